#!/bin/bash
# diagnostic: rebuild given sources with extra -D flags and run one pytest selection
files=$1; sel=$2; shift 2
for extra in "$@"; do
  ( cd fastegnn_amd/csrc && for f in $files; do rm -f $f.o; done && make -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include $extra" > /dev/null 2>&1 ) || { echo "build failed: $extra"; continue; }
  echo "[$extra]"; python -m pytest $sel -x -q -m gpu 2>&1 | grep -E "^E  +Assert|passed|failed" | head -4
done
