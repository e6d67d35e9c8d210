#!/bin/bash
# diagnostic: rebuild the library with in-kernel phase stamps (on the GPU box's scratch copy) and print phase shares
cd fastegnn_amd/csrc && rm -f *.o && make -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DFE_STAMP $EXTRA" > /dev/null 2>&1 && cd ../.. && python tools/gpu_stamp.py
