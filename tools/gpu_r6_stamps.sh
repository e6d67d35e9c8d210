#!/bin/bash
# round 6: rebuild the default library with in-kernel phase stamps (-DFE_STAMP; on the GPU box's scratch copy) and print the phase table of
# the two backward kernels' producers.  usage: bash tools/gpu_r6_stamps.sh [extra -D flags]
( cd fastegnn_amd/csrc && rm -f layer_bwd.o virt_bwd.o && make -j32 ../libfastegnn_hip.so EXTRA="-DFE_STAMP $1" > /dev/null 2>&1 ) || { echo "stamp build failed"; exit 1; }
python tools/gpu_stamp_vbs.py
( cd fastegnn_amd/csrc && rm -f layer_bwd.o virt_bwd.o && make -j32 ../libfastegnn_hip.so > /dev/null 2>&1 )
