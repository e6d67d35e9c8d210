#!/bin/bash
# diagnostic: rebuild virt_bwd.hip with in-kernel phase stamps (on the GPU box's scratch copy) and print the phase shares
cd fastegnn_amd/csrc && rm -f virt_bwd.o && make -j8 ../libfastegnn_hip.so EXTRA="-DFE_STAMP $EXTRA" > /dev/null 2>&1 && cd ../.. && python tools/gpu_stamp_vb2.py
