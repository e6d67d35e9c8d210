#!/bin/bash
# diagnostic: rebuild layer_fwd.hip with in-kernel phase stamps of virt_fwd_kernel (on the GPU box's scratch copy) and print the shares
cd fastegnn_amd/csrc && rm -f layer_fwd.o && make -j8 ../libfastegnn_hip.so EXTRA="-DFE_STAMP_VF $EXTRA" > /dev/null 2>&1 && cd ../.. && python tools/gpu_stamp_vf.py
