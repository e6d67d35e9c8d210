#!/bin/bash
# diagnostic: rebuild the backward kernels with in-kernel phase stamps (on the GPU box's scratch copy) and print the phase shares (default: the producers of edge_bwd; STAMP_SCRIPT= selects another)
cd fastegnn_amd/csrc && rm -f layer_bwd.o layer_fwd.o && make -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DFE_STAMP $EXTRA" > /dev/null 2>&1 && cd ../.. && python ${STAMP_SCRIPT:-tools/gpu_stamp_eb.py}
