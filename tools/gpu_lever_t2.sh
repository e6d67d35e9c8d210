#!/bin/bash
# diagnostic: the two-part split of the first chained layer's operand in edge_fwd (-DFE_EDGE_T2):
# kernel time on one box against the default build, and what it does to the parity tests
bash tools/gpu_ab.sh "base:" "t2:-DFE_EDGE_T2"
python -m pytest tests/test_gpu_parity.py "tests/test_gpu_properties.py::test_cfg4_headline_shape_vs_oracle" -q -m gpu 2>&1 | tail -15 | cut -c1-700
cd fastegnn_amd/csrc && rm -f layer_fwd.o && make -j8 ../libfastegnn_hip.so > /dev/null 2>&1
