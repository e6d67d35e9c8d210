#!/bin/bash
mkdir -p gpurun_out/suite
python -m pytest tests -m gpu -q > gpurun_out/suite/out.txt 2>&1
echo "exit $?" >> gpurun_out/suite/out.txt
grep -v "amdgpu.ids\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" gpurun_out/suite/out.txt | tail -40 | cut -c1-300
