#!/bin/bash
bash tools/gpu_r4_final.sh r04c
bash tools/gpu_r4_sharded_final.sh r04c
