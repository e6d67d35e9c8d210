#!/bin/bash
# diagnostic: A/B of build variants on ONE box (boxes differ by several percent): for every "TAG:flags" argument rebuild the
# library with those -D flags and print the per-kernel table of a short bench run.
# usage: bash tools/gpu_ab.sh "base:" "noflush:-DFE_PC_FLUSH=1000000" ...
for arg in "$@"; do
  TAG="${arg%%:*}" EXTRA="${arg#*:}" bash tools/gpu_variant_bench.sh
done
