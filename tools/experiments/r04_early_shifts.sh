#!/bin/bash
# Round 4 experiment: all of a row step's ds_swizzle shifts issued at the head
# of the step behind a scheduling barrier (13 in flight) against the default
# (one stage ahead, which the compiler sinks to one at a time).
set -o pipefail
out=gpurun_out
for mode in stage tickbar stage tickbar; do
  SODA_HIP_EARLY_SHIFTS=$mode python tools/sweep.py --extent 8192 8192 --waves 1x1 --nt-load 1 --xcd 1 --launches 4 --rounds 3 --reps 8 \
    --fuse 13 12 8 --prefetch 4 --shift mixh --chunk 0 --out $out/r04_early_$mode.json 2>&1 | grep -o '"fuse": [0-9]*\|"ms_min": [0-9.]*\|"kernel": "[^"]*"' | paste - - - | sed "s/^/$mode /"
done
SODA_HIP_EARLY_SHIFTS=tickbar python -m pytest tests/test_hip_parity.py -m gpu -x -q -p no:cacheprovider -k "temporal_blocking or every_kernel_of_the_benched or awkward or explicit_chunk" 2>&1 | tail -3
