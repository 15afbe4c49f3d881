#!/bin/bash
# Round 4, item 4, second pass on the 8192 x 1224 slab: chunk length against
# residency (one wave per SIMD with 50-row chunks? three with prefetch 2?),
# and 2 cells per lane with mixed shifts.  -> gpurun_out/r04_slab2_*.json
set -o pipefail
out=gpurun_out
common="--extent 8192 1224 --waves 1x1 --nt-load 1 --xcd 1 --launches 8 --rounds 3 --reps 6"
python tools/sweep.py $common --fuse 13 12 --prefetch 2 4 --shift mixh --chunk 0 17 20 25 33 50 --out $out/r04_slab2_chunk.json > $out/r04_slab2_chunk.log 2>&1 || exit 1
python tools/sweep.py $common --fuse 13 12 8 --prefetch 2 4 --shift mixh --vec 2 --chunk 0 --out $out/r04_slab2_v2mixh.json > $out/r04_slab2_v2mixh.log 2>&1 || exit 1
python tools/sweep.py $common --fuse 10 9 --prefetch 2 4 --shift mixh dpp --chunk 0 --out $out/r04_slab2_t10.json > $out/r04_slab2_t10.log 2>&1 || exit 1
echo done
