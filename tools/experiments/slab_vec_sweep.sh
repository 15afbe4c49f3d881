#!/bin/bash
# Slab extents of an 8-GPU run of jacobi2d 8192^2: 4 / 2 cells per lane, one wave or four pipelined waves.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
out=gpurun_out
for ext in 1224 1048; do
  python tools/sweep.py --extent 8192 $ext --fuse 12 --chunk 0 --prefetch 4 --vec 4 2 --shift mixh dpp --waves 1x1 --pipe 1 4 \
    --nt-load 1 --xcd 1 --tile-rows 4 --rounds 3 --reps 30 --out $out/r03_slabvec_${ext}.json > $out/r03_slabvec_${ext}.log 2>&1 || exit 1
done
cut -c1-20,150-175,330-700 $out/r03_slabvec_*.log
