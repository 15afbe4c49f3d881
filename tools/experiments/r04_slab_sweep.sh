#!/bin/bash
# Round 4, VERDICT item 4: the middle slab of the 8-GPU C2 run (8192 x 1224)
# and of the 4-GPU run (8192 x 2248) under every lever the generator already
# has -- pipelined blocks (the warm-up is paid per block, chunks W times
# longer), narrower strips (V = 2), shallower fusion, both shift flavours.
# Sustained: 8 launches per call, arrays rotating.  -> gpurun_out/r04_slab_*.json
set -o pipefail
out=gpurun_out
common="--waves 1x1 --nt-load 1 --xcd 1 --chunk 0 --launches 8 --rounds 3 --reps 6"
for ext in "8192 1224" "8192 2248"; do
  tag=$(echo $ext | tr ' ' 'x')
  python tools/sweep.py --extent $ext $common --fuse 13 12 8 --prefetch 4 --shift mixh dpp --pipe 1 --out $out/r04_slab_${tag}_base.json > $out/r04_slab_${tag}_base.log 2>&1 || exit 1
  python tools/sweep.py --extent $ext $common --fuse 12 8 --prefetch 2 --shift mixh dpp --pipe 2 4 --out $out/r04_slab_${tag}_pipe.json > $out/r04_slab_${tag}_pipe.log 2>&1 || exit 1
  python tools/sweep.py --extent $ext $common --fuse 12 8 --prefetch 4 --shift dpp --vec 2 --pipe 1 --out $out/r04_slab_${tag}_v2.json > $out/r04_slab_${tag}_v2.log 2>&1 || exit 1
  echo "done $tag"
done
