#!/bin/bash
# contrast through ldswin: rows per step (= waves per block) against LDS-limited
# residency: S4 51 KB -> 3 blocks = 3 waves per SIMD; S8 68 KB -> 2 blocks = 4
# waves per SIMD; S2, S6.  -> gpurun_out/r04_contrast2.jsonl
set -o pipefail
out=gpurun_out/r04_contrast2.jsonl
: > $out
for wy in 1 8 2 6; do
  for chunk in 32 64 128; do
    python tools/corpus_bench.py --only contrast.soda --strategy ldswin --chunk $chunk --waves-y $wy --reps 10 | sed "s/^{/{\"chunk\": $chunk, \"waves_y\": $wy, /" >> $out || exit 1
  done
done
cat $out | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['waves_y'], r['chunk'], r['kernels'][0], r['us_per_iteration'])"
