#!/bin/bash
# Round 4, VERDICT item 6: contrast through the LDS row ring (ldswin) against
# the 7-stage marching form, chunk lengths swept.  -> gpurun_out/r04_contrast.jsonl
set -o pipefail
out=gpurun_out/r04_contrast.jsonl
: > $out
for chunk in 32 64 128 171 256; do
  python tools/corpus_bench.py --only contrast.soda --strategy ldswin --chunk $chunk --reps 10 | sed "s/^{/{\"chunk\": $chunk, /" >> $out || exit 1
done
python tools/corpus_bench.py --only contrast.soda --strategy march --reps 10 >> $out || exit 1
python tools/corpus_bench.py --only contrast.soda --reps 10 >> $out || exit 1
cat $out
