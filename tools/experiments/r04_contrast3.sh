#!/bin/bash
# contrast through ldswin: literal operands (8-byte encodings) against constants
# held in registers (4-byte encodings): is the step bound by instruction fetch?
set -o pipefail
out=gpurun_out/r04_contrast3.jsonl
: > $out
for sg in 1 0; do
  for wy in 1 8; do
    for chunk in 64 128; do
      SODA_HIP_LDSWIN_SGPR=$sg python tools/corpus_bench.py --only contrast.soda --strategy ldswin --chunk $chunk --waves-y $wy --reps 10 | sed "s/^{/{\"registers\": $sg, \"chunk\": $chunk, \"waves_y\": $wy, /" >> $out || exit 1
    done
  done
done
cat $out | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('const-in-registers', r['registers'], 'waves_y', r['waves_y'], 'chunk', r['chunk'], r['vgprs'], r['us_per_iteration'])"
