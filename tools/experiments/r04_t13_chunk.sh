#!/bin/bash
# Does the library's chunk length (168 rows: 1960 waves, one round at two waves
# per SIMD) leave anything on the table for the benched T13 / T12 passes on
# 8192^2?  Sustained (4 launches per call).  -> gpurun_out/r04_t13_chunk.json
set -o pipefail
python tools/sweep.py --extent 8192 8192 --waves 1x1 --nt-load 1 --xcd 1 --launches 4 --rounds 3 --reps 6 \
  --fuse 13 12 --prefetch 4 --shift mixh --chunk 0 120 137 147 158 171 186 205 228 256 --out gpurun_out/r04_t13_chunk.json > gpurun_out/r04_t13_chunk.log 2>&1 || exit 1
python - <<'PY'
import json
rows = json.load(open('gpurun_out/r04_t13_chunk.json'))
for r in sorted(rows, key=lambda r: (r['fuse'], r['ms_min'])):
    print('T%d chunk %3d  %.1f us  (median %.1f)' % (r['fuse'], r['chunk'], r['ms_min'] * 1e3, r['ms_med'] * 1e3))
PY
