#!/bin/bash
# Four waves per SIMD (<= 128 VGPRs) with mixed shifts at T = 8..11 against the
# benched two waves at T = 12 / 13: waves issue in pairs, 2.0 cycles per
# instruction at 2 / 4 / 8 resident waves, 2.7 at 3.
set -o pipefail
python tools/sweep.py --extent 8192 8192 --waves 1x1 --nt-load 1 --xcd 1 --launches 4 --rounds 3 --reps 8 \
  --fuse 8 9 10 11 --prefetch 2 4 --shift mixh --chunk 0 --occ 0 4 --out gpurun_out/r04_occupancy4.json > gpurun_out/r04_occupancy4.log 2>&1 || { tail -5 gpurun_out/r04_occupancy4.log; exit 1; }
python - <<'PY'
import json
for r in sorted(json.load(open('gpurun_out/r04_occupancy4.json')), key=lambda r: r['ms_min'] / r['fuse']):
    print('T%-2d pf%d occ%d  %.1f us  %.2f us/iter  %s' % (r['fuse'], r['prefetch'], r['occ'], r['ms_min'] * 1e3, r['ms_min'] * 1e3 / r['fuse'], r['kernel'][-28:]))
PY
