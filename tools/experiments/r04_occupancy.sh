#!/bin/bash
# T13 / T12 mixh at three waves per SIMD (<= 168 VGPRs: shallower prefetch or a
# forced occupancy) against the benched two (172-173 VGPRs): SQ counters say a
# wave issues VALU 54 % of its life at two waves per SIMD.
set -o pipefail
python tools/sweep.py --extent 8192 8192 --waves 1x1 --nt-load 1 --xcd 1 --launches 4 --rounds 3 --reps 8 \
  --fuse 13 12 --prefetch 2 3 4 --shift mixh --chunk 0 --occ 0 3 --out gpurun_out/r04_occupancy.json > gpurun_out/r04_occupancy.log 2>&1 || { tail -5 gpurun_out/r04_occupancy.log; exit 1; }
python - <<'PY'
import json
for r in sorted(json.load(open('gpurun_out/r04_occupancy.json')), key=lambda r: (r['fuse'], r['ms_min'])):
    print('T%d pf%d occ%d  %.1f us  %s' % (r['fuse'], r['prefetch'], r['occ'], r['ms_min'] * 1e3, r['kernel'][-28:]))
PY
