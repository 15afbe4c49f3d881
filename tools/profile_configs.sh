#!/bin/bash
# Kernel durations and PMC traffic of the other BASELINE configs (heat3d 512^3
# T=1 and T=2, blur 16384^2, contrast 8192^2) -> gpurun_out/<tag>_cfg_traffic.json
set -o pipefail
tag=${1:-r01}
out=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {  # name, args...
  name=$1; shift
  # JIT and peel selection first, OUTSIDE the profiler: kernels compiled by
  # hiprtc inside a rocprofv3-run process came out different once (heat3d T=2:
  # 253 instead of 260 VGPRs, another peel choice), and what is profiled must
  # be what production runs
  python3 tools/run_program.py "$@" > $out/${tag}_cfg_${name}.warm.log 2>&1 || return 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_cfg_${name}_trace -- python3 tools/run_program.py "$@" > $out/${tag}_cfg_${name}.log 2>&1 || return 1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_cfg_${name}_fetch -- python3 tools/run_program.py "$@" >> $out/${tag}_cfg_${name}.log 2>&1 || return 1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_cfg_${name}_write -- python3 tools/run_program.py "$@" >> $out/${tag}_cfg_${name}.log 2>&1 || return 1
  python tools/pmc_summary.py $out/${tag}_cfg_${name}_trace $out/${tag}_cfg_${name}_fetch $out/${tag}_cfg_${name}_write $out/${tag}_cfg_${name}_traffic.json > /dev/null
}
run heat3d_t1 heat3d.soda 512 512 512 --iterate 20 --reps 3 || exit 1
run heat3d_t2 heat3d.soda 512 512 512 --iterate 20 --fuse 2 --reps 3 || exit 1
run blur blur.soda 16384 16384 --reps 10 || exit 1
run contrast contrast.soda 8192 8192 --reps 5 || exit 1
run denoise2d denoise2d.soda 8192 8192 --reps 5 || exit 1
run erosion erosion.soda 8192 8192 --reps 10 || exit 1
run xcorr xcorr.soda 8192 8192 --reps 10 || exit 1
run denoise3d denoise3d.soda 512 512 512 --reps 3 || exit 1
python3 - <<PY
import json, glob
merged = {}
for f in sorted(glob.glob('$out/${tag}_cfg_*_traffic.json')):
    merged.update(json.load(open(f)))
json.dump(merged, open('$out/${tag}_cfg_traffic.json', 'w'), indent=1, sort_keys=True)
for k, v in merged.items():
    if 'copy' in k: continue
    print(k[:60].ljust(62), '%.1f us' % v['avg_us'], 'fetch %.0f MB' % (v.get('fetch_bytes_per_launch', 0) / 1e6), 'write %.0f MB' % (v.get('write_bytes_per_launch', 0) / 1e6))
PY
