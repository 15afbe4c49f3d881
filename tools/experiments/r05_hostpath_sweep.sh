#!/bin/bash
# host-array entry: worker threads x staging chunk size (tools/hostpath.py)
out=gpurun_out/r05_hostpath_sweep.jsonl
: > $out
for th in 4 8 16; do
  for mb in 8 16 32; do
    SODA_HIP_HOST_THREADS=$th SODA_HIP_HOST_CHUNK_MB=$mb python tools/hostpath.py --only c2 >> $out 2>/dev/null
    SODA_HIP_HOST_THREADS=$th SODA_HIP_HOST_CHUNK_MB=$mb python tools/hostpath.py --only c3 >> $out 2>/dev/null
  done
done
SODA_HIP_HOST_BANDS=0 python tools/hostpath.py --only c2 >> $out 2>/dev/null
python - <<'PY'
import json
for l in open('gpurun_out/r05_hostpath_sweep.jsonl'):
  r = json.loads(l)
  print(r['workload'][:12], 'threads', r['host_threads'], 'chunk', r['chunk_MiB'], 'reused %.2f fresh %.2f' % (r['reused_ms'], r['fresh_ms']))
PY
