#!/bin/bash
# First contact with a real multi-GPU node.  No session of
# rounds 1-5 ever saw more than one GPU, so three calls have never executed:
# ncclSend / ncclRecv between two devices (batch_isend_irecv over RCCL, world >
# 1), hipMemcpyPeerAsync across devices (soda_hip_group_*), and all_gather on
# the nccl backend (bench.py's result check).  This script walks them in order
# of increasing dependence, stops at the first failure, and leaves one file a
# human (or the next round) can read: gpurun_out/first_multi_gpu.jsonl, one
# JSON object per step {step, cmd, rc, seconds, line: <bench JSON or null>,
# parity: <mismatches or null>, rccl_world}.  Every bench step checks its
# result against the CPU oracle (bench.py `parity`) -- a wrong exchange fails
# the step, not just the number.
#
#   tools/first_multi_gpu.sh [max GPUs, default: all visible]
#
# Run from the repository root on the node; ~6 minutes for 8 GPUs.
set -o pipefail
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0        # dmabuf IPC only on this pool
out=gpurun_out/first_multi_gpu.jsonl
mkdir -p gpurun_out
: > "$out"
have=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
# SODA_FIRST_CONTACT_REHEARSAL=1: the script's own plumbing on a ONE-GPU box --
# two ranks on device 0 over gloo (bench.py's documented rehearsal switches),
# virtual slabs for the group steps; proves nothing about RCCL or xGMI
rehearsal=${SODA_FIRST_CONTACT_REHEARSAL:-}
virtual=
if [ -n "$rehearsal" ]; then
  export SODA_BENCH_ONE_GPU=1 SODA_BENCH_BACKEND=gloo
  have=2
  virtual=--virtual
fi
max=${1:-$have}
[ "$max" -gt "$have" ] && max=$have
echo "first_multi_gpu: $have GPUs visible, using up to $max${rehearsal:+ (REHEARSAL on one GPU)}"
if [ "$have" -lt 2 ]; then
  echo '{"step": "probe", "rc": 2, "note": "fewer than 2 GPUs visible: nothing to do"}' >> "$out"
  exit 2
fi

step() {   # name, command...
  local name=$1; shift
  local log=gpurun_out/first_multi_gpu_${name}.log
  local t0=$(date +%s.%N)
  "$@" > "$log" 2>&1
  local rc=$?
  local t1=$(date +%s.%N)
  python - "$name" "$rc" "$t0" "$t1" "$log" "$out" "$*" <<'PY'
import json, sys
name, rc, t0, t1, log, out, cmd = sys.argv[1:8]
line = None
for l in open(log, errors='replace'):
  if l.startswith('{'):
    try:
      line = json.loads(l)
    except ValueError:
      pass
rec = {'step': name, 'cmd': cmd, 'rc': int(rc),
       'seconds': round(float(t1) - float(t0), 1), 'line': line,
       'parity': (line or {}).get('parity', {}).get('mismatches')
                 if line else None,
       'rccl_world': (line or {}).get('rccl_world') if line else None,
       'value': (line or {}).get('value') if line else None,
       'ms_per_step': (line or {}).get('ms_per_step') if line else None}
if int(rc) != 0:
  rec['log_tail'] = open(log, errors='replace').read()[-1500:]
with open(out, 'a') as f:
  f.write(json.dumps(rec) + '\n')
print('%-34s rc=%s  %ss  value=%s  ms/step=%s  parity=%s' % (
    name, rc, rec['seconds'], rec['value'], rec['ms_per_step'], rec['parity']))
PY
  return $rc
}

bench() {   # N, bench.py arguments...: one rank per GPU under torchrun
  local n=$1; shift
  if [ -n "$rehearsal" ]; then     # (bench.py's own launcher counts devices)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $n \
        --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
        bench.py --gpus $n "$@"
  else
    python bench.py --gpus $n "$@"
  fi
}

quick="--steps 20 --warmup 5 --no-cpu-baseline --no-rehearsal --no-single-iter --no-other-configs"
# 0. the single-GPU line on this node, for the ratio
step n1 python bench.py $quick || exit 1
# 1. two ranks over RCCL, exchange NOT hidden: ncclSend/ncclRecv + the nccl
#    all_gather of the result check, nothing else new
step n2_serial bench 2 --overlap off $quick || exit 1
# 2. the same with the exchange under the interior kernels (StreamOverlap)
step n2_overlap bench 2 --overlap on $quick || exit 1
# 3. the bench's own choice between the two (what the driver runs)
step n2_auto bench 2 $quick || exit 1
# 4. one process, two GPUs: hipMemcpyPeerAsync between devices
step n2_group python bench.py --gpus 2 --group $virtual $quick || exit 1
# 5. wider
for n in 4 8; do
  [ "$n" -le "$max" ] || continue
  step n${n}_auto bench $n $quick || exit 1
  step n${n}_group python bench.py --gpus $n --group $virtual $quick || exit 1
done
n=$max
# 6. the other multi-GPU configs of BASELINE.json: C4 heat3d 512^3 x 50 (a halo
#    exchange every K < 50 iterations: the exchange path under load) and C5
#    jacobi2d x 1000 (several exchanges per step)
step c4_n${n} bench $n --soda tests/golden/soda/heat3d.soda \
    --extent 512 512 512 --iterate 50 --fuse 2 $quick || exit 1
step c5_n${n} bench $n --iterate 1000 --steps 5 --warmup 2 \
    --no-cpu-baseline --no-rehearsal --no-single-iter --no-other-configs || exit 1
# 7. weak scaling: every rank keeps the single-GPU grid
step weak_n${n} bench $n --scaling weak $quick || exit 1
if [ -n "$rehearsal" ]; then echo "first_multi_gpu: rehearsal done, see $out"; exit 0; fi
# 8. what the exchange looks like on the wire: a kernel trace of two ranks,
#    one step, exchange hidden (rank 0's trace: the send/recv kernels of RCCL
#    against the interior kernels; tools/overlap_report.py reads it)
step trace_n2 rocprofv3 --kernel-trace --output-format csv \
    -d gpurun_out/first_multi_gpu_trace -- python3 -m torch.distributed.run \
    --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 \
    bench.py --gpus 2 --overlap on --steps 3 --warmup 1 --no-cpu-baseline \
    --no-rehearsal --no-single-iter --no-other-configs --no-parity
python tools/overlap_report.py gpurun_out/first_multi_gpu_trace \
    gpurun_out/first_multi_gpu_overlap.json > /dev/null 2>&1
# 9. the GPU tests that need peers (skipped on one GPU)
step tests python -m pytest tests/test_dist.py tests/test_group.py -m gpu -q -x
echo "first_multi_gpu: done, see $out"
python - "$out" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
base = next((r['value'] for r in rows if r['step'] == 'n1' and r['value']), None)
for r in rows:
  if r.get('value') and base:
    n = (r.get('line') or {}).get('n_gpus')
    print('%-14s n=%s  %.3g cells*iters/s  x%.2f of one GPU' % (
        r['step'], n, r['value'], r['value'] / base))
PY
