#!/bin/bash
# bench.py's two-rank path on the one GPU (gloo, point-to-point staged through
# host buffers), 12 runs per exchange mode: every run checks its result against
# the oracle (exit code 3 on a mismatch).  -> gpurun_out/r04_rehearsal_loop.txt
out=gpurun_out/r04_rehearsal_loop.txt
: > $out
export SODA_BENCH_ONE_GPU=1 SODA_BENCH_BACKEND=gloo
port=29600
for mode in on off auto; do
  ok=0; bad=0
  for i in $(seq 1 12); do
    port=$((port + 1))
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port \
      bench.py --gpus 2 --steps 3 --warmup 1 --extent 2048 2048 --iterate 48 --fuse 12 4 --exchange-every 24 \
      --overlap $mode --no-cpu-baseline --no-single-iter --clock-warm-seconds 0.05 > /tmp/rl.out 2> /tmp/rl.err
    rc=$?
    if [ $rc -eq 0 ] && grep -q '"mismatches": 0' /tmp/rl.out; then ok=$((ok + 1)); else bad=$((bad + 1)); tail -3 /tmp/rl.err >> $out; fi
  done
  echo "overlap $mode: $ok runs checked clean, $bad not" | tee -a $out
done
