out=gpurun_out/r03_denoise3d_sweep.jsonl
for v in 1 2; do for r in 6 8 12; do for pf in 1 2; do
  timeout -k 10 300 python tools/corpus_bench.py --only denoise3d.soda --vec $v --tile-rows $r --prefetch $pf --reg-budget 100000 --out $out
done; done; done
