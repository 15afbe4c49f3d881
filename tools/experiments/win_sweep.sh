for prog in erosion.soda xcorr.soda; do
  python tools/corpus_bench.py --only $prog --no-windows --out gpurun_out/r03_windows_sweep.jsonl
  python tools/corpus_bench.py --only $prog --out gpurun_out/r03_windows_sweep.jsonl
  for v in 8 4 2; do for pf in 1 2 4; do
    python tools/corpus_bench.py --only $prog --vec $v --prefetch $pf --reg-budget 400 --out gpurun_out/r03_windows_sweep.jsonl
  done; done
done
