out=gpurun_out/r03_xwindow_sweep.jsonl
for prog in erosion.soda xcorr.soda; do
  python tools/corpus_bench.py --only $prog --out $out
  for v in 8 4; do for pf in 1 2; do
    python tools/corpus_bench.py --only $prog --vec $v --prefetch $pf --reg-budget 100000 --out $out
  done; done
done
python -m pytest tests/test_hip_parity.py -x -q -k "corpus_2d or hand_written or fuzz or golden" 2>&1 | tail -3
python -m pytest tests/test_fuzz.py -x -q -m gpu 2>&1 | tail -3
