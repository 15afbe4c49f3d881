out=gpurun_out/r03_slide_sweep.jsonl
python tools/corpus_bench.py --only xcorr.soda erosion.soda --out $out
for v in 8 4; do for pf in 1 2 4; do
  python tools/corpus_bench.py --only xcorr.soda --vec $v --prefetch $pf --reg-budget 100000 --out $out
done; done
python -m pytest tests/test_hip_parity.py -x -q -k "corpus_2d or hand_written or golden or language or awkward" 2>&1 | tail -3
python -m pytest tests/test_fuzz.py -x -q -m gpu 2>&1 | tail -3
