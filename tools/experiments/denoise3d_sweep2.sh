out=gpurun_out/r03_denoise3d_sweep.jsonl
for v in 1 2 4; do
  SODA_HIP_XSHARE=1 timeout -k 10 300 python tools/corpus_bench.py --only denoise3d.soda --vec $v --reg-budget 100000 --out $out
done
SODA_HIP_XSHARE=1 timeout -k 10 300 python tools/corpus_bench.py --only heat3d.soda jacobi3d.soda --out $out
python -m pytest tests/test_hip_parity.py -q -x -k "corpus_3d or rows_shared" 2>&1 | tail -3
SODA_HIP_XSHARE=1 python -m pytest tests/test_hip_parity.py -q -x -k "corpus_3d" 2>&1 | tail -3
