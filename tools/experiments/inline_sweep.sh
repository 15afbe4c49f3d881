out=gpurun_out/r03_inline_sweep.jsonl
python tools/corpus_bench.py --only denoise3d.soda --no-inline --out $out
python tools/corpus_bench.py --only denoise3d.soda --out $out
for v in 4 2 1; do for r in 4 2; do
  timeout -k 10 300 python tools/corpus_bench.py --only denoise3d.soda --vec $v --tile-rows $r --reg-budget 100000 --out $out
done; done
python tools/corpus_bench.py --only denoise2d.soda --no-inline --out $out
python tools/corpus_bench.py --only denoise2d.soda --out $out
python tools/corpus_bench.py --only denoise2d.soda --vec 4 --prefetch 2 --out $out
python tools/corpus_bench.py --only sobel2d.soda --no-inline --out $out
python tools/corpus_bench.py --only sobel2d.soda --out $out
python tools/corpus_bench.py --only contrast.soda --out $out
python tools/corpus_bench.py --only contrast.soda --vec 2 --prefetch 2 --reg-budget 100000 --out $out
