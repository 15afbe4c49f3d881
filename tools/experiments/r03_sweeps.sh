# round 3: f3 microbenchmark, fused-kernel shift/pipe sweep, blur chunk sweep
python tools/mfmabench.py --out gpurun_out/r03_f3_mfma.json > gpurun_out/mfma.log 2>&1
python tools/sweep.py --fuse 12 --chunk 0 --prefetch 2 --waves 1x1 --nt-load 1 --xcd 1 --shift dpp swzh mixh --pipe 1 4 --pipe-rows 2 --reps 20 --rounds 3 --launches 8 --out gpurun_out/r03_sweep_t12_shift.json > gpurun_out/sweep_t12.log 2>&1
python tools/sweep.py --soda tests/golden/soda/blur.soda --extent 16384 16384 --fuse 1 --chunk 0 48 56 64 72 80 96 128 --prefetch 8 --vec 8 --waves 1x1 --nt-store 1 --nt-load 0 --xcd 1 --reps 10 --rounds 3 --out gpurun_out/r03_sweep_blur_chunk.json > gpurun_out/sweep_blur.log 2>&1
python tools/sweep.py --soda tests/golden/soda/blur.soda --extent 16384 16384 --fuse 1 --chunk 0 --prefetch 2 4 8 12 --vec 8 --waves 1x1 --nt-store 1 --nt-load 0 1 --xcd 0 1 --reps 10 --rounds 3 --out gpurun_out/r03_sweep_blur_pf.json > gpurun_out/sweep_blur2.log 2>&1
tail -22 gpurun_out/mfma.log
cut -c1-200 gpurun_out/sweep_t12.log
python - <<'PY'
import json
for f in ('gpurun_out/r03_sweep_blur_chunk.json','gpurun_out/r03_sweep_blur_pf.json'):
    for r in json.load(open(f)):
        print(f[-14:], r['chunk'], r['prefetch'], r['nt_load'], r['xcd'], round(r['ms_min']*1e3,1), round(r['ms_med']*1e3,1))
PY
