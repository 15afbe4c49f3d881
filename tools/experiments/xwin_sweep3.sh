python tools/sweep.py --soda tests/golden/soda/xcorr.soda --extent 8192 8192 --fuse 1 --chunk 0 32 64 128 256 --prefetch 2 --vec 8 --waves 1x1 --nt-store 1 --nt-load 0 --xcd 1 --reg-budget 100000 --reps 10 --rounds 2 --out gpurun_out/r03_sweep_xcorr_slide_chunk.json > gpurun_out/sweep_xc.log 2>&1
SODA_HIP_SLIDE=0 python tools/sweep.py --soda tests/golden/soda/xcorr.soda --extent 8192 8192 --fuse 1 --chunk 0 64 256 --prefetch 2 --vec 8 --waves 1x1 --nt-store 1 --nt-load 0 --xcd 1 --reg-budget 100000 --reps 10 --rounds 2 --out gpurun_out/r03_sweep_xcorr_chain_chunk.json >> gpurun_out/sweep_xc.log 2>&1
python - <<'PY'
import json
for f in ('gpurun_out/r03_sweep_xcorr_slide_chunk.json','gpurun_out/r03_sweep_xcorr_chain_chunk.json'):
    for r in json.load(open(f)):
        print(f[24:-5], r['chunk'], round(r['ms_min']*1e3,1), r['kernel'][-20:])
PY
