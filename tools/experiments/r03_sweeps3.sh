python tools/sweep.py --soda tests/golden/soda/blur.soda --extent 16384 16384 --fuse 1 --chunk 0 --prefetch 2 4 6 --vec 8 16 --waves 1x1 --nt-store 1 --nt-load 0 --xcd 1 --reg-budget 100000 --reps 10 --rounds 3 --out gpurun_out/r03_sweep_blur_vec.json > gpurun_out/sweep_blur3.log 2>&1
python - <<'PY'
import json
for r in json.load(open('gpurun_out/r03_sweep_blur_vec.json')):
    print(r['vec'], r['prefetch'], round(r['ms_min']*1e3,1), round(r['ms_med']*1e3,1), r['kernel'])
PY
python -m pytest tests/test_hip_parity.py -x -q -k "temporal or explicit_chunk or model_schedule or calibrated or border or full_size" 2>&1 | tail -4
python bench.py --no-cpu-baseline > gpurun_out/r03_bench_c.json 2> gpurun_out/r03_bench_c.err; python -c "
import json; d=json.loads(open('gpurun_out/r03_bench_c.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config']['schedule_per_exchange_interval'], d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('rehearsed_scaling'))"
