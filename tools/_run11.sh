mkdir -p gpurun_out/r02
SW="python tools/sweep.py --waves 1x1 --xcd 1 --rounds 3 --reps 5 --chunk 0"
$SW --soda tests/golden/soda/heat3d.soda --extent 512 512 512 --fuse 2 --prefetch 1 --tile-rows 4 6 8 10 12 --nt-load 0 --launches 4 --reg-budget 460 --out gpurun_out/r02/sweep_heat_rows.json > gpurun_out/r02/sweep_heat_rows.log 2>&1
$SW --soda tests/golden/soda/blur.soda --extent 16384 16384 --fuse 1 --vec 8 --prefetch 2 4 8 --nt-store 1 --nt-load 0 1 --out gpurun_out/r02/sweep_blur.json > gpurun_out/r02/sweep_blur.log 2>&1
python tools/sweep.py --waves 1x1 --xcd 1 --rounds 3 --reps 5 --chunk 24 32 48 66 96 128 --soda tests/golden/soda/blur.soda --extent 16384 16384 --fuse 1 --vec 8 --prefetch 4 --nt-store 1 --nt-load 0 --out gpurun_out/r02/sweep_blur_chunk.json > gpurun_out/r02/sweep_blur_chunk.log 2>&1
for f in heat_rows blur blur_chunk; do echo "== $f"; python - <<PY
import json
for r in json.load(open('gpurun_out/r02/sweep_$f.json')):
  print(r['fuse'], 'pf', r['prefetch'], 'rows', r['tile_rows'], 'chunk', r['chunk'], 'ntl', r['nt_load'], 'us %.1f' % (r['ms_min']*1e3), 'med %.1f' % (r['ms_med']*1e3), 'GB/s %.0f' % r['GBs'], r['kernel'][-30:])
PY
grep -h "^skip" gpurun_out/r02/sweep_$f.log | cut -c1-200
done
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "scheduler or calibrated" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single-iter 2>/dev/null | cut -c1-900
