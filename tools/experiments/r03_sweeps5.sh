python tools/sweep.py --soda tests/golden/soda/heat3d.soda --extent 512 512 512 --strategy march --fuse 2 3 4 --chunk 0 --prefetch 1 --vec 2 4 --tile-rows 2 4 --waves 1x1 --xcd 1 --reg-budget 1000000 --reps 5 --rounds 2 --launches 4 --out gpurun_out/r03_sweep_heat_deep.json > gpurun_out/sweep_heat_deep.log 2>&1
python tools/sweep.py --soda tests/golden/soda/heat3d.soda --extent 512 512 512 --strategy march --fuse 4 --chunk 0 --prefetch 1 --vec 4 --tile-rows 2 4 --pipe 2 --waves 1x1 --xcd 1 --reg-budget 1000000 --reps 5 --rounds 2 --launches 4 --out gpurun_out/r03_sweep_heat_deep_pipe.json >> gpurun_out/sweep_heat_deep.log 2>&1
python - <<'PY'
import json
for f in ('gpurun_out/r03_sweep_heat_deep.json','gpurun_out/r03_sweep_heat_deep_pipe.json'):
    for r in json.load(open(f)):
        print(r['fuse'], r['vec'], r['tile_rows'], r['pipe'], round(r['ms_min']*1e3,1), round(r['ms_min']*1e3/r['fuse'],1), r['kernel'][-28:])
PY
grep skip gpurun_out/sweep_heat_deep.log | cut -c1-160
