for ext in "8192 8192" "8192 1224"; do
  tag=$(echo $ext | tr ' ' 'x')
  python tools/sweep.py --extent $ext --fuse 12 13 14 16 --chunk 0 --prefetch 2 4 --waves 1x1 --nt-load 1 --xcd 1 --shift mixh --reps 20 --rounds 3 --launches 8 --out gpurun_out/r03_sweep_mixh_deep_$tag.json > gpurun_out/sweep_mixh_deep_$tag.log 2>&1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03_sweep_mixh_deep_*.json')):
    for r in json.load(open(f)):
        print(f[31:-5], r['fuse'], r['prefetch'], round(r['ms_min']*1e3,1), round(r['ms_med']*1e3,1), round(r['ms_min']*1e3/r['fuse'],2), r['kernel'][-12:])
PY
grep -h skip gpurun_out/sweep_mixh_deep_*.log | cut -c1-200
