for ext in "8192 8192" "8192 1224" "8192 4296"; do
  tag=$(echo $ext | tr ' ' 'x')
  python tools/sweep.py --extent $ext --fuse 12 8 4 --chunk 0 --prefetch 2 4 --waves 1x1 --nt-load 1 --xcd 1 --shift dpp mixh --reps 20 --rounds 3 --launches 8 --out gpurun_out/r03_sweep_mixh_$tag.json > gpurun_out/sweep_mixh_$tag.log 2>&1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03_sweep_mixh_*.json')):
    for r in json.load(open(f)):
        print(f[26:-5], r['fuse'], r['prefetch'], r['shift'], round(r['ms_min']*1e3,1), round(r['ms_med']*1e3,1))
PY
