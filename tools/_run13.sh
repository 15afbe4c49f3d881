mkdir -p gpurun_out/r02
SW="python tools/sweep.py --waves 1x1 --xcd 1 --rounds 4 --reps 5 --chunk 0"
$SW --edge 0 1 --soda tests/golden/soda/blur.soda --extent 16384 16384 --fuse 1 --vec 8 --prefetch 4 8 --nt-store 1 --nt-load 0 --out gpurun_out/r02/sweep_blur_edge.json > gpurun_out/r02/sweep_blur_edge.log 2>&1
$SW --edge 0 1 --fuse 1 --prefetch 8 --nt-load 1 --launches 8 --out gpurun_out/r02/sweep_j1_edge.json > gpurun_out/r02/sweep_j1_edge.log 2>&1
$SW --edge 0 1 --soda tests/golden/soda/heat3d.soda --extent 512 512 512 --fuse 1 --prefetch 1 --tile-rows 4 --nt-load 0 --launches 4 --out gpurun_out/r02/sweep_h1_edge.json > gpurun_out/r02/sweep_h1_edge.log 2>&1
for f in blur_edge j1_edge h1_edge; do echo "== $f"; python - <<PY
import json
for r in json.load(open('gpurun_out/r02/sweep_$f.json')):
  print(r['fuse'], 'pf', r['prefetch'], 'edge', r['edge'], 'us %.1f' % (r['ms_min']*1e3), 'med %.1f' % (r['ms_med']*1e3), 'GB/s %.0f' % r['GBs'])
PY
grep -h "^skip" gpurun_out/r02/sweep_$f.log | cut -c1-200
done
