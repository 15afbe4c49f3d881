mkdir -p gpurun_out/r02
SW="python tools/sweep.py --waves 1x1 --xcd 1 --rounds 3 --reps 5"
$SW --chunk 11 16 21 22 32 33 43 64 --soda tests/golden/soda/blur.soda --extent 16384 16384 --fuse 1 --vec 8 --prefetch 4 --nt-store 1 --nt-load 0 --out gpurun_out/r02/sweep_blur_chunk2.json > gpurun_out/r02/sweep_blur_chunk2.log 2>&1
$SW --chunk 8 11 16 22 32 43 64 --fuse 1 --prefetch 8 --nt-load 1 --launches 8 --out gpurun_out/r02/sweep_j1_chunk.json > gpurun_out/r02/sweep_j1_chunk.log 2>&1
$SW --chunk 16 22 32 43 64 --fuse 4 --prefetch 4 --nt-load 1 --launches 8 --out gpurun_out/r02/sweep_j4_chunk.json > gpurun_out/r02/sweep_j4_chunk.log 2>&1
$SW --chunk 8 11 16 22 32 64 --soda tests/golden/soda/heat3d.soda --extent 512 512 512 --fuse 1 --prefetch 1 --tile-rows 4 --nt-load 0 --launches 4 --out gpurun_out/r02/sweep_h1_chunk.json > gpurun_out/r02/sweep_h1_chunk.log 2>&1
$SW --chunk 8 11 16 22 32 64 --soda tests/golden/soda/heat3d.soda --extent 512 512 512 --fuse 2 --prefetch 1 --tile-rows 4 --nt-load 0 --launches 4 --out gpurun_out/r02/sweep_h2_chunk.json > gpurun_out/r02/sweep_h2_chunk.log 2>&1
for f in blur_chunk2 j1_chunk j4_chunk h1_chunk h2_chunk; do echo "== $f"; python - <<PY
import json
for r in json.load(open('gpurun_out/r02/sweep_$f.json')):
  print(r['fuse'], 'pf', r['prefetch'], 'chunk', r['chunk'], 'us %.1f' % (r['ms_min']*1e3), 'med %.1f' % (r['ms_med']*1e3), 'GB/s %.0f' % r['GBs'])
PY
grep -h "^skip" gpurun_out/r02/sweep_$f.log | cut -c1-200
done
