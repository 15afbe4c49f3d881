#!/usr/bin/env python3
"""VERDICT r4 item 6 (erosion / xcorr re-read their 18-row window per chunk:
fetch 1.48x / 1.27x) -- what is the re-read worth, measured before anything is
built to avoid it.

A marching wave that starts a chunk must load the window rows above it again
(its neighbour read them ~40 us earlier: L2 has long forgotten them).  The
cost of exactly that, isolated: two int16 programs with ONE operation per
cell, `min(i(0,0), i(0,1))` (window 2 rows) and `min(i(0,0), i(0,18))` (19
rows), same cells per lane, same chunk lengths -- the difference is the
re-read (and the taller register window).  Then erosion and xcorr themselves
at several chunk lengths: longer chunks re-read less and offer fewer waves.

  python tools/experiments/r05_window_reread.py --out gpurun_out/r05_window_reread.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

PROBE = """kernel: win%(h)d
burst width: 64
unroll factor: 4
input dram 0 int16: i(480, *)
output dram 1 int16: o(0, 0) = min(i(0, 0), i(0, %(h)d))
iterate: 1
border: ignore
cluster: none
"""


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--out', default=None)
  ap.add_argument('--reps', type=int, default=30)
  ap.add_argument('--rounds', type=int, default=4)
  args = ap.parse_args()
  import torch
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  extent = (8192, 8192)
  shape = extent[::-1]
  stream = torch.cuda.current_stream().cuda_stream
  a = torch.randint(0, 30000, shape, device='cuda', dtype=torch.int16)
  b = torch.empty_like(a)
  cases = []
  for h in (1, 18):
    for chunk in (0, 64, 128):
      cases.append(('probe: min over rows 0 and %d' % h,
                    core.from_text(PROBE % {'h': h}), chunk))
  soda = os.path.join(ROOT, 'tests', 'golden', 'soda')
  for name in ('erosion.soda', 'xcorr.soda'):
    for chunk in (0, 48, 64, 96, 128, 192):
      cases.append((name[:-5], core.from_file(os.path.join(soda, name)), chunk))
  progs = []
  for label, st, chunk in cases:
    try:
      progs.append(runtime.Program(
          st, lower.LowerOptions(chunk_rows=chunk or None), extent=extent))
    except Exception as e:   # noqa
      print('skip', label, chunk, str(e)[:160], flush=True)
      progs.append(None)
  times = [[] for _ in cases]
  for _ in range(args.rounds):
    for i, prog in enumerate(progs):
      if prog is None:
        continue

      def go():
        prog.run_device([b.data_ptr()], [a.data_ptr()], extent, stream=stream)
      go()
      e0, e1 = runtime.Event(), runtime.Event()
      e0.record(stream)
      for _ in range(args.reps):
        go()
      e1.record(stream)
      times[i].append(e0.elapsed_ms(e1) / args.reps * 1e3)
  rows = []
  for (label, st, chunk), prog, ts in zip(cases, progs, times):
    if prog is None or not ts:
      continue
    k = prog.module.kernels[0]
    tile = prog.geometry(extent)[0][k.name]
    warm = (k.tune or {}).get('warm')
    r = dict(case=label, chunk_asked=chunk, chunk_rows=tile[1], warm_rows=warm,
             reread=(tile[1] + (warm or 0)) / float(tile[1]),
             us_min=round(min(ts), 2), us_med=round(sorted(ts)[len(ts) // 2], 2),
             kernel=k.name, vgprs=prog.resources.get(k.name, {}).get('vgpr'),
             algorithmic_GBs=8192 * 8192 * 4 / min(ts) / 1e3)
    rows.append(r)
    print(json.dumps(r), flush=True)
  if args.out:
    with open(args.out, 'w') as f:
      json.dump(rows, f, indent=1)


if __name__ == '__main__':
  main()
