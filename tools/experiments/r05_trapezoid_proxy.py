#!/usr/bin/env python3
"""VERDICT r4 item 2 -- split (trapezoid) tiling along the marching dimension,
bounded by measurement BEFORE building it.

Phase A of a split tiling with chunk pitch C computes, per chunk, level l on
rows [m + l, m + C - l): exactly what today's kernel computes for an output
chunk of C - 2T rows (its warm-up IS that trapezoid, peeled).  So phase A on a
grid of N rows = today's kernel, chunk c = C - 2T, on a grid of (N / C) * c
rows: same waves, same row steps per wave, same loads (it stores 2 edge rows
per level more, not counted here: the proxy is a LOWER bound).  Phase B (the
inverted trapezoids over the N / C chunk boundaries, T (T + 1) level-rows in
2T row steps each) is bracketed by today's kernel with chunk 1 (T + 1 row
steps, half the level-rows, a third of the traffic) and with chunk 2T (4T row
steps, twice the level-rows).

Usage: python tools/experiments/r05_trapezoid_proxy.py [--fuse 13] [--out F]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--fuse', type=int, default=13)
  ap.add_argument('--cols', type=int, default=8192)
  ap.add_argument('--rounds', type=int, default=4)
  ap.add_argument('--reps', type=int, default=20)
  ap.add_argument('--out', default=None)
  args = ap.parse_args()
  import torch
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  T = args.fuse
  soda = os.path.join(ROOT, 'tests/golden/soda/jacobi2d.soda')
  # (label, rows, chunk or 0 = the library's choice)
  cases = [('full grid, today', 8192, 0)]
  for pitch in (204, 128, 90):
    n = 8192 // pitch
    c = pitch - 2 * T
    cases += [('full: phase A pitch %d' % pitch, n * c, c),
              ('full: phase B lower, %d seams' % n, n, 1),
              ('full: phase B upper, %d seams' % n, n * 2 * T, 2 * T)]
  cases += [('slab 1224, today', 1224, 0)]
  for pitch in (51, 38, 30):
    n = 1224 // pitch
    c = pitch - 2 * T
    cases += [('slab: phase A pitch %d' % pitch, n * c, c),
              ('slab: phase B lower, %d seams' % n, n, 1),
              ('slab: phase B upper, %d seams' % n, n * 2 * T, 2 * T)]
  dev = torch.device('cuda', 0)
  stream = torch.cuda.current_stream().cuda_stream
  st = core.from_file(soda, iterate=T)
  progs = []
  seen = {}
  for label, rows, chunk in cases:
    key = (rows, chunk)
    if key not in seen:
      extent = (args.cols, rows)
      try:
        prog = runtime.Program(
            st, lower.LowerOptions(fuse=(T,), chunk_rows=chunk or None),
            extent=extent)
        a = torch.rand((rows, args.cols), device=dev, dtype=torch.float32)
        b = torch.empty_like(a)
        seen[key] = (prog, a, b, extent)
      except Exception as e:  # noqa
        print('skip', label, str(e)[:200], flush=True)
        seen[key] = None
    progs.append(seen[key])
  times = [[] for _ in cases]
  for _ in range(args.rounds):
    for i, item in enumerate(progs):
      if item is None:
        continue
      prog, a, b, extent = item

      def go():
        prog.run_device([b.data_ptr()], [a.data_ptr()], extent, iterate=T,
                        stream=stream)
      go()
      e0, e1 = runtime.Event(), runtime.Event()
      e0.record(stream)
      for _ in range(args.reps):
        go()
      e1.record(stream)
      times[i].append(e0.elapsed_ms(e1) / args.reps * 1e3)
  rows_out = []
  for (label, rows, chunk), item, ts in zip(cases, progs, times):
    if item is None or not ts:
      continue
    prog = item[0]
    tile = prog.geometry(item[3])[0] if hasattr(prog, 'geometry') else None
    r = dict(case=label, rows=rows, chunk=chunk, us_min=round(min(ts), 2),
             us_med=round(sorted(ts)[len(ts) // 2], 2),
             kernel=prog.module.kernels[0].name, tile=tile)
    rows_out.append(r)
    print(json.dumps(r), flush=True)
  if args.out:
    with open(args.out, 'w') as f:
      json.dump(rows_out, f, indent=1)


if __name__ == '__main__':
  main()
