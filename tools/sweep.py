#!/usr/bin/env python3
"""Kernel-parameter sweep on one GPU: times the dominant pass of a program for
every combination of knobs, interleaved over rounds in ONE process (so numbers
are comparable), HIP events on the launch stream.  Prints one line per config,
sorted by time.  Usage: python tools/sweep.py --fuse 1 4 --chunk 32 64 ..."""
import argparse
import itertools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--soda', default=os.path.join(ROOT, 'tests/golden/soda/jacobi2d.soda'))
  ap.add_argument('--extent', type=int, nargs='+', default=[8192, 8192])
  ap.add_argument('--fuse', type=int, nargs='+', default=[1])
  ap.add_argument('--chunk', type=int, nargs='+', default=[64])
  ap.add_argument('--prefetch', type=int, nargs='+', default=[2])
  ap.add_argument('--waves', nargs='+', default=['1x4'])
  ap.add_argument('--nt-store', type=int, nargs='+', default=[0])
  ap.add_argument('--nt-load', type=int, nargs='+', default=[0])
  ap.add_argument('--xcd', type=int, nargs='+', default=[0])
  ap.add_argument('--vec', type=int, nargs='+', default=[4])
  ap.add_argument('--tile-rows', type=int, nargs='+', default=[6])
  ap.add_argument('--edge', type=int, nargs='+', default=[1])
  ap.add_argument('--wg', type=int, nargs='+', default=[0])
  ap.add_argument('--il', type=int, nargs='+', default=[0])
  ap.add_argument('--shift', nargs='+', default=['dpp'])
  ap.add_argument('--mw', type=int, nargs='+', default=[0])
  ap.add_argument('--occ', type=int, nargs='+', default=[0])
  ap.add_argument('--buf', type=int, nargs='+', default=[1])
  ap.add_argument('--pipe', type=int, nargs='+', default=[1])
  ap.add_argument('--pipe-rows', type=int, nargs='+', default=[4])
  ap.add_argument("--peel", type=int, nargs="+", default=[-2])
  ap.add_argument('--reg-budget', type=int, default=None)
  ap.add_argument('--rounds', type=int, default=3)
  ap.add_argument('--reps', type=int, default=20)
  ap.add_argument('--launches', type=int, default=1, help='launches per call (ping-pong through the program temporaries, as in a real run)')
  ap.add_argument('--strategy', default='auto')
  ap.add_argument('--out', default=None)
  args = ap.parse_args()
  import torch
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  dev = torch.device('cuda', 0)
  shape = tuple(args.extent[::-1])
  configs = list(itertools.product(args.fuse, args.chunk, args.prefetch,
                                   args.waves, args.nt_store, args.nt_load,
                                   args.xcd, args.vec, args.tile_rows, args.edge, args.wg, args.il, args.shift, args.mw, args.occ, args.buf, args.pipe, args.pipe_rows, args.peel))
  progs = []
  stream = torch.cuda.current_stream().cuda_stream
  for fuse, chunk, pf, waves, nts, ntl, xcd, vec, trows, edge, wg, il, shift, mw, occ, buf, pipe, prow, peel in configs:
    st = core.from_file(args.soda, iterate=fuse * args.launches)
    wx, wy = map(int, waves.split('x'))
    opts = lower.LowerOptions(strategy=args.strategy, fuse=(fuse,) if fuse > 1 else (), chunk_rows=chunk if chunk > 0 else None,
                              prefetch=pf, waves_x=wx, waves_y=wy,
                              nt_store=bool(nts), nt_load=bool(ntl),
                              xcd_swizzle=bool(xcd), vec=vec if vec > 0 else None,
                              tile_rows=trows, edge_loads=bool(edge), warm_guards=bool(wg), interleave=bool(il), lane_shift=shift, min_waves=mw, occupancy=occ, buffer_ops=bool(buf), pipe=pipe, pipe_rows=prow, reg_budget=args.reg_budget, peel=(None if peel == -2 else peel))
    try:
      progs.append((runtime.Program(st, opts, extent=args.extent), st, fuse))
    except Exception as e:  # noqa
      print('skip', fuse, chunk, pf, waves, str(e)[:200])
      progs.append(None)
  st0 = core.from_file(args.soda)
  tdt = {'float32': torch.float32, 'float64': torch.float64, 'uint16': torch.int16,
         'int16': torch.int16, 'int32': torch.int32, 'uint8': torch.uint8}
  def mk(t, rand):
    dt = tdt[t.np_name]
    if rand and dt.is_floating_point:
      return torch.rand(shape, device=dev, dtype=dt)
    if rand:
      return torch.randint(0, 30000, shape, device=dev, dtype=dt)
    return torch.empty(shape, device=dev, dtype=dt)
  ins = [mk(t, True) for t in st0.input_types]
  outs = [mk(t, False) for t in st0.output_types]
  bytes_cell = sum(t.size_in_bytes for t in st0.input_types) + sum(t.size_in_bytes for t in st0.output_types)
  times = {i: [] for i in range(len(configs))}
  for r in range(args.rounds):
    for i, item in enumerate(progs):
      if item is None:
        continue
      prog, st, fuse = item
      def go():
        prog.run_device([t.data_ptr() for t in outs], [t.data_ptr() for t in ins],
                        args.extent, iterate=fuse * args.launches, stream=stream)
      go()
      a, b = runtime.Event(), runtime.Event()
      a.record(stream)
      for _ in range(args.reps):
        go()
      b.record(stream)
      times[i].append(a.elapsed_ms(b) / args.reps / args.launches)
  cells = 1
  for e in args.extent:
    cells *= e
  rows = []
  for i, cfg in enumerate(configs):
    if not times[i]:
      continue
    best = min(times[i]); med = sorted(times[i])[len(times[i]) // 2]
    fuse = cfg[0]
    rows.append(dict(fuse=fuse, chunk=cfg[1], prefetch=cfg[2], waves=cfg[3],
                     nt_store=cfg[4], nt_load=cfg[5], xcd=cfg[6], vec=cfg[7], tile_rows=cfg[8], edge=cfg[9], wg=cfg[10], il=cfg[11], shift=cfg[12], mw=cfg[13], occ=cfg[14], buf=cfg[15], pipe=cfg[16], pipe_rows=cfg[17], peel=cfg[18], ms_min=best,
                     ms_med=med, GBs=cells * bytes_cell / best / 1e6,
                     Gcell_iters=cells * fuse / best / 1e6,
                     kernel=progs[i][0].module.kernels[0].name))
  rows.sort(key=lambda r: (r['fuse'], r['ms_min']))
  for r in rows:
    print(json.dumps(r))
  if args.out:
    with open(args.out, 'w') as f:
      json.dump(rows, f, indent=1)


if __name__ == '__main__':
  main()
