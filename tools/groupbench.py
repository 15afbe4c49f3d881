#!/usr/bin/env python3
"""Times the slab group (soda_hip_group_*) with N virtual devices on one GPU.

  python tools/groupbench.py [--slabs 8] [--iterate 100] [--steps 10]
                             [--soda jacobi2d.soda] [--extent 8192 8192]

Per configuration (overlap on / off, exchange interval): milliseconds per step
of `iterate` iterations (all slabs, sharing the one GPU), host time spent
enqueueing a step, launches / copies / split passes per step.  On one GPU the
slabs run one after the other, so ms per step ~ N x one slab's time: what this
shows is the cost of the split launches and of the event chain, and that the
host keeps ahead of the GPU -- not the transfer time of a real xGMI link.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--soda', default='jacobi2d.soda')
  ap.add_argument('--extent', type=int, nargs='+', default=[8192, 8192])
  ap.add_argument('--iterate', type=int, default=100)
  ap.add_argument('--fuse', type=int, nargs='*', default=[12, 8, 4])
  ap.add_argument('--slabs', type=int, default=8)
  ap.add_argument('--steps', type=int, default=10)
  ap.add_argument('--every', type=int, nargs='*', default=[0])
  ap.add_argument('--no-calibrate', action='store_true')
  ap.add_argument('--only', choices=('overlap', 'serial'), default=None)
  ap.add_argument('--out', default=None)
  args = ap.parse_args()
  import numpy as np
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  path = os.path.join(ROOT, 'tests', 'golden', 'soda', args.soda)
  stencil = core.from_file(path, iterate=args.iterate)
  extent = tuple(args.extent)
  rng = np.random.default_rng(0)
  inputs = {n: rng.random(extent[::-1], dtype=np.float32)
            for n in stencil.input_names}
  rows = []
  for every in args.every:
    for overlap in (True, False):
      if args.only and (args.only == 'overlap') != overlap:
        continue
      with runtime.Group(stencil, extent, [0] * args.slabs,
                         lower.LowerOptions(fuse=tuple(args.fuse)),
                         exchange_every=every, overlap=overlap,
                         calibrate=not args.no_calibrate) as group:
        group.load(inputs)
        for _ in range(3):
          group.run()
        group.synchronize()
        enq = 0.0
        t0 = time.perf_counter()
        for _ in range(args.steps):
          group.run()
          enq += group.stats()['enqueue_ms']
        t_enq = time.perf_counter() - t0
        group.synchronize()
        dt = time.perf_counter() - t0
        st = group.stats()
        slab = group.slab(args.slabs // 2)
        row = {
            'program': stencil.app_name, 'extent': list(extent),
            'iterate': args.iterate, 'slabs': args.slabs,
            'overlap': overlap, 'exchange_every': st['exchange_every'],
            'ms_per_step': dt / args.steps * 1e3,
            'host_enqueue_ms_per_step': enq / args.steps,
            'host_loop_ms_per_step': t_enq / args.steps * 1e3,
            'slab_rows': slab.extent[stencil.dim - 1],
            'ghost_rows': [slab.ghost_lo, slab.ghost_hi],
            **{k: st[k] for k in ('intervals', 'exchanges', 'copies',
                                  'copy_bytes', 'launches', 'split_passes')},
        }
        rows.append(row)
        print(json.dumps(row), flush=True)
  if args.out:
    with open(args.out, 'w') as f:
      for r in rows:
        f.write(json.dumps(r) + '\n')


if __name__ == '__main__':
  main()
