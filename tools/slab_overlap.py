#!/usr/bin/env python3
"""What hiding a halo exchange costs ONE rank: the middle slab of an N-GPU run
alone on the GPU (as on its own GPU in the real run), chained steps of
`iterate` iterations, the exchange emulated by device copies of the same size
on a second stream.

  none     no exchange at all (lower bound)
  serial   copies, then the passes (what soda_amd.dist did in round 2)
  split    soda_hip_run_device_slab: first / last pass in two parts around the
           copies (SODA_HIP_SPLIT=inorder: both parts on the launch stream)

The copies here are on-device (a few us); over xGMI they take tens of us, which
`serial` pays in full and `split` hides.  What this measures is the PRICE of
the split launches.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--soda', default='jacobi2d.soda')
  ap.add_argument('--extent', type=int, nargs='+', default=[8192, 8192])
  ap.add_argument('--iterate', type=int, default=100)
  ap.add_argument('--every', type=int, default=0)
  ap.add_argument('--fuse', type=int, nargs='*', default=[12, 8, 4])
  ap.add_argument('--gpus', type=int, default=8)
  ap.add_argument('--steps', type=int, default=30)
  ap.add_argument('--out', default=None)
  args = ap.parse_args()
  import torch
  from soda_amd import core, dist as sdist, runtime
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(os.path.join(ROOT, 'tests', 'golden', 'soda',
                                        args.soda), iterate=args.iterate)
  fuse = tuple(args.fuse)
  every = args.every or sdist.auto_exchange_every(
      stencil, args.extent, args.gpus, args.iterate,
      multiple_of=max(fuse) if fuse else 1)
  slab = sdist.Slab(stencil, args.extent, args.gpus, args.gpus // 2, every)
  lext = slab.local_extent
  prog = runtime.Program(stencil, lower.LowerOptions(fuse=fuse), extent=lext,
                         calibrate=True)
  shape = tuple(lext[::-1])
  a = torch.rand(shape, device='cuda')
  b = torch.empty_like(a)
  row_bytes = slab.row_cells * 4
  main_stream = runtime.Stream()
  comm = runtime.Stream()
  ready, sendable = runtime.Event(), runtime.Event()
  msgs = slab.messages()

  def copies(arr, stream):
    base = arr.data_ptr()
    for _, (s0, s1), (r0, r1) in msgs:
      stream.copy(base + r0 * row_bytes, base + s0 * row_bytes,
                  (s1 - s0) * row_bytes)

  def interval(dst, src, iters, mode):
    kw = dict(iterate=iters, stream=main_stream.handle, origin=slab.origin,
              global_extent=slab.extent, keep=slab.keep)
    if mode == 'none':
      prog.run_device([dst.data_ptr()], [src.data_ptr()], lext, **kw)
    elif mode == 'serial':
      copies(src, main_stream)
      prog.run_device([dst.data_ptr()], [src.data_ptr()], lext, **kw)
    else:
      comm.wait_event(sendable)
      copies(src, comm)
      ready.record(comm.handle)
      prog.run_device([dst.data_ptr()], [src.data_ptr()], lext,
                      ghosts=(slab.ghost_lo, slab.ghost_hi),
                      sends=(slab.reach_hi * every, slab.reach_lo * every),
                      ghosts_ready=ready.handle(), sendable=sendable.handle(),
                      **kw)

  def step(mode, state):
    done = 0
    while done < args.iterate:
      k = min(every, args.iterate - done)
      interval(state[1], state[0], k, mode)
      state[0], state[1] = state[1], state[0]
      done += k

  rows = []
  for mode in ('none', 'serial', 'split', 'none', 'serial', 'split'):
    state = [a, b]
    sendable.record(main_stream.handle)
    for _ in range(5):
      step(mode, state)
    main_stream.synchronize()
    t0, t1 = runtime.Event(), runtime.Event()
    t0.record(main_stream.handle)
    for _ in range(args.steps):
      step(mode, state)
    t1.record(main_stream.handle)
    ms = t0.elapsed_ms(t1) / args.steps
    comm.synchronize()
    row = {'program': stencil.app_name, 'slab': list(lext), 'gpus': args.gpus,
           'iterate': args.iterate, 'exchange_every': every, 'mode': mode,
           'split_env': os.environ.get('SODA_HIP_SPLIT', 'side'),
           'ms_per_step': ms, 'launches': prog.last_launches()[0],
           'split_passes': prog.last_split(),
           'schedule': prog.schedule(lext, min(every, args.iterate))}
    rows.append(row)
    print(json.dumps(row), flush=True)
  if args.out:
    with open(args.out, 'a') as f:
      for r in rows:
        f.write(json.dumps(r) + '\n')


if __name__ == '__main__':
  main()
