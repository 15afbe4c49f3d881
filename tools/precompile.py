#!/usr/bin/env python3
"""Pre-JITs the kernels a sweep will need into soda_amd/_jit_cache (hiprtc, no
GPU needed) so that a gpurun call spends its minutes measuring, not compiling.
Takes the same knob lists as tools/sweep.py."""
import argparse
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--soda', default=os.path.join(ROOT, 'tests/golden/soda/jacobi2d.soda'))
  ap.add_argument('--extent', type=int, nargs='+', default=[8192, 8192])
  ap.add_argument('--fuse', type=int, nargs='+', default=[12])
  ap.add_argument('--prefetch', type=int, nargs='+', default=[2])
  ap.add_argument('--vec', type=int, nargs='+', default=[4])
  ap.add_argument('--shift', nargs='+', default=['dpp'])
  ap.add_argument('--pipe', type=int, nargs='+', default=[1])
  ap.add_argument('--pipe-rows', type=int, nargs='+', default=[2])
  ap.add_argument('--nt-load', type=int, nargs='+', default=[1])
  ap.add_argument('--nt-store', type=int, nargs='+', default=[0])
  ap.add_argument('--tile-rows', type=int, nargs='+', default=[6])
  ap.add_argument('--peel', type=int, nargs='+', default=[-2])
  ap.add_argument('--mw', type=int, nargs='+', default=[0])
  ap.add_argument('--reg-budget', type=int, default=None)
  ap.add_argument('--launches', type=int, default=1)
  args = ap.parse_args()
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  for fuse, pf, vec, shift, pipe, prow, ntl, nts, trows, peel, mw in itertools.product(
      args.fuse, args.prefetch, args.vec, args.shift, args.pipe,
      args.pipe_rows, args.nt_load, args.nt_store, args.tile_rows,
      args.peel, args.mw):
    st = core.from_file(args.soda, iterate=fuse * args.launches)
    opts = lower.LowerOptions(fuse=(fuse,) if fuse > 1 else (), prefetch=pf,
                              vec=vec, lane_shift=shift, pipe=pipe,
                              pipe_rows=prow, nt_load=bool(ntl),
                              nt_store=bool(nts), xcd_swizzle=True,
                              tile_rows=trows, waves_x=1, waves_y=1,
                              peel=(None if peel == -2 else peel),
                              min_waves=mw, reg_budget=args.reg_budget)
    try:
      opts = runtime.resolve_options(st, opts, args.extent)
      mod = lower.lower(st, opts)
      code = runtime.compile_source(mod.source, '%s.hip' % st.app_name)
      res = runtime.kernel_resources(code)
      k = mod.kernels[0].name
      print(k, res.get(k))
    except Exception as e:  # noqa
      print('skip', fuse, pf, vec, shift, pipe, str(e)[:200])


if __name__ == '__main__':
  main()
