#!/usr/bin/env python3
"""One-off scan of random programs beyond the seeds the test suite holds: the
generic generator (tests/fuzz.py program) and the window generator
(window_program), GPU kernels (auto and direct) against the C oracle, bit for
bit.  Usage: python tools/fuzz_scan.py window|generic|rich FIRST LAST"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
  import fuzz
  from oracle import c_oracle
  from soda_amd import core, runtime, util
  from soda_amd.codegen.hip import lower
  kind, first, last = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
  gen = (fuzz.window_program if kind == 'window' else
         (lambda s: fuzz.program(s, rich=True)) if kind == 'rich' else
         fuzz.program)
  ext_for = fuzz.window_extent_for if kind == 'window' else fuzz.extent_for
  ran = failed = 0
  t0 = time.time()
  for seed in range(first, last):
    text, dim, _ = gen(seed)
    try:
      stencil = core.from_text(text)
    except util.SodaError:
      continue
    extent = ext_for(seed, dim)
    lo, hi = stencil.valid_box(extent)
    if not all(h > l for l, h in zip(lo, hi)):
      continue
    ins = fuzz.inputs_for(stencil, extent, seed)
    want = c_oracle.COracle(stencil, openmp=False).run(ins)
    ran += 1
    for strategy in ('auto', 'direct'):
      try:
        with runtime.Program(stencil, lower.LowerOptions(strategy=strategy,
                                                         fuse=(2,)),
                             extent=extent) as prog:
          got = prog.run(ins)
      except Exception as e:  # noqa
        failed += 1
        print('seed %d %s: %s: %s\n%s' % (seed, strategy, type(e).__name__,
                                          str(e)[:300], text), flush=True)
        continue
      for o in stencil.output_names:
        lo, hi = stencil.valid_box(extent, o)
        idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
        if not np.array_equal(got[o][idx], want[o][idx], equal_nan=True):
          failed += 1
          print('seed %d %s output %s: %d cells differ\n%s' %
                (seed, strategy, o, int((got[o][idx] != want[o][idx]).sum()),
                 text), flush=True)
    if ran % 25 == 0:
      print('... %d programs, %d failures, %.0f s' % (ran, failed,
                                                     time.time() - t0),
            flush=True)
  print('%s seeds [%d, %d): %d programs run, %d failures' %
        (kind, first, last, ran, failed))
  return 1 if failed else 0


if __name__ == '__main__':
  sys.exit(main())
