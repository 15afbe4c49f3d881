#!/usr/bin/env python3
"""Repro of tools/fuzz_scan.py deep, seed 159."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import fuzz_nest
from oracle import c_oracle
from soda_amd import core, runtime
from soda_amd.codegen.hip import lower

nest, _ = fuzz_nest.program(159, 'plain')
extent = (520, 291)
for iterate in (1, 2, 3, 4, 5, 8, 13, 26):
  nest.iterate = iterate
  text = nest.soda_text()
  st = core.from_text(text)
  ins = fuzz_nest.inputs_for(nest, extent, 159)
  want = c_oracle.COracle(st).run(ins)
  own = nest.run(ins, extent)
  for kw in (dict(fuse=(13, 12, 4)), dict(fuse=()), dict(strategy='direct')):
    with runtime.Program(st, lower.LowerOptions(**kw), extent=extent) as prog:
      got = prog.run(ins)
      kinds = [(p.kind, p.fused_iters) for p in prog.module.passes]
      names = [k.name for k in prog.module.kernels]
    line = []
    for o in st.output_names:
      lo, hi = st.valid_box(extent, o)
      idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
      line.append('%s box %s..%s gpu-vs-oracle %d oracle-vs-nest %d' % (
          o, lo, hi, int((got[o][idx] != want[o][idx]).sum()),
          int((want[o] != own[o]).sum())))
    print(iterate, kw, kinds, names[:2], ' | '.join(line), flush=True)
