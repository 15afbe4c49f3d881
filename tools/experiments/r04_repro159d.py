#!/usr/bin/env python3
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import fuzz_nest
from oracle import c_oracle
from soda_amd import core, runtime
from soda_amd.codegen.hip import lower
nest, _ = fuzz_nest.program(159, 'plain'); nest.iterate = 1
texts = {'orig': nest.soda_text()}
# without the unused let; without the dead local; dead local as uint8
texts['no_let'] = texts['orig'].replace('output uint8:\n  uint8 tmp = in1(-2, -1) * 8 * 5\n  out0(0, 0) = in1(2, 1) * 1', 'output uint8: out0(0, 0) = in1(2, 1) * 1')
texts['no_loc'] = '\n'.join(l for l in texts['orig'].splitlines() if not l.startswith('local')) + '\n'
texts['loc_u8'] = texts['orig'].replace('local uint16:', 'local uint8:')
texts['loc_i32'] = texts['orig'].replace('local uint16:', 'local int32:')
extent = (520, 291)
rng = np.random.default_rng(4401)
for tag, text in texts.items():
  st = core.from_text(text)
  ins = {n: rng.integers(0, 201, extent[::-1]).astype(np.uint8) for n in st.input_names}
  want = c_oracle.COracle(st).run(ins)
  for kw in (dict(peel=0), dict(peel=0, lane_shift='bperm'), dict(peel=0, lane_shift='swz'), dict(peel=0, edge_loads=False),
             dict(peel=0, prefetch=1), dict(peel=0, prefetch=4), dict(peel=0, buffer_ops=False)):
    try:
      with runtime.Program(st, lower.LowerOptions(**kw), extent=extent) as prog:
        got = prog.run(ins)
        name = prog.module.kernels[0].name
    except Exception as e:
      print(tag, kw, 'ERR', str(e)[:120]); continue
    lo, hi = st.valid_box(extent, 'out1')
    idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
    bad = got['out1'][idx] != want['out1'][idx]
    ys, xs = np.nonzero(bad)
    print(tag, kw, name[-44:], 'bad', int(bad.sum()), 'rows mod 7', sorted(set((ys + lo[1]) % 7))[:7], 'cols mod 8', sorted(set((xs + lo[0]) % 8)), flush=True)
