#!/usr/bin/env python3
"""Localises tools/fuzz_scan.py deep seed 159 (march2d T1 V8 on uint8 tensors,
520 x 291): which cells, which knobs."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import fuzz_nest
from oracle import c_oracle
from soda_amd import core, runtime
from soda_amd.codegen.hip import lower

nest, _ = fuzz_nest.program(159, 'plain')
nest.iterate = 1
text = nest.soda_text()
if len(sys.argv) > 1 and sys.argv[1] == 'min':
  text = '''kernel: m159
burst width: 64
unroll factor: 2
iterate: 1
input uint8: in0(32, *)
input uint8: in1
output uint8: out0(0, 0) = in1(2, 1) * 1
output uint8: out1(0, 0) = (min(int32(in0(-2, 1)), 4) + in1(-2, -2) * in1(2, -1) - (in0(2, -2) + 8 * 16) - in0(1, -2) * in1(0, 1)) / 6
'''
st = core.from_text(text)
for extent in ((520, 291), (512, 291), (520, 64), (1032, 100), (300, 80)):
  rng = np.random.default_rng(4401)
  ins = {n: rng.integers(0, 201, extent[::-1]).astype(np.uint8) for n in st.input_names}
  want = c_oracle.COracle(st).run(ins)
  for kw in (dict(), dict(vec=4), dict(inline=False), dict(windows=False), dict(peel=0), dict(chunk_rows=16), dict(nt_store=False)):
    try:
      with runtime.Program(st, lower.LowerOptions(**kw), extent=extent) as prog:
        got = prog.run(ins)
        name = prog.module.kernels[0].name
    except Exception as e:
      print(extent, kw, 'ERR', str(e)[:100]); continue
    out = []
    for o in st.output_names:
      lo, hi = st.valid_box(extent, o)
      idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
      bad = got[o][idx] != want[o][idx]
      ys, xs = np.nonzero(bad)
      out.append('%s: %d bad' % (o, bad.sum()) + ('' if not bad.any() else
                 ' rows %d..%d cols %d..%d, cols mod 8: %s, first (y,x)=(%d,%d) got %d want %d' % (
                     ys.min(), ys.max(), xs.min(), xs.max(), sorted(set((xs + lo[0]) % 8))[:8],
                     ys[0] + lo[1], xs[0] + lo[0], got[o][idx][ys[0], xs[0]], want[o][idx][ys[0], xs[0]])))
    print(extent, kw, name[-40:], ' | '.join(out), flush=True)
