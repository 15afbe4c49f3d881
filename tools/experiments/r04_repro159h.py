#!/usr/bin/env python3
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import c_oracle
from soda_amd import core, runtime
from soda_amd.codegen.hip import lower
HEAD = 'kernel: d159\nburst width: 64\nunroll factor: 2\niterate: 1\ninput uint8: in0(32, *)\ninput uint8: in1\n'
LOC = 'local uint16: loc0(0, 0) = in1(2, 0) * 43 * (min(16, 32) * in0(1, 0)) - (in1(-1, -1) + 10) * in1(0, -2) + (in0(2, -2) - in0(2, 1)) * 40 + in0(1, 2)\n'
OUT0 = 'output uint8: out0(0, 0) = in1(2, 1) * 1\n'
expr = sys.argv[1] if len(sys.argv) > 1 else '(min(int32(in0(-2, 1)), 4) + in1(-2, -2) * in1(2, -1) - in0(1, -2) * in1(0, 1)) / 6'
extent = (520, 291)
rng = np.random.default_rng(4401)
ins = {n: rng.integers(1, 201, extent[::-1]).astype(np.uint8) for n in ('in0', 'in1')}
st = core.from_text(HEAD + LOC + OUT0 + 'output uint8: out1(0, 0) = %s\n' % expr)
want = c_oracle.COracle(st).run(ins)
with runtime.Program(st, lower.LowerOptions(peel=0), extent=extent) as prog:
  got = prog.run(ins)
  name = prog.module.kernels[0].name
lo, hi = st.valid_box(extent, 'out1')
idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
bad = got['out1'][idx] != want['out1'][idx]
ys, xs = np.nonzero(bad)
print(os.environ.get('SODA_HIP_EXTRA_FLAGS'), name[-30:], 'bad', int(bad.sum()), 'cols mod 8', sorted(set((xs + lo[0]) % 8)))
if bad.any():
  y, x = ys[0] + lo[1], xs[0] + lo[0]
  a, b = ins['in0'].astype(int), ins['in1'].astype(int)
  print(' first bad (y,x)=(%d,%d) got %d want %d; in0(-2,1)=%d in1(-2,-2)=%d in1(2,-1)=%d in0(1,-2)=%d in1(0,1)=%d' % (
      y, x, got['out1'][y, x], want['out1'][y, x], a[y + 1, x - 2], b[y - 2, x - 2], b[y - 1, x + 2], a[y - 2, x + 1], b[y + 1, x]))
