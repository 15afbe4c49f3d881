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
extent = (520, 291)
rng = np.random.default_rng(4401)
ins = {n: rng.integers(1, 201, extent[::-1]).astype(np.uint8) for n in ('in0', 'in1')}
cases = {
  'U124': '(min(int32(in0(-2, 1)), 4) + in1(-2, -2) * in1(2, -1) - in0(1, -2) * in1(0, 1)) / 6',
  'in1(0,1)': 'in1(0, 1)', 'in0(-2,1)': 'in0(-2, 1)', 'in1(2,-1)': 'in1(2, -1)', 'in1(-2,-2)': 'in1(-2, -2)',
  'in0(1,-2)': 'in0(1, -2)', 'prod': 'in1(-2, -2) * in1(2, -1)', 'prod2': 'in0(1, -2) * in1(0, 1)',
  'sum4': 'in0(-2, 1) + in1(-2, -2) + in1(2, -1) + in0(1, -2) + in1(0, 1)',
}
for tag, expr in cases.items():
  st = core.from_text(HEAD + LOC + OUT0 + 'output uint8: out1(0, 0) = %s\n' % expr)
  want = c_oracle.COracle(st).run(ins)
  with runtime.Program(st, lower.LowerOptions(peel=0), extent=extent) as prog:
    got = prog.run(ins)
    name = prog.module.kernels[0].name
    tile = prog.geometry(extent)[0][name]
  o = 'out1'
  lo, hi = st.valid_box(extent, o)
  idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
  bad = got[o][idx] != want[o][idx]
  ys, xs = np.nonzero(bad)
  msg = '%-10s %s tile %s bad %d' % (tag, name[-34:], tile, bad.sum())
  if bad.any():
    ay, ax = ys + lo[1], xs + lo[0]
    msg += ' rows-in-chunk %s cols mod 8 %s' % (sorted(set(ay % tile[1])), sorted(set(ax % 8)))
    if '(' in tag and tag.count('(') == 1:
      name_in = tag[:3]
      src = ins[name_in].astype(int)
      found = {}
      for dy in range(-8, 9):
        for dx in range(-10, 11):
          ok = n = 0
          for y, x in list(zip(ay, ax))[:300]:
            if 0 <= y + dy < extent[1] and 0 <= x + dx < extent[0]:
              n += 1; ok += int(src[y + dy, x + dx]) == int(got[o][y, x])
          if n and ok > 0.9 * n:
            found[(dx, dy)] = ok
      msg += ' source offsets %s zeros %d' % (found, int((got[o][idx][bad] == 0).sum()))
  print(msg, flush=True)
