#!/usr/bin/env python3
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import c_oracle
from soda_amd import core, runtime
from soda_amd.codegen.hip import lower
HEAD = 'kernel: d159\nburst width: 64\nunroll factor: 2\niterate: 1\ninput uint8: in0(32, *)\ninput uint8: in1\n'
progs = {
  'A': HEAD + 'local uint16: loc0(0, 0) = in0(1, 2) + in1(2, 0)\noutput uint8: out0(0, 0) = in1(2, 1)\noutput uint8: out1(0, 0) = in1(2, -1)\n',
  'B': HEAD + 'local uint16: loc0(0, 0) = in0(1, 2)\noutput uint8: out0(0, 0) = in1(2, 1)\noutput uint8: out1(0, 0) = in1(2, -1)\n',
  'C': HEAD + 'local uint16: loc0(0, 0) = in1(2, 0)\noutput uint8: out0(0, 0) = in1(2, 1)\noutput uint8: out1(0, 0) = in1(2, -1)\n',
  'D': HEAD + 'local uint16: loc0(0, 0) = in0(1, 2) + in1(2, 0)\noutput uint8: out0(0, 0) = in0(0, 0)\noutput uint8: out1(0, 0) = in1(2, -1)\n',
}
extent = (520, 291)
rng = np.random.default_rng(4401)
for tag, text in progs.items():
  st = core.from_text(text)
  ins = {n: rng.integers(0, 201, extent[::-1]).astype(np.uint8) for n in st.input_names}
  want = c_oracle.COracle(st).run(ins)
  with runtime.Program(st, lower.LowerOptions(peel=0), extent=extent) as prog:
    got = prog.run(ins)
    name = prog.module.kernels[0].name
    tile = prog.geometry(extent)[0][name]
  for o in st.output_names:
    lo, hi = st.valid_box(extent, o)
    idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
    bad = got[o][idx] != want[o][idx]
    ys, xs = np.nonzero(bad)
    msg = '%s %s %s tile %s bad %d' % (tag, o, name[-30:], tile, bad.sum())
    if bad.any():
      ay, ax = ys + lo[1], xs + lo[0]
      msg += ' rows-in-chunk %s cols mod 8 %s' % (sorted(set(ay % tile[1])), sorted(set(ax % 8)))
      # where did the wrong values come from?
      src = ins['in1'].astype(int)
      found = {}
      for dy in range(-6, 7):
        for dx in range(-10, 11):
          ok = 0; n = 0
          for y, x in list(zip(ay, ax))[:200]:
            if 0 <= y + dy < extent[1] and 0 <= x + dx < extent[0]:
              n += 1; ok += int(src[y + dy, x + dx]) == int(got[o][y, x])
          if n and ok > 0.9 * n:
            found[(dx, dy)] = ok
      zeros = int((got[o][idx][bad] == 0).sum())
      msg += ' source offsets %s zeros %d' % (found, zeros)
    print(msg, flush=True)
