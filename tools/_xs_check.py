import sys, os, json
sys.path.insert(0, '.')
import numpy as np, torch
from soda_amd import core, runtime
from soda_amd.codegen.hip import lower
from oracle import c_oracle
ok = True
for name, extent, it, fuse in [('heat3d', (300, 20, 24), 4, (2,)), ('heat3d', (512, 16, 20), 6, (2,)),
                               ('heat3d', (256, 12, 18), 4, (2,)), ('jacobi3d', (508, 14, 16), 4, (2,)),
                               ('heat3d', (1024, 9, 12), 2, (2,)), ('heat3d', (64, 9, 40), 5, (2,))]:
  st = core.from_file('tests/golden/soda/%s.soda' % name, iterate=it)
  rng = np.random.default_rng(1)
  x = rng.random(tuple(extent[::-1]), dtype=np.float32)
  want = c_oracle.COracle(st).run({st.input_names[0]: x})[st.output_names[0]]
  with runtime.Program(st, lower.LowerOptions(fuse=fuse), extent=extent) as prog:
    got = prog.run({st.input_names[0]: x})[st.output_names[0]]
    names = [k.name for k in prog.module.kernels]
    vg = {k: v['vgpr'] for k, v in prog.resources.items()}
  lo, hi = st.valid_box(extent)
  idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
  same = np.array_equal(got[idx], want[idx])
  ok &= same
  print(name, extent, it, 'OK' if same else 'MISMATCH %d' % int((got[idx] != want[idx]).sum()), names[0][-20:], vg)
print('ALL OK' if ok else 'FAILED')
