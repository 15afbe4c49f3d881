#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-array entry (Program.run = soda_hip_run_host_box)
for the headline workload: copies in, 100 iterations, copies the valid box out."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from soda_amd import core, runtime
from soda_amd.codegen.hip import lower

st = core.from_file(os.path.join(ROOT, 'tests/golden/soda/jacobi2d.soda'), iterate=100)
extent = (8192, 8192)
a = np.random.default_rng(0).random(extent[::-1], dtype=np.float32)
out = {'t0': np.zeros_like(a)}
with runtime.Program(st, lower.LowerOptions(fuse=(12, 4)), extent=extent) as prog:
  prog.run({'t1': a}, outputs=out)
  t0 = time.time()
  for _ in range(3):
    prog.run({'t1': a}, outputs=out)
  dt = (time.time() - t0) / 3
print(json.dumps({'workload': 'jacobi2d 8192x8192 iterate=100 via host arrays',
                  'seconds': dt, 'cells_iters_per_s_incl_pcie': 8192 * 8192 * 100 / dt}))
