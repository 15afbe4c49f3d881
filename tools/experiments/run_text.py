"""GPU kernels against the C oracle for .soda files given on the command line.
usage: run_text.py "EXTENT" "KW-dict" file.soda ..."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import fuzz
from oracle import c_oracle
from soda_amd import core, runtime
from soda_amd.codegen.hip import lower
extent, kw = eval(sys.argv[1]), eval(sys.argv[2])
for path in sys.argv[3:]:
    stencil = core.from_text(open(path).read())
    ins = fuzz.inputs_for(stencil, extent, 1)
    want = c_oracle.COracle(stencil, openmp=False).run(ins)
    try:
        with runtime.Program(stencil, lower.LowerOptions(**kw), extent=extent) as prog:
            got = prog.run(ins)
            names = [k.name for k in prog.module.kernels]
    except Exception as e:
        print(os.path.basename(path), 'ERR', str(e)[:200]); continue
    res = []
    for o in stencil.output_names:
        lo, hi = stencil.valid_box(extent, o)
        idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
        bad = got[o][idx] != want[o][idx]
        res.append((o, int(bad.sum()), int(bad.size)))
    print(os.path.basename(path), res, names[:1], flush=True)
