import sys, os, ast
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import fuzz
from oracle import c_oracle
from soda_amd import core, runtime, util
from soda_amd.codegen.hip import lower
seed, extent = int(sys.argv[1]), ast.literal_eval(sys.argv[2])
variants = [eval(v) for v in sys.argv[3:]] or [dict(fuse=(2,))]
text, dim, _ = fuzz.program(seed)
stencil = core.from_text(text)
ins = fuzz.inputs_for(stencil, extent, seed)
want = c_oracle.COracle(stencil, openmp=False).run(ins)
print('flags', os.environ.get('SODA_HIP_EXTRA_FLAGS'))
for kw in variants:
    try:
        with runtime.Program(stencil, lower.LowerOptions(**kw), extent=extent) as prog:
            got = prog.run(ins)
            names = [k.name for k in prog.module.kernels]
    except Exception as e:
        print(kw, 'ERR', str(e)[:100]); continue
    out = []
    for o in stencil.output_names:
        lo, hi = stencil.valid_box(extent, o)
        idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
        bad = got[o][idx] != want[o][idx]
        w = np.argwhere(bad)
        out.append((o, int(bad.sum()), sorted(set((w[:, 0] + lo[-1]).tolist()))[:8] if len(w) else None, sorted(set((w[:, 1] + lo[0]).tolist()))[:10] if len(w) else None))
    print(kw, out, names[:1], flush=True)
