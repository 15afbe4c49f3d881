"""Re-runs one tools/fuzz_scan.py `options` failure with one knob at a time put
back to its default.  usage: triage_one.py SEED KIND "EXTENT" "KW-dict" """
import sys, os, ast
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import fuzz
from oracle import c_oracle
from soda_amd import core, runtime, util
from soda_amd.codegen.hip import lower

seed, kind = int(sys.argv[1]), sys.argv[2]
extent, kw = ast.literal_eval(sys.argv[3]), ast.literal_eval(sys.argv[4])
text, dim, _ = (fuzz.window_program(seed) if kind == 'window' else fuzz.program(seed, rich=kind == 'rich'))
border = os.environ.get('TRIAGE_BORDER')
stencil = core.from_text(text, **({'border': border} if border else {}))
ins = fuzz.inputs_for(stencil, extent, seed)
want = c_oracle.COracle(stencil, openmp=False).run(ins)

def check(kw, tag):
    try:
        with runtime.Program(stencil, lower.LowerOptions(**kw), extent=extent) as prog:
            got = prog.run(ins)
            names = [k.name for k in prog.module.kernels]
    except Exception as e:
        print(tag, 'ERR', str(e)[:100]); return
    out = []
    for o in stencil.output_names:
        lo, hi = stencil.valid_box(extent, o)
        if border:
            lo, hi = (0,) * len(extent), tuple(extent)
        idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
        bad = ~((got[o][idx] == want[o][idx]) | ((got[o][idx] != got[o][idx]) & (want[o][idx] != want[o][idx])))
        w = np.argwhere(bad)
        out.append((o, int(bad.sum()), (w.min(0) + np.array(lo[::-1])).tolist() if len(w) else None,
                    (w.max(0) + np.array(lo[::-1])).tolist() if len(w) else None,
                    sorted(set((w[:, 0] + lo[-1]).tolist()))[:12] if len(w) else None))
    print(tag, out, names[:1], flush=True)

check(kw, 'as scanned')
check(kw, 'again     ')
for k in kw:
    if kw[k] not in (None,) and k != 'fuse':
        d = dict(kw); d[k] = None if k not in ('waves_y', 'waves_x') else 1
        check(d, 'default %-10s' % k)
check(dict(kw, fuse=(3,)), 'fuse (3,)')
check(dict(kw, fuse=(2,)), 'fuse (2,)')
check(dict(kw, fuse=()), 'fuse ()')
