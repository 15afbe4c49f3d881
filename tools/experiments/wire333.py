import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import fuzz
from oracle import frt_layout, numpy_oracle
from soda_amd import core, stream
text, dim, _ = fuzz.program(333)
stencil = core.from_text(text)
for extent in [(122, 27), (32, 27), (61, 27)]:
    ins = fuzz.inputs_for(stencil, extent, 333)
    layout = stream.WireLayout(stencil, extent)
    in_banks = frt_layout.scatter(layout, ins)
    ref = {o: np.zeros(tuple(extent[::-1]), np.dtype(t.np_name)) for o, t in zip(stencil.output_names, stencil.output_types)}
    frt_layout.gather(layout, frt_layout.kernel_on_streams(layout, in_banks), ref)
    want = numpy_oracle.run(stencil, ins)
    boxes = [stencil.valid_box(extent, o) for o in stencil.output_names]
    lo = [max(b[0][d] for b in boxes) for d in range(dim)]
    hi = [min(b[1][d] for b in boxes) for d in range(dim)]
    idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
    for mode in ('dense', 'linear'):
        out_banks = frt_layout.alloc(layout, stencil.output_names)
        prog = stream.StreamProgram(stencil, dense=mode != 'linear')
        try:
            prog.run_banked_host(out_banks, in_banks, layout.cycle_count)
            used = prog.last_mode
        finally:
            prog.close()
        got = {o: np.zeros_like(ref[o]) for o in ref}
        frt_layout.gather(layout, out_banks, got)
        for o in stencil.output_names:
            b = got[o][idx] != ref[o][idx]
            w = np.argwhere(b)
            b2 = ref[o][idx] != want[o][idx]
            print(extent, 'tiles', layout.tiles, mode, 'ran', used, o, 'gpu!=restatement', int(b.sum()),
                  'cols', sorted(set((w[:, 1] + lo[0]).tolist()))[:12] if len(w) else None,
                  'rows', sorted(set((w[:, 0] + lo[1]).tolist()))[:6] if len(w) else None,
                  '| restatement!=nD', int(b2.sum()), 'stencil_offset', layout.stencil_offset, flush=True)
