import sys, os
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import numpy as np
import fuzz
from oracle import c_oracle
from soda_amd import core, runtime, util
from soda_amd.codegen.hip import lower

def run(seed, kind, iterate, border, extent, slabs, every, fuse, threads, overlap):
    text, dim, _ = (fuzz.window_program(seed) if kind == 'window' else fuzz.program(seed, rich=kind == 'rich'))
    stencil = core.from_text(text, iterate=iterate, **({'border': border} if border else {}))
    ins = fuzz.inputs_for(stencil, extent, seed)
    want = c_oracle.COracle(stencil, openmp=False).run(ins)
    def cmp(got, tag):
        for o in stencil.output_names:
            if border: g,w = got[o], want[o]
            else:
                lo, hi = stencil.valid_box(extent, o)
                idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
                g, w = got[o][idx], want[o][idx]
            bad = ~((g == w) | (np.isnan(g) & np.isnan(w)))
            n = int(bad.sum())
            where = np.argwhere(bad)
            print(tag, o, 'differ', n, 'first', where[:3].tolist(), 'last', where[-3:].tolist(), 'box', stencil.valid_box(extent, o) if not border else None, flush=True)
    with runtime.Program(stencil, lower.LowerOptions(fuse=fuse), extent=extent) as prog:
        cmp(prog.run(ins), 'single')
    print('reach', stencil.reach_along(stencil.dim-1))
    for s_, e_, t_, o_ in [(slabs, every, threads, overlap), (slabs, every, False, overlap), (slabs, every, False, False), (2, every, False, False), (slabs, 1, False, False), (slabs, 0, False, True)]:
        try:
            with runtime.Group(stencil, extent, [0]*s_, lower.LowerOptions(fuse=fuse), exchange_every=e_, overlap=o_, threads=t_) as group:
                got = group.run_host(ins)
                st = group.stats()
                infos = [(group.slab(i).begin, group.slab(i).end, group.slab(i).own_begin, group.slab(i).own_end, group.slab(i).ghost_lo, group.slab(i).ghost_hi) for i in range(s_)]
            cmp(got, 'group slabs=%d every=%d threads=%s overlap=%s' % (s_, e_, t_, o_))
            print('   stats', st, infos)
        except util.SodaError as e:
            print('group', s_, e_, 'error', e)

run(263, 'plain', 4, None, (130, 11, 127), 6, 2, (3, 2), True, True)
run(78, 'rich', 6, 'preserve', (130, 20, 82), 4, 3, (2,), False, True)
