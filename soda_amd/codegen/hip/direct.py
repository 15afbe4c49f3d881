"""`direct` kernels: one kernel per stage, parents read from global memory --
the correctness baseline and the fallback of lower.lower (see lower.py)."""
from typing import Dict, List, Tuple

from soda_amd import core, ir

from soda_amd.codegen.hip.module import (KernelDesc, Module, PassDesc, _COORDS)

# ---------------------------------------------------------------------------
# direct kernels
# ---------------------------------------------------------------------------

DIRECT_BLOCK = 256


def _interior(st: core.Stencil, stage: core.Stage):
  """(lo, hi, keep): cells at least `lo[d]` from the low and `hi[d]` from the
  high end of every dimension of the arrays can be computed (their taps lie
  in the arrays); the others get 0 -- or, for an output under `border:
  preserve`, the value of input `keep` (core.py check_preserve), as do the
  cells outside _global_interior()."""
  dim = st.dim
  lo = [0] * dim
  hi = [0] * dim
  for parent in stage.taps:
    tlo, thi = stage.tap_bounds(parent)
    for d in range(dim):
      lo[d] = max(lo[d], -tlo[d])
      hi[d] = max(hi[d], thi[d])
  keep = None
  if st.preserve_border and stage.is_output:
    keep = st.preserved_from(stage.name)
  return lo, hi, keep


def _global_interior(st: core.Stencil, stage: core.Stage, coord) -> List[str]:
  """border: preserve -- C conditions that a cell lies inside the box ONE
  iteration can compute on the GLOBAL grid (the arrays may be one GPU's slab:
  kargs origin / gextent).  `coord(d)` spells the local coordinate."""
  wlo, whi = st.interior_bounds(stage.name)
  conds = []
  for d in range(st.dim):
    if wlo[d] < 0:
      conds.append('a.origin[%d] + %s >= %d' % (d, coord(d), -wlo[d]))
    if whi[d] > 0:
      conds.append('a.origin[%d] + %s < a.gextent[%d] - %d' %
                   (d, coord(d), d, whi[d]))
  return conds


def _wide(table, name: str, text: str) -> str:
  """A one-byte cell enters an expression as the int C promotes it to, its
  value range hidden from the compiler (soda_rt.h soda_wide: hipcc's
  packed-byte instruction selection is wrong in places)."""
  return 'soda_wide(%s)' % text if table[name].size_in_bytes == 1 else text


def _direct_rows_kernel(mod: Module, stage: core.Stage, name: str,
                        vec: int) -> Tuple[List[str], Tuple[int, ...]]:
  """`direct` with `vec` cells per thread: every row of a parent the stage taps
  is fetched ONCE per thread into registers (span of the taps + vec - 1 cells)
  and shared by the thread's cells; the expression is emitted operation-major
  (ir.c_statements) so a row is fetched right before its first use and dies
  after its last.  contrast (17 x 17 taps): 289 loads per cell -> <= 18 per
  row per 4 cells.  Threads whose cells are not all interior along dimension 0
  take the one-cell-at-a-time path; results are those of the scalar kernel."""
  st = mod.stencil
  dim = st.dim
  table = st.symbol_table
  parents = list(stage.taps)
  lo, hi, keep = _interior(st, stage)
  if keep is not None and keep not in parents:
    parents.append(keep)
  ct = stage.haoda_type.c_type
  V = vec
  L = [
      '// stage `%s`, %d cells per thread: %s' %
      (stage.name, V, ' '.join(str(stage.stmt).split())[:400]),
      'extern "C" __global__ void __launch_bounds__(%d) %s(soda_hip_kargs_t a) {'
      % (DIRECT_BLOCK, name),
      '  unsigned b = blockIdx.x;',
      '  const int %s = ((int)(b %% (unsigned)a.ntile[0]) * %d + '
      '(int)threadIdx.x) * %d;' % (_COORDS[0], DIRECT_BLOCK, V),
      '  b /= (unsigned)a.ntile[0];',
  ]
  for d in range(1, dim):
    L.append('  const int %s = (int)(b %% (unsigned)a.ntile[%d]); '
             'b /= (unsigned)a.ntile[%d];' % (_COORDS[d], d, d))
  L.append('  if (%s >= a.extent[0]) return;' % _COORDS[0])
  L.append('  %s* __restrict__ out = (%s*)a.buf[%d];' %
           (ct, ct, mod.slot[stage.name]))
  for parent in parents:
    pt = table[parent].c_type
    L.append('  const %s* __restrict__ in_%s = (const %s*)a.buf[%d];' %
             (pt, parent, pt, mod.slot[parent]))
  L.extend(mod.param_decls(stage))
  L.append('  const int64_t soda_o = %s;' % ' + '.join(
      ['(int64_t)%s' % _COORDS[0]] +
      ['(int64_t)%s * a.stride[%d]' % (_COORDS[d], d) for d in range(1, dim)]))
  outer = []
  for d in range(1, dim):
    if lo[d]:
      outer.append('%s >= %d' % (_COORDS[d], lo[d]))
    if hi[d]:
      outer.append('%s < a.extent[%d] - %d' % (_COORDS[d], d, hi[d]))
  L.append('  %s soda_r[%d];' % (ct, V))
  if keep is None:
    L.append('  soda_zero_frag<%s, %d>(soda_r);' % (ct, V))
  else:    # border: preserve -- cells that are not computed keep the input
    L.append('  soda_load_frag<%s, %d, false>(soda_r, in_%s + soda_o);' %
             (ct, V, keep))
  glob = []      # border: preserve -- the box on the GLOBAL grid
  if keep is not None:
    glob = _global_interior(st, stage, lambda d: _COORDS[d])
    outer += [c for c in glob if 'origin[0]' not in c]
  L.append('  const bool soda_rows_ok = %s;' %
           (' && '.join(outer) if outer else 'true'))
  first_ok = ['%s >= %d' % (_COORDS[0], lo[0]),
              '%s + %d < a.extent[0] - %d' % (_COORDS[0], V - 1, hi[0])]
  for c in glob:
    if 'origin[0]' in c:      # all V cells: the first for >=, the last for <
      first_ok.append(c.replace('+ %s <' % _COORDS[0],
                                '+ %s + %d <' % (_COORDS[0], V - 1)))
  L.append('  if (soda_rows_ok && %s) {' % ' && '.join(first_ok))

  # ---- all cells interior: shared row buffers --------------------------------
  # A row's taps far apart along dimension 0 (the linearised 1-D form of a
  # wire stream: every tap of the n-D program lies in ONE row, whole tile rows
  # apart) are fetched as separate runs -- one buffer over the whole span would
  # be a tile row or more per thread (512 x 512 tiles: 2 MB of stack, refused
  # by the compiler).
  rows: Dict[Tuple, Tuple[str, int]] = {}
  taps0: Dict[Tuple[str, Tuple[int, ...]], set] = {}
  for ref in ir.get_loads(stage.stmt.expr):
    if ref.name in st.param_names:
      continue
    off = tuple(a - b for a, b in zip(ref.idx, stage.st_idx))
    taps0.setdefault((ref.name, off[1:]), set()).add(off[0])
  gap = max(V, 8)
  run_of: Dict[Tuple, int] = {}
  spans: Dict[Tuple, List[int]] = {}
  for (pname, rest), offs in taps0.items():
    run, last = 0, None
    for o in sorted(offs):
      if last is not None and o - last > gap:
        run += 1
      last = o
      run_of[(pname, rest, o)] = run
      sp = spans.setdefault((pname, rest, run), [o, o])
      sp[1] = o
  body: List[str] = []

  def mk_load(e: int):
    def load(ref: ir.Ref) -> str:
      prm = mod.param_load(ref)
      if prm is not None:
        return prm
      off = tuple(a - b for a, b in zip(ref.idx, stage.st_idx))
      key = (ref.name, off[1:], run_of[(ref.name, off[1:], off[0])])
      if key not in rows:
        mn, mx = spans[key]
        var = 'rb%d_%s' % (len(rows), ref.name)
        n = mx - mn + V
        pt = table[ref.name].c_type
        terms = ['soda_o', '(%d)' % mn]
        for d in range(1, dim):
          if off[d]:
            terms.append('(%d) * a.stride[%d]' % (off[d], d))
        body.append('%s %s[%d];' % (pt, var, n))
        body.append('{ const %s* __restrict__ soda_p = in_%s + (%s);' %
                    (pt, ref.name, ' + '.join(terms)))
        body.append('  _Pragma("unroll") for (int i = 0; i < %d; ++i) '
                    '%s[i] = soda_p[i]; }' % (n, var))
        rows[key] = (var, mn)
      var, mn = rows[key]
      return _wide(table, ref.name, '%s[%d]' % (var, off[0] - mn + e))
    return load

  counter = [0]

  def fresh() -> str:
    counter[0] += 1
    return 'v%d' % counter[0]

  _, results = ir.c_statements(stage.stmt.expr, [mk_load(e) for e in range(V)],
                               fresh, var=mod.param_var, stmts=body)
  L.extend('    ' + x for x in body)
  for e, r in enumerate(results):
    L.append('    soda_r[%d] = (%s)(%s);' % (e, ct, r))
  L.append('  } else if (soda_rows_ok) {')

  # ---- strip ends: one cell at a time, as the scalar kernel ------------------
  def scalar_load(ref: ir.Ref) -> str:
    prm = mod.param_load(ref)
    if prm is not None:
      return prm
    terms = ['soda_o', 'e']
    for d in range(dim):
      off = ref.idx[d] - stage.st_idx[d]
      if off:
        terms.append('(%d)' % off if d == 0 else
                     '(%d) * a.stride[%d]' % (off, d))
    return _wide(table, ref.name, 'in_%s[%s]' % (ref.name, ' + '.join(terms)))

  L.append('    _Pragma("unroll") for (int e = 0; e < %d; ++e) {' % V)
  cell_ok = ['%s + e >= %d' % (_COORDS[0], lo[0]),
             '%s + e < a.extent[0] - %d' % (_COORDS[0], hi[0])]
  cell_ok += [c.replace('+ %s ' % _COORDS[0], '+ %s + e ' % _COORDS[0])
              for c in glob if 'origin[0]' in c]
  L.append('      if (%s)' % ' && '.join(cell_ok))
  L.append('        soda_r[e] = (%s)(%s);' %
           (ct, ir.c_expr(stage.stmt.expr, scalar_load, mod.param_var)))
  L.append('    }')
  L.append('  }')
  L.append('  soda_store_frag<%s, %d, false>(out + soda_o, soda_r);' % (ct, V))
  L.append('}')
  return L, (DIRECT_BLOCK * V,) + (1,) * (dim - 1)


def add_direct_pass(mod: Module, vec: int = 1) -> PassDesc:
  st = mod.stencil
  dim = st.dim
  table = st.symbol_table
  kernel_ids = []
  for stage in st.ordered_stages:
    name = '%s_direct_%s' % (st.app_name, stage.name)
    if vec > 1 and not stage.stmt.let and stage.taps:
      lines, tile = _direct_rows_kernel(mod, stage, name + '_v%d' % vec, vec)
      kernel_ids.append(mod.add_kernel(
          KernelDesc(name + '_v%d' % vec, (DIRECT_BLOCK, 1, 1), tile,
                     note='direct rows V%d' % vec, tune=dict(vec=vec)),
          '\n'.join(lines) + '\n'))
      continue
    parents = list(stage.taps)
    lo, hi, keep = _interior(st, stage)
    if keep is not None and keep not in parents:
      parents.append(keep)
    lines = [
        '// stage `%s`: %s' % (stage.name,
                               ' '.join(str(stage.stmt).split())),
        'extern "C" __global__ void __launch_bounds__(%d) %s(soda_hip_kargs_t a) {'
        % (DIRECT_BLOCK, name),
        '  unsigned b = blockIdx.x;',
        '  const int %s = (int)(b %% (unsigned)a.ntile[0]) * %d + (int)threadIdx.x;'
        % (_COORDS[0], DIRECT_BLOCK),
        '  b /= (unsigned)a.ntile[0];',
    ]
    for d in range(1, dim):
      lines.append('  const int %s = (int)(b %% (unsigned)a.ntile[%d]); '
                   'b /= (unsigned)a.ntile[%d];' % (_COORDS[d], d, d))
    lines.append('  if (%s >= a.extent[0]) return;' % _COORDS[0])
    ct = stage.haoda_type.c_type
    lines.append('  %s* __restrict__ out = (%s*)a.buf[%d];' %
                 (ct, ct, mod.slot[stage.name]))
    for parent in parents:
      pt = table[parent].c_type
      lines.append('  const %s* __restrict__ in_%s = (const %s*)a.buf[%d];' %
                   (pt, parent, pt, mod.slot[parent]))
    lines.extend(mod.param_decls(stage))
    lines.append('  const int64_t soda_o = %s;' % ' + '.join(
        ['(int64_t)%s' % _COORDS[0]] +
        ['(int64_t)%s * a.stride[%d]' % (_COORDS[d], d) for d in range(1, dim)]))
    conds = []
    for d in range(dim):
      if lo[d]:
        conds.append('%s >= %d' % (_COORDS[d], lo[d]))
      if hi[d]:
        conds.append('%s < a.extent[%d] - %d' % (_COORDS[d], d, hi[d]))
    if keep is None:
      lines.append('  %s soda_r = (%s)0;' % (ct, ct))
    else:   # border: preserve -- cells that are not computed keep the input
      lines.append('  %s soda_r = in_%s[soda_o];' % (ct, keep))
      conds += _global_interior(st, stage, lambda d: _COORDS[d])
    lines.append('  if (%s) {' % (' && '.join(conds) if conds else 'true'))

    def load(ref: ir.Ref, _stage=stage) -> str:
      prm = mod.param_load(ref)
      if prm is not None:
        return prm
      terms = ['soda_o']
      for d in range(dim):
        off = ref.idx[d] - _stage.st_idx[d]
        if off:
          terms.append('(%d)' % off if d == 0 else
                       '(%d) * a.stride[%d]' % (off, d))
      return _wide(table, ref.name, 'in_%s[%s]' % (ref.name, ' + '.join(terms)))

    for let in stage.stmt.let:
      lines.append('    const %s %s = %s;' %
                   (let.haoda_type.c_type, let.name,
                    ir.c_expr(let.expr, load, mod.param_var)))
    lines.append('    soda_r = (%s)(%s);' %
                 (ct, ir.c_expr(stage.stmt.expr, load, mod.param_var)))
    lines.append('  }')
    lines.append('  out[soda_o] = soda_r;')
    lines.append('}')
    kernel_ids.append(
        mod.add_kernel(
            KernelDesc(name, (DIRECT_BLOCK, 1, 1), (DIRECT_BLOCK,) + (1,) *
                       (dim - 1), note='direct'), '\n'.join(lines) + '\n'))
  # traffic: every stage reads its parents once (ideal caching) and writes once
  p = PassDesc(1, kernel_ids, 'direct')
  mod.passes.append(p)
  return p


