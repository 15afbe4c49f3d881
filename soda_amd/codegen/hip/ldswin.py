"""`ldswin`: wide 2-D windows through an LDS row ring.

For programs the register-marching kernels serve badly: ONE stage (after the
pointwise locals are folded) that taps ONE input over a window many cells wide
in dimension 0 -- contrast.soda (reference tests/src/contrast.soda: 17 x 17
taps, 393 operations per cell).  `march2d` brings a neighbour's cell into a
lane with a lane shift, so a window that reaches 16 cells costs 16 halo lanes
of every 64 (48 valid: a quarter of an issue-bound kernel's slots wasted) and
~200 lane moves per cell; `direct` reads every row fragment through the
texture path, which one CU's four SIMDs share (85 16-byte loads per 4 cells:
the L1 is the bound, 698 us).

Here the window rows live in LDS, written once per block and row:

  * a block is 8 (or 4) waves; wave w computes output row y + w of an 8-row
    step, lane l the 8 cells x0 + 8 l ... + 7: ALL 64 lanes valid, no lane ever
    shifts;
  * LDS holds a ring of `window height + 15` input rows of 512 + window width
    cells (rounded to 16 bytes).  A step reads, per thread and window row, the
    32 + window-width bytes its 8 cells tap with `ds_read_b128`s (consecutive
    lanes 32 bytes apart: conflict-free) -- 6 reads per row for contrast, shared
    by 8 cells, against 85 texture loads per 4 cells in `direct`;
  * while a step computes, every wave fetches ONE of the next step's input rows
    from global memory into registers (coalesced 16-byte loads, 1 KiB per
    instruction) and files it in the ring slot that went dead a step ago; one
    barrier per step orders both directions;
  * the expression is emitted operation-major for the 8 cells (ir.c_statements,
    as `direct` does), a window row's fragment fetched right before the first
    term that taps it: textual order, left-to-right association, no FMA -- the
    bits of the oracle.

Chunks of `chunk` rows per block along dimension 1; a chunk re-reads the
`window height - 1` rows above it (loads only, nothing is computed twice).
"""
from typing import Dict, List, Optional, Tuple

from soda_amd import core, ir, util

from soda_amd.codegen.hip.module import KernelDesc, Module, PassDesc

V = 8                   # cells per lane
STEP = 4                # rows per step = waves per block (the fallback)
WIDTH = 64 * V          # columns per block
MIN_OPS = 96            # below this a program is not compute-bound enough
MIN_SPAN = 5            # window cells along dimension 0 beyond which lanes are
                        # cheaper in LDS than shifted (2 halo lanes per side)
CHUNK = 128


def ldswin_supported(stencil: core.Stencil) -> Optional[str]:
  if stencil.dim != 2:
    return 'ldswin needs a 2-dimensional program'
  if len(stencil.ordered_stages) != 1 or len(stencil.input_names) != 1 or \
      len(stencil.output_names) != 1:
    return 'ldswin handles single-stage, single-input programs'
  stage = stencil.ordered_stages[0]
  if stencil.input_names[0] not in stage.taps:
    return 'ldswin: the stage reads no tensor'
  if stage.stmt.let:
    return 'ldswin does not handle let variables'
  if stencil.param_stmts:
    return 'ldswin does not handle param arrays'
  if stencil.preserve_border:
    return 'ldswin does not handle border: preserve'
  table = stencil.symbol_table
  if table[stencil.input_names[0]].size_in_bytes != 4 or \
      stage.haoda_type.size_in_bytes != 4:
    return 'ldswin handles 4-byte cells'
  return None


def ldswin_candidate(stencil: core.Stencil) -> bool:
  """Cheap test on the program AS WRITTEN, before `lower` folds its pointwise
  locals into one stage (which re-parses the program and duplicates a local
  per read): everything `ldswin_pays` will ask of the folded form that can
  already be read off the unfolded one."""
  if stencil.dim != 2 or stencil.iterate != 1 or \
      len(stencil.input_names) != 1 or len(stencil.output_names) != 1 or \
      stencil.param_stmts or stencil.preserve_border:
    return False
  table = stencil.symbol_table
  out = stencil.output_names[0]
  if table[stencil.input_names[0]].size_in_bytes != 4 or \
      table[out].size_in_bytes != 4:
    return False
  lo, hi = stencil.iteration_boxes()[out]
  work = sum(ir.op_count(s.stmt.expr) + sum(ir.op_count(l.expr)
                                            for l in s.stmt.let)
             for s in stencil.ordered_stages)
  return hi[0] - lo[0] >= MIN_SPAN and work >= MIN_OPS


def ldswin_pays(stencil: core.Stencil) -> bool:
  """Whether `auto` should pick it: a wide window and enough arithmetic."""
  if ldswin_supported(stencil):
    return False
  stage = stencil.ordered_stages[0]
  tlo, thi = stage.tap_bounds(stencil.input_names[0])
  span = thi[0] - tlo[0]
  work = ir.op_count(stage.stmt.expr)
  return span >= MIN_SPAN and work >= MIN_OPS and stencil.iterate == 1


def add_ldswin_pass(mod: Module, chunk: int = CHUNK,
                    step: Optional[int] = None) -> PassDesc:
  """`step` rows per step = waves per block; None: 8 where the ring of window
  height + 15 rows fits 80 KB of LDS (two blocks, four waves per SIMD), else 4.
  contrast 8192^2 (profiles/r04_contrast2.jsonl, us): 8 rows 512-524, 4 rows
  533-544, 6 rows 685-712 (six waves do not spread over four SIMDs), 2 rows
  739-936."""
  if step is None:
    try:
      return add_ldswin_pass(mod, chunk, 8)
    except util.SemanticError as e:
      if 'too tall' not in str(e):
        raise
      return add_ldswin_pass(mod, chunk, STEP)
  st = mod.stencil
  why = ldswin_supported(st)
  if why:
    raise util.SemanticError('ldswin: %s' % why)
  stage = st.ordered_stages[0]
  iname = st.input_names[0]
  table = st.symbol_table
  ct_in, ct_out = table[iname].c_type, stage.haoda_type.c_type
  tlo, thi = stage.tap_bounds(iname)          # offsets relative to the cell
  xl, xh, yl, yh = tlo[0], thi[0], tlo[1], thi[1]
  wy = yh - yl                                # window rows - 1
  # the box one iteration computes (every load inside the grid, and the cell)
  blo = (max(0, -xl), max(0, -yl))
  bhi = (max(0, xh), max(0, yh))
  # quads (4 cells, 16 bytes) of a ring row, relative to the block's column 0:
  # lane l taps cells 8 l + xl ... 8 l + 7 + xh = quads 2 l + klo ... 2 l + khi
  klo = xl // 4
  khi = (V - 1 + xh) // 4
  nquad = 2 * 63 + khi - klo + 1               # quads klo ... 126 + khi
  halo = nquad - 128                           # quads beyond the 128 own ones
  if halo > 64 or klo < -32:
    raise util.SemanticError('ldswin: window too wide')
  # A ring row keeps its even quads in one region and its odd quads in another
  # (`odd0` cells further on): lane l reads quads 2 l + k, so consecutive lanes
  # read consecutive 16-byte quads of ONE region -- conflict-free, where the
  # plain layout had lanes 32 bytes apart and two lanes of every 16 on the
  # same banks (round 4, SQ counters of the plain layout: 4.7e7 of 9.9e7 LDS
  # cycles were bank conflicts).  The odd region starts 128 bytes off a
  # 256-byte boundary, so a filing instruction's even and odd lanes (which
  # write the two regions at equal offsets) miss each other's banks as well.
  half = (nquad + 1) // 2                      # quads per region
  odd0 = half * 4                              # cells
  while (odd0 * 4) % 256 != 128:
    odd0 += 4
  pitch = odd0 + half * 4                      # cells
  live = wy + step                             # rows a step reads
  ring = live + step                           # + the rows it files
  lds_bytes = ring * pitch * 4
  if lds_bytes > 80 * 1024:      # two blocks per CU at least
    raise util.SemanticError('ldswin: window too tall for LDS')
  frag = (khi - klo + 1) * 4                   # cells a thread reads per row
  name = '%s_ldswin_V%d_S%d_R%d_C%d' % (st.app_name, V, step, ring, chunk)
  L: List[str] = []
  w = L.append
  w('// ldswin: %d x %d cells per step and block, ring of %d rows x %d cells in '
    'LDS (%d bytes); window x %d..%d, y %d..%d' %
    (WIDTH, step, ring, pitch, lds_bytes, xl, xh, yl, yh))
  w('// stage `%s`: %s' % (stage.name, ' '.join(str(stage.stmt).split())[:300]))
  w('extern "C" __global__ void __launch_bounds__(%d) %s(soda_hip_kargs_t a) {'
    % (64 * step, name))
  w('  __shared__ __attribute__((aligned(16))) %s ring[%d * %d];' %
    (ct_in, ring, pitch))
  w('  const int lane = (int)(threadIdx.x & 63u);')
  w('  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));')
  w('  const unsigned nblk = gridDim.x;')
  w('  const unsigned bid = (nblk % 8u == 0u) ? (blockIdx.x % 8u) * (nblk / 8u)'
    ' + blockIdx.x / 8u : blockIdx.x;')
  w('  const int x0 = (int)(bid %% (unsigned)a.ntile[0]) * %d;' % WIDTH)
  w('  const int yb = (int)(bid / (unsigned)a.ntile[0]) * a.tile[1];')
  w('  const int n0 = a.extent[0], n1 = a.extent[1];')
  w('  const int yend = min(yb + a.tile[1], n1 - %d);' % bhi[1])
  w('  const int ybeg = max(yb, %d);' % blo[1])
  w('  if (ybeg >= yend) return;')
  w('  const int64_t pitch_g = a.stride[1];')
  w('  const %s* __restrict__ in = (const %s*)a.buf[%d];' %
    (ct_in, ct_in, mod.slot[iname]))
  w('  %s* __restrict__ out = (%s*)a.buf[%d];' %
    (ct_out, ct_out, mod.slot[stage.name]))
  # ---- filing one input row: quads 64 f + lane (+ the halo quads) ----------
  nload = 2 + (1 if halo > 0 else 0)
  w('  // row `y` of the input -> registers (zeros outside the grid)')
  w('  auto fetch = [&](int y, %s (&q)[%d][4]) {' % (ct_in, nload))
  w('    const bool row_ok = y >= 0 && y < n1;')
  for f in range(nload):
    quad = '%d + lane' % (klo + 64 * f)
    cond = 'row_ok && xq >= 0 && xq + 3 < n0'
    if f == 2:
      cond += ' && lane < %d' % halo
    w('    { const int xq = x0 + 4 * (%s);' % quad)
    w('      if (%s) soda_load_frag<%s, 4, false>(q[%d], in + (int64_t)y * '
      'pitch_g + xq);' % (cond, ct_in, f))
    w('      else soda_zero_frag<%s, 4>(q[%d]); }' % (ct_in, f))
  w('  };')
  w('  auto file = [&](int slot, const %s (&q)[%d][4]) {' % (ct_in, nload))
  for f in range(nload):
    guard = 'if (lane < %d) ' % halo if f == 2 else ''
    # quad 64 f + lane of the row: region by its parity, place by its half
    w('    %ssoda_store_frag<%s, 4>(&ring[slot * %d + (lane & 1) * %d + '
      '4 * ((%d + lane) >> 1)], q[%d]);' % (guard, ct_in, pitch, odd0, 64 * f, f))
  w('  };')
  # ---- prologue: the rows the first step reads ------------------------------
  w('  // ring slot of input row r: (r - first row) mod %d, kept incrementally'
    % ring)
  w('  const int r0 = ybeg + (%d);      // first input row of the chunk' % yl)
  w('  for (int i = wave; i < %d; i += %d) {' % (live, step))
  w('    %s q[%d][4];' % (ct_in, nload))
  w('    fetch(r0 + i, q);')
  w('    file(i, q);')
  w('  }')
  w('  int base = 0;                    // slot of the step\'s first input row')
  w('  const int xc = x0 + lane * %d;' % V)
  w('  for (int y = ybeg; y < yend; y += %d) {' % step)
  w('    __syncthreads();')
  w('    // the next %d rows travel while this step computes' % step)
  w('    %s nq[%d][4];' % (ct_in, nload))
  w('    fetch(y + (%d) + %d + wave, nq);' % (yl, live))
  w('    const int yo = y + wave;')
  w('    if (yo < yend) {')
  # ---- the expression, operation-major, fragments fetched on first use ------
  body: List[str] = []
  rows: Dict[int, str] = {}

  def mk_load(e: int):
    def load(ref: ir.Ref) -> str:
      dx = ref.idx[0] - stage.st_idx[0]
      dy = ref.idx[1] - stage.st_idx[1]
      if dy not in rows:
        var = 'rw%d' % len(rows)
        body.append('%s %s[%d];' % (ct_in, var, frag))
        body.append('{ int s = base + wave + (%d); if (s >= %d) s -= %d;' %
                    (dy - yl, ring, ring))
        body.append('  const %s* p = &ring[s * %d + lane * 4];' %
                    (ct_in, pitch))
        for k in range(khi - klo + 1):    # quad 2 lane + k of the row
          body.append('  soda_load_frag<%s, 4, false>(*(%s(*)[4])&%s[%d], '
                      'p + %d);' % (ct_in, ct_in, var, 4 * k,
                                    (odd0 if k % 2 else 0) + 4 * (k // 2)))
        body.append('}')
        rows[dy] = var
      # cell e of the lane taps column 8 l + e + dx = fragment cell e + dx - 4 klo
      return '%s[%d]' % (rows[dy], e + dx - 4 * klo)
    return load

  counter = [0]

  def fresh() -> str:
    counter[0] += 1
    return 'v%d' % counter[0]

  # (Literal operands make half of the step's instructions 8 bytes long; the
  # same constants held in registers -- 15 KB of code instead of 20 -- ran 5 %
  # SLOWER, 118 VGPRs: the step is not bound by instruction fetch,
  # profiles/r04_contrast3.jsonl.)
  _, results = ir.c_statements(stage.stmt.expr, [mk_load(e) for e in range(V)],
                               fresh, stmts=body)
  L.extend('      ' + x for x in body)
  w('      %s res[%d];' % (ct_out, V))
  for e, r in enumerate(results):
    w('      res[%d] = (%s)(%s);' % (e, ct_out, r))
  # ---- store: whole fragments inside the box, cell by cell at its edges -----
  w('      %s* o = out + (int64_t)yo * pitch_g + xc;' % ct_out)
  w('      if (xc >= %d && xc + %d < n0 - %d) {' % (blo[0], V - 1, bhi[0]))
  w('        soda_store_frag<%s, 4>(o, *(const %s(*)[4])&res[0]);' %
    (ct_out, ct_out))
  w('        soda_store_frag<%s, 4>(o + 4, *(const %s(*)[4])&res[4]);' %
    (ct_out, ct_out))
  w('      } else {')
  w('        _Pragma("unroll") for (int e = 0; e < %d; ++e)' % V)
  w('          if (xc + e >= %d && xc + e < n0 - %d) o[e] = res[e];' %
    (blo[0], bhi[0]))
  w('      }')
  w('    }')
  w('    // file the fetched rows where the rows this step read first went dead')
  w('    { int s = base + %d + wave; if (s >= %d) s -= %d; file(s, nq); }' %
    (live, ring, ring))
  w('    base += %d; if (base >= %d) base -= %d;' % (step, ring, ring))
  w('  }')
  w('}')
  idx = mod.add_kernel(
      KernelDesc(name, (64 * step, 1, 1), (WIDTH, chunk), lds_bytes=0,
                 note='ldswin', tune=dict(vec=4)), '\n'.join(L) + '\n')
  p = PassDesc(1, [idx], 'ldswin',
               dict(bytes_per_cell_min=8, lds_bytes=lds_bytes))
  mod.passes.append(p)
  return p
