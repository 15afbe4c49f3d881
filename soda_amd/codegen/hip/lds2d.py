"""`lds2d`: the classic LDS halo tile, kept as the measured alternative to the
register-window kernels."""
from typing import List, Optional

from soda_amd import core, ir, util

from soda_amd.codegen.hip.module import KernelDesc, Module, PassDesc

# ---------------------------------------------------------------------------
# lds2d: the classic LDS halo tile (kept as the measured alternative)
# ---------------------------------------------------------------------------

LDS2D_TILE_ROWS = 32


def lds2d_supported(stencil: core.Stencil) -> Optional[str]:
  if stencil.dim != 2:
    return 'lds2d needs a 2-dimensional program'
  if len(stencil.ordered_stages) != 1 or len(stencil.input_names) != 1:
    return 'lds2d handles single-stage, single-input programs'
  if stencil.ordered_stages[0].stmt.let:
    return 'lds2d does not handle let variables'
  return None


def add_lds2d_pass(mod: Module, tile_rows: int = LDS2D_TILE_ROWS,
                   nt_load: bool = True) -> PassDesc:
  """One iteration per launch, the textbook way: a 256-thread block stages a
  (tile_rows + halo) x (256 + halo) input tile in LDS with coalesced 16-byte
  loads, synchronises, and every thread computes 4 consecutive cells per row
  from LDS.  The north star names this design; it is generated so that the
  choice between it and the register-marching kernel is a MEASUREMENT
  (tools/sweep.py --strategy lds; profiles/r01_sweeps.md), not an assertion."""
  st = mod.stencil
  why = lds2d_supported(st)
  if why:
    raise util.SemanticError('lds2d: %s' % why)
  stage = st.ordered_stages[0]
  iname = st.input_names[0]
  tlo, thi = stage.tap_bounds(iname)
  rxl, rxh = max(0, -tlo[0]), max(0, thi[0])
  ryl, ryh = max(0, -tlo[1]), max(0, thi[1])
  if max(rxl, rxh) > 4:
    raise util.SemanticError('lds2d: x radius above 4')
  table = st.symbol_table
  ct_in, ct_out = table[iname].c_type, stage.haoda_type.c_type
  V = 4
  width = 64 * V
  pitch = width + 8                       # 4 halo cells each side, 16-B aligned
  rows = tile_rows + ryl + ryh
  name = '%s_lds2d_T1_R%d%s' % (st.app_name, tile_rows, '_ntl' if nt_load else '')
  L: List[str] = []
  w = L.append
  w('// lds2d: %dx%d output tile per 256-thread block, %d x %d cells staged in'
    ' LDS' % (width, tile_rows, rows, pitch))
  w('extern "C" __global__ void __launch_bounds__(256) %s(soda_hip_kargs_t a) {'
    % name)
  w('  __shared__ __attribute__((aligned(16))) %s tile[%d][%d];' %
    (ct_in, rows, pitch))
  w('  const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);')
  w('  const unsigned nblk = gridDim.x;')
  w('  const unsigned bid = (nblk % 8u == 0u) ? (blockIdx.x % 8u) * (nblk / 8u)'
    ' + blockIdx.x / 8u : blockIdx.x;')
  w('  const int x0 = (int)(bid %% (unsigned)a.ntile[0]) * %d;' % width)
  w('  const int y0 = (int)(bid / (unsigned)a.ntile[0]) * %d;' % tile_rows)
  w('  const int n0 = a.extent[0], n1 = a.extent[1];')
  w('  const int64_t pitch_g = a.stride[1];')
  w('  const %s* __restrict__ in = (const %s*)a.buf[%d];' %
    (ct_in, ct_in, mod.slot[iname]))
  w('  %s* __restrict__ out = (%s*)a.buf[%d];' %
    (ct_out, ct_out, mod.slot[stage.name]))
  w('  const int x = x0 + lane * %d;' % V)
  w('  const bool lane_ok = x + %d <= n0;' % V)
  # stage the tile: waves take rows round-robin
  w('  for (int r = wave; r < %d; r += 4) {' % rows)
  w('    const int y = y0 - %d + r;' % ryl)
  w('    const bool row_ok = y >= 0 && y < n1;')
  w('    %s v[%d];' % (ct_in, V))
  w('    if (row_ok && lane_ok) soda_load_frag<%s, %d, %s>(v, in + (int64_t)y * '
    'pitch_g + x);' % (ct_in, V, 'true' if nt_load else 'false'))
  w('    else soda_zero_frag<%s, %d>(v);' % (ct_in, V))
  w('    soda_store_frag<%s, %d>(&tile[r][4 + lane * %d], v);' % (ct_in, V, V))
  if rxl or rxh:
    w('    if (lane < %d) {  // left halo' % max(rxl, 1))
    w('      const int hx = x0 - 1 - lane;')
    w('      tile[r][3 - lane] = (row_ok && hx >= 0 && lane < %d) ? '
      'in[(int64_t)y * pitch_g + hx] : (%s)0;' % (rxl, ct_in))
    w('    } else if (lane >= 60 && lane < 60 + %d) {  // right halo' %
      max(rxh, 1))
    w('      const int hx = x0 + %d + (lane - 60);' % width)
    w('      tile[r][4 + %d + (lane - 60)] = (row_ok && hx < n0) ? '
      'in[(int64_t)y * pitch_g + hx] : (%s)0;' % (width, ct_in))
    w('    }')
  w('  }')
  w('  __syncthreads();')
  w('  for (int r = wave; r < %d; r += 4) {' % tile_rows)
  w('    const int y = y0 + r;')
  w('    if (y >= n1 || !lane_ok) continue;')
  w('    %s res[%d];' % (ct_out, V))
  for e in range(V):

    def load(ref: ir.Ref, _e=e) -> str:
      dx = ref.idx[0] - stage.st_idx[0]
      dy = ref.idx[1] - stage.st_idx[1]
      return 'tile[r + %d][4 + lane * %d + %d]' % (ryl + dy, V, _e + dx)

    w('    res[%d] = (%s)(%s);' % (e, ct_out, ir.c_expr(stage.stmt.expr, load)))
  w('    soda_store_frag<%s, %d>(out + (int64_t)y * pitch_g + x, res);' %
    (ct_out, V))
  w('  }')
  w('}')
  idx = mod.add_kernel(
      KernelDesc(name, (256, 1, 1), (width, tile_rows),
                 lds_bytes=0, note='lds2d', tune=dict(vec=4)),
      '\n'.join(L) + '\n')
  p = PassDesc(1, [idx], 'lds2d',
               dict(bytes_per_cell_min=table[iname].size_in_bytes +
                    stage.haoda_type.size_in_bytes,
                    lds_bytes=rows * pitch * table[iname].size_in_bytes))
  mod.passes.append(p)
  return p


