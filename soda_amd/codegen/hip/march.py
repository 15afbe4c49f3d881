"""`march2d` / `march3d`: register-window marching kernels (see lower.py for
the design; DESIGN.md section 4.1 for the measurements behind the choices)."""
from typing import Dict, List, Optional, Tuple

from soda_amd import core, ir, util

from soda_amd.codegen.hip.module import KernelDesc, Module, PassDesc

MAX_UNROLL = 24        # tallest register window (planes) a marching wave holds
MAX_FUSE_PRESERVE = 8      # deepest temporal blocking under `border: preserve`
MAX_FUSE_3D = 2            # deepest temporal blocking of the 3-D kernels
MAX_SHIFT_TEMPS = 64       # lane-shifted operand copies per row step
REG_BUDGET = 160           # estimated VGPRs (windows + shifted copies) a shape may need

# ---------------------------------------------------------------------------
# march: register-window marching along the last (streamed) dimension
# ---------------------------------------------------------------------------

class _Node:
  """A tensor of the T-times unrolled chain."""

  def __init__(self, key, ctype: str, stage: Optional[core.Stage], it: int):
    self.key = key              # ('in', name) or (stage name, iteration)
    self.ctype = ctype
    self.stage = stage          # None for a global input
    self.it = it
    self.parents: Dict[str, '_Node'] = {}   # DSL name -> node
    self.delay = 0              # ticks between issue of input plane t and plane t here
    self.first_use = None       # age of the youngest plane any child reads
    self.window = 1             # planes (rows in 2-D) kept
    self.slots = 1              # window padded to a divisor of the unroll
    self.margin = [0, 0]        # invalid cells at the low/high end of a strip
    self.rmargin = [0, 0]       # 3-D: invalid rows at the low/high end of a tile
    self.store_slot: Optional[int] = None   # plan slot if stored to memory
    # stage-pipelined blocks: the wave that owns (computes / loads) the tensor;
    # a `mirror` is the copy a wave keeps of a tensor the previous wave of the
    # block produces, filled from the LDS ring one tick after its production
    self.owner = 0
    self.mirror_of: Optional['_Node'] = None
    self.to_lds = False         # some wave mirrors this tensor
    # border: preserve -- DSL name of the input whose value this output keeps
    # on cells one iteration cannot compute (it is then also a parent, tapped
    # at offset 0)
    self.keep: Optional[str] = None
    # x-halo sharing: the strip's neighbours hand this tensor's end cells over
    # through LDS (one register per row slot holds them, like edge loads)
    self.xs = False

  def tap_bounds(self, pname: str):
    """Bounds of the taps on parent `pname`, the offset-0 tap of a preserved
    border included."""
    dim = len(self.stage.st_idx)
    if pname in self.stage.taps:
      lo, hi = self.stage.tap_bounds(pname)
    else:
      lo, hi = (0,) * dim, (0,) * dim
    if pname == self.keep:
      lo = tuple(min(0, v) for v in lo)
      hi = tuple(max(0, v) for v in hi)
    return lo, hi

  @property
  def is_input(self) -> bool:
    return self.stage is None and self.mirror_of is None

  @property
  def fill_delay(self) -> int:
    """Ticks between the issue of input plane t and the tick at which plane t
    of THIS tensor is put into its register slot (loaded, read from LDS or
    computed).  Loads are put in flight `delay` ticks before they are used."""
    if self.mirror_of is not None:
      return self.delay - MIRROR_PREFETCH
    return 0 if self.stage is None else self.delay

  @property
  def var(self) -> str:
    if self.mirror_of is not None:
      return 'm%d_%s' % (self.owner, self.mirror_of.var)
    if self.stage is None:
      return 'g_%s' % self.key[1]
    return 't%d_%s' % (self.it, self.stage.name)


LDS_PER_CU = 160 * 1024     # MI355X_MICROARCH.md
MIRROR_PREFETCH = 1         # ticks between a mirror's LDS read and its first use


class MarchConfig:

  def __init__(self, fused_iters: int = 1, vec: int = 4, chunk_rows: int = 64,
               prefetch: int = 2, waves_x: int = 1, waves_y: int = 1,
               nt_store: bool = False, nt_load: bool = True,
               xcd_swizzle: bool = True, edge_loads: bool = True,
               tile_rows: int = 6, warm_guards: bool = False,
               interleave: bool = False, lane_shift: str = 'dpp',
               min_waves: int = 0, occupancy: int = 0,
               buffer_ops: bool = True, pipe: int = 1, pipe_rows: int = 4,
               stamps: bool = False,
               peel: int = -1, align_lanes: int = 1, xshare: int = 0,
               xwindow: bool = True, slide: bool = True):
    # integer window reductions along dimension 0 evaluated for all cells of
    # a lane jointly (_emit_xwindow); integer sums along the streamed
    # dimension as sliding sums
    self.xwindow = xwindow
    self.slide = slide
    # x-halos shared through LDS: a block of `xshare` waves covers the WHOLE
    # row (extent[0] <= xshare * 64 * vec, checked at launch), every wave a
    # strip of 64 fully valid lanes; the one cell a fused iteration needs from
    # the neighbouring strip's intermediate results is handed over through LDS
    # (lanes 0 and 63 write their end cells, one barrier per row step, the
    # neighbour reads them into the register DPP shifts take their `old`
    # operand from -- the mechanism edge loads use for the program inputs).
    # Without it a T = 2 strip of heat3d has 62 valid lanes = 248 cells and a
    # 512-cell row needs THREE waves, the third nearly idle.
    self.xshare = int(xshare)
    # valid lanes of a strip rounded down to a multiple of this (extra halo
    # lanes on the high side): 4 makes a strip's rows start and end on 64-byte
    # boundaries.  Pays where rows are written with non-temporal stores -- a
    # 63-lane strip's 1008-byte rows end in partial lines that go to memory
    # twice (blur 16384^2: 225 -> 202 us at 60 lanes) -- and not with plain
    # stores, which L2 merges (jacobi2d T = 4 / 8 / 12: no gain, slower)
    self.align_lanes = max(1, int(align_lanes))
    # the pipeline warm-up of a chunk as straight-line code in front of the
    # loop, WITHOUT the stages whose rows cannot reach an output row of the
    # chunk yet (fused iteration l first matters 2l row steps into the chunk:
    # T = 12 computes 156 of its first 288 stage-rows for nothing otherwise)
    # (in whole trips of the unrolled loop; -1: all the warm-up, 0: none.  The
    # last trips are nearly full row steps, save little and can cost registers
    # -- T = 12: 159 VGPRs with 2 of its 4 trips peeled, 231 with all four --
    # so runtime.select_peel picks the count per kernel from compiled code)
    self.peel = int(peel)
    # diagnostics: every wave records s_memtime at entry and exit and where it
    # ran (tools/timeline.py)
    self.stamps = stamps
    # row steps per barrier of a stage-pipelined block (power of two): the
    # waves synchronise once per `pipe_rows` rows through a ring of twice
    # that many slots
    self.pipe_rows = pipe_rows if pipe > 1 else 1
    # stage-pipelined blocks: the fused iterations are split over `pipe` waves
    # of a block; wave w runs iterations [w*T/pipe, (w+1)*T/pipe) of the SAME
    # strip and chunk one tick behind wave w-1, rows travel through a 2-slot
    # LDS ring, one barrier per tick.  Fewer registers per wave (more waves per
    # SIMD) and `pipe` times longer chunks for the same number of waves (the
    # 2T-row warm-up is paid per block, not per wave).
    self.pipe = pipe
    # vector memory through buffer resources with out-of-range offsets instead
    # of `if (row_ok)` branches (soda_rt.h): straight-line loop body, exact
    # s_waitcnt counts
    self.buffer_ops = buffer_ops
    # cap the waves resident per SIMD (0 = whatever the registers allow) by
    # giving every block an LDS allocation it never touches: 3 waves per SIMD
    # issue VALU work slower than 2 or 4 (tools/valubench.py)
    self.occupancy = occupancy
    # ask the register allocator for at least this many waves per SIMD
    # (amdgpu_waves_per_eu); 0 = let it use what it wants
    self.min_waves = min_waves
    # how a row is shifted by one lane: 'dpp' (wave_shr/shl fused into the
    # consuming add), 'bperm' (ds_bpermute_b32, issued one stage early), 'swz'
    # (ds_swizzle rotate + a readlane/writelane patch for the lane that crosses
    # the 32-lane halves) or 'swzh' (ds_swizzle rotate alone: the wave holds
    # two independent 32-lane half strips, each with its own halo lanes) or
    # 'lds': every lane files the end cells of a row it has computed in a
    # wave-private LDS line and takes its neighbours' from there one row step
    # later (plain ds_write / ds_read, no barrier: one wave, in-order LDS), so
    # the vector ALU sees no cross-lane operation for computed rows at all
    self.lane_shift = lane_shift
    # emit a stage's cells operation-major (independent statements back to
    # back).  Measured SLOWER on gfx950 (T=12: 186 vs 151 us): a wave64 VALU op
    # runs as two 32-lane passes, so a dependent op already issues without a
    # bubble, and the interleaved order only adds register-bank pressure.
    self.interleave = interleave
    self.warm_guards = warm_guards  # skip a stage while its rows cannot
    #                                 reach any output row of this chunk yet
    self.fused_iters = fused_iters
    self.vec = vec
    self.chunk_rows = chunk_rows  # cells one wave marches over (last dim)
    self.prefetch = prefetch
    self.waves_x = waves_x
    self.waves_y = waves_y
    self.nt_store = nt_store      # non-temporal stores of the output rows
    self.nt_load = nt_load        # non-temporal loads of the input rows
    self.xcd_swizzle = xcd_swizzle  # neighbouring tiles on one XCD (one L2)
    self.edge_loads = edge_loads  # halo cells of the inputs fetched by the
    #                               strip's edge lanes instead of overlap
    self.tile_rows = tile_rows    # 3-D: output rows (dim 1) per wave
    self.chunk_fixed = False      # True: the host must not re-size the chunk

  def key(self) -> str:
    # chunk_rows is a launch-time value, not part of the code
    return 'T%d_V%d_P%d_W%dx%d_R%d%s%s%s%s' % (
        self.fused_iters, self.vec, self.prefetch,
        self.waves_x, self.waves_y, self.tile_rows,
        '_nts' if self.nt_store else '', '_ntl' if self.nt_load else '',
        '_xcd' if self.xcd_swizzle else '',
        '_edge' if self.edge_loads else '') + (
            '_wg' if self.warm_guards else '') + (
                '_il' if self.interleave else '') + (
                    '_bp' if self.lane_shift == 'bperm' else
                    '_swz' if self.lane_shift == 'swz' else
                    '_swzh' if self.lane_shift == 'swzh' else
                    '_mixh' if self.lane_shift == 'mixh' else
                    '_mix64' if self.lane_shift == 'mix64' else
                    '_mix64d' if self.lane_shift == 'mix64d' else
                    '_ldsx' if self.lane_shift == 'lds' else
                    '_noshift' if self.lane_shift == 'none' else '') + (
                        '_mw%d' % self.min_waves if self.min_waves else '') + (
                            '_occ%d' % self.occupancy if self.occupancy else '') + (
                                '_buf' if self.buffer_ops else '') + (
                                    '_pipe%dx%d' % (self.pipe, self.pipe_rows)
                                    if self.pipe > 1 else '') + (
                                        '_st' if self.stamps else '') + (
                                            '' if self.peel < 0 else
                                            '_k%d' % self.peel) + (
                                                '_al%d' % self.align_lanes
                                                if self.align_lanes > 1 else '') + (
                                                    '_xs%d' % self.xshare
                                                    if self.xshare else '')


March2DConfig = MarchConfig   # older name


def default_vec(stencil: core.Stencil) -> int:
  """16 bytes per lane per row for the widest tensor."""
  widest = max(t.size_in_bytes for t in stencil.symbol_table.values())
  return max(1, 16 // widest)


def march_supported(stencil: core.Stencil) -> Optional[str]:
  """None if the marching kernels can run the program, else why not."""
  if stencil.dim not in (2, 3):
    return 'the marching kernels need a 2- or 3-dimensional program'
  return None


march2d_supported = march_supported


def _build_chain(st: core.Stencil, T: int, pf: int, edge: Tuple[int, int],
                 pipe: int = 1, pipe_rows: int = 1, xshare: bool = False):
  """The tensors of T chained iterations with their schedule (delay in ticks
  behind the load of the input plane, window of planes kept, invalid margins).
  `pipe` > 1 splits the iterations evenly over that many waves of a block:
  a tensor consumed by the next wave gets a mirror there (see _Node)."""
  dim = st.dim
  ax = dim - 1                         # march axis
  table = st.symbol_table
  per_wave = T // pipe
  nodes: List[_Node] = []
  inputs = {}
  for name in st.input_names:
    n = _Node(('in', name), table[name].c_type, None, -1)
    n.delay = pf
    n.margin = [-edge[0], -edge[1]]    # edge lanes fetch that many halo cells
    inputs[name] = n
    nodes.append(n)
  cur_inputs = dict(inputs)
  last_outputs = {}
  for it in range(T):
    owner = it // per_wave if pipe > 1 else 0
    env = {}
    for name, src in cur_inputs.items():
      if src.owner != owner:
        m = _Node(('mirror', src.key, owner), src.ctype, None, it)
        m.mirror_of = src
        m.owner = owner
        src.to_lds = True
        nodes.append(m)
        src = m
      env[name] = src
    for stage in st.ordered_stages:
      n = _Node((stage.name, it), stage.haoda_type.c_type, stage, it)
      n.owner = owner
      for parent in stage.taps:
        n.parents[parent] = env[parent]
      if st.preserve_border and stage.is_output:
        n.keep = st.preserved_from(stage.name)
        n.parents.setdefault(n.keep, env[n.keep])
      env[stage.name] = n
      nodes.append(n)
    last_outputs = {o: env[o] for o in st.output_names}
    if it < T - 1:
      cur_inputs = {i: env[o] for i, o in zip(st.input_names, st.output_names)}
  def share(n: _Node) -> None:
    """x-halo sharing: if some consumer taps `n` off-centre along dim 0, the
    neighbouring strips hand its end cells over (one valid cell per side)."""
    reach = 0
    for c in nodes:
      if c.stage is None or c.mirror_of is not None:
        continue
      for pname, p in c.parents.items():
        if p is not n:
          continue
        tlo, thi = c.tap_bounds(pname)
        reach = max(reach, -tlo[0], thi[0])
        if tlo[0] < 0 or thi[0] > 0:
          # the halo cells of a plane arrive at the END of the row step that
          # computes it (or first reads it, for an input): taps off-centre in
          # x must not touch the newest plane the consumer reads
          newest = max(off[ax] for off in c.stage.taps.get(pname, ())
                       if off[0] != 0)
          if newest >= thi[ax]:
            raise util.SemanticError(
                'march: x-halo sharing needs a row step between a plane and '
                'its off-centre taps (%s reads %s)' % (c.var, n.var))
    if reach:
      if reach > 1 or n.margin[0] > 0 or n.margin[1] > 0:
        raise util.SemanticError(
            'march: x-halo sharing hands over one valid cell per side')
      n.xs = True
      n.margin = [n.margin[0] - 1, n.margin[1] - 1]

  if xshare:
    for n in inputs.values():
      share(n)
  # delays and margins, in chain order
  for n in nodes:
    if n.mirror_of is not None:
      src = n.mirror_of
      # read `pipe_rows` ticks (one barrier period) after it was written, used
      # MIRROR_PREFETCH ticks after that
      n.delay = src.delay + pipe_rows + MIRROR_PREFETCH
      n.margin = list(src.margin)
      n.rmargin = list(src.rmargin)
      continue
    if n.stage is None:
      continue
    delay = None
    margin = [0, 0]
    rmargin = [0, 0]
    for pname, p in n.parents.items():
      tlo, thi = n.tap_bounds(pname)
      d = p.delay + thi[ax]
      delay = d if delay is None else max(delay, d)
      margin[0] = max(margin[0], p.margin[0] + max(0, -tlo[0]))
      margin[1] = max(margin[1], p.margin[1] + max(0, thi[0]))
      if dim == 3:
        rmargin[0] = max(rmargin[0], p.rmargin[0] + max(0, -tlo[1]))
        rmargin[1] = max(rmargin[1], p.rmargin[1] + max(0, thi[1]))
    n.delay = delay if delay is not None else 0
    n.margin = margin
    n.rmargin = rmargin
    if xshare:
      share(n)
  # windows: a parent keeps planes from its newest (age 0) to the oldest any
  # child still reads
  for n in nodes:
    if n.stage is None:
      continue
    for pname, p in n.parents.items():
      tlo, thi = n.tap_bounds(pname)
      p.window = max(p.window, n.delay - tlo[ax] - p.fill_delay + 1)
      # the youngest plane of the parent any child reads (an input plane is
      # first touched that many ticks after its load was issued)
      first = n.delay - thi[ax] - p.fill_delay
      p.first_use = first if p.first_use is None else min(p.first_use, first)
  return nodes, inputs, last_outputs


def _choose_unroll(nodes: List[_Node], regs_per_plane: int,
                   multiple_of: int = 1) -> Optional[int]:
  max_w = max(n.window for n in nodes)
  best = None
  for u in range(max_w, MAX_UNROLL + 1):
    if u % multiple_of:
      continue
    divisors = [d for d in range(1, u + 1) if u % d == 0]
    pad = 0
    for n in nodes:
      slots = min(d for d in divisors if d >= n.window)
      pad += slots - n.window
    cost = pad * regs_per_plane + 2 * u
    if best is None or cost < best[0]:
      best = (cost, u)
  return None if best is None else best[1]


class _MarchKernel:
  """One marching kernel: the schedule of the T-times unrolled stage chain
  (constructor) and its HIP text (`emit`)."""

  def __init__(self, mod: Module, cfg: MarchConfig):
    self.mod, self.cfg = mod, cfg
    self.st = self.mod.stencil
    why = march_supported(self.st)
    if why:
      raise util.SemanticError('march: %s' % why)
    self.dim = self.st.dim
    self.ax = self.dim - 1
    self.T, self.V, self.PF = self.cfg.fused_iters, self.cfg.vec, self.cfg.prefetch
    if self.T > 1 and len(self.st.input_names) != len(self.st.output_names):
      raise util.SemanticError('march: cannot fuse iterations of a program '
                               'whose inputs and outputs differ in number')
    # halo cells of the program inputs that the strip's edge lanes fetch with
    # separate narrow loads (so a 1-iteration strip keeps all 64 lanes valid and
    # its rows start on a 64*V-cell boundary)
    self.edge = (0, 0)
    # lanes that form one strip: the whole wave, or each 32-lane half
    self.group = 32 if self.cfg.lane_shift in ('swzh', 'mixh') else 64
    if self.cfg.edge_loads and self.cfg.lane_shift in ('dpp', 'none', 'lds'):
      # (the edge cells enter through DPP's `old` operand)
      lo = hi = 0
      for stage in self.st.ordered_stages:
        for pname in stage.taps:
          if pname in self.st.input_names:
            tlo, thi = stage.tap_bounds(pname)
            lo, hi = max(lo, -tlo[0]), max(hi, thi[0])
      if max(lo, hi) <= min(self.V, 2):
        self.edge = (lo, hi)
    self.W = self.cfg.pipe
    if self.W > 1:
      if self.T % self.W or not self.cfg.buffer_ops or \
          self.cfg.waves_x * self.cfg.waves_y != 1:
        raise util.SemanticError(
            'march: %d pipelined waves need a fusion depth that is a multiple '
            'of it, buffer addressing and one strip per block' % self.W)
      self.edge = (0, 0)
    self.R = self.cfg.pipe_rows if self.W > 1 else 1
    if self.R & (self.R - 1):
      raise util.SemanticError('march: rows per barrier must be a power of two')
    self.xs = self.cfg.xshare
    if self.xs:
      self.edge = (0, 0)    # the inputs' halo cells travel through LDS as well
      if self.W > 1 or self.cfg.lane_shift != 'dpp' or not self.cfg.buffer_ops \
          or self.cfg.waves_x * self.cfg.waves_y != 1:
        raise util.SemanticError(
            'march: x-halo sharing needs DPP shifts, buffer addressing and '
            'one wave per strip')
      if not 1 <= self.xs <= 16:
        raise util.SemanticError('march: 1 to 16 waves can share a row')
      if self.dim == 3 and self.cfg.tile_rows > 24:
        raise util.SemanticError('march: x-halo sharing holds a row per lane')
    self.nodes, self.inputs, self.outputs = _build_chain(
        self.st, self.T, self.PF, self.edge, self.W, self.R, bool(self.xs))
    self.out_nodes = list(self.outputs.values())
    self.margin_lo = max(0, max(n.margin[0] for n in self.out_nodes))
    self.margin_hi = max(0, max(n.margin[1] for n in self.out_nodes))
    self.lanes_lo = -(-self.margin_lo // self.V)
    self.lanes_hi = -(-self.margin_hi // self.V)
    if self.edge != (0, 0) and not self.xs:
      # edge loads only pay when they save a halo lane
      plain = _build_chain(self.st, self.T, self.PF, (0, 0), self.W, self.R)[2]
      p_lo = -(-max(n.margin[0] for n in plain.values()) // self.V)
      p_hi = -(-max(n.margin[1] for n in plain.values()) // self.V)
      if (p_lo, p_hi) == (self.lanes_lo, self.lanes_hi):
        self.edge = (0, 0)
        self.nodes, self.inputs, self.outputs = _build_chain(
            self.st, self.T, self.PF, self.edge, self.W, self.R, bool(self.xs))
        self.out_nodes = list(self.outputs.values())
    if self.cfg.align_lanes > 1 and self.group == 64:
      spare = (64 - self.lanes_lo - self.lanes_hi) % self.cfg.align_lanes
      if self.lanes_lo + self.lanes_hi + spare < 16:
        self.lanes_hi += spare
    if self.lanes_lo + self.lanes_hi >= self.group // 2:
      raise util.SemanticError('march: halo of %d+%d cells is too wide for a '
                               '%d-lane strip at %d cells per lane' %
                               (self.margin_lo, self.margin_hi, self.group,
                                self.V))
    # 3-D: rows (dim 1) of a tile held in registers
    if self.dim == 3:
      self.rhalo_lo = max(n.rmargin[0] for n in self.out_nodes)
      self.rhalo_hi = max(n.rmargin[1] for n in self.out_nodes)
      self.rows_in = self.cfg.tile_rows + self.rhalo_lo + self.rhalo_hi
    else:
      self.rhalo_lo = self.rhalo_hi = 0
      self.rows_in = 1
    self.tile_rows = self.rows_in - self.rhalo_lo - self.rhalo_hi
    self.U = _choose_unroll(self.nodes, self.V * self.rows_in, self.R)
    if self.U is None:
      raise util.SemanticError(
          'march: a tensor needs a window of %d planes (> %d)' %
          (max(n.window for n in self.nodes), MAX_UNROLL))
    for n in self.nodes:
      n.slots = min(d for d in range(1, self.U + 1) if self.U % d == 0 and d >= n.window)
    # rows (3-D) of every tensor that some output row of the tile depends on:
    # the generator emits all rows a tensor can hold, the compiler drops the
    # statements nobody reads -- the estimate must not count them (denoise3d:
    # 10 stages, most of them needed on the 2-4 output rows only)
    need = {id(n): None for n in self.nodes}
    for n in self.out_nodes:
      need[id(n)] = (self.rhalo_lo, self.rows_in - self.rhalo_hi)
    for n in reversed(self.nodes):
      if need[id(n)] is None:
        continue
      src = n.mirror_of
      if src is not None:
        need[id(src)] = need[id(n)]
        continue
      if n.stage is None:
        continue
      a, b = need[id(n)]
      for pname, pnode in n.parents.items():
        tlo, thi = n.tap_bounds(pname)
        lo = a + (tlo[1] if self.dim == 3 else 0)
        hi = b + (thi[1] if self.dim == 3 else 0)
        cur = need[id(pnode)]
        need[id(pnode)] = (lo, hi) if cur is None else (min(cur[0], lo),
                                                         max(cur[1], hi))

    def rows_needed(n: _Node) -> int:
      if need[id(n)] is None:
        return 0
      lo = max(need[id(n)][0], n.rmargin[0])
      hi = min(need[id(n)][1], self.rows_in - n.rmargin[1])
      return max(0, hi - lo)

    self.est_regs = max(
        sum(n.slots * self.V * rows_needed(n)
            for n in self.nodes if n.owner == wv) for wv in range(self.W))
    if self.est_regs > 400:
      raise util.SemanticError(
          'march: the register windows need about %d VGPRs per lane' % self.est_regs)
    for o, n in self.outputs.items():
      n.store_slot = self.mod.slot[o]

    # lowest plane of every tensor (relative to the chunk's first output plane)
    # that some output plane of the chunk depends on: before that, computing the
    # tensor is wasted pipeline warm-up
    self.back_lo = {id(n): None for n in self.nodes}
    for n in self.out_nodes:
      self.back_lo[id(n)] = 0
    for n in reversed(self.nodes):
      if self.back_lo[id(n)] is None:
        continue
      if n.mirror_of is not None:      # the same planes, held by another wave
        src = n.mirror_of
        if self.back_lo[id(src)] is None or \
            self.back_lo[id(n)] < self.back_lo[id(src)]:
          self.back_lo[id(src)] = self.back_lo[id(n)]
        continue
      if n.stage is None:
        continue
      for pname, pnode in n.parents.items():
        tlo, _ = n.tap_bounds(pname)
        cand = self.back_lo[id(n)] + tlo[self.ax]
        if self.back_lo[id(pnode)] is None or cand < self.back_lo[id(pnode)]:
          self.back_lo[id(pnode)] = cand

    self.group_lanes = self.group - self.lanes_lo - self.lanes_hi
    self.strip_lanes = self.group_lanes * (64 // self.group)
    self.strip_cells = self.strip_lanes * self.V
    self.max_delay = max(n.delay for n in self.out_nodes)
    bounds = self.st.window_bounds(self.T)
    self.m_lo = min(0, min(bounds[o][0][self.ax] for o in self.st.output_names))
    self.m_hi = max(0, max(bounds[o][1][self.ax] for o in self.st.output_names))
    # load-only ticks at the head of a chunk (buffer addressing only: the peeled
    # loads must not need branches)
    # A stage that lags the loads by d computes, in the first `lead` <= d ticks,
    # only planes below every plane the chunk needs; a stage without inputs
    # (a constant) has d = 0 and forbids the peeling.
    self.lead = 0
    if self.cfg.buffer_ops:
      self.lead = min([self.PF] + [n.delay for n in self.nodes if n.stage is not None])
    # compute ticks before a chunk's first output
    self.warm = self.max_delay - self.m_lo - self.lead

    # peeled warm-up: row steps in front of the loop (a whole number of loop
    # trips, so the register slots line up) in which a stage is emitted only
    # from the step on at which its rows start to matter
    self.peeled = 0
    self.peel_trips_max = 0
    if self.W == 1:
      steps = -(-self.warm // self.U) * self.U
      stages = [n for n in self.nodes if n.stage is not None]
      skipped = sum(1 for i in range(steps) for n in stages
                    if not self.stage_needed(n, i))
      if skipped * 4 >= steps * len(stages):
        self.peel_trips_max = steps // self.U
        trips = self.peel_trips_max if self.cfg.peel < 0 else min(
            self.cfg.peel, self.peel_trips_max)
        self.peeled = trips * self.U

    self.kind = 'march%dd' % self.dim
    self.name = '%s_%s_%s' % (self.st.app_name, self.kind, self.cfg.key())
    self.waves = self.cfg.waves_x * self.cfg.waves_y if self.dim == 2 else 1
    self.wx = self.cfg.waves_x if self.dim == 2 else 1
    if self.W > 1:
      self.waves = self.W
    if self.xs:
      self.waves = self.xs
    self.block = 64 * self.waves
    self.L: List[str] = []
    self.w = self.L.append
    self.shift_temps = 0    # lane-shifted copies alive within one tick

  def rows_of(self, n: _Node):
    return range(n.rmargin[0], self.rows_in - n.rmargin[1])

  def stage_needed(self, n: _Node, step: int) -> bool:
    """Does the plane `n` computes `step` row steps into a chunk (counted from
    the first step of the loop) reach any output plane of the chunk?"""
    lo = self.back_lo[id(n)]
    if lo is None:
      return False
    return self.m_lo + self.lead + step >= lo + n.delay

  def store_rows(self, n: _Node):
    return range(max(n.rmargin[0], self.rhalo_lo),
                 self.rows_in - max(n.rmargin[1], self.rhalo_hi))

  @staticmethod
  def slot_of(n: _Node, k: int, age: int) -> int:
    return (k - age) % n.slots

  # -- emission ---------------------------------------------------------------
  def emit(self) -> PassDesc:
    self._emit_header()
    self._emit_addressing()
    for wv in range(self.W):
      self._emit_wave(wv)
    if self.cfg.stamps:
      self.w('  if (lane == 0) {')
      self.w('    unsigned long long* d = (unsigned long long*)a.buf[15] + '
             '4ull * ((unsigned long long)blockIdx.x * %d + wave);' % self.waves)
      self.w('    d[0] = soda_t0;')
      self.w('    d[1] = __builtin_amdgcn_s_memtime();')
      self.w('    d[2] = __builtin_amdgcn_s_getreg(63492);   // HW_REG_HW_ID')
      self.w('    d[3] = __builtin_amdgcn_s_getreg(63508);   // HW_REG_XCC_ID')
      self.w('  }')
    self.w('}')
    return self._finish()

  def _emit_header(self) -> None:
    self.w('// %s: T=%d fused iteration(s), %d cells/lane, chunks of %d along dim %d,'
      ' prefetch %d, unroll %d' % (self.kind, self.T, self.V,
                                   self.cfg.chunk_rows, self.ax, self.PF,
                                   self.U))
    self.w('// strip: %d lanes valid of 64 (halo %d+%d cells, edge loads %d+%d); '
      'pipeline depth %d' % (self.strip_lanes, self.margin_lo, self.margin_hi,
                             self.edge[0], self.edge[1],
                             self.warm))
    if self.dim == 3:
      self.w('// tile: %d rows held in registers for %d output rows (halo %d+%d)' %
        (self.rows_in, self.tile_rows, self.rhalo_lo, self.rhalo_hi))
    for n in self.nodes:
      self.w('//   %-24s delay %2d  window %2d (slots %2d)  margin %d/%d  rows %d/%d' %
        (n.var, n.delay, n.window, n.slots, n.margin[0], n.margin[1],
         n.rmargin[0], n.rmargin[1]))
    self.w('extern "C" __global__ void __launch_bounds__(%d) %s%s(soda_hip_kargs_t a) {'
      % (self.block, '__attribute__((amdgpu_waves_per_eu(%d, %d))) ' %
         (self.cfg.min_waves, max(self.cfg.min_waves, 8))
         if self.cfg.min_waves else '', self.name))
    if self.cfg.stamps:
      self.w('  const unsigned long long soda_t0 = __builtin_amdgcn_s_memtime();')
    self.w('  const int lane = (int)(threadIdx.x & 63u);')
    self.w('  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));')
    if self.cfg.xcd_swizzle:
      # blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous
      # run of tiles so halo rows/columns shared by neighbours hit one L2
      # (speed only; any placement is correct)
      self.w('  const unsigned nblk = gridDim.x;')
      self.w('  const unsigned bid = (nblk % 8u == 0u) ? (blockIdx.x % 8u) * (nblk / 8u)'
        ' + blockIdx.x / 8u : blockIdx.x;')
    else:
      self.w('  const unsigned bid = blockIdx.x;')
    self.w('  const int tile_x = (int)(bid % (unsigned)a.ntile[0]);')
    if self.dim == 2:
      self.w('  const int tile_l = (int)(bid / (unsigned)a.ntile[0]);')
      self.w('  const int tile_m = tile_l + (tile_l >= a.skip_from ? a.skip_count : 0);')
      if self.xs:           # the block's waves cover the row side by side
        self.w('  const int strip = wave;')
        self.w('  const int chunk = tile_m;')
      elif self.W > 1:      # all waves of the block share the strip and the chunk
        self.w('  const int strip = tile_x;')
        self.w('  const int chunk = tile_m;')
      else:
        self.w('  const int strip = tile_x * %d + wave %% %d;' % (self.wx, self.wx))
        self.w('  const int chunk = tile_m * %d + wave / %d;' % (self.cfg.waves_y, self.wx))
      self.w('  const int n0 = a.extent[0], nm = a.extent[1];')
    else:
      self.w('  const int tile_y = (int)((bid / (unsigned)a.ntile[0]) % '
        '(unsigned)a.ntile[1]);')
      self.w('  const int chunk_l = (int)(bid / ((unsigned)a.ntile[0] * '
        '(unsigned)a.ntile[1]));')
      self.w('  const int chunk = chunk_l + (chunk_l >= a.skip_from ? a.skip_count : 0);')
      self.w('  const int strip = %s;' % ('wave' if self.xs else 'tile_x'))
      self.w('  const int n0 = a.extent[0], n1 = a.extent[1], nm = a.extent[2];')
      self.w('  const int y0 = tile_y * %d - %d;  // first row held' %
        (self.tile_rows, self.rhalo_lo))
      self.w('  const int64_t pitch_y = a.stride[1];')
    if self.group == 64:
      self.w('  const int sub = lane;')
      self.w('  const int x0 = strip * %d + (lane - %d) * %d;' %
        (self.strip_cells, self.lanes_lo, self.V))
    else:     # two half strips side by side, each with its own halo lanes
      self.w('  const int sub = lane & %d;' % (self.group - 1))
      self.w('  const int x0 = strip * %d + (lane >> 5) * %d + (sub - %d) * %d;'
        % (self.strip_cells, self.group_lanes * self.V, self.lanes_lo, self.V))
    # chunk length is a launch-time value (kernel descriptor tile / waves): the
    # host sizes it so the grid fills the GPU in whole rounds of waves
    self.w('  const int chunk_len = a.tile[%d] / %d;' %
      (self.ax, self.cfg.waves_y if self.dim == 2 else 1))
    self.w('  const int m_begin = chunk * chunk_len;')
    self.w('  const int m_end = min(m_begin + chunk_len, nm);')
    if self.xs:
      # every wave of the block takes part in the barriers, whatever its strip
      self.w('  if (m_begin >= nm) return;  // block-uniform')
    else:
      self.w('  if (m_begin >= nm || strip * %d >= n0) return;  // wave-uniform' %
        self.strip_cells)
    self.w('  const bool lane_ok = x0 >= 0 && x0 + %d <= n0;' % self.V)
    self.w('  const bool store_ok = lane_ok && sub >= %d && sub < %d;' %
      (self.lanes_lo, self.group - self.lanes_hi))
    self.w('  const int64_t pitch = a.stride[%d];' % self.ax)
    self.w('  const int64_t x0c = lane_ok ? (int64_t)x0 : 0;')

  def _emit_addressing(self) -> None:
    self.buf = self.cfg.buffer_ops
    table0 = self.st.symbol_table
    self.esz = {nme: table0[nme].size_in_bytes for nme in list(self.inputs) + list(self.outputs)}
    self.n_edge = max(self.edge)
    if self.buf:
      # every tensor is addressed through a window that starts at the first
      # plane this wave touches; see soda_rt.h for the offset encoding
      self.w('  const int in_end = min(nm, m_end + %d);  // last input plane needed + 1'
        % self.m_hi)
      self.w('  const int wlo = max(0, m_begin + (%d));' % self.m_lo)
      for es in sorted(set(self.esz.values())):
        self.w('  const unsigned xb%d = lane_ok ? (unsigned)x0 * %du : SODA_OOB_X;' %
          (es, es))
        self.w('  const unsigned sxb%d = store_ok ? (unsigned)x0 * %du : SODA_OOB_X;' %
          (es, es))
        self.w('  const unsigned pitch_b%d = (unsigned)pitch * %du;' % (es, es))
        if self.dim == 3:
          self.w('  const unsigned pitch_yb%d = (unsigned)pitch_y * %du;' % (es, es))
          for j in range(self.rows_in):
            self.w('  const unsigned yo%d_%d = (y0 + %d >= 0 && y0 + %d < n1) ? '
                   '(unsigned)(y0 + %d) * pitch_yb%d : SODA_OOB_X;' %
                   (es, j, j, j, j, es))
      for nme, n in self.inputs.items():
        self.w('  const soda_rsrc_t r_%s = soda_make_rsrc((const %s*)a.buf[%d] + '
          '(int64_t)wlo * pitch, (int64_t)(in_end - wlo) * pitch * %d);' %
          (nme, n.ctype, self.mod.slot[nme], self.esz[nme]))
      for o, n in self.outputs.items():
        self.w('  const soda_rsrc_t w_%s = soda_make_rsrc((%s*)a.buf[%d] + '
          '(int64_t)m_begin * pitch, (int64_t)(m_end - m_begin) * pitch * %d);' %
          (o, n.ctype, self.mod.slot[o], self.esz[o]))
    else:
      for nme, n in self.inputs.items():
        self.w('  const %s* __restrict__ p_%s = (const %s*)a.buf[%d] + x0c;' %
          (n.ctype, nme, n.ctype, self.mod.slot[nme]))
      for o, n in self.outputs.items():
        self.w('  %s* __restrict__ q_%s = (%s*)a.buf[%d] + x0c;' %
          (n.ctype, o, n.ctype, self.mod.slot[o]))
    if self.xs or self.n_edge:
      self.w('  const bool edge_lane = lane == 0 || lane == 63;')
    self.xs_nodes = [n for n in self.nodes if n.xs]
    if self.xs_nodes:    # (none: no tap leaves its column, nothing to share)
      self.xs_rows = max(len(self.rows_of(n)) for n in self.xs_nodes)
      if self.xs_rows > 32:
        raise util.SemanticError('march: x-halo sharing holds a row per lane')
      self.xs_stride = self.xs * 2 * self.xs_rows
      for n in self.xs_nodes:
        # [parity of the row step][wave][0: its first cell, 1: its last][row],
        # then one cell that stays zero (what lies beyond the row's ends)
        self.w('  __shared__ %s soda_xs_%s[%d];' %
               (n.ctype, n.var, 2 * self.xs_stride + 1))
      self.w('  if (threadIdx.x == 0) {')
      for n in self.xs_nodes:
        self.w('    soda_xs_%s[%d] = (%s)0;' % (n.var, 2 * self.xs_stride, n.ctype))
      self.w('  }')
      self.w('  soda_pipe_barrier();')
      # writing: lane 0 files its first cell, lane 63 its last
      self.w('  const int xs_wr = (wave * 2 + (lane == 0 ? 0 : 1)) * %d;' %
             self.xs_rows)
      # reading: lane r takes row r's cell from the strip on the left (its last
      # cell), lane 32 + r from the strip on the right (its first cell); lanes
      # without a neighbour or a row read the zero cell
      self.w('  const int xs_nb = lane < 32 ? wave - 1 : wave + 1;')
      self.w('  const int xs_rd = (xs_nb >= 0 && xs_nb < %d && (lane & 31) < %d) ? '
             '(xs_nb * 2 + (lane < 32 ? 1 : 0)) * %d + (lane & 31) : %d;' %
             (self.xs, self.xs_rows, self.xs_rows, 2 * self.xs_stride))
    if self.n_edge:
      # lane 0 fetches the cells left of the strip, lane 63 those right of it;
      # DPP hands them to the shifted reads through the `old` operand
      for i in range(self.n_edge):
        self.w('  const int ex%d = lane == 0 ? x0 - %d : x0 + %d;' %
          (i, 1 + i, self.V + i))
        self.w('  const bool edge_ok%d = edge_lane && ex%d >= 0 && ex%d < n0;' %
          (i, i, i))
        if self.buf:
          for es in sorted({self.esz[nme] for nme in self.inputs}):
            self.w('  const unsigned exb%d_%d = edge_ok%d ? (unsigned)ex%d * %du : '
              'SODA_OOB_X;' % (i, es, i, i, es))
        else:
          self.w('  const int64_t eoff%d = edge_ok%d ? (int64_t)(ex%d - x0c) : 0;' %
            (i, i, i))

    if self.st.preserve_border:
      # border: preserve -- which of a lane's cells lie outside the columns one
      # iteration can compute (the same for every fused iteration)
      for o in self.st.output_names:
        wlo, whi = self.st.interior_bounds(o)
        for e in range(self.V):
          self.w('  const bool keepx_%s_%d = !(a.origin[0] + x0 + %d >= %d && '
                 'a.origin[0] + x0 + %d < a.gextent[0] - %d);'
                 % (o, e, e, max(0, -wlo[0]), e, max(0, whi[0])))
    self.use_bperm = self.cfg.lane_shift == 'bperm'
    self.use_swz = self.cfg.lane_shift in ('swz', 'swzh')
    # 'mixh': the two shift directions on two different pipes -- down through
    # DPP (vector ALU), up through ds_swizzle (the LDS crossbar, shared by the
    # four SIMDs of a CU) -- on half strips, as 'swzh'
    # 'mix64' / 'mix64d' (round 5 experiment): the same split on WHOLE 64-lane
    # strips -- the DPP wave shift crosses the halves by itself, the swizzle
    # rotates within 32 lanes and the one lane that must cross is patched
    # (v_readlane + v_writelane / one DPP move restricted to lanes 28-31): 56
    # of 64 lanes valid at T = 13 instead of 48, for 2 / 1 more instructions
    # per fused iteration and row step
    self.use_mix = self.cfg.lane_shift in ('mixh', 'mix64', 'mix64d')
    # Integer sums over a run of taps along the STREAMED dimension as a sliding
    # sum: an int32 accumulator per cell that lives across row steps,
    #     acc += newest row;  result = cast(acc);  acc -= oldest row
    # -- 2 operations per cell whatever the window (xcorr: 19 rows), no
    # auxiliary tensors.  State that survives row steps: the stage must run in
    # EVERY step from its first on (it does: a stage is never dropped again
    # once the peeled warm-up has started it) and the accumulator is set up
    # right before that first step from the rows the window holds then.
    # Exact: integers narrower than 32 bits (optimization/windows.py).
    self.slide: Dict[int, Tuple[str, Tuple[int, ...], int]] = {}
    if self.cfg.xwindow and self.cfg.slide and self.W == 1 and \
        not self.cfg.warm_guards and not self.cfg.interleave:
      from soda_amd.optimization import windows
      table = dict(self.st.symbol_table)
      for n in self.nodes:
        if n.stage is None or n.mirror_of is not None or n.keep is not None:
          continue
        m = windows._match(n.stage.stmt, table)
        if m is None or m[0] != '+' or m[2] != self.ax:
          continue
        off = tuple(a - b for a, b in zip(m[5], n.stage.st_idx))
        self.slide[id(n)] = (m[1], off, m[4])
    if self.use_bperm:
      self.w('  const int lane_dn_addr = ((lane + 63) & 63) << 2;  // byte address of lane-1')
      self.w('  const int lane_up_addr = ((lane + 1) & 63) << 2;')
    # 'lds' lane shifts: which computed tensors hand their end cells to the
    # neighbouring lanes through LDS, and how many cells per side.  A tensor
    # qualifies when every off-centre tap on it reads the row computed ONE row
    # step ago (the line holds one row; jacobi-like stencils: taps off-centre
    # in x sit on the centre row, the newest row is one ahead) and reaches at
    # most one lane.  The others keep DPP.
    self.ldsx: Dict[int, Tuple[int, int]] = {}
    if self.cfg.lane_shift == 'lds' and self.W == 1 and not self.xs and \
        self.dim == 2:
      for n in self.nodes:
        if n.stage is None or n.mirror_of is not None:
          continue
        lo = hi = 0
        ok = True
        for c in self.nodes:
          if c.stage is None:
            continue
          for pname, pnode in c.parents.items():
            if pnode is not n:
              continue
            for off in c.stage.taps.get(pname, ()):
              if off[0] == 0:
                continue
              first, last = off[0], off[0] + self.V - 1   # cells 0 and V-1
              if first >= 0 and last < self.V:
                continue                    # stays inside the lane
              if c.delay - off[self.ax] - n.fill_delay != 1 or \
                  abs(off[0]) > self.V:
                ok = False
              lo, hi = max(lo, -off[0]), max(hi, off[0])
        if ok and (lo or hi):
          self.ldsx[id(n)] = (lo, hi)
      for n in self.nodes:
        if id(n) not in self.ldsx:
          continue
        lo, hi = self.ldsx[id(n)]
        # L<i>[j]: cell V-1-i of lane j-1;  R<i>[j]: cell i of lane j
        for i in range(lo):
          self.w('  __shared__ %s soda_lx_%s_L%d[65];' % (n.ctype, n.var, i))
        for i in range(hi):
          self.w('  __shared__ %s soda_lx_%s_R%d[65];' % (n.ctype, n.var, i))
      if self.ldsx:
        self.w('  if (lane == 0) {   // what lies beyond the wave\'s end lanes')
        for n in self.nodes:
          if id(n) in self.ldsx:
            lo, hi = self.ldsx[id(n)]
            for i in range(lo):
              self.w('    soda_lx_%s_L%d[0] = (%s)0;' % (n.var, i, n.ctype))
            for i in range(hi):
              self.w('    soda_lx_%s_R%d[64] = (%s)0;' % (n.var, i, n.ctype))
        self.w('  }')
    declared = set()
    for stage in self.st.ordered_stages:      # `param` arrays: plain pointers
      for line in self.mod.param_decls(stage):
        if line not in declared:
          declared.add(line)
          self.w(line)
    for n in self.nodes:
      if n.to_lds:     # written at tick t, read by the next wave at tick t + R
        self.w('  __shared__ %s soda_ring_%s[%d][%d][%d];' % (
            n.ctype, n.var, 2 * self.R, self.rows_in, 64 * self.V))
    if not self.buf:
      self.w('  const int in_end = min(nm, m_end + %d);  // last input plane needed + 1'
        % self.m_hi)
    self.nt_l = 'true' if self.cfg.nt_load else 'false'
    self.nt_s = 'true' if self.cfg.nt_store else 'false'

  def emit_decls(self, wv: int) -> None:
      for n in self.nodes:
        if n.owner != wv:
          continue
        if id(n) in self.slide:
          for j in self.rows_of(n):
            self.w('  int xa_%s_r%d[%d];' % (n.var, j, self.V))
            self.w('  soda_zero_frag<int, %d>(xa_%s_r%d);' % (self.V, n.var, j))
        for s in range(n.slots):
          for j in self.rows_of(n):
            self.w('  %s %s_s%d_r%d[%d];' % (n.ctype, n.var, s, j, self.V))
            self.w('  soda_zero_frag<%s, %d>(%s_s%d_r%d);' % (n.ctype, self.V, n.var, s, j))
            if n.is_input and self.n_edge:
              self.w('  %s %s_s%d_r%d_e[%d];' % (n.ctype, n.var, s, j, self.n_edge))
              self.w('  soda_zero_frag<%s, %d>(%s_s%d_r%d_e);' %
                (n.ctype, self.n_edge, n.var, s, j))
          if n.xs:
            # halo cells of the plane in slot s: lane r / 32 + r holds row r's
            # left / right neighbour cell
            self.w('  %s hv_%s_s%d = (%s)0;' % (n.ctype, n.var, s, n.ctype))

  def emit_loads(self, k: int, t_expr: str, pin: bool = False) -> None:
      """Issues the loads of input plane t (tick phase k).  `pin`: the step is
      straight-line code (peeled warm-up); its loads take their lane offset
      from an opaque copy made here, or the compiler hoists the loads of ALL
      peeled steps to the top of the kernel (T = 12: 229 instead of 159
      VGPRs)."""
      self.w('      const int t = %s;' % t_expr)
      self.w('      const bool plane_ok = t >= 0 && t < in_end;')
      if pin and self.buf:
        for es in sorted({self.esz[nme] for nme in self.inputs}):
          self.w('      unsigned xb%dp = xb%d; asm volatile("" : "+v"(xb%dp));'
                 % (es, es, es))
      pinned = 'p' if pin and self.buf else ''
      if self.buf:
        in_es = sorted({self.esz[nme] for nme in self.inputs})
        for es in in_es:
          if self.dim == 2:
            self.w('      const unsigned ro%d = plane_ok ? (unsigned)(t - wlo) * '
              'pitch_b%d : SODA_OOB_ROW;' % (es, es))
          else:
            # plane part + row part (loop-invariant, emitted once) + lane
            # part, each either in range or 2^30: the sum is out of range as
            # soon as one of them is, and cannot wrap (soda_rt.h)
            self.w('      const unsigned po%d = plane_ok ? (unsigned)(t - wlo) * '
              'pitch_b%d : SODA_OOB_X;' % (es, es))
            for j in sorted({j for n in self.inputs.values() for j in self.rows_of(n)}):
              self.w('      const unsigned ro%d_%d = po%d + yo%d_%d;' %
                     (es, j, es, es, j))
      for nme, n in self.inputs.items():
        s = self.slot_of(n, k, 0)
        for j in (self.rows_of(n) if self.buf else []):
          es = self.esz[nme]
          ro = 'ro%d' % es if self.dim == 2 else 'ro%d_%d' % (es, j)
          reg = '%s_s%d_r%d' % (n.var, s, j)
          self.w('      soda_buf_load_frag<%s, %d, %s>(%s, r_%s, %s + xb%d%s);' %
            (n.ctype, self.V, self.nt_l, reg, nme, ro, es, pinned))
          for i in range(self.n_edge):
            self.w('      { %s e1[1]; soda_buf_load_frag<%s, 1, false>(e1, r_%s, %s + '
              'exb%d_%d); %s_e[%d] = e1[0]; }' %
              (n.ctype, n.ctype, nme, ro, i, es, reg, i))
        for j in ([] if self.buf else self.rows_of(n)):
          if self.dim == 3:
            cond = 'plane_ok && y0 + %d >= 0 && y0 + %d < n1' % (j, j)
            addr = 'p_%s + (int64_t)t * pitch + (int64_t)(y0 + %d) * pitch_y' % (
                nme, j)
          else:
            cond = 'plane_ok'
            addr = 'p_%s + (int64_t)t * pitch' % nme
          reg = '%s_s%d_r%d' % (n.var, s, j)
          self.w('      if (lane_ok && %s) soda_load_frag<%s, %d, %s>(%s, %s);' %
            (cond, n.ctype, self.V, self.nt_l, reg, addr))
          self.w('      else soda_zero_frag<%s, %d>(%s);' % (n.ctype, self.V, reg))
          for i in range(self.n_edge):
            self.w('      %s_e[%d] = (edge_ok%d && %s) ? (%s)[eoff%d] : (%s)0;' %
              (reg, i, i, cond, addr, i, n.ctype))

  def _emit_wave(self, wv: int) -> None:
    """The tick loop of one wave of the block (the only one unless the
    block is stage-pipelined)."""
    if self.W > 1:
      self.w('  if (wave == %d) {  // iterations %d..%d of the chain' %
             (wv, wv * (self.T // self.W), (wv + 1) * (self.T // self.W) - 1))
    self.emit_decls(wv)
    # The first `lead` ticks of a chunk only fill the prefetch queue: nothing a
    # stage could compute then reaches an output plane (every stage lags its
    # inputs by at least the prefetch depth), so they are peeled into a
    # load-only prologue whose slot phases continue into tick 0 of the loop.
    self.w('  int tau = m_begin + (%d);' % (self.m_lo + self.lead))
    if wv == 0:
      for i in range(self.lead):
        self.w('    {  // prologue: loads of plane tau - %d' % (self.lead - i))
        self.emit_loads((i - self.lead) % self.U, 'tau - %d' % (self.lead - i))
        self.w('    }')
    self.w('  const int tau_end = m_end + %d;' % self.max_delay)
    if self.peeled:
      self.w('  {  // warm-up: %d row steps, stages enter as their rows start '
             'to matter' % self.peeled)
      for i in range(self.peeled):
        self._emit_tick(wv, i % self.U, step=i)
      self.w('  }')
      self.w('  tau += %d;' % self.peeled)
    for n in self.nodes:
      if id(n) in self.slide and n.owner == wv and not (
          self.peeled and self.stage_needed(n, self.peeled - 1)):
        # the stage's first step is the loop's first: set its accumulator up
        self.w('  {  // sliding sum of %s: the rows its window holds now' % n.var)
        self.w('    const int t = tau;')
        self._shifted = {}
        self._wide = {}
        self._lx_read = set()
        self._stage_mark = len(self.L)
        self._emit_stage(n, 0, slide_init='only')
        self.w('  }')
    self.w('  for (; tau < tau_end; tau += %d) {' % self.U)
    for k in range(self.U):
      self._emit_tick(wv, k)
    self.w('  }')
    if self.W > 1:
      self.w('  }')

  def _emit_tick(self, wv: int, k: int, step: Optional[int] = None) -> None:
    """One row step at slot phase k; `step` = its number within the peeled
    warm-up (None: a step of the loop)."""
    at = k if step is None else step
    self.w('    {  // tick %d of %d' % (k, self.U))

    if self.W > 1 and k % self.R == 0:
      self.w('      soda_pipe_barrier();')
    if wv == 0:
      self.emit_loads(k, 'tau + %d' % at, pin=step is not None)
    else:
      self.w('      const int t = tau + %d;' % at)
    for n in self.nodes:
      if n.mirror_of is not None and n.owner == wv:
        # the plane the previous wave wrote R ticks (one barrier) ago
        for j in self.rows_of(n):
          self.w('      soda_load_frag<%s, %d, false>(%s_s%d_r%d, &soda_ring_%s'
                 '[(t + %d) & %d][%d][lane * %d]);' %
                 (n.ctype, self.V, n.var, self.slot_of(n, k, 0), j,
                  n.mirror_of.var, self.R, 2 * self.R - 1, j, self.V))
    # 2. compute every tensor's new plane
    self._shifted: Dict[Tuple[str, int, int, int, int], str] = {}
    self._wide: Dict[str, str] = {}   # one-byte cells widened in this step
    self._lx_read = set()      # ('lds' shifts) lines already read in this step
    self._stage_mark = len(self.L)
    # (tensor, slot) whose end cells change hands at the end of this step: the
    # planes computed in it, and the input plane whose load it is first to use
    fresh_xs: List[Tuple[_Node, int]] = []
    if self.xs and wv == 0:
      for n in self.inputs.values():
        if n.xs:
          fresh_xs.append((n, self.slot_of(n, k, self.PF)))
      # the neighbours' end cells of the planes handed over at the end of the
      # previous step: one LDS read per tensor, row r in lanes r and 32 + r
      # (stale or unwritten values reach only planes no output depends on)
      for n in self.nodes:
        if not n.xs:
          continue
        slot = self.slot_of(n, k - 1, self.PF if n.is_input else 0)
        self.w('      hv_%s_s%d = soda_xs_%s[xs_rd < %d ? xs_rd + ((t - 1) & 1) * %d '
               ': xs_rd];' % (n.var, slot, n.var, 2 * self.xs_stride,
                              self.xs_stride))
    for n in self.nodes:
      if n.stage is not None and n.owner == wv:
        if step is not None and not self.stage_needed(n, step):
          continue
        first = step is not None and (step == 0 or
                                      not self.stage_needed(n, step - 1))
        self._emit_stage(n, k, slide_init='first' if first else None)
        if n.xs:
          fresh_xs.append((n, self.slot_of(n, k, 0)))
    self.shift_temps = max(self.shift_temps, len(self._shifted))
    if fresh_xs:
      # x-halo hand-over of the planes computed (first read, for an input) in
      # this step: lanes 0 and 63 put their end cells into LDS; the barrier
      # also keeps the next write of this parity behind every wave's reads
      self.w('      if (edge_lane) {')
      for n, slot in fresh_xs:
        for jj, j in enumerate(self.rows_of(n)):
          reg = '%s_s%d_r%d' % (n.var, slot, j)
          self.w('        soda_xs_%s[xs_wr + (t & 1) * %d + %d] = lane == 0 ? %s[0] '
                 ': %s[%d];' % (n.var, self.xs_stride, jj, reg, reg, self.V - 1))
      self.w('      }')
      self.w('      soda_pipe_barrier();')
    self.w('    }')

  def _lx_reads(self, n: _Node) -> List[str]:
    lo, hi = self.ldsx[id(n)]
    # compiler fences (no instruction): a lane reads what ANOTHER lane wrote,
    # which single-thread reordering rules know nothing about -- the reads must
    # stay behind the previous step's writes and ahead of this step's
    out = ['      asm volatile("" ::: "memory");']
    for i in range(lo):
      out.append('      const %s lx_%s_L%d = soda_lx_%s_L%d[lane];'
                 % (n.ctype, n.var, i, n.var, i))
    for i in range(hi):
      out.append('      const %s lx_%s_R%d = soda_lx_%s_R%d[lane + 1];'
                 % (n.ctype, n.var, i, n.var, i))
    return out

  def _emit_stage(self, n: _Node, k: int,
                  slide_init: Optional[str] = None) -> None:
    """One tensor's new plane at tick phase k: lane-shifted operands first
    (shared by the stages of a tick), then one statement per cell.
    `slide_init` (sliding sums only): 'first' -- this is the stage's first
    step, set the accumulator up in front of it; 'only' -- emit nothing but
    that set-up (the first step is the loop's first)."""
    stage = n.stage
    pre: List[str] = []
    guard = None
    if self.cfg.warm_guards and self.back_lo[id(n)] is not None:
      first = self.back_lo[id(n)] + n.delay   # tick offset from m_begin
      if first > self.m_lo:                   # the loop starts at m_begin + m_lo
        guard = 't >= m_begin + (%d)' % first
        self._shifted = {}                     # temporaries live inside the guard
        self._wide = {}

    early: List[str] = []

    def operand(pname: str, off: Tuple[int, ...], j: int, e: int, _n=n,
                _k=k, _pre=pre, _early=early) -> str:
      p = _n.parents[pname]
      age = _n.delay - off[self.ax] - p.fill_delay
      slot = self.slot_of(p, _k, age)
      row = j + (off[1] if self.dim == 3 else 0)
      if row < p.rmargin[0] or row >= self.rows_in - p.rmargin[1]:
        raise util.InternalError('march: row %d of %s is not held' %
                                 (row, p.var))
      reg = '%s_s%d_r%d' % (p.var, slot, row)
      c = e + off[0]
      lane_off, sub = c // self.V, c % self.V
      src = '%s[%d]' % (reg, sub)
      # one-byte cells enter an expression as the int C promotes them to, with
      # the value range hidden from the compiler (soda_rt.h soda_wide: hipcc's
      # packed-byte instruction selection is wrong in places), once per cell
      # and tick
      byte = p.ctype in ('uint8_t', 'int8_t')

      def wide(text: str, tag: str) -> str:
        if not byte:
          return text
        if tag not in self._wide:
          self._wide[tag] = 'w%d_%s' % (len(self._wide), tag)
          _pre.append('      const int %s = soda_wide(%s);' %
                      (self._wide[tag], text))
        return self._wide[tag]

      if lane_off == 0:
        return wide(src, '%s_e%d' % (reg, sub))
      if self.cfg.lane_shift == 'none':
        return src       # TIMING EXPERIMENTS ONLY: wrong results
      if id(p) in self.ldsx and abs(lane_off) == 1 and age == 1:
        # the neighbouring lane filed this cell one row step ago
        i = (-c - 1) if lane_off < 0 else (c - self.V)
        name = 'lx_%s_%s%d' % (p.var, 'L' if lane_off < 0 else 'R', i)
        if (p.var, _k) not in self._lx_read:
          # (the tensor's own stage is skipped in this peeled step, so its
          # line still holds the row wanted: read it here)
          self._lx_read.add((p.var, _k))
          _pre.extend(self._lx_reads(p))
        return wide(name, name)
      key = (p.var, slot, row, sub, lane_off)
      if key not in self._shifted:
        tmp = 'sh_%s_e%d_%s%d' % (reg, sub, 'm' if lane_off < 0 else 'p',
                                  abs(lane_off))
        if (p.is_input and self.n_edge) or p.xs:
          if abs(lane_off) != 1:
            raise util.InternalError('march: edge loads reach one lane')
          # cell index relative to the strip end, served by the edge lane
          ei = (-c - 1) if lane_off < 0 else (c - self.V)
          if p.xs:
            # broadcast from the halo vector of the plane (a scalar register)
            rr = row - p.rmargin[0] + (0 if lane_off < 0 else 32)
            old = ('soda_bcast(hv_%s_s%d, %d)' % (p.var, slot, rr)
                   if ei == 0 else '(%s)0' % p.ctype)
          else:
            old = '%s_e[%d]' % (reg, ei) if 0 <= ei < self.n_edge else \
                '(%s)0' % p.ctype
          expr = ('soda_lane_dn_or(%s, %s)' if lane_off < 0 else
                  'soda_lane_up_or(%s, %s)') % (src, old)
        elif self.use_bperm:
          expr = src
          for _ in range(abs(lane_off)):
            expr = 'soda_lane_from(%s, %s)' % (
                'lane_dn_addr' if lane_off < 0 else 'lane_up_addr', expr)
        elif self.use_swz or (self.use_mix and lane_off > 0):
          expr = src
          for _ in range(abs(lane_off)):
            expr = 'soda_lane_%s%d%s(%s)' % (
                'dn' if lane_off < 0 else 'up', self.group,
                'd' if self.cfg.lane_shift == 'mix64d' else '', expr)
        else:
          expr = src
          for _ in range(abs(lane_off)):
            expr = ('soda_lane_dn(%s)' if lane_off < 0 else
                    'soda_lane_up(%s)') % expr
        line = '      const %s %s = %s;' % (p.ctype, tmp, expr)
        # a shift of a row produced in an EARLIER tick can be issued ahead
        # of the previous stage's arithmetic (latency hidden behind it)
        early = (self.use_bperm or self.use_swz or
                 (self.use_mix and lane_off > 0)) and (
            p.is_input or age > 0) and not (p.is_input and self.n_edge) \
            and not p.xs
        (_early if early else _pre).append(line)
        self._shifted[key] = tmp
      return wide(self._shifted[key], self._shifted[key])

    body: List[str] = []
    dst_slot = self.slot_of(n, k, 0)
    xwin = self._xwindow(stage) if n.keep is None and not guard else None
    slide = self.slide.get(id(n))
    if slide is not None:
      pname, off, taps = slide
      for j in self.rows_of(n):
        acc = 'xa_%s_r%d' % (n.var, j)
        for e in range(self.V):
          def tap(i, _j=j, _e=e):
            o = list(off)
            o[self.ax] = off[self.ax] + i
            return '(int)%s' % operand(pname, tuple(o), _j, _e)
          if slide_init:
            body.append('      %s[%d] = %s;' % (acc, e, ' + '.join(
                tap(i) for i in range(taps - 1))))
          if slide_init != 'only':
            body.append('      %s[%d] += %s;' % (acc, e, tap(taps - 1)))
            body.append('      %s_s%d_r%d[%d] = (%s)%s[%d];' %
                        (n.var, dst_slot, j, e, n.ctype, acc, e))
            body.append('      %s[%d] -= %s;' % (acc, e, tap(0)))
      if slide_init == 'only':
        self.L.extend(early)     # (shifts on the LDS pipe: nothing to hide behind here)
        self.L.extend(pre)
        self.L.extend(body)
        return
    elif xwin is not None:
      self._emit_xwindow(n, stage, xwin, dst_slot, operand, body)
    elif self.cfg.interleave and not stage.stmt.let:
      # all cells of the row tile at once, operation-major
      cells = [(j, e) for j in self.rows_of(n) for e in range(self.V)]

      def mk_load(j, e, _stage=stage):
        def load(ref: ir.Ref) -> str:
          prm = self.mod.param_load(ref)
          if prm is not None:
            return prm
          off = tuple(a - b for a, b in zip(ref.idx, _stage.st_idx))
          return operand(ref.name, off, j, e)
        return load

      counter = [0]

      def fresh(_n=n, _k=k) -> str:
        counter[0] += 1
        return 'v_%s_k%d_%d' % (_n.var, _k, counter[0])

      stmts, results = ir.c_statements(stage.stmt.expr,
                                       [mk_load(j, e) for j, e in cells],
                                       fresh, var=self.mod.param_var)
      body.extend('      ' + x for x in stmts)
      for (j, e), r in zip(cells, results):
        body.append('      %s_s%d_r%d[%d] = (%s)(%s);' %
                    (n.var, dst_slot, j, e, n.ctype, r))
    for j in ([] if (xwin is not None or slide is not None or
                     (self.cfg.interleave and not stage.stmt.let))
              else self.rows_of(n)):
      for e in range(self.V):

        def load(ref: ir.Ref, _e=e, _j=j, _stage=stage) -> str:
          prm = self.mod.param_load(ref)
          if prm is not None:
            return prm
          off = tuple(a - b for a, b in zip(ref.idx, _stage.st_idx))
          return operand(ref.name, off, _j, _e)

        dst = '%s_s%d_r%d[%d]' % (n.var, dst_slot, j, e)
        if stage.stmt.let:
          body.append('      {')
          for let in stage.stmt.let:
            body.append('        const %s %s = %s;' %
                        (let.haoda_type.c_type, let.name,
                         ir.c_expr(let.expr, load, self.mod.param_var)))
          body.append('        %s = (%s)(%s);' %
                      (dst, n.ctype, ir.c_expr(stage.stmt.expr, load,
                                               self.mod.param_var)))
          body.append('      }')
        else:
          body.append('      %s = (%s)(%s);' %
                      (dst, n.ctype, ir.c_expr(stage.stmt.expr, load,
                                               self.mod.param_var)))
    if guard:
      self.w('      if (%s) {  // wave-uniform' % guard)
      self.L.extend(early)
    elif early:
      # place them in front of the previous stage's block of this tick.
      # (The compiler sinks each to one stage's arithmetic ahead of its use,
      # one swizzle in flight at a time.  Round 4 pinned all 13 of a T=13 step
      # at the head of the step behind a scheduling barrier -- 13 in flight,
      # 162 VGPRs, bit-exact -- and gained nothing: T12 145-146 us either way,
      # T13 170 against 156-161, profiles/r04_early_*.json.  The swizzles'
      # latency is not what the step waits for.)
      self.L[self._stage_mark:self._stage_mark] = early
    if n.keep is not None:
      # border: preserve -- cells one iteration cannot compute keep the value
      # of the input this output replaces (same cell, previous iteration)
      wlo, whi = self.st.interior_bounds(stage.name)
      zero = (0,) * self.dim
      body.append('      {')
      body.append('        const int mk = a.origin[%d] + t - %d;  // global plane'
                  % (self.ax, n.delay))
      body.append('        const bool keep_m = !(mk >= %d && mk < a.gextent[%d] - '
                  '%d);' % (max(0, -wlo[self.ax]), self.ax,
                            max(0, whi[self.ax])))
      for j in self.rows_of(n):
        keep_row = 'keep_m'
        if self.dim == 3:
          body.append('        const bool keep_r%d = keep_m || !(a.origin[1] + '
                      'y0 + %d >= %d && a.origin[1] + y0 + %d < a.gextent[1] - '
                      '%d);' % (j, j, max(0, -wlo[1]), j, max(0, whi[1])))
          keep_row = 'keep_r%d' % j
        for e in range(self.V):
          dst = '%s_s%d_r%d[%d]' % (n.var, dst_slot, j, e)
          body.append('        %s = (%s || keepx_%s_%d) ? %s : %s;' %
                      (dst, keep_row, stage.name, e,
                       operand(n.keep, zero, j, e), dst))
      body.append('      }')
    self._stage_mark = len(self.L)
    if id(n) in self.ldsx and (n.var, k) not in self._lx_read:
      # the cells the neighbouring lanes filed one row step ago, fetched
      # BEFORE this step's row overwrites the line; consumed by the stages
      # that follow in this step
      self._lx_read.add((n.var, k))
      self.L.extend(self._lx_reads(n))
    self.L.extend(pre)
    self.L.extend(body)
    if self.st.symbol_table[stage.name].size_in_bytes == 1:
      # every cell of a one-byte tensor through a register of its own: left to
      # itself the compiler packs the cells of a row into one register, and one
      # of its instruction-selection combines on that form is wrong on gfx950
      # (ROCm 7.2: a min whose operands come out of the packed register picked
      # the wrong side in one cell of one unrolled step of a fused kernel --
      # tools/fuzz_scan.py options, seed 613; exact with the pass that selects
      # instructions run unoptimised)
      for j in self.rows_of(n):
        self.w('      soda_own_register<%s, %d>(%s_s%d_r%d);' %
               (n.ctype, self.V, n.var, dst_slot, j))
    if id(n) in self.ldsx:
      lo, hi = self.ldsx[id(n)]
      reg = '%s_s%d_r0' % (n.var, dst_slot)
      self.w('      asm volatile("" ::: "memory");')
      for i in range(lo):
        self.w('      soda_lx_%s_L%d[lane + 1] = %s[%d];' %
               (n.var, i, reg, self.V - 1 - i))
      for i in range(hi):
        self.w('      soda_lx_%s_R%d[lane] = %s[%d];' % (n.var, i, reg, i))
    if guard:
      self.w('      }')
      self._shifted = {}
      self._wide = {}
    if n.to_lds:   # hand the new plane to the next wave of the block
      for j in self.rows_of(n):
        self.w('      soda_store_frag<%s, %d, false>(&soda_ring_%s[t & %d][%d]'
               '[lane * %d], %s_s%d_r%d);' %
               (n.ctype, self.V, n.var, 2 * self.R - 1, j, self.V, n.var,
                dst_slot, j))
    if n.store_slot is not None:
      self._emit_store(n, stage.name, dst_slot)

  # -- integer window reductions along dimension 0, all cells of a lane jointly --
  def _xwindow(self, stage):
    """(op, parent, first offset relative to the cell, taps) if `stage` is an
    association-free reduction over a contiguous run of dimension-0 taps of one
    tensor (optimization/windows.py states when it is), else None."""
    if not self.cfg.xwindow:
      return None
    from soda_amd.optimization import windows
    table = dict(self.st.symbol_table)
    m = windows._match(stage.stmt, table)
    if m is None or m[2] != 0:
      return None
    op, parent, _, _, taps, base = m
    off = tuple(a - b for a, b in zip(base, stage.st_idx))
    return op, parent, off, taps

  def _emit_xwindow(self, n: _Node, stage, xwin, dst_slot: int, operand,
                    body: List[str]) -> None:
    """A window of `taps` cells along dimension 0, for the V cells of a lane at
    once: the V + taps - 1 cells the lane's windows cover are brought into the
    lane ONCE (whole fragments of the lanes to the right: ~ (taps - 1) lane
    moves instead of a move per tap per cell), then

      +        out[0] = the first window, out[e + 1] = out[e] + in[e + taps]
               - in[e]: taps - 1 + 2 (V - 1) operations for V cells, in int32
               (exact: the values are narrower than 32 bits, windows.py);
      min/max  in blocks of b = min(V, taps) cells: the windows of a block
               share the middle cells in[e0+b-1 .. e0+taps-1]; suffix
               reductions of in[e0 .. e0+b-2] and prefix reductions of
               in[e0+taps .. e0+taps+b-2] complete them: ~ taps + 3 b
               operations per block.  (One block only when V <= taps; with
               more cells per lane than taps -- uint8: 16 -- no cell is common
               to all V windows.)

    Integer arithmetic in any order gives the same bits as the statement's
    left-to-right text; the statement's cast is applied per cell as before."""
    op, pname, off, taps = xwin
    V = self.V
    for j in self.rows_of(n):
      tag = '%s_k%d_r%d' % (n.var, dst_slot, j)
      cells = []
      for i in range(V + taps - 1):
        o = list(off)
        o[0] = off[0] + i
        cells.append(operand(pname, tuple(o), j, 0))
      dst = ['%s_s%d_r%d[%d]' % (n.var, dst_slot, j, e) for e in range(V)]
      body.append('      {')
      if op == '+':
        body.append('        int xw_%s = %s;' % (tag, ' + '.join(
            '(int)%s' % c for c in cells[:taps])))
        body.append('        %s = (%s)xw_%s;' % (dst[0], n.ctype, tag))
        for e in range(1, V):
          body.append('        xw_%s = xw_%s + (int)%s - (int)%s;' %
                      (tag, tag, cells[e + taps - 1], cells[e - 1]))
          body.append('        %s = (%s)xw_%s;' % (dst[e], n.ctype, tag))
      else:
        fn = 'SODA_MIN' if op == 'min' else 'SODA_MAX'
        pt = self.st.symbol_table[pname].c_type
        # blocks of at most `taps` cells: the windows of cells e0 .. e0+b-1
        # all contain in[e0+b-1 .. e0+taps-1] (b <= taps, so never empty)
        for blk, e0 in enumerate(range(0, V, taps)):
          b = min(taps, V - e0)
          lo, hi = e0 + b - 1, e0 + taps - 1
          btag = '%s_b%d' % (tag, blk)
          mid = '(%s)%s' % (pt, cells[lo])
          for c in cells[lo + 1:hi + 1]:
            mid = '%s((%s)%s, %s)' % (fn, pt, c, mid)
          body.append('        const %s xm_%s = %s;' % (pt, btag, mid))
          # suffix reductions of the cells below the middle block ...
          suf = {}
          prev = None
          for i in range(lo - 1, e0 - 1, -1):
            name = 'xs_%s_%d' % (btag, i)
            expr = '(%s)%s' % (pt, cells[i]) if prev is None else \
                '%s((%s)%s, %s)' % (fn, pt, cells[i], prev)
            body.append('        const %s %s = %s;' % (pt, name, expr))
            suf[i] = prev = name
          # ... and prefix reductions of the cells above it
          pre = {}
          prev = None
          for i in range(hi + 1, e0 + b + taps - 1):
            name = 'xp_%s_%d' % (btag, i)
            expr = '(%s)%s' % (pt, cells[i]) if prev is None else \
                '%s(%s, (%s)%s)' % (fn, prev, pt, cells[i])
            body.append('        const %s %s = %s;' % (pt, name, expr))
            pre[i] = prev = name
          for e in range(e0, e0 + b):
            parts = ['xm_%s' % btag]
            if e < lo:
              parts.append(suf[e])
            if e + taps - 1 > hi:
              parts.append(pre[e + taps - 1])
            expr = parts[0]
            for q in parts[1:]:
              expr = '%s(%s, %s)' % (fn, expr, q)
            body.append('        %s = (%s)(%s);' % (dst[e], n.ctype, expr))
      body.append('      }')

  def _emit_store(self, n: _Node, oname: str, dst_slot: int) -> None:
    """Stores the new plane of a last-iteration output (rows and lanes that
    are not this wave's to write are dropped by the addressing)."""
    if self.buf:
      es = self.esz[oname]
      self.w('      {')
      self.w('        const int m = t - %d;' % n.delay)
      self.w('        const bool m_ok = m >= m_begin && m < m_end;')
      for j in self.store_rows(n):
        reg = '%s_s%d_r%d' % (n.var, dst_slot, j)
        if self.dim == 3:
          self.w('        soda_buf_store_frag<%s, %d, %s>(w_%s, (m_ok ? (unsigned)'
            '(m - m_begin) * pitch_b%d : SODA_OOB_X) + yo%d_%d + sxb%d, %s);' %
            (n.ctype, self.V, self.nt_s, oname, es, es, j, es, reg))
        else:
          self.w('        soda_buf_store_frag<%s, %d, %s>(w_%s, (m_ok ? '
            '(unsigned)(m - m_begin) * pitch_b%d : SODA_OOB_ROW) + sxb%d, '
            '%s);' % (n.ctype, self.V, self.nt_s, oname, es, es, reg))
      self.w('      }')
      return
    self.w('      {')
    self.w('        const int m = t - %d;' % n.delay)
    self.w('        if (store_ok && m >= m_begin && m < m_end) {')
    for j in range(max(n.rmargin[0], self.rhalo_lo),
                   self.rows_in - max(n.rmargin[1], self.rhalo_hi)):
      reg = '%s_s%d_r%d' % (n.var, dst_slot, j)
      if self.dim == 3:
        self.w('          if (y0 + %d < n1) soda_store_frag<%s, %d, %s>(q_%s + '
          '(int64_t)m * pitch + (int64_t)(y0 + %d) * pitch_y, %s);' %
          (j, n.ctype, self.V, self.nt_s, oname, j, reg))
      else:
        self.w('          soda_store_frag<%s, %d, %s>(q_%s + (int64_t)m * pitch, '
          '%s);' % (n.ctype, self.V, self.nt_s, oname, reg))
    self.w('        }')
    self.w('      }')

  def _finish(self) -> PassDesc:
    if self.shift_temps > MAX_SHIFT_TEMPS:
      raise util.SemanticError(
          'march: %d lane-shifted operands per row step (> %d); the taps reach '
          'too far along dim 0 for %d cells per lane' %
          (self.shift_temps, MAX_SHIFT_TEMPS, self.V))
    self.est_regs += self.shift_temps
    # launch-time model (include/soda_hip.h, soda_hip_kernel_desc_t): vector
    # instructions one row step of the busiest wave issues, and the part of
    # the warm-up the peeled steps skip, in row steps
    stages = [n for n in self.nodes if n.stage is not None]
    per_wave = []
    for wv in range(self.W):
      ops = 0.0
      for n in stages:
        if n.owner != wv:
          continue
        cost = sum(ir.op_count(l.expr) for l in n.stage.stmt.let) + \
            ir.op_count(n.stage.stmt.expr) + 0.5
        wide = 2.0 if n.ctype in ('double', 'int64_t', 'uint64_t') else 1.0
        ops += cost * wide * self.V * len(self.rows_of(n))
      per_wave.append(ops + 8.0)
    step_ops = max(per_wave)
    saved = sum(1 for i in range(self.peeled) for n in stages
                if not self.stage_needed(n, i)) / float(max(1, len(stages)))
    if self.dim == 2:
      tile = (self.strip_cells * (self.xs or self.wx),
              self.cfg.chunk_rows * self.cfg.waves_y)
    else:
      tile = (self.strip_cells * (self.xs or 1), self.tile_rows,
              self.cfg.chunk_rows)
    lds_pad = 0
    if self.cfg.occupancy:
      blocks_per_cu = max(1, 4 * self.cfg.occupancy // self.waves)
      # smallest allocation of which blocks_per_cu + 1 no longer fit
      lds_pad = min(65536, (LDS_PER_CU // (blocks_per_cu + 1)) // 1024 * 1024
                    + 1024)
    idx = self.mod.add_kernel(
        KernelDesc(self.name, (self.block, 1, 1), tile, lds_bytes=lds_pad,
                   note='%s %s' % (self.kind, self.cfg.key()),
                   tune=dict(axis=self.ax, waves_along=self.cfg.waves_y if self.dim == 2 else 1,
                             waves_per_block=self.waves, warm=self.warm,
                             fixed=self.cfg.chunk_fixed, occupancy=self.cfg.occupancy,
                             pipe=self.W, vec=self.V, step_ops=step_ops,
                             lane_shift=self.cfg.lane_shift,
                             warm_saved=saved,
                             lane_redundancy=64.0 / self.strip_lanes,
                             peel_trips=self.peeled // self.U,
                             unroll=self.U, tile_rows=self.tile_rows,
                             peel_trips_max=self.peel_trips_max,
                             fused=self.T,
                             window_extra=(self.m_hi - self.m_lo) if self.buf else None,
                             max_elem=max(self.esz.values()),
                             max_extent0=self.strip_cells * self.xs)),
        '\n'.join(self.L) + '\n')
    table = self.st.symbol_table
    bytes_in = sum(table[i].size_in_bytes for i in self.st.input_names)
    bytes_out = sum(table[o].size_in_bytes for o in self.st.output_names)
    redundancy = (64.0 / self.strip_lanes) * (
        (self.cfg.chunk_rows + self.warm) / float(self.cfg.chunk_rows)) * (
            self.rows_in / float(self.tile_rows))
    p = PassDesc(
        self.T, [idx], self.kind,
        dict(bytes_per_cell_min=bytes_in + bytes_out,
             read_redundancy=redundancy, strip_cells=self.strip_cells, warm_rows=self.warm,
             unroll=self.U, edge=self.edge, rows_in=self.rows_in, tile_rows=self.tile_rows,
             est_window_regs=self.est_regs))
    self.mod.passes.append(p)
    return p


def add_march_pass(mod: Module, cfg: MarchConfig) -> PassDesc:
  """Adds one marching kernel (and its pass) for `cfg.fused_iters` iterations;
  raises SemanticError if the program or the shape does not fit."""
  return _MarchKernel(mod, cfg).emit()


add_march2d_pass = add_march_pass


