"""Lowering of a Stencil to gfx950 HIP kernels + a launch plan.

What the reference's HLS emitter does for an FPGA (reference
src/soda/codegen/xilinx/hls_kernel.py:338-410 `print_code`, :665-886
`print_module_definition`; src/soda/dataflow.py:336-625) is done here for a
GPU.  Two families of kernels are generated:

`direct`   one kernel per stage, 16 bytes' worth of cells per thread, parents
           read straight from global memory into per-thread row buffers
           (L1/L2 give the reuse between threads).  Handles every program
           the front-end accepts (1-4 dimensions, any DAG); locals live in
           HBM scratch.  It is the correctness baseline and the fallback.

`march2d`  2-D programs.  One wavefront owns a strip of 64*V columns and
           marches along dimension 1 (the dimension SODA streams).  Every
           tensor of the -- possibly T-times unrolled -- stage chain keeps a
           sliding window of rows in REGISTERS; neighbours along dimension 0
           come from adjacent lanes by DPP wave shifts; inputs arrive by one
           coalesced 16-byte load per lane per row, PF rows ahead of use;
           only the last iteration's outputs are stored.  That is SODA's own
           micro-architecture (line buffers chained through all iterations so
           DRAM is touched once; reference core.py:338-369, README.md:155-156)
           mapped onto a wavefront: the FIFOs become registers, the `unroll
           factor` PEs become 64*V lanes, and `iterate` stages fused in one
           pipeline become T fused iterations per launch (temporal blocking).
           Waves never talk to each other (no LDS, no barrier): the halo a
           strip needs is recomputed by overlapping strips/chunks.

Floating-point results are bit-identical to the CPU oracle because the kernels
compile the same C expression text with -ffp-contract=off.

Layout of the package: module.py (Module, descriptors, the device runtime
text), march.py (`march2d` / `march3d`), direct.py, lds2d.py (the classic LDS
halo tile, kept as the measured alternative); this file holds the options and
`lower()`, which picks the shape.
"""
import os
from typing import Optional, Sequence

from soda_amd import core, util

from soda_amd.codegen.hip.module import (KernelDesc, Module, PassDesc,  # noqa: F401
                                         _check_native, runtime_text)
from soda_amd.codegen.hip.direct import DIRECT_BLOCK, add_direct_pass  # noqa: F401
from soda_amd.codegen.hip.lds2d import (add_lds2d_pass,  # noqa: F401
                                        lds2d_supported)
from soda_amd.codegen.hip.ldswin import (add_ldswin_pass,  # noqa: F401
                                         ldswin_candidate, ldswin_pays,
                                         ldswin_supported)
from soda_amd.codegen.hip.march import (MAX_FUSE_3D, MAX_FUSE_PRESERVE,  # noqa: F401
                                        MAX_SHIFT_TEMPS,
                                        MAX_UNROLL, REG_BUDGET, MarchConfig,
                                        add_march_pass, default_vec,
                                        march_supported)

# ---------------------------------------------------------------------------
# whole module
# ---------------------------------------------------------------------------

MAX_TENSORS = 16       # SODA_HIP_MAX_TENSORS (include/soda_hip.h)
# The fusion depths iterated programs are offered by default -- ONE definition
# for `sodac --hip-fuse`, bench.py, __graft_entry__.prebuild_list and the parity
# tests of the benched schedule (tests/test_hip_parity.py): the library mixes
# them per extent (jacobi2d 8192^2 x 100 = 4 x T13 + 4 x T12).
DEFAULT_FUSE = (13, 12, 8, 4)
# integer sums along the streamed dimension as sliding sums in the marching
# kernels (march.py `slide`) instead of power-of-two chains (SODA_HIP_SLIDE=0/1)
SLIDING_SUMS = os.environ.get('SODA_HIP_SLIDE', '1') != '0'


class LowerOptions:
  """Knobs of the HIP backend (command-line spelling: --hip-*).  `None` means
  "the measured default for this program's dimensionality" (profiles/
  r01_sweeps.md): 2-D strips prefetch 2 rows with non-temporal loads (every
  input row is used by one wave only); 3-D tiles re-read their halo rows, so
  they use plain loads (the overlap then hits in L2), prefetch 1 plane and hold
  4 output rows per wave to stay under 128 VGPRs."""

  def __init__(self, strategy: str = 'auto', fuse: Sequence[int] = (4,),
               vec: Optional[int] = None, chunk_rows: Optional[int] = None,
               prefetch: Optional[int] = None, waves_x: int = 1,
               waves_y: int = 1, nt_store: Optional[bool] = None,
               nt_load: Optional[bool] = None, xcd_swizzle: bool = True,
               edge_loads: bool = True, tile_rows: Optional[int] = None,
               warm_guards: bool = False, interleave: bool = False,
               lane_shift: Optional[str] = None, min_waves: int = 0,
               occupancy: int = 0, buffer_ops: bool = True,
               pipe: Optional[int] = None, pipe_rows: int = 2,
               reg_budget: Optional[int] = None,
               stamps: bool = False,
               peel=None, align_lanes: Optional[int] = None,
               xshare: Optional[bool] = None,
               row_cells: Optional[int] = None,
               windows: Optional[bool] = None,
               inline: Optional[bool] = None):
    # locals read only at the cell being computed are folded into their
    # consumers (optimization/pointwise.py): denoise3d is 4 tensors instead
    # of 10.  SODA_HIP_INLINE=0/1 overrides for A/B runs
    if inline is None:
      inline = os.environ.get('SODA_HIP_INLINE', '1') != '0'
    self.inline = inline
    # long integer window reductions (erosion's 19-tap min, xcorr's 19-tap
    # sums) as chains of power-of-two windows: 6 instead of 18 operations per
    # cell, bit-exact (optimization/windows.py).  SODA_HIP_WINDOWS=0/1
    # overrides for A/B runs
    if windows is None:
      windows = os.environ.get('SODA_HIP_WINDOWS', '1') != '0'
    self.windows = windows
    self.stamps = stamps
    # fused 3-D kernels whose block covers the whole row, x-halos handed over
    # through LDS (MarchConfig.xshare).  Needs the row length the program will
    # run on (`row_cells` = extent[0]; runtime.Program fills it in from its
    # extent): the kernels then REFUSE any other row length at launch.
    # None: where it applies (3-D, fused, <= 4 waves per row); True also in
    # 2-D; SODA_HIP_XSHARE=0/1 overrides for A/B runs
    if xshare is None and os.environ.get('SODA_HIP_XSHARE'):
      xshare = os.environ['SODA_HIP_XSHARE'] == '1'
    self.xshare = xshare
    self.row_cells = row_cells
    # valid lanes of a strip as a multiple of this; None: as many as make a
    # strip's output rows start on 64-byte boundaries where they are written
    # with non-temporal stores, else 1
    if align_lanes is None and os.environ.get('SODA_HIP_ALIGN_LANES'):
      align_lanes = int(os.environ['SODA_HIP_ALIGN_LANES'])   # A/B runs
    self.align_lanes = align_lanes
    # trips of the unrolled loop whose warm-up is peeled into straight-line
    # code without the stages that do not matter yet: an int for every fusion
    # depth, a dict {depth: trips}, -1 = all of the warm-up, None = let
    # runtime.select_peel choose per kernel from the compiled register counts
    # (lower() itself treats None as -1)
    self.peel = peel
    # estimated VGPRs a marching shape may need before the ladder in lower()
    # moves on to a leaner one (None: REG_BUDGET)
    self.reg_budget = reg_budget
    # waves of a block sharing the fused iterations of a 2-D kernel; None =
    # default_pipe(fusion depth)
    self.pipe = pipe
    self.pipe_rows = pipe_rows
    self.buffer_ops = buffer_ops
    self.occupancy = occupancy
    self.min_waves = min_waves
    self.lane_shift = lane_shift
    self.interleave = interleave
    self.warm_guards = warm_guards
    self.strategy = strategy      # auto | direct | march | lds
    self.fuse = tuple(fuse)       # temporal-blocking depths to generate
    self.vec = vec
    self.chunk_rows = chunk_rows
    self.prefetch = prefetch
    self.waves_x = waves_x
    self.waves_y = waves_y
    # Cache hints, measured on MI355X with buffers ping-ponging as in a real
    # iterated run (tools/sweep.py --launches N): inputs are read once per
    # launch, so their loads are non-temporal; the outputs are the next
    # launch's inputs, so their stores are plain.
    self.nt_store = nt_store
    self.nt_load = nt_load
    self.xcd_swizzle = xcd_swizzle
    self.edge_loads = edge_loads
    self.tile_rows = tile_rows

  def resolved(self, dim: int, iterated: bool = True) -> 'LowerOptions':
    out = LowerOptions(self.strategy, self.fuse, self.vec, self.chunk_rows,
                       self.prefetch, self.waves_x, self.waves_y,
                       self.nt_store, self.nt_load, self.xcd_swizzle,
                       self.edge_loads, self.tile_rows, self.warm_guards,
                       self.interleave, self.lane_shift, self.min_waves,
                       self.occupancy, self.buffer_ops, self.pipe,
                       self.pipe_rows, self.reg_budget,
                       self.stamps, self.peel, self.align_lanes, self.xshare,
                       self.row_cells, self.windows, self.inline)
    if out.prefetch is None and dim == 3:
      out.prefetch = 1
    # 2-D: resolved per fusion depth in lower() (default_prefetch)
    # Cache hints.  Iterated programs (buffers ping-pong): inputs are read once
    # per launch -> non-temporal loads; outputs are the next launch's inputs ->
    # plain stores.  One-shot programs (iterate 1, e.g. blur): the output is
    # not re-read -> non-temporal stores, plain loads (blur 16384^2: 232 us vs
    # 294 us with the iterated defaults).  3-D tiles re-read halo rows -> plain
    # loads either way.
    if out.nt_load is None:
      out.nt_load = dim != 3 and iterated
    if out.nt_store is None:
      out.nt_store = not iterated
    if out.tile_rows is None:
      out.tile_rows = 4
    return out

  def key(self) -> str:
    return '%s_f%s_v%s_c%s_p%s_w%dx%d_%d%s%d%d_r%s' % (
        self.strategy, '-'.join(map(str, self.fuse)), self.vec,
        self.chunk_rows, self.prefetch, self.waves_x, self.waves_y,
        self.nt_store, self.nt_load, self.xcd_swizzle, self.edge_loads,
        self.tile_rows)


# operations per cell (as written) between which a one-iteration 2-D program
# of fp32 cells runs on narrow strips (prefers_narrow_strips)
NARROW_STRIP_OPS = (48, 128)


def prefers_narrow_strips(stencil: core.Stencil) -> bool:
  """One iteration per launch of a 2-D program that computes for 50-130
  operations per fp32 cell (denoise2d: 55): the kernel is bound by neither
  HBM nor the load queue but by how many waves can take turns, so 8 bytes per
  lane with 4 rows in flight (80 registers, six waves per SIMD) beat the usual
  16 bytes with 8 rows (127 registers, four waves): denoise2d 8192^2 171.6 /
  170.6 us -> 164.2 / 164.2 at 8 bytes per lane -> 157.9 with 4 rows in flight
  as well (4 bytes per lane 197.5; 16 bytes with 4 rows 168.0; 12 rows 176.6;
  profiles/r05_denoise_shapes.jsonl, one process; sustained over 900 launches
  each, alternating: 163.6 against 151.8 us, r05_denoise_sustained.json).
  Above the band a program goes through `ldswin` or takes 2 rows in flight
  (lower()); below it the loads are what a wave waits for.  fp32 cells only:
  that is what was measured.  runtime.resolve_options applies it where the
  caller fixed neither the cells per lane nor the prefetch depth."""
  if os.environ.get('SODA_HIP_NARROW', '1') == '0':
    return False
  if stencil.dim != 2 or stencil.iterate != 1:
    return False
  if any(str(t) != 'float' for t in stencil.symbol_table.values()):
    return False
  from soda_amd import ir as _ir
  work = sum(_ir.op_count(s.stmt.expr) +
             sum(_ir.op_count(l.expr) for l in s.stmt.let)
             for s in stencil.ordered_stages)
  return NARROW_STRIP_OPS[0] <= work < NARROW_STRIP_OPS[1]


def default_pipe(fused_iters: int) -> int:
  """Waves of a block that share the fused iterations of a 2-D kernel
  (MarchConfig.pipe).  One: in kernel sweeps four waves with three iterations
  each beat one wave with twelve by 5-10 % (jacobi2d T=12 on 8192 x {8192, 4296,
  2248, 1224}: 148 vs 156, 91 vs 102, 65 vs 70, 44 vs 47 us), but in the
  sustained 9-launch steps of bench.py on one box they do not (1.383 vs 1.357
  ms on 8192^2; 0.552 vs 0.603 ms on a 4-GPU slab, 0.419 vs 0.399 ms on an
  8-GPU slab).  `--hip-pipe 4` selects them."""
  return 1


def default_lane_shift(fused_iters: int, dim: int) -> str:
  """How a lane gets its neighbours' cells.  One iteration to a few per launch:
  DPP whole-wave shifts, folded into the consuming add.  Deep 2-D fusion
  (T >= 8): 'mixh' -- the two directions on two different pipes, down through
  DPP (vector ALU), up through ds_swizzle (the LDS crossbar), on 32-lane half
  strips.  A DPP-carrying instruction costs the issuing SIMD ~4-8 cycles, a
  swizzle occupies the CU's shared LDS pipe; all of either kind is the same
  speed (T=12: 148.3 / 148.5 us), half of each is faster although a half strip
  has fewer valid lanes, and needs 11 registers less, which buys two more rows
  in flight (profiles/r03_sweep_mixh_*.json, jacobi2d, us per launch, dpp ->
  mixh: T=12 on 8192^2 146.3 -> 136.2, on the 4296- / 1224-row slabs of a 2- /
  8-GPU run 89.4 -> 79.7 / 39.1 -> 34.8; T=8 114.3 -> 114.0, 31.9 -> 29.7;
  T=4 100.4 -> 101.1, 19.5 -> 20.0: DPP stays)."""
  return 'mixh' if dim == 2 and fused_iters >= 8 else 'dpp'


def default_prefetch(fused_iters: int, lane_shift: str = 'dpp') -> int:
  """Rows a 2-D marching wave keeps in flight ahead of the one it computes.
  With the loop body branch-free (buffer addressing) the depth is real:
  jacobi2d 8192^2, one iteration per launch, 99.9 / 94.0 / 92.6 / 90.9 us at
  depth 1 / 2 / 4 / 8; the fused kernels have less to hide and fewer registers
  to spare (T=8: 116 us at 4 vs 119 at 2; T=12 with DPP shifts: 149 at 2 vs
  153 at 4; with 'mixh' shifts, 11 registers leaner: 149.6 at 2, 136.2 at 4)."""
  if fused_iters <= 2:
    return 8
  if fused_iters <= 8 or lane_shift in ('mixh', 'mix64', 'mix64d'):
    return 4
  return 2


def lower(stencil: core.Stencil, opts: Optional[LowerOptions] = None) -> Module:
  """Builds the module: passes for every requested fusion depth plus a
  1-iteration pass (always present: the scheduler needs it for remainders)."""
  opts = (opts or LowerOptions()).resolved(
      stencil.dim, iterated=stencil.iterate > 1)
  _check_native(stencil)
  stencil.check_preserve()
  # arithmetic per cell of one iteration AS WRITTEN (the rewrites below fold
  # and split statements; what the shape ladder wants to know is how much the
  # program computes)
  from soda_amd import ir as _ir
  work = sum(_ir.op_count(s.stmt.expr) +
             sum(_ir.op_count(l.expr) for l in s.stmt.let)
             for s in stencil.ordered_stages)
  # the module's program from here on is a DERIVED one: same tensors the
  # caller sees (inputs, outputs, their windows and boxes), other locals
  def same_boxes(other: core.Stencil) -> bool:
    """A rewrite must leave the outputs' windows alone: they decide the valid
    box and, under `border: preserve`, which cells an iteration computes.  A
    tensor's window includes the tensor's own cell (core.iteration_boxes), so
    folding a local that is STORED off-centre can widen what its consumer may
    compute (`loc(0, -1) = in(1, 0); out(0, 0) = loc(0, -1)`: row 0 of `out`
    needs row -1 of `loc`; folded, it needs nothing outside the grid -- found
    by tools/fuzz_scan.py options, a preserved row computed instead)."""
    mine, theirs = stencil.iteration_boxes(), other.iteration_boxes()
    return all(mine[o] == theirs[o] for o in stencil.output_names)

  # wide 2-D windows through an LDS row ring (ldswin.py): a whole stage per
  # thread, so every pointwise local -- the groups of a rebalanced sum included
  # -- is a sub-expression there.  On request, and by itself where it pays
  # (contrast: one input, 17 x 17 taps, 393 operations per cell)
  if opts.strategy in ('auto', 'ldswin') and stencil.dim == 2 and \
      os.environ.get('SODA_HIP_LDSWIN', '1') != '0' and \
      (opts.strategy == 'ldswin' or ldswin_candidate(stencil)):
    from soda_amd.optimization import pointwise
    # (`auto`: a finite cap on the folded expression -- a local read k times
    # at one index is duplicated k times per level; contrast folds to ~400)
    whole = pointwise.inline_pointwise(
        stencil, fold_groups=True,
        max_ops=1 << 20 if opts.strategy == 'ldswin' else 1 << 13)
    if same_boxes(whole) and (
        (opts.strategy == 'ldswin' and ldswin_supported(whole) is None) or
        (opts.strategy == 'auto' and ldswin_pays(whole) and
         (opts.vec is None or opts.vec % 4 == 0))):
      mod = Module(whole)
      try:
        add_ldswin_pass(mod, chunk=opts.chunk_rows or 64,
                        step=opts.waves_y if opts.waves_y > 1 else None)
        return mod
      except util.SemanticError:
        # the ring does not fit LDS (window too tall / too wide): only an
        # explicit `ldswin` request hears about it, `auto` moves on to the
        # marching / direct ladder with the program as written
        if opts.strategy == 'ldswin':
          raise
  if opts.strategy == 'ldswin':
    raise util.SemanticError('ldswin: %s' % (
        ldswin_supported(stencil) or 'the program does not fold to one stage'))
  if opts.inline and opts.strategy != 'lds':
    from soda_amd.optimization import pointwise
    folded = pointwise.inline_pointwise(stencil)
    if same_boxes(folded):
      stencil = folded
  if opts.windows and opts.strategy != 'lds':
    from soda_amd.optimization import windows
    # (the marching kernels reduce dimension-0 windows themselves, all cells
    # of a lane jointly; `direct` kernels get chains in every dimension)
    marching = opts.strategy in ('auto', 'march') and \
        march_supported(stencil) is None
    skip = ()
    if marching:
      skip = ((0, None),) + (((stencil.dim - 1, '+'),) if SLIDING_SUMS else ())
    derived = windows.decompose(stencil, skip=skip)
    # (the auxiliaries are tensors of the launch plan: a program that would
    # exceed the argument block's slots keeps its windows as written)
    if len(derived.symbol_table) + len(derived.param_stmts) <= MAX_TENSORS \
        and same_boxes(derived):
      stencil = derived
  # operations whose operand ranges the text proves: cheaper sequences with
  # the same bits (exact.py; the program as written unless one is enabled)
  from soda_amd.codegen.hip import exact
  stencil = exact.specialize(stencil)
  mod = Module(stencil)
  if opts.strategy == 'lds':
    if stencil.preserve_border:
      raise util.SemanticError('lds2d: border: preserve is not supported')
    if stencil.symbol_table[stencil.input_names[0]].size_in_bytes != 4:
      raise util.SemanticError('lds2d: 4-byte cells only')
    add_lds2d_pass(mod, nt_load=bool(opts.nt_load))
    return mod
  use_march = opts.strategy in ('auto', 'march') and \
      march_supported(stencil) is None
  if opts.strategy == 'march' and not use_march:
    raise util.SemanticError('march: %s' % march_supported(stencil))
  if use_march:
    vec0 = opts.vec or default_vec(stencil)
    iterable = (len(stencil.input_names) == len(stencil.output_names) and
                stencil.input_types == stencil.output_types)
    # Fusion depths: each requested depth, clipped to the iteration count the
    # program asks for (a 2-iteration program gets a T=2 kernel, not T=4) and,
    # in 3-D, to MAX_FUSE_3D: a tile of rows x planes per tensor per fused
    # iteration; two iterations still fit (heat3d 512^3: 118-125 us per
    # iteration at T=2, 260 VGPRs, against 194 us at T=1; T=3 with 2 cells per
    # lane: 129 us).
    cap = stencil.iterate
    if stencil.dim == 3 and opts.strategy != 'march':
      cap = min(cap, MAX_FUSE_3D)
    if stencil.preserve_border and opts.strategy != 'march':
      # every fused iteration also keeps the window of the tensor it preserves
      # the border from: jacobi2d T=12 needs 186 VGPRs (2 waves per SIMD) and
      # runs 96 iterations in 2.51 ms, T=8 in 2.04 ms (1.54 / 1.61 ms without
      # preserve)
      cap = min(cap, MAX_FUSE_PRESERVE)
    depths = sorted({min(t, cap) for t in opts.fuse if iterable} - {0, 1},
                    reverse=True)

    def peel_for(t: int) -> int:
      if opts.peel is None:
        return -1
      if isinstance(opts.peel, dict):
        return int(opts.peel.get(t, -1))
      return int(opts.peel)

    def pipe_for(t: int) -> int:
      want = default_pipe(t) if opts.pipe is None else opts.pipe
      ok = (want > 1 and t > 1 and t % want == 0 and opts.buffer_ops and
            opts.waves_x * opts.waves_y == 1)
      return want if ok else 1

    out_bytes = min(t.size_in_bytes for t in stencil.output_types)

    def shift_for(t: int, xshare: int = 0) -> str:
      if opts.lane_shift is not None:
        return opts.lane_shift
      if xshare or pipe_for(t) > 1 or opts.waves_x * opts.waves_y != 1 or \
          not opts.buffer_ops:
        return 'dpp'
      return default_lane_shift(t, stencil.dim)

    def config(t: int, vec: int, pf: Optional[int],
               rows: Optional[int] = None, xshare: int = 0) -> MarchConfig:
      shift = shift_for(t, xshare)
      if pf is None:
        pf = default_prefetch(t, shift)
      cfg = MarchConfig(t, vec, opts.chunk_rows or 64, pf,
                        opts.waves_x, opts.waves_y, opts.nt_store,
                        opts.nt_load, opts.xcd_swizzle, opts.edge_loads,
                        rows or opts.tile_rows, opts.warm_guards,
                        opts.interleave,
                        shift, opts.min_waves, opts.occupancy,
                        opts.buffer_ops,
                        pipe_for(t), opts.pipe_rows,
                        opts.stamps, peel_for(t),
                        opts.align_lanes if opts.align_lanes is not None else
                        (max(1, 64 // (vec * out_bytes)) if opts.nt_store
                         else 1), xshare, bool(opts.windows),
                        bool(opts.windows) and SLIDING_SUMS)
      cfg.chunk_fixed = opts.chunk_rows is not None
      return cfg

    # The one-iteration kernel is mandatory.  Tall windows (erosion and xcorr
    # read 19 rows) or many live tensors may not fit the register budget at
    # the preferred shape: try shallower prefetch, then fewer cells per lane,
    # and take the first shape whose estimate stays
    # within the register budget (else the leanest); only if none exists fall
    # back to `direct` kernels.
    budget = opts.reg_budget or REG_BUDGET
    # arithmetic per cell of one iteration: a program that computes for
    # hundreds of instructions per cell has nothing to hide behind a deep
    # prefetch queue and pays for its registers in resident waves (contrast,
    # 393 operations per cell: 703 us at 8 rows in flight, 660 at 2)
    first = default_prefetch(1) if stencil.dim == 2 and work < 128 else \
        (2 if stencil.dim == 2 else 1)
    pfs = [opts.prefetch] if opts.prefetch else [first, 2, 1]
    pfs = sorted(set(pfs), reverse=True)
    # (3-D: fewer rows per tile never paid -- denoise3d 512^3: 651 / 691 / 910
    # us at 4 / 2 / 1 rows, 1 cell per lane -- so the tile height stays)
    rows_list = [opts.tile_rows]
    vecs = []
    v = vec0
    while v >= 1:
      vecs.append(v)
      v //= 2
    chosen = None
    last_error = None
    for v in vecs:
      for rows in rows_list:
        for pf in pfs:
          trial = Module(stencil)
          try:
            est = add_march_pass(trial, config(1, v, pf, rows)).traffic_model[
                'est_window_regs']
          except util.SemanticError as e:
            last_error = e
            continue
          if chosen is None or est < chosen[0]:
            chosen = (est, v, pf, rows)
          if est <= budget:
            break
        if chosen and chosen[0] <= budget:
          break
      if chosen and chosen[0] <= budget:
        break
    if chosen is None:
      if opts.strategy == 'march':
        raise last_error
      use_march = False
    else:
      _, vec, pf1, rows1 = chosen
      # waves that cover a whole row side by side, sharing x-halos through LDS
      share = 0
      if opts.row_cells and opts.xshare is not False and \
          (stencil.dim == 3 or opts.xshare) and opts.pipe in (None, 1):
        share = -(-opts.row_cells // (64 * vec))
        if share > (4 if opts.xshare is None else 16):
          share = 0
      for t in depths:
        keep = (len(mod.kernels), len(mod.passes), len(mod.chunks))
        for xs in ([share, 0] if share and t > 1 else [0]):
          try:
            add_march_pass(mod, config(t, vec, opts.prefetch or None,
                                       xshare=xs))
            break
          except util.SemanticError:
            # this shape does not fit: try without sharing; if the depth does
            # not fit at all the scheduler uses the others
            del mod.kernels[keep[0]:], mod.passes[keep[1]:], \
                mod.chunks[keep[2]:]
      # (the one-iteration kernel shares x-halos only on request: measured)
      done1 = False
      if share and opts.xshare:
        keep = (len(mod.kernels), len(mod.passes), len(mod.chunks))
        try:
          add_march_pass(mod, config(1, vec, pf1, rows1, xshare=share))
          done1 = True
        except util.SemanticError:
          del mod.kernels[keep[0]:], mod.passes[keep[1]:], mod.chunks[keep[2]:]
      if not done1:
        add_march_pass(mod, config(1, vec, pf1, rows1))
  if not use_march:
    add_direct_pass(mod, opts.vec or 1)
  return mod

