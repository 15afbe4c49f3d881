"""Stencil: the middle-end object every backend receives.

Restates, for execution on a GPU, the parts of reference src/soda/core.py that
define *what* a SODA program computes:

  * validation and messages of `Stencil.__init__` (core.py:52-142);
  * the per-iteration tensor chain and its names (core.py:307-369,
    `name_in_iter` :320-336; order pinned by src/tests/test_core.py:84-88);
  * stencil windows, their offset/dim/distance (core.py:858-926) and from them
    the valid box every tensor is defined on (frt/host.py:565-577).

The FPGA-only parts (reuse buffers, ILP-scheduled FIFO depths, dataflow
modules) have no counterpart: a GPU keeps halos in LDS/registers.  Windows of
`iterate` = 100..1000 are computed analytically as per-dimension bounds (the
reference enumerates point sets, O((iterate*r)^dim)); point-set enumeration is
kept for small cases because `stencil_distance` is defined on the point set.
"""
import collections
import itertools
import logging
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

from soda_amd import grammar, ir, util

_logger = logging.getLogger(__name__)

Box = Tuple[Tuple[int, ...], Tuple[int, ...]]  # (lo offsets, hi offsets)


class Stage:
  """One local/output statement seen as a producer: taps grouped by parent.

  Offsets are relative to the produced cell (load idx - store idx), which is
  how the reference's loop nest addresses parents (frt/host.py:587-594)."""

  def __init__(self, stmt, is_output: bool, param_names: Sequence[str] = ()):
    self.stmt = stmt
    self.name = stmt.name
    self.haoda_type = stmt.haoda_type
    self.is_output = is_output
    self.st_idx = stmt.ref.idx
    nodes = [l.expr for l in stmt.let] + [stmt.expr]
    self.loads = ir.get_load_dict(nodes)
    # `param` arrays are not tensors of the grid: their elements are addressed
    # absolutely, they take no part in windows, boxes or the stage DAG
    self.params = [p for p in self.loads if p in param_names]
    for node in nodes:
      for v in ir.get_vars(node):
        if v.name in param_names and v.name not in self.params:
          self.params.append(v.name)
    self.taps: Dict[str, List[Tuple[int, ...]]] = collections.OrderedDict()
    for parent, refs in self.loads.items():
      if parent in param_names:
        continue
      seen = []
      for r in refs:
        off = tuple(a - b for a, b in zip(r.idx, self.st_idx))
        if off not in seen:
          seen.append(off)
      self.taps[parent] = seen

  def tap_bounds(self, parent: str) -> Box:
    pts = self.taps[parent]
    dim = len(self.st_idx)
    return (tuple(min(p[d] for p in pts) for d in range(dim)),
            tuple(max(p[d] for p in pts) for d in range(dim)))


class Stencil:
  """A validated SODA program plus the geometry a backend needs.

  Constructor keywords are the reference's (core.py:52-76), so the unit tests
  of the reference read the same here."""

  def __init__(self, **kwargs):
    self.iterate = kwargs.pop('iterate')
    if self.iterate < 1:
      raise util.SemanticError('cannot iterate %d times' % self.iterate)
    self.border = kwargs.pop('border', None) or 'ignore'
    self.preserve_border = self.border == 'preserve'
    self.cluster = kwargs.pop('cluster', None) or 'none'
    self.burst_width = kwargs.pop('burst_width')
    self.app_name = kwargs.pop('app_name')
    self.tile_size = tuple(kwargs.pop('tile_size'))
    self.unroll_factor = kwargs.pop('unroll_factor')
    self.replication_factor = kwargs.pop('replication_factor', 1)
    self.dim = kwargs.pop('dim')
    self.param_stmts = list(kwargs.pop('param_stmts'))
    self.input_stmts = list(kwargs.pop('input_stmts'))
    self.local_stmts = list(kwargs.pop('local_stmts'))
    self.output_stmts = list(kwargs.pop('output_stmts'))
    self.optimizations = kwargs.pop('optimizations', {}) or {}
    for key in ('dram_in', 'dram_out'):  # FPGA bank maps: accepted, unused
      kwargs.pop(key, None)

    if self.dim > util.MAX_DIM:
      raise util.SemanticError('at most %d dimensions are supported, got %d' %
                               (util.MAX_DIM, self.dim))

    if self.iterate > 1:
      if len(self.input_stmts) != len(self.output_stmts):
        raise util.SemanticError(
            'number of input tensors must be the same as output if iterate > 1 '
            'times, currently there are %d input(s) but %d output(s)' %
            (len(self.input_stmts), len(self.output_stmts)))
      if self.input_types != self.output_types:
        raise util.SemanticError(
            'input must have the same type(s) as output if iterate > 1 '
            'times, current input has type %s but output has type %s' %
            (util.lst2str(self.input_types), util.lst2str(self.output_types)))

    table = self.symbol_table  # raises on duplicate names
    for stmt in self.local_stmts + self.output_stmts:
      stmt.expr = ir.flatten(stmt.expr)
      stmt.let = tuple(ir.flatten(l) for l in stmt.let)
      if len(stmt.ref.idx) != self.dim:
        raise util.SemanticError(
            '`%s` is stored with %d indices in a %d-dimensional program' %
            (stmt.name, len(stmt.ref.idx), self.dim))
    known = set(table) | set(self.param_names)
    for stmt in self.local_stmts + self.output_stmts:
      lets = set()
      for node in [l for l in stmt.let] + [stmt.expr]:
        expr = node.expr if isinstance(node, ir.Let) else node
        for ref in ir.get_loads(expr):
          if ref.name not in known:
            raise util.SemanticError('`%s` loads unknown tensor `%s`' %
                                     (stmt.name, ref.name))
          if ref.name not in self.param_names and len(ref.idx) != self.dim:
            raise util.SemanticError(
                '`%s` is loaded with %d indices in a %d-dimensional program' %
                (ref.name, len(ref.idx), self.dim))
          if ref.name in self.param_names:
            size = self.param_table[ref.name].size
            if len(ref.idx) != len(size) or not all(
                0 <= i < n for i, n in zip(ref.idx, size)):
              raise util.SemanticError(
                  '`%s` reads element (%s) of param `%s%s`' %
                  (stmt.name, ', '.join(map(str, ref.idx)), ref.name,
                   ''.join('[%d]' % n for n in size)))
        for var in ir.get_vars(expr):
          if var.name not in lets and var.name not in self.param_names:
            raise util.SemanticError('`%s` uses unknown variable `%s`' %
                                     (stmt.name, var.name))
          if var.name in self.param_names and var.name not in lets and \
              self.param_table[var.name].size:
            raise util.SemanticError(
                '`%s` uses param array `%s` without an index' %
                (stmt.name, var.name))
        if isinstance(node, ir.Let):
          lets.add(node.name)
    # the one optimisation pass the reference always runs, at this very point
    # (ref core.py:134-138: after simplify, before type propagation)
    self._cr_counter = 0
    from soda_amd.optimization import inline
    inline.rebalance(self)
    table = self.symbol_table
    typed = dict(table)
    typed.update((p.name, p.haoda_type) for p in self.param_stmts)
    for stmt in self.local_stmts + self.output_stmts:
      stmt.propagate_type(typed)

    self.stages: List[Stage] = (
        [Stage(s, False, self.param_names) for s in self.local_stmts] +
        [Stage(s, True, self.param_names) for s in self.output_stmts])
    self._check_dag()

  def new_cr_var(self) -> str:
    """A fresh statement name `cr_var_<n>` (ref core.py:183-191)."""
    while True:
      var = 'cr_var_%d' % self._cr_counter
      self._cr_counter += 1
      if var not in {s.name for s in
                     self.input_stmts + self.local_stmts + self.output_stmts}:
        return var

  # -- names and types -----------------------------------------------------
  @property
  def kernel_name(self) -> str:
    return '%s_kernel' % self.app_name

  @property
  def input_names(self):
    return tuple(s.name for s in self.input_stmts)

  @property
  def param_names(self):
    return tuple(s.name for s in self.param_stmts)

  @property
  def param_table(self):
    return {s.name: s for s in self.param_stmts}

  @staticmethod
  def param_index(stmt, idx: Sequence[int]) -> int:
    """Element `name(i, j, ...)` of `param T: name[s0][s1]...` is the C array
    element name[i][j]... (row-major: the reference host declares and fills
    it that way, frt/host.py:497-541); a scalar param has the single index 0."""
    lin = 0
    for i, n in zip(idx, stmt.size):
      lin = lin * n + i
    return lin

  @staticmethod
  def param_elems(stmt) -> int:
    n = 1
    for v in stmt.size:
      n *= v
    return n

  @property
  def local_names(self):
    return tuple(s.name for s in self.local_stmts)

  @property
  def output_names(self):
    return tuple(s.name for s in self.output_stmts)

  @property
  def input_types(self):
    return tuple(s.haoda_type for s in self.input_stmts)

  @property
  def local_types(self):
    return tuple(s.haoda_type for s in self.local_stmts)

  @property
  def output_types(self):
    return tuple(s.haoda_type for s in self.output_stmts)

  @property
  def symbol_table(self) -> Dict[str, ir.Type]:
    table: Dict[str, ir.Type] = collections.OrderedDict()
    for stmt in self.input_stmts + self.local_stmts + self.output_stmts:
      if stmt.name in table:
        raise util.InputError('conflicting stmt name: %s' % stmt.name)
      table[stmt.name] = stmt.haoda_type
    return table

  def __str__(self) -> str:
    stmts = (self.input_stmts + self.param_stmts + self.local_stmts +
             self.output_stmts)
    return ('kernel: {0.app_name}\nburst width: {0.burst_width}\n'
            'iterate: {0.iterate}\nunroll factor: {0.unroll_factor}\n{stmts}\n'
            'border: {0.border}\ncluster: {0.cluster}').format(
                self, stmts='\n'.join(map(str, stmts)))

  # -- DAG -----------------------------------------------------------------
  def _check_dag(self) -> None:
    """Orders stages so that parents come first (the reference toposorts,
    core.py:458-471); a cycle inside one iteration is an error."""
    produced = set(self.input_names)
    pending = list(self.stages)
    ordered: List[Stage] = []
    while pending:
      ready = [s for s in pending if all(p in produced or
                                         p in self.param_names
                                         for p in s.taps)]
      if not ready:
        raise util.SemanticError(
            'cyclic dependence among %s' % util.lst2str(s.name for s in pending))
      # stable: keep file order among ready stages
      nxt = ready[0]
      ordered.append(nxt)
      produced.add(nxt.name)
      pending.remove(nxt)
    self.ordered_stages = ordered

  def name_in_iter(self, name: str, iteration: int) -> str:
    """Name of `name` as seen by statements of iteration `iteration`
    (reference core.py:320-336)."""
    if name in self.input_names:
      return name if iteration == 0 else '%s_iter%d' % (name, iteration)
    if name in self.output_names:
      if iteration < self.iterate - 1:
        return '%s_iter%d' % (
            self.input_names[self.output_names.index(name)], iteration + 1)
      return name
    if name in self.local_names:
      return name if iteration == 0 else '%s_iter%d' % (name, iteration)
    if name in self.param_names:
      return name
    raise util.InternalError('unknown name: %s' % name)

  def produce_offsets(self) -> Dict[str, int]:
    """Stream position at which every INPUT tensor is produced relative to
    the earliest one -- the delay the reference host applies when it lays the
    inputs out (`produce_offset`, ref frt/host.py:241-246).

    Restates the integer program of ref core.py:371-426 over the tensors of
    all iterations: variables p_T (produced) and c_T (kept until), minimise
    sum(c_T - p_T) subject to c_T >= p_T and, for every load of L by a
    statement S stored at linear offset s with linear load offsets o:
        p_L <= p_S + (s - max o)      (the newest element exists)
        c_L >= p_S + (s - min o)      (the oldest one has not been dropped)
    with p of input 0 pinned at 0; the result is shifted so the earliest
    input sits at 0 (core.py:416-421).  Linear offsets are `serialize`d with
    the program's tile size (ref tensor.py:71-92).  Solved with SciPy's HiGHS
    (the reference uses PuLP/CBC: where the optimum is not unique the two may
    pick different offsets -- one-input programs have nothing to solve)."""
    if len(self.input_names) == 1:
      return {self.input_names[0]: 0}
    import numpy as np
    from scipy import optimize
    tile = self.tile_size
    names = list(self.tensor_names)
    index = {n: i for i, n in enumerate(names)}
    nt = len(names)
    # variables: p_0..p_{nt-1}, c_0..c_{nt-1}
    rows, lo, hi = [], [], []

    def constraint(coeffs, lower, upper):
      row = np.zeros(2 * nt)
      for j, v in coeffs:
        row[j] += v
      rows.append(row)
      lo.append(lower)
      hi.append(upper)

    for i in range(nt):
      constraint([(nt + i, 1.0), (i, -1.0)], 0.0, np.inf)       # c >= p
    for it in range(self.iterate):
      for stage in self.ordered_stages:
        s_name = self.name_in_iter(stage.name, it)
        s_off = util.serialize(stage.st_idx, tile)
        for parent, refs in stage.loads.items():
          if parent in self.param_names:
            continue
          l_name = self.name_in_iter(parent, it)
          offs = [util.serialize(r.idx, tile) for r in refs]
          ps, pl, cl = index[s_name], index[l_name], nt + index[l_name]
          # p_L - p_S <= s - newest
          constraint([(pl, 1.0), (ps, -1.0)], -np.inf, s_off - max(offs))
          # c_L - p_S >= s - oldest
          constraint([(cl, 1.0), (ps, -1.0)], s_off - min(offs), np.inf)
    cost = np.concatenate([-np.ones(nt), np.ones(nt)])
    bounds_lo = np.full(2 * nt, -np.inf)
    bounds_hi = np.full(2 * nt, np.inf)
    bounds_lo[index[self.input_names[0]]] = 0.0
    bounds_hi[index[self.input_names[0]]] = 0.0
    res = optimize.milp(
        cost, constraints=optimize.LinearConstraint(np.array(rows), lo, hi),
        integrality=np.ones(2 * nt),
        bounds=optimize.Bounds(bounds_lo, bounds_hi))
    if not res.success:
      raise util.InternalError('unexpected ILP status: %s' % res.message)
    p = {n: int(round(res.x[index[n]])) for n in self.input_names}
    base = min(p.values())
    return {n: v - base for n, v in p.items()}

  @property
  def tensor_names(self) -> Tuple[str, ...]:
    """Every tensor of the unrolled program in chronological order."""
    names = list(self.input_names)
    for it in range(self.iterate):
      for stage in self.ordered_stages:
        names.append(self.name_in_iter(stage.name, it))
    return tuple(names)

  # the reference exposes dicts of Tensor objects; only the names matter here
  @property
  def tensors(self) -> Dict[str, str]:
    return collections.OrderedDict((n, n) for n in self.tensor_names)

  @property
  def chronological_tensors(self):
    Named = collections.namedtuple('Named', 'name')
    return [Named(n) for n in self.tensor_names]

  # -- windows: analytic bounds --------------------------------------------
  def iteration_boxes(self, in_boxes: Optional[Dict[str, Box]] = None
                      ) -> Dict[str, Box]:
    """Bounds, relative to the program inputs, of the window every tensor of
    ONE iteration depends on, given the bounds of this iteration's inputs."""
    zero = (0,) * self.dim
    boxes: Dict[str, Box] = {}
    for name in self.input_names:
      boxes[name] = (in_boxes or {}).get(name, (zero, zero))
    for stage in self.ordered_stages:
      lo = [None] * self.dim
      hi = [None] * self.dim
      for parent in stage.taps:
        if parent in self.param_names:
          continue
        tlo, thi = stage.tap_bounds(parent)
        plo, phi = boxes[parent]
        for d in range(self.dim):
          # min(plo, 0) / max(phi, 0): the loaded element itself must lie in
          # the grid too.  Identical to the reference's window whenever every
          # tensor's window spans offset 0 (true for its whole corpus); where
          # it does not (`p(0) = in(2)`, `t(0) = p(-3)`) the reference's
          # self-check loop would read p[-2] out of bounds.
          l = tlo[d] + min(plo[d], 0)
          h = thi[d] + max(phi[d], 0)
          lo[d] = l if lo[d] is None else min(lo[d], l)
          hi[d] = h if hi[d] is None else max(hi[d], h)
      if lo[0] is None:  # a stage of constants only
        lo, hi = list(zero), list(zero)
      boxes[stage.name] = (tuple(lo), tuple(hi))
    return boxes

  def window_bounds(self, iterate: Optional[int] = None) -> Dict[str, Box]:
    """Bounds of the overall stencil window of every tensor of the LAST of
    `iterate` iterations (default: self.iterate), relative to the program
    inputs.  Equals the bounding box of reference
    `get_overall_stencil_window(inputs, tensor)` (core.py:876-919)."""
    iterate = self.iterate if iterate is None else iterate
    boxes = self.iteration_boxes()
    for _ in range(iterate - 1):
      nxt = {i: boxes[o] for i, o in zip(self.input_names, self.output_names)}
      boxes = self.iteration_boxes(nxt)
    return boxes

  def valid_box(self, extent: Sequence[int], name: Optional[str] = None,
                iterate: Optional[int] = None
                ) -> Tuple[Tuple[int, ...], Tuple[int, ...]]:
    """[lo, hi) per dimension of the region where tensor `name` (default: the
    first output) is defined after `iterate` iterations on a grid of `extent`
    (loop bounds of reference frt/host.py:570-577)."""
    name = name or self.output_names[0]
    if self.preserve_border:
      # outputs are defined everywhere (their border cells carry the input
      # they replace); locals only where one iteration can compute them
      if name in self.output_names:
        return (0,) * self.dim, tuple(extent)
      return self.interior_box(extent, name)
    lo, hi = self.window_bounds(iterate)[name]
    return (tuple(max(0, -l) for l in lo),
            tuple(n - max(0, h) for n, h in zip(extent, hi)))

  # -- border: preserve -----------------------------------------------------
  # The reference parses and stores `border: preserve` (grammar.py:32,
  # core.py:56-57) but no live code path implements it.  Here it means: in
  # every iteration, an output cell outside the box ONE iteration can compute
  # takes the value of the input that output replaces (outputs pair with
  # inputs by position, as `iterate` pairs them), so boundary values persist
  # and the defined region does not shrink with the iteration count.
  def check_preserve(self) -> None:
    """Raises unless `border: preserve` is well defined for this program."""
    if not self.preserve_border:
      return
    if (len(self.input_stmts) != len(self.output_stmts) or
        self.input_types != self.output_types):
      raise util.SemanticError(
          'border: preserve copies every output\'s border from the input it '
          'replaces: it needs as many outputs as inputs, of the same types '
          '(inputs %s, outputs %s)' % (util.lst2str(self.input_types),
                                      util.lst2str(self.output_types)))

  def preserved_from(self, output: str) -> str:
    """The input whose values `output` keeps on its border."""
    return self.input_names[self.output_names.index(output)]

  def interior_bounds(self, name: str) -> Box:
    """Window of `name` over ONE iteration, relative to that iteration's
    inputs (what decides where a cell can be computed under preserve)."""
    return self.iteration_boxes()[name]

  def interior_box(self, extent: Sequence[int], name: str
                   ) -> Tuple[Tuple[int, ...], Tuple[int, ...]]:
    lo, hi = self.interior_bounds(name)
    return (tuple(max(0, -l) for l in lo),
            tuple(n - max(0, h) for n, h in zip(extent, hi)))

  @property
  def radius(self) -> Box:
    """Per-iteration growth of the window of the outputs: (lo, hi) with
    lo <= 0 <= hi taken over all outputs (halo depth per iteration)."""
    boxes = self.iteration_boxes()
    lo = tuple(min(0, min(boxes[o][0][d] for o in self.output_names))
               for d in range(self.dim))
    hi = tuple(max(0, max(boxes[o][1][d] for o in self.output_names))
               for d in range(self.dim))
    return lo, hi

  def reach_along(self, d: int) -> Tuple[int, int]:
    """Cells one iteration reads below / above a cell along dimension d."""
    lo, hi = self.radius
    return -lo[d], hi[d]

  # -- windows: point sets (small cases; pins kStencilDistance etc.) -------
  def stencil_window_points(self, name: Optional[str] = None,
                            iterate: Optional[int] = None,
                            limit: int = 2_000_000
                            ) -> Tuple[Tuple[int, ...], ...]:
    """Sorted point set of the overall stencil window (core.py:876-919)."""
    name = name or self.output_names[0]
    iterate = self.iterate if iterate is None else iterate
    zero = (0,) * self.dim
    cur = {n: {zero} for n in self.input_names}
    for it in range(iterate):
      sets: Dict[str, set] = dict(cur)
      for stage in self.ordered_stages:
        pts = set()
        for parent, offs in stage.taps.items():
          if parent in self.param_names:
            continue
          for off in offs:
            for p in sets[parent]:
              pts.add(tuple(a + b for a, b in zip(p, off)))
          if len(pts) > limit:
            raise util.InputError(
                'stencil window of %s has more than %d points; use '
                'window_bounds()' % (name, limit))
        sets[stage.name] = pts
      if it < iterate - 1:
        cur = {i: sets[o] for i, o in zip(self.input_names, self.output_names)}
    return tuple(sorted(sets[name]))

  @property
  def stencil_window(self):
    return self.stencil_window_points()

  @property
  def stencil_distance(self) -> int:
    """kStencilDistance as the reference's Stencil holds it (core.py:616-625):
    the larger of the window's distance and its stencil offset.  They differ
    when every tap lies AHEAD of the cell in streaming order (`o(0, 0) = i(-1,
    2)` on 32-cell rows: distance 0, offset 63 -- the host's void tail and
    cycle count must cover the 63).  Rounds 1-5 returned the bare distance
    here; a second derivation of the host's constants in the test
    infrastructure (tests/test_stream.py) found it."""
    window = self.stencil_window
    distance = get_stencil_distance(window, self.tile_size)
    offset = distance - util.serialize(get_stencil_window_offset(window),
                                       self.tile_size)
    return max(distance, offset)

  @property
  def stencil_dim(self) -> List[int]:
    return get_stencil_dim(self.stencil_window)


def get_stencil_window_offset(points: Iterable[Sequence[int]]) -> Tuple[int, ...]:
  """-min per dimension of a store-normalised window (core.py:922-926)."""
  points = list(points)
  return tuple(-min(p[d] for p in points) for d in range(len(points[0])))


def get_stencil_dim(points: Iterable[Sequence[int]]) -> List[int]:
  """max-min+1 per dimension (core.py:864-870)."""
  points = list(points)
  return [
      max(p[d] for p in points) - min(p[d] for p in points) + 1
      for d in range(len(points[0]))
  ]


def get_stencil_distance(points: Iterable[Sequence[int]],
                         tile_size: Sequence[int]) -> int:
  """Elements between the first input a cell needs and the cell itself in the
  reference's streaming order (core.py:858-861)."""
  points = list(points)
  return (max(util.serialize_iter(points, tile_size)) +
          util.serialize(get_stencil_window_offset(points), tile_size))


def from_program(program: grammar.SodaProgram, **overrides) -> Stencil:
  """What reference sodac.py:150-194 does between parse and backends:
  directive values overridden by command-line values."""
  tile_size = list(program.tile_size[:-1])
  over_tile = overrides.pop('tile_size', None) or ()
  for d, t in enumerate(over_tile):
    if d < len(tile_size) and t and t > 0:
      tile_size[d] = t
  tile_size.append(0)
  replication_factor = overrides.pop('replication_factor', None)
  unroll_factor = overrides.pop('unroll_factor', None)
  if replication_factor is None:
    unroll = unroll_factor if unroll_factor is not None else program.unroll_factor
    replication = 1
  else:
    unroll = replication = replication_factor

  def pick(key):
    value = overrides.pop(key, None)
    return value if value is not None else getattr(program, key)

  stencil = Stencil(
      burst_width=pick('burst_width'),
      border=pick('border'),
      iterate=pick('iterate'),
      cluster=pick('cluster'),
      dram_in=overrides.pop('dram_in', None),
      dram_out=overrides.pop('dram_out', None),
      app_name=program.app_name,
      input_stmts=program.input_stmts,
      param_stmts=program.param_stmts,
      local_stmts=program.local_stmts,
      output_stmts=program.output_stmts,
      dim=program.dim,
      tile_size=tile_size,
      unroll_factor=unroll,
      replication_factor=replication,
      optimizations=overrides.pop('optimizations', None),
  )
  if overrides:
    raise util.InternalError('unknown overrides: %s' % sorted(overrides))
  return stencil


def from_text(text: str, **overrides) -> Stencil:
  return from_program(grammar.parse(text), **overrides)


def from_file(path: str, **overrides) -> Stencil:
  return from_program(grammar.parse_file(path), **overrides)
