"""`decompose`: long integer window reductions as chains of power-of-two windows.

A kernel-quality pass of the HIP backend, NOT one of the reference's (its
computation-reuse pass, reference src/soda/optimization/computation_reuse.py,
reorders floating-point sums and is off by default; this one touches only
reductions whose result does not depend on association).

A stage such as erosion's (reference tests/src/erosion.soda:5-15)

    tmp(0, 9) = min(input(0, 0), input(0, 1), ..., input(0, 18))

costs 18 operations per cell as written.  A window of n = 19 taps is the
disjoint union of windows of 16, 2 and 1 taps, and a window of 2k taps is two
windows of k taps side by side, so with the auxiliary tensors

    w2(x) = min(input(x), input(x + 1))        w4(x) = min(w2(x), w2(x + 2))
    w8(x) = min(w4(x), w4(x + 4))              w16(x) = min(w8(x), w8(x + 8))

and the rest of the window reduced directly, r3(x) = min(input(x), input(x + 1),
input(x + 2)), the stage becomes `min(w16(0), r3(16))`: 4 + 2 + 1 = 7 operations
per cell.  (w2(16) and input(18) would do for the rest with one operation
less, but then two tensors are tapped 16 and 18 cells away: along the streamed
dimension they would have to be kept in registers that long, along dimension 0
fetched from lanes away.)  Exact -- bit for bit -- where the reduction is associative and
commutative on the values it sees:

  * `min` / `max` of integers (any integer type; the auxiliaries keep it);
  * `+` of integers narrower than 32 bits: C promotes them to int, a sum of
    fewer than 2^16 such values cannot overflow int32, so the auxiliaries are
    int32 tensors holding the exact partial sums and the statement's own cast
    (xcorr: int16, reference tests/src/xcorr.soda:5-15) is applied once, to the
    same integer, as before.

Floating-point sums are never touched (their association is the reference's,
DESIGN.md section 3), nor are floating-point min / max (signed zeros, NaNs).
The auxiliaries are ordinary `local` statements of a derived program: the
marching kernels keep them in registers like any other local; windows, boxes
and results of the original tensors do not change.  The oracle never sees the
derived program.
"""
from typing import Dict, List, Optional, Tuple

from soda_amd import core, ir

MIN_RUN = 6          # shorter windows are left alone
MAX_DIRECT = 4       # a rest of at most this many taps is reduced directly
_MINMAX = ('min', 'max')


def _int_bits(t: ir.Type) -> Optional[int]:
  """Width of a native integer type, None for anything else."""
  if t is None or t.is_float or not t.is_native:
    return None
  name = str(t)
  if not (name.startswith('int') or name.startswith('uint')):
    return None
  return t.width_in_bits


def _run(refs: List[ir.Ref]) -> Optional[Tuple[int, int, int]]:
  """(dimension, first offset, length) if the taps differ along exactly one
  dimension and form a contiguous run there without repeats."""
  if len(refs) < MIN_RUN or len({r.idx for r in refs}) != len(refs):
    return None
  dim = len(refs[0].idx)
  varying = [d for d in range(dim) if len({r.idx[d] for r in refs}) > 1]
  if len(varying) != 1:
    return None
  d = varying[0]
  offs = sorted(r.idx[d] for r in refs)
  if offs != list(range(offs[0], offs[0] + len(offs))):
    return None
  return d, offs[0], len(offs)


def _match(stmt, table) -> Optional[Tuple[str, str, int, int, int, Tuple[int, ...]]]:
  """(op, parent, dimension, first offset, length, index of the first tap)."""
  if stmt.let:
    return None
  expr = stmt.expr
  while isinstance(expr, ir.Cast) and expr.haoda_type == stmt.haoda_type:
    expr = expr.expr
  if isinstance(expr, ir.Chain) and set(expr.operators) == {'+'}:
    op, terms = '+', list(expr.operands)
  elif isinstance(expr, ir.Call) and expr.name in _MINMAX:
    op, terms = expr.name, list(expr.args)
  else:
    return None
  if not all(isinstance(t, ir.Ref) for t in terms):
    return None
  parent = terms[0].name
  if parent not in table or any(t.name != parent for t in terms):
    return None
  bits = _int_bits(table[parent])
  if bits is None or _int_bits(stmt.haoda_type) is None:
    return None
  if op == '+' and bits >= 32:
    return None
  run = _run(terms)
  if run is None:
    return None
  d, first, n = run
  if op == '+' and n >= (1 << 15):
    return None
  base = list(terms[0].idx)
  base[d] = first
  return op, parent, d, first, n, tuple(base)


def _ref(name: str, idx) -> str:
  return '%s(%s)' % (name, ', '.join(str(i) for i in idx))


def decompose(stencil: core.Stencil, skip=()) -> core.Stencil:
  """The derived program (a new Stencil), or `stencil` itself if no statement
  qualifies.  `skip`: (dimension, op) pairs left as written, op None = any --
  the marching kernels reduce dimension-0 windows for all cells of a lane
  jointly (march._emit_xwindow: cheaper than chains there, where every far tap
  is a lane move) and keep sums along the streamed dimension as sliding sums
  (two operations per cell, no auxiliary tensors)."""
  table = stencil.symbol_table
  taken = set(table) | set(stencil.param_names)
  made: Dict[Tuple[str, str, int, int], str] = {}   # (op, parent, dim, size)
  new_locals: Dict[str, List[str]] = {}             # consumer -> DSL lines
  rewritten: Dict[str, str] = {}
  dim = stencil.dim

  def aux(op: str, parent: str, d: int, size: int, lines: List[str]) -> str:
    """Name of the tensor holding the `size`-tap window of `parent` along `d`
    that starts at a cell; `lines` gets the statements still to be made."""
    if size == 1:
      return parent
    key = (op, parent, d, size)
    if key in made:
      return made[key]
    half = aux(op, parent, d, size // 2, lines)
    tag = {'+': 'sum', 'min': 'min', 'max': 'max'}[op]
    name = '%s_%s%d_%d' % (parent, tag, d, size)
    while name in taken:
      name += '_'
    taken.add(name)
    made[key] = name
    zero = [0] * dim
    far = list(zero)
    far[d] = size // 2
    ctype = 'int32' if op == '+' else str(table[parent])
    a, b = _ref(half, zero), _ref(half, far)
    body = '%s + %s' % (a, b) if op == '+' else '%s(%s, %s)' % (op, a, b)
    lines.append('local %s: %s = %s' % (ctype, _ref(name, zero), body))
    return name

  def direct(op: str, parent: str, d: int, size: int, lines: List[str]) -> str:
    """The `size`-tap window of `parent` written out (size - 1 operations)."""
    if size == 1:
      return parent
    key = (op + '/direct', parent, d, size)
    if key in made:
      return made[key]
    tag = {'+': 'sum', 'min': 'min', 'max': 'max'}[op]
    name = '%s_%s%d_r%d' % (parent, tag, d, size)
    while name in taken:
      name += '_'
    taken.add(name)
    made[key] = name
    taps = []
    for i in range(size):
      idx = [0] * dim
      idx[d] = i
      taps.append(_ref(parent, idx))
    ctype = 'int32' if op == '+' else str(table[parent])
    body = ' + '.join(taps) if op == '+' else '%s(%s)' % (op, ', '.join(taps))
    lines.append('local %s: %s = %s' % (ctype, _ref(name, [0] * dim), body))
    return name

  for stmt in stencil.local_stmts + stencil.output_stmts:
    m = _match(stmt, table)
    if m is None or (m[2], None) in skip or (m[2], m[0]) in skip:
      continue
    op, parent, d, first, n, base = m
    lines: List[str] = []
    terms = []
    at = list(base)
    left = n
    while left:
      size = 1 << (left.bit_length() - 1)
      if left < n and left <= MAX_DIRECT:
        # the short rest of the window directly from the parent: one tensor
        # with near taps only, instead of several auxiliaries tapped far away
        # (each of which would have to be kept -- registers -- or fetched from
        # lanes away -- shifts -- until the far end of the window arrives)
        terms.append(_ref(direct(op, parent, d, left, lines), at))
        break
      terms.append(_ref(aux(op, parent, d, size, lines), at))
      at[d] += size
      left -= size
    body = ' + '.join(terms) if op == '+' else '%s(%s)' % (op, ', '.join(terms))
    new_locals[stmt.name] = lines
    head = str(stmt).split('=', 1)[0]
    rewritten[stmt.name] = '%s= %s' % (head, body)
  if not rewritten:
    return stencil
  out = ['kernel: %s' % stencil.app_name,
         'burst width: %d' % stencil.burst_width,
         'iterate: %d' % stencil.iterate,
         'unroll factor: %d' % stencil.unroll_factor]
  out.extend(str(s) for s in stencil.input_stmts + stencil.param_stmts)
  for stmt in stencil.local_stmts:
    out.extend(new_locals.get(stmt.name, ()))
    out.append(rewritten.get(stmt.name, str(stmt)))
  # (locals first: auxiliaries an output needs come after every local)
  for stmt in stencil.output_stmts:
    out.extend(new_locals.get(stmt.name, ()))
  for stmt in stencil.output_stmts:
    out.append(rewritten.get(stmt.name, str(stmt)))
  out.append('border: %s' % stencil.border)
  out.append('cluster: %s' % stencil.cluster)
  derived = core.from_text('\n'.join(out) + '\n')
  derived.replication_factor = stencil.replication_factor
  derived.derived_from = stencil
  return derived
