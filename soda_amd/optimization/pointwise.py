"""`inline_pointwise`: locals that are only ever read at the cell being computed
are folded into their consumers.

A kernel-quality pass of the HIP backend.  (The reference has an `inline`
optimisation of its own, reference src/soda/optimization/inline.py:20-168,
off by default and aimed at FPGA resources; this one only removes tensors that
cost the GPU kernels register windows for nothing.)

denoise3d (reference tests/src/denoise3d.soda:8-29) writes its update in ten
statements: six differences `diff_* = u(0,0,0) - u(...)` that only `g` reads,
each at (0,0,0); `r0`, read only by `r1` at (0,0,0); `r1`, read only by the
output at (0,0,0).  As tensors, every one of them gets a register window in the
marching kernels (rows held x planes kept x cells per lane) and a slot in the
pipeline; folded into their consumers they are sub-expressions the compiler
keeps in a register for a few instructions.  Four statements instead of ten
(denoise2d: three instead of eight).

Bit-exact by construction: a statement `local T: L(s) = e` read as `L(s)` by a
consumer is replaced there by `T(e)` -- the same C expression, the same cast to
the statement's type the store would have applied (a no-op for `float`, the
wrap for a narrow integer).  The consumer is evaluated only inside its own
valid box, which lies inside L's (its window contains L's), so the zeros the
reference's loop nest keeps outside L's box are never seen.  Statements with
`let` bindings, outputs, and locals read at any other offset stay tensors; so
do the `cr_var_*` groups of a rebalanced sum (they ARE the association).
"""
from typing import Dict, List

from soda_amd import core, ir

MAX_OPS = 96         # a local costing more than this per cell stays a tensor


def _shift(expr: ir.Node, by, params) -> ir.Node:
  def fn(n):
    if isinstance(n, ir.Ref) and n.name not in params:
      return ir.Ref(n.name, tuple(a + b for a, b in zip(n.idx, by)), n.lat,
                    n.haoda_type)
    return n
  return expr.transform(fn)


def inline_pointwise(stencil: core.Stencil, fold_groups: bool = False,
                     max_ops: int = MAX_OPS) -> core.Stencil:
  """The derived program (a new Stencil), or `stencil` itself.  `fold_groups`:
  also fold the `cr_var_*` groups of a rebalanced sum -- each becomes the
  parenthesised, cast sub-expression `T(group)` of the final sum, i.e. the
  SAME association (a group is summed on its own, rounded to T, then added),
  only no longer a tensor; for kernel families that evaluate a whole stage per
  thread (ldswin) and have no use for a register window per group."""
  params = set(stencil.param_names)
  stmts = {s.name: s for s in stencil.local_stmts}
  consumers: Dict[str, List] = {}
  for s in stencil.local_stmts + stencil.output_stmts:
    for node in [l.expr for l in s.let] + [s.expr]:
      for ref in ir.get_loads(node):
        if ref.name in stmts:
          consumers.setdefault(ref.name, []).append((s, ref))
  fold = {}
  for name, s in stmts.items():
    if s.let or name not in consumers:
      continue
    if name.startswith('cr_var_') and not fold_groups:
      continue
    if any(ref.idx != s.ref.idx for _, ref in consumers[name]):
      continue
    if ir.op_count(s.expr) > max_ops:
      continue
    fold[name] = s
  if not fold:
    return stencil
  # fold in statement order, so that a folded local's expression already has
  # the locals it reads folded in (r0 into r1, then r1 into the output)
  bodies: Dict[str, ir.Node] = {}

  def substitute(expr: ir.Node) -> ir.Node:
    def fn(n):
      if isinstance(n, ir.Ref) and n.name in bodies:
        return ir.Cast(fold[n.name].haoda_type, bodies[n.name])
      return n
    return expr.transform(fn)

  lines = ['kernel: %s' % stencil.app_name,
           'burst width: %d' % stencil.burst_width,
           'iterate: %d' % stencil.iterate,
           'unroll factor: %d' % stencil.unroll_factor]
  lines.extend(str(s) for s in stencil.input_stmts + stencil.param_stmts)
  for s in stencil.local_stmts + stencil.output_stmts:
    expr = substitute(s.expr)
    if s.name in fold:
      # (refs of a folded local are at its own store index: no shift needed)
      bodies[s.name] = expr
      continue
    head = str(s).split(':', 1)[0] if not s.let else None
    if s.let:
      # a consumer with lets: substitute inside them as well
      lets = ' '.join('%s = %s' % ((('%s ' % l.haoda_type) if l.haoda_type
                                    else '') + l.name,
                                   substitute(l.expr).text()) for l in s.let)
      lines.append('%s: %s %s = %s' % (str(s).split(':', 1)[0], lets,
                                       s.ref.text(), expr.text()))
    else:
      lines.append('%s: %s = %s' % (head, s.ref.text(), expr.text()))
  lines.append('border: %s' % stencil.border)
  lines.append('cluster: %s' % stencil.cluster)
  derived = core.from_text('\n'.join(lines) + '\n')
  derived.replication_factor = stencil.replication_factor
  derived.derived_from = stencil
  return derived
