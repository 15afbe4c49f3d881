"""`rebalance`: long fp32 sums are cut into groups of at most 32 terms.

Restates reference src/soda/optimization/inline.py:170-262, the one
optimisation pass the reference ALWAYS runs (src/soda/core.py:138, right
after `arithmetic.simplify` and before type propagation).  It changes results:
every group but the last becomes a local statement `cr_var_<n>` whose value is
rounded on its own, and the statement itself becomes `last group + cr_var_0 +
cr_var_1 + ...` -- a different association of the fp32 sum than the text of
the program.  Only programs with more than 32 terms in one fp32 `+` chain are
touched (`tests/src/contrast.soda` is the one in the reference's corpus), but
for those this IS the reference's arithmetic, so the oracle and the kernels
must follow it.

What the reference does, kept step by step:
  * only statements of a type listed in REBALANCE_THRESHOLDS (`float`);
  * only if the whole expression is one chain of `+` (no `-`);
  * a term `coeff * (a + b + ...)` or `(a + b + ...) * coeff` counts as
    len(a + b + ...) items and is rebuilt as `(a + b + ...) * coeff`; any other
    term counts as one item;
  * terms are sorted by item count, largest first (stable);
  * groups are filled greedily in that order, a new group starting when the
    next term would push the count past the threshold;
  * groups 0 .. n-2 become local statements (names from Stencil.new_cr_var,
    store index all zeros, the statement's `let`s copied, type = the type of
    the group's sum), appended to the program's locals; the statement keeps
    group n-1 followed by references to the new locals, in order;
  * start over until nothing changes.
"""
import logging
from typing import List, Optional, Tuple

from soda_amd import grammar, ir

_logger = logging.getLogger(__name__)

REBALANCE_THRESHOLDS = {
    ir.Type('float'): 32,
}


def _is_sum(node: ir.Node) -> bool:
  return isinstance(node, ir.Chain) and node.level == 'add_sub'


def _typed(node: ir.Node, table) -> ir.Type:
  """Type the expression would get from type propagation (operand types from
  `table`), without touching `node`."""

  def tag(n):
    if isinstance(n, (ir.Ref, ir.Var)) and n.haoda_type is None and \
        n.name in table:
      if isinstance(n, ir.Ref):
        return ir.Ref(n.name, n.idx, n.lat, table[n.name])
      return ir.Var(n.name, n.idx, table[n.name])
    return n

  return node.transform(tag).haoda_type


def rebalance(stencil):
  """Modifies `stencil` in place (reference inline.py:175-262); returns it."""
  for stmt in list(stencil.local_stmts) + list(stencil.output_stmts):
    if stmt.haoda_type not in REBALANCE_THRESHOLDS:
      continue
    expr = stmt.expr
    if not (_is_sum(expr) and set(expr.operators) == {'+'}):
      continue
    threshold = REBALANCE_THRESHOLDS[stmt.haoda_type]
    reduction: List[Tuple[Optional[ir.Node], ir.Node]] = []
    for operand in expr.operands:
      if isinstance(operand, ir.Chain) and operand.operators == ('*',):
        a, b = operand.operands
        if _is_sum(a):
          reduction.append((b, a))
        elif _is_sum(b):
          reduction.append((a, b))
        else:
          reduction.append((None, operand))
      else:
        reduction.append((None, operand))

    def num_items(x) -> int:
      return 1 if x[0] is None else len(x[1].operands)

    reduction.sort(key=num_items, reverse=True)     # stable, as the reference's
    count = 0
    groups: List[list] = [[]]
    for item in reduction:
      if count + num_items(item) > threshold:
        groups.append([])
        count = 0
      groups[-1].append(item)
      count += num_items(item)
    if len(groups) == 1:
      continue
    _logger.info("stmt %s has too many operations, breaking'em into %d",
                 stmt.name, len(groups))
    table = dict(stencil.symbol_table)
    table.update((p.name, p.haoda_type) for p in stencil.param_stmts)
    for let in stmt.let:
      if let.haoda_type is not None:
        table[let.name] = let.haoda_type
    new_exprs = []
    for group in groups:
      operands = []
      for coeff, opds in group:
        operands.append(opds if coeff is None else
                        ir.Chain((opds, coeff), ('*',)))
      # (a group of one term is that term: a chain needs two operands)
      new_exprs.append(operands[0] if len(operands) == 1 else
                       ir.Chain(operands, ('+',) * (len(operands) - 1)))
    new_stmts = []
    for new_expr in new_exprs[:-1]:
      name = stencil.new_cr_var()
      new_stmts.append(
          grammar.LocalStmt(_typed(new_expr, table),
                            ir.Ref(name, (0,) * len(stmt.ref.idx)), new_expr,
                            let=stmt.let))
      stencil.local_stmts.append(new_stmts[-1])
      _logger.debug('new stmt: %s', new_stmts[-1])
    last = new_exprs[-1]
    last_operands = list(last.operands) if _is_sum(last) else [last]
    refs = [ir.Ref(s.name, s.ref.idx) for s in new_stmts]
    stmt.expr = ir.Chain(last_operands + refs,
                         ('+',) * (len(last_operands) + len(refs) - 1))
    return rebalance(stencil)
  return stencil
