"""Expression IR of the .soda DSL: types, nodes, printing, C rendering.

The reference keeps its expression IR in the un-vendored `haoda` package
(reference src/setup.py:42; call sites src/soda/grammar.py:46,118-136,
src/soda/codegen/frt/host.py:613-624).  This is an independent design with the
same observable behaviour at those call sites:

  * `str(node)` reproduces the DSL text in the normal form the reference's
    unit test pins (reference src/tests/test_grammar.py:28-54): every
    multi-operand operator chain is parenthesised except at statement, let,
    cast and call-argument level; references print as `a(0, 1) ~lat`.
  * `c_expr(node, ...)` prints C/C++ text whose evaluation follows the usual
    C arithmetic rules (integer promotion, truncating `/`, left-to-right
    association) -- that text is what both the HIP kernels and the CPU oracle
    compile, exactly as the reference pastes `c_expr` into its generated host
    (frt/host.py:616-623) and kernel.

Operator chains are stored n-ary per precedence level so the textual
left-to-right order of floating-point operations is never changed.
"""
import re
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

from soda_amd import util

# --------------------------------------------------------------------------
# types
# --------------------------------------------------------------------------

_FIXED_RE = re.compile(r'^(u?)int([1-9]\d*)(?:_([1-9]\d*))?$')
_FLOAT_RE = re.compile(r'^float([1-9]\d*)(?:_([1-9]\d*))?$')


class Type:
  """A DSL scalar type such as `uint16`, `float`, `int27`, `float18_3`."""

  __slots__ = ('name',)

  def __init__(self, name: str):
    name = str(name)
    if not (name in ('float', 'double', 'half') or _FIXED_RE.match(name) or
            _FLOAT_RE.match(name)):
      raise util.SemanticError('unknown type `%s`' % name)
    self.name = name

  def __str__(self) -> str:
    return self.name

  def __repr__(self) -> str:
    return 'Type(%r)' % self.name

  def __eq__(self, other) -> bool:
    return str(self) == str(other)

  def __ne__(self, other) -> bool:
    return not self == other

  def __hash__(self) -> int:
    return hash(self.name)

  @property
  def is_float(self) -> bool:
    return self.name in ('float', 'double', 'half') or bool(
        _FLOAT_RE.match(self.name))

  @property
  def is_fixed(self) -> bool:
    return bool(_FIXED_RE.match(self.name))

  @property
  def is_signed(self) -> bool:
    return self.is_float or not self.name.startswith('u')

  @property
  def width_in_bits(self) -> int:
    if self.name == 'float':
      return 32
    if self.name == 'double':
      return 64
    if self.name == 'half':
      return 16
    m = _FIXED_RE.match(self.name)
    if m:
      return int(m.group(2))
    m = _FLOAT_RE.match(self.name)
    return int(m.group(1))

  @property
  def is_native(self) -> bool:
    """True if the type maps onto a plain C scalar (what the HIP path runs)."""
    if self.name in ('float', 'double'):
      return True
    m = _FIXED_RE.match(self.name)
    if m and m.group(3) is None:
      return int(m.group(2)) in (8, 16, 32, 64)
    m = _FLOAT_RE.match(self.name)
    if m and m.group(2) is None:
      return int(m.group(1)) in (32, 64)
    return False

  @property
  def c_type(self) -> str:
    if self.name in ('float', 'float32'):
      return 'float'
    if self.name in ('double', 'float64'):
      return 'double'
    m = _FIXED_RE.match(self.name)
    if m and m.group(3) is None and int(m.group(2)) in (8, 16, 32, 64):
      return '%sint%s_t' % (m.group(1), m.group(2))
    raise util.SemanticError(
        'type `%s` has no plain C equivalent; the HIP backend and the oracle '
        'only run 8/16/32/64-bit integers, float and double' % self.name)

  @property
  def np_name(self) -> str:
    """numpy dtype name, for the host side."""
    c = self.c_type
    return {'float': 'float32', 'double': 'float64'}.get(c, c[:-2])

  @property
  def size_in_bytes(self) -> int:
    return self.width_in_bits // 8


def common_type(types: Iterable[Optional[Type]]) -> Optional[Type]:
  """Type of an operator chain for *printing* purposes (float beats fixed, wider
  beats narrower).  Numerics never depend on this: they come from compiling
  the C text."""
  best = None
  for t in types:
    if t is None:
      continue
    if best is None:
      best = t
    elif t.is_float and not best.is_float:
      best = t
    elif t.is_float == best.is_float and t.width_in_bits > best.width_in_bits:
      best = t
  return best


# --------------------------------------------------------------------------
# nodes
# --------------------------------------------------------------------------

class Node:
  """Base class.  Nodes are immutable-by-convention; `rebuild` maps children."""

  haoda_type: Optional[Type] = None

  def children(self) -> Tuple['Node', ...]:
    return ()

  def rebuild(self, children: Sequence['Node']) -> 'Node':
    return self

  def transform(self, fn: Callable[['Node'], 'Node']) -> 'Node':
    """Post-order rewrite: children first, then `fn` on the rebuilt node."""
    kids = self.children()
    node = self
    if kids:
      new_kids = tuple(k.transform(fn) for k in kids)
      if any(a is not b for a, b in zip(kids, new_kids)):
        node = self.rebuild(new_kids)
    return fn(node)

  def walk(self):
    yield self
    for k in self.children():
      yield from k.walk()

  @property
  def is_chain(self) -> bool:
    return False

  def text(self) -> str:
    raise NotImplementedError

  def __str__(self) -> str:
    return self.text()


def _operand_text(node: Node) -> str:
  s = node.text()
  return '(%s)' % s if node.is_chain else s


class Num(Node):
  """A literal, kept as written (`0.2f`, `.125f`, `65535`, `0x10u`)."""

  def __init__(self, text: str):
    self.literal = text

  def text(self) -> str:
    return self.literal

  @property
  def is_float_literal(self) -> bool:
    t = self.literal.lower()
    if t.startswith('0x') or t.startswith('0b'):
      return False
    return '.' in t or 'e' in t or t.endswith('f')

  @property
  def haoda_type(self) -> Type:
    t = self.literal.lower()
    if self.is_float_literal:
      return Type('float') if t.endswith('f') else Type('double')
    suffix = t[len(t.rstrip('ul')):]
    return Type('%sint%d' % ('u' if 'u' in suffix else '',
                             64 if 'l' in suffix else 32))

  @property
  def c_literal(self) -> str:
    t = self.literal
    if t.lower().startswith('0b'):  # not C; print the value
      digits = t.rstrip('uUlL')
      return '%d%s' % (int(digits[2:], 2), t[len(digits):])
    return t


class Ref(Node):
  """`name(i, j, ...) ~lat`: a tensor element relative to the current cell."""

  def __init__(self, name: str, idx: Sequence[int], lat: Optional[int] = None,
               haoda_type: Optional[Type] = None):
    self.name = name
    self.idx = tuple(int(i) for i in idx)
    self.lat = lat
    self.haoda_type = haoda_type

  def text(self) -> str:
    s = '%s(%s)' % (self.name, ', '.join(map(str, self.idx)))
    if self.lat is not None:
      s += ' ~%d' % self.lat
    return s

  def with_name(self, name: str) -> 'Ref':
    return Ref(name, self.idx, self.lat, self.haoda_type)

  def __eq__(self, other) -> bool:
    return (isinstance(other, Ref) and self.name == other.name and
            self.idx == other.idx)

  def __hash__(self) -> int:
    return hash((self.name, self.idx))


class Var(Node):
  """A let-bound scalar, or a `param` element `p[1][3]`."""

  def __init__(self, name: str, idx: Sequence[int] = (),
               haoda_type: Optional[Type] = None):
    self.name = name
    self.idx = tuple(int(i) for i in idx)
    self.haoda_type = haoda_type

  def text(self) -> str:
    return self.name + ''.join('[%d]' % i for i in self.idx)


class Cast(Node):

  def __init__(self, haoda_type: Type, expr: Node):
    self.haoda_type = haoda_type
    self.expr = expr

  def children(self):
    return (self.expr,)

  def rebuild(self, children):
    return Cast(self.haoda_type, children[0])

  def text(self) -> str:
    return '%s(%s)' % (self.haoda_type, self.expr.text())


class Call(Node):

  def __init__(self, name: str, args: Sequence[Node]):
    self.name = name
    self.args = tuple(args)

  def children(self):
    return self.args

  def rebuild(self, children):
    return Call(self.name, children)

  def text(self) -> str:
    return '%s(%s)' % (self.name, ', '.join(a.text() for a in self.args))

  @property
  def haoda_type(self):
    return common_type(a.haoda_type for a in self.args)


class Unary(Node):
  """A prefix chain such as `+-+-l` (operators kept in source order)."""

  def __init__(self, ops: Sequence[str], operand: Node):
    self.ops = tuple(ops)
    self.operand = operand

  def children(self):
    return (self.operand,)

  def rebuild(self, children):
    return Unary(self.ops, children[0])

  def text(self) -> str:
    return ''.join(self.ops) + _operand_text(self.operand)

  @property
  def haoda_type(self):
    return self.operand.haoda_type


# precedence levels, loosest first (reference src/soda/grammar.py:215-226 gives
# the class order; the operator spellings are C's)
LEVELS = (
    ('logic_or', ('||',)),
    ('logic_and', ('&&',)),
    ('binary_or', ('|',)),
    ('xor', ('^',)),
    ('binary_and', ('&',)),
    ('eq_cmp', ('==', '!=')),
    ('lt_cmp', ('<=', '>=', '<', '>')),
    ('add_sub', ('+', '-')),
    ('mul_div', ('*', '/', '%')),
)
_LEVEL_OF_OP = {op: name for name, ops in LEVELS for op in ops}


class Chain(Node):
  """`a op b op c ...` with operators of one precedence level, left-assoc."""

  def __init__(self, operands: Sequence[Node], operators: Sequence[str]):
    if len(operands) != len(operators) + 1 or len(operands) < 2:
      raise util.InternalError('malformed operator chain')
    self.operands = tuple(operands)
    self.operators = tuple(operators)

  @property
  def level(self) -> str:
    return _LEVEL_OF_OP[self.operators[0]]

  @property
  def is_chain(self) -> bool:
    return True

  def children(self):
    return self.operands

  def rebuild(self, children):
    return Chain(children, self.operators)

  def text(self) -> str:
    parts = [_operand_text(self.operands[0])]
    for op, operand in zip(self.operators, self.operands[1:]):
      parts.append(op)
      parts.append(_operand_text(operand))
    return ' '.join(parts)

  @property
  def haoda_type(self):
    if self.level in ('logic_or', 'logic_and', 'eq_cmp', 'lt_cmp'):
      return Type('uint1')
    return common_type(o.haoda_type for o in self.operands)


class Let(Node):
  """`[type] name = expr` in front of a local/output statement."""

  def __init__(self, haoda_type: Optional[Type], name: str, expr: Node):
    self.haoda_type = haoda_type
    self.name = name
    self.expr = expr

  def children(self):
    return (self.expr,)

  def rebuild(self, children):
    return Let(self.haoda_type, self.name, children[0])

  def text(self) -> str:
    prefix = '%s ' % self.haoda_type if self.haoda_type is not None else ''
    return '%s%s = %s' % (prefix, self.name, self.expr.text())


# --------------------------------------------------------------------------
# queries
# --------------------------------------------------------------------------

def get_loads(node: Node) -> List[Ref]:
  """Every `Ref` in evaluation (source) order, duplicates kept."""
  return [n for n in node.walk() if isinstance(n, Ref)]


def get_load_dict(nodes: Iterable[Node]) -> Dict[str, List[Ref]]:
  """name -> loads, in first-appearance order (ref visitor.py:68-90)."""
  out: Dict[str, List[Ref]] = {}
  for node in nodes:
    for ref in get_loads(node):
      out.setdefault(ref.name, []).append(ref)
  return out


def get_vars(node: Node) -> List[Var]:
  return [n for n in node.walk() if isinstance(n, Var)]


def op_count(node: Node) -> int:
  """Arithmetic operations one evaluation of the expression costs (a rough
  instruction count for the launch-time model of the HIP backend)."""
  n = 0
  for x in node.walk():
    if isinstance(x, Chain):
      n += len(x.operators) * (4 if x.level == 'mul_div' and any(
          op in '/%' for op in x.operators) else 1)
    elif isinstance(x, Unary):
      n += len(x.ops)
    elif isinstance(x, Call):
      n += 8 if x.name not in ('min', 'max', 'fmin', 'fmax', 'abs', 'fabs',
                               'select') else max(1, len(x.args) - 1)
    elif isinstance(x, Cast):
      n += 1
  return n


def flatten(node: Node) -> Node:
  """The only simplification the reference applies by default
  (`arithmetic.simplify`, core.py:131): drop no-op wrappers.  Operand order
  and association are untouched."""

  def fn(n: Node) -> Node:
    if isinstance(n, Unary) and not n.ops:
      return n.operand
    return n

  return node.transform(fn)


# --------------------------------------------------------------------------
# C rendering
# --------------------------------------------------------------------------

# DSL call name -> (float spelling, double spelling, integer spelling)
_C_FUNCS = {
    'sqrt': ('sqrtf', 'sqrt', 'sqrt'),
    'cbrt': ('cbrtf', 'cbrt', 'cbrt'),
    'exp': ('expf', 'exp', 'exp'),
    'exp2': ('exp2f', 'exp2', 'exp2'),
    'log': ('logf', 'log', 'log'),
    'log2': ('log2f', 'log2', 'log2'),
    'log10': ('log10f', 'log10', 'log10'),
    'sin': ('sinf', 'sin', 'sin'),
    'cos': ('cosf', 'cos', 'cos'),
    'tan': ('tanf', 'tan', 'tan'),
    'asin': ('asinf', 'asin', 'asin'),
    'acos': ('acosf', 'acos', 'acos'),
    'atan': ('atanf', 'atan', 'atan'),
    'floor': ('floorf', 'floor', 'floor'),
    'ceil': ('ceilf', 'ceil', 'ceil'),
    'round': ('roundf', 'round', 'round'),
    'fabs': ('fabsf', 'fabs', 'fabs'),
    'pow': ('powf', 'pow', 'pow'),
}
FUNC_NAMES = frozenset(_C_FUNCS) | {'min', 'max', 'fmin', 'fmax', 'abs',
                                    'select'}


def c_expr(node: Node,
           load: Callable[[Ref], str],
           var: Callable[[Var], str] = lambda v: v.text()) -> str:
  """C text of `node`.  `load(ref)` spells a tensor element, `var(v)` a let
  variable or param element.  Sub-expressions are always parenthesised, so C
  precedence can never regroup what the DSL grouped."""
  if isinstance(node, _Text):
    return node.text()
  if isinstance(node, Num):
    return node.c_literal
  if isinstance(node, Ref):
    return load(node)
  if isinstance(node, Var):
    return var(node)
  if isinstance(node, Cast):
    return '((%s)(%s))' % (node.haoda_type.c_type, c_expr(node.expr, load, var))
  if isinstance(node, Unary):
    inner = c_expr(node.operand, load, var)
    for op in reversed(node.ops):
      inner = '(%s%s)' % (op, inner)
    return inner
  if isinstance(node, Chain):
    parts = [c_expr(node.operands[0], load, var)]
    for op, operand in zip(node.operators, node.operands[1:]):
      parts.append(op)
      parts.append(c_expr(operand, load, var))
    return '(%s)' % ' '.join(parts)
  if isinstance(node, Call):
    args = [c_expr(a, load, var) for a in node.args]
    t = node.haoda_type
    if node.name in ('min', 'max', 'fmin', 'fmax'):
      # n-ary, left fold; SODA_MIN/SODA_MAX are type-generic (soda_rt.h / oracle prelude)
      macro = 'SODA_MIN' if node.name in ('min', 'fmin') else 'SODA_MAX'
      out = args[0]
      for a in args[1:]:
        out = '%s(%s, %s)' % (macro, out, a)
      return out
    if node.name == 'abs':
      return 'SODA_ABS(%s)' % args[0]
    if node.name == 'select':
      if len(args) != 3:
        raise util.SemanticError('select() takes 3 arguments')
      return '((%s) ? (%s) : (%s))' % tuple(args)
    if node.name.startswith('soda_'):
      # a backend intrinsic (codegen/hip/exact.py): made by a lowering, never
      # parsed -- FUNC_NAMES is what the grammar accepts
      return '%s(%s)' % (node.name, ', '.join(args))
    if node.name in _C_FUNCS:
      f32, f64, fint = _C_FUNCS[node.name]
      if t is not None and t.is_float and t.width_in_bits <= 32:
        fn = f32
      elif t is not None and t.is_float:
        fn = f64
      else:
        fn = fint
      return '%s(%s)' % (fn, ', '.join(args))
    raise util.SemanticError('unknown function `%s`' % node.name)
  if isinstance(node, Let):
    raise util.InternalError('render lets through their .expr')
  raise util.InternalError('cannot render %r' % (node,))


def c_statements(node: Node, loads: Sequence[Callable[[Ref], str]],
                 fresh: Callable[[], str],
                 var: Callable[[Var], str] = lambda v: v.text(),
                 stmts: Optional[List[str]] = None
                 ) -> Tuple[List[str], List[str]]:
  """`node` evaluated for several cells at once, as three-address C++
  statements emitted OPERATION-major, cell-minor: every binary operation of
  the tree becomes one `const auto t = a op b;` per cell, and the statements of
  the different cells are interleaved.  Same operations in the same order per
  cell as `c_expr` (so the same bits); what changes is that consecutive
  statements are independent, which is the order the GPU's in-order waves want
  (a wave issues back-to-back only if the next instruction does not depend on
  the previous one).  Returns (statements, one result expression per cell).
  `loads[c]` spells a tensor element for cell c.  A caller-supplied `stmts`
  list is appended to, so a load callback may itself emit statements (e.g. the
  fetch of the row it is about to name) right before their first use."""
  n = len(loads)
  if stmts is None:
    stmts = []

  def leaf(texts):
    return texts

  def ev(nd) -> List[str]:
    if isinstance(nd, Num):
      return [nd.c_literal] * n
    if isinstance(nd, Ref):
      return [loads[c](nd) for c in range(n)]
    if isinstance(nd, Var):
      return [var(nd)] * n
    if isinstance(nd, Cast):
      inner = ev(nd.expr)
      return ['((%s)(%s))' % (nd.haoda_type.c_type, x) for x in inner]
    if isinstance(nd, Unary):
      inner = ev(nd.operand)
      for op in reversed(nd.ops):
        inner = ['(%s%s)' % (op, x) for x in inner]
      return inner
    if isinstance(nd, Chain):
      acc = ev(nd.operands[0])
      for op, operand in zip(nd.operators, nd.operands[1:]):
        rhs = ev(operand)
        names = []
        for c in range(n):
          t = fresh()
          stmts.append('const auto %s = %s %s %s;' % (t, acc[c], op, rhs[c]))
          names.append(t)
        acc = names
      return acc
    # calls: fall back to the nested text per cell (arguments evaluated first)
    if isinstance(nd, Call):
      args = [ev(a) for a in nd.args]
      out = []
      for c in range(n):
        sub = Call(nd.name, [_Text(args[k][c], nd.args[k].haoda_type)
                             for k in range(len(nd.args))])
        out.append(c_expr(sub, loads[c], var))
      return out
    raise util.InternalError('cannot render %r' % (nd,))

  return stmts, ev(node)


class _Text(Node):
  """Already-rendered C text standing in for a sub-expression."""

  def __init__(self, text: str, haoda_type: Optional[Type]):
    self._text = text
    self.haoda_type = haoda_type

  def text(self) -> str:
    return self._text


C_PRELUDE = '''\
/* each argument appears ONCE in the expansion (a nested 19-way min() must not
 * blow up textually) and is evaluated once; the result type is that of the
 * conditional operator, i.e. the usual arithmetic conversions */
#define SODA_MIN(a, b) ({ __auto_type soda_a_ = (a); __auto_type soda_b_ = (b); \\
                          soda_b_ < soda_a_ ? soda_b_ : soda_a_; })
#define SODA_MAX(a, b) ({ __auto_type soda_a_ = (a); __auto_type soda_b_ = (b); \\
                          soda_a_ < soda_b_ ? soda_b_ : soda_a_; })
#define SODA_ABS(a) ({ __auto_type soda_a_ = (a); \\
                       soda_a_ < 0 ? -soda_a_ : soda_a_; })
'''
