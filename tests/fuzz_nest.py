"""Random programs with an oracle of their own (TEST INFRASTRUCTURE).

tests/fuzz.py prints random programs as text; product AND oracle then read
that text through ONE front-end (the product's parser, tap extraction, box
computation and C-expression printer), so a front-end bug is invisible there.
This generator builds every program as a small expression tree and emits from
the TREE, side by side,

  (i)  the .soda text the product parses -- with the FEWEST parentheses C
       precedence and left-to-right association allow, so grouping is the
       parser's job (the reference's expression grammar: `Expr` ... `Operand`,
       reference src/soda/grammar.py:209-232, haoda's `ir.GRAMMAR`), and
  (ii) a C++ loop nest, fully parenthesised from the tree, whose iteration
       boxes come from this file's own tap lists and whose arithmetic
       conversions are the host compiler's (g++), the way the reference's
       generated host checks itself (reference src/soda/codegen/frt/host.py:
       558-624: zero-initialised tensors, one nest per tensor per iteration,
       load index x + idx - store idx, cast on store; boxes :565-577 with the
       always-truthy `tensor.is_output` of :567, i.e. every tensor's window is
       taken relative to the program INPUTS; iteration chaining reference
       src/soda/core.py:320-336).

Nothing here imports the product or the oracle package.  Three families, as in tests/fuzz.py:
`plain` (mixed-precedence arithmetic, lets, locals, store offsets, min / max,
integer division, a top-level sqrt), `window` (integer sums / min / max over
contiguous taps, several stages), `rich` (select over comparisons and logic,
unary minus, casts, float division, %, &, |, ^, 64-bit detours, abs).  Only
constructs whose C++ meaning does not depend on a library overload choice are
generated (min / max of equal types; sqrt only as a whole float statement).
"""
import ctypes
import hashlib
import os
import subprocess
import tempfile

import numpy as np

CTYPES = {'uint8': 'uint8_t', 'int16': 'int16_t', 'uint16': 'uint16_t',
          'int32': 'int32_t', 'int64': 'int64_t', 'float': 'float',
          'double': 'double'}
NPTYPES = {'uint8': np.uint8, 'int16': np.int16, 'uint16': np.uint16,
           'int32': np.int32, 'int64': np.int64, 'float': np.float32,
           'double': np.float64}
FLOAT_TYPES = ['float', 'double']
INT_TYPES = ['uint8', 'int16', 'uint16', 'int32']

# binding strength, C's (and the reference grammar's class order)
PREC = {'||': 1, '&&': 2, '|': 3, '^': 4, '&': 5, '==': 6, '!=': 6,
        '<': 7, '<=': 7, '>': 7, '>=': 7, '+': 8, '-': 8,
        '*': 9, '/': 9, '%': 9}
UNARY_PREC, ATOM_PREC = 10, 11


def _rank(t):
  """Usual arithmetic conversions on this file's type names."""
  return {'double': 5, 'float': 4, 'int64': 3}.get(t, 2)    # narrow ints -> int


def _common(a, b):
  r = max(_rank(a), _rank(b))
  return {5: 'double', 4: 'float', 3: 'int64', 2: 'int32'}[r]


class Ref:
  def __init__(self, name, idx, typ):
    self.name, self.idx, self.typ = name, tuple(idx), typ

  def soda(self):
    return '%s(%s)' % (self.name, ', '.join(map(str, self.idx))), ATOM_PREC

  def cpp(self, load):
    return load(self)

  def ctype(self):
    return self.typ

  def refs(self):
    yield self


class Var:
  def __init__(self, name, typ):
    self.name, self.typ = name, typ

  def soda(self):
    return self.name, ATOM_PREC

  def cpp(self, load):
    return self.name

  def ctype(self):
    return self.typ

  def refs(self):
    return iter(())


class Lit:
  def __init__(self, text):
    self.text = text

  def soda(self):
    return self.text, ATOM_PREC

  def cpp(self, load):
    return self.text

  def ctype(self):
    if self.text.endswith('f'):
      return 'float'
    return 'double' if '.' in self.text else 'int32'

  def refs(self):
    return iter(())


class Bin:
  def __init__(self, op, a, b):
    self.op, self.a, self.b = op, a, b

  def soda(self):
    p = PREC[self.op]
    ta, pa = self.a.soda()
    tb, pb = self.b.soda()
    if pa < p:                 # left operand binds weaker: must be grouped
      ta = '(%s)' % ta
    if pb <= p:                # right operand of a left-associative operator
      tb = '(%s)' % tb
    return '%s %s %s' % (ta, self.op, tb), p

  def cpp(self, load):
    return '(%s %s %s)' % (self.a.cpp(load), self.op, self.b.cpp(load))

  def ctype(self):
    if PREC[self.op] <= 2 or 6 <= PREC[self.op] <= 7:
      return 'int32'           # bool: only ever a select() condition here
    return _common(self.a.ctype(), self.b.ctype())

  def refs(self):
    yield from self.a.refs()
    yield from self.b.refs()


class Neg:
  def __init__(self, a):
    self.a = a

  def soda(self):
    t, p = self.a.soda()
    if p < UNARY_PREC or t.startswith('-'):
      t = '(%s)' % t
    return '-' + t, UNARY_PREC

  def cpp(self, load):
    return '(-%s)' % self.a.cpp(load)

  def ctype(self):
    return _common(self.a.ctype(), 'int32')

  def refs(self):
    yield from self.a.refs()


class Cast:
  def __init__(self, typ, a):
    self.typ, self.a = typ, a

  def soda(self):
    return '%s(%s)' % (self.typ, self.a.soda()[0]), ATOM_PREC

  def cpp(self, load):
    return '((%s)(%s))' % (CTYPES[self.typ], self.a.cpp(load))

  def ctype(self):
    return self.typ

  def refs(self):
    yield from self.a.refs()


class Call:
  """min / max (n-ary, equal argument types), select, sqrt, abs."""

  def __init__(self, name, args):
    self.name, self.args = name, list(args)

  def soda(self):
    return '%s(%s)' % (self.name,
                       ', '.join(a.soda()[0] for a in self.args)), ATOM_PREC

  def cpp(self, load):
    args = [a.cpp(load) for a in self.args]
    if self.name in ('min', 'max'):
      out = args[0]
      for a in args[1:]:
        out = 'std::%s(%s, %s)' % (self.name, out, a)
      return out
    if self.name == 'select':
      return '((%s) ? (%s) : (%s))' % tuple(args)
    if self.name == 'sqrt':
      return 'std::sqrt(%s)' % args[0]
    if self.name == 'abs':
      return 'std::abs(%s)' % args[0]
    raise ValueError(self.name)

  def ctype(self):
    if self.name in ('min', 'max'):
      return self.args[0].ctype()      # std::min<T>: T (equal types, below)
    if self.name == 'select':          # ?: converts only operands that differ
      a, b = self.args[1].ctype(), self.args[2].ctype()
      return a if a == b else _common(a, b)
    return _common(self.args[0].ctype(), 'int32')

  def refs(self):
    for a in self.args:
      yield from a.refs()


def _same_type(args):
  """min / max take arguments of ONE type (std::min deduces a single T): when
  the arguments' types differ, those that are not of the usual-conversion type
  of all of them are cast to it.  (`ctype()` is the EXACT C++ type of a node:
  a tap of a uint8 tensor is a uint8_t, a sum of two is an int.)"""
  types = [a.ctype() for a in args]
  if all(t == types[0] for t in types):
    return list(args)
  want = types[0]
  for t in types[1:]:
    want = _common(want, t)
  return [a if t == want else Cast(want, a) for a, t in zip(args, types)]


class Stmt:
  def __init__(self, kind, typ, name, store, expr, lets=()):
    self.kind, self.typ, self.name = kind, typ, name
    self.store, self.expr, self.lets = tuple(store), expr, list(lets)

  def refs(self):
    for _, _, e in self.lets:
      yield from e.refs()
    yield from self.expr.refs()


class Program:
  """inputs: [(name, type)]; stmts: locals and outputs in file order."""

  def __init__(self, name, dim, iterate, inputs, stmts):
    self.name, self.dim, self.iterate = name, dim, iterate
    self.inputs, self.stmts = list(inputs), list(stmts)
    self.outputs = [s for s in stmts if s.kind == 'output']

  # ---- (i) the text the product reads --------------------------------------
  def soda_text(self):
    lines = ['kernel: %s' % self.name, 'burst width: 64', 'unroll factor: 2',
             'iterate: %d' % self.iterate]
    for i, (n, t) in enumerate(self.inputs):
      decl = n
      if self.dim > 1 and i == 0:
        decl += '(%s, *)' % ', '.join(['32'] * (self.dim - 1))
      lines.append('input %s: %s' % (t, decl))
    for s in self.stmts:
      ref = '%s(%s)' % (s.name, ', '.join(map(str, s.store)))
      head = '%s %s:' % (s.kind, s.typ)
      if s.lets:
        lines.append(head)
        for lt, ln, le in s.lets:
          lines.append('  %s %s = %s' % (lt, ln, le.soda()[0]))
        lines.append('  %s = %s' % (ref, s.expr.soda()[0]))
      else:
        lines.append('%s %s = %s' % (head, ref, s.expr.soda()[0]))
    return '\n'.join(lines) + '\n'

  # ---- boxes from the tap lists ---------------------------------------------
  def margins(self):
    """[(iteration, stmt, lo, margin)] in execution order: the nest of a
    tensor runs over [lo, N - margin) per dimension -- the cells for which
    EVERY load, at every level down to the program inputs, lies inside the
    grid.  By interval propagation: an input is defined on the whole grid; a
    cell x of T is computable iff x + (load index - store index) lies in the
    producer's box, for all loads.

    Where every tensor's window contains its own cell this is the reference's
    box (frt/host.py:565-577 over core.py:876-926, `literal_margins` below);
    where it does not, the reference's nest would index an intermediate array
    out of bounds (`p(0) = in(2)`, `t(0) = p(-3)` reads p[-2] at cell 1), so
    there is no reference behaviour to match and the tighter box is the only
    defined one."""
    dim = self.dim
    zero = tuple([0] * dim)
    box = {n: (zero, zero) for n, _ in self.inputs}     # name -> (lo, margin)
    order = []
    for k in range(self.iterate):
      for s in self.stmts:
        lo, margin = [0] * dim, [0] * dim
        for r in s.refs():
          plo, pmargin = box[r.name]
          for d in range(dim):
            off = r.idx[d] - s.store[d]
            lo[d] = max(lo[d], plo[d] - off)
            margin[d] = max(margin[d], pmargin[d] + off)
        box[s.name] = (tuple(lo), tuple(margin))
        order.append((k, s, tuple(lo), tuple(margin)))
      # the next iteration's inputs are this one's outputs, by position
      for (n, _), o in zip(self.inputs, self.outputs):
        box[n] = box[o.name]
    return order

  def literal_margins(self):
    """The reference's formula to the letter: the window of a tensor = the
    offsets of the program INPUTS it reaches through all its loads (load index
    minus store index, accumulated through the producers: core.py:876-919),
    lo = max(0, -min), margin = max(0, max) (frt/host.py:570-577).  Also
    returns whether every window contains offset 0 in every dimension."""
    dim = self.dim
    zero = (tuple([0] * dim), tuple([0] * dim))
    win = {n: zero for n, _ in self.inputs}
    order, spans_zero = [], True
    for k in range(self.iterate):
      for s in self.stmts:
        mins, maxs = [None] * dim, [None] * dim
        for r in s.refs():
          pmin, pmax = win[r.name]
          for d in range(dim):
            off = r.idx[d] - s.store[d]
            a, b = pmin[d] + off, pmax[d] + off
            mins[d] = a if mins[d] is None else min(mins[d], a)
            maxs[d] = b if maxs[d] is None else max(maxs[d], b)
        win[s.name] = (tuple(mins), tuple(maxs))
        spans_zero = spans_zero and all(a <= 0 <= b
                                        for a, b in zip(mins, maxs))
        order.append((k, s, tuple(max(0, -m) for m in mins),
                      tuple(max(0, m) for m in maxs)))
      for (n, _), o in zip(self.inputs, self.outputs):
        win[n] = win[o.name]
    return order, spans_zero

  def boxes(self, extent):
    """[(iteration, stmt, lo, hi)] on a grid of `extent`."""
    return [(k, s, lo, tuple(extent[d] - margin[d] for d in range(self.dim)))
            for k, s, lo, margin in self.margins()]

  def valid_box(self, extent, out_name):
    return [(lo, hi) for k, s, lo, hi in self.boxes(extent)
            if s.name == out_name][-1]

  # ---- (ii) the loop nest ---------------------------------------------------
  def cpp_text(self):
    dim = self.dim
    out = ['#include <algorithm>', '#include <cmath>', '#include <cstdint>',
           '#include <cstdlib>', '#include <cstring>', '',
           'extern "C" int nest(const void* const* ins, void* const* outs, '
           'const int64_t* N) {']
    cells = ' * '.join('N[%d]' % d for d in range(dim))
    out.append('  const int64_t cells = %s;' % cells)
    boxes = self.margins()                          # [lo, N - margin)
    live = {}
    for i, (n, t) in enumerate(self.inputs):
      out.append('  const %s* %s_0 = (const %s*)ins[%d];' %
                 (CTYPES[t], n, CTYPES[t], i))
      live[n] = '%s_0' % n
    allocs = []
    for k, s, lo, margin in boxes:
      last = k == self.iterate - 1
      var = '%s_%d' % (s.name, k)
      ct = CTYPES[s.typ]
      if s.kind == 'output' and last:
        out.append('  %s* %s = (%s*)outs[%d];' %
                   (ct, var, ct, self.outputs.index(s)))
        out.append('  memset(%s, 0, cells * sizeof(%s));' % (var, ct))
      else:
        out.append('  %s* %s = (%s*)calloc(cells, sizeof(%s));' %
                   (ct, var, ct, ct))
        out.append('  if (!%s) return 1;' % var)
        allocs.append(var)
      pad = '  '
      for d in reversed(range(dim)):
        out.append('%sfor (int64_t x%d = %d; x%d < N[%d] - %d; ++x%d) {' %
                   (pad, d, lo[d], d, d, margin[d], d))
        pad += '  '

      def load(r, _s=s, _live=dict(live)):
        idx = None
        for d in reversed(range(dim)):
          term = '(x%d + (%d))' % (d, r.idx[d] - _s.store[d])
          idx = term if idx is None else '(%s + N[%d] * %s)' % (term, d, idx)
        return '%s[%s]' % (_live[r.name], idx)

      for lt, ln, le in s.lets:
        out.append('%sconst %s %s = (%s)(%s);' %
                   (pad, CTYPES[lt], ln, CTYPES[lt], le.cpp(load)))
      here = None
      for d in reversed(range(dim)):
        here = 'x%d' % d if here is None else '(x%d + N[%d] * %s)' % (d, d, here)
      out.append('%s%s[%s] = (%s)(%s);' %
                 (pad, var, here, ct, s.expr.cpp(load)))
      for d in range(dim):
        pad = pad[:-2]
        out.append('%s}' % pad)
      live[s.name] = var
      if s is self.stmts[-1]:      # end of an iteration: outputs become inputs
        for (n, _), o in zip(self.inputs, self.outputs):
          live[n] = live[o.name]
    for var in allocs:
      out.append('  free((void*)%s);' % var)
    out.append('  return 0;')
    out.append('}')
    return '\n'.join(out) + '\n'

  # ---- run it ----------------------------------------------------------------
  def run(self, inputs, extent):
    """{output: array}: the nest compiled by g++ (-O1 -ffp-contract=off
    -fwrapv) on `inputs` ({name: array shaped extent[::-1]})."""
    lib = _compile(self.cpp_text())
    shape = tuple(extent[::-1])
    ins = [np.ascontiguousarray(inputs[n], dtype=NPTYPES[t])
           for n, t in self.inputs]
    assert all(a.shape == shape for a in ins)
    outs = [np.empty(shape, dtype=NPTYPES[s.typ]) for s in self.outputs]
    c_ins = (ctypes.c_void_p * len(ins))(*[a.ctypes.data for a in ins])
    c_outs = (ctypes.c_void_p * len(outs))(*[a.ctypes.data for a in outs])
    n = (ctypes.c_int64 * self.dim)(*extent)
    lib.nest.restype = ctypes.c_int
    if lib.nest(c_ins, c_outs, n) != 0:
      raise MemoryError('nest: calloc failed')
    return {s.name: a for s, a in zip(self.outputs, outs)}


_BUILD_DIR = None


def _compile(source):
  global _BUILD_DIR
  if _BUILD_DIR is None:
    _BUILD_DIR = tempfile.mkdtemp(prefix='soda_fuzz_nest_')
  key = hashlib.sha1(source.encode()).hexdigest()[:20]
  so = os.path.join(_BUILD_DIR, 'nest_%s.so' % key)
  if not os.path.exists(so):
    src = os.path.join(_BUILD_DIR, 'nest_%s.cpp' % key)
    with open(src, 'w') as f:
      f.write(source)
    subprocess.run(['g++', '-O1', '-ffp-contract=off', '-fwrapv', '-fPIC',
                    '-shared', '-o', so, src], check=True,
                   capture_output=True)
  return ctypes.CDLL(so)


# ---------------------------------------------------------------------------
# generators
# ---------------------------------------------------------------------------

def _idx(rng, dim, radius):
  return tuple(int(rng.integers(-radius, radius + 1)) for _ in range(dim))


def _arith(rng, leaf, is_float, depth, rich, to_int):
  """A random tree over leaf() nodes."""
  sub = lambda: _arith(rng, leaf, is_float, depth + 1, rich, to_int)
  r = rng.random()
  if depth >= 3 or r < 0.22:
    return leaf()
  if r < 0.62:                          # a chain, mixed precedence, left to right
    ops = ['+', '-', '*'] if is_float else ['+', '-', '*', '+', '-']
    node = sub()
    for _ in range(int(rng.integers(1, 4))):
      op = ops[int(rng.integers(len(ops)))]
      rhs = sub()
      if rng.random() < 0.3:            # a tighter-binding group on the right
        rhs = Bin('*', rhs, leaf())
      node = Bin(op, node, rhs)
    return node
  if r < 0.70:
    fn = ['min', 'max'][int(rng.integers(2))]
    return Call(fn, _same_type([sub(), sub()]))
  if r < 0.76:
    if is_float:
      d = sub()
      return Bin('/', sub(), Bin('+', Lit('1.5f'), Bin('*', d, d)))
    return Bin('/', sub(), Lit(str(int(rng.integers(2, 7)))))
  if not rich:
    return Bin('*', sub(), Lit('%.3ff' % rng.uniform(0.1, 2.0) if is_float
                               else str(int(rng.integers(1, 9)))))
  if r < 0.83:
    cmp_op = ['<', '<=', '>', '>=', '==', '!='][int(rng.integers(6))]
    cond = Bin(cmp_op, sub(), sub())
    if rng.random() < 0.4:
      other = Bin(['<', '>'][int(rng.integers(2))], sub(), sub())
      cond = Bin(['&&', '||'][int(rng.integers(2))], cond, other)
    return Call('select', [cond, sub(), sub()])
  if r < 0.88:
    return Neg(sub())
  if is_float:
    if r < 0.94:
      return Cast(['float', 'double'][int(rng.integers(2))], sub())
    if not to_int:     # (beyond int32 the conversion is undefined: iterated
      return sub()     # programs, whose values grow, do without)
    return Cast('float', Cast('int32', Bin('*', sub(), Lit('5.0f'))))
  if r < 0.92:
    return Bin('%', sub(), Lit(str(int(rng.integers(2, 9)))))
  if r < 0.97:
    return Bin(['&', '|', '^'][int(rng.integers(3))], sub(), sub())
  if rng.random() < 0.5:
    return Call('abs', [Bin('-', sub(), sub())])
  return Cast('int32', Bin('%', Bin('*', Cast('int64', sub()),
                                    Lit('100003')), Lit('1009')))


def program(seed, family='plain'):
  """(Program, extent) for `family` in plain / rich / window / wide."""
  if family == 'window':
    return _window_program(seed)
  if family == 'wide':
    return _wide_program(seed)
  rich = family == 'rich'
  rng = np.random.default_rng(seed + (881000 if rich else 880000))
  dim = int(rng.choice([1, 2, 2, 2, 3]))
  is_float = bool(rng.random() < 0.55)
  types = FLOAT_TYPES if is_float else INT_TYPES
  n_in = int(rng.choice([1, 1, 2]))
  n_out = n_in if rng.random() < 0.6 else int(rng.choice([1, 2]))
  in_types = [types[int(rng.integers(len(types)))] for _ in range(n_in)]
  iterable = n_in == n_out
  out_types = list(in_types) if iterable else \
      [types[int(rng.integers(len(types)))] for _ in range(n_out)]
  iterate = int(rng.choice([1, 2, 3])) if iterable else 1
  radius = 1 if dim == 3 else int(rng.choice([1, 2]))
  inputs = [('in%d' % i, t) for i, t in enumerate(in_types)]
  produced = list(inputs)
  stmts = []
  n_loc = int(rng.integers(0, 4))
  for k in range(n_loc + n_out):
    is_out = k >= n_loc
    name = 'out%d' % (k - n_loc) if is_out else 'loc%d' % k
    typ = out_types[k - n_loc] if is_out else \
        types[int(rng.integers(len(types)))]
    store = _idx(rng, dim, 1) if rng.random() < 0.3 else (0,) * dim
    parents = [produced[int(rng.integers(len(produced)))]
               for _ in range(int(rng.integers(1, 3)))]
    if is_out and k - n_loc < n_in and rng.random() < 0.5:
      parents.append(inputs[k - n_loc])

    def tap(_p=parents):
      n, t = _p[int(rng.integers(len(_p)))]
      return Ref(n, _idx(rng, dim, radius), t)

    leaves = [tap, tap]
    lets = []
    if rng.random() < 0.3:
      lets.append((typ, 'tmp', _arith(rng, tap, is_float, 1, rich,
                                      iterate == 1)))
      leaves.append(lambda _t=typ: Var('tmp', _t))
    if is_float:
      leaves.append(lambda _t=typ: Lit(
          '%.3ff' % rng.uniform(0.05, 1.0) if _t == 'float' or
          rng.random() < 0.5 else '%.3f' % rng.uniform(0.05, 1.0)))
    else:
      leaves.append(lambda: Lit(str(int(rng.integers(0, 50)))))

    def leaf(_l=leaves):
      return _l[int(rng.integers(len(_l)))]()

    expr = _arith(rng, leaf, is_float, 0, rich, iterate == 1)
    if not any(True for _ in expr.refs()):
      expr = Bin('+', expr, tap())             # every statement reads a tensor
    if is_float and typ == 'float' and expr.ctype() == 'float' and \
        rng.random() < 0.12:
      # sqrt only as a whole float statement: (float)sqrt((double)x) and
      # sqrtf(x) are the same number, whichever overload a host picks
      expr = Call('sqrt', [Bin('+', Lit('1.5f'), Bin('*', expr, expr))])
    stmts.append(Stmt('output' if is_out else 'local', typ, name, store, expr,
                      lets))
    if not is_out:
      produced.append((name, typ))
  prog = Program('%snest%d' % ('r' if rich else 'p', seed), dim, iterate,
                 inputs, stmts)
  rng2 = np.random.default_rng(seed + 9999)
  if dim == 1:
    extent = (int(rng2.integers(200, 700)),)
  elif dim == 2:
    extent = (int(rng2.choice([64, 100, 258, 300])), int(rng2.integers(24, 90)))
  else:
    extent = (int(rng2.choice([40, 64, 260])), int(rng2.integers(12, 20)),
              int(rng2.integers(14, 30)))
  return prog, extent


def _window_program(seed):
  rng = np.random.default_rng(seed + 882000)
  dim = int(rng.choice([2, 2, 2, 3]))
  typ = ['int16', 'uint16', 'uint8', 'int32'][int(rng.integers(4))]
  iterate = int(rng.choice([1, 1, 2, 3]))
  longest = 24 if dim == 2 else 8
  inputs = [('in0', typ)]
  produced = list(inputs)
  stmts = []
  n_stage = int(rng.integers(1, 4))
  for k in range(n_stage):
    pn, pt = produced[-1] if rng.random() < 0.7 else \
        produced[int(rng.integers(len(produced)))]
    op = ['+', 'min', 'max'][int(rng.integers(3))]
    d = int(rng.integers(dim))
    n = int(rng.integers(2, longest + 1))
    first = int(rng.integers(-n + 1, 2))
    base = [int(rng.integers(-1, 2)) if rng.random() < 0.2 else 0
            for _ in range(dim)]
    taps = []
    for j in range(n):
      idx = list(base)
      idx[d] = first + j
      taps.append(Ref(pn, idx, pt))
    if rng.random() < 0.15 and len(taps) > 2:      # not a contiguous run
      taps.pop(int(rng.integers(len(taps))))
    if rng.random() < 0.2:
      taps = [taps[int(i)] for i in rng.permutation(len(taps))]
    if op == '+':
      body = taps[0]
      for t in taps[1:]:
        body = Bin('+', body, t)
    else:
      body = Call(op, taps)       # one tensor: equal types by construction
    store = _idx(rng, dim, 1) if rng.random() < 0.3 else (0,) * dim
    last = k == n_stage - 1
    if last:
      en, et = produced[int(rng.integers(len(produced)))]
      r = rng.random()
      if r < 0.5:
        body = Bin('/', body, Lit(str(int(rng.integers(2, 9)))))
      if r < 0.3:
        body = Bin('+', body, Ref(en, _idx(rng, dim, 1), et))
      stmts.append(Stmt('output', typ, 'out0', store, body))
    else:
      lt = typ if rng.random() < 0.7 else \
          ['int16', 'uint16', 'uint8', 'int32'][int(rng.integers(4))]
      stmts.append(Stmt('local', lt, 'w%d' % k, store, body))
      produced.append(('w%d' % k, lt))
  prog = Program('wnest%d' % seed, dim, iterate, inputs, stmts)
  rng2 = np.random.default_rng(seed + 78000)
  if dim == 2:
    extent = (int(rng2.choice([128, 258, 300, 520])),
              int(rng2.integers(80, 200)))
  else:
    extent = (int(rng2.choice([64, 130, 260])), int(rng2.integers(24, 40)),
              int(rng2.integers(30, 60)))
  return prog, extent


def _wide_program(seed):
  """One input of 4-byte cells, one or two statements (a local read at the
  consumer's own cell, or none), 12-40 taps spread over a window up to 21 cells
  wide and 9 rows tall on both sides of the cell, mixed-precedence arithmetic
  with literals: the shape the `ldswin` kernels serve (an LDS row ring)."""
  rng = np.random.default_rng(seed + 883000)
  typ = ['float', 'float', 'int32'][int(rng.integers(3))]
  is_float = typ == 'float'
  xl, xh = -int(rng.integers(0, 11)), int(rng.integers(1, 11))
  yl, yh = -int(rng.integers(0, 5)), int(rng.integers(0, 5))
  inputs = [('a', typ)]

  def tap(name='a'):
    return Ref(name, (int(rng.integers(xl, xh + 1)),
                      int(rng.integers(yl, yh + 1))), typ)

  def lit():
    return Lit('%.3ff' % rng.uniform(0.05, 1.5) if is_float
               else str(int(rng.integers(1, 9))))

  def chain(n, leaf):
    node = leaf()
    for _ in range(n):
      r = rng.random()
      rhs = leaf()
      if r < 0.5:
        rhs = Bin('*', rhs, lit())
      elif r < 0.6:
        rhs = Bin('*', rhs, leaf())
      elif r < 0.65 and is_float:
        d = leaf()
        rhs = Bin('/', rhs, Bin('+', Lit('1.5f'), Bin('*', d, d)))
      elif r < 0.7:
        rhs = Call(['min', 'max'][int(rng.integers(2))],
                   _same_type([rhs, leaf()]))
      node = Bin(['+', '-', '+'][int(rng.integers(3))], node, rhs)
    return node

  stmts = []
  store = _idx(rng, 2, 1) if rng.random() < 0.3 else (0, 0)
  if rng.random() < 0.5:
    lstore = _idx(rng, 2, 1) if rng.random() < 0.3 else (0, 0)
    stmts.append(Stmt('local', typ, 's', lstore,
                      chain(int(rng.integers(6, 20)), tap)))
    # the consumer reads the local at its own store index only: the local is
    # "pointwise" (what it computed for the consumer's base cell)
    sref = lambda: Ref('s', lstore, typ)
    expr = Bin('+', Bin('*', sref(), lit()),
               chain(int(rng.integers(6, 20)), tap))
    if rng.random() < 0.5:
      expr = Bin('-', expr, sref())
  else:
    expr = chain(int(rng.integers(12, 40)), tap)
  stmts.append(Stmt('output', typ, 'b', store, expr))
  prog = Program('widenest%d' % seed, 2, 1, inputs, stmts)
  rng2 = np.random.default_rng(seed + 884000)
  extent = (int(rng2.choice([36, 520, 1028, 1540])), int(rng2.integers(40, 160)))
  return prog, extent


def inputs_for(prog, extent, seed):
  rng = np.random.default_rng(seed + 4242)
  shape = tuple(extent[::-1])
  out = {}
  for name, t in prog.inputs:
    dt = NPTYPES[t]
    if t in FLOAT_TYPES:
      out[name] = rng.uniform(0.25, 2.0, shape).astype(dt)
    else:
      out[name] = rng.integers(0, min(int(np.iinfo(dt).max), 200) + 1,
                               shape).astype(dt)
  return out


def has_empty_box(prog, extent):
  return any(any(h <= l for l, h in zip(lo, hi))
             for _, _, lo, hi in prog.boxes(extent))
