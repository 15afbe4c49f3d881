"""CPU ORACLE (numpy restatement) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path (soda_amd/) never does and fails loudly when
its HIP library is missing.

Restates the reference's semantic definition of a SODA program's result: the
naive loop nest its host generator prints for self-checking
(reference src/soda/codegen/frt/host.py:558-624), with

  * the valid box of every tensor from the input-relative overall stencil
    window (frt/host.py:565-577; core.py:876-926) -- note `tensor.is_output`
    at host.py:567 is a bound method, always truthy, so EVERY tensor uses the
    window relative to the program inputs;
  * loads addressed as parent[x + idx - st_idx] (frt/host.py:587-594);
  * iteration chaining output_k -> input_k of the next iteration
    (core.py:320-336);
  * C++ expression semantics: integer promotion, usual arithmetic
    conversions, truncating division, fp32 ops in textual order without FMA,
    conversion to the statement type on store (grammar.py:123-136);
  * zero-initialised tensors (frt/host.py:472-475): cells outside a tensor's
    box are 0.

PARITY STATUS: the reference's own tests hold no numeric vector for this path
 and the reference cannot be built or imported here
(haoda/textx/pulp missing, no Xilinx headers), so NUMERIC PARITY IS UNPINNED by
reference fixtures.  What pins this file instead: closed-form KATs derived from
the semantics (tests/test_oracle.py), agreement with the independently written
C loop nests in oracle/kat_kernels.c, and agreement with the generated-C oracle
(oracle/c_oracle.py).

Array convention: numpy shape is extent reversed (dimension 0 of the DSL is the
fastest-varying one = the last numpy axis), C-contiguous.
"""
import os
import re
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from soda_amd import core, ir, util

# ---------------------------------------------------------------------------
# C scalar type lattice (LP64)
# ---------------------------------------------------------------------------

_NP = {
    'i8': np.int8, 'u8': np.uint8, 'i16': np.int16, 'u16': np.uint16,
    'i32': np.int32, 'u32': np.uint32, 'i64': np.int64, 'u64': np.uint64,
    'f32': np.float32, 'f64': np.float64,
}
_RANK = {'i8': 1, 'u8': 1, 'i16': 2, 'u16': 2, 'i32': 3, 'u32': 3, 'i64': 4,
         'u64': 4}


def ctype_of(t: ir.Type) -> str:
  c = t.c_type
  if c == 'float':
    return 'f32'
  if c == 'double':
    return 'f64'
  return ('u' if c.startswith('u') else 'i') + c.lstrip('u')[3:-2]


def _promote(t: str) -> str:
  """Integer promotion: everything narrower than int becomes int."""
  if t in _RANK and _RANK[t] < 3:
    return 'i32'
  return t


def _usual(a: str, b: str) -> str:
  """Usual arithmetic conversions."""
  if 'f64' in (a, b):
    return 'f64'
  if 'f32' in (a, b):
    return 'f32'
  a, b = _promote(a), _promote(b)
  if a == b:
    return a
  sa, sb = a[0] == 'i', b[0] == 'i'
  if sa == sb:
    return a if _RANK[a] >= _RANK[b] else b
  u, s = (b, a) if sa else (a, b)
  if _RANK[u] >= _RANK[s]:
    return u
  return s  # i64 holds every u32


def _conv(x, src: str, dst: str):
  """C conversion of values of type src to dst."""
  if src == dst:
    return x
  if dst in ('f32', 'f64'):
    return np.asarray(x).astype(_NP[dst])
  if src in ('f32', 'f64'):
    # float -> int truncates toward zero (in range); go through int64
    wide = np.trunc(np.asarray(x, dtype=np.float64)).astype(np.int64)
    return wide.astype(_NP[dst])
  return np.asarray(x).astype(_NP[dst])  # modular, like every real compiler


def _literal(num: ir.Num):
  text = num.literal.lower()
  if num.is_float_literal:
    if text.endswith('f'):
      return np.float32(text[:-1]), 'f32'
    return np.float64(text.rstrip('l')), 'f64'
  digits = text.rstrip('ul')
  suffix = text[len(digits):]
  value = int(digits, 0) if not digits.startswith('0b') else int(digits[2:], 2)
  unsigned = 'u' in suffix
  long_ = 'l' in suffix
  if unsigned:
    t = 'u64' if long_ or value >= 2**32 else 'u32'
  else:
    t = 'i64' if long_ or value >= 2**31 else 'i32'
  return _NP[t](value), t


def _trunc_div(a, b, t):
  if t in ('f32', 'f64'):
    return a / b
  if t[0] == 'u':
    return a // b
  q = a // b
  fix = ((a % b) != 0) & ((a < 0) != (b < 0))
  return (q + fix.astype(q.dtype)).astype(_NP[t])


def _trunc_mod(a, b, t):
  if t[0] == 'u':
    return a % b
  return (a - _trunc_div(a, b, t) * b).astype(_NP[t])


class _Eval:
  """Evaluates one statement's expression over a box with C semantics."""

  def __init__(self, load, let_values):
    self.load = load          # Ref -> (array, ctype)
    self.lets = let_values    # name -> (array, ctype)

  def __call__(self, node):
    with np.errstate(all='ignore'):
      return self.ev(node)

  def ev(self, n):
    if isinstance(n, ir.Num):
      return _literal(n)
    if isinstance(n, ir.Ref):
      return self.load(n)
    if isinstance(n, ir.Var):
      if n.name in self.lets and not n.idx:
        return self.lets[n.name]
      raise util.SemanticError('oracle: unknown variable %s' % n)
    if isinstance(n, ir.Cast):
      v, t = self.ev(n.expr)
      dst = ctype_of(n.haoda_type)
      return _conv(v, t, dst), dst
    if isinstance(n, ir.Unary):
      v, t = self.ev(n.operand)
      for op in reversed(n.ops):
        if op == '!':
          v, t = (v == 0).astype(np.int32), 'i32'
          continue
        p = _promote(t)
        v = _conv(v, t, p)
        t = p
        if op == '-':
          v = (-v).astype(_NP[t]) if t not in ('f32', 'f64') else -v
        elif op == '~':
          v = ~v
      return v, t
    if isinstance(n, ir.Chain):
      v, t = self.ev(n.operands[0])
      for op, operand in zip(n.operators, n.operands[1:]):
        w, u = self.ev(operand)
        v, t = self.binop(op, v, t, w, u)
      return v, t
    if isinstance(n, ir.Call):
      return self.call(n)
    raise util.InternalError('oracle cannot evaluate %r' % (n,))

  def binop(self, op, a, ta, b, tb):
    if op in ('&&', '||'):
      x, y = (a != 0), (b != 0)
      r = (x & y) if op == '&&' else (x | y)
      return r.astype(np.int32), 'i32'
    t = _usual(ta, tb)
    a, b = _conv(a, ta, t), _conv(b, tb, t)
    if op in ('==', '!=', '<', '>', '<=', '>='):
      r = {'==': np.equal, '!=': np.not_equal, '<': np.less, '>': np.greater,
           '<=': np.less_equal, '>=': np.greater_equal}[op](a, b)
      return r.astype(np.int32), 'i32'
    dt = _NP[t]
    if op == '+':
      r = np.add(a, b, dtype=dt)
    elif op == '-':
      r = np.subtract(a, b, dtype=dt)
    elif op == '*':
      r = np.multiply(a, b, dtype=dt)
    elif op == '/':
      r = _trunc_div(np.asarray(a), np.asarray(b), t)
    elif op == '%':
      r = _trunc_mod(np.asarray(a), np.asarray(b), t)
    elif op == '&':
      r = a & b
    elif op == '|':
      r = a | b
    elif op == '^':
      r = a ^ b
    else:
      raise util.InternalError('operator %s' % op)
    return np.asarray(r).astype(dt, copy=False), t

  def call(self, n):
    args = [self.ev(a) for a in n.args]
    name = n.name
    if name in ('min', 'max', 'fmin', 'fmax'):
      v, t = args[0]
      for w, u in args[1:]:
        c = _usual(t, u)
        a, b = _conv(v, t, c), _conv(w, u, c)
        # SODA_MIN(a,b) = (b < a ? b : a); SODA_MAX(a,b) = (a < b ? b : a)
        v = np.where(b < a, b, a) if name in ('min', 'fmin') else np.where(
            a < b, b, a)
        t = c
      return v, t
    if name == 'abs':
      v, t = args[0]
      p = _promote(t)
      v = _conv(v, t, p)
      return np.where(v < 0, -v, v).astype(_NP[p]), p
    if name == 'select':
      (c, _), (a, ta), (b, tb) = args
      t = _usual(ta, tb)
      return np.where(c != 0, _conv(a, ta, t), _conv(b, tb, t)), t
    fn = {'sqrt': np.sqrt, 'cbrt': np.cbrt, 'exp': np.exp, 'exp2': np.exp2,
          'log': np.log, 'log2': np.log2, 'log10': np.log10, 'sin': np.sin,
          'cos': np.cos, 'tan': np.tan, 'asin': np.arcsin, 'acos': np.arccos,
          'atan': np.arctan, 'floor': np.floor, 'ceil': np.ceil,
          'fabs': np.fabs, 'round': None, 'pow': np.power}[name]
    # which C function is called follows the DSL type of the call
    # (ir.c_expr): float -> sqrtf, otherwise the double function
    dsl_t = n.haoda_type
    t = 'f32' if (dsl_t is not None and dsl_t.is_float and
                  dsl_t.width_in_bits <= 32) else 'f64'
    vals = [_conv(v, u, t) for v, u in args]
    if name == 'round':
      v = vals[0]
      r = np.where(v < 0, np.ceil(v - 0.5), np.floor(v + 0.5))
    else:
      r = fn(*vals)
    return np.asarray(r).astype(_NP[t]), t


# ---------------------------------------------------------------------------
# the loop nest
# ---------------------------------------------------------------------------

def _box_slices(lo: Sequence[int], hi: Sequence[int], off: Sequence[int]):
  """numpy index of box [lo, hi) shifted by `off` (axes reversed)."""
  return tuple(slice(l + o, h + o)
               for l, h, o in zip(lo[::-1], hi[::-1], off[::-1]))


def run(stencil: core.Stencil, inputs: Dict[str, np.ndarray],
        iterate: Optional[int] = None,
        keep_locals: bool = False,
        origin: Optional[Sequence[int]] = None,
        global_extent: Optional[Sequence[int]] = None
        ) -> Dict[str, np.ndarray]:
  """Runs `iterate` iterations (default: the program's) and returns the output
  tensors (plus the last iteration's locals if asked), zero outside their
  valid boxes."""
  iterate = stencil.iterate if iterate is None else iterate
  if iterate > 1 and len(stencil.input_names) != len(stencil.output_names):
    raise util.SemanticError('iterate > 1 needs as many outputs as inputs')
  # `param` arrays ride in `inputs` under their names, C order (element
  # name(i, j) = array[i][j]); a scalar param is a 0-d or 1-element array
  params = {}
  for p in stencil.param_stmts:
    arr = np.ascontiguousarray(inputs[p.name]).reshape(-1)
    if arr.dtype != np.dtype(p.haoda_type.np_name) or \
        arr.size != stencil.param_elems(p):
      raise util.InputError('param %s must be %d x %s' % (
          p.name, stencil.param_elems(p), p.haoda_type.np_name))
    params[p.name] = (arr, p)
  first = inputs[stencil.input_names[0]]
  extent = first.shape[::-1]
  dim = stencil.dim
  if len(extent) != dim:
    raise util.InputError('inputs must be %d-dimensional' % dim)
  cur = {}
  for name, t in zip(stencil.input_names, stencil.input_types):
    arr = np.ascontiguousarray(inputs[name])
    if arr.shape != first.shape:
      raise util.InputError('input shapes differ')
    if arr.dtype != np.dtype(t.np_name):
      raise util.InputError('input %s must be %s' % (name, t.np_name))
    cur[name] = arr

  preserve = stencil.preserve_border
  stencil.check_preserve()
  in_boxes = None
  result = {}
  for it in range(iterate):
    # border: preserve -- every iteration starts from fully defined inputs
    boxes = stencil.iteration_boxes(None if preserve else in_boxes)
    tensors = dict(cur)
    for stage in stencil.ordered_stages:
      wlo, whi = boxes[stage.name]
      lo = tuple(max(0, -l) for l in wlo)
      hi = tuple(n - max(0, h) for n, h in zip(extent, whi))
      out = np.zeros(first.shape, dtype=np.dtype(stage.haoda_type.np_name))
      if preserve and stage.is_output:
        out[...] = cur[stencil.preserved_from(stage.name)]
        if origin is not None:
          # the arrays are a window (slab) of a larger grid: the border is the
          # GLOBAL one; cells the window cannot compute keep the input too
          glo, ghi = stencil.interior_box(global_extent, stage.name)
          tl = [0] * dim
          th = [0] * dim
          for parent in stage.taps:
            a, b = stage.tap_bounds(parent)
            for d in range(dim):
              tl[d] = max(tl[d], -a[d])
              th[d] = max(th[d], b[d])
          lo = tuple(max(tl[d], glo[d] - origin[d]) for d in range(dim))
          hi = tuple(min(extent[d] - th[d], ghi[d] - origin[d])
                     for d in range(dim))
      if all(h > l for l, h in zip(lo, hi)):
        st = stage.st_idx

        def load(ref, _lo=lo, _hi=hi, _st=st):
          if ref.name in params:
            arr, pstmt = params[ref.name]
            return (arr[stencil.param_index(pstmt, ref.idx)],
                    ctype_of(pstmt.haoda_type))
          off = tuple(a - b for a, b in zip(ref.idx, _st))
          parent = tensors[ref.name]
          return (parent[_box_slices(_lo, _hi, off)],
                  ctype_of(stencil.symbol_table[ref.name]))

        lets = {name: (arr[0], ctype_of(pstmt.haoda_type))
                for name, (arr, pstmt) in params.items() if not pstmt.size}
        ev = _Eval(load, lets)
        for let in stage.stmt.let:
          v, t = ev(let.expr)
          if let.haoda_type is not None:
            dst = ctype_of(let.haoda_type)
            v, t = _conv(v, t, dst), dst
          lets[let.name] = (v, t)
        v, t = ev(stage.stmt.expr)
        dst = ctype_of(stage.haoda_type)
        out[_box_slices(lo, hi, (0,) * dim)] = _conv(v, t, dst)
      tensors[stage.name] = out
    if it < iterate - 1:
      cur = {i: tensors[o]
             for i, o in zip(stencil.input_names, stencil.output_names)}
      in_boxes = {i: boxes[o]
                  for i, o in zip(stencil.input_names, stencil.output_names)}
    else:
      for name in stencil.output_names:
        result[name] = tensors[name]
      if keep_locals:
        for name in stencil.local_names:
          result[name] = tensors[name]
  return result


# ---------------------------------------------------------------------------
# the reference harness's compare rule (frt/host.py:625-657)
# ---------------------------------------------------------------------------

def _atof(text: str) -> float:
  """C's atof: the longest leading floating-point literal, 0.0 if none."""
  m = re.match(r'\s*[+-]?(\d+\.?\d*([eE][+-]?\d+)?|\.\d+([eE][+-]?\d+)?'
               r'|inf(inity)?|nan)', text, re.I)
  return float(m.group(0)) if m else 0.0


def compare(got: np.ndarray, want: np.ndarray, lo: Sequence[int],
            hi: Sequence[int], threshold: Optional[float] = None) -> int:
  """Number of mismatching cells inside box [lo, hi): integers must be equal;
  floats fail iff (d^2 > t^2) and (d^2 / ref^2 > t^2), t = 0.00001 unless the
  environment says otherwise -- `$THRESHOLD`, read with atof() per compare as
  the reference's generated host does (frt/host.py:634-637); an explicit
  `threshold` argument outranks both."""
  if threshold is None:
    threshold = 1e-5
    env = os.environ.get('THRESHOLD')
    if env is not None:
      threshold = _atof(env)
  idx = _box_slices(lo, hi, (0,) * len(lo))
  g, w = got[idx], want[idx]
  if np.issubdtype(w.dtype, np.floating):
    d = g.astype(np.float64) - w.astype(np.float64)
    d2 = d * d
    t2 = threshold * threshold
    with np.errstate(all='ignore'):
      bad = (d2 > t2) & (d2 / (w.astype(np.float64)**2) > t2)
    bad |= np.isnan(d)
    return int(bad.sum())
  return int((g != w).sum())
