"""Random .soda programs for property tests (shared by CPU and GPU suites).

Generates small but structurally varied programs: 1-3 dimensions, 1-2 inputs,
0-3 locals, 1-2 outputs, random taps within a small radius (asymmetric, with
non-zero store indices), integer or floating types, let variables, nested
parentheses, the function/operator set both the oracle and the kernels
implement.  Everything is seeded, so a failing case is reproducible from its
seed."""
import numpy as np

FLOAT_TYPES = ['float', 'double']
INT_TYPES = ['uint8', 'int16', 'uint16', 'int32']


_BUDGET = None


def _cache_is_warm() -> bool:
  """Was the JIT cache (soda_amd/_jit_cache) filled by a run of the FULL seed
  sets (tools/warm_jit_cache.sh)?  Asked precisely: the code object of one
  program only the full sets contain -- the last valid plain seed below 100 --
  lowered as test_gpu_matches_oracle lowers it, is looked up by its content
  key.  hiprtc takes 1-3 s per module; with the cache the full sets cost the
  GPU run ~1 minute more, without it ~5."""
  import os
  try:
    from soda_amd import core, runtime, util
    from soda_amd.codegen.hip import lower
    for seed in range(99, 49, -1):
      text, dim, _ = program(seed)
      try:
        stencil = core.from_text(text)
      except util.SodaError:
        continue
      extent = extent_for(seed, dim)
      lo, hi = stencil.valid_box(extent)
      if not all(h > l for l, h in zip(lo, hi)):
        continue
      opts = runtime.resolve_options(
          stencil, lower.LowerOptions(strategy='direct', fuse=(2,)), extent)
      mod = lower.lower(stencil, opts)
      key = runtime.source_key(mod.source)
      name = ('%s.hip' % stencil.app_name).replace('.', '_')
      return os.path.exists(os.path.join(runtime.CACHE_DIR,
                                         '%s_%s.hsaco' % (name, key)))
  except Exception:      # noqa: BLE001 -- no library, no cache: the default sets
    return False
  return False


def budget() -> float:
  """`--fuzz-budget` / $SODA_FUZZ_BUDGET: how much of the GPU run's time the
  random-program tests may take, relative to the sets sized for a COLD run
  (1).  Not given: 2 -- the full sets of rounds 2-3 -- when the JIT cache holds
  their code objects already, else 1."""
  global _BUDGET
  import os
  given = os.environ.get('SODA_FUZZ_BUDGET')
  if given:
    try:
      return max(0.0, float(given))
    except ValueError:
      return 1.0
  if _BUDGET is None:
    _BUDGET = 2.0 if _cache_is_warm() else 1.0
  return _BUDGET


def budget_seeds(default: int, full: int, pinned=()):
  """Seeds 0 .. n-1 with n = `default` scaled by the budget (never more than
  `full`, the set earlier rounds ran; never fewer than 4), plus `pinned`
  (seeds that once found a defect stay in whatever the budget)."""
  n = min(full, max(4, int(round(default * budget()))))
  return list(range(n)) + [s for s in pinned if s >= n]


def _idx(rng, dim, radius):
  return tuple(int(rng.integers(-radius, radius + 1)) for _ in range(dim))


def _ref(name, idx):
  return '%s(%s)' % (name, ', '.join(map(str, idx)))


def _expr(rng, leaves, is_float, depth=0):
  """Random expression over `leaves` (callables returning a leaf string)."""
  r = rng.random()
  if depth >= 3 or r < 0.25:
    return leaves[int(rng.integers(len(leaves)))]()
  if r < 0.80:
    n = int(rng.integers(2, 5))
    ops = ['+', '-', '*'] if is_float else ['+', '-', '*', '+', '-']
    parts = [_expr(rng, leaves, is_float, depth + 1)]
    for _ in range(n - 1):
      parts.append(ops[int(rng.integers(len(ops)))])
      parts.append(_expr(rng, leaves, is_float, depth + 1))
    return '(' + ' '.join(parts) + ')'
  if r < 0.88:
    lit = ('%.3ff' % rng.uniform(0.1, 2.0)) if is_float else str(
        int(rng.integers(1, 9)))
    return '(%s * %s)' % (_expr(rng, leaves, is_float, depth + 1), lit)
  if r < 0.94:
    fn = ['min', 'max'][int(rng.integers(2))]
    return '%s(%s, %s)' % (fn, _expr(rng, leaves, is_float, depth + 1),
                           _expr(rng, leaves, is_float, depth + 1))
  if is_float:
    return 'sqrt(1.5f + %s * %s)' % ((_expr(rng, leaves, True, depth + 1),) * 2)
  return '(%s / %d)' % (_expr(rng, leaves, False, depth + 1),
                        int(rng.integers(2, 7)))


def _expr_rich(rng, leaves, is_float, depth=0, to_int=True):
  """As _expr, plus the rest of the operator set whose results are defined bit
  for bit on both sides: float division (IEEE), unary minus, abs / fabs,
  floor / ceil, select over comparisons and logic, casts between the float
  types and to integers and back (`to_int`), integer %, &, |, ^ and 64-bit
  detours."""
  sub = lambda: _expr_rich(rng, leaves, is_float, depth + 1, to_int)
  r = rng.random()
  if depth >= 3 or r < 0.22:
    return leaves[int(rng.integers(len(leaves)))]()
  if r < 0.55:
    return _expr(rng, [sub], is_float, 2 if depth >= 2 else 1)
  if r < 0.63:
    cmp_op = ['<', '<=', '>', '>=', '==', '!='][int(rng.integers(6))]
    cond = '%s %s %s' % (sub(), cmp_op, sub())
    if rng.random() < 0.3:
      cond = '(%s) %s (%s %s %s)' % (cond, ['&&', '||'][int(rng.integers(2))],
                                     sub(), ['<', '>'][int(rng.integers(2))],
                                     sub())
    return 'select(%s, %s, %s)' % (cond, sub(), sub())
  if r < 0.70:
    return '(-%s)' % sub()
  if r < 0.76:
    return '%s(%s)' % ('abs' if not is_float or rng.random() < 0.5 else 'fabs',
                       sub())
  if is_float:
    if r < 0.86:
      d = sub()
      return '(%s / (1.5f + %s * %s))' % (sub(), d, d)
    if r < 0.91:
      return '%s(%s * 3.7f)' % (['floor', 'ceil'][int(rng.integers(2))], sub())
    if r < 0.96:
      return '%s(%s)' % (['float', 'double'][int(rng.integers(2))], sub())
    if not to_int:     # (out of int32's range the conversion is undefined in
      return sub()     # C: iterated programs, whose values grow, do without)
    return 'float(int32(%s * 5.0f))' % sub()
  if r < 0.84:
    return '(%s %% %d)' % (sub(), int(rng.integers(2, 9)))
  if r < 0.93:
    return '(%s %s %s)' % (sub(), ['&', '|', '^'][int(rng.integers(3))], sub())
  return 'int32(int64(%s) * 100003 %% 1009)' % sub()


def program(seed: int, rich: bool = False):
  """Returns (soda text, dim, iterate).  `rich`: the wider operator set of
  _expr_rich (its own seeds: the programs of the plain generator never
  change)."""
  rng = np.random.default_rng(seed + 555000 if rich else seed)
  _gen = _expr
  dim = int(rng.choice([1, 2, 2, 2, 3]))
  is_float = bool(rng.random() < 0.6)
  types = FLOAT_TYPES if is_float else INT_TYPES
  n_in = int(rng.choice([1, 1, 2]))
  n_out = n_in if rng.random() < 0.6 else int(rng.choice([1, 2]))
  iterable = n_in == n_out
  in_types = [types[int(rng.integers(len(types)))] for _ in range(n_in)]
  if iterable:
    out_types = list(in_types)
  else:
    out_types = [types[int(rng.integers(len(types)))] for _ in range(n_out)]
  iterate = int(rng.choice([1, 2, 3])) if iterable else 1
  if rich:
    def _gen(r, leaves, flt, depth=0, _to_int=iterate == 1):
      return _expr_rich(r, leaves, flt, depth, _to_int)
  radius = 1 if dim == 3 else int(rng.choice([1, 2]))
  tile = ['32'] * (dim - 1)
  lines = ['kernel: %sfuzz%d' % ('r' if rich else '', seed),
           'burst width: 64', 'unroll factor: 2', 'iterate: %d' % iterate]
  names = []
  for i, t in enumerate(in_types):
    decl = 'in%d' % i
    if dim > 1 and i == 0:
      decl += '(%s, *)' % ', '.join(tile)
    lines.append('input %s: %s' % (t, decl))
    names.append('in%d' % i)
  n_loc = int(rng.integers(0, 4))
  produced = list(names)
  for k in range(n_loc + n_out):
    is_out = k >= n_loc
    name = ('out%d' % (k - n_loc)) if is_out else ('loc%d' % k)
    t = out_types[k - n_loc] if is_out else types[int(rng.integers(len(types)))]
    st_idx = _idx(rng, dim, 1) if rng.random() < 0.3 else (0,) * dim
    parents = [produced[int(rng.integers(len(produced)))]
               for _ in range(int(rng.integers(1, 3)))]
    if is_out and k - n_loc < n_in and ('in%d' % (k - n_loc)) not in parents \
        and rng.random() < 0.5:
      parents.append('in%d' % (k - n_loc))

    def leaf(_parents=parents):
      p = _parents[int(rng.integers(len(_parents)))]
      return _ref(p, _idx(rng, dim, radius))

    leaves = [leaf]
    lets = []
    if rng.random() < 0.3:
      lt = t
      lets.append('  %s tmp = %s' % (lt, _gen(rng, [leaf], is_float, 1)))
      leaves.append(lambda: 'tmp')
    if is_float:
      leaves.append(lambda: '%.3ff' % rng.uniform(-1.0, 1.0)
                    if t == 'float' else '%.3f' % rng.uniform(-1.0, 1.0))
    else:
      leaves.append(lambda: str(int(rng.integers(0, 50))))
    body = _gen(rng, leaves, is_float)
    head = ('output %s:' if is_out else 'local %s:') % t
    if lets:
      lines.append(head)
      lines.extend(lets)
      lines.append('  %s = %s' % (_ref(name, st_idx), body))
    else:
      lines.append('%s %s = %s' % (head, _ref(name, st_idx), body))
    if not is_out:
      produced.append(name)
  return '\n'.join(lines) + '\n', dim, iterate


def extent_for(seed: int, dim: int):
  rng = np.random.default_rng(seed + 9999)
  if dim == 1:
    return (int(rng.integers(200, 700)),)
  if dim == 2:
    return (int(rng.choice([64, 100, 258, 300])), int(rng.integers(20, 90)))
  return (int(rng.choice([40, 64, 260])), int(rng.integers(10, 20)),
          int(rng.integers(12, 30)))


def inputs_for(stencil, extent, seed: int):
  rng = np.random.default_rng(seed + 4242)
  shape = tuple(extent[::-1])
  out = {}
  for name, t in zip(stencil.input_names, stencil.input_types):
    dt = np.dtype(t.np_name)
    if t.is_float:
      out[name] = rng.uniform(0.25, 2.0, shape).astype(dt)
    else:
      hi = min(int(np.iinfo(dt).max), 200)
      out[name] = rng.integers(0, hi + 1, shape).astype(dt)
  return out


WINDOW_TYPES = ['int16', 'uint16', 'uint8', 'int32']


def window_program(seed: int):
  """Random programs made of integer window reductions (sums, min, max over
  2-24 contiguous taps along one dimension, chained 1-3 stages deep, random
  first offsets and store indices, narrow types whose sums wrap in the cast):
  the shapes the backend evaluates as sliding sums, power-of-two chains and
  joint per-lane windows instead of tap by tap.  Returns (text, dim, iterate)."""
  rng = np.random.default_rng(seed + 77000)
  dim = int(rng.choice([2, 2, 2, 3]))
  t = WINDOW_TYPES[int(rng.integers(len(WINDOW_TYPES)))]
  iterate = int(rng.choice([1, 1, 2, 3]))
  longest = 24 if dim == 2 else 8
  lines = ['kernel: wfuzz%d' % seed, 'burst width: 64', 'unroll factor: 2',
           'iterate: %d' % iterate,
           'input %s: in0(%s, *)' % (t, ', '.join(['32'] * (dim - 1)))]
  n_stage = int(rng.integers(1, 4))
  produced = ['in0']
  for k in range(n_stage):
    parent = produced[-1] if rng.random() < 0.7 else \
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
      taps.append(_ref(parent, idx))
    if rng.random() < 0.15:           # not a contiguous run: left as written
      taps.pop(int(rng.integers(len(taps))))
    if rng.random() < 0.2:
      order = rng.permutation(len(taps))
      taps = [taps[int(i)] for i in order]
    body = ' + '.join(taps) if op == '+' else '%s(%s)' % (op, ', '.join(taps)) \
        if len(taps) > 1 else taps[0]
    st_idx = _idx(rng, dim, 1) if rng.random() < 0.3 else (0,) * dim
    last = k == n_stage - 1
    if last:
      extra = produced[int(rng.integers(len(produced)))]
      r = rng.random()
      if r < 0.3:
        body = '(%s) / %d + %s' % (body, int(rng.integers(2, 9)),
                                   _ref(extra, _idx(rng, dim, 1)))
      elif r < 0.5:
        body = '(%s) / %d' % (body, int(rng.integers(2, 9)))
      lines.append('output %s: out0%s = %s' %
                   (t, '(%s)' % ', '.join(map(str, st_idx)), body))
    else:
      name = 'w%d' % k
      lt = t if rng.random() < 0.7 else \
          WINDOW_TYPES[int(rng.integers(len(WINDOW_TYPES)))]
      lines.append('local %s: %s = %s' % (lt, _ref(name, st_idx), body))
      produced.append(name)
  return '\n'.join(lines) + '\n', dim, iterate


def window_extent_for(seed: int, dim: int):
  rng = np.random.default_rng(seed + 78000)
  if dim == 2:
    return (int(rng.choice([128, 258, 300, 520])), int(rng.integers(80, 200)))
  return (int(rng.choice([64, 130, 260])), int(rng.integers(24, 40)),
          int(rng.integers(30, 60)))
