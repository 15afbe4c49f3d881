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


def program(seed: int):
  """Returns (soda text, dim, iterate)."""
  rng = np.random.default_rng(seed)
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
  radius = 1 if dim == 3 else int(rng.choice([1, 2]))
  tile = ['32'] * (dim - 1)
  lines = ['kernel: fuzz%d' % seed, 'burst width: 64', 'unroll factor: 2',
           'iterate: %d' % iterate]
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
      lets.append('  %s tmp = %s' % (lt, _expr(rng, [leaf], is_float, 1)))
      leaves.append(lambda: 'tmp')
    if is_float:
      leaves.append(lambda: '%.3ff' % rng.uniform(-1.0, 1.0)
                    if t == 'float' else '%.3f' % rng.uniform(-1.0, 1.0))
    else:
      leaves.append(lambda: str(int(rng.integers(0, 50))))
    body = _expr(rng, leaves, is_float)
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
