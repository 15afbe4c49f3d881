"""Pins the CPU oracle (test infrastructure under oracle/).

The reference holds no numeric vectors for this path, so the
oracle is pinned by: closed-form known answers derived from the reference
semantics (frt/host.py:558-624), three restatements that must agree bit for
bit (numpy, generated C, hand-written C that never touches this repo's
parser), the reference harness's input recipe, and committed golden outputs.
"""
import ctypes
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT, soda_path
from soda_amd import core


@pytest.fixture(scope='module')
def kat(built):
  lib = ctypes.CDLL(os.path.join(ROOT, 'oracle', '_build', 'libkat_kernels.so'))
  return lib


@pytest.fixture(scope='module')
def ref_init(built):
  return ctypes.CDLL(os.path.join(ROOT, 'oracle', '_build', 'libref_init.so'))


def _ptr(a):
  return a.ctypes.data_as(ctypes.c_void_p)


def _reference_floats(ref_init, shape):
  a = np.empty(shape, np.float32)
  ref_init.ref_init_float(_ptr(a), ctypes.c_int64(a.size))
  return a


def test_reference_input_recipe(ref_init):
  """libstdc++ default_random_engine + uniform_real_distribution<double>(0,1),
  memory order (frt/host.py:503-528): minstd_rand0, two draws per double."""
  a = _reference_floats(ref_init, (4, 5))
  x, vals = 1, []
  for _ in range(4):
    lo = x = x * 16807 % 2147483647
    hi = x = x * 16807 % 2147483647
    vals.append(((lo - 1) + (hi - 1) * 2147483646.0) / 2147483646.0**2)
  assert np.array_equal(a.ravel()[:4], np.array(vals).astype(np.float32))
  assert a.min() >= 0.0 and a.max() < 1.0


@pytest.mark.parametrize('iterate', [1, 2, 5])
def test_jacobi2d_three_restatements_agree(kat, ref_init, iterate):
  from oracle import c_oracle, numpy_oracle
  st = core.from_file(soda_path('jacobi2d.soda'))
  a = _reference_floats(ref_init, (40, 48))
  want = np.empty_like(a)
  assert kat.kat_jacobi2d(_ptr(a), _ptr(want), 48, 40, iterate) == 0
  assert np.array_equal(numpy_oracle.run(st, {'t1': a}, iterate)['t0'], want)
  assert np.array_equal(c_oracle.COracle(st).run({'t1': a}, iterate)['t0'], want)
  lo, hi = st.valid_box((48, 40), iterate=iterate)
  assert lo == (iterate, iterate) and hi == (48 - iterate, 40 - iterate)
  assert not want[:iterate].any() and not want[:, :iterate].any()


def test_blur_three_restatements_agree(kat):
  from oracle import c_oracle, numpy_oracle
  st = core.from_file(soda_path('blur.soda'))
  a = np.random.default_rng(0).integers(0, 65536, (30, 50)).astype(np.uint16)
  want = np.empty_like(a)
  assert kat.kat_blur(_ptr(a), _ptr(want), 50, 30) == 0
  assert np.array_equal(numpy_oracle.run(st, {'input': a})['blur_y'], want)
  assert np.array_equal(c_oracle.COracle(st).run({'input': a})['blur_y'], want)


@pytest.mark.parametrize('iterate', [1, 2, 3])
def test_heat3d_three_restatements_agree(kat, iterate):
  from oracle import c_oracle, numpy_oracle
  st = core.from_file(soda_path('heat3d.soda'))
  a = np.random.default_rng(2).random((12, 14, 16), dtype=np.float32)
  want = np.empty_like(a)
  assert kat.kat_heat3d(_ptr(a), _ptr(want), 16, 14, 12, iterate) == 0
  assert np.array_equal(numpy_oracle.run(st, {'in': a}, iterate)['out'], want)
  assert np.array_equal(c_oracle.COracle(st).run({'in': a}, iterate)['out'], want)


def test_skew2d_store_index_and_asymmetric_taps(kat):
  """Non-zero store index + asymmetric taps: orientation errors that the
  symmetric benchmark stencils cannot show."""
  from oracle import c_oracle, numpy_oracle
  st = core.from_file(soda_path('skew2d.soda'))
  a = np.random.default_rng(5).random((20, 24), dtype=np.float32)
  want = np.empty_like(a)
  assert kat.kat_skew2d(_ptr(a), _ptr(want), 24, 20) == 0
  assert np.array_equal(numpy_oracle.run(st, {'a': a})['c'], want)
  assert np.array_equal(c_oracle.COracle(st).run({'a': a})['c'], want)
  assert st.valid_box((24, 20)) == ((2, 2), (22, 18))


def test_blur_closed_form_on_reference_init():
  """BASELINE config 1 (tests/src/blur.soda, 2000 x 1024, CPU): the reference
  harness's integer init p+q gives blur_x = p+q+1 and blur_y = p+q+2 exactly
  on [0,1998) x [0,1022), zero elsewhere."""
  from oracle import c_oracle, numpy_oracle
  st = core.from_file(soda_path('blur.soda'))
  q, p = np.indices((1024, 2000))
  a = (p + q).astype(np.uint16)
  for got in (numpy_oracle.run(st, {'input': a}, keep_locals=True),
              c_oracle.COracle(st).run({'input': a})):
    assert np.array_equal(got['blur_y'][:1022, :1998], (p + q + 2)[:1022, :1998])
    assert not got['blur_y'][1022:].any() and not got['blur_y'][:, 1998:].any()
    if 'blur_x' in got:
      assert np.array_equal(got['blur_x'][:1022], (p + q + 1)[:1022])


@pytest.mark.parametrize('iterate', [1, 4])
def test_heat3d_ramp_is_a_fixed_point(iterate):
  """Coefficients are powers of two summing to 1: p+q+r is reproduced exactly
  (robust to FMA/association; pins 3-D indexing and the box shrink)."""
  from oracle import numpy_oracle
  st = core.from_file(soda_path('heat3d.soda'))
  a = np.indices((20, 18, 16)).sum(axis=0).astype(np.float32)
  got = numpy_oracle.run(st, {'in': a}, iterate)['out']
  n = iterate
  assert np.array_equal(got[n:-n, n:-n, n:-n], a[n:-n, n:-n, n:-n])
  assert not got[:n].any()


def test_jacobi2d_linear_field_within_tolerance():
  from oracle import numpy_oracle
  st = core.from_file(soda_path('jacobi2d.soda'))
  q, p = np.indices((40, 50))
  a = (0.25 * p + 0.5 * q + 1).astype(np.float32)
  got = numpy_oracle.run(st, {'t1': a}, 6)['t0']
  lo, hi = st.valid_box((50, 40), iterate=6)
  assert numpy_oracle.compare(got, a, lo, hi) == 0          # 1e-5 rule
  assert numpy_oracle.compare(got, a * 1.001, lo, hi) > 0


def test_integer_semantics():
  """C promotion, truncating division, wrap on store (grammar.py:123-136)."""
  from oracle import c_oracle, numpy_oracle
  st = core.from_text('kernel: k\nburst width: 64\nunroll factor: 1\n'
                      'iterate: 1\ninput int16: a(8, *)\n'
                      'local int16: d(0, 0) = (a(0, 0) - a(1, 0)) / 3\n'
                      'local uint16: m(0, 0) = a(0, 0) * a(0, 0)\n'
                      'output int16: o(0, 0) = int32(d(0, 0)) * 257 % 7 + '
                      'm(0, 0) / 256 - (a(0, 1) > a(0, 0))')
  a = np.array([[-7, 5, 300, -300, 32767, -32768, 11, 0],
                [1, 2, 3, 4, 5, 6, 7, 8]], np.int16)
  got = numpy_oracle.run(st, {'a': a}, keep_locals=True)
  assert got['d'][0, :3].tolist() == [-4, -98, 200]      # trunc toward zero
  assert got['m'][0, 2] == np.uint16(300 * 300 % 65536)
  assert np.array_equal(c_oracle.COracle(st).run({'a': a})['o'], got['o'])


def test_min_max_calls_and_sqrt():
  from oracle import c_oracle, numpy_oracle
  for name, feed in (('erosion.soda', 'input'), ('denoise2d.soda', None)):
    st = core.from_file(soda_path(name))
    rng = np.random.default_rng(9)
    ins = {}
    for n, t in zip(st.input_names, st.input_types):
      if t.is_float:
        ins[n] = rng.random((40, 48), dtype=np.float32)
      else:
        ins[n] = rng.integers(-2000, 2000, (40, 48)).astype(t.np_name)
    a = numpy_oracle.run(st, ins)
    b = c_oracle.COracle(st).run(ins)
    for o in st.output_names:
      assert np.array_equal(a[o], b[o]), name


@pytest.mark.parametrize('name', ['jacobi2d', 'blur', 'heat3d', 'skew2d',
                                  'sobel2d', 'denoise2d', 'jacobi3d',
                                  'denoise3d'])
def test_committed_golden_vectors(name):
  """tests/golden/*.npz, written by tests/golden/make_golden.py."""
  from oracle import numpy_oracle
  path = os.path.join(GOLDEN_DIR, '%s.npz' % name)
  data = np.load(path)
  st = core.from_file(soda_path('%s.soda' % name), iterate=int(data['iterate']))
  ins = {n: data['in_' + n] for n in st.input_names}
  got = numpy_oracle.run(st, ins)
  for o in st.output_names:
    assert np.array_equal(got[o], data['out_' + o])


GOLDEN_EXTRA = [('jacobi2d_preserve', 'jacobi2d.soda', 'preserve'),
                ('heat3d_preserve', 'heat3d.soda', 'preserve'),
                ('conv2d', 'conv2d.soda', None),
                ('conv2d_preserve', 'conv2d.soda', 'preserve')]


@pytest.mark.parametrize('tag,soda,border', GOLDEN_EXTRA)
def test_committed_golden_vectors_preserve_and_params(tag, soda, border):
  """`border: preserve` and `param` arrays as this build defines them
  (DESIGN.md 4.5, 4.6), frozen; both oracles reproduce them."""
  from oracle import c_oracle, numpy_oracle
  data = np.load(os.path.join(GOLDEN_DIR, '%s.npz' % tag))
  st = core.from_file(soda_path(soda), iterate=int(data['iterate']),
                      border=border)
  ins = {n: data['in_' + n] for n in st.input_names + st.param_names}
  for run in (numpy_oracle.run, c_oracle.COracle(st, openmp=False).run):
    got = run(st, ins) if run is numpy_oracle.run else run(ins)
    for o in st.output_names:
      assert np.array_equal(got[o], data['out_' + o])


@pytest.mark.parametrize('name', ['coupled2d.soda', 'lets2d.soda',
                                  'ints2d.soda'])
def test_language_surface_numpy_vs_generated_c(name):
  """Multi-input/-output iteration chaining, let variables, double, integer
  widths, select/abs/max: the two generic restatements agree bit for bit."""
  from oracle import c_oracle, numpy_oracle
  st = core.from_file(soda_path(name))
  rng = np.random.default_rng(1)
  ins = {}
  for n, t in zip(st.input_names, st.input_types):
    ins[n] = (rng.random((30, 40)).astype(t.np_name) if t.is_float else
              rng.integers(0, 256, (30, 40)).astype(t.np_name))
  a = numpy_oracle.run(st, ins)
  b = c_oracle.COracle(st).run(ins)
  for o in st.output_names:
    assert np.array_equal(a[o], b[o])
    lo, hi = st.valid_box((40, 30), o)
    assert a[o][lo[1]:hi[1], lo[0]:hi[0]].any()


def test_compare_rule():
  from oracle import numpy_oracle
  want = np.full((4, 4), 100.0, np.float32)
  got = want.copy()
  got[1, 1] += 0.0005       # abs > 1e-5 but rel = 5e-6: passes, as in the ref
  assert numpy_oracle.compare(got, want, (0, 0), (4, 4)) == 0
  got[2, 2] += 0.01
  assert numpy_oracle.compare(got, want, (0, 0), (4, 4)) == 1
  assert numpy_oracle.compare(got, want, (0, 0), (2, 2)) == 0
  a = np.arange(16, dtype=np.uint16).reshape(4, 4)
  b = a.copy()
  b[3, 3] += 1
  assert numpy_oracle.compare(b, a, (0, 0), (4, 4)) == 1


def test_compare_rule_honours_threshold_from_the_environment(monkeypatch):
  """`$THRESHOLD` replaces the 1e-5 of the rule, read with atof() as the
  reference's generated host does (frt/host.py:634-637); integers ignore it."""
  from oracle import numpy_oracle
  want = np.full((4, 4), 100.0, np.float32)
  got = want.copy()
  got[2, 2] += 0.01                      # rel 1e-4
  assert numpy_oracle.compare(got, want, (0, 0), (4, 4)) == 1
  monkeypatch.setenv('THRESHOLD', '0.001')
  assert numpy_oracle.compare(got, want, (0, 0), (4, 4)) == 0
  monkeypatch.setenv('THRESHOLD', '1e-6')
  got[1, 1] += 0.0005                    # rel 5e-6: now a mismatch too
  assert numpy_oracle.compare(got, want, (0, 0), (4, 4)) == 2
  monkeypatch.setenv('THRESHOLD', '1e-3 (loose)')       # atof: leading number
  assert numpy_oracle.compare(got, want, (0, 0), (4, 4)) == 0
  monkeypatch.setenv('THRESHOLD', 'none')               # atof: 0.0
  assert numpy_oracle.compare(got, want, (0, 0), (4, 4)) == 2
  assert numpy_oracle.compare(got, want, (0, 0), (4, 4), threshold=1e-3) == 0
  a = np.arange(16, dtype=np.uint16).reshape(4, 4)
  b = a.copy()
  b[3, 3] += 1
  monkeypatch.setenv('THRESHOLD', '10')
  assert numpy_oracle.compare(b, a, (0, 0), (4, 4)) == 1


# -- border: preserve (this build's definition; the reference only parses it) --
@pytest.mark.parametrize('name,iterate,extent', [
    ('jacobi2d.soda', 3, (37, 21)),
    ('blur.soda', 2, (40, 17)),            # two stages: box through the local
    ('heat3d.soda', 2, (12, 11, 9)),
    ('seidel2d.soda', 4, (33, 20)),
])
def test_preserve_border_oracles_agree(name, iterate, extent):
  from oracle import c_oracle, numpy_oracle
  st = core.from_file(soda_path(name), iterate=iterate, border='preserve')
  rng = np.random.default_rng(3)
  ins = {n: (rng.random(tuple(extent[::-1])).astype(t.np_name) if t.is_float
             else rng.integers(0, 1000, tuple(extent[::-1])).astype(t.np_name))
         for n, t in zip(st.input_names, st.input_types)}
  a = numpy_oracle.run(st, ins)
  b = c_oracle.COracle(st, openmp=False).run(ins)
  plain = core.from_file(soda_path(name), iterate=1)
  one = numpy_oracle.run(core.from_file(soda_path(name), iterate=1,
                                        border='preserve'), ins)
  ref1 = numpy_oracle.run(plain, ins)
  for o, i in zip(st.output_names, st.input_names):
    assert np.array_equal(a[o].view(np.uint8), b[o].view(np.uint8))
    assert st.valid_box(extent, o) == ((0,) * st.dim, tuple(extent))
    # one iteration: the plain result inside its box, the input outside
    lo, hi = plain.valid_box(extent, o)
    idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
    assert np.array_equal(one[o][idx], ref1[o][idx])
    mask = np.ones(one[o].shape, bool)
    mask[idx] = False
    assert np.array_equal(one[o][mask], ins[i][mask])
    # any number of iterations: the outermost layer is the input's
    rim = np.ones(a[o].shape, bool)
    rim[tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))] = False
    assert np.array_equal(a[o][rim], ins[i][rim])


def test_preserve_border_heat3d_ramp_is_a_fixed_point_everywhere():
  """heat3d's weights are powers of two summing to 1: a linear ramp is exactly
  reproduced inside, and preserved on the border -> output == input on the
  WHOLE grid for any iteration count."""
  from oracle import numpy_oracle
  st = core.from_file(soda_path('heat3d.soda'), iterate=5, border='preserve')
  r, q, p = np.meshgrid(np.arange(10), np.arange(13), np.arange(16),
                        indexing='ij')
  a = (p + q + r).astype(np.float32)
  out = numpy_oracle.run(st, {st.input_names[0]: a})
  assert np.array_equal(out[st.output_names[0]], a)


def test_preserve_border_needs_paired_tensors():
  from oracle import numpy_oracle
  from soda_amd import util
  st = core.from_text(
      'kernel: k\nburst width: 64\nunroll factor: 2\niterate: 1\n'
      'border: preserve\ninput float: a(32, *)\ninput float: b\n'
      'output float: c(0, 0) = a(0, 1) + b(0, -1)\n')
  with pytest.raises(util.SemanticError, match='border: preserve'):
    numpy_oracle.run(st, {'a': np.zeros((4, 4), np.float32),
                          'b': np.zeros((4, 4), np.float32)})


# -- param arrays ---------------------------------------------------------------
def test_param_arrays_closed_form_and_oracles_agree():
  from oracle import c_oracle, numpy_oracle
  st = core.from_file(soda_path('conv2d.soda'), iterate=1)
  assert st.param_names == ('w', 'bias')
  assert [dict(s.taps).keys() for s in st.stages][0] == {'img': 0}.keys()
  rng = np.random.default_rng(0)
  ins = {'img': rng.random((20, 30), dtype=np.float32),
         'w': rng.random((3, 3), dtype=np.float32),
         'bias': np.array([0.5], np.float32)}
  a = numpy_oracle.run(st, ins)['out']
  b = c_oracle.COracle(st, openmp=False).run(ins)['out']
  assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
  # w(i, j) is the C array element w[i][j]; the sum associates left to right
  y, x = 7, 9
  acc = None
  for dy in (-1, 0, 1):
    for dx in (-1, 0, 1):
      term = np.float32(ins['img'][y + dy, x + dx] * ins['w'][dy + 1, dx + 1])
      acc = term if acc is None else np.float32(acc + term)
  assert a[y, x] == np.float32(acc + np.float32(0.5))
  assert st.valid_box((30, 20)) == ((1, 1), (29, 19))


@pytest.mark.parametrize('bad,why', [
    ('w(3, 0)', 'reads element'),          # out of range
    ('w(0)', 'reads element'),             # wrong arity
    ('w', 'without an index'),             # array used as a scalar
])
def test_param_reference_errors(bad, why):
  from soda_amd import util
  text = ('kernel: k\nburst width: 64\nunroll factor: 2\niterate: 1\n'
          'input float: a(32, *)\nparam float: w[3][3]\n'
          'output float: b(0, 0) = a(0, 0) * %s\n' % bad)
  with pytest.raises(util.SemanticError, match=why):
    core.from_text(text)


# ---------------------------------------------------------------------------
# the rest of the reference's 2-D corpus against hand-written loop nests
# (oracle/kat_kernels.c: written from the DSL text, never through
# soda_amd.grammar / core / ir)
# ---------------------------------------------------------------------------

def _i16(rng, shape, lo=-120, hi=120):
  return rng.integers(lo, hi, shape).astype(np.int16)


def _corpus_kat(kat, name, rng):
  """(inputs by DSL name, expected output by DSL name) from the hand-written
  kernel of corpus program `name`."""
  shape = (30, 44)                # 44 columns (dim 0), 30 rows
  h, w = shape
  if name == 'sobel2d':
    img = _i16(rng, shape)
    out = np.empty(shape, np.uint16)
    assert kat.kat_sobel2d(_ptr(img), _ptr(out), w, h) == 0
    return {'img': img}, {'mag': out}
  if name == 'seidel2d':
    a = rng.random(shape, dtype=np.float32)
    out = np.empty_like(a)
    assert kat.kat_seidel2d(_ptr(a), _ptr(out), w, h, 2) == 0
    return {'input': a}, {'output': out}
  if name == 'denoise2d':
    f = rng.random(shape, dtype=np.float32)
    u = rng.random(shape, dtype=np.float32)
    out = np.empty_like(f)
    assert kat.kat_denoise2d(_ptr(f), _ptr(u), _ptr(out), w, h) == 0
    return {'f': f, 'u': u}, {'output': out}
  if name == 'erosion':
    a = _i16(rng, shape, -30000, 30000)
    out = np.empty_like(a)
    assert kat.kat_erosion(_ptr(a), _ptr(out), w, h) == 0
    return {'input': a}, {'output': out}
  if name == 'xcorr':
    a = _i16(rng, shape, -3000, 3000)
    out = np.empty_like(a)
    assert kat.kat_xcorr(_ptr(a), _ptr(out), w, h) == 0
    return {'input': a}, {'tmp3': out}
  shape3 = (14, 18, 22)           # 22 cells along dim 0, 14 planes
  nz, ny, nx = shape3
  if name == 'jacobi3d':
    a = rng.random(shape3, dtype=np.float32)
    out = np.empty_like(a)
    assert kat.kat_jacobi3d(_ptr(a), _ptr(out), nx, ny, nz, 2) == 0
    return {'t1': a}, {'t0': out}
  if name == 'denoise3d':
    f = rng.random(shape3, dtype=np.float32)
    u = rng.random(shape3, dtype=np.float32)
    out = np.empty_like(f)
    assert kat.kat_denoise3d(_ptr(f), _ptr(u), _ptr(out), nx, ny, nz) == 0
    return {'f': f, 'u': u}, {'output': out}
  raise KeyError(name)


@pytest.mark.parametrize('name', ['sobel2d', 'seidel2d', 'denoise2d',
                                  'erosion', 'xcorr', 'jacobi3d',
                                  'denoise3d'])
def test_corpus_against_hand_written_kernels(kat, name):
  """Front-end + both generated oracles against an independent reading of the
  program text, bit for bit on the valid box; zero outside it."""
  from oracle import c_oracle, numpy_oracle
  st = core.from_file(soda_path(name + '.soda'))
  ins, want = _corpus_kat(kat, name, np.random.default_rng(99))
  got_np = numpy_oracle.run(st, ins)
  got_c = c_oracle.COracle(st).run(ins)
  for o, w in want.items():
    lo, hi = st.valid_box(w.shape[::-1], o)
    idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
    assert w[idx].any(), 'degenerate test data'
    assert np.array_equal(got_np[o][idx], w[idx])
    assert np.array_equal(got_c[o][idx], w[idx])
    outside = np.ones(w.shape, bool)
    outside[idx] = False
    assert not w[outside].any()       # the hand-derived box IS the valid box


def _contrast_taps():
  """(dx, dy, coef) of every term of contrast.soda, read from the DSL text by
  a regular expression (not by this repo's parser)."""
  import re
  with open(soda_path('contrast.soda')) as f:
    text = f.read()
  taps = re.findall(r'input\s*\(\s*(-?\d+)\s*,\s*(-?\d+)\s*\)\s*\*\s*(-?\d+)', text)
  return [tuple(map(int, t)) for t in taps]


def test_contrast_follows_the_reference_rebalance(kat):
  """contrast.soda is a 197-term fp32 sum; the reference's always-on
  `inline.rebalance` (inline.py:175-262) evaluates it as six 32-term locals
  plus a 5-term remainder.  The hand-written kernel restates that association;
  front-end + oracles must equal it bit for bit -- and must NOT equal the
  plain left-to-right sum of the program text."""
  from oracle import c_oracle, numpy_oracle
  taps = _contrast_taps()
  assert len(taps) == 197
  dx, dy, cf = (np.array(v, np.int32) for v in zip(*taps))
  rng = np.random.default_rng(5)
  a = rng.random((40, 56), dtype=np.float32)
  h, w = a.shape
  want = np.empty_like(a)
  assert kat.kat_rebalanced_sum(_ptr(a), _ptr(want), w, h, len(taps), _ptr(dx),
                                _ptr(dy), _ptr(cf), 32) == 0
  plain = np.empty_like(a)
  assert kat.kat_rebalanced_sum(_ptr(a), _ptr(plain), w, h, len(taps), _ptr(dx),
                                _ptr(dy), _ptr(cf), 1000) == 0
  st = core.from_file(soda_path('contrast.soda'))
  lo, hi = st.valid_box((w, h))
  assert (lo, hi) == ((0, 0), (w - 16, h - 16))
  idx = (slice(lo[1], hi[1]), slice(lo[0], hi[0]))
  got_np = numpy_oracle.run(st, {'input': a})['output']
  got_c = c_oracle.COracle(st).run({'input': a})['output']
  assert np.array_equal(got_np[idx], want[idx])
  assert np.array_equal(got_c[idx], want[idx])
  assert not np.array_equal(plain[idx], want[idx])
  # The association is not a detail: the coefficients sum to ~0, the terms
  # cancel, and on uniform [0, 1) data the plain left-to-right sum FAILS the
  # reference's own compare rule (frt/host.py:634-657, 1e-5) against the
  # rebalanced sum in dozens of cells.  Any re-associated evaluation -- an
  # MFMA fma chain included -- is in the same position.
  assert numpy_oracle.compare(plain, want, lo, hi) >= 10
  # The evaluation most favourable to an MFMA path: the reference's own seven
  # groups in their textual tap order, only with every multiply-add FUSED --
  # one rounding per term, what `v_mfma_f32_*` does along K.  (fp64 holds the
  # product of two fp32 exactly; the double rounding of the sum is far below
  # the effect measured.)  It still fails the compare rule in several cells.
  fused = np.zeros_like(a)
  hh, ww = hi[1] - lo[1], hi[0] - lo[0]
  parts = []
  for g0 in range(0, len(taps), 32):
    acc = None
    for x, y, c in taps[g0:g0 + 32]:
      term = a[y:y + hh, x:x + ww].astype(np.float64) * np.float64(np.float32(c))
      acc = term.astype(np.float32) if acc is None else \
          (acc.astype(np.float64) + term).astype(np.float32)
    parts.append(acc)
  total = parts[-1]
  for part in parts[:-1]:
    total = total + part
  fused[idx] = total
  assert numpy_oracle.compare(fused, want, lo, hi) >= 3
