"""Stencil core KATs: messages and tensor names restated from reference
src/tests/test_core.py:62-88, layout from src/tests/test_util.py:7-11, window
numbers from SURVEY.md section 8(c) (derived from core.py:858-926)."""
import pytest

from conftest import soda_path
from soda_amd import core, grammar, ir, util

BLUR_ITER2 = r'''
kernel: blur
burst width: 512
unroll factor: 16
input uint16: input(2000, *)
local uint16: tmp(0,0)=(input(-1,0)+input(0,0)+input(1,0))/3
output uint16: output(0,0)=(tmp(0,-1)+tmp(0,0)+tmp(0,1))/3
iterate: 2
border: preserve
cluster: none
'''


def _args(program, **extra):
  args = dict(program.__dict__)
  args['replication_factor'] = 1
  args.update(extra)
  return args


def test_number_of_inputs_differs_from_outputs():
  program = grammar.parse(BLUR_ITER2)
  extra = grammar.InputStmt(haoda_type=ir.Type('uint16'), name='bar',
                            tile_size=[233], dram=())
  with pytest.raises(util.SemanticError) as e:
    core.Stencil(**_args(program,
                         input_stmts=[program.input_stmts[0], extra]))
  assert str(e.value) == (
      'number of input tensors must be the same as output if iterate > 1 times,'
      ' currently there are 2 input(s) but 1 output(s)')


def test_input_type_differs_from_output():
  program = grammar.parse(BLUR_ITER2)
  other = grammar.InputStmt(haoda_type=ir.Type('half'), name='input',
                            tile_size=[2000], dram=())
  with pytest.raises(util.SemanticError) as e:
    core.Stencil(**_args(program, input_stmts=[other]))
  assert str(e.value) == (
      'input must have the same type(s) as output if iterate > 1 times, '
      'current input has type [half] but output has type [uint16]')


def test_cannot_iterate_zero_times():
  with pytest.raises(util.SemanticError) as e:
    core.from_text(BLUR_ITER2, iterate=0)
  assert str(e.value) == 'cannot iterate 0 times'


def test_high_level_dag_construction():
  stencil = core.from_text(BLUR_ITER2)
  names = ('input', 'tmp', 'input_iter1', 'tmp_iter1', 'output')
  assert tuple(stencil.tensors) == names
  assert tuple(t.name for t in stencil.chronological_tensors) == names


def test_serialize_round_trip():
  idx, tile = (42, 23, 233), (2333, 233, 0)
  assert util.deserialize(util.serialize(idx, tile), tile) == idx
  assert util.serialize((3, 2), (2000, 0)) == 4003


@pytest.mark.parametrize('name,iterate,tile,dim,offset,distance', [
    ('blur.soda', 1, None, [3, 3], (0, 0), 4002),
    ('jacobi2d.soda', 1, None, [3, 3], (1, 1), 65),
    ('jacobi2d.soda', 2, None, [5, 5], (2, 2), 130),
    ('heat3d.soda', 1, None, [3, 3, 3], (1, 1, 1), 2081),
    ('heat3d.soda', 2, None, [5, 5, 5], (2, 2, 2), 4162),
])
def test_window_kats(name, iterate, tile, dim, offset, distance):
  stencil = core.from_file(soda_path(name), iterate=iterate)
  window = stencil.stencil_window
  assert core.get_stencil_dim(window) == dim
  assert core.get_stencil_window_offset(window) == offset
  assert stencil.stencil_distance == distance


def test_window_point_counts():
  assert len(core.from_file(soda_path('blur.soda')).stencil_window) == 9
  assert len(core.from_file(soda_path('jacobi2d.soda'),
                            iterate=1).stencil_window) == 5
  # iterate n: a diamond of radius n
  assert len(core.from_file(soda_path('jacobi2d.soda'),
                            iterate=3).stencil_window) == 25


@pytest.mark.parametrize('name', ['jacobi2d.soda', 'heat3d.soda', 'blur.soda',
                                  'denoise2d.soda', 'skew2d.soda',
                                  'sobel2d.soda', 'xcorr.soda'])
@pytest.mark.parametrize('iterate', [1, 2, 3])
def test_analytic_bounds_equal_enumerated_window(name, iterate):
  """window_bounds() (O(iterate)) is the bounding box of the enumerated
  point set the reference computes (core.py:876-919)."""
  stencil = core.from_file(soda_path(name))
  if iterate > 1 and len(stencil.input_names) != len(stencil.output_names):
    pytest.skip('not iterable')
  for out in stencil.output_names:
    pts = stencil.stencil_window_points(out, iterate)
    lo, hi = stencil.window_bounds(iterate)[out]
    assert lo == tuple(min(p[d] for p in pts) for d in range(stencil.dim))
    assert hi == tuple(max(p[d] for p in pts) for d in range(stencil.dim))


def test_valid_boxes_of_baseline_configs():
  j = core.from_file(soda_path('jacobi2d.soda'), iterate=100)
  assert j.valid_box((8192, 8192)) == ((100, 100), (8092, 8092))
  j = core.from_file(soda_path('jacobi2d.soda'), iterate=1000)
  assert j.valid_box((8192, 8192)) == ((1000, 1000), (7192, 7192))
  h = core.from_file(soda_path('heat3d.soda'), iterate=50)
  assert h.valid_box((512, 512, 512)) == ((50, 50, 50), (462, 462, 462))
  b = core.from_file(soda_path('blur.soda'))
  assert b.valid_box((2000, 1024)) == ((0, 0), (1998, 1022))
  assert b.valid_box((16384, 16384)) == ((0, 0), (16382, 16382))
  assert j.radius == ((-1, -1), (1, 1))
  assert b.radius == ((0, 0), (2, 2))


def test_overrides_and_str():
  s = core.from_file(soda_path('jacobi2d.soda'), iterate=7, tile_size=[64],
                     unroll_factor=4, border='preserve')
  assert (s.iterate, s.tile_size, s.unroll_factor, s.border) == (
      7, (64, 0), 4, 'preserve')
  assert s.kernel_name == 'jacobi2d_kernel'
  text = str(s)
  assert text.startswith('kernel: jacobi2d\nburst width: 64\niterate: 7\n')
  assert 'output dram 1 float: t0(0, 0) = (t1(0, 1) + t1(1, 0)' in text
  # replication factor overrides unroll factor (reference sodac.py:161-170)
  s = core.from_file(soda_path('jacobi2d.soda'), replication_factor=3)
  assert (s.unroll_factor, s.replication_factor) == (3, 3)


def test_semantic_checks():
  base = ('kernel: k\nburst width: 64\nunroll factor: 1\niterate: 1\n'
          'input float: a(8, *)\n')
  with pytest.raises(util.SemanticError, match='unknown tensor'):
    core.from_text(base + 'output float: o(0, 0) = b(0, 0)')
  with pytest.raises(util.SemanticError, match='indices'):
    core.from_text(base + 'output float: o(0, 0) = a(0)')
  with pytest.raises(util.SemanticError, match='unknown variable'):
    core.from_text(base + 'output float: o(0, 0) = a(0, 0) + v')
  with pytest.raises(util.InputError, match='conflicting'):
    core.from_text(base + 'local float: o(0, 0) = a(0, 0)\n'
                   'output float: o(0, 0) = a(0, 0)')
  with pytest.raises(util.SemanticError, match='cyclic'):
    core.from_text(base + 'local float: x(0, 0) = y(0, 0)\n'
                   'local float: y(0, 0) = x(0, 0)\n'
                   'output float: o(0, 0) = y(0, 0)')


def test_stage_order_follows_dependences_not_file_order():
  s = core.from_text('kernel: k\nburst width: 64\nunroll factor: 1\n'
                     'iterate: 1\ninput float: a(8, *)\n'
                     'local float: second(0, 0) = first(0, 1)\n'
                     'local float: first(0, 0) = a(1, 0)\n'
                     'output float: o(0, 0) = second(0, 0)')
  assert [st.name for st in s.ordered_stages] == ['first', 'second', 'o']
  # the enumerated window is the single point (1, 1) ...
  assert s.stencil_window_points('o') == ((1, 1),)
  # ... but the box also keeps every LOADED element in the grid (the element
  # second(0,0) reads, first(0,1), exists only where 0 is in range too)
  assert s.window_bounds()['o'] == ((0, 0), (1, 1))


def test_rebalance_of_contrast_known_answer():
  """`inline.rebalance` (ref src/soda/optimization/inline.py:175-262, always
  run from core.py:138) on the one corpus program it touches, derived by hand
  from the text of contrast.soda: its 197 terms come in rows of 1, 7, 11, 13,
  13, 15, 15, 17, 17, 15, 15, 13, 13, 11, 7, 1+... taps; 32-term groups in
  textual order end after the rows' running totals 32, 64, ... ; six locals
  cr_var_0..5 (ref core.py:183-191 naming) and a 5-term remainder."""
  from conftest import soda_path
  from soda_amd import ir
  st = core.from_file(soda_path('contrast.soda'))
  assert st.local_names == tuple('cr_var_%d' % i for i in range(6))
  assert all(str(t) == 'float' for t in st.local_types)
  bare = lambda e: e.expr if isinstance(e, ir.Cast) else e
  groups = [bare(s.expr) for s in st.local_stmts]
  assert [len(g.operands) for g in groups] == [32] * 6
  assert all(set(g.operators) == {'+'} for g in groups)
  # first and last term of every group, counted off the program text:
  # rows hold 1, 7, 11, 13 (=32) | 13, 15, +4 of row 6 (=32) | ...
  ends = [('input(8, 0) * -106', 'input(14, 3) * -98'),
          ('input(2, 4) * -73', 'input(4, 6) * 37'),
          ('input(5, 6) * 67', 'input(5, 8) * 84')]
  for g, (first, last) in zip(groups, ends):
    assert (str(g.operands[0]), str(g.operands[-1])) == (first, last)
  assert all(s.ref.idx == (0, 0) for s in st.local_stmts)
  out = bare(st.output_stmts[0].expr)
  assert len(out.operands) == 5 + 6 and set(out.operators) == {'+'}
  assert [str(o) for o in out.operands[5:]] == [
      'cr_var_%d(0, 0)' % i for i in range(6)]
  assert str(out.operands[4]) == 'input(8, 16) * -106'
  # windows: the output still spans the whole 17 x 17 footprint
  assert st.window_bounds(1)['output'] == ((0, 0), (16, 16))


@pytest.mark.parametrize('name', ['jacobi2d', 'blur', 'heat3d', 'sobel2d',
                                  'seidel2d', 'denoise2d', 'denoise3d',
                                  'erosion', 'xcorr', 'jacobi3d'])
def test_rebalance_leaves_the_rest_of_the_corpus_alone(name):
  """No other corpus program has an fp32 `+` chain of more than 32 terms."""
  from conftest import soda_path
  st = core.from_file(soda_path(name + '.soda'))
  assert not any(n.startswith('cr_var_') for n in st.local_names)


def test_rebalance_groups_by_item_count():
  """Terms `coeff * (a + b + ...)` count as their inner length, are sorted
  largest first (stably) and rebuilt as `(a + b + ...) * coeff` (ref
  inline.py:191-203, :228-233)."""
  inner = lambda k, n: ' + '.join('a(%d, %d)' % (i, k) for i in range(n))
  text = '''kernel: t
burst width: 64
unroll factor: 2
iterate: 1
input dram 0 float: a(64, *)
output dram 1 float: b(0, 0) = a(0, 9) + 2.0f * (%s) + (%s) * 3.0f + a(1, 9)
''' % (inner(0, 20), inner(1, 30))
  st = core.from_text(text)
  # items 1, 20, 30, 1 -> sorted 30, 20, 1, 1 -> groups [30] and [20, 1, 1]
  # (30 + 20 > 32 opens the second group; 20 + 1 + 1 fits): ONE new local
  from soda_amd import ir
  bare = lambda e: e.expr if isinstance(e, ir.Cast) else e
  assert st.local_names == ('cr_var_0',)
  first = bare(st.local_stmts[0].expr)
  assert first.operators == ('*',) and str(first.operands[1]) == '3.0f'
  assert len(first.operands[0].operands) == 30
  out = bare(st.output_stmts[0].expr)
  assert [len(getattr(o, 'operands', ())) for o in out.operands] == [2, 0, 0, 0]
  assert str(out.operands[0].operands[1]) == '2.0f'       # coefficient last
  assert [str(o) for o in out.operands[1:]] == [
      'a(0, 9)', 'a(1, 9)', 'cr_var_0(0, 0)']


@pytest.mark.parametrize('name', ['jacobi2d', 'blur', 'heat3d', 'sobel2d',
                                  'seidel2d', 'denoise2d', 'erosion', 'xcorr',
                                  'jacobi3d', 'skew2d', 'coupled2d'])
@pytest.mark.parametrize('iterate', [1, 2, 3])
def test_valid_boxes_against_brute_force_windows(name, iterate):
  """Stencil.valid_box (analytic per-dimension bounds, O(iterate)) against the
  reference's own procedure (ref core.py:876-926): enumerate the overall
  stencil window of every tensor as a POINT SET -- Minkowski sums of tap sets
  along the producer chain, iteration after iteration, written here from
  scratch on plain tuples -- and take its per-dimension extremes (the box of
  frt/host.py:565-577)."""
  import re
  from conftest import soda_path
  st = core.from_file(soda_path(name + '.soda'), iterate=1)
  if iterate > 1 and (len(st.input_names) != len(st.output_names) or
                      st.input_types != st.output_types):
    pytest.skip('not iterable')
  st = core.from_file(soda_path(name + '.soda'), iterate=iterate)
  dim = st.dim
  zero = (0,) * dim
  add = lambda a, b: tuple(x + y for x, y in zip(a, b))
  # window of every tensor relative to the program inputs, by enumeration
  window = {n: {zero} for n in st.input_names}
  final = {}
  for it in range(iterate):
    for stage in st.ordered_stages:
      pts = set()
      for parent, taps in stage.taps.items():
        for tap in taps:
          pts |= {add(tap, p) for p in window[parent]}
      window[stage.name] = pts or {zero}
    final = dict(window)
    for i, o in zip(st.input_names, st.output_names):
      window[i] = window[o]
  extent = tuple(40 + 7 * d for d in range(dim))
  for o in st.output_names:
    pts = final[o]
    lo = tuple(max(0, -min(p[d] for p in pts)) for d in range(dim))
    hi = tuple(extent[d] - max(0, max(p[d] for p in pts)) for d in range(dim))
    assert st.valid_box(extent, o) == (lo, hi), (o, iterate)
