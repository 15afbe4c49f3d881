"""The HIP backend's two program rewrites (soda_amd/optimization/pointwise.py,
windows.py): the derived program computes the caller's tensors bit for bit --
checked here by running ORIGINAL and DERIVED program through the CPU oracle
(test infrastructure) on random data -- with the same windows and valid boxes,
and is the smaller program the kernels are meant to see.  The GPU parity tests
(tests/test_hip_parity.py: corpus, fuzz, goldens) run the kernels lowered from
the derived programs against the oracle's evaluation of the originals."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, SODA_DIR, soda_path
from soda_amd import core, ir

PROGRAMS = sorted(glob.glob(os.path.join(SODA_DIR, '*.soda'))) + sorted(
    glob.glob(os.path.join(GOLDEN_DIR, '*.soda')))


def _inputs(st, rng):
  shape = {2: (26, 44), 3: (12, 14, 26), 4: (6, 7, 8, 9)}[st.dim]
  ins = {}
  for n, t in zip(st.input_names, st.input_types):
    if t.is_float:
      ins[n] = rng.random(shape, dtype=np.float64).astype(t.np_name)
    else:
      info = np.iinfo(np.dtype(t.np_name))
      ins[n] = rng.integers(max(info.min, -40000), min(info.max, 40000),
                            shape).astype(t.np_name)
  for p in st.param_stmts:
    ins[p.name] = rng.random(p.size or (1,)).astype(p.haoda_type.np_name)
  return ins, shape[::-1]


@pytest.mark.parametrize('path', PROGRAMS, ids=os.path.basename)
def test_derived_programs_compute_the_same_tensors(path):
  from oracle import numpy_oracle
  from soda_amd.optimization import pointwise, windows
  st = core.from_file(path)
  derived = windows.decompose(pointwise.inline_pointwise(st))
  if derived is st:
    pytest.skip('nothing to rewrite')
  assert derived.input_names == st.input_names
  assert derived.output_names == st.output_names
  assert derived.window_bounds() .keys() >= set(st.output_names)
  ins, extent = _inputs(st, np.random.default_rng(17))
  want = numpy_oracle.run(st, ins)
  got = numpy_oracle.run(derived, ins)
  for o in st.output_names:
    assert st.valid_box(extent, o) == derived.valid_box(extent, o)
    lo, hi = st.valid_box(extent, o)
    idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
    assert want[o][idx].size and np.array_equal(got[o][idx], want[o][idx]), o


def test_groups_of_a_rebalanced_sum_fold_into_one_stage():
  """pointwise.inline_pointwise(fold_groups=True): contrast's six `cr_var_*`
  groups become cast sub-expressions of the output -- ONE stage for the ldswin
  kernels -- with the association unchanged: the CPU oracle gets the same bits
  from the derived program as from the seven-stage one, and the text is not
  rebalanced a second time."""
  from oracle import c_oracle
  from soda_amd.optimization import pointwise
  st = core.from_file(soda_path('contrast.soda'))
  assert len(st.ordered_stages) == 7
  whole = pointwise.inline_pointwise(st, fold_groups=True, max_ops=1 << 20)
  assert len(whole.ordered_stages) == 1 and not whole.local_stmts
  assert whole.iteration_boxes()['output'] == st.iteration_boxes()['output']
  rng = np.random.default_rng(3)
  extent = (70, 40)
  ins = {'input': rng.random(extent[::-1], dtype=np.float32)}
  want = c_oracle.COracle(st, openmp=False).run(ins)['output']
  got = c_oracle.COracle(whole, openmp=False).run(ins)['output']
  lo, hi = st.valid_box(extent)
  assert hi[0] - lo[0] == 54 and hi[1] - lo[1] == 24
  assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_pointwise_locals_are_folded():
  """denoise3d (ref tests/src/denoise3d.soda:8-29): ten statements, of which
  six differences, r0 and r1 are only read at the cell being computed."""
  from soda_amd.optimization import pointwise
  st = core.from_file(soda_path('denoise3d.soda'))
  d = pointwise.inline_pointwise(st)
  assert [s.name for s in st.local_stmts] == [
      'diff_u', 'diff_d', 'diff_l', 'diff_r', 'diff_i', 'diff_o', 'g', 'r0',
      'r1']
  assert [s.name for s in d.local_stmts] == ['g']       # read at six offsets
  assert d.derived_from is st
  d2 = pointwise.inline_pointwise(core.from_file(soda_path('denoise2d.soda')))
  assert [s.name for s in d2.local_stmts] == ['g']
  # nothing to fold: the very same object comes back
  j = core.from_file(soda_path('jacobi2d.soda'))
  assert pointwise.inline_pointwise(j) is j
  # the groups of a rebalanced sum ARE the reference's association: kept
  c = core.from_file(soda_path('contrast.soda'))
  assert pointwise.inline_pointwise(c) is c


def test_window_reductions_become_power_of_two_chains():
  """erosion's 19-tap min (ref tests/src/erosion.soda:5-15) and xcorr's 19-tap
  sums (ref tests/src/xcorr.soda:5-15): 18 operations per cell as written, 7
  as chains; integer sums through int32 auxiliaries, the statement's own cast
  kept."""
  from soda_amd.optimization import windows
  for name, op, aux_type in (('erosion', 'min', 'int16'),
                             ('xcorr', '+', 'int32')):
    st = core.from_file(soda_path(name + '.soda'))
    d = windows.decompose(st)
    before = {s.name: ir.op_count(s.expr) for s in st.local_stmts +
              st.output_stmts}
    after = {s.name: ir.op_count(s.expr) for s in d.local_stmts +
             d.output_stmts}
    long_ones = [n for n, c in before.items() if c >= 18]
    assert len(long_ones) == 2
    for n in long_ones:
      assert after[n] <= 2
    added = [s for s in d.local_stmts if s.name not in before]
    assert len(added) == 10 and all(str(s.haoda_type) == aux_type
                                    for s in added)
    assert sum(after.values()) <= sum(before.values()) - 15
  # floating point is never touched, nor short windows
  for name in ('jacobi2d', 'heat3d', 'contrast', 'blur', 'sobel2d'):
    st = core.from_file(soda_path(name + '.soda'))
    assert windows.decompose(st) is st


def test_lowering_uses_the_derived_program_and_can_be_told_not_to():
  from soda_amd import runtime
  from soda_amd.codegen.hip import lower
  assert lower.MAX_TENSORS == runtime.MAX_TENSORS
  # a program whose chains would not fit the argument block keeps its windows
  w = core.from_file(soda_path('winsum2d.soda'))
  d = lower.lower(w, lower.LowerOptions(strategy='direct', vec=1))
  assert len(d.stencil.symbol_table) <= runtime.MAX_TENSORS
  st = core.from_file(soda_path('erosion.soda'))
  on = lower.lower(st, lower.LowerOptions(vec=4))
  off = lower.lower(st, lower.LowerOptions(vec=4, windows=False, inline=False))
  # (chains along the streamed dimension; the dimension-0 window is reduced
  # by the marching kernel itself, all cells of a lane jointly)
  assert len(on.stencil.local_stmts) == 6 and on.stencil.derived_from is st
  assert off.stencil is st
  assert 'input_min1_16' in on.source and 'input_min1_16' not in off.source
  assert 'xm_t0_output' in on.source and 'tmp_min0_16' not in on.source
  direct = lower.lower(st, lower.LowerOptions(vec=4, strategy='direct'))
  assert len(direct.stencil.local_stmts) == 11      # chains in both directions


OFF_CENTRE_LOCAL = """kernel: offc
burst width: 64
unroll factor: 2
iterate: 2
border: preserve
input float: a(32, *)
local float: l(0, -1) = a(1, 0) + 1.0f
output float: b(0, 0) = l(0, -1) * 0.5f
"""


def test_rewrites_keep_the_outputs_windows():
  """A local stored off-centre and read back at that very offset is pointwise
  -- but its own cell has to lie in the grid, so row 0 of `b` (which needs row
  -1 of `l`) is NOT computable and keeps its input under `border: preserve`.
  Folded into `b` the constraint would vanish: the lowering keeps the program
  as written when a rewrite would change an output's window."""
  from soda_amd import core
  from soda_amd.codegen.hip import lower
  from soda_amd.optimization import pointwise
  st = core.from_text(OFF_CENTRE_LOCAL)
  assert st.interior_bounds('b') == ((0, -1), (1, 0))
  folded = pointwise.inline_pointwise(st)
  assert [s.name for s in folded.ordered_stages] == ['b']
  assert folded.interior_bounds('b') != st.interior_bounds('b')
  for strategy in ('auto', 'direct'):
    mod = lower.lower(st, lower.LowerOptions(strategy=strategy, fuse=(2,)))
    assert [s.name for s in mod.stencil.ordered_stages] == ['l', 'b']
    assert mod.stencil.interior_bounds('b') == st.interior_bounds('b')


@pytest.mark.gpu
@pytest.mark.parametrize('strategy,fuse', [('auto', (2,)), ('auto', ()),
                                           ('direct', ())])
def test_off_centre_local_under_preserved_border(built, strategy, fuse):
  import numpy as np
  from oracle import c_oracle, numpy_oracle
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  st = core.from_text(OFF_CENTRE_LOCAL)
  extent = (300, 90)
  a = np.random.default_rng(3).random(extent[::-1], dtype=np.float32)
  want = numpy_oracle.run(st, {'a': a})['b']
  assert np.array_equal(want, c_oracle.COracle(st, openmp=False).run({'a': a})['b'])
  assert np.array_equal(want[0], a[0])          # row 0 keeps the input
  with runtime.Program(st, lower.LowerOptions(strategy=strategy, fuse=fuse),
                       extent=extent) as prog:
    got = prog.run({'a': a})['b']
  assert np.array_equal(got, want)
