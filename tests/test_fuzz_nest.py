"""Random programs against loop nests that never touch the product's front-end
(tests/fuzz_nest.py: text and C++ nest emitted side by side from one tree).

CPU: the generic oracle (which READS the text through soda_amd's parser, tap
extraction, boxes and C printer) must equal the independent nest on the whole
array -- values inside the nest's own box, zeros outside -- and the front-end's
valid boxes must be the nest's.  GPU: the kernels against the same nests."""
import numpy as np
import pytest

import fuzz_nest

FAMILIES = ('plain', 'rich', 'window')
CPU_SEEDS = range(0, 60)
GPU_SEEDS = range(0, 25)


def _same(a, b):
  """Bit for bit; a NaN on both sides counts as equal (its payload and sign
  are not defined by the operation that made it)."""
  if a.dtype != b.dtype or a.shape != b.shape:
    return False
  if a.dtype.kind != 'f':
    return bool((a == b).all())
  bits = {4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
  eq = np.ascontiguousarray(a).view(bits) == np.ascontiguousarray(b).view(bits)
  return bool((eq | (np.isnan(a) & np.isnan(b))).all())


def _build(seed, family):
  from soda_amd import core, util
  prog, extent = fuzz_nest.program(seed, family)
  if fuzz_nest.has_empty_box(prog, extent):
    pytest.skip('empty box')
  text = prog.soda_text()
  try:
    stencil = core.from_text(text)
  except util.SodaError as e:     # the generator's programs are all legal
    raise AssertionError('front-end refused a legal program: %s\n%s' %
                         (e, text))
  return prog, extent, text, stencil


@pytest.mark.parametrize('family', FAMILIES)
@pytest.mark.parametrize('seed', CPU_SEEDS)
def test_front_end_and_oracle_against_independent_nests(seed, family):
  from oracle import c_oracle
  prog, extent, text, stencil = _build(seed, family)
  assert stencil.iterate == prog.iterate and stencil.dim == prog.dim
  assert list(stencil.input_names) == [n for n, _ in prog.inputs]
  assert list(stencil.output_names) == [s.name for s in prog.outputs]
  ins = fuzz_nest.inputs_for(prog, extent, seed)
  want = prog.run(ins, extent)
  got = c_oracle.COracle(stencil, openmp=False).run(ins)
  for o in stencil.output_names:
    lo, hi = stencil.valid_box(extent, o)
    assert (tuple(lo), tuple(hi)) == prog.valid_box(extent, o), text
    assert _same(got[o], want[o]), '%s\n%s: %d cells differ' % (
        text, o, int((got[o] != want[o]).sum()))


def test_boxes_are_the_references_where_it_defines_them():
  """The nests' boxes (interval propagation: every load inside the grid) are
  the reference's literal window formula wherever every tensor's window
  contains its own cell -- which holds for enough of the random programs to
  matter, and for the reference's whole corpus."""
  literal = 0
  for family in FAMILIES:
    for seed in CPU_SEEDS:
      prog, _ = fuzz_nest.program(seed, family)
      ref, spans_zero = prog.literal_margins()
      if spans_zero:
        literal += 1
        assert [(lo, m) for _, _, lo, m in ref] == \
            [(lo, m) for _, _, lo, m in prog.margins()]
      else:      # tighter, never looser
        for (_, _, lo, m), (_, _, rlo, rm) in zip(prog.margins(), ref):
          assert all(a >= b for a, b in zip(lo, rlo))
          assert all(a >= b for a, b in zip(m, rm))
  assert literal >= 60


@pytest.mark.parametrize('seed', range(0, 30))
def test_wide_window_programs_against_independent_nests_cpu(seed):
  """The `wide` family (what the ldswin kernels serve) through front-end and C
  oracle against its own nests, on the CPU."""
  from oracle import c_oracle
  prog, extent, text, stencil = _build(seed, 'wide')
  ins = fuzz_nest.inputs_for(prog, extent, seed)
  want = prog.run(ins, extent)
  got = c_oracle.COracle(stencil, openmp=False).run(ins)
  for o in stencil.output_names:
    lo, hi = stencil.valid_box(extent, o)
    assert (tuple(lo), tuple(hi)) == prog.valid_box(extent, o), text
    assert _same(got[o], want[o]), text


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(0, 30))
def test_lds_window_kernels_match_independent_nests(built, seed):
  """`--hip-strategy ldswin` on random wide-window programs (taps on both
  sides of the cell, off-centre stores, a pointwise local folded in, float and
  int32 cells, row lengths from narrower than a lane's reach to three blocks)
  against nests that share nothing with the product."""
  from soda_amd import runtime, util
  from soda_amd.codegen.hip import lower
  prog, extent, text, stencil = _build(seed, 'wide')
  ins = fuzz_nest.inputs_for(prog, extent, seed)
  want = prog.run(ins, extent)
  try:
    hip = runtime.Program(stencil, lower.LowerOptions(strategy='ldswin'),
                          extent=extent)
  except util.SemanticError as e:
    # (a local stored off-centre whose folding would move the output's window
    # stays a tensor: two stages are not ldswin's business)
    assert 'single-stage' in str(e) or 'fold' in str(e), str(e)
    pytest.skip(str(e))
  with hip:
    assert [p.kind for p in hip.module.passes] == ['ldswin']
    got = hip.run(ins)
  for o in stencil.output_names:
    assert _same(got[o], want[o]), (
        'seed %d output %s: %d cells differ\n%s' %
        (seed, o, int((got[o] != want[o]).sum()), text))


def test_nest_generator_covers_the_language():
  blob = '\n'.join(fuzz_nest.program(s, f)[0].soda_text()
                   for f in FAMILIES for s in CPU_SEEDS)
  for needle in ('local ', 'tmp = ', 'min(', 'max(', 'sqrt(', 'double',
                 'uint8', 'int16', 'iterate: 3', ', *)', '(32, 32, *)', ' / ',
                 'select(', ' && ', ' || ', ' % ', ' & ', ' | ', ' ^ ',
                 'int64(', 'abs(', ' - -', '* -', ' == ', ' != ', '<= ',
                 'float(', 'int32('):
    assert needle in blob, needle
  dims = {fuzz_nest.program(s, 'plain')[0].dim for s in CPU_SEEDS}
  assert dims == {1, 2, 3}


def test_nest_files_never_import_the_product():
  import os
  import re
  here = os.path.dirname(os.path.abspath(__file__))
  src = open(os.path.join(here, 'fuzz_nest.py')).read()
  assert not re.search(r'^\s*(import|from)\s+(soda_amd|oracle)', src, re.M)


@pytest.mark.gpu
@pytest.mark.parametrize('family', FAMILIES)
@pytest.mark.parametrize('seed', GPU_SEEDS)
def test_gpu_matches_independent_nests(built, seed, family):
  """The HIP path (the family the lowering picks, and `direct`) against nests
  that share nothing with it but the program's tree."""
  from soda_amd import runtime
  from soda_amd.codegen.hip import lower
  prog, extent, text, stencil = _build(seed, family)
  ins = fuzz_nest.inputs_for(prog, extent, seed)
  want = prog.run(ins, extent)
  for strategy in ('auto', 'direct'):
    with runtime.Program(stencil, lower.LowerOptions(strategy=strategy,
                                                     fuse=(2,)),
                         extent=extent) as hip:
      got = hip.run(ins)
      kinds = sorted({p.kind for p in hip.module.passes})
    for o in stencil.output_names:      # whole arrays: zeros outside the box
      assert _same(got[o], want[o]), (
          'seed %d %s, %s (%s), output %s: %d cells differ\n%s' %
          (seed, family, strategy, kinds, o, int((got[o] != want[o]).sum()),
           text))
