"""The three hipcc code-generation faults soda_rt.h works around, pinned.

Each was found by random programs on ROCm 7.2 and papered over with an empty
inline asm the optimiser cannot see through (soda_rt.h: `soda_opaque` -- the
DPP-combine pass folding integer lane shifts into their consumers;
`soda_wide` -- packed-byte instruction selection next to a min and lane
shifts; `soda_own_register` -- a min fed from a register that packs four
one-byte cells).  Nothing tied them to a compiler version:
a ROCm bump could re-open one silently, or make one dead weight.

Per fault, on the GPU: the reduced reproducer is built WITH the workaround
(product build: must equal the oracle bit for bit -- asserted) and WITHOUT it
(`#define SODA_UNGUARDED_*` in front of the module: recorded, not asserted --
`still_needed` says whether this compiler still mis-compiles it).  The record
goes to gpurun_out/compiler_pins.json with the compiler's version, and the
bench line carries `roofline.compiler`.  The reference has no counterpart: it
emits HLS C++ for Vivado (reference src/soda/codegen/xilinx/hls_kernel.py)."""
import json
import os

import numpy as np
import pytest

import fuzz
from conftest import ROOT
from soda_amd import core, util

RECORD = os.path.join(ROOT, 'gpurun_out', 'compiler_pins.json')


def _mismatches(stencil, extent, ins, want, opts, prefix=''):
  """Cells of the valid boxes that differ from `want`, or the text of the
  compile error the build ends in."""
  from soda_amd import runtime
  from soda_amd.codegen.hip import lower
  try:
    with runtime.Program(stencil, lower.LowerOptions(**opts), extent=extent,
                         source_prefix=prefix) as prog:
      got = prog.run(ins)
  except util.SodaError as e:
    return 'build failed: %s' % str(e).splitlines()[0][:200]
  bad = 0
  for o in stencil.output_names:
    lo, hi = stencil.valid_box(extent, o)
    idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
    bad += int((got[o][idx] != want[o][idx]).sum())
  return bad


def _record(fault, entry):
  from soda_amd import runtime
  table = {}
  try:
    with open(RECORD) as f:
      table = json.load(f)
  except (OSError, ValueError):
    pass
  entry['compiler'] = runtime.compiler_version()
  table[fault] = entry
  try:
    os.makedirs(os.path.dirname(RECORD), exist_ok=True)
    with open(RECORD, 'w') as f:
      json.dump(table, f, indent=1, sort_keys=True)
  except OSError:
    pass


BYTES_NEXT_TO_A_MIN = """kernel: bytes159
burst width: 64
unroll factor: 2
iterate: 1
input uint8: in0(32, *)
input uint8: in1
local uint16: loc0(0, 0) = in1(2, 0) * 43 * (min(16, 32) * in0(1, 0)) - (in1(-1, -1) + 10) * in1(0, -2) + (in0(2, -2) - in0(2, 1)) * 40 + in0(1, 2)
output uint8: out0(0, 0) = in1(2, 1) * 1
output uint8: out1(0, 0) = (min(int32(in0(-2, 1)), 4) + in1(-2, -2) * in1(2, -1) - (in0(2, -2) + 8 * 16) - in0(1, -2) * in1(0, 1)) / 6
"""


@pytest.mark.gpu
def test_packed_byte_selection_soda_wide(built):
  """Round 4, tools/fuzz_scan.py deep seed 159, reduced: v_dot4_u32_u8 over
  v_perm_b32-assembled operands / SDWA byte selects next to a min and
  lane-shifted copies -- a quarter of the cells wrong above -O0."""
  from oracle import c_oracle
  stencil = core.from_text(BYTES_NEXT_TO_A_MIN)
  extent = (520, 291)
  rng = np.random.default_rng(4401)
  ins = {n: rng.integers(1, 201, extent[::-1]).astype(np.uint8)
         for n in stencil.input_names}
  want = c_oracle.COracle(stencil).run(ins)
  guarded = _mismatches(stencil, extent, ins, want, {})
  bare = _mismatches(stencil, extent, ins, want, {},
                     '#define SODA_UNGUARDED_WIDE 1\n')
  _record('soda_wide', {'guarded_mismatches': guarded,
                        'unguarded': bare, 'still_needed': bare != 0,
                        'reproducer': 'tests/test_compiler_pins.py '
                                      'BYTES_NEXT_TO_A_MIN, 520 x 291'})
  assert guarded == 0


@pytest.mark.gpu
def test_min_from_a_packed_register_soda_own_register(built):
  """Round 3, tools/fuzz_scan.py options seed 613: a uint8 local between
  int16 / int32 tensors in a fused kernel; v_min_i32_sdwa picked the wrong
  side in one cell of one unrolled step."""
  from oracle import c_oracle
  text, dim, _ = fuzz.program(613)
  stencil = core.from_text(text)
  extent = (1100, 207)
  ins = fuzz.inputs_for(stencil, extent, 613)
  want = c_oracle.COracle(stencil, openmp=False).run(ins)
  rows = {}
  worst_guarded = 0
  for label, opts in (('fuse 2', dict(fuse=(2,))),
                      ('fuse 2, all warm-up peeled', dict(fuse=(2,), peel=-1)),
                      ('fuse 3 2, 9-row chunks, two waves along',
                       dict(fuse=(3, 2), chunk_rows=9, waves_y=2))):
    g = _mismatches(stencil, extent, ins, want, opts)
    b = _mismatches(stencil, extent, ins, want, opts,
                    '#define SODA_UNGUARDED_OWN 1\n')
    # (round 4's soda_wide hands one-byte cells to every expression as ints of
    # hidden range, which may be what keeps this fault away now: both off)
    both = _mismatches(stencil, extent, ins, want, opts,
                       '#define SODA_UNGUARDED_OWN 1\n'
                       '#define SODA_UNGUARDED_WIDE 1\n')
    rows[label] = {'guarded_mismatches': g, 'unguarded': b,
                   'unguarded_and_soda_wide_off': both}
    worst_guarded = max(worst_guarded, g if isinstance(g, int) else 1 << 30)
  _record('soda_own_register', {
      'cases': rows,
      'still_needed': any(r['unguarded'] != 0 for r in rows.values()),
      'needed_without_soda_wide': any(r['unguarded_and_soda_wide_off'] != 0
                                      for r in rows.values()),
      'reproducer': 'tests/fuzz.py program(613), 1100 x 207'})
  assert worst_guarded == 0


INT_FUSED = """kernel: intshift
burst width: 64
unroll factor: 2
iterate: 4
input %(t)s: a(32, *)
output %(t)s: b(0, 0) = a(-1, 0) - a(1, 0) + (a(0, 1) + a(0, -1)) / 2 + a(0, 0) - a(1, 1)
"""


@pytest.mark.gpu
def test_dpp_combine_on_integer_shifts_soda_opaque(built):
  """Round 1: LLVM's DPP-combine pass folded integer lane shifts into
  v_add_u32_dpp / v_subrev_u32_dpp with wrong results in fused integer
  stencils, and produced illegal DPP encodings for 64-bit types.  Fused
  integer and double programs with taps to either side, with and without the
  empty asm behind every non-fp32 shift."""
  from oracle import c_oracle
  rows = {}
  worst_guarded = 0
  rng = np.random.default_rng(77)
  for t, np_t in (('int32', np.int32), ('int16', np.int16),
                  ('uint16', np.uint16), ('double', np.float64)):
    stencil = core.from_text(INT_FUSED % {'t': t})
    extent = (640, 203)
    a = (rng.random(extent[::-1]) if t == 'double'
         else rng.integers(0, 5000, extent[::-1])).astype(np_t)
    want = c_oracle.COracle(stencil, openmp=False).run({'a': a})
    for label, opts in (('fuse 4', dict(fuse=(4,))),
                        ('fuse 2', dict(fuse=(2,)))):
      g = _mismatches(stencil, extent, {'a': a}, want, opts)
      b = _mismatches(stencil, extent, {'a': a}, want, opts,
                      '#define SODA_UNGUARDED_OPAQUE 1\n')
      rows['%s, %s' % (t, label)] = {'guarded_mismatches': g, 'unguarded': b}
      worst_guarded = max(worst_guarded, g if isinstance(g, int) else 1 << 30)
  _record('soda_opaque', {
      'cases': rows,
      'still_needed': any(r['unguarded'] != 0 for r in rows.values()),
      'reproducer': 'tests/test_compiler_pins.py INT_FUSED, 640 x 203'})
  assert worst_guarded == 0


def test_the_switches_exist_and_a_product_build_never_sets_them():
  """soda_rt.h carries the three switches; no generator or option defines
  them."""
  from soda_amd import runtime
  from soda_amd.codegen.hip import lower
  text = lower.runtime_text()
  for name in ('SODA_UNGUARDED_OPAQUE', 'SODA_UNGUARDED_WIDE',
               'SODA_UNGUARDED_OWN'):
    assert '#ifndef %s' % name in text
    assert not any(name in o for o in runtime.COMPILE_OPTIONS)
    assert '#define %s' % name not in text
  stencil = core.from_text(BYTES_NEXT_TO_A_MIN)
  src = lower.lower(stencil, lower.LowerOptions()).source
  assert 'soda_wide(' in src and '#define SODA_UNGUARDED' not in src
