"""soda_amd/codegen/hip/exact.py: `1.0f / sqrt(x)` by a cheaper sequence where
the program text proves x >= a positive constant.

The claim is "same bits as hipcc's correctly rounded expansions", which are
the bits the oracle's `1.0f / sqrtf(x)` has on the CPU (IEEE root, IEEE
quotient).  It rests on three legs, each tested here:

  * the range analysis only fires on what it can prove (CPU);
  * the rewrite is private to the HIP lowering -- the caller's program, which
    the oracle reads, is never touched (CPU);
  * the sequence equals the compiler's for EVERY fp32 operand the analysis
    admits: an enumeration of all 1.9e9 of them on the GPU, run with the GPU
    suite, its record committed under profiles/ and keyed by the hash of the
    sequence's text, so the text cannot change without the proof going stale
    (CPU test below).

The reference has no counterpart: its kernel is HLS C++ whose `sqrt` / `/` are
Vivado's IEEE cores (reference src/soda/codegen/xilinx/hls_kernel.py); its
host's oracle nest evaluates the same expression with the CPU's
(reference src/soda/codegen/frt/host.py:558-624)."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, SODA_DIR
from soda_amd import core, ir
from soda_amd.codegen.hip import exact, lower

RECORD = os.path.join(ROOT, 'profiles', 'rsqrt_exact.json')


def _program(expr: str, decl: str = 'input float: u(32, *)') -> core.Stencil:
  return core.from_text('kernel: p\nburst width: 64\nunroll factor: 2\n'
                        'iterate: 1\n%s\noutput float: o(0, 0) = %s\n' %
                        (decl, expr))


def _expr(expr: str, **kw) -> ir.Node:
  return _program(expr, **kw).output_stmts[0].expr


def _calls(st: core.Stencil):
  stmts = st.local_stmts + st.output_stmts
  return [n.name for s in stmts for n in s.expr.walk() if isinstance(n, ir.Call)]


def _unwrap(node: ir.Node) -> ir.Node:
  while isinstance(node, ir.Cast):
    node = node.expr
  return node


@pytest.mark.parametrize('expr,bound', [
    ('1.0f + u(0, 0) * u(0, 0)', 1.0),
    ('0.00005f + u(0, 0) * u(0, 0) + u(1, 0) * u(1, 0)', float(np.float32(0.00005))),
    ('0.5f + 0.25f + (u(0, 0) - u(1, 0)) * (u(0, 0) - u(1, 0))', 0.75),
    ('u(0, 0) * u(0, 0)', 0.0),
    # nothing the text proves:
    ('1.0f + u(0, 0) * u(1, 0)', None),             # not a square
    ('1.0f + u(0, 0)', None),                       # a load can be anything
    ('1.0f - u(0, 0) * u(0, 0)', None),             # a difference
    ('-1.0f + u(0, 0) * u(0, 0)', None),
    ('1.0 + u(0, 0) * u(0, 0)', None),              # evaluated in double
])
def test_lower_bound_is_what_the_text_proves(expr, bound):
  got = exact.lower_bound(_unwrap(_expr(expr)))
  assert got == bound, (expr, got)


def test_lower_bound_rounds_like_the_kernel():
  """The bound of a sum is computed in fp32, left to right as the chain is
  evaluated: monotonic rounding makes it a bound of the rounded sum."""
  got = exact.lower_bound(_unwrap(_expr('1.0f + 0.00000001f + u(0, 0) * u(0, 0)')))
  assert got == float(np.float32(np.float32(1.0) + np.float32(0.00000001)))


@pytest.mark.parametrize('expr,rewritten', [
    ('1.0f / sqrt(1.0f + u(0, 0) * u(0, 0))', True),
    ('1.0f / sqrt(0.00005f + u(0, 0) * u(0, 0)) * u(0, 1)', True),   # chain goes on
    ('u(0, 1) * (1.0f / sqrt(1.0f + u(0, 0) * u(0, 0)))', True),
    ('1.0f / sqrt(u(0, 0) * u(0, 0))', False),          # x may be 0 or denormal
    ('1.0f / sqrt(1.0e-30f + u(0, 0) * u(0, 0))', False),   # below the margin
    ('2.0f / sqrt(1.0f + u(0, 0) * u(0, 0))', False),   # another numerator
    ('u(0, 1) / sqrt(1.0f + u(0, 0) * u(0, 0))', False),
    ('1.0 / sqrt(1.0f + u(0, 0) * u(0, 0))', False),    # a double quotient
    ('1.0f / sqrt(1.0f + u(0, 0))', False),
])
def test_rewrite_fires_only_on_proven_operands(expr, rewritten, monkeypatch):
  monkeypatch.setenv('SODA_HIP_RSQRT', 'g')
  st = _program(expr)
  derived = exact.specialize(st)
  assert ('soda_rsqrt_lb' in _calls(derived)) == rewritten
  if rewritten:
    assert 'sqrt' not in _calls(derived)
  else:
    assert derived is st


def test_the_callers_program_is_never_touched(monkeypatch):
  """The oracle reads the caller's Stencil: the rewrite works on a copy, and
  the oracle's generated C never sees the intrinsic."""
  from oracle import c_oracle
  monkeypatch.setenv('SODA_HIP_RSQRT', 'g')
  for name in ('denoise2d.soda', 'denoise3d.soda'):
    st = core.from_file(os.path.join(SODA_DIR, name))
    before = [str(s.expr) for s in st.local_stmts + st.output_stmts]
    mod = lower.lower(st, lower.LowerOptions(peel=0))
    assert mod.stencil is not st
    assert [str(s.expr) for s in st.local_stmts + st.output_stmts] == before
    assert 'soda_rsqrt_lb' not in _calls(st)
    assert 'soda_rsqrt_lb' not in c_oracle.generate(st)
    src = mod.source
    assert src.count('SODA_DEV float soda_rsqrt_lb(float x)') == 1
    assert 'sqrtf(' not in src.split('SODA_DEV float soda_rsqrt_lb')[1]


def test_off_is_the_program_as_written(monkeypatch):
  """SODA_HIP_RSQRT=off: not one character of any module changes (the JIT
  cache and the counter evidence of the other kernels depend on it), and a
  program without the pattern is the same text under every setting."""
  st = core.from_file(os.path.join(SODA_DIR, 'denoise2d.soda'))
  monkeypatch.setenv('SODA_HIP_RSQRT', 'off')
  off = lower.lower(st, lower.LowerOptions(peel=0))
  assert off.stencil is not None and 'soda_rsqrt_lb' not in off.source
  jac = core.from_file(os.path.join(SODA_DIR, 'jacobi2d.soda'))
  texts = set()
  for v in ('off', 'c', 'g'):
    monkeypatch.setenv('SODA_HIP_RSQRT', v)
    texts.add(lower.lower(jac, lower.LowerOptions(peel=0)).source)
  assert len(texts) == 1
  monkeypatch.setenv('SODA_HIP_RSQRT', 'nonsense')
  with pytest.raises(ValueError):
    lower.lower(st, lower.LowerOptions(peel=0))


def test_the_intrinsic_cannot_be_spelled_in_a_program():
  from soda_amd import util
  with pytest.raises(util.SodaError):
    _program('soda_rsqrt_lb(1.0f + u(0, 0) * u(0, 0))')


def test_fewer_instructions_no_scaling_steps(monkeypatch):
  """What the rewrite buys, from the code objects (hiprtc, no GPU needed): the
  loop of the denoise2d kernel loses its v_div_scale / v_div_fmas for `g`
  (the output's own quotient is a product in this program) and a fifth of its
  vector instructions."""
  from soda_amd import isa, runtime
  if isa.objdump() is None:
    pytest.skip('no llvm-objdump')
  st = core.from_file(os.path.join(SODA_DIR, 'denoise2d.soda'))
  counts = {}
  for v in ('off', 'g'):
    monkeypatch.setenv('SODA_HIP_RSQRT', v)
    mod = lower.lower(st, lower.LowerOptions(peel=0, vec=4, prefetch=8))
    code = runtime.compile_source(mod.source, 'denoise2d.hip')
    instrs = isa.disassemble(code)[mod.kernels[0].name]
    counts[v] = (isa.static_profile(instrs)['loop'].get('valu', 0),
                 sum(1 for _, m, _, _ in instrs if m.startswith('v_div_scale')),
                 sum(1 for _, m, _, _ in instrs if m.startswith('v_sqrt_f32')))
  assert counts['off'][1] > 0 and counts['g'][1] == 0
  assert counts['g'][2] == counts['off'][2]           # as many roots
  assert counts['g'][0] < 0.85 * counts['off'][0], counts


def test_the_committed_enumeration_is_for_this_text():
  """profiles/rsqrt_exact.json is the GPU's verdict on the default variant,
  keyed by the hash of the variant's text: editing the sequence without
  re-running the enumeration (tests/test_exact.py on the GPU box writes
  gpurun_out/rsqrt_exact.json; copy it to profiles/) turns this red."""
  if exact.DEFAULT_VARIANT == 'off':
    return
  with open(RECORD) as f:
    rec = json.load(f)
  entry = rec['variants'][exact.DEFAULT_VARIANT]
  assert entry['text'] == exact.text_key(exact.DEFAULT_VARIANT), (
      'exact.py variant `%s` changed since the enumeration ran' %
      exact.DEFAULT_VARIANT)
  assert entry['mismatch'] == 0 and entry['nan_other_payload'] == 0
  # every operand the analysis can admit was among the cases:
  # [2^-96, +inf] = 0x7f800000 - 0x0f800000 + 1 patterns, + 2 x (2^23 - 1) NaNs
  assert rec['cases'] == (0x7f800000 - 0x0f800000 + 1) + 2 * ((1 << 23) - 1)
  assert float.fromhex(rec['lower_bound']) == exact.ROOT_SCALING_BOUND
  assert exact.REQUIRED_LOWER_BOUND >= exact.ROOT_SCALING_BOUND


def test_scan_program_holds_the_product_text():
  src = exact.scan_source()
  for name, text in exact.VARIANTS.items():
    assert 'namespace v_%s {%s}' % (name, text) in src
    assert exact.text_key(name) in src


@pytest.mark.gpu
def test_every_operand_gets_the_compilers_bits():
  """All 1.9e9 fp32 operands >= 2^-96, +inf and every NaN: each variant of
  exact.py against hipcc's own `1.0f / sqrtf(x)`.  ~1 s of GPU time."""
  path = os.path.join(ROOT, 'soda_amd', '_exact', 'rsqrt_scan')
  exact.build_scan(path)        # (build() made it; a no-op then)
  proc = subprocess.run([path], capture_output=True, text=True, timeout=300)
  assert proc.returncode == 0, proc.stdout + proc.stderr
  rec = json.loads(proc.stdout)
  try:
    from soda_amd import runtime
    rec['compiler'] = runtime.compiler_version()
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'rsqrt_exact.json'), 'w') as f:
      json.dump(rec, f, indent=1, sort_keys=True)
  except OSError:
    pass
  assert rec['cases'] == (0x7f800000 - 0x0f800000 + 1) + 2 * ((1 << 23) - 1)
  for name in exact.VARIANTS:
    entry = rec['variants'][name]
    assert entry['text'] == exact.text_key(name)
    assert entry['mismatch'] == 0, (name, entry)
    assert entry['nan_other_payload'] == 0, (name, entry)


@pytest.mark.gpu
@pytest.mark.parametrize('name,extent', [('denoise2d.soda', (1024, 700)),
                                         ('denoise3d.soda', (96, 80, 40))])
def test_denoise_kernels_keep_the_oracles_bits(name, extent):
  """The two programs of the corpus the rewrite fires on, against the C oracle
  by bits, on the reference's kind of input and on fields spread over 60
  binades with cells whose squares overflow (x = +inf, g = 0)."""
  from oracle import c_oracle
  from soda_amd import runtime
  st = core.from_file(os.path.join(SODA_DIR, name))
  oracle = c_oracle.COracle(st)
  rng = np.random.default_rng(5)
  shape = extent[::-1]
  for kind in ('uniform', 'wide'):
    ins = {}
    for n in st.input_names:
      a = rng.random(shape, dtype=np.float32)
      if kind == 'wide':
        a = ((1 + a) * np.exp2(rng.integers(-30, 30, shape)) *
             rng.choice([-1.0, 1.0], shape)).astype(np.float32)
        a[rng.random(shape) < 1e-3] = 1e25
      ins[n] = a
    want = oracle.run(ins)
    with runtime.Program(st, lower.LowerOptions(), extent=extent) as prog:
      assert 'soda_rsqrt_lb(' in prog.module.source
      got = prog.run(ins)
    for o in st.output_names:
      lo, hi = st.valid_box(extent, o)
      idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
      g, w = got[o][idx], want[o][idx]
      same = (g.view(np.int32) == w.view(np.int32)) | (np.isnan(g) & np.isnan(w))
      assert same.all(), (kind, o, int((~same).sum()))


# --- random programs around the pattern --------------------------------------

_CONSTS = ('1.0f', '0.00005f', '1.0e-25f', '3.0e20f', '0.5f + 0.25f', '7.0e-27f')


def _random_program(seed: int):
  """(text, extent): a program whose locals / output hold `1.0f / sqrt(c +
  squares)` with c from 7e-27 (just above the 2^-90 the rewrite asks for) to
  3e20, squares of differences of random taps, in 1 to 3 dimensions, the
  quotient used at shifted positions, inside products and longer chains."""
  rng = np.random.default_rng(1000 + seed)
  dim = int(rng.integers(1, 4))
  tile = {1: '', 2: '32, ', 3: '16, 8, '}[dim]

  def tap(r=1):
    return '(%s)' % ', '.join(str(int(rng.integers(-r, r + 1))) for _ in range(dim))

  def square():
    if rng.random() < 0.3:
      t = 'u%s' % tap()
      return '%s * %s' % (t, t)
    d = '(u%s - u%s)' % (tap(), tap())
    return '%s * %s' % (d, d)

  def root(c):
    terms = [c] + [square() for _ in range(int(rng.integers(1, 4)))]
    return '1.0f / sqrt(%s)' % ' + '.join(terms)

  c1, c2 = rng.choice(_CONSTS, 2)
  zero = '(%s)' % ', '.join(['0'] * dim)
  lines = ['kernel: rs%d' % seed, 'burst width: 64', 'unroll factor: 2',
           'iterate: 1', 'input float: u(%s*)' % tile,
           'local float: g%s = %s' % (zero, root(c1)),
           'output float: o%s = u%s * g%s + g%s * u%s - %s * u%s / 3.0f' %
           (zero, zero, tap(), tap(), tap(), root(c2), tap())]
  extent = {1: (5000,), 2: (300, 200), 3: (70, 40, 30)}[dim]
  return '\n'.join(lines) + '\n', extent


RANDOM_SEEDS = tuple(range(8))


def test_random_programs_around_the_pattern_are_rewritten(monkeypatch):
  monkeypatch.setenv('SODA_HIP_RSQRT', 'g')
  dims = set()
  for seed in RANDOM_SEEDS:
    text, extent = _random_program(seed)
    st = core.from_text(text)
    dims.add(st.dim)
    for strategy in ('auto', 'direct'):
      mod = lower.lower(st, lower.LowerOptions(strategy=strategy, peel=0))
      src = mod.source
      assert src.count('soda_rsqrt_lb(') >= 3, (seed, strategy)
      assert 'sqrtf(' not in src.split('SODA_DEV float soda_rsqrt_lb')[1]
  assert dims == {1, 2, 3}


@pytest.mark.gpu
@pytest.mark.parametrize('strategy', ['auto', 'direct'])
@pytest.mark.parametrize('seed', RANDOM_SEEDS)
def test_random_programs_around_the_pattern_keep_the_oracles_bits(seed, strategy):
  """... against the C oracle by bits, on inputs whose magnitudes run from
  1e-15 to 1e15 (x from the constant alone up to 1e30 and, for a few cells,
  +inf), through the marching and the direct kernels."""
  from oracle import c_oracle
  from soda_amd import runtime
  text, extent = _random_program(seed)
  st = core.from_text(text)
  rng = np.random.default_rng(seed)
  shape = extent[::-1]
  a = ((1 + rng.random(shape)) * np.power(10.0, rng.integers(-15, 16, shape)) *
       rng.choice([-1.0, 1.0], shape)).astype(np.float32)
  a[rng.random(shape) < 2e-3] = 3e22
  ins = {'u': a}
  want = c_oracle.COracle(st).run(ins)
  with runtime.Program(st, lower.LowerOptions(strategy=strategy),
                       extent=extent) as prog:
    assert 'soda_rsqrt_lb(' in prog.module.source
    got = prog.run(ins)
  lo, hi = st.valid_box(extent, 'o')
  idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
  g, w = got['o'][idx], want['o'][idx]
  same = (g.view(np.int32) == w.view(np.int32)) | (np.isnan(g) & np.isnan(w))
  assert same.all(), (seed, strategy, int((~same).sum()))
  assert np.isfinite(w).mean() > 0.5


# --- shape choice for arithmetic-heavy one-iteration programs -----------------

def test_heavy_fp32_programs_take_narrow_strips():
  """lower.prefers_narrow_strips / runtime.resolve_options: denoise2d (55
  operations per fp32 cell, one iteration) gets 8 bytes per lane and 4 rows in
  flight -- unless the caller fixed either; nothing else of the corpus moves,
  nor any program with narrower or wider cells."""
  from soda_amd import runtime
  d2 = core.from_file(os.path.join(SODA_DIR, 'denoise2d.soda'))
  assert lower.prefers_narrow_strips(d2)
  got = runtime.resolve_options(d2, lower.LowerOptions(), (8192, 8192), probe=False)
  assert (got.vec, got.prefetch) == (2, 4)
  got = runtime.resolve_options(d2, lower.LowerOptions(), (1001, 64), probe=False)
  assert got.vec == 1                                  # rows still decide first
  got = runtime.resolve_options(d2, lower.LowerOptions(vec=4), (8192, 8192),
                                probe=False)
  assert got.vec == 4 and got.prefetch is None         # the caller's word
  got = runtime.resolve_options(d2, lower.LowerOptions(prefetch=8), (8192, 8192),
                                probe=False)
  assert (got.vec, got.prefetch) == (4, 8)
  for name in ('jacobi2d', 'blur', 'sobel2d', 'seidel2d', 'erosion', 'xcorr',
               'contrast', 'heat3d', 'jacobi3d', 'denoise3d'):
    st = core.from_file(os.path.join(SODA_DIR, name + '.soda'))
    assert not lower.prefers_narrow_strips(st), name
  # the same arithmetic on doubles or iterated: not what was measured
  text = open(os.path.join(SODA_DIR, 'denoise2d.soda')).read()
  assert not lower.prefers_narrow_strips(
      core.from_text(text.replace('float', 'double')))
  assert not lower.prefers_narrow_strips(
      core.from_text(text.replace('iterate: 1', 'iterate: 2')
                     .replace('input dram 0 float: f\n', '')
                     .replace('f(0, 0)', '0.5f')))


@pytest.mark.parametrize('seed', RANDOM_SEEDS)
def test_the_two_oracles_agree_on_the_random_programs(seed):
  """What the GPU cases above are compared with: the C oracle (gcc, sqrtss /
  divss) against the numpy restatement (IEEE root and quotient of its own), by
  bits, on the same kind of input -- on the CPU."""
  from oracle import c_oracle, numpy_oracle
  text, extent = _random_program(seed)
  st = core.from_text(text)
  extent = tuple(min(e, 90) for e in extent)
  rng = np.random.default_rng(seed)
  shape = extent[::-1]
  a = ((1 + rng.random(shape)) * np.power(10.0, rng.integers(-15, 16, shape)) *
       rng.choice([-1.0, 1.0], shape)).astype(np.float32)
  a[rng.random(shape) < 2e-3] = 3e22
  with np.errstate(all='ignore'):
    n = numpy_oracle.run(st, {'u': a})['o']
  c = c_oracle.COracle(st).run({'u': a})['o']
  lo, hi = st.valid_box(extent, 'o')
  idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
  g, w = n[idx], c[idx]
  same = (g.view(np.int32) == w.view(np.int32)) | (np.isnan(g) & np.isnan(w))
  assert same.all(), int((~same).sum())
