"""Cheaper evaluations of fp32 operations that are PROVEN to give the bits of
hipcc's correctly rounded expansions -- the ones the oracle's `1.0f / sqrtf(x)`
gives on the CPU -- for the operands the program text allows.

`1.0f / sqrt(x)` costs 27 vector instructions as hipcc expands it (a
correctly rounded root: scale, v_sqrt, two +-1 ulp residuals and selects,
unscale, class check; a correctly rounded quotient: two v_div_scale, v_rcp,
five fused steps, v_div_fmas, v_div_fixup).  Most of that guards operands the
denoise programs cannot produce: their x is a positive constant plus squares,
so it is >= that constant (or +inf / NaN), the root needs no denormal scaling
and the quotient 1 / root no v_div_scale.  The composite is a function of ONE
fp32 variable, so "same bits" is checked by enumeration over every x >= 2^-96,
+inf and every NaN on the GPU (`scan_source` below, built by build(), run by
tests/test_exact.py on every GPU run; results in profiles/rsqrt_exact.json),
not argued.

The rewrite is a derived program of the HIP lowering only (a backend
intrinsic `soda_rsqrt_lb` no .soda text can spell); the oracle keeps
evaluating the program as written."""
import copy
import os
from typing import Optional

import numpy as np

from soda_amd import core, ir

# below this the compiler's root starts scaling (0x0f800000 = 2^-96); the
# enumeration covers [2^-96, +inf]; the rewrite asks for a margin
ROOT_SCALING_BOUND = float(np.float32(2.0 ** -96))
REQUIRED_LOWER_BOUND = float(np.float32(2.0 ** -90))

# The evaluations the enumeration compares (scan_source() pastes these very
# strings); `variant()` names the one the lowering emits.
_COMMON = '''
// root of x >= 2^-96 (or +inf, NaN): hipcc's own expansion minus the denormal
// scaling -- v_sqrt (1 ulp), then the neighbour whose residual says so
SODA_DEV float soda_sqrt_lb(float x) {
  const float s0 = __builtin_amdgcn_sqrtf(x);
  const float sd = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, s0) - 1u);
  const float su = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, s0) + 1u);
  const float rd = __builtin_fmaf(-sd, s0, x), ru = __builtin_fmaf(-su, s0, x);
  float s = (0.0f >= rd) ? sd : s0;
  s = (0.0f < ru) ? su : s;
  return s;
}
'''
VARIANTS = {
    # hipcc's two expansions minus the scaling steps, nothing else changed
    'c': _COMMON + '''
SODA_DEV float soda_rsqrt_lb(float x) {
  const float s = soda_sqrt_lb(x);
  float r = __builtin_amdgcn_rcpf(s);
  r = __builtin_fmaf(__builtin_fmaf(-s, r, 1.0f), r, r);
  float q = r;
  q = __builtin_fmaf(__builtin_fmaf(-s, q, 1.0f), r, q);
  q = __builtin_fmaf(__builtin_fmaf(-s, q, 1.0f), r, q);
  return __builtin_amdgcn_div_fixupf(q, s, 1.0f);
}
''',
    # two refinements of the quotient instead of three
    'd': _COMMON + '''
SODA_DEV float soda_rsqrt_lb(float x) {
  const float s = soda_sqrt_lb(x);
  float r = __builtin_amdgcn_rcpf(s);
  r = __builtin_fmaf(__builtin_fmaf(-s, r, 1.0f), r, r);
  const float q = __builtin_fmaf(__builtin_fmaf(-s, r, 1.0f), r, r);
  return __builtin_amdgcn_div_fixupf(q, s, 1.0f);
}
''',
    # one refinement
    'g': _COMMON + '''
SODA_DEV float soda_rsqrt_lb(float x) {
  const float s = soda_sqrt_lb(x);
  const float r = __builtin_amdgcn_rcpf(s);
  const float q = __builtin_fmaf(__builtin_fmaf(-s, r, 1.0f), r, r);
  return __builtin_amdgcn_div_fixupf(q, s, 1.0f);
}
''',
}

# (Refuted by the enumeration and not kept here -- profiles/
# r05_rsqrt_candidates.json: b / f, v_rsq seeding root and reciprocal, one
# transcendental instead of two: 224 of 1.9e9 operands differ; e, a select on
# +inf instead of v_div_fixup: other NaN payloads; h / i / k / l, the root by one
# fused Newton step from v_sqrt with v_rcp or v_rsq as slope: 113-561 operands
# and +inf.  None of them ran faster in the denoise kernels than `g`
# (profiles/r05_rsqrt_variants.jsonl): the quotient is no longer what binds.)

# What the lowering emits when SODA_HIP_RSQRT is not set ('off': the program
# as written).  Only a variant with ZERO mismatches in
# profiles/rsqrt_exact.json -- keyed by the hash of its text -- may stand
# here (tests/test_exact.py).
DEFAULT_VARIANT = 'g'


def variant() -> str:
  v = os.environ.get('SODA_HIP_RSQRT', DEFAULT_VARIANT)
  if v != 'off' and v not in VARIANTS:
    raise ValueError('SODA_HIP_RSQRT=%s: off or one of %s' %
                     (v, ', '.join(sorted(VARIANTS))))
  return v


def helper_text(stencil: core.Stencil) -> str:
  """Device functions the kernels of a module over `stencil` call
  (Module.source): the variant `specialize` rewrote the program for -- noted
  on the derived program itself, not read from the environment again."""
  v = getattr(stencil, 'exact_rsqrt_variant', None)
  return VARIANTS[v] if v else ''


def _is_f32(node: ir.Node) -> bool:
  t = node.haoda_type
  return t is not None and t.is_float and t.width_in_bits == 32


def lower_bound(node: ir.Node) -> Optional[float]:
  """A value the fp32 expression `node` cannot fall below whatever its loads
  hold (it may still be +inf or NaN), or None when the text does not give
  one.  Rounding is monotonic, so a bound computed in fp32 holds for the
  rounded sums."""
  if not _is_f32(node):
    return None
  if isinstance(node, ir.Num):
    if not node.literal.lower().endswith('f'):
      return None
    try:
      v = np.float32(node.literal.rstrip('fF'))
    except ValueError:
      return None
    return float(v) if np.isfinite(v) else None
  if isinstance(node, ir.Cast):
    return lower_bound(node.expr)       # float -> float: the value itself
  if isinstance(node, ir.Chain):
    if node.operators == ('*',) and \
        node.operands[0].text() == node.operands[1].text() and \
        _is_f32(node.operands[0]):
      return 0.0                         # a square (expressions are pure)
    if all(op == '+' for op in node.operators):
      acc = np.float32(0.0)
      for operand in node.operands:
        b = lower_bound(operand)
        if b is None or b < 0.0:
          return None
        acc = np.float32(acc + np.float32(b))
      return float(acc)
  return None


def _rewrite(node: ir.Node) -> ir.Node:
  """`1.0f / sqrt(x) ...` with x provably >= REQUIRED_LOWER_BOUND: the first
  two operands of the chain become one backend intrinsic."""
  if not isinstance(node, ir.Chain) or node.operators[0] != '/':
    return node
  one, root = node.operands[0], node.operands[1]
  if not (isinstance(one, ir.Num) and one.literal.lower().endswith('f') and
          _is_f32(one)):
    return node
  try:
    if float(one.literal.rstrip('fF')) != 1.0:
      return node
  except ValueError:
    return node
  if not (isinstance(root, ir.Call) and root.name == 'sqrt' and
          len(root.args) == 1 and _is_f32(root.args[0])):
    return node
  lb = lower_bound(root.args[0])
  if lb is None or lb < REQUIRED_LOWER_BOUND:
    return node
  fused = ir.Call('soda_rsqrt_lb', [root.args[0]])
  if len(node.operands) == 2:
    return fused
  return ir.Chain((fused,) + node.operands[2:], node.operators[1:])


def specialize(stencil: core.Stencil) -> core.Stencil:
  """The program with every provable `1.0f / sqrt(x)` as the intrinsic: the
  stencil itself when there is none (or SODA_HIP_RSQRT=off), else a private
  copy -- the caller's object, which the oracle also reads, is never touched."""
  if variant() == 'off':
    return stencil
  stmts = stencil.local_stmts + stencil.output_stmts
  exprs = [s.expr for s in stmts] + [l.expr for s in stmts for l in s.let]
  if not any(e.transform(_rewrite) is not e for e in exprs):
    return stencil
  derived = copy.deepcopy(stencil)
  derived.exact_rsqrt_variant = variant()
  for s in derived.local_stmts + derived.output_stmts:
    s.expr = s.expr.transform(_rewrite)
    for l in s.let:
      l.expr = l.expr.transform(_rewrite)
  return derived


# ---------------------------------------------------------------------------
# the enumeration
# ---------------------------------------------------------------------------

def text_key(name: str) -> str:
  """Names a variant's TEXT in the enumeration's record."""
  import hashlib
  return hashlib.sha256(VARIANTS[name].encode()).hexdigest()[:16]


def scan_source() -> str:
  """A stand-alone HIP program: every variant above against hipcc's own
  `1.0f / sqrtf(x)` (same -O3 -ffp-contract=off as the kernels) for every
  fp32 x in [2^-96, +inf] and every NaN; one JSON object on stdout."""
  names = sorted(VARIANTS)
  spaces = '\n'.join('namespace v_%s {%s}' % (n, VARIANTS[n]) for n in names)
  calls = ', '.join('v_%s::soda_rsqrt_lb(x)' % n for n in names)
  prints = '\n'.join(
      '  printf("%%s\\"%s\\": {\\"text\\": \\"%s\\", \\"mismatch\\": %%llu, '
      '\\"nan_other_payload\\": %%llu, \\"first\\": \\"0x%%08x\\"}", '
      '%d ? ", " : "", h.bad[%d], h.nan_pair[%d], h.first[%d]);' %
      (n, text_key(n), i, i, i, i) for i, n in enumerate(names))
  return _SCAN % dict(spaces=spaces, calls=calls, nc=len(names), prints=prints,
                      lb='0x1p-96f')


def build_scan(path: str, hipcc: str = '/opt/rocm/bin/hipcc') -> str:
  """hipcc: scan_source() -> an executable (needs no GPU to build)."""
  import subprocess
  os.makedirs(os.path.dirname(path), exist_ok=True)
  src = path + '.hip'
  text = scan_source()
  if os.path.exists(path) and os.path.exists(src) and open(src).read() == text:
    return path
  with open(src, 'w') as f:
    f.write(text)
  if not os.path.exists(hipcc):
    hipcc = 'hipcc'
  tmp = '%s.%d.tmp' % (path, os.getpid())
  proc = subprocess.run([hipcc, '--offload-arch=gfx950', '-O3',
                         '-ffp-contract=off', '-w', '-o', tmp, src],
                        capture_output=True, text=True)
  if proc.returncode != 0:
    raise RuntimeError('building the rsqrt enumeration failed:\n' + proc.stderr)
  os.replace(tmp, path)
  return path


_SCAN = r'''// generated by soda_amd/codegen/hip/exact.py scan_source(): do not edit
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#define SODA_DEV static __device__ inline __attribute__((always_inline))
%(spaces)s
#define NC %(nc)d
struct Tally { unsigned long long bad[NC], nan_pair[NC], cases; uint32_t first[NC]; };
static __device__ inline uint32_t bits_of(float f) { return __builtin_bit_cast(uint32_t, f); }

extern "C" __global__ void scan(Tally* t, uint32_t lb_bits) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long bad[NC] = {0}, nn[NC] = {0}, cases = 0;
  for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
       u < (1ull << 32); u += stride) {
    const uint32_t b = (uint32_t)u;
    const float x = __builtin_bit_cast(float, b);
    const bool is_nan = (b & 0x7fffffffu) > 0x7f800000u;
    if (!is_nan && !(b >= lb_bits && b <= 0x7f800000u)) continue;
    ++cases;
    const float want = 1.0f / sqrtf(x);
    const float got[NC] = {%(calls)s};
    for (int c = 0; c < NC; ++c) {
      if (bits_of(got[c]) == bits_of(want)) continue;
      if (want != want && got[c] != got[c]) { ++nn[c]; continue; }
      if (bad[c]++ == 0) atomicCAS(&t->first[c], 0u, b);
    }
  }
  atomicAdd(&t->cases, cases);
  for (int c = 0; c < NC; ++c) {
    if (bad[c]) atomicAdd(&t->bad[c], bad[c]);
    if (nn[c]) atomicAdd(&t->nan_pair[c], nn[c]);
  }
}

int main() {
  const float lb = %(lb)s;
  uint32_t lb_bits; memcpy(&lb_bits, &lb, 4);
  Tally* d; Tally h;
  if (hipMalloc(&d, sizeof h) != hipSuccess) { puts("{\"error\": \"hipMalloc\"}"); return 2; }
  if (hipMemset(d, 0, sizeof h) != hipSuccess) return 2;
  hipLaunchKernelGGL(scan, dim3(256 * 32), dim3(256), 0, 0, d, lb_bits);
  if (hipDeviceSynchronize() != hipSuccess) { puts("{\"error\": \"kernel\"}"); return 2; }
  if (hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  printf("{\"lower_bound\": \"%%a\", \"cases\": %%llu, \"variants\": {", lb, h.cases);
%(prints)s
  puts("}}");
  return 0;
}
'''
