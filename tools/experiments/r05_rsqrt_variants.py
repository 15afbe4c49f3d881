#!/usr/bin/env python3
"""denoise2d / denoise3d with `1.0f / sqrt(x)` evaluated by each variant of
soda_amd/codegen/hip/exact.py: time per launch and bits against (i) the
program as written on the GPU, at full size, on the reference's kind of input
(uniform [0, 1)) and on inputs spread over 90 binades with overflowing cells,
(ii) the C oracle on a smaller grid.  (The run recorded in
profiles/r05_rsqrt_variants.jsonl also had the refuted candidates b, e, f, which
exact.py no longer carries.)  One process, one JSON line per case into
gpurun_out/r05_rsqrt_variants.jsonl.  Pre-JIT here first (--compile-only)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument('--compile-only', action='store_true')
ap.add_argument('--variants', nargs='+', default=['off', 'c', 'd', 'g'])
ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out/r05_rsqrt_variants.jsonl'))
args = ap.parse_args()

from soda_amd import core, runtime
from soda_amd.codegen.hip import lower

CASES = [('denoise2d', (8192, 8192), (2048, 1500)),
         ('denoise3d', (512, 512, 512), (160, 96, 40))]


def stencil_of(name):
  return core.from_file(os.path.join(ROOT, 'tests/golden/soda/%s.soda' % name))


if args.compile_only:
  for name, extent, small in CASES:
    st = stencil_of(name)
    for v in args.variants:
      os.environ['SODA_HIP_RSQRT'] = v
      for ext in (extent, small):
        t = time.time()
        opts = runtime.resolve_options(st, lower.LowerOptions(), ext)
        mod = lower.lower(st, opts)
        code = runtime.compile_source(mod.source, '%s.hip' % st.app_name)
        k = mod.kernels[0].name
        print(name, v, ext, k, runtime.kernel_resources(code).get(k),
              '%.1fs' % (time.time() - t), flush=True)
  sys.exit(0)

import torch
from oracle import c_oracle

dev = torch.device('cuda', 0)
os.makedirs(os.path.dirname(args.out), exist_ok=True)
log = open(args.out, 'a')


def emit(rec):
  line = json.dumps(rec)
  print(line, flush=True)
  log.write(line + '\n')
  log.flush()


def fields(shape, kind, n, seed):
  g = torch.Generator(device=dev)
  g.manual_seed(seed)
  out = []
  for _ in range(n):
    u = torch.rand(shape, generator=g, device=dev, dtype=torch.float32)
    if kind == 'wide':
      e = torch.rand(shape, generator=g, device=dev) * 90.0 - 45.0
      s = torch.where(torch.rand(shape, generator=g, device=dev) < 0.5, -1.0, 1.0)
      u = s * (1.0 + u) * torch.exp2(e.floor())
      big = torch.rand(shape, generator=g, device=dev) < 1e-4
      u = torch.where(big, torch.full_like(u, 1e25), u)   # squares overflow
      u = u.to(torch.float32)
    out.append(u.contiguous())
  return out


def bits(t):
  return t.view(torch.int32)


for name, extent, small in CASES:
  st = stencil_of(name)
  shape = tuple(extent[::-1])
  n_in = len(st.input_names)
  inputs = {k: fields(shape, k, n_in, 7) for k in ('uniform', 'wide')}
  small_in = {n: (np.random.RandomState(3 + i).rand(*small[::-1])
                  .astype(np.float32)) for i, n in enumerate(st.input_names)}
  t0 = time.time()
  want_small = c_oracle.COracle(st).run(small_in)
  oracle_s = time.time() - t0
  base = {}
  stream = torch.cuda.current_stream().cuda_stream
  for v in args.variants:
    os.environ['SODA_HIP_RSQRT'] = v
    rec = {'program': name, 'extent': list(extent), 'variant': v}
    try:
      prog = runtime.Program(st, lower.LowerOptions(), extent=extent)
      rec['kernel'] = prog.module.kernels[0].name
      for kind in ('uniform', 'wide'):
        out = [torch.zeros(shape, device=dev, dtype=torch.float32)
               for _ in st.output_names]
        ins = inputs[kind]
        prog.run_device([t.data_ptr() for t in out],
                        [t.data_ptr() for t in ins], extent, stream=stream)
        torch.cuda.synchronize()
        if kind == 'uniform':
          best = 1e9
          for _ in range(3):
            a, b = runtime.Event(), runtime.Event()
            a.record(stream)
            for _ in range(args.reps):
              prog.run_device([t.data_ptr() for t in out],
                              [t.data_ptr() for t in ins], extent, stream=stream)
            b.record(stream)
            torch.cuda.synchronize()
            best = min(best, a.elapsed_ms(b) / args.reps)
          rec['us_per_launch'] = round(best * 1000, 1)
        if v == args.variants[0]:
          base[kind] = [t.clone() for t in out]
          rec['nan_cells_' + kind] = int(sum(torch.isnan(t).sum() for t in out))
          rec['zero_g_probe_' + kind] = int(sum((t == 0).sum() for t in out))
        else:
          bad = nanpay = 0
          for t, w in zip(out, base[kind]):
            diff = bits(t) != bits(w)
            both_nan = torch.isnan(t) & torch.isnan(w)
            bad += int((diff & ~both_nan).sum())
            nanpay += int((diff & both_nan).sum())
          rec['mismatch_vs_%s_%s' % (args.variants[0], kind)] = bad
          rec['nan_other_payload_' + kind] = nanpay
      # the C oracle on the small grid, through the numpy entry
      sprog = runtime.Program(st, lower.LowerOptions(), extent=small)
      got = sprog.run(small_in)
      bad = 0
      for o in st.output_names:
        g_, w_ = got[o], want_small[o]
        lo_, hi_ = st.valid_box(small, o)     # by bits, where it is defined
        sl = tuple(slice(l, h) for l, h in zip(lo_[::-1], hi_[::-1]))
        bad += int((g_[sl].view(np.int32) != w_[sl].view(np.int32)).sum())
        rec['oracle_cells'] = int(g_[sl].size)
      rec['mismatch_vs_c_oracle_small'] = bad
      rec['oracle_s'] = round(oracle_s, 2)
    except Exception as e:      # noqa
      rec['error'] = '%s: %s' % (type(e).__name__, str(e)[:300])
    emit(rec)
