#!/usr/bin/env python3
"""After the exact `1.0f / sqrt(x)` (codegen/hip/exact.py) the denoise kernels
are no longer bound by their quotient: do the library's shape choices still
stand?  One knob at a time around the defaults, denoise2d 8192^2 and denoise3d
512^3, us per launch (best of 3 x 10), one process.  --compile-only here first
(hiprtc, no GPU), then on the box:
  python tools/experiments/r05_denoise_shapes.py > gpurun_out/r05_denoise_shapes.jsonl"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from soda_amd import core, runtime
from soda_amd.codegen.hip import lower

CASES = {
    'denoise2d': ((8192, 8192), [
        {}, {'prefetch': 4}, {'prefetch': 12}, {'nt_load': True},
        {'tile_rows': 6}, {'tile_rows': 8}, {'chunk_rows': 64},
        {'chunk_rows': 128}, {'chunk_rows': 256}, {'waves_y': 2},
        {'vec': 8, 'reg_budget': 1 << 20}, {'vec': 2}]),
    'denoise3d': ((512, 512, 512), [
        {}, {'xshare': True}, {'tile_rows': 2}, {'tile_rows': 6},
        {'prefetch': 2}, {'vec': 2, 'reg_budget': 1 << 20},
        {'vec': 2, 'xshare': True, 'reg_budget': 1 << 20}]),
}
if '--second' in sys.argv:      # around the two shapes the first pass liked
  CASES = {
      'denoise2d': ((8192, 8192), [
          {}, {'vec': 2}, {'vec': 2, 'prefetch': 4}, {'vec': 2, 'prefetch': 2},
          {'vec': 2, 'prefetch': 12}, {'vec': 2, 'chunk_rows': 48},
          {'vec': 2, 'chunk_rows': 96}, {'vec': 1}, {'prefetch': 4}, {},
          {'vec': 2}]),
      'denoise3d': ((512, 512, 512), [
          {}, {'vec': 2, 'xshare': True, 'reg_budget': 1 << 20}, {},
          {'vec': 2, 'xshare': True, 'reg_budget': 1 << 20}]),
  }

compile_only = '--compile-only' in sys.argv
if not compile_only:
  import torch
  dev = torch.device('cuda', 0)
  stream = torch.cuda.current_stream().cuda_stream

for name, (extent, knobs) in CASES.items():
  st = core.from_file(os.path.join(ROOT, 'tests/golden/soda/%s.soda' % name))
  if not compile_only:
    shape = tuple(extent[::-1])
    ins = [torch.rand(shape, device=dev) for _ in st.input_names]
    outs = [torch.empty(shape, device=dev) for _ in st.output_names]
  for kw in knobs:
    rec = {'program': name, 'knobs': kw}
    try:
      opts = lower.LowerOptions(**kw)
      if compile_only:
        ro = runtime.resolve_options(st, opts, extent)
        mod = lower.lower(st, ro)
        code = runtime.compile_source(mod.source, '%s.hip' % st.app_name)
        k = mod.kernels[0].name
        rec.update(kernel=k, res=runtime.kernel_resources(code).get(k))
      else:
        prog = runtime.Program(st, opts, extent=extent)
        rec['kernel'] = prog.module.kernels[0].name
        args = ([t.data_ptr() for t in outs], [t.data_ptr() for t in ins], extent)
        prog.run_device(*args, stream=stream)
        best = 1e9
        for _ in range(3):
          a, b = runtime.Event(), runtime.Event()
          a.record(stream)
          for _ in range(10):
            prog.run_device(*args, stream=stream)
          b.record(stream)
          torch.cuda.synchronize()
          best = min(best, a.elapsed_ms(b) / 10)
        rec['us_per_launch'] = round(best * 1000, 1)
    except Exception as e:    # noqa
      rec['error'] = '%s: %s' % (type(e).__name__, str(e)[:200])
    print(json.dumps(rec), flush=True)
