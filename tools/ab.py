#!/usr/bin/env python3
"""Sustained A/B (/C ...) of backend option sets for one program: the shapes
alternate in blocks of launches inside ONE process, times are reported per
sub-block, so clock drift over the first milliseconds of load shows instead of
deciding (a cold 31-launch process of denoise2d drifts 150 -> 215 -> 194 us,
profiles/r05_denoise2d_kernel_trace.txt; best-of-short-windows flatters).

  here (hiprtc, no GPU):  python tools/ab.py denoise2d.soda 8192 8192 \\
        --arm '{}' --arm '{"vec": 4, "prefetch": 8}' --compile-only
  on the box:             ... the same without --compile-only  (~2 s of GPU)

An arm is a JSON object of LowerOptions keywords; a key "env" holds
environment variables set while that arm's program is built
(e.g. {"env": {"SODA_HIP_RSQRT": "off"}}).  One JSON object on stdout.
(Written when round 5's GPU minutes were spent: the measuring loop is that of
tools/experiments/r05_denoise_sustained.py, which ran; --compile-only ran here.)"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('soda')
  ap.add_argument('extent', type=int, nargs='+')
  ap.add_argument('--arm', action='append', required=True,
                  help='JSON object of LowerOptions keywords (+ "env")')
  ap.add_argument('--iterate', type=int, default=None)
  ap.add_argument('--rounds', type=int, default=3)
  ap.add_argument('--blocks', type=int, default=6, help='sub-blocks per round')
  ap.add_argument('--launches', type=int, default=50, help='launches per sub-block')
  ap.add_argument('--compile-only', action='store_true')
  args = ap.parse_args()
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  path = args.soda if os.path.exists(args.soda) else \
      os.path.join(ROOT, 'tests/golden/soda', args.soda)
  st = core.from_file(path, iterate=args.iterate)
  extent = tuple(args.extent)
  arms = [json.loads(a) for a in args.arm]

  class _Env:
    def __init__(self, env):
      self.env, self.old = env or {}, {}

    def __enter__(self):
      for k, v in self.env.items():
        self.old[k] = os.environ.get(k)
        os.environ[k] = str(v)

    def __exit__(self, *exc):
      for k, v in self.old.items():
        if v is None:
          os.environ.pop(k, None)
        else:
          os.environ[k] = v

  out = {'program': st.app_name, 'extent': list(extent),
         'iterate': st.iterate, 'arms': []}
  if args.compile_only:
    for arm in arms:
      kw = {k: v for k, v in arm.items() if k != 'env'}
      with _Env(arm.get('env')):
        opts = runtime.resolve_options(st, lower.LowerOptions(**kw), extent)
        mod = lower.lower(st, opts)
        code = runtime.compile_source(mod.source, '%s.hip' % st.app_name)
      res = runtime.kernel_resources(code)
      out['arms'].append({'arm': arm, 'kernels': {
          k.name: res.get(k.name) for k in mod.kernels}})
    print(json.dumps(out))
    return
  import torch
  dev = torch.device('cuda', 0)
  stream = torch.cuda.current_stream().cuda_stream
  tdt = {'float32': torch.float32, 'float64': torch.float64,
         'uint16': torch.int16, 'int16': torch.int16, 'int32': torch.int32,
         'uint8': torch.uint8, 'int8': torch.int8}
  shape = extent[::-1]

  def field(t, rand):
    dt = tdt[t.np_name]
    if not rand:
      return torch.empty(shape, device=dev, dtype=dt)
    if dt.is_floating_point:
      return torch.rand(shape, device=dev, dtype=dt)
    return torch.randint(0, 100, shape, device=dev, dtype=dt)

  ins = [field(t, True) for t in st.input_types]
  outs = [field(t, False) for t in st.output_types]
  call = ([t.data_ptr() for t in outs], [t.data_ptr() for t in ins], extent)
  progs = []
  for arm in arms:
    kw = {k: v for k, v in arm.items() if k != 'env'}
    with _Env(arm.get('env')):
      prog = runtime.Program(st, lower.LowerOptions(**kw), extent=extent)
    progs.append(prog)
    out['arms'].append({'arm': arm, 'kernels': [k.name for k in prog.module.kernels],
                        'us_per_run_by_block': []})
    prog.run_device(*call, stream=stream)
  torch.cuda.synchronize()
  for _ in range(args.rounds):
    for prog, rec in zip(progs, out['arms']):
      row = []
      for _ in range(args.blocks):
        a, b = runtime.Event(), runtime.Event()
        a.record(stream)
        for _ in range(args.launches):
          prog.run_device(*call, stream=stream)
        b.record(stream)
        torch.cuda.synchronize()
        row.append(round(a.elapsed_ms(b) * 1000 / args.launches, 1))
      rec['us_per_run_by_block'].append(row)
  for rec in out['arms']:
    flat = [x for r in rec['us_per_run_by_block'] for x in r]
    rec['mean_us'] = round(sum(flat) / len(flat), 1)
    last = rec['us_per_run_by_block'][-1]
    rec['last_round_mean_us'] = round(sum(last) / len(last), 1)
  print(json.dumps(out))


if __name__ == '__main__':
  main()
