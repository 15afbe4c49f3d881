#!/usr/bin/env python3
"""Sustained comparison of denoise2d's two shapes (16 bytes per lane, 8 rows in
flight against 8 bytes, 4 rows): the launch time drifts with the clocks over
the first milliseconds of load (profiles/r05_denoise2d_kernel_trace.txt: 150 ->
215 -> 194 us), so best-of-short-windows is not the last word.  Alternating
blocks of 300 launches, times per 50 launches, one process."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from soda_amd import core, runtime
from soda_amd.codegen.hip import lower
import torch
extent = (8192, 8192)
st = core.from_file(os.path.join(ROOT, 'tests/golden/soda/denoise2d.soda'))
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream().cuda_stream
shape = extent[::-1]
ins = [torch.rand(shape, device=dev) for _ in st.input_names]
outs = [torch.empty(shape, device=dev) for _ in st.output_names]
args = ([t.data_ptr() for t in outs], [t.data_ptr() for t in ins], extent)
progs = {}
for tag, env in (('wide_V4_P8', '0'), ('narrow_V2_P4', '1')):
  os.environ['SODA_HIP_NARROW'] = env
  progs[tag] = runtime.Program(st, lower.LowerOptions(), extent=extent)
out = {t: {'kernel': p.module.kernels[0].name, 'us_per_launch_by_50': []}
       for t, p in progs.items()}
for rnd in range(3):
  for tag, prog in progs.items():
    row = []
    for _ in range(6):
      a, b = runtime.Event(), runtime.Event()
      a.record(stream)
      for _ in range(50):
        prog.run_device(*args, stream=stream)
      b.record(stream)
      torch.cuda.synchronize()
      row.append(round(a.elapsed_ms(b) * 20, 1))
    out[tag]['us_per_launch_by_50'].append(row)
for t in out:
  flat = [x for r in out[t]['us_per_launch_by_50'] for x in r]
  out[t]['mean_us'] = round(sum(flat) / len(flat), 1)
  out[t]['last_round_mean_us'] = round(sum(out[t]['us_per_launch_by_50'][-1]) / 6, 1)
print(json.dumps(out))
