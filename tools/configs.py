#!/usr/bin/env python3
"""Measures every BASELINE.json config on ONE GPU (device-resident data, HIP
events on the launch stream) and writes a JSON table.  Multi-GPU configs are
measured as the middle rank's slab of an exchange-free run (what each of the N
GPUs would execute); config 1 is the CPU plumbing case and only reports the
host-array path."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from soda_amd import core, dist as sdist, runtime
from soda_amd.codegen.hip import lower

SODA = os.path.join(ROOT, 'tests', 'golden', 'soda')
TORCH = {'float32': torch.float32, 'uint16': torch.int16, 'int16': torch.int16}


def measure(name, extent, iterate, fuse, world=1, reps=5, label=''):
  st = core.from_file(os.path.join(SODA, name), iterate=iterate)
  every = sdist.auto_exchange_every(st, extent, world, iterate,
                                    multiple_of=max(fuse) if fuse else 1)
  rounds = sdist.rounds(iterate, every)
  st = core.from_file(os.path.join(SODA, name), iterate=every)   # one round
  slab = sdist.Slab(st, extent, world, world // 2, every)
  lext = slab.local_extent
  shape = tuple(lext[::-1])
  dev = torch.device('cuda', 0)
  ins = []
  for t in st.input_types:
    dt = TORCH[t.np_name]
    ins.append(torch.rand(shape, device=dev, dtype=dt) if dt.is_floating_point
               else torch.randint(0, 30000, shape, device=dev, dtype=dt))
  outs = [torch.empty(shape, device=dev, dtype=TORCH[t.np_name])
          for t in st.output_types]
  prog = runtime.Program(st, lower.LowerOptions(fuse=fuse), extent=lext,
                         calibrate=True)
  stream = torch.cuda.current_stream().cuda_stream

  def go():
    prog.run_device([t.data_ptr() for t in outs], [t.data_ptr() for t in ins],
                    lext, stream=stream, keep=slab.keep)

  go()
  go()
  a, b = runtime.Event(), runtime.Event()
  a.record(stream)
  for _ in range(reps):
    go()
  b.record(stream)
  ms = a.elapsed_ms(b) / reps * rounds   # compute only; exchanges not included
  table = st.symbol_table
  bpc = (sum(table[n].size_in_bytes for n in st.input_names) +
         sum(table[n].size_in_bytes for n in st.output_names))
  cells = float(np.prod(extent))
  local = float(np.prod(lext))
  launches = prog.last_launches()[0] * rounds
  res = dict(config=label, program=name, extent=list(extent), iterate=iterate,
             n_gpus=world, local_extent=list(lext), ms=ms, launches=launches,
             exchange_every=every, exchanges=rounds - 1,
             kernels=sorted({k.name for k in prog.module.kernels}),
             schedule=prog.schedule(lext, every),
             pass_us={t: round(v, 1) for t, v in prog.pass_times(lext)[0].items()},
             cells_iters_per_s_job=cells * iterate / (ms * 1e-3),
             algorithmic_GBs_per_gpu=local * bpc * launches / (ms * 1e-3) / 1e9
             if all(p.kind != 'direct' for p in prog.module.passes) else None)
  prog.close()
  return res


def main():
  out = []
  out.append(measure('blur.soda', (2000, 1024), 1, (), label='C1 blur 2000x1024 (GPU run of the CPU plumbing case)', reps=50))
  out.append(measure('jacobi2d.soda', (8192, 8192), 100, (), label='C2 jacobi2d 8192^2 it=100, one iteration per launch'))
  out.append(measure('jacobi2d.soda', (8192, 8192), 100, (13, 12, 8, 4), label='C2 jacobi2d 8192^2 it=100, T=12 fused'))
  out.append(measure('blur.soda', (16384, 16384), 1, (), label='C3 blur 16384^2 fused two-stage', reps=20))
  out.append(measure('heat3d.soda', (512, 512, 512), 50, (), label='C4 heat3d 512^3 it=50, one iteration per launch, 1 GPU'))
  out.append(measure('heat3d.soda', (512, 512, 512), 50, (2,), label='C4 heat3d 512^3 it=50, T=2 fused, 1 GPU'))
  out.append(measure('heat3d.soda', (512, 512, 512), 50, (2,), world=8, label='C4 heat3d 512^3 it=50, T=2, slab of an 8-GPU run (compute only, exchanges not timed)'))
  out.append(measure('jacobi2d.soda', (8192, 8192), 1000, (13, 12, 8, 4), label='C5 jacobi2d 8192^2 it=1000, T=12 fused, 1 GPU', reps=2))
  out.append(measure('jacobi2d.soda', (8192, 8192), 100, (13, 12, 8, 4), world=8, label='C2 jacobi2d 8192^2 it=100, slab of an 8-GPU run (exchange-free)'))
  out.append(measure('jacobi2d.soda', (8192, 8192), 1000, (13, 12, 8, 4), world=8, label='C5 jacobi2d 8192^2 it=1000, slab of an 8-GPU run (compute only, exchanges not timed)', reps=2))
  out.append(measure('jacobi2d.soda', (8192, 8192), 1000, (4,), label='C5 jacobi2d 8192^2 it=1000, T=4 fused (as BASELINE words it), 1 GPU', reps=2))
  for r in out:
    print(json.dumps(r))
  path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'configs.json')
  with open(path, 'w') as f:
    json.dump(out, f, indent=1)


if __name__ == '__main__':
  main()
