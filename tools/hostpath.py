#!/usr/bin/env python3
"""PCIe-inclusive time of the operator-level entry on HOST arrays (Program.run =
soda_hip_run_host_box, the soda::app::<app>() analogue; reference
src/soda/codegen/frt/host.py:62-88,319-322): copies in, all iterations, copies
the valid box out.  Never the bench value (inputs are resident there).

Per workload: `reused` -- the same arrays call after call (pages present,
what a caller in a loop sees); `fresh` -- new input and output arrays for every
call (the output's pages untouched until the library writes the valid box:
their page faults are inside the timed call)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from soda_amd import core, runtime
from soda_amd.codegen.hip import lower

SODA = os.path.join(ROOT, 'tests', 'golden', 'soda')


def measure(label, name, extent, iterate, fuse, reps=3, check=True):
  st = core.from_file(os.path.join(SODA, name), iterate=iterate)
  rng = np.random.default_rng(0)
  shape = extent[::-1]

  def make_inputs():
    out = {}
    for n, t in zip(st.input_names, st.input_types):
      dt = np.dtype(t.np_name)
      out[n] = (rng.random(shape, dtype=np.float32) if dt.kind == 'f'
                else rng.integers(0, 20000, shape).astype(dt))
    return out

  ins = make_inputs()
  outs = {n: np.zeros(shape, dtype=np.dtype(t.np_name))
          for n, t in zip(st.output_names, st.output_types)}
  res = {'workload': label, 'extent': list(extent), 'iterate': iterate,
         'bytes_in': int(sum(a.nbytes for a in ins.values())),
         'bytes_out': int(sum(a.nbytes for a in outs.values())),
         'host_threads': int(os.environ.get('SODA_HIP_HOST_THREADS', 8)),
         'chunk_MiB': int(os.environ.get('SODA_HIP_HOST_CHUNK_MB', 16))}
  with runtime.Program(st, lower.LowerOptions(fuse=fuse), extent=extent) as prog:
    t0 = time.perf_counter()
    prog.run(ins, outputs=outs)            # JIT load, calibration, staging
    res['first_call_ms'] = (time.perf_counter() - t0) * 1e3
    ts = []
    for _ in range(reps):
      t0 = time.perf_counter()
      prog.run(ins, outputs=outs)
      ts.append((time.perf_counter() - t0) * 1e3)
    res['reused_ms'] = min(ts)
    res['reused_ms_all'] = [round(t, 2) for t in ts]
    if check:
      # the device-resident path on the same input: bit for bit on the box
      import torch
      dev = [torch.from_numpy(ins[n].view(np.int16) if ins[n].dtype == np.uint16
                              else ins[n]).cuda() for n in st.input_names]
      dout = [torch.empty_like(dev[0]) for _ in st.output_names]
      prog.run_device([t.data_ptr() for t in dout],
                      [t.data_ptr() for t in dev], extent)
      torch.cuda.synchronize()
      bad = 0
      for n, t in zip(st.output_names, dout):
        lo, hi = st.valid_box(extent, n)
        idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
        got = outs[n][idx]
        want = t.cpu().numpy().view(got.dtype)[idx]
        bad += int((got.view(np.uint8) != want.view(np.uint8)).sum())
      res['bytes_differing_from_the_device_path'] = bad
      del dev, dout
    ts = []
    for _ in range(reps):
      fresh_in = {n: a.copy() for n, a in ins.items()}
      fresh_out = {n: np.zeros(shape, dtype=a.dtype) for n, a in outs.items()}
      t0 = time.perf_counter()
      prog.run(fresh_in, outputs=fresh_out)
      ts.append((time.perf_counter() - t0) * 1e3)
    res['fresh_ms'] = min(ts)
    res['fresh_ms_all'] = [round(t, 2) for t in ts]
    # the same values in page-aligned memory registered with the GPU
    # (runtime.PinnedBuffer = soda_hip_host_register, as a host that keeps its
    # arrays across calls would): DMA from / to where they are, no staging
    # slots, no worker threads
    def page(n):
      return -(-n // 4096) * 4096
    sizes = [page(a.nbytes) for a in list(ins.values()) + list(outs.values())]
    t0 = time.perf_counter()
    with runtime.PinnedBuffer(sum(sizes)) as buf:
      res['allocate_and_register_ms'] = (time.perf_counter() - t0) * 1e3
      before = {n: a.copy() for n, a in outs.items()}
      at, pin_in, pin_out = 0, {}, {}
      for n, a in ins.items():
        pin_in[n] = buf.array(a.shape, a.dtype, at)
        pin_in[n][...] = a
        at += page(a.nbytes)
      for n, a in outs.items():
        pin_out[n] = buf.array(a.shape, a.dtype, at)
        pin_out[n][...] = 0
        at += page(a.nbytes)
      ts = []
      for _ in range(reps):
        t0 = time.perf_counter()
        prog.run(pin_in, outputs=pin_out)
        ts.append((time.perf_counter() - t0) * 1e3)
      outs = pin_out
      res['pinned_ms'] = min(ts)
      res['pinned_ms_all'] = [round(t, 2) for t in ts]
      bad = 0
      for n in st.output_names:
        lo, hi = st.valid_box(extent, n)
        idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
        bad += int((outs[n][idx].view(np.uint8) !=
                    before[n][idx].view(np.uint8)).sum())
      res['pinned_bytes_differing_from_the_pageable_run'] = bad
  cells = float(np.prod(extent))
  res['cells_iters_per_s_incl_pcie'] = cells * iterate / (res['reused_ms'] * 1e-3)
  res['GBs_moved_reused'] = (res['bytes_in'] + res['bytes_out']) / (
      res['reused_ms'] * 1e-3) / 1e9
  return res


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--only', default=None)
  args = ap.parse_args()
  rows = []
  if args.only in (None, 'c2'):
    rows.append(measure('C2 jacobi2d 8192x8192 iterate=100 via host arrays',
                        'jacobi2d.soda', (8192, 8192), 100,
                        lower.DEFAULT_FUSE))
  if args.only in (None, 'c3'):
    rows.append(measure('C3 blur 16384x16384 via host arrays', 'blur.soda',
                        (16384, 16384), 1, ()))
  if args.only in (None, 'c4'):
    rows.append(measure('C4 heat3d 512^3 iterate=50 via host arrays',
                        'heat3d.soda', (512, 512, 512), 50, (2,)))
  for r in rows:
    print(json.dumps(r))


if __name__ == '__main__':
  main()
