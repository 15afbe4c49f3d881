#!/usr/bin/env python3
"""Where and when did every wavefront of a marching kernel run?

Builds the dominant pass of a program with time stamps (`stamps=True`: each
wave records s_memtime at entry and exit plus HW_ID / XCC_ID), launches it once
on warm clocks and prints what the launch looked like from the inside:

  * span of the launch (first entry -> last exit), wave lifetimes;
  * occupancy actually reached: time-averaged resident waves per SIMD, the
    largest number of waves any SIMD held at once, waves per SIMD;
  * when waves started and ended (deciles): a second round of waves or a long
    tail shows up here.

  python tools/timeline.py --fuse 12 [--pipe 4] [--chunk 147] [--shift swzh]
"""
import argparse
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--soda', default=os.path.join(ROOT, 'tests/golden/soda/jacobi2d.soda'))
  ap.add_argument('--extent', type=int, nargs='+', default=[8192, 8192])
  ap.add_argument('--fuse', type=int, default=12)
  ap.add_argument('--chunk', type=int, default=0)
  ap.add_argument('--prefetch', type=int, default=None)
  ap.add_argument('--pipe', type=int, default=1)
  ap.add_argument('--pipe-rows', type=int, default=2)
  ap.add_argument('--shift', default='dpp')
  ap.add_argument('--vec', type=int, default=None)
  ap.add_argument('--tag', default='')
  ap.add_argument('--out', default=None)
  args = ap.parse_args()
  import numpy as np
  import torch
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  dev = torch.device('cuda', 0)
  st = core.from_file(args.soda, iterate=args.fuse)
  opts = lower.LowerOptions(fuse=(args.fuse,) if args.fuse > 1 else (),
                            chunk_rows=args.chunk or None,
                            prefetch=args.prefetch, pipe=args.pipe,
                            pipe_rows=args.pipe_rows, lane_shift=args.shift,
                            vec=args.vec, 
                            stamps=True)
  prog = runtime.Program(st, opts, extent=args.extent)
  shape = tuple(args.extent[::-1])
  tdt = {'float32': torch.float32, 'uint16': torch.int16, 'int16': torch.int16}
  ins = [torch.rand(shape, device=dev).to(tdt[t.np_name]) for t in st.input_types]
  outs = [torch.empty(shape, device=dev, dtype=tdt[t.np_name]) for t in st.output_types]
  dbg = torch.zeros(1 << 22, device=dev, dtype=torch.int64)     # 32 MiB
  prog.set_debug_buffer(dbg.data_ptr())
  stream = torch.cuda.current_stream().cuda_stream

  def go():
    prog.run_device([t.data_ptr() for t in outs], [t.data_ptr() for t in ins],
                    args.extent, iterate=args.fuse, stream=stream)

  for _ in range(30):          # warm clocks
    go()
  torch.cuda.synchronize()
  dbg.zero_()
  a, b = runtime.Event(), runtime.Event()
  a.record(stream)
  go()
  b.record(stream)
  ms = a.elapsed_ms(b)
  torch.cuda.synchronize()
  raw = dbg.cpu().numpy().reshape(-1, 4)
  raw = raw[raw[:, 1] != 0]
  t0, t1, hw, xcc = raw[:, 0].copy(), raw[:, 1].copy(), raw[:, 2], raw[:, 3] & 0xf
  # s_memtime counts from a different origin on every XCD: line the XCDs up at
  # their first wave's entry (the dispatcher starts all eight within ~1 us)
  for x in np.unique(xcc):
    m = xcc == x
    base = t0[m].min()
    t0[m] -= base
    t1[m] -= base
  begin = t0.min()
  span = float(t1.max() - begin)
  life = (t1 - t0).astype(np.float64)
  # gfx9 HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
  simd = (hw >> 4) & 3
  cu = (hw >> 8) & 0xf
  sh = (hw >> 12) & 1
  se = (hw >> 13) & 7
  key = ((xcc * 8 + se) * 2 + sh) * 64 + cu * 4 + simd
  per = collections.Counter(key.tolist())
  # largest number of waves a SIMD held at once (sweep line per SIMD)
  peak = collections.Counter()
  order = np.argsort(t0, kind='stable')
  events = collections.defaultdict(list)
  for i in order:
    events[int(key[i])].append((int(t0[i]), 1))
    events[int(key[i])].append((int(t1[i]), -1))
  for k, ev in events.items():
    cur = best = 0
    for _, d in sorted(ev):
      cur += d
      best = max(best, cur)
    peak[k] = best
  q = lambda v, p: float(np.percentile(v, p))
  kernel = prog.module.sorted_passes()[0].kernels[0]
  res = prog.resources.get(prog.module.kernels[kernel].name, {})
  out = dict(
      tag=args.tag, kernel=prog.module.kernels[kernel].name,
      vgpr=res.get('vgpr'), tile=list(prog.module.kernels[kernel].tile[:2]),
      event_us=ms * 1e3, waves=int(len(raw)), simds_used=len(per),
      span_ticks=span, ticks_per_us=span / (ms * 1e3),
      life_ticks=dict(min=float(life.min()), median=q(life, 50),
                      max=float(life.max())),
      life_over_span=dict(min=life.min() / span, p10=q(life, 10) / span,
                          median=q(life, 50) / span, p90=q(life, 90) / span,
                          max=life.max() / span),
      mean_resident_waves_per_simd=float(life.sum() / span / max(1, len(per))),
      waves_per_simd=dict(collections.Counter(per.values())),
      peak_waves_per_simd=dict(collections.Counter(peak.values())),
      start_deciles=[round(q(t0 - begin, p) / span, 3) for p in range(0, 101, 10)],
      end_deciles=[round(q(t1 - begin, p) / span, 3) for p in range(0, 101, 10)],
  )
  print(json.dumps(out))
  if args.out:
    with open(args.out, 'a') as f:
      f.write(json.dumps(out) + '\n')


if __name__ == '__main__':
  main()
