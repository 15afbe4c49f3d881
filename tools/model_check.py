#!/usr/bin/env python3
"""Launch-time model of the library (soda_hip_plan_geometry) against the clock:
for every pass of a program and a list of extents, the modelled and the
measured time of one launch (HIP events, arrays rotating as in an iterated
run), and the schedule the library picks for `--iterate`.  One JSON line per
extent."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--soda', default=os.path.join(ROOT, 'tests/golden/soda/jacobi2d.soda'))
  ap.add_argument('--fuse', type=int, nargs='+', default=[12, 10, 8, 4])
  ap.add_argument('--iterate', type=int, default=100)
  ap.add_argument('--extents', nargs='+', default=['8192x8192', '8192x4296', '8192x2248', '8192x1224'])
  ap.add_argument('--launches', type=int, default=8)
  ap.add_argument('--reps', type=int, default=5)
  ap.add_argument('--out', default=None)
  args = ap.parse_args()
  import torch
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  dev = torch.device('cuda', 0)
  stream = torch.cuda.current_stream().cuda_stream
  tdt = {'float32': torch.float32, 'uint16': torch.int16, 'int16': torch.int16}
  for text in args.extents:
    extent = [int(v) for v in text.split('x')]
    shape = tuple(extent[::-1])
    row = dict(extent=extent, passes={})
    st_all = core.from_file(args.soda, iterate=args.iterate)
    with runtime.Program(st_all, lower.LowerOptions(fuse=tuple(args.fuse)),
                         extent=extent, calibrate=False) as prog:
      tiles, model = prog.geometry(extent)
      row['schedule'] = prog.schedule(extent, args.iterate)
      row['model_total_us'] = sum(model[t] * c for t, c in row['schedule'].items())
      ins = [torch.rand(shape, device=dev).to(tdt[t.np_name]) for t in st_all.input_types]
      outs = [torch.empty_like(t) for t in ins]

      def whole():
        prog.run_device([t.data_ptr() for t in outs], [t.data_ptr() for t in ins],
                        extent, iterate=args.iterate, stream=stream)

      whole()
      a, b = runtime.Event(), runtime.Event()
      a.record(stream)
      for _ in range(args.reps):
        whole()
      b.record(stream)
      row['measured_total_us'] = a.elapsed_ms(b) / args.reps * 1e3
    for t in sorted(set(args.fuse) | {1}, reverse=True):
      st = core.from_file(args.soda, iterate=t * args.launches)
      with runtime.Program(st, lower.LowerOptions(fuse=(t,) if t > 1 else ()),
                           extent=extent, calibrate=False) as prog:
        tiles, model = prog.geometry(extent)

        def go():
          prog.run_device([x.data_ptr() for x in outs], [x.data_ptr() for x in ins],
                          extent, iterate=t * args.launches, stream=stream)

        go()
        best = 1e9
        for _ in range(3):
          a, b = runtime.Event(), runtime.Event()
          a.record(stream)
          for _ in range(args.reps):
            go()
          b.record(stream)
          best = min(best, a.elapsed_ms(b) / args.reps / args.launches * 1e3)
        kk = [k for k in prog.module.kernels if k.tune and k.tune.get('fused') == t][0]
        name = kk.name
        idx = prog.module.kernels.index(kk)
        d = prog.plan.kernels[idx]
        row['passes'][t] = dict(model_us=round(model[t], 1), measured_us=round(best, 1),
                                tile=tiles[name], vgpr=prog.resources[name]['vgpr'],
                                desc=dict(warm=d.warm, warm_saved=d.warm_saved,
                                          step_ns=d.step_ns, lanes=d.lane_redundancy,
                                          bytes_per_cell=d.bytes_per_cell,
                                          block=list(d.block), vec=d.vec,
                                          waves_along=d.waves_along, pipe=d.pipe,
                                          strip_cells=kk.tile[0],
                                          shift=(kk.tune or {}).get('lane_shift'),
                                          ns_per_op=runtime.NS_PER_VALU_OP,
                                          step_ops=(kk.tune or {}).get('step_ops')))
    print(json.dumps(row), flush=True)
    if args.out:
      with open(args.out, 'a') as f:
        f.write(json.dumps(row) + '\n')


if __name__ == '__main__':
  main()
