#!/usr/bin/env python3
"""Every program of the reference corpus (tests/golden/soda = reference
tests/src) on device-resident random data: time per iteration, algorithmic
GB/s (bytes of the program's inputs + outputs per cell), kernel family.
One JSON line per program; `--strategy direct` gives the fallback's numbers."""
import argparse, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from soda_amd import core, runtime
from soda_amd.codegen.hip import lower

T = {'float32': torch.float32, 'float64': torch.float64, 'uint16': torch.int16,
     'int16': torch.int16, 'int32': torch.int32, 'uint8': torch.uint8,
     'int8': torch.int8, 'uint32': torch.int32, 'int64': torch.int64}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--strategy', default='auto')
  ap.add_argument('--extent2', type=int, nargs=2, default=[8192, 8192])
  ap.add_argument('--extent3', type=int, nargs=3, default=[512, 512, 512])
  ap.add_argument('--reps', type=int, default=5)
  ap.add_argument('--only', nargs='*', default=[])
  ap.add_argument('--vec', type=int, default=None)
  ap.add_argument('--prefetch', type=int, default=None)
  ap.add_argument('--no-windows', action='store_true')
  ap.add_argument('--no-inline', action='store_true')
  ap.add_argument('--tile-rows', type=int, default=None)
  ap.add_argument('--chunk', type=int, default=None)
  ap.add_argument('--waves-y', type=int, default=1)
  ap.add_argument('--reg-budget', type=int, default=None)
  ap.add_argument('--out', default=None)
  args = ap.parse_args()
  dev = torch.device('cuda', 0)
  s = torch.cuda.current_stream().cuda_stream
  for path in sorted(glob.glob(os.path.join(ROOT, 'tests/golden/soda/*.soda'))):
    name = os.path.basename(path)
    if args.only and name not in args.only:
      continue
    st = core.from_file(path)
    extent = args.extent2 if st.dim == 2 else args.extent3
    shape = tuple(extent[::-1])
    try:
      prog = runtime.Program(
          st, lower.LowerOptions(strategy=args.strategy, vec=args.vec,
                                 prefetch=args.prefetch,
                                 reg_budget=args.reg_budget,
                                 windows=False if args.no_windows else None,
                                 inline=False if args.no_inline else None,
                                 tile_rows=args.tile_rows,
                                 chunk_rows=args.chunk, waves_y=args.waves_y),
          extent=extent)
    except Exception as e:   # noqa
      print(json.dumps(dict(program=name, error=str(e)[:200])), flush=True)
      continue
    ins = [torch.rand(shape, device=dev, dtype=T[t.np_name])
           if T[t.np_name].is_floating_point else
           torch.randint(0, 200, shape, device=dev, dtype=T[t.np_name])
           for t in st.input_types]
    outs = [torch.empty(shape, device=dev, dtype=T[t.np_name])
            for t in st.output_types]

    def go():
      prog.run_device([t.data_ptr() for t in outs],
                      [t.data_ptr() for t in ins], extent, stream=s)
    go()
    a, b = runtime.Event(), runtime.Event()
    a.record(s)
    for _ in range(args.reps):
      go()
    b.record(s)
    ms = a.elapsed_ms(b) / args.reps
    cells = 1
    for e in extent:
      cells *= e
    bytes_cell = sum(t.size_in_bytes for t in st.input_types) + sum(
        t.size_in_bytes for t in st.output_types)
    launches = prog.last_launches()[0]
    row = dict(
        program=name, extent=extent, iterate=st.iterate,
        kernels=[k.name for k in prog.module.kernels],
        vgprs=[prog.resources.get(k.name, {}).get('vgpr')
               for k in prog.module.kernels],
        stages=len(st.local_stmts) + len(st.output_stmts),
        families=sorted({p.kind for p in prog.module.passes}),
        launches=launches, ms_per_run=round(ms, 4),
        us_per_iteration=round(ms * 1e3 / st.iterate, 1),
        algorithmic_bytes_per_cell_iter=bytes_cell,
        algorithmic_GBs=round(cells * bytes_cell * st.iterate / ms / 1e6, 1),
        cells_iters_per_s=cells * st.iterate / ms * 1e3)
    print(json.dumps(row), flush=True)
    if args.out:
      with open(args.out, 'a') as f:
        f.write(json.dumps(row) + '\n')
    prog.close()
    del ins, outs


if __name__ == '__main__':
  main()
