#!/usr/bin/env python3
"""Runs one .soda program on device-resident random data (for rocprofv3):
  python tools/run_program.py heat3d.soda 512 512 512 --iterate 20"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from soda_amd import core, runtime
from soda_amd.codegen.hip import lower

ap = argparse.ArgumentParser()
ap.add_argument('soda')
ap.add_argument('extent', type=int, nargs='+')
ap.add_argument('--iterate', type=int, default=None)
ap.add_argument('--fuse', type=int, nargs='*', default=[])
ap.add_argument('--strategy', default='auto')
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--pipe', type=int, default=None)
ap.add_argument('--border', default=None)
ap.add_argument('--pipe-rows', type=int, default=2)
ap.add_argument('--shift', default='dpp')
ap.add_argument('--chunk', type=int, default=0)
ap.add_argument('--peel', type=int, default=None)
ap.add_argument('--xshare', type=int, default=None)
args = ap.parse_args()
path = args.soda if os.path.exists(args.soda) else os.path.join(ROOT, 'tests/golden/soda', args.soda)
st = core.from_file(path, iterate=args.iterate, border=args.border)
T = {'float32': torch.float32, 'float64': torch.float64, 'uint16': torch.int16, 'int16': torch.int16, 'int32': torch.int32, 'uint8': torch.uint8}
shape = tuple(args.extent[::-1])
dev = torch.device('cuda', 0)
ins = [torch.rand(shape, device=dev, dtype=T[t.np_name]) if T[t.np_name].is_floating_point else torch.randint(0, 200, shape, device=dev, dtype=T[t.np_name]) for t in st.input_types]
outs = [torch.empty(shape, device=dev, dtype=T[t.np_name]) for t in st.output_types]
prog = runtime.Program(st, lower.LowerOptions(strategy=args.strategy, fuse=tuple(args.fuse), pipe=args.pipe, pipe_rows=args.pipe_rows, lane_shift=args.shift, chunk_rows=args.chunk or None, peel=args.peel, xshare=None if args.xshare is None else bool(args.xshare)), extent=args.extent)
s = torch.cuda.current_stream().cuda_stream
a, b = runtime.Event(), runtime.Event()
prog.run_device([t.data_ptr() for t in outs], [t.data_ptr() for t in ins], args.extent, stream=s)
a.record(s)
for _ in range(args.reps):
  prog.run_device([t.data_ptr() for t in outs], [t.data_ptr() for t in ins], args.extent, stream=s)
b.record(s)
print('%s %s iterate=%d: %.3f ms per run, kernels %s' % (st.app_name, args.extent, st.iterate, a.elapsed_ms(b) / args.reps, [k.name for k in prog.module.kernels]))
