#!/usr/bin/env python3
"""Soak of the host-array entry (soda_host.cpp): random programs, extents,
strides, staging chunk sizes and thread counts, every run compared bit for
bit with the device-resident path of the same program on the valid box, and
the caller's array outside the box checked untouched.  Looks for what single
tests do not: a slot re-used while its DMA is in flight, a band launched
before its rows arrived, a fetch overtaking its kernels.

  python tools/experiments/r05_host_soak.py [--trials 150] [--seed 0]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--trials', type=int, default=150)
  ap.add_argument('--seed', type=int, default=0)
  ap.add_argument('--out', default=None)
  ap.add_argument('--verbose', action='store_true',
                  help='print every trial\'s draw before running it')
  ap.add_argument('--dry', action='store_true',
                  help='print every trial\'s draw, run nothing (no GPU needed)')
  args = ap.parse_args()
  import torch
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  soda = os.path.join(ROOT, 'tests', 'golden', 'soda')
  golden = os.path.join(ROOT, 'tests', 'golden')
  rng = np.random.default_rng(args.seed)
  menu = [
      # file, dim, iterate choices, fuse
      (os.path.join(soda, 'jacobi2d.soda'), 2, (1, 2, 5, 13, 26, 40), (13, 12, 8, 4)),
      (os.path.join(soda, 'blur.soda'), 2, (1,), ()),
      (os.path.join(soda, 'denoise2d.soda'), 2, (1,), ()),
      (os.path.join(soda, 'sobel2d.soda'), 2, (1,), ()),
      (os.path.join(golden, 'coupled2d.soda'), 2, (2, 4, 6), (2,)),
      (os.path.join(soda, 'heat3d.soda'), 3, (1, 2, 4, 7), (2,)),
  ]
  tdt = {'float32': torch.float32, 'uint16': torch.int16, 'int16': torch.int16,
         'int32': torch.int32, 'uint8': torch.uint8}
  bad = 0
  t0 = time.time()
  # 2-D: <= 4800 x 3008 x 4 B x 4 tensors; room for padding
  pinned = None if args.dry else runtime.PinnedBuffer(4 * (60 << 20))
  for trial in range(args.trials):
    path, dim, its, fuse = menu[int(rng.integers(len(menu)))]
    iterate = int(rng.choice(its))
    if dim == 2:
      extent = (int(rng.integers(16, 600)) * 8, int(rng.integers(100, 3000)))
    else:
      extent = (int(rng.integers(4, 20)) * 8, int(rng.integers(20, 70)),
                int(rng.integers(40, 300)))
    st = core.from_file(path, iterate=iterate)
    lo_hi = {o: st.valid_box(extent, o) for o in st.output_names}
    if not all(h > l for lo, hi in lo_hi.values() for l, h in zip(lo, hi)):
      continue
    os.environ['SODA_HIP_HOST_CHUNK_KB'] = str(int(rng.choice(
        [16, 64, 256, 1024, 4096, 16384])))
    # never / by the rule / by the library's own estimate
    bands = ['0', '1', '1', ''][int(rng.integers(0, 4))]
    os.environ.pop('SODA_HIP_HOST_BANDS', None)
    if bands:
      os.environ['SODA_HIP_HOST_BANDS'] = bands
    shape = extent[::-1]
    pad_in, pad_out = int(rng.integers(0, 3)) * 8, int(rng.integers(0, 3)) * 4
    ins, big = {}, {}
    for n, t in zip(st.input_names, st.input_types):
      dt = np.dtype(t.np_name)
      full = (rng.random(shape[:-1] + (shape[-1] + pad_in,)).astype(dt)
              if dt.kind == 'f' else
              rng.integers(0, 3000, shape[:-1] + (shape[-1] + pad_in,)).astype(dt))
      big[n] = full
      ins[n] = full[..., pad_in // 2:pad_in // 2 + shape[-1]]
    outs_big = {n: np.full(shape[:-1] + (shape[-1] + pad_out,), 7,
                           np.dtype(t.np_name))
                for n, t in zip(st.output_names, st.output_types)}
    outs = {n: a[..., pad_out // 2:pad_out // 2 + shape[-1]]
            for n, a in outs_big.items()}
    prog = None if args.dry else runtime.Program(
        st, lower.LowerOptions(fuse=fuse), extent=extent)
    # the caller's arrays registered with the GPU or not: dense ones (no
    # padding) then go by DMA where they are, padded ones through the slots
    # tensors live in ONE page-aligned buffer registered for the whole soak
    # (runtime.PinnedBuffer), or in numpy's own memory
    pin = ['none', 'all', 'inputs', 'outputs'][int(rng.integers(0, 4))]
    if pinned is not None:
      at = 0
      for group, names, which in ((big, st.input_names, ('all', 'inputs')),
                                  (outs_big, st.output_names, ('all', 'outputs'))):
        for n in names:
          if pin in which:
            home = pinned.array(group[n].shape, group[n].dtype, at)
            home[...] = group[n]
            group[n] = home
            at += -(-home.nbytes // 4096) * 4096
      ins = {n: big[n][..., pad_in // 2:pad_in // 2 + shape[-1]] for n in ins}
      outs = {n: a[..., pad_out // 2:pad_out // 2 + shape[-1]]
              for n, a in outs_big.items()}
    if args.dry or args.verbose:
      print(json.dumps({'trial': trial, 'program': os.path.basename(path),
                        'extent': extent, 'iterate': iterate,
                        'chunk_kb': os.environ['SODA_HIP_HOST_CHUNK_KB'],
                        'bands': bands, 'pinned': pin,
                        'pads': [pad_in, pad_out],
                        'out_at': [hex(outs_big[n].ctypes.data)
                                   for n in st.output_names]}), flush=True)
    if args.dry:
      continue
    try:
      prog.run(ins, outputs=outs)
      dev_in = [torch.from_numpy(np.ascontiguousarray(ins[n]).view(
          np.int16 if ins[n].dtype == np.uint16 else ins[n].dtype)).cuda()
                for n in st.input_names]
      dev_out = [torch.zeros(shape, device='cuda', dtype=tdt[t.np_name])
                 for t in st.output_types]
      prog.run_device([t.data_ptr() for t in dev_out],
                      [t.data_ptr() for t in dev_in], extent)
      torch.cuda.synchronize()
      for n, t in zip(st.output_names, dev_out):
        lo, hi = lo_hi[n]
        idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
        want = t.cpu().numpy().view(outs[n].dtype)
        wrong = int((outs[n][idx].view(np.uint8) !=
                     want[idx].view(np.uint8)).sum())
        mask = np.ones(outs_big[n].shape, bool)
        mask[..., pad_out // 2:pad_out // 2 + shape[-1]][idx] = False
        touched = int((outs_big[n][mask] != 7).sum())
        if wrong or touched:
          bad += 1
          print(json.dumps({'trial': trial, 'program': os.path.basename(path),
                            'extent': extent, 'iterate': iterate,
                            'chunk_kb': os.environ['SODA_HIP_HOST_CHUNK_KB'],
                            'bands': bands, 'pinned': pin,
                            'pads': [pad_in, pad_out], 'wrong_bytes': wrong,
                            'touched_outside': touched}), flush=True)
    finally:
      if prog is not None:
        prog.close()
    if (trial + 1) % 25 == 0:
      print('... %d trials, %d bad, %.0f s' % (trial + 1, bad, time.time() - t0),
            flush=True)
  print('host soak: %d trials, %d bad' % (args.trials, bad))
  return 1 if bad else 0


if __name__ == '__main__':
  sys.exit(main())
