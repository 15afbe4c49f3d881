#!/usr/bin/env python3
"""Times `<app>_kernel` on wire streams: unwire + program + wire,
device-resident banks, `dense` (marching kernels on the (tile..., rows) view)
against `linear` (causal 1-D form).  One JSON line per mode."""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--soda', default='tests/golden/soda/blur.soda')
  ap.add_argument('--extent', type=int, nargs='+', default=[2000, 16384])
  ap.add_argument('--steps', type=int, default=10)
  ap.add_argument('--tile', type=int, nargs='+', default=None,
                  help='tile size of dimensions 0..dim-2 instead of the file\'s')
  ap.add_argument('--iterate', type=int, default=None)
  ap.add_argument('--banks', type=int, default=None,
                  help='DRAM banks of every tensor instead of the file\'s')
  ap.add_argument('--host', action='store_true',
                  help='also time soda_hip_stream_run_host on pageable banks')
  args = ap.parse_args()
  from soda_amd import core, runtime, stream
  if args.banks:
    import re
    dram = '.'.join(map(str, range(args.banks)))
    text = re.sub(r'(input|output) dram [\d.]+', r'\1 dram ' + dram,
                  open(args.soda).read())
    st = core.from_text(text, iterate=args.iterate, tile_size=args.tile)
  else:
    st = core.from_file(args.soda, iterate=args.iterate, tile_size=args.tile)
  lay = stream.WireLayout(st, args.extent)
  lib = runtime.library()
  table = st.symbol_table

  def dev_banks(names):
    out = {}
    for n in names:
      nb = lay.bank_count[n]
      out[n] = []
      for _ in range(nb):
        p = ctypes.c_void_p()
        nbytes = lay.buf_elems[n] // nb * table[n].size_in_bytes
        runtime.check(lib.soda_hip_malloc(0, nbytes, ctypes.byref(p)), 'malloc')
        runtime.check(lib.soda_hip_memset(p, 1, nbytes, None), 'memset')
        out[n].append(p.value)
    return out

  ins, outs = dev_banks(st.input_names), dev_banks(st.output_names)
  cells = 1
  for e in args.extent:
    cells *= e
  for mode, direct in (('dense', True), ('dense', False), ('linear', True),
                       ('linear', False)):
    prog = stream.StreamProgram(st, dense=mode == 'dense', direct=direct)
    for _ in range(3):
      prog.run_banked_device(outs, ins, lay.cycle_count)
    runtime.synchronize()
    e0, e1 = runtime.Event(), runtime.Event()
    e0.record()
    for _ in range(args.steps):
      prog.run_banked_device(outs, ins, lay.cycle_count)
    e1.record()
    runtime.synchronize()
    ms = e0.elapsed_ms(e1) / args.steps
    print(json.dumps({'soda': os.path.basename(args.soda), 'extent': args.extent,
                      'mode': prog.last_mode,
                      'outputs': ('shift + copy pass' if not direct else
                                  'stored late, 16-byte interleave pass' if any(
                                      t.startswith('wire_') for t in prog.specs)
                                  else 'stored in place'),
                      'ms_per_call': round(ms, 4),
                      'cells_iters_per_s': cells * st.iterate / ms * 1e3,
                      'tiles': lay.tiles,
                      'banks': max(lay.bank_count.values())}))
    if args.host:
      # <app>_kernel on HOST banks, as the generated host calls it
      # (SODA_CPP_BINDING): pageable arrays of the reference's sizes
      import numpy as np
      import time
      hin = {n: [np.ones(lay.buf_elems[n] // lay.bank_count[n],
                         np.dtype(table[n].np_name))
                 for _ in range(lay.bank_count[n])] for n in st.input_names}
      hout = {n: [np.zeros(lay.buf_elems[n] // lay.bank_count[n],
                           np.dtype(table[n].np_name))
                  for _ in range(lay.bank_count[n])] for n in st.output_names}
      prog.run_banked_host(hout, hin, lay.cycle_count)
      ts = []
      for _ in range(3):
        t0 = time.perf_counter()
        prog.run_banked_host(hout, hin, lay.cycle_count)
        ts.append((time.perf_counter() - t0) * 1e3)
      nbytes = sum(a.nbytes for v in list(hin.values()) + list(hout.values())
                   for a in v)
      print(json.dumps({'soda': os.path.basename(args.soda), 'mode': prog.last_mode,
                        'banks': max(lay.bank_count.values()),
                        'host_banks_ms_per_call': round(min(ts), 3),
                        'bytes_moved': nbytes,
                        'GBs': nbytes / (min(ts) * 1e-3) / 1e9}))
    prog.close()


if __name__ == '__main__':
  main()
