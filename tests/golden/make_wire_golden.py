#!/usr/bin/env python3
"""Writes tests/golden/wire_*.npz: `<app>_kernel` at the level of the reference
host's WIRE streams -- the caller's arrays, the banks the
generated host lays them out in (tiles end to end, burst padding, bank
interleave, delayed inputs, the kStencilDistance tail;
reference src/soda/codegen/frt/host.py:124-249), the banks the kernel contract
leaves (every output cell late by its stencil offset, :401-408) and what the
host gathers from them (:340-427).

The reference cannot be run here: the vectors come from
oracle/frt_layout.py and the numpy oracle AFTER the checks of
tests/test_stream.py, and -- for the single-tile cases -- the gathered outputs
are asserted HERE against the hand-written C loop nests of oracle/kat_kernels.c,
which share nothing with this repo's parser, layout code or generators.  They
freeze the layout arithmetic so later rounds cannot drift silently."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from soda_amd import core, stream  # noqa: E402
from oracle import frt_layout  # noqa: E402

# tag, program, text substitutions, extent, iterate
CASES = [
    ('blur', 'blur.soda', (), (2000, 5), None),
    ('jacobi2d_4tiles', 'jacobi2d.soda', (), (100, 9), None),     # iterate 2
    ('heat3d_2x2tiles', 'heat3d.soda', (), (40, 40, 5), None),    # iterate 2
    ('denoise2d', 'denoise2d.soda', (), (32, 14), None),          # f delayed
    ('blur_4banks', 'blur.soda',
     ((r'input dram [^\n]*', 'input dram 0.1.2.3 uint16: input(2048, *)'),
      (r'output dram [\d.]+ \w+:', 'output dram 0.1.2.3 uint16:')),
     (2048, 5), None),
    ('jacobi2d_2banks', 'jacobi2d.soda',
     ((r'input dram [^\n]*', 'input dram 0.1 float: t1(32, *)'),
      (r'output dram [\d.]+ \w+:', 'output dram 2.3 float:')),
     (32, 13), None),
]


def program(case):
  tag, name, subs, extent, iterate = case
  text = open(os.path.join(HERE, 'soda', name)).read()
  for pat, rep in subs:
    text = re.sub(pat, rep, text)
  return core.from_text(text, iterate=iterate)


def vectors(case, rng):
  st = program(case)
  extent = case[3]
  ins = {}
  for n, t in zip(st.input_names, st.input_types):
    shape = tuple(extent[::-1])
    ins[n] = (rng.random(shape, dtype=np.float32) if t.is_float else
              rng.integers(0, 60000, shape).astype(t.np_name))
  lay = stream.WireLayout(st, extent)
  in_banks = frt_layout.scatter(lay, ins)
  out_banks = frt_layout.kernel_on_streams(lay, in_banks)
  got = {o: np.zeros(tuple(extent[::-1]), np.dtype(t.np_name))
         for o, t in zip(st.output_names, st.output_types)}
  frt_layout.gather(lay, out_banks, got)
  return st, lay, ins, in_banks, out_banks, got


def main():
  subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle')], check=True)
  kat = ctypes.CDLL(os.path.join(ROOT, 'oracle', '_build', 'libkat_kernels.so'))
  rng = np.random.default_rng(20260105)

  def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)

  for case in CASES:
    tag, _, _, extent, _ = case
    st, lay, ins, in_banks, out_banks, got = vectors(case, rng)
    # single tile: what the host hands back is the n-D stencil -- the
    # hand-written C nests say so too
    if lay.tiles == 1 and tag.startswith(('blur', 'jacobi2d')):
      a = ins[st.input_names[0]]
      want = np.empty_like(a)
      if tag.startswith('blur'):
        kat.kat_blur(ptr(a), ptr(want), extent[0], extent[1])
      else:
        kat.kat_jacobi2d(ptr(a), ptr(want), extent[0], extent[1], st.iterate)
      o = st.output_names[0]
      lo, hi = st.valid_box(extent, o)
      idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
      assert np.array_equal(got[o][idx], want[idx]), tag
    arrays = {'extent': np.array(extent), 'cycle_count': lay.cycle_count}
    for n, a in ins.items():
      arrays['in_' + n] = a
    for n, bs in in_banks.items():
      for b, a in enumerate(bs):
        arrays['inbank%d_%s' % (b, n)] = a
    for n, bs in out_banks.items():
      for b, a in enumerate(bs):
        arrays['outbank%d_%s' % (b, n)] = a
    for n, a in got.items():
      arrays['out_' + n] = a
    np.savez_compressed(os.path.join(HERE, 'wire_%s.npz' % tag), **arrays)
    print(tag, 'tiles', lay.tiles, 'banks', max(lay.bank_count.values()),
          'stream elements', lay.cycle_count * lay.epc[st.input_names[0]])


if __name__ == '__main__':
  main()
