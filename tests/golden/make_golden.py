#!/usr/bin/env python3
"""Writes tests/golden/*.npz: small seeded inputs and the expected outputs.

The reference cannot be run here, so these vectors come from
this repo's CPU oracle AFTER it passed the closed-form KATs and the
three-restatement agreement tests in tests/test_oracle.py; they freeze that
behaviour so later rounds cannot drift silently.  For jacobi2d, blur, heat3d,
skew2d, jacobi3d and denoise3d the expected outputs are produced by the
HAND-WRITTEN C kernels in oracle/kat_kernels.c (independent of this repo's
parser and generators)."""
import ctypes
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from soda_amd import core  # noqa: E402
from oracle import numpy_oracle  # noqa: E402


def ptr(a):
  return a.ctypes.data_as(ctypes.c_void_p)


def main():
  subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle')], check=True)
  kat = ctypes.CDLL(os.path.join(ROOT, 'oracle', '_build', 'libkat_kernels.so'))
  init = ctypes.CDLL(os.path.join(ROOT, 'oracle', '_build', 'libref_init.so'))
  rng = np.random.default_rng(20260101)

  a = np.empty((36, 44), np.float32)     # the reference harness's own inputs
  init.ref_init_float(ptr(a), ctypes.c_int64(a.size))
  out = np.empty_like(a)
  kat.kat_jacobi2d(ptr(a), ptr(out), 44, 36, 3)
  np.savez_compressed(os.path.join(HERE, 'jacobi2d.npz'), iterate=3, in_t1=a,
                      out_t0=out)

  a = rng.integers(0, 65536, (24, 40)).astype(np.uint16)
  out = np.empty_like(a)
  kat.kat_blur(ptr(a), ptr(out), 40, 24)
  np.savez_compressed(os.path.join(HERE, 'blur.npz'), iterate=1, in_input=a,
                      out_blur_y=out)

  a = rng.random((12, 14, 16), dtype=np.float32)
  out = np.empty_like(a)
  kat.kat_heat3d(ptr(a), ptr(out), 16, 14, 12, 2)
  np.savez_compressed(os.path.join(HERE, 'heat3d.npz'), iterate=2, in_in=a,
                      out_out=out)

  a = rng.random((20, 24), dtype=np.float32)
  out = np.empty_like(a)
  kat.kat_skew2d(ptr(a), ptr(out), 24, 20)
  np.savez_compressed(os.path.join(HERE, 'skew2d.npz'), iterate=1, in_a=a,
                      out_c=out)

  for name in ('sobel2d', 'denoise2d'):
    st = core.from_file(os.path.join(HERE, 'soda', name + '.soda'))
    ins = {}
    for n, t in zip(st.input_names, st.input_types):
      if t.is_float:
        ins[n] = rng.random((20, 36), dtype=np.float32)
      else:
        ins[n] = rng.integers(-3000, 3000, (20, 36)).astype(t.np_name)
    outs = numpy_oracle.run(st, ins)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), iterate=1,
                        **{'in_' + k: v for k, v in ins.items()},
                        **{'out_' + k: v for k, v in outs.items()})

  # features the reference declares but never delivered, as defined by this
  # build: `border: preserve` and `param` arrays.  Own
  # seed, so the vectors above never move.
  rng2 = np.random.default_rng(20260102)
  for tag, name, kw, shape in (
      ('jacobi2d_preserve', 'jacobi2d', dict(iterate=3, border='preserve'),
       (20, 36)),
      ('heat3d_preserve', 'heat3d', dict(iterate=2, border='preserve'),
       (10, 12, 16)),
      ('conv2d', '../conv2d', dict(), (20, 36)),
      ('conv2d_preserve', '../conv2d', dict(border='preserve'), (20, 36))):
    st = core.from_file(os.path.join(HERE, 'soda', name + '.soda'), **kw)
    ins = {n: rng2.random(shape, dtype=np.float32) for n in st.input_names}
    for p in st.param_stmts:
      ins[p.name] = rng2.random(p.size or (1,)).astype(p.haoda_type.np_name)
    outs = numpy_oracle.run(st, ins)
    np.savez_compressed(os.path.join(HERE, tag + '.npz'), iterate=st.iterate,
                        **{'in_' + k: v for k, v in ins.items()},
                        **{'out_' + k: v for k, v in outs.items()})


  # the reference's two 3-D programs left (round 3): expected outputs from the
  # HAND-WRITTEN nests in oracle/kat_kernels.c.  Own seed again.
  rng3 = np.random.default_rng(20260103)
  a = rng3.random((12, 14, 20), dtype=np.float32)
  out = np.empty_like(a)
  kat.kat_jacobi3d(ptr(a), ptr(out), 20, 14, 12, 2)
  np.savez_compressed(os.path.join(HERE, 'jacobi3d.npz'), iterate=2, in_t1=a,
                      out_t0=out)
  f = rng3.random((12, 14, 20), dtype=np.float32)
  u = rng3.random((12, 14, 20), dtype=np.float32)
  out = np.empty_like(f)
  kat.kat_denoise3d(ptr(f), ptr(u), ptr(out), 20, 14, 12)
  np.savez_compressed(os.path.join(HERE, 'denoise3d.npz'), iterate=1, in_f=f,
                      in_u=u, out_output=out)


if __name__ == '__main__':
  main()
