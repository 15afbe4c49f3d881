#!/usr/bin/env python3
"""SURVEY.md 8(f3), the measurement behind "not built": fp32 matrix
instructions against the vector ALU on gfx950, FLOP per clock per SIMD.

  v_mfma_f32_32x32x2_f32   2 * 32*32*2 = 4096 FLOP per instruction
  v_mfma_f32_16x16x4_f32   2 * 16*16*4 = 2048 FLOP per instruction
  v_fma_f32                2 * 64      =  128 FLOP per instruction (wave64)
  v_mul_f32 + v_add_f32    the exact (unfused) evaluation of one multiply-add:
                           2 instructions, 128 FLOP

A row-convolution on the matrix pipe evaluates every tap as a fused
multiply-add inside the instruction; the reference's result needs the product
rounded BEFORE the add, so the path is only interesting if the
matrix pipe is much faster than the vector pair -- this prints by how much.
Independent accumulators (no dependent chain), 1-8 waves per SIMD, cycles from
s_memtime inside the kernel (no clock assumption), wall time beside it.
Writes profiles-style JSON to stdout / --out."""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SRC_TAIL = r'''
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define TIMED_BEGIN \
  unsigned long long* stamp = (unsigned long long*)a.buf[15];   /* debug slot */ \
  float* out = (float*)a.buf[1]; \
  const float c = out[threadIdx.x & 63], d = out[64 + (threadIdx.x & 63)]; \
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define TIMED_END(value) \
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
  out[128 + blockIdx.x * 64 + threadIdx.x] = (value); \
  if (threadIdx.x == 0) stamp[blockIdx.x] = t1 - t0;

extern "C" __global__ void __launch_bounds__(64) k_mfma32(soda_hip_kargs_t a) {
  TIMED_BEGIN
  v16f x0 = {0}, x1 = {0}, x2 = {0}, x3 = {0};
  for (int i = 0; i < a.extent[1]; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      x0 = __builtin_amdgcn_mfma_f32_32x32x2f32(c, d, x0, 0, 0, 0);
      x1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c, d, x1, 0, 0, 0);
      x2 = __builtin_amdgcn_mfma_f32_32x32x2f32(c, d, x2, 0, 0, 0);
      x3 = __builtin_amdgcn_mfma_f32_32x32x2f32(c, d, x3, 0, 0, 0);
    }
  }
  TIMED_END(x0[0] + x1[1] + x2[2] + x3[3])
}
extern "C" __global__ void __launch_bounds__(64) k_mfma16(soda_hip_kargs_t a) {
  TIMED_BEGIN
  v4f x0 = {0}, x1 = {0}, x2 = {0}, x3 = {0};
  for (int i = 0; i < a.extent[1]; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      x0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c, d, x0, 0, 0, 0);
      x1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c, d, x1, 0, 0, 0);
      x2 = __builtin_amdgcn_mfma_f32_16x16x4f32(c, d, x2, 0, 0, 0);
      x3 = __builtin_amdgcn_mfma_f32_16x16x4f32(c, d, x3, 0, 0, 0);
    }
  }
  TIMED_END(x0[0] + x1[1] + x2[2] + x3[3])
}
extern "C" __global__ void __launch_bounds__(64) k_fma(soda_hip_kargs_t a) {
  TIMED_BEGIN
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  float x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < a.extent[1]; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x0 = __builtin_fmaf(x0, c, d); x1 = __builtin_fmaf(x1, c, d);
      x2 = __builtin_fmaf(x2, c, d); x3 = __builtin_fmaf(x3, c, d);
      x4 = __builtin_fmaf(x4, c, d); x5 = __builtin_fmaf(x5, c, d);
      x6 = __builtin_fmaf(x6, c, d); x7 = __builtin_fmaf(x7, c, d);
    }
  }
  TIMED_END(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7)
}
// the exact form: product rounded, then added (compiled with -ffp-contract=off)
extern "C" __global__ void __launch_bounds__(64) k_mul_add(soda_hip_kargs_t a) {
  TIMED_BEGIN
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  float x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < a.extent[1]; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x0 = x0 * c + d; x1 = x1 * c + d; x2 = x2 * c + d; x3 = x3 * c + d;
      x4 = x4 * c + d; x5 = x5 * c + d; x6 = x6 * c + d; x7 = x7 * c + d;
    }
  }
  TIMED_END(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7)
}
'''

# kernel: (instructions per inner trip, FLOP per instruction, what, loop trips
# relative to --iters: the vector kernels run 16x as many so that every launch
# lasts milliseconds and the dispatch ramp of thousands of one-wave blocks does
# not show)
KERNELS = {
    'k_mfma32': (32, 4096, 'v_mfma_f32_32x32x2_f32', 1),
    'k_mfma16': (32, 2048, 'v_mfma_f32_16x16x4_f32', 2),
    'k_fma': (32, 128, 'v_fma_f32', 16),
    'k_mul_add': (64, 64, 'v_mul_f32 + v_add_f32 (one multiply-add = 2 instr.)',
                  16),
}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--out', default=None)
  ap.add_argument('--iters', type=int, default=2000)
  args = ap.parse_args()
  import torch
  from soda_amd import runtime
  from soda_amd.codegen.hip import lower
  code = runtime.compile_source(lower.runtime_text() + SRC_TAIL, 'mfmabench.hip')
  res = runtime.kernel_resources(code)
  lib = runtime.library()
  dev = torch.device('cuda', 0)
  stream = torch.cuda.current_stream().cuda_stream
  rows = []
  for kname, (per_trip, flop, what, scale) in KERNELS.items():
    iters = args.iters * scale
    for wps in (1, 2, 4, 8):
      nblocks = 1024 * wps
      buf = torch.zeros(128 + nblocks * 64, device=dev)
      stamps = torch.zeros(nblocks, device=dev, dtype=torch.int64)
      inp = torch.zeros(64, device=dev)
      plan = runtime.Plan()
      plan.abi_version = runtime.ABI_VERSION
      plan.dim = 2
      plan.num_inputs = plan.num_outputs = 1
      plan.elem_size[0] = plan.elem_size[1] = 4
      plan.num_kernels = 1
      plan.kernels[0].name = kname.encode()
      plan.kernels[0].block[0] = 64
      plan.kernels[0].block[1] = plan.kernels[0].block[2] = 1
      plan.kernels[0].tile[0] = 1
      plan.kernels[0].tile[1] = iters
      plan.kernels[0].tile[2] = plan.kernels[0].tile[3] = 1
      plan.kernels[0].march_dim = 0
      plan.num_passes = 1
      plan.passes[0].fused_iters = 1
      plan.passes[0].num_kernels = 1
      h = ctypes.c_void_p()
      runtime.check(lib.soda_hip_program_create(code, len(code), ctypes.byref(plan), 0,
                                                ctypes.byref(h)), 'create')
      # per-wave cycle counts go to the library's debug buffer slot (buf[15])
      runtime.check(lib.soda_hip_program_set_debug_buffer(h, ctypes.c_void_p(stamps.data_ptr())),
                    'debug buffer')
      outs = (ctypes.c_void_p * 1)(buf.data_ptr())
      ins = (ctypes.c_void_p * 1)(inp.data_ptr())
      ext = (ctypes.c_int32 * 2)(nblocks, iters)

      def go():
        runtime.check(lib.soda_hip_run_device(h, outs, ins, ext, 1, ctypes.c_void_p(stream)),
                      'run')

      go()
      a, b = runtime.Event(), runtime.Event()
      a.record(stream)
      for _ in range(3):
        go()
      b.record(stream)
      ms = a.elapsed_ms(b) / 3
      torch.cuda.synchronize()
      cycles = float(stamps.double().median().item())
      instr = iters * per_trip
      # Wall clock is the measure: instructions of all waves / time.  The
      # per-wave cycle count (s_memtime at entry and exit) is meaningful for a
      # LONE wave per SIMD only -- with more, the waves of a SIMD are not all
      # resident for each other's whole life (register limits, staggered
      # dispatch) -- and is reported as such.
      rows.append(dict(kernel=kname, instruction=what, waves_per_simd=wps,
                       vgprs=res.get(kname, {}).get('vgpr'),
                       agprs=res.get(kname, {}).get('agpr'),
                       ms=ms, median_wave_cycles=cycles,
                       lone_wave_cycles_per_instruction=(
                           cycles / instr if wps == 1 else None),
                       tflops_chip_wall=nblocks * instr * flop / (ms * 1e-3) / 1e12))
      print(json.dumps(rows[-1]), flush=True)
      lib.soda_hip_program_destroy(h)
  best = {}
  for r in rows:
    k = r['kernel']
    if k not in best or r['tflops_chip_wall'] > best[k]['tflops_chip_wall']:
      best[k] = r
  lone = {r['kernel']: r['lone_wave_cycles_per_instruction'] for r in rows
          if r['waves_per_simd'] == 1}
  summary = {
      'what': 'fp32 throughput on gfx950 (MI355X, 1024 SIMDs), wall clock, best '
              'over 1-8 waves per SIMD; one multiply-add counted as 2 FLOP',
      'best': {k: dict(instruction=v['instruction'],
                       tflops_chip_wall=round(v['tflops_chip_wall'], 1),
                       waves_per_simd=v['waves_per_simd'],
                       lone_wave_cycles_per_instruction=round(lone[k], 2))
               for k, v in best.items()},
      'rows': rows,
  }
  m = max(best['k_mfma32']['tflops_chip_wall'], best['k_mfma16']['tflops_chip_wall'])
  summary['mfma_over_fma'] = m / best['k_fma']['tflops_chip_wall']
  summary['mfma_over_exact_mul_add'] = m / best['k_mul_add']['tflops_chip_wall']
  print(json.dumps({k: v for k, v in summary.items() if k != 'rows'}, indent=1))
  if args.out:
    with open(args.out, 'w') as f:
      json.dump(summary, f, indent=1)


if __name__ == '__main__':
  main()
