#!/usr/bin/env python3
"""VALU issue-rate microbenchmark on gfx950: scalar v_add_f32 vs packed
v_pk_add_f32, dependent chains, N waves per SIMD.  Prints cycles per wave-
instruction per SIMD (chip clock from the measured time is assumed 2.2 GHz)."""
import ctypes, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from soda_amd import runtime
from soda_amd.codegen.hip import lower

SRC = lower.runtime_text() + '''
typedef float v2 __attribute__((ext_vector_type(2)));
extern "C" __global__ void __launch_bounds__(64) k_scalar(soda_hip_kargs_t a) {
  float* out = (float*)a.buf[1];
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, c = out[0];
  for (int i = 0; i < a.extent[1]; ++i) {
#pragma unroll
    for (int j = 0; j < 64; ++j) { x0 += c; x1 += c; x2 += c; x3 += c; }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3;
}
extern "C" __global__ void __launch_bounds__(64) k_packed(soda_hip_kargs_t a) {
  float* out = (float*)a.buf[1];
  v2 x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, c = {out[0], out[1]};
  for (int i = 0; i < a.extent[1]; ++i) {
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      asm volatile("v_pk_add_f32 %0, %0, %2\\n\\tv_pk_add_f32 %1, %1, %2" : "+v"(x0), "+v"(x1) : "v"(c));
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0.x + x0.y + x1.x + x1.y;
}
extern "C" __global__ void __launch_bounds__(64) k_fma(soda_hip_kargs_t a) {
  float* out = (float*)a.buf[1];
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, c = out[0], one = out[1] + 1.0f;
  for (int i = 0; i < a.extent[1]; ++i) {
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      x0 = __builtin_fmaf(x0, one, c); x1 = __builtin_fmaf(x1, one, c);
      x2 = __builtin_fmaf(x2, one, c); x3 = __builtin_fmaf(x3, one, c);
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3;
}
extern "C" __global__ void __launch_bounds__(64) k_mul(soda_hip_kargs_t a) {
  float* out = (float*)a.buf[1];
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, c = out[0] + 1.0f;
  for (int i = 0; i < a.extent[1]; ++i) {
#pragma unroll
    for (int j = 0; j < 64; ++j) { x0 *= c; x1 *= c; x2 *= c; x3 *= c; }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3;
}
extern "C" __global__ void __launch_bounds__(64) k_pkfma(soda_hip_kargs_t a) {
  float* out = (float*)a.buf[1];
  v2 x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, c = {out[0], out[1]}, one = {out[2] + 1.f, out[3] + 1.f};
  for (int i = 0; i < a.extent[1]; ++i) {
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      asm volatile("v_pk_fma_f32 %0, %0, %3, %2\\n\\tv_pk_fma_f32 %1, %1, %3, %2" : "+v"(x0), "+v"(x1) : "v"(c), "v"(one));
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0.x + x0.y + x1.x + x1.y;
}
extern "C" __global__ void __launch_bounds__(64) k_addv(soda_hip_kargs_t a) {
  float* out = (float*)a.buf[1];
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, c = out[threadIdx.x];
  for (int i = 0; i < a.extent[1]; ++i) {
#pragma unroll
    for (int j = 0; j < 64; ++j) { x0 += c; x1 += c; x2 += c; x3 += c; }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3;
}
extern "C" __global__ void __launch_bounds__(64) k_mullit(soda_hip_kargs_t a) {
  float* out = (float*)a.buf[1];
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  for (int i = 0; i < a.extent[1]; ++i) {
#pragma unroll
    for (int j = 0; j < 64; ++j) { x0 *= 0.2f; x1 *= 0.2f; x2 *= 0.2f; x3 *= 0.2f; }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3;
}
extern "C" __global__ void __launch_bounds__(64) k_chain1(soda_hip_kargs_t a) {
  float* out = (float*)a.buf[1];
  float x0 = threadIdx.x, c = out[threadIdx.x];
  for (int i = 0; i < a.extent[1]; ++i) {
#pragma unroll
    for (int j = 0; j < 256; ++j) { x0 += c; }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0;
}
extern "C" __global__ void __launch_bounds__(64) k_dppv(soda_hip_kargs_t a) {
  float* out = (float*)a.buf[1];
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, c = out[threadIdx.x];
  for (int i = 0; i < a.extent[1]; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      x0 += c; x1 += c; x2 += c; x3 += soda_lane_up(x0);
      x0 += c; x1 += c; x2 += c; x3 += c;
      x0 += c; x1 += c; x2 += c; x3 += c;
      x0 += soda_lane_dn(x3); x1 += c; x2 += c; x3 += c;
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3;
}
#define MIX(SH_UP, SH_DN) \
  float* out = (float*)a.buf[1]; \
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, c = out[threadIdx.x]; \
  const int up_addr = (((int)threadIdx.x + 1) & 63) * 4, dn_addr = (((int)threadIdx.x + 63) & 63) * 4; \
  (void)up_addr; (void)dn_addr; \
  for (int i = 0; i < a.extent[1]; ++i) { \
    _Pragma("unroll") for (int j = 0; j < 16; ++j) { \
      x0 += c; x1 += c; x2 += c; x3 += SH_UP(x0); \
      x0 += c; x1 += c; x2 += c; x3 += c; \
      x0 += c; x1 += c; x2 += c; x3 += c; \
      x0 += SH_DN(x3); x1 += c; x2 += c; x3 += c; \
    } \
  } \
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3;
#define BPERM_UP(v) __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(up_addr, __builtin_bit_cast(int, v)))
#define BPERM_DN(v) __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(dn_addr, __builtin_bit_cast(int, v)))
#define ROWSHR(v) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true))
#define ROWSHL(v) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true))
#define BCAST15(v) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false))
#define WAVEROR(v) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x13c, 0xf, 0xf, false))
#define WAVEROL(v) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x134, 0xf, 0xf, false))
extern "C" __global__ void __launch_bounds__(64) k_bperm(soda_hip_kargs_t a) { MIX(BPERM_UP, BPERM_DN) }
extern "C" __global__ void __launch_bounds__(64) k_rowsh(soda_hip_kargs_t a) { MIX(ROWSHL, ROWSHR) }
extern "C" __global__ void __launch_bounds__(64) k_bcast(soda_hip_kargs_t a) { MIX(BCAST15, BCAST15) }
extern "C" __global__ void __launch_bounds__(64) k_rot(soda_hip_kargs_t a) { MIX(WAVEROL, WAVEROR) }
extern "C" __global__ void __launch_bounds__(64) k_dpp(soda_hip_kargs_t a) {
  float* out = (float*)a.buf[1];
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, c = out[0];
  for (int i = 0; i < a.extent[1]; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      x0 += c; x1 += c; x2 += c; x3 += soda_lane_up(x0);
      x0 += c; x1 += c; x2 += c; x3 += c;
      x0 += c; x1 += c; x2 += c; x3 += c;
      x0 += soda_lane_dn(x3); x1 += c; x2 += c; x3 += c;
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3;
}
'''

def main():
  code = runtime.compile_source(SRC, 'valubench.hip')
  lib = runtime.library()
  dev = torch.device('cuda', 0)
  buf = torch.zeros(1 << 22, device=dev)
  inp = torch.zeros(64, device=dev)
  stream = torch.cuda.current_stream().cuda_stream
  iters = 200
  for kname, per_iter in (('k_addv', 256), ('k_dppv', 256), ('k_bperm', 256), ('k_rowsh', 256), ('k_bcast', 256), ('k_rot', 256)):
    for waves_per_simd in (2, 3, 4, 8):
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
      plan.num_passes = 1
      plan.passes[0].fused_iters = 1
      plan.passes[0].num_kernels = 1
      h = ctypes.c_void_p()
      runtime.check(lib.soda_hip_program_create(code, len(code), ctypes.byref(plan), 0, ctypes.byref(h)), 'create')
      nblocks = 1024 * waves_per_simd
      outs = (ctypes.c_void_p * 1)(buf.data_ptr()); ins = (ctypes.c_void_p * 1)(inp.data_ptr())
      ext = (ctypes.c_int32 * 2)(nblocks, iters)
      def go():
        runtime.check(lib.soda_hip_run_device(h, outs, ins, ext, 1, ctypes.c_void_p(stream)), 'run')
      go(); a, b = runtime.Event(), runtime.Event()
      a.record(stream)
      for _ in range(5): go()
      b.record(stream)
      ms = a.elapsed_ms(b) / 5
      instr_per_wave = iters * per_iter
      per_simd = instr_per_wave * waves_per_simd
      cyc = ms * 1e-3 * 2.2e9 / per_simd
      print(json.dumps(dict(kernel=kname, waves_per_simd=waves_per_simd, ms=ms, cycles_per_instr_per_simd=cyc)))
      lib.soda_hip_program_destroy(h)

if __name__ == '__main__':
  main()
