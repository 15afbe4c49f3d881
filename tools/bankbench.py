#!/usr/bin/env python3
"""VGPR bank-conflict microbenchmark on gfx950: v_add_f32 with both VGPR
sources in the same bank (index mod 4) against sources in different banks;
explicit registers through one asm block.  Cycles per wave64 instruction per
SIMD at an assumed 2.2 GHz."""
import ctypes, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from soda_amd import runtime
from soda_amd.codegen.hip import lower

HEAD = '''
extern "C" __global__ void __launch_bounds__(64) %(name)s(soda_hip_kargs_t a) {
  float* out = (float*)a.buf[1];
  float c = out[threadIdx.x], r;
  asm volatile(
    "v_mov_b32 v40, %%1\\n\\tv_mov_b32 v41, %%1\\n\\tv_mov_b32 v42, %%1\\n\\tv_mov_b32 v43, %%1\\n\\t"
    "v_mov_b32 v44, %%1\\n\\tv_mov_b32 v45, %%1\\n\\tv_mov_b32 v46, %%1\\n\\tv_mov_b32 v47, %%1\\n\\t"
    "v_mov_b32 v48, %%1\\n\\tv_mov_b32 v49, %%1\\n\\tv_mov_b32 v50, %%1\\n\\tv_mov_b32 v51, %%1\\n\\t"
    "s_mov_b32 s20, %%2\\n\\t"
    "1:\\n\\t"
    %(body)s
    "s_sub_u32 s20, s20, 1\\n\\t"
    "s_cmp_lg_u32 s20, 0\\n\\t"
    "s_cbranch_scc1 1b\\n\\t"
    "v_add_f32 %%0, v40, v41\\n\\tv_add_f32 %%0, %%0, v42\\n\\tv_add_f32 %%0, %%0, v43\\n\\t"
    : "=v"(r) : "v"(c), "s"(a.extent[1])
    : "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","s20","scc");
  out[blockIdx.x * 64 + threadIdx.x] = r;
}
'''

def body(pattern, n=256):
  lines = []
  for i in range(n):
    lines.append('"%s\\n\\t"' % pattern[i % len(pattern)])
  return '\n    '.join(lines)

KERNELS = {
  # dst, src0, src1: accumulators v40..v43, constants v44..v51
  'diff_bank': ['v_add_f32 v40, v45, v40', 'v_add_f32 v41, v46, v41', 'v_add_f32 v42, v47, v42', 'v_add_f32 v43, v44, v43'],
  'same_bank': ['v_add_f32 v40, v44, v40', 'v_add_f32 v41, v45, v41', 'v_add_f32 v42, v46, v42', 'v_add_f32 v43, v47, v43'],
  'same_reg':  ['v_add_f32 v40, v40, v40', 'v_add_f32 v41, v41, v41', 'v_add_f32 v42, v42, v42', 'v_add_f32 v43, v43, v43'],
  'dst_other_bank_srcs_same': ['v_add_f32 v41, v44, v48', 'v_add_f32 v42, v45, v49', 'v_add_f32 v43, v46, v50', 'v_add_f32 v40, v47, v51'],
  'dst_other_bank_srcs_diff': ['v_add_f32 v40, v45, v50', 'v_add_f32 v41, v46, v51', 'v_add_f32 v42, v47, v48', 'v_add_f32 v43, v44, v49'],
  'literal_mul': ['v_mul_f32 v40, 0x3e4ccccd, v40', 'v_mul_f32 v41, 0x3e4ccccd, v41', 'v_mul_f32 v42, 0x3e4ccccd, v42', 'v_mul_f32 v43, 0x3e4ccccd, v43'],
  'snop_mix': ['v_add_f32 v40, v45, v40', 'v_add_f32 v41, v46, v41', 'v_add_f32 v42, v47, v42', 's_nop 1'],
  # serial dependent chain through one register, like one cell's 4 adds + mul
  'chain5': ['v_add_f32 v40, v45, v46', 'v_add_f32 v40, v47, v40', 'v_add_f32 v40, v49, v40', 'v_add_f32 v40, v50, v40', 'v_mul_f32 v41, 0x3e4ccccd, v40'],
}


def main():
  only = sys.argv[1:]
  lib = runtime.library()
  dev = torch.device('cuda', 0)
  buf = torch.zeros(1 << 22, device=dev)
  inp = torch.zeros(64, device=dev)
  stream = torch.cuda.current_stream().cuda_stream
  iters = 200
  for name, pattern in KERNELS.items():
    if only and name not in only:
      continue
    n = 256 - 256 % len(pattern)
    src = lower.runtime_text() + HEAD % dict(name='k_' + name, body=body(pattern, n))
    code = runtime.compile_source(src, 'bankbench_%s.hip' % name)
    for waves_per_simd in (1, 2, 3, 4, 8):
      plan = runtime.Plan()
      plan.abi_version = runtime.ABI_VERSION
      plan.dim = 2
      plan.num_inputs = plan.num_outputs = 1
      plan.elem_size[0] = plan.elem_size[1] = 4
      plan.num_kernels = 1
      plan.kernels[0].name = ('k_' + name).encode()
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
      per_simd = iters * n * waves_per_simd
      cyc = ms * 1e-3 * 2.2e9 / per_simd
      print(json.dumps(dict(kernel=name, waves_per_simd=waves_per_simd, ms=round(ms, 4), cycles_per_instr_per_simd=round(cyc, 2))), flush=True)
      lib.soda_hip_program_destroy(h)


if __name__ == '__main__':
  main()
