#!/usr/bin/env python3
"""Replays the VALU instruction sequence of one row step of the fused jacobi2d
kernel (tools/experiments/tick_variants.json, cut from the compiler's ISA) in a
register-only loop: what does the SIMD sustain on exactly this mix, without any
memory, LDS or barrier?  Variants: as compiled; DPP modifiers dropped; literal
multiplier replaced by a register."""
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
    %(init)s
    "s_mov_b32 s20, %%2\\n\\t"
    "1:\\n\\t"
    %(body)s
    "s_sub_u32 s20, s20, 1\\n\\t"
    "s_cmp_lg_u32 s20, 0\\n\\t"
    "s_cbranch_scc1 1b\\n\\t"
    "v_add_f32 %%0, v0, v1\\n\\t"
    : "=v"(r) : "v"(c), "s"(a.extent[1])
    : %(clob)s, "s20", "scc");
  out[blockIdx.x * 64 + threadIdx.x] = r;
}
'''

def main():
  variants = json.load(open(os.path.join(ROOT, 'tools/experiments/tick_variants.json')))
  lib = runtime.library()
  dev = torch.device('cuda', 0)
  buf = torch.zeros(1 << 22, device=dev)
  inp = torch.zeros(64, device=dev)
  stream = torch.cuda.current_stream().cuda_stream
  iters = 100
  regs = ['v%d' % i for i in range(48)]
  init = '\n    '.join('"v_mov_b32 %s, %%1\\n\\t"' % r for r in regs)
  clob = ', '.join('"%s"' % r for r in regs)
  for name, seq in variants.items():
    reps = 4
    body = '\n    '.join('"%s\\n\\t"' % l for l in seq * reps)
    src = lower.runtime_text() + HEAD % dict(name='k_' + name, init=init, body=body, clob=clob)
    code = runtime.compile_source(src, 'tickbench_%s.hip' % name)
    n = sum(1 for l in seq if l.startswith('v_')) * reps   # VALU instructions only
    for waves_per_simd in (1, 2, 3, 4, 6, 8):
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
