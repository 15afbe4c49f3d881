#!/usr/bin/env python3
"""Cross-lane issue-rate microbenchmark on gfx950: what does one DPP-carrying
VALU instruction cost, by DPP control and by mix with plain adds?  Inline asm
so the compiler cannot combine or hoist.  Prints cycles per wave64 instruction
per SIMD at an assumed 2.2 GHz."""
import ctypes, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from soda_amd import runtime
from soda_amd.codegen.hip import lower

HEAD = '''
extern "C" __global__ void __launch_bounds__(64) %(name)s(soda_hip_kargs_t a) {
  float* out = (float*)a.buf[1];
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  float c = out[threadIdx.x], d = out[64 + threadIdx.x];
  int ci = threadIdx.x; (void)ci; (void)d;
  for (int i = 0; i < a.extent[1]; ++i) {
#pragma unroll
    for (int j = 0; j < %(reps)d; ++j) {
      asm volatile(%(body)s : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(d) : "v"(c), "v"(ci));
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + d;
}
'''
PLAIN = 'v_add_f32 %{r}, %5, %{r}'
def dpp(ctrl, src='%5'):
  return 'v_add_f32_dpp %{r}, ' + src + ', %{r} ' + ctrl + ' row_mask:0xf bank_mask:0xf bound_ctrl:1'

def body(instrs):
  """instrs: list of templates with {r} = accumulator index placeholder"""
  out = []
  for k, t in enumerate(instrs):
    out.append(t.replace('{r}', str(k % 4)))
  return '"' + '\\n\\t'.join(out) + '"'

KERNELS = {
    # name: (instruction list per asm block, reps)
    'plain': ([PLAIN] * 16, 16),
    'dpp_wave_shr': ([dpp('wave_shr:1')] * 16, 16),
    'dpp_row_shr': ([dpp('row_shr:1')] * 16, 16),
    'dpp_quad': ([dpp('quad_perm:[1,2,3,0]')] * 16, 16),
    'dpp_row_ror': ([dpp('row_ror:1')] * 16, 16),
    'dpp_row_mirror': ([dpp('row_mirror')] * 16, 16),
    'mov_dpp_wave_shr': (['v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1'] * 16, 16),
    'mix_1in8_wave_shr': (([PLAIN] * 7 + [dpp('wave_shr:1')]) * 2, 16),
    'mix_1in8_row_shr': (([PLAIN] * 7 + [dpp('row_shr:1')]) * 2, 16),
    'mix_1in16_wave_shr': ([PLAIN] * 15 + [dpp('wave_shr:1')], 16),
    # shifted operand = the accumulator another chain has just written
    'mix_1in8_hazard': (([PLAIN] * 7 + ['v_add_f32_dpp %3, %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1']) * 2, 16),
    'permlane32_swap': (['v_permlane32_swap_b32 %{r}, %4'] * 16, 16),
    'permlane16_swap': (['v_permlane16_swap_b32 %{r}, %4'] * 16, 16),
    'bpermute': (['ds_bpermute_b32 %{r}, %6, %{r}'] * 16 + ['s_waitcnt lgkmcnt(0)'], 16),
    'mix_1in8_bpermute': (([PLAIN] * 7 + ['ds_bpermute_b32 %4, %6, %4']) * 2 + ['s_waitcnt lgkmcnt(0)'], 16),
    'swizzle': (['ds_swizzle_b32 %{r}, %{r} offset:swizzle(BITMASK_PERM, "01pip")'] * 0 + ['ds_swizzle_b32 %{r}, %{r} offset:0x8000'] * 16 + ['s_waitcnt lgkmcnt(0)'], 16),
}


def main():
  only = sys.argv[1:]
  lib = runtime.library()
  dev = torch.device('cuda', 0)
  buf = torch.zeros(1 << 22, device=dev)
  inp = torch.zeros(64, device=dev)
  stream = torch.cuda.current_stream().cuda_stream
  iters = 200
  for name, (instrs, reps) in KERNELS.items():
    if only and name not in only:
      continue
    src = lower.runtime_text() + HEAD % dict(name='k_' + name, reps=reps, body=body(instrs))
    try:
      code = runtime.compile_source(src, 'dppbench_%s.hip' % name)
    except Exception as e:   # noqa
      print(json.dumps(dict(kernel=name, error=str(e)[-300:])))
      continue
    n_instr = sum(1 for t in instrs if not t.startswith('s_waitcnt')) * reps
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
      plan.kernels[0].lds_bytes = int(os.environ.get('DPPBENCH_LDS', '0'))
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
      per_simd = iters * n_instr * waves_per_simd
      cyc = ms * 1e-3 * 2.2e9 / per_simd
      print(json.dumps(dict(kernel=name, waves_per_simd=waves_per_simd, ms=round(ms, 4), cycles_per_instr_per_simd=round(cyc, 2))), flush=True)
      lib.soda_hip_program_destroy(h)


if __name__ == '__main__':
  main()
