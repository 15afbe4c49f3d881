#!/usr/bin/env python3
"""Where does the dispatcher put the waves of a grid?  Every single-wave block
spins for a while (so the whole grid is resident at once) and records HW_ID /
XCC_ID; prints the histogram of waves per SIMD and per CU, with and without an
LDS allocation that caps residency."""
import collections, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from soda_amd import runtime
from soda_amd.codegen.hip import lower

SRC = lower.runtime_text() + '''
extern "C" __global__ void __launch_bounds__(256) k_where(soda_hip_kargs_t a) {
  unsigned* out = (unsigned*)a.buf[1];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float x = threadIdx.x;
  for (int i = 0; i < a.extent[1] * 2000; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x));
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
  if ((threadIdx.x & 63) == 0) { out[2 * wave] = hw; out[2 * wave + 1] = xcc; }
  if (x == 12345.f) out[0] = 0;
}
'''

def main():
  lib = runtime.library()
  code = runtime.compile_source(SRC, 'placement.hip')
  dev = torch.device('cuda', 0)
  stream = torch.cuda.current_stream().cuda_stream
  inp = torch.zeros(64, device=dev)
  for waves_per_block in (1, 4):
    for nwaves in (1024, 2048, 2560, 3072):
      for lds in (0, 19 * 1024 * waves_per_block if waves_per_block == 1 else 65536):
        buf = torch.zeros(2 * nwaves, device=dev, dtype=torch.int32)
        plan = runtime.Plan()
        plan.abi_version = runtime.ABI_VERSION
        plan.dim = 2
        plan.num_inputs = plan.num_outputs = 1
        plan.elem_size[0] = plan.elem_size[1] = 4
        plan.num_kernels = 1
        plan.kernels[0].name = b'k_where'
        plan.kernels[0].block[0] = 64 * waves_per_block
        plan.kernels[0].block[1] = plan.kernels[0].block[2] = 1
        plan.kernels[0].tile[0] = 1
        plan.kernels[0].tile[1] = 50
        plan.kernels[0].tile[2] = plan.kernels[0].tile[3] = 1
        plan.kernels[0].lds_bytes = lds
        plan.num_passes = 1
        plan.passes[0].fused_iters = 1
        plan.passes[0].num_kernels = 1
        h = ctypes.c_void_p()
        runtime.check(lib.soda_hip_program_create(code, len(code), ctypes.byref(plan), 0, ctypes.byref(h)), 'create')
        outs = (ctypes.c_void_p * 1)(buf.data_ptr()); ins = (ctypes.c_void_p * 1)(inp.data_ptr())
        ext = (ctypes.c_int32 * 2)(nwaves // waves_per_block, 50)
        runtime.check(lib.soda_hip_run_device(h, outs, ins, ext, 1, ctypes.c_void_p(stream)), 'run')
        torch.cuda.synchronize()
        v = buf.cpu().numpy().astype('uint32').reshape(-1, 2)
        per_simd = collections.Counter()
        per_cu = collections.Counter()
        for hw, xcc in v:
          hw = int(hw); xcc = int(xcc) & 0xf
          simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
          per_simd[(xcc, se, sh, cu, simd)] += 1
          per_cu[(xcc, se, sh, cu)] += 1
        hs = collections.Counter(per_simd.values()); hc = collections.Counter(per_cu.values())
        print(json.dumps(dict(waves_per_block=waves_per_block, waves=nwaves, lds=lds,
                              cus_used=len(per_cu), simds_used=len(per_simd),
                              waves_per_simd_hist=dict(sorted(hs.items())),
                              waves_per_cu_hist=dict(sorted(hc.items())))), flush=True)
        lib.soda_hip_program_destroy(h)

if __name__ == '__main__':
  main()
