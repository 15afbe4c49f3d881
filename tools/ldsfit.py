#!/usr/bin/env python3
"""How many single-wave blocks with L bytes of dynamic LDS are resident per CU?
A spin kernel of fixed per-wave duration is launched with 256*k blocks; the time
steps up when k exceeds the residency."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
from soda_amd import runtime
import placement

def main():
  lib = runtime.library()
  code = runtime.compile_source(placement.SRC, 'placement.hip')
  dev = torch.device('cuda', 0)
  stream = torch.cuda.current_stream().cuda_stream
  inp = torch.zeros(64, device=dev)
  buf = torch.zeros(2 * 8192, device=dev, dtype=torch.int32)
  for lds in (0, 8192, 10240, 13312, 16384, 18432, 19456, 20480, 24576, 32768, 65536):
    row = {}
    for k in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13, 16):
      plan = runtime.Plan()
      plan.abi_version = runtime.ABI_VERSION
      plan.dim = 2
      plan.num_inputs = plan.num_outputs = 1
      plan.elem_size[0] = plan.elem_size[1] = 4
      plan.num_kernels = 1
      plan.kernels[0].name = b'k_where'
      plan.kernels[0].block[0] = 64
      plan.kernels[0].block[1] = plan.kernels[0].block[2] = 1
      plan.kernels[0].tile[0] = 1
      plan.kernels[0].tile[1] = 10
      plan.kernels[0].tile[2] = plan.kernels[0].tile[3] = 1
      plan.kernels[0].lds_bytes = lds
      plan.num_passes = 1
      plan.passes[0].fused_iters = 1
      plan.passes[0].num_kernels = 1
      h = ctypes.c_void_p()
      runtime.check(lib.soda_hip_program_create(code, len(code), ctypes.byref(plan), 0, ctypes.byref(h)), 'create')
      outs = (ctypes.c_void_p * 1)(buf.data_ptr()); ins = (ctypes.c_void_p * 1)(inp.data_ptr())
      ext = (ctypes.c_int32 * 2)(256 * k, 10)
      def go():
        runtime.check(lib.soda_hip_run_device(h, outs, ins, ext, 1, ctypes.c_void_p(stream)), 'run')
      go()
      a, b = runtime.Event(), runtime.Event()
      a.record(stream); go(); b.record(stream); runtime.synchronize()
      row[k] = round(a.elapsed_ms(b) * 1e3)
      lib.soda_hip_program_destroy(h)
    print(json.dumps(dict(lds=lds, us_by_blocks_per_cu=row)), flush=True)

if __name__ == '__main__':
  main()
