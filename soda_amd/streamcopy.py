"""A plain float4 stream-copy kernel run through the same C ABI.

Measurement aid, not part of the stencil path: SURVEY.md section 8(d) defines
the roofline denominator the north star asks for ("measured HBM3E streaming
roofline") as a stream-copy HIP kernel timed on the same GPU.  bench.py and
tools/sweep.py time this next to the stencil kernels.
"""
import ctypes

from soda_amd import runtime, util
from soda_amd.codegen.hip import lower

_SOURCE = '''
// stream copy: every thread moves %(unroll)d float4, 16 bytes per lane per access
extern "C" __global__ void __launch_bounds__(256) soda_stream_copy(soda_hip_kargs_t a) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  const v4* __restrict__ src = (const v4*)a.buf[0];
  v4* __restrict__ dst = (v4*)a.buf[1];
  const int64_t n4 = (int64_t)a.extent[0] / 4;
  const int64_t base = (int64_t)blockIdx.x * (256 * %(unroll)d) + threadIdx.x;
  v4 r[%(unroll)d];
#pragma unroll
  for (int i = 0; i < %(unroll)d; ++i) {
    const int64_t j = base + (int64_t)i * 256;
    if (j < n4) r[i] = %(load)s;
  }
#pragma unroll
  for (int i = 0; i < %(unroll)d; ++i) {
    const int64_t j = base + (int64_t)i * 256;
    if (j < n4) %(store)s;
  }
}
'''


class StreamCopy:
  """dst[i] = src[i] over `floats` fp32 values (a multiple of 4)."""

  def __init__(self, device: int = 0, unroll: int = 4, nt_store: bool = False,
               nt_load: bool = False):
    self.unroll = unroll
    src = lower.runtime_text() + _SOURCE % dict(
        unroll=unroll,
        load='__builtin_nontemporal_load(src + j)' if nt_load else 'src[j]',
        store='__builtin_nontemporal_store(r[i], dst + j)' if nt_store
        else 'dst[j] = r[i]')
    self.code = runtime.compile_source(src, 'stream_copy.hip')
    plan = runtime.Plan()
    plan.abi_version = runtime.ABI_VERSION
    plan.dim = 1
    plan.num_inputs = plan.num_outputs = 1
    plan.elem_size[0] = plan.elem_size[1] = 4
    plan.num_kernels = 1
    plan.kernels[0].name = b'soda_stream_copy'
    plan.kernels[0].block[0] = 256
    plan.kernels[0].block[1] = plan.kernels[0].block[2] = 1
    plan.kernels[0].tile[0] = 256 * 4 * unroll
    for d in range(1, runtime.MAX_DIM):
      plan.kernels[0].tile[d] = 1
    plan.num_passes = 1
    plan.passes[0].fused_iters = 1
    plan.passes[0].num_kernels = 1
    self.plan = plan
    self._lib = runtime.library()
    self._h = ctypes.c_void_p()
    runtime.check(
        self._lib.soda_hip_program_create(self.code, len(self.code),
                                          ctypes.byref(plan), device,
                                          ctypes.byref(self._h)),
        'loading stream copy')

  def run(self, dst: int, src: int, floats: int, stream: int = 0) -> None:
    if floats % 4 or floats >= 2**31:
      raise util.InputError('floats must be a multiple of 4 below 2^31')
    outs = (ctypes.c_void_p * 1)(dst)
    ins = (ctypes.c_void_p * 1)(src)
    ext = (ctypes.c_int32 * 1)(floats)
    runtime.check(
        self._lib.soda_hip_run_device(self._h, outs, ins, ext, 1,
                                      ctypes.c_void_p(stream)), 'stream copy')

  def close(self) -> None:
    if self._h:
      self._lib.soda_hip_program_destroy(self._h)
      self._h = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass
