"""ctypes binding of libsoda_hip.so and the JIT driver on top of it.

The Python side of the drop-in boundary (include/soda_hip.h).  The reference's
host reaches its kernel either linked directly or through the FRT runtime
(`fpga::Instance(bitstream)`, SetArg/WriteToDevice/Exec/ReadFromDevice/Finish;
reference src/soda/codegen/frt/host.py:97-113,288-322).  `Program` is that
object for a GPU: built from HIP source text instead of a bitstream, it owns
the loaded code object and runs it on host or device arrays.

There is no CPU fallback: a missing library, a failed JIT or a missing GPU
raises `util.BackendError`.
"""
import ctypes
import hashlib
import os
from typing import Dict, Optional, Sequence

from soda_amd import core, util
from soda_amd.codegen.hip import lower

_HERE = os.path.dirname(os.path.abspath(__file__))
# SODA_HIP_LIBRARY: another build of the same sources -- the sanitizer builds
# of `make -C soda_amd/csrc asan tsan` (tools/sanitize.sh), never a substitute
LIB_PATH = os.environ.get('SODA_HIP_LIBRARY') or \
    os.path.join(_HERE, 'libsoda_hip.so')
CACHE_DIR = os.environ.get('SODA_HIP_CACHE',
                           os.path.join(_HERE, '_jit_cache'))
ARCH = 'gfx950'
# -ffp-contract=off: bit-identical fp32 results to the CPU oracle (no FMA).
# -fno-slp-vectorize: hipcc would otherwise pair neighbouring cells into
#   v_pk_add_f32/v_pk_mul_f32; on gfx950 those issue at half the rate of the
#   scalar forms and need v_mov shuffles to line registers up (measured: same
#   instruction count, ~25 % more cycles, 19 more VGPRs on jacobi2d T=8).
# * `-fwrapv`: signed integer arithmetic wraps.  The reference's kernel is
#   hardware (ap_int arithmetic: two's complement, no undefined behaviour);
#   without the flag hipcc may and does compile an overflowing int32 product
#   to something else than its wrapped value (tools/fuzz_scan.py, programs
#   2184 / 3775: 10^4-10^5 cells away from gcc's result), and the two sides of
#   a parity test would be comparing undefined behaviours.  The oracle's C is
#   built with the same flag.  Cost: within run-to-run noise on every corpus
#   program (profiles/r03_wrapv.json).
COMPILE_OPTIONS = ('--offload-arch=%s' % ARCH, '-O3', '-ffp-contract=off',
                   '-fno-slp-vectorize', '-fwrapv', '-std=c++17') + tuple(
                       os.environ.get('SODA_HIP_EXTRA_FLAGS', '').split())

MAX_DIM = 4
MAX_TENSORS = 16
MAX_KERNELS = 32
MAX_PASSES = 8
MAX_PASS_KERNELS = 16
MAX_PARAMS = 8
MAX_SLABS = 64
NAME_LEN = 96
GROUP_NO_OVERLAP = 1
GROUP_CALIBRATE = 2
GROUP_THREADS = 4
ABI_VERSION = 7


class KernelDesc(ctypes.Structure):
  _fields_ = [('name', ctypes.c_char * NAME_LEN),
              ('block', ctypes.c_int32 * 3),
              ('tile', ctypes.c_int32 * MAX_DIM),
              ('lds_bytes', ctypes.c_int32),
              ('vec', ctypes.c_int32),
              ('march_dim', ctypes.c_int32),
              ('waves_along', ctypes.c_int32),
              ('warm', ctypes.c_int32),
              ('window_extra', ctypes.c_int32),
              ('max_elem', ctypes.c_int32),
              ('vgprs', ctypes.c_int32),
              ('pipe', ctypes.c_int32),
              ('chunk_fixed', ctypes.c_int32),
              ('step_ns', ctypes.c_float),
              ('warm_saved', ctypes.c_float),
              ('bytes_per_cell', ctypes.c_float),
              ('lane_redundancy', ctypes.c_float),
              ('max_extent0', ctypes.c_int32)]


class PassDesc(ctypes.Structure):
  _fields_ = [('fused_iters', ctypes.c_int32),
              ('num_kernels', ctypes.c_int32),
              ('kernel', ctypes.c_int32 * MAX_PASS_KERNELS),
              ('cost', ctypes.c_float)]


class Plan(ctypes.Structure):
  _fields_ = [('abi_version', ctypes.c_int32), ('dim', ctypes.c_int32),
              ('num_inputs', ctypes.c_int32), ('num_outputs', ctypes.c_int32),
              ('num_locals', ctypes.c_int32),
              ('num_params', ctypes.c_int32),
              ('param_elems', ctypes.c_int32 * MAX_PARAMS),
              ('elem_size', ctypes.c_int32 * MAX_TENSORS),
              ('num_kernels', ctypes.c_int32),
              ('kernels', KernelDesc * MAX_KERNELS),
              ('num_passes', ctypes.c_int32),
              ('passes', PassDesc * MAX_PASSES),
              ('has_reach', ctypes.c_int32), ('reach_lo', ctypes.c_int32),
              ('reach_hi', ctypes.c_int32)]


class StreamDesc(ctypes.Structure):
  """Mirror of soda_hip_stream_desc_t (the wire-format kernel, stream.py)."""
  _fields_ = [('dim', ctypes.c_int32), ('num_inputs', ctypes.c_int32),
              ('num_outputs', ctypes.c_int32), ('iterate', ctypes.c_int32),
              ('tile', ctypes.c_int32 * MAX_DIM),
              ('stencil_distance', ctypes.c_int32),
              ('banks', ctypes.c_int32 * MAX_TENSORS),
              ('elem_size', ctypes.c_int32 * MAX_TENSORS),
              ('elems_per_cycle', ctypes.c_int32 * MAX_TENSORS),
              ('shift', ctypes.c_int32 * MAX_TENSORS),
              ('num_linear', ctypes.c_int32),
              ('linear_vec', ctypes.c_int32 * 4)]


class SlabRun(ctypes.Structure):
  """Mirror of soda_hip_slab_run_t: a run between two halo exchanges."""
  _fields_ = [('keep_lo', ctypes.c_int32), ('keep_hi', ctypes.c_int32),
              ('reach_lo', ctypes.c_int32), ('reach_hi', ctypes.c_int32),
              ('ghost_lo', ctypes.c_int32), ('ghost_hi', ctypes.c_int32),
              ('send_lo', ctypes.c_int32), ('send_hi', ctypes.c_int32),
              ('ghosts_ready', ctypes.c_void_p),
              ('sendable', ctypes.c_void_p)]


class LaunchInfo(ctypes.Structure):
  """Mirror of soda_hip_launch_info_t."""
  _fields_ = [(n, ctypes.c_int32) for n in (
      'fused_iters', 'lo', 'hi', 'wait', 'record', 'split', 'chunk', 'chunks',
      'bnd_lo', 'bnd_hi')]


class GroupDesc(ctypes.Structure):
  """Mirror of soda_hip_group_desc_t."""
  _fields_ = [('num_slabs', ctypes.c_int32),
              ('device', ctypes.c_int32 * MAX_SLABS),
              ('extent', ctypes.c_int32 * MAX_DIM),
              ('reach_lo', ctypes.c_int32), ('reach_hi', ctypes.c_int32),
              ('iterate', ctypes.c_int32),
              ('exchange_every', ctypes.c_int32),
              ('flags', ctypes.c_int32)]


class SlabInfo(ctypes.Structure):
  """Mirror of soda_hip_slab_info_t."""
  _fields_ = [('device', ctypes.c_int32),
              ('begin', ctypes.c_int32), ('end', ctypes.c_int32),
              ('own_begin', ctypes.c_int32), ('own_end', ctypes.c_int32),
              ('ghost_lo', ctypes.c_int32), ('ghost_hi', ctypes.c_int32),
              ('extent', ctypes.c_int32 * MAX_DIM),
              ('inputs', ctypes.c_void_p * MAX_TENSORS),
              ('outputs', ctypes.c_void_p * MAX_TENSORS)]


class GroupStats(ctypes.Structure):
  """Mirror of soda_hip_group_stats_t."""
  _fields_ = [('exchange_every', ctypes.c_int32),
              ('intervals', ctypes.c_int32), ('exchanges', ctypes.c_int32),
              ('copies', ctypes.c_int32), ('copy_bytes', ctypes.c_int64),
              ('launches', ctypes.c_int32), ('split_passes', ctypes.c_int32),
              ('enqueue_ms', ctypes.c_float)]


class HostTensor(ctypes.Structure):
  _fields_ = [('ptr', ctypes.c_void_p),
              ('extent', ctypes.POINTER(ctypes.c_int32)),
              ('stride', ctypes.POINTER(ctypes.c_int32)),
              ('min', ctypes.POINTER(ctypes.c_int32))]


# every symbol include/soda_hip.h declares: name -> (restype, argtypes)
_vp = ctypes.c_void_p
_i32 = ctypes.c_int32
_pvp = ctypes.POINTER(ctypes.c_void_p)
_pi32 = ctypes.POINTER(ctypes.c_int32)
API = {
    'soda_hip_abi_version': (ctypes.c_int, []),
    'soda_hip_status_string': (ctypes.c_char_p, [ctypes.c_int]),
    'soda_hip_last_error': (ctypes.c_size_t, [ctypes.c_char_p, ctypes.c_size_t]),
    'soda_hip_device_count': (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    'soda_hip_sizeof': (ctypes.c_size_t, [ctypes.c_int]),
    'soda_hip_compile': (ctypes.c_int, [
        ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p), _i32,
        _pvp, ctypes.POINTER(ctypes.c_size_t)
    ]),
    'soda_hip_free_code': (None, [_vp]),
    'soda_hip_compiler_version': (ctypes.c_int, [_pi32, _pi32]),
    'soda_hip_plan_geometry': (ctypes.c_int, [
        ctypes.POINTER(Plan), _pi32, _pi32, ctypes.POINTER(ctypes.c_float)
    ]),
    'soda_hip_plan_schedule': (ctypes.c_int, [
        ctypes.POINTER(Plan), _pi32, _i32, _pi32
    ]),
    'soda_hip_program_create': (ctypes.c_int, [
        _vp, ctypes.c_size_t, ctypes.POINTER(Plan), _i32, _pvp
    ]),
    'soda_hip_program_destroy': (ctypes.c_int, [_vp]),
    'soda_hip_run_device': (ctypes.c_int, [_vp, _pvp, _pvp, _pi32, _i32, _vp]),
    'soda_hip_run_device_window': (ctypes.c_int, [_vp, _pvp, _pvp, _pi32, _pi32,
                                                  _pi32, _i32, _vp]),
    'soda_hip_run_device_cone': (ctypes.c_int, [_vp, _pvp, _pvp, _pi32, _pi32,
                                                _pi32, _i32, _i32, _i32, _i32,
                                                _i32, _vp]),
    'soda_hip_run_device_slab': (ctypes.c_int, [_vp, _pvp, _pvp, _pi32, _pi32,
                                                _pi32, _i32,
                                                ctypes.POINTER(SlabRun), _vp]),
    'soda_hip_plan_launches': (ctypes.c_int, [
        ctypes.POINTER(Plan), _pi32, _i32, ctypes.POINTER(SlabRun), _i32,
        ctypes.POINTER(LaunchInfo), _pi32
    ]),
    'soda_hip_last_rows': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int64)]),
    'soda_hip_last_split': (ctypes.c_int, [_vp, _pi32]),
    'soda_hip_group_create': (ctypes.c_int, [
        _vp, ctypes.c_size_t, ctypes.POINTER(Plan), ctypes.POINTER(GroupDesc),
        _pvp
    ]),
    'soda_hip_group_destroy': (ctypes.c_int, [_vp]),
    'soda_hip_group_slab': (ctypes.c_int, [_vp, _i32,
                                           ctypes.POINTER(SlabInfo)]),
    'soda_hip_group_plan': (ctypes.c_int, [
        ctypes.POINTER(Plan), ctypes.POINTER(GroupDesc), _pi32
    ]),
    'soda_hip_group_load': (ctypes.c_int, [_vp, ctypes.POINTER(HostTensor)]),
    'soda_hip_group_loaded': (ctypes.c_int, [_vp]),
    'soda_hip_group_run': (ctypes.c_int, [_vp, _i32]),
    'soda_hip_group_synchronize': (ctypes.c_int, [_vp]),
    'soda_hip_group_store': (ctypes.c_int, [_vp, ctypes.POINTER(HostTensor),
                                            _pi32, _pi32]),
    'soda_hip_group_run_host': (ctypes.c_int, [
        _vp, ctypes.POINTER(HostTensor), ctypes.POINTER(HostTensor), _i32,
        _pi32, _pi32
    ]),
    'soda_hip_group_last_stats': (ctypes.c_int, [
        _vp, ctypes.POINTER(GroupStats)
    ]),
    'soda_hip_memcpy_d2d': (ctypes.c_int, [_vp, _i32, _vp, _i32,
                                           ctypes.c_size_t, _vp]),
    'soda_hip_run_host': (ctypes.c_int, [
        _vp, ctypes.POINTER(HostTensor), ctypes.POINTER(HostTensor), _i32
    ]),
    'soda_hip_run_host_box': (ctypes.c_int, [
        _vp, ctypes.POINTER(HostTensor), ctypes.POINTER(HostTensor), _i32,
        _pi32, _pi32
    ]),
    'soda_hip_host_copy_box': (ctypes.c_int, [
        _vp, _pi32, _vp, _pi32, _pi32, _pi32, _i32, _i32, _i32, _i32, _i32
    ]),
    'soda_hip_host_weave_banks': (ctypes.c_int, [
        ctypes.POINTER(ctypes.c_void_p), _i32, _vp, ctypes.c_int64,
        ctypes.c_int64, _i32, _i32, _i32
    ]),
    'soda_hip_host_register': (ctypes.c_int, [_vp, ctypes.c_size_t]),
    'soda_hip_host_unregister': (ctypes.c_int, [_vp]),
    'soda_hip_last_launches': (ctypes.c_int, [_vp, _pi32, _pi32]),
    'soda_hip_program_set_debug_buffer': (ctypes.c_int, [_vp, _vp]),
    'soda_hip_program_calibrate': (ctypes.c_int, [_vp, _pi32, _i32, _vp]),
    'soda_hip_program_set_auto_calibrate': (ctypes.c_int, [_vp, ctypes.c_int]),
    'soda_hip_program_schedule': (ctypes.c_int, [_vp, _pi32, _i32, _pi32]),
    'soda_hip_program_pass_times': (ctypes.c_int, [
        _vp, _pi32, ctypes.POINTER(ctypes.c_float), _pi32
    ]),
    'soda_hip_stream_create': (ctypes.c_int, [
        ctypes.POINTER(StreamDesc), _vp, _pvp, _pvp, _pvp, _pvp
    ]),
    'soda_hip_stream_destroy': (ctypes.c_int, [_vp]),
    'soda_hip_stream_run_device': (ctypes.c_int, [
        _vp, _pvp, _pvp, ctypes.c_uint64, _vp
    ]),
    'soda_hip_stream_run_host': (ctypes.c_int, [_vp, _pvp, _pvp,
                                                ctypes.c_uint64]),
    'soda_hip_stream_last_mode': (ctypes.c_int, [_vp]),
    'soda_hip_stream_set_device_dense_min_tile': (ctypes.c_int, [_vp, _i32]),
    'soda_hip_malloc': (ctypes.c_int, [_i32, ctypes.c_size_t, _pvp]),
    'soda_hip_free': (ctypes.c_int, [_i32, _vp]),
    'soda_hip_memcpy_h2d': (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp]),
    'soda_hip_memcpy_d2h': (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp]),
    'soda_hip_memset': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_size_t, _vp]),
    'soda_hip_stream_synchronize': (ctypes.c_int, [_vp]),
    'soda_hip_event_create': (ctypes.c_int, [_pvp]),
    'soda_hip_event_record': (ctypes.c_int, [_vp, _vp]),
    'soda_hip_event_elapsed_ms': (ctypes.c_int, [
        _vp, _vp, ctypes.POINTER(ctypes.c_float)
    ]),
    'soda_hip_event_destroy': (ctypes.c_int, [_vp]),
    'soda_hip_event_handle': (ctypes.c_int, [_vp, _pvp]),
    'soda_hip_hipstream_create': (ctypes.c_int, [_i32, _pvp]),
    'soda_hip_hipstream_destroy': (ctypes.c_int, [_vp]),
    'soda_hip_hipstream_wait_event': (ctypes.c_int, [_vp, _vp]),
}

_lib = None


def library() -> ctypes.CDLL:
  """Loads libsoda_hip.so once; fails loudly if it was not built."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise util.BackendError(
          '%s is missing: build it with `python -c "import __graft_entry__ as '
          'g; g.build()"` (hipcc). There is no CPU fallback.' % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own
    # libamdhip64/libhiprtc (same SONAMEs as /opt/rocm's).  Two copies in one
    # process cannot both open the GPU, so when torch is installed it is
    # loaded FIRST and libsoda_hip.so binds to the copy torch brought in.
    if not os.environ.get('SODA_HIP_NO_TORCH'):
      try:
        import torch  # noqa: F401
      except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in API.items():
      fn = getattr(lib, name)
      fn.restype = restype
      fn.argtypes = argtypes
    if lib.soda_hip_abi_version() != ABI_VERSION:
      raise util.BackendError('libsoda_hip.so has ABI version %d, expected %d' %
                              (lib.soda_hip_abi_version(), ABI_VERSION))
    for which, mirror in ((1, KernelDesc), (2, PassDesc), (3, Plan),
                          (4, HostTensor), (5, StreamDesc), (6, SlabRun),
                          (7, GroupDesc), (8, SlabInfo), (9, GroupStats),
                          (10, LaunchInfo)):
      if lib.soda_hip_sizeof(which) != ctypes.sizeof(mirror):
        raise util.BackendError(
            'struct layout mismatch between libsoda_hip.so and runtime.py '
            '(%s: %d vs %d bytes); rebuild the library' %
            (mirror.__name__, lib.soda_hip_sizeof(which),
             ctypes.sizeof(mirror)))
    _lib = lib
  return _lib


def last_error() -> str:
  lib = library()
  n = lib.soda_hip_last_error(None, 0)
  buf = ctypes.create_string_buffer(n + 1)
  lib.soda_hip_last_error(buf, n + 1)
  return buf.value.decode(errors='replace')


def check(status: int, what: str) -> None:
  if status != 0:
    lib = library()
    raise util.BackendError(
        '%s: %s: %s' % (what, lib.soda_hip_status_string(status).decode(),
                        last_error()))


def device_count() -> int:
  n = ctypes.c_int(0)
  if library().soda_hip_device_count(ctypes.byref(n)) != 0:
    return 0
  return n.value


_compiler = None


def compiler_version() -> str:
  """'hiprtc <major>.<minor>' of the library that JIT-compiles, or 'unknown'
  when it cannot be asked (no library built): part of every cache key, since
  another compiler gives the same source other registers and code sizes."""
  global _compiler
  if _compiler is None:
    try:
      a, b = ctypes.c_int32(), ctypes.c_int32()
      if library().soda_hip_compiler_version(ctypes.byref(a),
                                             ctypes.byref(b)) == 0:
        _compiler = 'hiprtc %d.%d' % (a.value, b.value)
      else:
        _compiler = 'unknown'
    except (util.SodaError, OSError):
      _compiler = 'unknown'
  return _compiler


def source_key(source: str, options: Sequence[str] = COMPILE_OPTIONS) -> str:
  """Content hash of a kernel module as it is built: compiler version, compile
  options, source text.  Names the code object in the JIT cache, and ties a
  measurement (profiles/traffic.json) to the kernels it was taken on."""
  return hashlib.sha256(
      (compiler_version() + '\0' + '\0'.join(options) + '\0' +
       source).encode()).hexdigest()[:24]


def compile_source(source: str, name: str = 'soda.hip',
                   options: Sequence[str] = COMPILE_OPTIONS,
                   cache_dir: Optional[str] = None) -> bytes:
  """HIP text -> gfx950 code object, cached on disk by content hash.  Runs
  without a GPU (hiprtc), so `build()` can pre-compile on the CPU box."""
  cache_dir = CACHE_DIR if cache_dir is None else cache_dir
  key = source_key(source, options)
  path = os.path.join(cache_dir, '%s_%s.hsaco' % (name.replace('.', '_'), key))
  if os.path.exists(path):
    with open(path, 'rb') as f:
      return f.read()
  lib = library()
  opts = [o.encode() for o in options]
  arr = (ctypes.c_char_p * len(opts))(*opts)
  code = ctypes.c_void_p()
  size = ctypes.c_size_t()
  check(
      lib.soda_hip_compile(source.encode(), name.encode(), arr, len(opts),
                           ctypes.byref(code), ctypes.byref(size)),
      'JIT of %s' % name)
  try:
    blob = ctypes.string_at(code, size.value)
  finally:
    lib.soda_hip_free_code(code)
  try:
    os.makedirs(cache_dir, exist_ok=True)
    tmp = '%s.%d.tmp' % (path, os.getpid())
    with open(tmp, 'wb') as f:
      f.write(blob)
    os.replace(tmp, path)
  except OSError:
    pass  # read-only tree: run uncached
  return blob


# ns of SIMD issue time per vector instruction of a marching kernel's row step,
# fitted with the other constants of the library's time model (soda_hip.cpp
# "Constants of the launch-time model", tools/fit_model.py); the fused kernels
# sustain 3.2-3.8 cycles per wave64 instruction at ~2.3 GHz with their lane
# shifts (profiles/)
NS_PER_VALU_OP = 1.28
# ... of a kernel whose lane shifts are split between DPP and ds_swizzle
MIXH_FACTOR = 0.849


def make_plan(mod: lower.Module,
              resources: Optional[Dict[str, dict]] = None) -> Plan:
  """The launch plan of a lowered module.  `resources` (kernel_resources of
  the compiled code) supplies the register counts the library sizes chunks
  from; without it chunks stay at their defaults."""
  st = mod.stencil
  table = st.symbol_table
  io_bytes = float(sum(table[n].size_in_bytes for n in st.input_names) +
                   sum(table[n].size_in_bytes for n in st.output_names))
  plan = Plan()
  plan.abi_version = ABI_VERSION
  plan.dim = st.dim
  plan.num_inputs = len(st.input_names)
  plan.num_outputs = len(st.output_names)
  plan.num_locals = len(st.local_names)
  if len(st.param_stmts) > MAX_PARAMS:
    raise util.SemanticError('more than %d param arrays' % MAX_PARAMS)
  plan.num_params = len(st.param_stmts)
  for i, pstmt in enumerate(st.param_stmts):
    plan.param_elems[i] = st.param_elems(pstmt)
  if len(mod.elem_size) > MAX_TENSORS:
    raise util.SemanticError('more than %d tensors' % MAX_TENSORS)
  for i, s in enumerate(mod.elem_size):
    plan.elem_size[i] = s
  if len(mod.kernels) > MAX_KERNELS:
    raise util.SemanticError('more than %d kernels' % MAX_KERNELS)
  plan.num_kernels = len(mod.kernels)
  for i, k in enumerate(mod.kernels):
    if len(k.name) >= NAME_LEN:
      raise util.SemanticError('kernel name too long: %s' % k.name)
    plan.kernels[i].name = k.name.encode()
    for d in range(3):
      plan.kernels[i].block[d] = k.block[d]
    for d in range(MAX_DIM):
      plan.kernels[i].tile[d] = k.tile[d]
    plan.kernels[i].lds_bytes = k.lds_bytes
    d = plan.kernels[i]
    d.window_extra = -1
    tune = k.tune or {}
    d.vec = int(tune.get('vec') or 0)
    if 'axis' in tune:                     # a marching kernel
      d.march_dim = tune['axis'] + 1
      d.waves_along = int(tune.get('waves_along') or 1)
      d.warm = int(tune.get('warm') or 0)
      extra = tune.get('window_extra')
      d.window_extra = -1 if extra is None else int(extra)
      d.max_elem = int(tune.get('max_elem') or 0)
      d.pipe = int(tune.get('pipe') or 1)
      d.chunk_fixed = 1 if tune.get('fixed') else 0
      d.vgprs = int((resources or {}).get(k.name, {}).get('vgpr') or 0)
      d.step_ns = float(tune.get('step_ops') or 0.0) * NS_PER_VALU_OP * (
          MIXH_FACTOR if tune.get('lane_shift') in ('mixh', 'mix64',
                                                    'mix64d') else 1.0)
      d.warm_saved = float(tune.get('warm_saved') or 0.0)
      d.bytes_per_cell = io_bytes
      d.lane_redundancy = float(tune.get('lane_redundancy') or 1.0)
      d.max_extent0 = int(tune.get('max_extent0') or 0)
  passes = mod.sorted_passes()
  if len(passes) > MAX_PASSES:
    raise util.SemanticError('more than %d passes' % MAX_PASSES)
  plan.num_passes = len(passes)
  for i, p in enumerate(passes):
    plan.passes[i].fused_iters = p.fused_iters
    if len(p.kernels) > MAX_PASS_KERNELS:
      raise util.SemanticError(
          'a pass of %d kernels exceeds the limit of %d' %
          (len(p.kernels), MAX_PASS_KERNELS))
    plan.passes[i].num_kernels = len(p.kernels)
    plan.passes[i].cost = 0.0    # marching passes carry a time model instead
    for j, k in enumerate(p.kernels):
      plan.passes[i].kernel[j] = k
  plan.has_reach = 1
  plan.reach_lo, plan.reach_hi = st.reach_along(st.dim - 1)
  return plan


def kernel_resources(code: bytes) -> Dict[str, dict]:
  """Per-kernel register/LDS use read from the code object's AMDGPU metadata
  note (msgpack): {kernel name: {'vgpr': n, 'sgpr': n, 'lds': bytes,
  'scratch': bytes, 'code': bytes of machine code}}.  Returns {} if the note
  cannot be read."""
  import struct
  try:
    import msgpack
    if code[:4] != b'\x7fELF':
      return {}
    try:
      sizes = _function_sizes(code)
    except Exception:
      sizes = {}
    shoff, = struct.unpack_from('<Q', code, 0x28)
    shentsize, shnum = struct.unpack_from('<HH', code, 0x3A)
    for i in range(shnum):
      off = shoff + i * shentsize
      sh_type, = struct.unpack_from('<I', code, off + 4)
      if sh_type != 7:       # SHT_NOTE
        continue
      sec_off, sec_size = struct.unpack_from('<QQ', code, off + 0x18)
      pos, end = sec_off, sec_off + sec_size
      while pos + 12 <= end:
        namesz, descsz, ntype = struct.unpack_from('<III', code, pos)
        pos += 12
        name = code[pos:pos + namesz].rstrip(b'\0')
        pos += (namesz + 3) & ~3
        desc = code[pos:pos + descsz]
        pos += (descsz + 3) & ~3
        if name == b'AMDGPU' and ntype == 32:     # NT_AMDGPU_METADATA
          meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
          out = {}
          for k in meta.get('amdhsa.kernels', []):
            out[k['.name']] = dict(
                vgpr=k.get('.vgpr_count', 0) + 0, sgpr=k.get('.sgpr_count', 0),
                agpr=k.get('.agpr_count', 0),
                lds=k.get('.group_segment_fixed_size', 0),
                scratch=k.get('.private_segment_fixed_size', 0),
                code=sizes.get(k['.name'], 0))
          return out
  except Exception:
    return {}
  return {}


def _function_sizes(code: bytes) -> Dict[str, int]:
  """{function symbol: bytes of machine code} from the ELF symbol table."""
  import struct
  out: Dict[str, int] = {}
  shoff, = struct.unpack_from('<Q', code, 0x28)
  shentsize, shnum = struct.unpack_from('<HH', code, 0x3A)
  heads = [struct.unpack_from('<IIQQQQIIQQ', code, shoff + i * shentsize)
           for i in range(shnum)]
  for h in heads:
    if h[1] != 2:                     # SHT_SYMTAB
      continue
    str_off = heads[h[6]][4]          # sh_link -> its string table
    for pos in range(h[4], h[4] + h[5], 24):
      st_name, st_info, _, _, _, st_size = struct.unpack_from('<IBBHQQ', code, pos)
      if st_info & 0xF == 2:          # STT_FUNC
        end = code.index(b'\0', str_off + st_name)
        out[code[str_off + st_name:end].decode()] = st_size
  return out


NUM_CUS = 256          # MI355X: 8 XCDs x 32 CUs, 4 SIMDs each
MAX_WAVES_PER_SIMD = 8


def waves_per_simd(vgprs: int) -> int:
  """Occupancy the register file allows (MI355X_MICROARCH.md, Register
  files: 512 VGPRs per lane per SIMD, allocation granule 8)."""
  alloc = max(8, -(-vgprs // 8) * 8)
  return max(1, min(MAX_WAVES_PER_SIMD, 512 // alloc))


def plan_geometry(plan: Plan, extent: Sequence[int]):
  """(tile of every kernel, modelled ns of every pass) the library would use
  for `extent` -- soda_hip_plan_geometry; no GPU needed.  Raises BackendError
  if a kernel of the plan cannot run this extent."""
  lib = library()
  ext = (ctypes.c_int32 * MAX_DIM)(*(list(extent) + [1] * (MAX_DIM - len(extent))))
  tiles = (ctypes.c_int32 * (MAX_DIM * plan.num_kernels))()
  ns = (ctypes.c_float * plan.num_passes)()
  check(lib.soda_hip_plan_geometry(ctypes.byref(plan), ext, tiles, ns),
        'launch geometry for extent %s' % (tuple(extent),))
  return ([tuple(tiles[k * MAX_DIM:(k + 1) * MAX_DIM])
           for k in range(plan.num_kernels)], list(ns))


def plan_schedule(plan: Plan, extent: Sequence[int], iterate: int):
  """How many times each pass runs for `iterate` iterations on `extent`."""
  lib = library()
  ext = (ctypes.c_int32 * MAX_DIM)(*(list(extent) + [1] * (MAX_DIM - len(extent))))
  count = (ctypes.c_int32 * plan.num_passes)()
  check(lib.soda_hip_plan_schedule(ctypes.byref(plan), ext, iterate, count),
        'schedule of %d iterations' % iterate)
  return list(count)


def plan_launches(plan: Plan, extent: Sequence[int], iterate: int,
                  run: Optional[SlabRun] = None):
  """The launches a run on `extent` would issue (soda_hip_plan_launches): a
  list of dicts, one per launch, in order.  No GPU needed."""
  lib = library()
  ext = (ctypes.c_int32 * MAX_DIM)(*(list(extent) + [1] * (MAX_DIM - len(extent))))
  n = ctypes.c_int32(0)
  cap = 4096
  arr = (LaunchInfo * cap)()
  check(lib.soda_hip_plan_launches(ctypes.byref(plan), ext, iterate,
                                   ctypes.byref(run) if run is not None else
                                   None, cap, arr, ctypes.byref(n)),
        'launches of %d iterations' % iterate)
  return [{name: getattr(arr[i], name) for name, _ in LaunchInfo._fields_}
          for i in range(min(n.value, cap))]


def pick_vec(stencil: core.Stencil, extent: Optional[Sequence[int]]) -> int:
  """Cells per lane per row: 16 bytes' worth, reduced until it divides the
  row length (rows must stay 16-byte aligned for the vector loads)."""
  vec = lower.default_vec(stencil)
  if extent is not None:
    while vec > 1 and extent[0] % vec:
      vec //= 2
  return vec


ICACHE_BYTES = 60 * 1024   # instruction cache a kernel's loop should stay in


def select_peel(stencil: core.Stencil, opts: 'lower.LowerOptions',
                extent: Optional[Sequence[int]] = None) -> Dict[int, int]:
  """{fusion depth: trips of the loop whose warm-up to peel} for the marching
  kernels of `stencil` under `opts` (MarchConfig.peel).  Peeling skips stages
  that do not matter yet -- less work per chunk -- but the late, nearly full
  trips buy little and the straight-line code can need more registers than the
  loop.  Occupancy decides: per depth, the largest trip count whose COMPILED
  kernel keeps the waves per SIMD of the unpeeled kernel and spills nothing
  (jacobi2d T=12: 2 of 4 trips, 158 VGPRs; all 4 would take 231).  Code size
  decides too: the peeled steps are whole trips of the unrolled loop, and a
  3-D kernel's trip is big -- heat3d T=2: 42 KB of code unpeeled, 73 KB with
  its one trip peeled, past the instruction cache, 220-230 us against 202-205
  on 512^3.  Where chunks are long (>= 8x the warm-up, judged by the library's
  geometry for `extent`) the warm-up is a few percent and the smaller code
  wins; on a thin slab (80 planes: chunks of 8-10) peeling wins, 38 against
  69 us, also because the peeled kernel happens to need fewer registers.
  Trials are JIT-compiled (no GPU needed) and remembered in the kernel
  cache."""
  import copy
  import json
  plain = copy.copy(opts)
  plain.peel = 0
  mod0 = lower.lower(stencil, plain)
  todo = {}
  for k in mod0.kernels:
    if k.tune and k.tune.get('peel_trips_max'):
      todo[k.tune['fused']] = k.tune['peel_trips_max']
  if not todo:
    return {}
  # long chunks on this extent?  (geometry of the unpeeled kernels)
  long_chunks = {}
  res0 = kernel_resources(compile_source(mod0.source,
                                         '%s.hip' % stencil.app_name))
  if extent is not None:
    try:
      tiles, _ = plan_geometry(make_plan(mod0, res0), extent)
      for k, tile in zip(mod0.kernels, tiles):
        if k.tune and k.tune.get('fused') in todo:
          long_chunks[k.tune['fused']] = \
              tile[k.tune['axis']] >= 8 * max(1, k.tune.get('warm') or 1)
    except util.SodaError:
      pass
  key = hashlib.sha256(('peel3\0' + compiler_version() + '\0' +
                        '\0'.join(COMPILE_OPTIONS) + '\0' +
                        repr(sorted(long_chunks.items())) + '\0' +
                        mod0.source).encode()).hexdigest()[:24]
  memo = os.path.join(CACHE_DIR, 'peel_%s.json' % key)
  try:
    with open(memo) as f:
      return {int(t): int(v) for t, v in json.load(f).items()}
  except (OSError, ValueError):
    pass

  def probe(depth: int, trips: int):
    if trips == 0:
      res, mod = res0, mod0
    else:
      one = copy.copy(opts)
      one.fuse = (depth,) if depth > 1 else ()
      one.peel = trips
      mod = lower.lower(stencil, one)
      res = kernel_resources(compile_source(mod.source,
                                            '%s.hip' % stencil.app_name))
    for k in mod.kernels:
      if k.tune and k.tune.get('fused') == depth:
        r = res.get(k.name)
        return None if not r else (waves_per_simd(r['vgpr']), r['scratch'],
                                   r.get('code', 0))
    return None

  chosen = {}
  for depth, most in todo.items():
    base = probe(depth, 0)
    chosen[depth] = 0
    if base is None:
      continue
    for trips in range(most, 0, -1):
      got = probe(depth, trips)
      if got is None or got[0] < base[0] or got[1] > base[1]:
        continue
      if long_chunks.get(depth) and got[2] > ICACHE_BYTES >= base[2]:
        continue          # the warm-up is small change here; keep the code small
      chosen[depth] = trips
      break
  try:
    os.makedirs(CACHE_DIR, exist_ok=True)
    tmp = '%s.%d.tmp' % (memo, os.getpid())
    with open(tmp, 'w') as f:
      json.dump(chosen, f)
    os.replace(tmp, memo)
  except OSError:
    pass
  return chosen


def select_shape(stencil: core.Stencil, opts: 'lower.LowerOptions',
                 extent: Optional[Sequence[int]] = None):
  """(cells per lane, prefetch depth) for programs whose register windows the
  shape ladder in lower() OVER-estimates, or None to leave its choice alone.
  The ladder counts a register per cell held; 16-bit cells are packed two to a
  register, so for tall windows of narrow integers (erosion and xcorr hold 19
  rows of int16 plus their partial-window tensors) it retreats to 2 cells per
  lane -- 8-byte loads, 129 us on 8192^2 -- where 8 cells per lane COMPILE to
  243-247 registers without spilling and run in 60-67 us
  (profiles/r03_windows_sweep.jsonl).  So, like select_peel, ask the compiler:
  the widest shape above the ladder's choice whose compiled one-iteration
  kernel fits the register file -- at most 512 registers per lane, accumulation
  registers included: one wave per SIMD (xcorr at 8 cells per lane: 314, still
  faster than 4 cells per lane at 160) -- and needs no scratch.  Trials are
  JIT-compiled (no GPU) and remembered in the kernel cache."""
  import copy
  import json
  if opts.strategy not in ('auto', 'march') or stencil.iterate > 1:
    return None
  # (the over-estimate comes from packed narrow cells only: a program of
  # 32-bit cells is left to the ladder -- and spared the trial compilations,
  # minutes for a 197-tap program like contrast)
  if min(t.size_in_bytes for t in stencil.symbol_table.values()) >= 4:
    return None
  base = copy.copy(opts)
  base.peel = 0
  mod0 = lower.lower(stencil, base)
  one = [k for k in mod0.kernels if k.tune and k.tune.get('fused') == 1 and
         'axis' in k.tune]
  if not one:
    return None
  chosen = int(one[0].tune.get('vec') or 1)
  want = int(opts.vec or lower.default_vec(stencil))
  if chosen >= want:
    return None
  key = hashlib.sha256(('shape2\0' + compiler_version() + '\0' +
                        '\0'.join(COMPILE_OPTIONS) + '\0%d\0' % want +
                        mod0.source).encode()).hexdigest()[:24]
  memo = os.path.join(CACHE_DIR, 'shape_%s.json' % key)
  try:
    with open(memo) as f:
      got = json.load(f)
    return tuple(got) if got else None
  except (OSError, ValueError):
    pass
  found = None
  vec = want
  while vec > chosen and found is None:
    for pf in (2, 1):
      trial = copy.copy(base)
      trial.vec, trial.prefetch, trial.reg_budget = vec, pf, 1 << 20
      try:
        mod = lower.lower(stencil, trial)
        res = kernel_resources(compile_source(mod.source,
                                              '%s.hip' % stencil.app_name))
      except util.SodaError:
        continue
      ks = [k for k in mod.kernels if k.tune and k.tune.get('fused') == 1]
      if not ks or int(ks[0].tune.get('vec') or 0) != vec:
        continue
      r = res.get(ks[0].name)
      # (the count includes accumulation registers used as spill space: up
      # to 512 a wave still runs, alone on its SIMD -- xcorr at 8 cells per
      # lane: 314, 60.7 us on 8192^2, against 84.7 us at 4 cells and 160)
      if r and r['scratch'] == 0 and 0 < r['vgpr'] <= 512:
        found = (vec, pf)
        break
    vec //= 2
  try:
    os.makedirs(CACHE_DIR, exist_ok=True)
    tmp = '%s.%d.tmp' % (memo, os.getpid())
    with open(tmp, 'w') as f:
      json.dump(list(found) if found else [], f)
    os.replace(tmp, memo)
  except OSError:
    pass
  return found


def resolve_options(stencil: core.Stencil,
                    opts: Optional['lower.LowerOptions'],
                    extent: Optional[Sequence[int]],
                    probe: bool = True) -> 'lower.LowerOptions':
  """A private copy of the caller's options with everything this module
  decides filled in: the vector width for `extent`, the peel depths.  The peel
  depths come from trial compilations (select_peel), which need the library
  and hiprtc: with probe=False, or where they are not to be had (a text-only
  use such as `sodac --hip-kernel` on a box without the library), the warm-up
  is not peeled -- deterministic, correct, a few percent slower."""
  import copy
  out = copy.copy(opts) if opts is not None else lower.LowerOptions()
  if out.vec is None:
    out.vec = pick_vec(stencil, extent)
    if out.prefetch is None and out.strategy in ('auto', 'march') and \
        lower.prefers_narrow_strips(stencil):
      # arithmetic-heavy one-iteration 2-D fp32 program: occupancy over width
      out.vec, out.prefetch = min(out.vec, 2), 4
  if out.row_cells is None and extent is not None:
    out.row_cells = int(extent[0])   # lets blocks cover whole rows (xshare)
  probing = probe and not os.environ.get('SODA_HIP_NO_PROBE')
  if probing and out.prefetch is None and out.reg_budget is None and \
      out.strategy in ('auto', 'march') and \
      lower.march_supported(stencil) is None:
    try:
      shape = select_shape(stencil, out, extent)
    except (util.SodaError, OSError):
      shape = None
    if shape:
      out.vec, out.prefetch = shape
      out.reg_budget = 1 << 20
  if out.peel is None and out.strategy in ('auto', 'march') and \
      lower.march_supported(stencil) is None:
    if not probe or os.environ.get('SODA_HIP_NO_PROBE'):
      out.peel = 0
    else:
      try:
        out.peel = select_peel(stencil, out, extent)
      except (util.SodaError, OSError):
        out.peel = 0
  return out


class Program:
  """A SODA program JIT-built for gfx950 and loaded on one GPU."""

  def __init__(self, stencil: core.Stencil,
               opts: Optional[lower.LowerOptions] = None, device: int = 0,
               extent: Optional[Sequence[int]] = None,
               calibrate: Optional[bool] = None,
               source_prefix: str = ''):
    """`calibrate`: True -- time the passes on `extent` now; None (default) --
    the library does it by itself on the first run of an extent (a few ms,
    once; soda_hip_program_set_auto_calibrate); False -- never: runs are
    scheduled by the model.  `source_prefix`: text compiled in front of the
    module (diagnostic builds only: tests/test_compiler_pins.py switches a
    compiler-fault workaround of soda_rt.h off with a #define)."""
    self.stencil = stencil
    self.opts = resolve_options(stencil, opts, extent)   # never the caller's
    self.device = device
    self.module = lower.lower(stencil, self.opts)
    self.code = compile_source(source_prefix + self.module.source,
                               '%s.hip' % stencil.app_name)
    self.resources = kernel_resources(self.code)
    # launch geometry (chunk lengths, pass schedule) is the library's, decided
    # per run from the extent it is given and the register counts in the plan
    self.plan = make_plan(self.module, self.resources)
    if extent is not None:
      self.geometry(extent)            # fail early if it cannot run
    self._lib = library()
    self._handle = ctypes.c_void_p()
    check(
        self._lib.soda_hip_program_create(self.code, len(self.code),
                                          ctypes.byref(self.plan), device,
                                          ctypes.byref(self._handle)),
        'loading `%s` on GPU %d' % (stencil.app_name, device))
    if calibrate is False:
      check(self._lib.soda_hip_program_set_auto_calibrate(self._handle, 0),
            'set_auto_calibrate')
    if calibrate and extent is not None:
      self.calibrate(extent)

  # -- launch geometry ------------------------------------------------------
  def geometry(self, extent: Sequence[int]):
    """What a run on `extent` uses: ({kernel name: tile}, {fused iterations
    of a pass: modelled microseconds})."""
    tiles, ns = plan_geometry(self.plan, extent)
    passes = self.module.sorted_passes()
    return ({k.name: t[:self.stencil.dim]
             for k, t in zip(self.module.kernels, tiles)},
            {p.fused_iters: v / 1e3 for p, v in zip(passes, ns)})

  def calibrate(self, extent: Sequence[int], launches: int = 4,
                stream: int = 0) -> Dict[int, float]:
    """Times one launch of every pass on `extent` on the GPU (a few ms, once)
    so that runs on this extent are scheduled by the clock; returns {fused
    iterations: microseconds}."""
    ext = (ctypes.c_int32 * MAX_DIM)(*(list(extent) +
                                        [1] * (MAX_DIM - len(extent))))
    check(self._lib.soda_hip_program_calibrate(self._handle, ext, launches,
                                               ctypes.c_void_p(stream)),
          'calibrating `%s`' % self.stencil.app_name)
    return self.pass_times(extent)[0]

  def pass_times(self, extent: Sequence[int]):
    """({fused iterations: microseconds per launch}, measured?)"""
    ext = (ctypes.c_int32 * MAX_DIM)(*(list(extent) +
                                        [1] * (MAX_DIM - len(extent))))
    ns = (ctypes.c_float * self.plan.num_passes)()
    measured = ctypes.c_int32(0)
    check(self._lib.soda_hip_program_pass_times(self._handle, ext, ns,
                                                ctypes.byref(measured)),
          'pass_times')
    return ({p.fused_iters: v / 1e3
             for p, v in zip(self.module.sorted_passes(), ns)},
            bool(measured.value))

  def schedule(self, extent: Sequence[int], iterate: int) -> Dict[int, int]:
    """{fused iterations of a pass: launches} for `iterate` iterations."""
    ext = (ctypes.c_int32 * MAX_DIM)(*(list(extent) +
                                        [1] * (MAX_DIM - len(extent))))
    count = (ctypes.c_int32 * self.plan.num_passes)()
    check(self._lib.soda_hip_program_schedule(self._handle, ext, iterate,
                                              count), 'schedule')
    return {p.fused_iters: c
            for p, c in zip(self.module.sorted_passes(), count) if c}

  # -- lifecycle -----------------------------------------------------------
  def close(self) -> None:
    if getattr(self, '_handle', None):
      self._lib.soda_hip_program_destroy(self._handle)
      self._handle = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # interpreter shutdown
      pass

  def __enter__(self):
    return self

  def __exit__(self, *exc):
    self.close()

  # -- checks shared by both entry points ----------------------------------
  def _check_extent(self, extent: Sequence[int]) -> None:
    if len(extent) != self.stencil.dim:
      raise util.InputError('extent must have %d entries' % self.stencil.dim)
    # (vector width and buffer-window limits are the library's to check:
    # soda_hip_plan_geometry, also behind every run entry of the C ABI)

  # -- device-resident arrays (the <app>_kernel analogue) ------------------
  def run_device(self, outputs: Sequence[int], inputs: Sequence[int],
                 extent: Sequence[int], iterate: Optional[int] = None,
                 stream: int = 0, origin: Optional[Sequence[int]] = None,
                 global_extent: Optional[Sequence[int]] = None,
                 keep: Optional[Sequence[int]] = None,
                 ghosts: Optional[Sequence[int]] = None,
                 sends: Optional[Sequence[int]] = None,
                 ghosts_ready: int = 0, sendable: int = 0) -> None:
    """`outputs` / `inputs` are device addresses (e.g. tensor.data_ptr()) of
    dense dim-0-fastest arrays; asynchronous on `stream`.  `inputs` holds the
    input tensors followed by the program's `param` arrays (C order).  For a
    slab of a larger grid pass where its cell 0 sits (`origin`) and the size of
    the whole grid (`global_extent`): `border: preserve` means the GLOBAL border.
    `keep` = (lo, hi): only cells [lo, hi) along the last dimension of the
    result are needed (a slab's own rows); passes then skip the rows nothing
    can carry into that range any more (soda_hip_run_device_cone).  NOTE: the
    last pass still writes up to `fused iterations x reach` rows on either
    side of [lo, hi) -- computed with zeros where the launch ended, i.e. wrong
    -- and leaves the rows beyond those untouched: outside [lo, hi) the output
    arrays are unspecified, not "unchanged".
    With `keep`, a halo exchange can run underneath (soda_hip_run_device_slab):
    `ghosts` = (lo, hi) rows at either end of the INPUT arrays that an exchange
    on another stream is writing, complete when the hipEvent_t `ghosts_ready`
    fires; `sends` = (lo, hi) rows at either end of the kept range the
    neighbours fetch next, complete when `sendable` (recorded by this call)
    fires."""
    st = self.stencil
    iterate = st.iterate if iterate is None else iterate
    self._check_extent(extent)
    if len(outputs) != len(st.output_names) or len(inputs) != len(
        st.input_names) + len(st.param_stmts):
      raise util.InputError('wrong number of tensors')
    outs = (ctypes.c_void_p * len(outputs))(*outputs)
    ins = (ctypes.c_void_p * len(inputs))(*inputs)
    ext = (ctypes.c_int32 * len(extent))(*extent)
    org = (ctypes.c_int32 * len(extent))(*(origin or [0] * len(extent)))
    gext = (ctypes.c_int32 * len(extent))(*(global_extent or extent))
    if ghosts is not None or sends is not None or ghosts_ready or sendable:
      keep = keep if keep is not None else (0, extent[-1])
      reach_lo, reach_hi = st.reach_along(st.dim - 1)
      run = SlabRun(int(keep[0]), int(keep[1]), reach_lo, reach_hi,
                    *(int(v) for v in (ghosts or (0, 0))),
                    *(int(v) for v in (sends or (0, 0))),
                    ghosts_ready or None, sendable or None)
      check(
          self._lib.soda_hip_run_device_slab(
              self._handle, outs, ins, ext, org, gext, iterate,
              ctypes.byref(run), ctypes.c_void_p(stream)),
          'running `%s`' % st.app_name)
      return
    if keep is not None and tuple(keep) != (0, extent[-1]):
      reach_lo, reach_hi = st.reach_along(st.dim - 1)
      check(
          self._lib.soda_hip_run_device_cone(
              self._handle, outs, ins, ext, org, gext, iterate, int(keep[0]),
              int(keep[1]), reach_lo, reach_hi, ctypes.c_void_p(stream)),
          'running `%s`' % st.app_name)
      return
    check(
        self._lib.soda_hip_run_device_window(self._handle, outs, ins, ext, org,
                                             gext, iterate,
                                             ctypes.c_void_p(stream)),
        'running `%s`' % st.app_name)

  def last_split(self) -> int:
    """Passes of the last run launched in two parts around a halo exchange."""
    n = ctypes.c_int32()
    check(self._lib.soda_hip_last_split(self._handle, ctypes.byref(n)),
          'last_split')
    return n.value

  def last_rows(self) -> int:
    """Cells along the last dimension the passes of the last run covered,
    summed over the passes."""
    n = ctypes.c_int64()
    check(self._lib.soda_hip_last_rows(self._handle, ctypes.byref(n)),
          'last_rows')
    return n.value

  def set_debug_buffer(self, ptr: int) -> None:
    """Device buffer the time stamps of `stamps=True` kernels go to."""
    check(self._lib.soda_hip_program_set_debug_buffer(self._handle,
                                                      ctypes.c_void_p(ptr)),
          'set_debug_buffer')

  def last_launches(self):
    a, b = ctypes.c_int32(), ctypes.c_int32()
    check(self._lib.soda_hip_last_launches(self._handle, ctypes.byref(a),
                                           ctypes.byref(b)), 'last_launches')
    return a.value, b.value

  # -- host arrays (the soda::app::<app> analogue) -------------------------
  def run(self, inputs: Dict[str, 'numpy.ndarray'],
          iterate: Optional[int] = None,
          outputs: Optional[Dict[str, 'numpy.ndarray']] = None
          ) -> Dict[str, 'numpy.ndarray']:
    """numpy in, numpy out.  Array shape is extent reversed (dim 0 fastest =
    last axis).  Only the valid box of each output is written, the rest keeps
    what the caller's array held (zeros for arrays allocated here)."""
    import numpy as np
    st = self.stencil
    iterate = st.iterate if iterate is None else iterate
    first = inputs[st.input_names[0]]
    extent = tuple(first.shape[::-1])
    self._check_extent(extent)
    dim = st.dim
    keep = []

    def describe(arr, np_name):
      if arr.dtype != np.dtype(np_name):
        raise util.InputError('expected dtype %s, got %s' % (np_name, arr.dtype))
      if arr.shape != first.shape:
        raise util.InputError('all tensors must share one shape')
      item = arr.dtype.itemsize
      strides = [s // item for s in arr.strides[::-1]]
      if any(s * item != b for s, b in zip(strides, arr.strides[::-1])):
        raise util.InputError('strides must be whole elements')
      ext = (ctypes.c_int32 * dim)(*extent)
      strd = (ctypes.c_int32 * dim)(*strides)
      mn = (ctypes.c_int32 * dim)(*([0] * dim))
      keep.extend((ext, strd, mn, arr))
      return HostTensor(arr.ctypes.data, ext, strd, mn)

    in_list = [describe(np.asarray(inputs[n]), t.np_name)
               for n, t in zip(st.input_names, st.input_types)]
    for pstmt in st.param_stmts:        # param arrays follow, C order
      arr = np.ascontiguousarray(inputs[pstmt.name]).reshape(-1)
      if arr.dtype != np.dtype(pstmt.haoda_type.np_name) or \
          arr.size != st.param_elems(pstmt):
        raise util.InputError('param %s must be %d x %s' % (
            pstmt.name, st.param_elems(pstmt), pstmt.haoda_type.np_name))
      keep.append(arr)
      in_list.append(HostTensor(arr.ctypes.data, None, None, None))
    ins = (HostTensor * len(in_list))(*in_list)
    result = {}
    for n, t in zip(st.output_names, st.output_types):
      if outputs is not None and n in outputs:
        result[n] = outputs[n]
      else:
        result[n] = np.zeros(first.shape, dtype=np.dtype(t.np_name))
    outs = (HostTensor * len(st.output_names))(*[
        describe(result[n], t.np_name)
        for n, t in zip(st.output_names, st.output_types)
    ])
    lo, hi = [], []
    for n in st.output_names:
      l, h = st.valid_box(extent, n, iterate)
      lo.extend(l)
      hi.extend(max(a, b) for a, b in zip(h, l))
    vlo = (ctypes.c_int32 * len(lo))(*lo)
    vhi = (ctypes.c_int32 * len(hi))(*hi)
    check(
        self._lib.soda_hip_run_host_box(self._handle, ins, outs, iterate, vlo,
                                        vhi), 'running `%s`' % st.app_name)
    return result


def _host_tensor(arr, np_name: str, shape, dim: int, keep: list):
  """HostTensor for a numpy array (shape = extent reversed), keeping the ctypes
  arrays it points to alive in `keep`."""
  import numpy as np
  if arr.dtype != np.dtype(np_name):
    raise util.InputError('expected dtype %s, got %s' % (np_name, arr.dtype))
  if tuple(arr.shape) != tuple(shape):
    raise util.InputError('all tensors must share one shape')
  item = arr.dtype.itemsize
  strides = [b // item for b in arr.strides[::-1]]
  if any(v * item != b for v, b in zip(strides, arr.strides[::-1])):
    raise util.InputError('strides must be whole elements')
  ext = (ctypes.c_int32 * dim)(*shape[::-1])
  strd = (ctypes.c_int32 * dim)(*strides)
  mn = (ctypes.c_int32 * dim)(*([0] * dim))
  keep.extend((ext, strd, mn, arr))
  return HostTensor(arr.ctypes.data, ext, strd, mn)


class Group:
  """A SODA program on N GPUs driven by one host thread: the grid cut into
  slabs along the streamed dimension, halo exchange by peer copies hidden under
  the compute (soda_hip_group_*, include/soda_hip.h).  `devices` may name one
  GPU several times ("virtual devices"): the whole N-slab schedule then runs on
  that GPU -- how a one-GPU box tests it."""

  def __init__(self, stencil: core.Stencil, extent: Sequence[int],
               devices: Sequence[int],
               opts: Optional[lower.LowerOptions] = None,
               iterate: Optional[int] = None, exchange_every: int = 0,
               overlap: bool = True, calibrate: bool = False,
               threads: Optional[bool] = None):
    """`threads`: one enqueueing thread per slab inside the library (default:
    when the slabs sit on more than one GPU)."""
    self.stencil = stencil
    self.extent = tuple(int(e) for e in extent)
    self.devices = tuple(int(d) for d in devices)
    if len(self.extent) != stencil.dim:
      raise util.InputError('extent must have %d entries' % stencil.dim)
    if not 1 <= len(self.devices) <= MAX_SLABS:
      raise util.InputError('1 to %d slabs' % MAX_SLABS)
    iterate = stencil.iterate if iterate is None else iterate
    n = len(self.devices)
    reach_lo, reach_hi = stencil.reach_along(stencil.dim - 1)
    desc = GroupDesc()
    desc.num_slabs = n
    for i, d in enumerate(self.devices):
      desc.device[i] = d
    for i, e in enumerate(self.extent):
      desc.extent[i] = e
    desc.reach_lo, desc.reach_hi = reach_lo, reach_hi
    desc.iterate = iterate
    desc.exchange_every = exchange_every
    if threads is None:
      threads = len(set(self.devices)) > 1
    desc.flags = (0 if overlap else GROUP_NO_OVERLAP) | (
        GROUP_CALIBRATE if calibrate else 0) | (
            GROUP_THREADS if threads else 0)
    self._lib = library()
    # The kernels are shaped for the extent of a slab (row-covering blocks,
    # how much warm-up to peel); a slab's extent depends on the exchange
    # interval, which the library picks from the plan's pass times: lower once
    # for the own rows, ask, lower again for the rows really held.
    own = self.extent[-1] // n
    every = ctypes.c_int32(exchange_every)
    local = self.extent[:-1] + (max(1, own),)
    for _ in range(2):
      self.opts = resolve_options(stencil, opts, local)
      self.module = lower.lower(stencil, self.opts)
      self.code = compile_source(self.module.source,
                                 '%s.hip' % stencil.app_name)
      self.resources = kernel_resources(self.code)
      self.plan = make_plan(self.module, self.resources)
      check(self._lib.soda_hip_group_plan(ctypes.byref(self.plan),
                                          ctypes.byref(desc),
                                          ctypes.byref(every)),
            'slabs of `%s`' % stencil.app_name)
      ghosts = (reach_lo + reach_hi if n > 2 else max(reach_lo, reach_hi)
                if n > 1 else 0) * every.value
      grown = self.extent[:-1] + (own + ghosts,)
      if grown == local:
        break
      local = grown
    # The model's interval shaped the kernels.  With calibrate=True and no
    # interval of the caller's, create is handed 0 so that the LIBRARY times
    # the model's best candidates on a middle slab's GPU and lets the clock
    # pick (soda_hip_group_create: exchange_every == 0 && GROUP_CALIBRATE);
    # the kernels run on any extent, only their peel / chunk choices were made
    # for the model's one.
    desc.exchange_every = 0 if (calibrate and exchange_every == 0) else \
        every.value
    self.exchange_every = every.value
    self._handle = ctypes.c_void_p()
    check(
        self._lib.soda_hip_group_create(self.code, len(self.code),
                                        ctypes.byref(self.plan),
                                        ctypes.byref(desc),
                                        ctypes.byref(self._handle)),
        'loading `%s` on GPUs %s' % (stencil.app_name, list(self.devices)))
    # (with calibrate=True the clock may have picked another interval)
    self.exchange_every = int(self.stats()['exchange_every'])

  def close(self) -> None:
    if getattr(self, '_handle', None):
      self._lib.soda_hip_group_destroy(self._handle)
      self._handle = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # interpreter shutdown
      pass

  def __enter__(self):
    return self

  def __exit__(self, *exc):
    self.close()

  def slab(self, i: int) -> SlabInfo:
    info = SlabInfo()
    check(self._lib.soda_hip_group_slab(self._handle, i, ctypes.byref(info)),
          'slab %d' % i)
    return info

  def load(self, inputs: Dict[str, 'numpy.ndarray']) -> None:
    """Scatters the global input arrays (and params) over the slabs."""
    import numpy as np
    st = self.stencil
    shape = self.extent[::-1]
    keep: list = []
    tensors = [_host_tensor(np.asarray(inputs[n]), t.np_name, shape, st.dim,
                            keep)
               for n, t in zip(st.input_names, st.input_types)]
    for pstmt in st.param_stmts:
      arr = np.ascontiguousarray(inputs[pstmt.name]).reshape(-1)
      if arr.dtype != np.dtype(pstmt.haoda_type.np_name) or \
          arr.size != st.param_elems(pstmt):
        raise util.InputError('param %s must be %d x %s' % (
            pstmt.name, st.param_elems(pstmt), pstmt.haoda_type.np_name))
      keep.append(arr)
      tensors.append(HostTensor(arr.ctypes.data, None, None, None))
    arr_t = (HostTensor * len(tensors))(*tensors)
    check(self._lib.soda_hip_group_load(self._handle, arr_t), 'group load')

  def loaded(self) -> None:
    """The caller filled the slabs' input arrays on the devices itself."""
    check(self._lib.soda_hip_group_loaded(self._handle), 'group loaded')

  def run(self, iterate: Optional[int] = None) -> None:
    """Enqueues `iterate` iterations on every slab; asynchronous."""
    iterate = self.stencil.iterate if iterate is None else iterate
    check(self._lib.soda_hip_group_run(self._handle, iterate),
          'running `%s` on %d slabs' % (self.stencil.app_name,
                                        len(self.devices)))

  def synchronize(self) -> None:
    check(self._lib.soda_hip_group_synchronize(self._handle),
          'group synchronize')

  def store(self, iterate: Optional[int] = None,
            outputs: Optional[Dict[str, 'numpy.ndarray']] = None
            ) -> Dict[str, 'numpy.ndarray']:
    """Gathers the results; only the valid box of `iterate` iterations of each
    output is written (the rest keeps what the caller's array held, zeros for
    arrays allocated here)."""
    import numpy as np
    st = self.stencil
    iterate = st.iterate if iterate is None else iterate
    shape = self.extent[::-1]
    keep: list = []
    result = {}
    for n, t in zip(st.output_names, st.output_types):
      result[n] = outputs[n] if outputs is not None and n in outputs else \
          np.zeros(shape, dtype=np.dtype(t.np_name))
    outs = (HostTensor * len(st.output_names))(*[
        _host_tensor(result[n], t.np_name, shape, st.dim, keep)
        for n, t in zip(st.output_names, st.output_types)])
    lo, hi = [], []
    for n in st.output_names:
      l, h = st.valid_box(self.extent, n, iterate)
      lo.extend(l)
      hi.extend(max(a, b) for a, b in zip(h, l))
    vlo = (ctypes.c_int32 * len(lo))(*lo)
    vhi = (ctypes.c_int32 * len(hi))(*hi)
    check(self._lib.soda_hip_group_store(self._handle, outs, vlo, vhi),
          'group store')
    return result

  def run_host(self, inputs: Dict[str, 'numpy.ndarray'],
               iterate: Optional[int] = None) -> Dict[str, 'numpy.ndarray']:
    """numpy in, numpy out: load, run, store (soda::app::<app>() on N GPUs)."""
    self.load(inputs)
    self.run(iterate)
    return self.store(iterate)

  def stats(self) -> Dict[str, float]:
    st = GroupStats()
    check(self._lib.soda_hip_group_last_stats(self._handle, ctypes.byref(st)),
          'group stats')
    return {name: getattr(st, name) for name, _ in GroupStats._fields_}


class Event:
  """hipEvent wrapper; records on the stream the kernels are launched on."""

  def __init__(self):
    self._lib = library()
    self._h = ctypes.c_void_p()
    check(self._lib.soda_hip_event_create(ctypes.byref(self._h)),
          'event_create')

  def record(self, stream: int = 0) -> None:
    check(self._lib.soda_hip_event_record(self._h, ctypes.c_void_p(stream)),
          'event_record')

  def handle(self) -> int:
    """The hipEvent_t inside (what run_device's ghosts_ready / sendable take)."""
    h = ctypes.c_void_p()
    check(self._lib.soda_hip_event_handle(self._h, ctypes.byref(h)),
          'event_handle')
    return h.value

  def elapsed_ms(self, stop: 'Event') -> float:
    ms = ctypes.c_float()
    check(self._lib.soda_hip_event_elapsed_ms(self._h, stop._h,
                                              ctypes.byref(ms)),
          'event_elapsed')
    return ms.value

  def __del__(self):
    try:
      if self._h:
        self._lib.soda_hip_event_destroy(self._h)
    except Exception:
      pass


class Stream:
  """A HIP stream of the caller's own (for a halo exchange that runs beside
  Program.run_device)."""

  def __init__(self, device: int = 0):
    self._lib = library()
    self._h = ctypes.c_void_p()
    check(self._lib.soda_hip_hipstream_create(device, ctypes.byref(self._h)),
          'hipstream_create')
    self.device = device

  @property
  def handle(self) -> int:
    return self._h.value

  def wait_event(self, event: Event) -> None:
    check(self._lib.soda_hip_hipstream_wait_event(self._h, event._h),
          'hipstream_wait_event')

  def copy(self, dst: int, src: int, nbytes: int,
           src_device: Optional[int] = None) -> None:
    check(self._lib.soda_hip_memcpy_d2d(
        ctypes.c_void_p(dst), self.device, ctypes.c_void_p(src),
        self.device if src_device is None else src_device, nbytes, self._h),
          'memcpy_d2d')

  def synchronize(self) -> None:
    check(self._lib.soda_hip_stream_synchronize(self._h), 'stream_synchronize')

  def __del__(self):
    try:
      if self._h:
        self._lib.soda_hip_hipstream_destroy(self._h)
    except Exception:
      pass


def synchronize(stream: int = 0) -> None:
  check(library().soda_hip_stream_synchronize(ctypes.c_void_p(stream)),
        'stream_synchronize')


class PinnedBuffer:
  """Page-aligned host memory of its own (an anonymous mmap), registered with
  the GPU for its lifetime (soda_hip_host_register): dense arrays carved from
  it travel by DMA from / to where they are when handed to Program.run, no
  staging slots, no worker threads.  Meant to live long -- allocate once, run
  many times, close() at the end -- like the buffers the reference host
  allocates once with aligned_alloc(4096, ...) (ref frt/host.py:165-178).

  Why not register any numpy array: heap memory shares pages with other
  objects and is recycled by malloc; registrations that come and go over such
  pages next to the HIP runtime's own pinning of pageable copies ended in GPU
  memory faults in a soak (tools/experiments/r05_host_soak.py), so the library
  only registers ranges that start on a page boundary."""

  def __init__(self, nbytes: int):
    import mmap
    self.nbytes = max(int(nbytes), 1)
    self._map = mmap.mmap(-1, self.nbytes)
    self._addr = ctypes.addressof(ctypes.c_char.from_buffer(self._map))
    check(library().soda_hip_host_register(ctypes.c_void_p(self._addr),
                                           self.nbytes), 'host_register')
    self._open = True

  def array(self, shape, dtype, offset: int = 0):
    """A dense numpy array of `shape` at byte `offset` of the buffer (the
    caller keeps arrays apart; offsets should be multiples of 4096)."""
    import numpy as np
    dtype = np.dtype(dtype)
    count = 1
    for n in shape:
      count *= int(n)
    if offset < 0 or offset + count * dtype.itemsize > self.nbytes:
      raise util.InputError('PinnedBuffer: %d bytes at offset %d do not fit %d'
                            % (count * dtype.itemsize, offset, self.nbytes))
    return np.frombuffer(self._map, dtype, count, offset).reshape(shape)

  def close(self) -> None:
    if not self._open:
      return
    self._open = False
    rc = library().soda_hip_host_unregister(ctypes.c_void_p(self._addr))
    try:
      self._map.close()
    except BufferError:       # arrays still alive: the pages go with them
      pass
    if rc:
      raise util.BackendError('host_unregister: %s' % last_error())

  def __enter__(self):
    return self

  def __exit__(self, *exc):
    self.close()
    return False

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass
