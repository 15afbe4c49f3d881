"""`--hip-host`: the generated C++ host, MI355X edition.

The reference's `--frt-host` prints a C++ translation unit that defines

    int soda::app::<app>(const T* var_<in>_ptr, const int32_t var_<in>_extent[d],
                         const int32_t var_<in>_stride[d],
                         const int32_t var_<in>_min[d],   ... per input, output
                         and param ..., const char* bitstream,
                         int burst_width = ..., int tile_size_0 = ..., ...,
                         int unroll_factor = ...)

(reference src/soda/codegen/frt/host.py:62-88) around the FPGA runtime.  This
module prints the same function -- same name, parameters and return value (0,
frt/host.py:429) -- around libsoda_hip.so: the HIP kernels travel as a string,
the launch plan as a static `soda_hip_plan_t`, and the body is
soda_hip_compile() / soda_hip_program_create() once, then
soda_hip_run_host_box(), which writes only each output's valid box back
(frt/host.py:357-375).  Build: g++ file.cpp -Iinclude -Lsoda_amd -lsoda_hip.
No Python at run time.  The FPGA-only trailing parameters are accepted and
ignored; `bitstream` may be NULL.
"""
from typing import List

from soda_amd import core
from soda_amd.codegen.hip import lower


def emit_source(w, var: str, source: str) -> None:
  """`const char* const <var>[]`: the kernel text, split so that no string
  literal exceeds compiler limits."""
  w('const char* const %s[] = {' % var)
  step = 8000
  for i in range(0, len(source), step):
    w('R"SODA_KERNELS(' + source[i:i + step] + ')SODA_KERNELS",')
  w('nullptr};')


def emit_plan(w, func: str, plan) -> None:
  """`soda_hip_plan_t <func>()` from a ctypes runtime.Plan."""
  w('soda_hip_plan_t %s() {' % func)
  w('  soda_hip_plan_t plan;')
  w('  memset(&plan, 0, sizeof plan);')
  w('  plan.abi_version = SODA_HIP_ABI_VERSION;')
  for field in ('dim', 'num_inputs', 'num_outputs', 'num_locals', 'num_params'):
    w('  plan.%s = %d;' % (field, getattr(plan, field)))
  for i in range(plan.num_params):
    w('  plan.param_elems[%d] = %d;' % (i, plan.param_elems[i]))
  slots = (plan.num_inputs + plan.num_outputs + plan.num_locals +
           plan.num_params)
  for i in range(slots):
    w('  plan.elem_size[%d] = %d;' % (i, plan.elem_size[i]))
  w('  plan.num_kernels = %d;' % plan.num_kernels)
  for i in range(plan.num_kernels):
    d = plan.kernels[i]
    w('  strcpy(plan.kernels[%d].name, "%s");' % (i, d.name.decode()))
    for k in range(3):
      w('  plan.kernels[%d].block[%d] = %d;' % (i, k, d.block[k]))
    for k in range(4):
      w('  plan.kernels[%d].tile[%d] = %d;' % (i, k, max(1, d.tile[k])))
    for field in ('lds_bytes', 'vec', 'march_dim', 'waves_along', 'warm',
                  'window_extra', 'max_elem', 'vgprs', 'pipe', 'chunk_fixed',
                  'max_extent0'):
      w('  plan.kernels[%d].%s = %d;' % (i, field, getattr(d, field)))
    for field in ('step_ns', 'warm_saved', 'bytes_per_cell',
                  'lane_redundancy'):
      w('  plan.kernels[%d].%s = %rf;' % (i, field, float(getattr(d, field))))
  w('  plan.num_passes = %d;' % plan.num_passes)
  for i in range(plan.num_passes):
    p = plan.passes[i]
    w('  plan.passes[%d].fused_iters = %d;' % (i, p.fused_iters))
    w('  plan.passes[%d].num_kernels = %d;' % (i, p.num_kernels))
    w('  plan.passes[%d].cost = %rf;' % (i, float(p.cost)))
    for j in range(p.num_kernels):
      w('  plan.passes[%d].kernel[%d] = %d;' % (i, j, p.kernel[j]))
  for field in ('has_reach', 'reach_lo', 'reach_hi'):
    w('  plan.%s = %d;' % (field, getattr(plan, field)))
  w('  return plan;')
  w('}')


def emit_fail(w, who: str) -> None:
  w('int Fail(const char* what) {')
  w('  char text[512];')
  w('  soda_hip_last_error(text, sizeof text);')
  w('  fprintf(stderr, "%s: %%s: %%s\\n", what, text);' % who)
  w('  return 1;')
  w('}')


def emit_program(w, func: str, source_var: str, plan_func: str,
                 label: str) -> None:
  """`soda_hip_program_t* <func>()`: JIT + load on first use."""
  from soda_amd import runtime
  w('soda_hip_program_t* %s() {' % func)
  w('  static soda_hip_program_t* program = nullptr;')
  w('  if (program) return program;')
  w('  size_t len = 0;')
  w('  for (const char* const* part = %s; *part; ++part) len += strlen(*part);'
    % source_var)
  w('  char* source = new char[len + 1];')
  w('  source[0] = 0;')
  w('  for (const char* const* part = %s; *part; ++part) strcat(source, *part);'
    % source_var)
  w('  const char* options[] = {%s};' % ', '.join(
      '"%s"' % o for o in runtime.COMPILE_OPTIONS))
  w('  void* code = nullptr;')
  w('  size_t code_size = 0;')
  w('  int rc = soda_hip_compile(source, "%s", options, %d, &code, '
    '&code_size);' % (label, len(runtime.COMPILE_OPTIONS)))
  w('  delete[] source;')
  w('  if (rc) { Fail("compiling the kernels"); return nullptr; }')
  w('  const soda_hip_plan_t plan = %s();' % plan_func)
  w('  rc = soda_hip_program_create(code, code_size, &plan, 0, &program);')
  w('  soda_hip_free_code(code);')
  w('  if (rc) { Fail("loading the kernels"); program = nullptr; }')
  w('  return program;')
  w('}')


def emit_group(w, stencil: core.Stencil, gpus: int, source_var: str,
               plan_func: str, label: str) -> None:
  """`soda_hip_group_t* Group(extent)`: JIT + one slab per GPU, remade when
  the extent changes."""
  from soda_amd import runtime
  dim = stencil.dim
  reach_lo, reach_hi = stencil.reach_along(dim - 1)
  w('')
  w('soda_hip_group_t* Group(const int32_t* extent) {')
  w('  static soda_hip_group_t* group = nullptr;')
  w('  static int32_t made_for[%d];' % dim)
  w('  if (group && !memcmp(made_for, extent, sizeof made_for)) return group;')
  w('  if (group) { soda_hip_group_destroy(group); group = nullptr; }')
  w('  size_t len = 0;')
  w('  for (const char* const* part = %s; *part; ++part) len += strlen(*part);'
    % source_var)
  w('  char* source = new char[len + 1];')
  w('  source[0] = 0;')
  w('  for (const char* const* part = %s; *part; ++part) strcat(source, *part);'
    % source_var)
  w('  const char* options[] = {%s};' % ', '.join(
      '"%s"' % o for o in runtime.COMPILE_OPTIONS))
  w('  void* code = nullptr;')
  w('  size_t code_size = 0;')
  w('  int rc = soda_hip_compile(source, "%s", options, %d, &code, '
    '&code_size);' % (label, len(runtime.COMPILE_OPTIONS)))
  w('  delete[] source;')
  w('  if (rc) { Fail("compiling the kernels"); return nullptr; }')
  w('  const soda_hip_plan_t plan = %s();' % plan_func)
  w('  soda_hip_group_desc_t desc;')
  w('  memset(&desc, 0, sizeof desc);')
  w('  desc.num_slabs = %d;' % gpus)
  w('  const char* virt = getenv("SODA_HIP_VIRTUAL_GPUS");')
  w('  for (int s = 0; s < %d; ++s) desc.device[s] = virt && *virt == \'1\' ? 0 '
    ': s;' % gpus)
  w('  for (int d = 0; d < %d; ++d) desc.extent[d] = extent[d];' % dim)
  w('  desc.reach_lo = %d;' % reach_lo)
  w('  desc.reach_hi = %d;' % reach_hi)
  w('  desc.iterate = %d;' % stencil.iterate)
  w('  desc.exchange_every = 0;     // the library picks')
  w('  desc.flags = SODA_HIP_GROUP_CALIBRATE | SODA_HIP_GROUP_THREADS;')
  w('  rc = soda_hip_group_create(code, code_size, &plan, &desc, &group);')
  w('  soda_hip_free_code(code);')
  w('  if (rc) { Fail("loading the kernels"); group = nullptr; return nullptr; }')
  w('  memcpy(made_for, extent, sizeof made_for);')
  w('  return group;')
  w('}')


def print_host(stencil: core.Stencil, opts: lower.LowerOptions,
               extent=None, gpus: int = 1, probe: bool = True) -> str:
  """C++ text of the host for `stencil` lowered with `opts` (chunks tuned for
  `extent` if given; the kernels are correct for any extent).  `gpus` > 1: the
  host cuts the grid into that many slabs, one per GPU, through
  soda_hip_group_* -- still one blocking call in one host thread, like the
  reference's (frt/host.py:319-322).  The environment variable
  SODA_HIP_VIRTUAL_GPUS=1 puts all slabs on device 0 at run time."""
  from soda_amd import runtime
  opts = runtime.resolve_options(stencil, opts, extent, probe=probe)
  mod = lower.lower(stencil, opts)
  # register counts of the kernels as this toolchain compiles them: the
  # library sizes chunk lengths from them at run time
  res = runtime.kernel_resources(
      runtime.compile_source(mod.source, '%s.hip' % stencil.app_name))
  plan = runtime.make_plan(mod, res)
  st = stencil
  dim = st.dim
  app = st.app_name
  table = dict(st.symbol_table)
  table.update((p.name, p.haoda_type) for p in st.param_stmts)
  L: List[str] = []
  w = L.append
  w('// generated by sodac --hip-host: soda::app::%s() on libsoda_hip.so' % app)
  w('// (signature of reference src/soda/codegen/frt/host.py:62-88)')
  w('#include <cstdint>')
  w('#include <cstdio>')
  w('#include <cstdlib>')
  w('#include <cstring>')
  w('#include "soda_hip.h"')
  w('')
  w('namespace {')
  emit_source(w, 'kSodaSource', mod.source)
  w('')
  emit_plan(w, 'MakePlan', plan)
  w('')
  emit_fail(w, 'soda::app::%s' % app)
  w('')
  if gpus > 1:
    emit_group(w, stencil, gpus, 'kSodaSource', 'MakePlan', '%s.hip' % app)
  else:
    emit_program(w, 'Program', 'kSodaSource', 'MakePlan', '%s.hip' % app)
  w('}  // namespace')
  w('')
  # the operator
  params = []
  for stmt in st.input_stmts + st.output_stmts + st.param_stmts:
    const = 'const ' if stmt in st.input_stmts or stmt in st.param_stmts else ''
    ct = table[stmt.name].c_type
    params.append('%s%s* var_%s_ptr' % (const, ct, stmt.name))
    params.append('const int32_t var_%s_extent[%d]' % (stmt.name, dim))
    params.append('const int32_t var_%s_stride[%d]' % (stmt.name, dim))
    params.append('const int32_t var_%s_min[%d]' % (stmt.name, dim))
  params.append('const char* bitstream = nullptr')
  params.append('const int burst_width = %d' % st.burst_width)
  for d in range(dim - 1):
    params.append('const int tile_size_%d = %d' % (d, st.tile_size[d]))
  params.append('const int unroll_factor = %d' % st.unroll_factor)
  w('namespace soda {')
  w('namespace app {')
  w('int %s(%s) {' % (app, ',\n    '.join(params)))
  w('  (void)bitstream; (void)burst_width; (void)unroll_factor;')
  for d in range(dim - 1):
    w('  (void)tile_size_%d;' % d)
  if opts.vec > 1:
    w('  if (var_%s_extent[0] %% %d) {' % (st.input_names[0], opts.vec))
    w('    fprintf(stderr, "soda::app::%s: rows must be a multiple of %d cells '
      '(regenerate with --hip-vec)\\n");' % (app, opts.vec))
    w('    return 1;')
    w('  }')
  if gpus == 1:
    w('  soda_hip_program_t* program = Program();')
    w('  if (!program) return 1;')
  ins = list(st.input_stmts) + list(st.param_stmts)
  w('  const soda_hip_host_tensor_t inputs[%d] = {' % len(ins))
  for stmt in ins:
    if stmt in st.param_stmts:      # only the pointer is read
      w('      {const_cast<%s*>(var_%s_ptr), nullptr, nullptr, nullptr},' %
        (table[stmt.name].c_type, stmt.name))
    else:
      w('      {const_cast<%s*>(var_%s_ptr), var_%s_extent, var_%s_stride, '
        'var_%s_min},' % (table[stmt.name].c_type, stmt.name, stmt.name,
                          stmt.name, stmt.name))
  w('  };')
  w('  const soda_hip_host_tensor_t outputs[%d] = {' % len(st.output_stmts))
  for stmt in st.output_stmts:
    w('      {var_%s_ptr, var_%s_extent, var_%s_stride, var_%s_min},' %
      (stmt.name, stmt.name, stmt.name, stmt.name))
  w('  };')
  # valid boxes (frt/host.py:357-375): compile-time windows, run-time extent
  first = st.input_names[0]
  w('  int32_t valid_lo[%d], valid_hi[%d];' % (len(st.output_names) * dim,
                                              len(st.output_names) * dim))
  for o, name in enumerate(st.output_names):
    if st.preserve_border:
      lo, hi = (0,) * dim, (0,) * dim
    else:
      wlo, whi = st.window_bounds()[name]
      lo = tuple(max(0, -v) for v in wlo)
      hi = tuple(max(0, v) for v in whi)
    for d in range(dim):
      w('  valid_lo[%d] = %d; valid_hi[%d] = var_%s_extent[%d] - %d;' %
        (o * dim + d, lo[d], o * dim + d, first, d, hi[d]))
      w('  if (valid_hi[%d] < valid_lo[%d]) valid_hi[%d] = valid_lo[%d];' %
        (o * dim + d, o * dim + d, o * dim + d, o * dim + d))
  if gpus > 1:
    w('  soda_hip_group_t* group = Group(var_%s_extent);' % first)
    w('  if (!group) return 1;')
    w('  if (soda_hip_group_run_host(group, inputs, outputs, %d, valid_lo, '
      'valid_hi))' % st.iterate)
    w('    return Fail("running");')
  else:
    w('  if (soda_hip_run_host_box(program, inputs, outputs, %d, valid_lo, '
      'valid_hi))' % st.iterate)
    w('    return Fail("running");')
  w('  return 0;')
  w('}')
  w('}  // namespace app')
  w('}  // namespace soda')
  return '\n'.join(L) + '\n'
