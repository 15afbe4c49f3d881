// libsoda_hip.so -- C-ABI shim between the Python host and the JIT-built HIP
// stencil kernels (interface and reference citations: include/soda_hip.h).
//
// Owns what the reference's generated C++ host owns at run time (reference
// src/soda/codegen/frt/host.py:62-431): device buffers, the launch sequence
// over all iterations, copy-in / copy-out of the valid box.  What it does NOT
// do, on purpose: no FPGA tiling/burst layout (frt/host.py:181-249) -- the
// kernels read the caller's dense dim-0-fastest arrays directly; no CPU
// fallback of any kind -- without a GPU every run entry fails with
// SODA_HIP_ERR_NODEVICE.
//
// Build: hipcc -O2 -fPIC -shared soda_hip.cpp -o libsoda_hip.so -lhiprtc
#include "soda_hip.h"

#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {

thread_local std::string g_error;

int fail(int status, const std::string& what) {
  g_error = what;
  return status;
}

int hip_fail(hipError_t e, const char* what) {
  char buf[512];
  snprintf(buf, sizeof buf, "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
  g_error = buf;
  if (e == hipErrorOutOfMemory) return SODA_HIP_ERR_NOMEM;
  if (e == hipErrorNoDevice || e == hipErrorInvalidDevice)
    return SODA_HIP_ERR_NODEVICE;
  return SODA_HIP_ERR_RUNTIME;
}

#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t e_ = (expr);                             \
    if (e_ != hipSuccess) return hip_fail(e_, #expr);   \
  } while (0)

struct DeviceBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
};

int ensure(DeviceBuffer& b, size_t bytes) {
  if (b.bytes >= bytes && b.ptr) return SODA_HIP_OK;
  if (b.ptr) {
    HIP_TRY(hipFree(b.ptr));
    b.ptr = nullptr;
    b.bytes = 0;
  }
  HIP_TRY(hipMalloc(&b.ptr, bytes));
  b.bytes = bytes;
  return SODA_HIP_OK;
}

}  // namespace

struct soda_hip_program {
  soda_hip_plan_t plan;
  int device = 0;
  hipModule_t module = nullptr;
  std::vector<hipFunction_t> functions;
  std::vector<DeviceBuffer> locals;   // one per local tensor
  std::vector<DeviceBuffer> temps;    // one per output: iteration ping-pong
  std::vector<DeviceBuffer> host_in;  // run_host staging on the device
  std::vector<DeviceBuffer> host_prm; // ... of the param arrays
  std::vector<DeviceBuffer> host_out;
  int32_t last_launches = 0;
  int32_t last_fused = 0;
  void* debug = nullptr;              // time-stamp buffer of diagnostic builds
};

struct soda_hip_event {
  hipEvent_t ev;
};

extern "C" {

int soda_hip_abi_version(void) { return SODA_HIP_ABI_VERSION; }

const char* soda_hip_status_string(int status) {
  switch (status) {
    case SODA_HIP_OK: return "ok";
    case SODA_HIP_ERR_INVALID: return "invalid argument";
    case SODA_HIP_ERR_COMPILE: return "kernel compilation failed";
    case SODA_HIP_ERR_RUNTIME: return "HIP runtime error";
    case SODA_HIP_ERR_NOMEM: return "out of memory";
    case SODA_HIP_ERR_NODEVICE: return "no usable GPU";
    case SODA_HIP_ERR_UNSUPPORTED: return "unsupported";
    default: return "unknown status";
  }
}

size_t soda_hip_last_error(char* buf, size_t cap) {
  if (buf && cap) {
    size_t n = g_error.size() < cap - 1 ? g_error.size() : cap - 1;
    memcpy(buf, g_error.data(), n);
    buf[n] = 0;
  }
  return g_error.size();
}

int soda_hip_device_count(int* count) {
  if (!count) return fail(SODA_HIP_ERR_INVALID, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return hip_fail(e, "hipGetDeviceCount");
  }
  *count = n;
  return SODA_HIP_OK;
}

size_t soda_hip_sizeof(int which) {
  switch (which) {
    case 0: return sizeof(soda_hip_kargs_t);
    case 1: return sizeof(soda_hip_kernel_desc_t);
    case 2: return sizeof(soda_hip_pass_desc_t);
    case 3: return sizeof(soda_hip_plan_t);
    case 4: return sizeof(soda_hip_host_tensor_t);
    default: return 0;
  }
}

int soda_hip_compile(const char* source, const char* name,
                     const char* const* options, int32_t num_options,
                     void** code, size_t* code_size) {
  if (!source || !code || !code_size || num_options < 0)
    return fail(SODA_HIP_ERR_INVALID, "soda_hip_compile: NULL argument");
  *code = nullptr;
  *code_size = 0;
  hiprtcProgram prog;
  hiprtcResult r = hiprtcCreateProgram(&prog, source, name ? name : "soda.hip",
                                       0, nullptr, nullptr);
  if (r != HIPRTC_SUCCESS)
    return fail(SODA_HIP_ERR_COMPILE,
                std::string("hiprtcCreateProgram: ") + hiprtcGetErrorString(r));
  r = hiprtcCompileProgram(prog, num_options, const_cast<const char**>(options));
  size_t log_size = 0;
  hiprtcGetProgramLogSize(prog, &log_size);
  std::string log(log_size, '\0');
  if (log_size) hiprtcGetProgramLog(prog, &log[0]);
  if (r != HIPRTC_SUCCESS) {
    hiprtcDestroyProgram(&prog);
    return fail(SODA_HIP_ERR_COMPILE,
                std::string("hiprtcCompileProgram: ") + hiprtcGetErrorString(r) +
                    "\n" + log);
  }
  size_t size = 0;
  r = hiprtcGetCodeSize(prog, &size);
  if (r != HIPRTC_SUCCESS || size == 0) {
    hiprtcDestroyProgram(&prog);
    return fail(SODA_HIP_ERR_COMPILE, "hiprtcGetCodeSize failed");
  }
  void* out = malloc(size);
  if (!out) {
    hiprtcDestroyProgram(&prog);
    return fail(SODA_HIP_ERR_NOMEM, "malloc(code)");
  }
  r = hiprtcGetCode(prog, static_cast<char*>(out));
  hiprtcDestroyProgram(&prog);
  if (r != HIPRTC_SUCCESS) {
    free(out);
    return fail(SODA_HIP_ERR_COMPILE, "hiprtcGetCode failed");
  }
  *code = out;
  *code_size = size;
  g_error = log;  // warnings, if any
  return SODA_HIP_OK;
}

void soda_hip_free_code(void* code) { free(code); }

static int check_plan(const soda_hip_plan_t* p) {
  if (p->abi_version != SODA_HIP_ABI_VERSION)
    return fail(SODA_HIP_ERR_INVALID, "plan: ABI version mismatch");
  if (p->dim < 1 || p->dim > SODA_HIP_MAX_DIM)
    return fail(SODA_HIP_ERR_INVALID, "plan: bad dim");
  if (p->num_inputs < 1 || p->num_outputs < 1 || p->num_locals < 0 ||
      p->num_params < 0 || p->num_params > SODA_HIP_MAX_PARAMS ||
      p->num_inputs + p->num_outputs + p->num_locals + p->num_params >
          SODA_HIP_MAX_TENSORS)
    return fail(SODA_HIP_ERR_INVALID, "plan: bad tensor counts");
  for (int k = 0; k < p->num_params; ++k)
    if (p->param_elems[k] < 1)
      return fail(SODA_HIP_ERR_INVALID, "plan: bad param size");
  if (p->num_kernels < 1 || p->num_kernels > SODA_HIP_MAX_KERNELS)
    return fail(SODA_HIP_ERR_INVALID, "plan: bad kernel count");
  if (p->num_passes < 1 || p->num_passes > SODA_HIP_MAX_PASSES)
    return fail(SODA_HIP_ERR_INVALID, "plan: bad pass count");
  int slots = p->num_inputs + p->num_outputs + p->num_locals + p->num_params;
  for (int s = 0; s < slots; ++s)
    if (p->elem_size[s] < 1 || p->elem_size[s] > 16)
      return fail(SODA_HIP_ERR_INVALID, "plan: bad element size");
  for (int k = 0; k < p->num_kernels; ++k) {
    const soda_hip_kernel_desc_t& d = p->kernels[k];
    if (!memchr(d.name, 0, SODA_HIP_NAME_LEN) || !d.name[0])
      return fail(SODA_HIP_ERR_INVALID, "plan: bad kernel name");
    int64_t threads = 1;
    for (int i = 0; i < 3; ++i) {
      if (d.block[i] < 1) return fail(SODA_HIP_ERR_INVALID, "plan: bad block");
      threads *= d.block[i];
    }
    if (threads > 1024) return fail(SODA_HIP_ERR_INVALID, "plan: block > 1024");
    for (int i = 0; i < p->dim; ++i)
      if (d.tile[i] < 1) return fail(SODA_HIP_ERR_INVALID, "plan: bad tile");
    if (d.lds_bytes < 0 || d.lds_bytes > 160 * 1024)
      return fail(SODA_HIP_ERR_INVALID, "plan: bad lds_bytes");
  }
  int prev = 1 << 30;
  for (int i = 0; i < p->num_passes; ++i) {
    const soda_hip_pass_desc_t& q = p->passes[i];
    if (q.fused_iters < 1 || q.fused_iters >= prev)
      return fail(SODA_HIP_ERR_INVALID,
                  "plan: passes must be sorted by fused_iters, descending");
    prev = q.fused_iters;
    if (q.num_kernels < 1 || q.num_kernels > SODA_HIP_MAX_PASS_KERNELS)
      return fail(SODA_HIP_ERR_INVALID, "plan: bad pass kernel count");
    for (int k = 0; k < q.num_kernels; ++k)
      if (q.kernel[k] < 0 || q.kernel[k] >= p->num_kernels)
        return fail(SODA_HIP_ERR_INVALID, "plan: bad pass kernel index");
  }
  if (p->passes[p->num_passes - 1].fused_iters != 1)
    return fail(SODA_HIP_ERR_INVALID, "plan: last pass must advance 1 iteration");
  return SODA_HIP_OK;
}

int soda_hip_program_create(const void* code, size_t code_size,
                            const soda_hip_plan_t* plan, int32_t device,
                            soda_hip_program_t** program) {
  if (!code || !code_size || !plan || !program)
    return fail(SODA_HIP_ERR_INVALID, "program_create: NULL argument");
  *program = nullptr;
  int rc = check_plan(plan);
  if (rc) return rc;
  int n = 0;
  rc = soda_hip_device_count(&n);
  if (rc) return rc;
  if (n < 1) return fail(SODA_HIP_ERR_NODEVICE, "no GPU visible");
  if (device < 0 || device >= n)
    return fail(SODA_HIP_ERR_INVALID, "program_create: no such device");
  HIP_TRY(hipSetDevice(device));
  soda_hip_program* p = new (std::nothrow) soda_hip_program;
  if (!p) return fail(SODA_HIP_ERR_NOMEM, "new program");
  p->plan = *plan;
  p->device = device;
  hipError_t e = hipModuleLoadData(&p->module, code);
  if (e != hipSuccess) {
    delete p;
    return hip_fail(e, "hipModuleLoadData");
  }
  p->functions.resize(plan->num_kernels);
  for (int k = 0; k < plan->num_kernels; ++k) {
    e = hipModuleGetFunction(&p->functions[k], p->module, plan->kernels[k].name);
    if (e != hipSuccess) {
      std::string what = std::string("hipModuleGetFunction(") +
                         plan->kernels[k].name + ")";
      (void)hipModuleUnload(p->module);
      delete p;
      return hip_fail(e, what.c_str());
    }
  }
  p->locals.resize(plan->num_locals);
  p->temps.resize(plan->num_outputs);
  p->host_in.resize(plan->num_inputs);
  p->host_prm.resize(plan->num_params);
  p->host_out.resize(plan->num_outputs);
  *program = p;
  return SODA_HIP_OK;
}

int soda_hip_program_destroy(soda_hip_program_t* p) {
  if (!p) return SODA_HIP_OK;
  (void)hipSetDevice(p->device);
  for (auto* v : {&p->locals, &p->temps, &p->host_in, &p->host_out,
                  &p->host_prm})
    for (auto& b : *v)
      if (b.ptr) (void)hipFree(b.ptr);
  if (p->module) (void)hipModuleUnload(p->module);
  delete p;
  return SODA_HIP_OK;
}

static int launch(soda_hip_program* p, int k, const soda_hip_kargs_t& base,
                  hipStream_t stream) {
  const soda_hip_kernel_desc_t& d = p->plan.kernels[k];
  soda_hip_kargs_t args = base;
  int64_t blocks = 1;
  for (int i = 0; i < SODA_HIP_MAX_DIM; ++i) {
    int32_t t = i < p->plan.dim ? d.tile[i] : 1;
    args.tile[i] = t;
    args.ntile[i] = (args.extent[i] + t - 1) / t;
    blocks *= args.ntile[i];
  }
  if (blocks < 1 || blocks > 0x7fffffffLL)
    return fail(SODA_HIP_ERR_INVALID, "grid does not fit a 1-D launch");
  size_t size = sizeof args;
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args,
                   HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  HIP_TRY(hipModuleLaunchKernel(p->functions[k], (unsigned)blocks, 1, 1,
                                d.block[0], d.block[1], d.block[2],
                                d.lds_bytes, stream, nullptr, extra));
  return SODA_HIP_OK;
}

static int schedule(const soda_hip_plan_t& plan, int32_t iterate,
                    int32_t* count, int32_t* total) {
  bool costed = false;
  for (int i = 0; i < plan.num_passes; ++i)
    costed = costed || plan.passes[i].cost > 0.f;
  *total = 0;
  if (!costed || iterate > (1 << 20)) {      // greedy, deepest first
    int32_t remaining = iterate;
    for (int i = 0; i < plan.num_passes; ++i) {
      count[i] = remaining / plan.passes[i].fused_iters;
      remaining -= count[i] * plan.passes[i].fused_iters;
      *total += count[i];
    }
    return remaining ? fail(SODA_HIP_ERR_INVALID, "iterate not schedulable")
                     : SODA_HIP_OK;
  }
  // unbounded knapsack over the iteration count: best[n] = least cost of n
  // iterations, pick[n] = the pass used last
  std::vector<double> best(iterate + 1, 1e300);
  std::vector<int8_t> pick(iterate + 1, -1);
  best[0] = 0.0;
  for (int32_t n = 1; n <= iterate; ++n)
    for (int i = 0; i < plan.num_passes; ++i) {
      const int32_t t = plan.passes[i].fused_iters;
      const double c = plan.passes[i].cost > 0.f ? plan.passes[i].cost : 1e6;
      if (t <= n && best[n - t] < 1e299 && best[n - t] + c < best[n]) {
        best[n] = best[n - t] + c;
        pick[n] = (int8_t)i;
      }
    }
  if (pick[iterate] < 0)
    return fail(SODA_HIP_ERR_INVALID, "iterate not schedulable");
  for (int i = 0; i < plan.num_passes; ++i) count[i] = 0;
  for (int32_t n = iterate; n > 0; n -= plan.passes[pick[n]].fused_iters) {
    ++count[pick[n]];
    ++*total;
  }
  return SODA_HIP_OK;
}

int soda_hip_run_device(soda_hip_program_t* p, void* const* outputs,
                        const void* const* inputs, const int32_t* extent,
                        int32_t iterate, void* stream_) {
  return soda_hip_run_device_window(p, outputs, inputs, extent, nullptr,
                                    nullptr, iterate, stream_);
}

int soda_hip_run_device_window(soda_hip_program_t* p, void* const* outputs,
                               const void* const* inputs,
                               const int32_t* extent, const int32_t* origin,
                               const int32_t* global_extent, int32_t iterate,
                               void* stream_) {
  if (!p || !outputs || !inputs || !extent)
    return fail(SODA_HIP_ERR_INVALID, "run_device: NULL argument");
  const soda_hip_plan_t& plan = p->plan;
  if (iterate < 1) return fail(SODA_HIP_ERR_INVALID, "cannot iterate < 1 times");
  if (iterate > 1 && plan.num_inputs != plan.num_outputs)
    return fail(SODA_HIP_ERR_INVALID,
                "number of input tensors must be the same as output if "
                "iterate > 1 times");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  HIP_TRY(hipSetDevice(p->device));

  soda_hip_kargs_t base;
  memset(&base, 0, sizeof base);
  int64_t cells = 1;
  for (int d = 0; d < SODA_HIP_MAX_DIM; ++d) {
    int32_t e = d < plan.dim ? extent[d] : 1;
    if (e < 1) return fail(SODA_HIP_ERR_INVALID, "run_device: extent < 1");
    base.extent[d] = e;
    base.stride[d] = cells;
    cells *= e;
    base.origin[d] = (origin && d < plan.dim) ? origin[d] : 0;
    base.gextent[d] = (global_extent && d < plan.dim) ? global_extent[d] : e;
    if (base.origin[d] < 0 || base.origin[d] + e > base.gextent[d])
      return fail(SODA_HIP_ERR_INVALID,
                  "run_device: the window does not lie in the global grid");
  }
  for (int i = 0; i < plan.num_inputs + plan.num_params; ++i)
    if (!inputs[i]) return fail(SODA_HIP_ERR_INVALID, "run_device: NULL input");
  for (int i = 0; i < plan.num_outputs; ++i)
    if (!outputs[i]) return fail(SODA_HIP_ERR_INVALID, "run_device: NULL output");
  for (int o = 0; o < plan.num_outputs; ++o)
    for (int i = 0; i < plan.num_inputs; ++i)
      if (outputs[o] == inputs[i])
        return fail(SODA_HIP_ERR_INVALID,
                    "run_device: an output aliases an input (in-place runs are "
                    "not supported: inputs are read while outputs are written)");

  // schedule: the multiset of passes that adds up to `iterate` at least total
  // cost (100 iterations with passes of 12 / 8 / 4 / 1: 7 x 12 + 2 x 8 beats
  // 8 x 12 + 4); without costs, as many of the deepest kind as fit, then the
  // next...
  int32_t count[SODA_HIP_MAX_PASSES];
  int32_t total = 0;
  if (int rc = schedule(plan, iterate, count, &total)) return rc;

  if (p->debug) base.buf[SODA_HIP_MAX_TENSORS - 1] = p->debug;
  const int in0 = 0, out0 = plan.num_inputs, loc0 = out0 + plan.num_outputs;
  const int prm0 = loc0 + plan.num_locals;
  for (int k = 0; k < plan.num_params; ++k)   // the same in every iteration
    base.buf[prm0 + k] = const_cast<void*>(inputs[plan.num_inputs + k]);
  for (int l = 0; l < plan.num_locals; ++l) {
    int rc = ensure(p->locals[l], (size_t)cells * plan.elem_size[loc0 + l]);
    if (rc) return rc;
    base.buf[loc0 + l] = p->locals[l].ptr;
  }
  if (total > 1)
    for (int o = 0; o < plan.num_outputs; ++o) {
      int rc = ensure(p->temps[o], (size_t)cells * plan.elem_size[out0 + o]);
      if (rc) return rc;
    }

  p->last_launches = 0;
  p->last_fused = 0;
  std::vector<const void*> src(inputs, inputs + plan.num_inputs);
  int done = 0;
  for (int i = 0; i < plan.num_passes; ++i) {
    for (int c = 0; c < count[i]; ++c, ++done) {
      // the last pass writes the caller's outputs; before that alternate
      // between the program's temporaries and the caller's outputs
      bool to_out = ((total - 1 - done) % 2) == 0;
      for (int j = 0; j < plan.num_inputs; ++j)
        base.buf[in0 + j] = const_cast<void*>(src[j]);
      for (int o = 0; o < plan.num_outputs; ++o)
        base.buf[out0 + o] = to_out ? outputs[o] : p->temps[o].ptr;
      for (int k = 0; k < plan.passes[i].num_kernels; ++k) {
        int rc = launch(p, plan.passes[i].kernel[k], base, stream);
        if (rc) return rc;
        ++p->last_launches;
        if (i == 0) ++p->last_fused;
      }
      if (done + 1 < total)
        for (int j = 0; j < plan.num_inputs; ++j) src[j] = base.buf[out0 + j];
    }
  }
  return SODA_HIP_OK;
}

int soda_hip_program_set_debug_buffer(soda_hip_program_t* p, void* buf) {
  if (!p) return fail(SODA_HIP_ERR_INVALID, "NULL program");
  p->debug = buf;
  return SODA_HIP_OK;
}

int soda_hip_last_launches(soda_hip_program_t* p, int32_t* launches,
                           int32_t* fused_launches) {
  if (!p) return fail(SODA_HIP_ERR_INVALID, "NULL program");
  if (launches) *launches = p->last_launches;
  if (fused_launches) *fused_launches = p->last_fused;
  return SODA_HIP_OK;
}

// -- host-array entry (soda::app::<app> analogue) ----------------------------

static bool is_dense(const soda_hip_host_tensor_t& t, int dim) {
  int64_t s = 1;
  for (int d = 0; d < dim; ++d) {
    if (t.stride[d] != s) return false;
    s *= t.extent[d];
  }
  return true;
}

// copies box [lo, hi) between a strided host array and a dense staging array
static void copy_box(char* strided, const int32_t* stride, char* dense,
                     const int32_t* extent, const int32_t* lo,
                     const int32_t* hi, int dim, int elem, bool to_dense) {
  int32_t idx[SODA_HIP_MAX_DIM];
  int32_t l[SODA_HIP_MAX_DIM], h[SODA_HIP_MAX_DIM];
  int64_t dstride[SODA_HIP_MAX_DIM], s = 1;
  for (int d = 0; d < SODA_HIP_MAX_DIM; ++d) {
    l[d] = d < dim ? lo[d] : 0;
    h[d] = d < dim ? hi[d] : 1;
    dstride[d] = s;
    s *= d < dim ? extent[d] : 1;
    if (h[d] <= l[d]) return;
  }
  for (idx[3] = l[3]; idx[3] < h[3]; ++idx[3])
    for (idx[2] = l[2]; idx[2] < h[2]; ++idx[2])
      for (idx[1] = l[1]; idx[1] < h[1]; ++idx[1]) {
        int64_t so = 0, dof = 0;
        for (int d = 1; d < SODA_HIP_MAX_DIM; ++d) {
          so += d < dim ? (int64_t)idx[d] * stride[d] : 0;
          dof += idx[d] * dstride[d];
        }
        if (stride[0] == 1) {
          char* a = strided + (so + l[0]) * elem;
          char* b = dense + (dof + l[0]) * elem;
          if (to_dense) memcpy(b, a, (size_t)(h[0] - l[0]) * elem);
          else memcpy(a, b, (size_t)(h[0] - l[0]) * elem);
        } else {
          for (int32_t x = l[0]; x < h[0]; ++x) {
            char* a = strided + (so + (int64_t)x * stride[0]) * elem;
            char* b = dense + (dof + x) * elem;
            if (to_dense) memcpy(b, a, elem);
            else memcpy(a, b, elem);
          }
        }
      }
}

int soda_hip_run_host_box(soda_hip_program_t* p,
                          const soda_hip_host_tensor_t* inputs,
                          const soda_hip_host_tensor_t* outputs,
                          int32_t iterate, const int32_t* valid_lo,
                          const int32_t* valid_hi) {
  if (!p || !inputs || !outputs)
    return fail(SODA_HIP_ERR_INVALID, "run_host: NULL argument");
  const soda_hip_plan_t& plan = p->plan;
  const int dim = plan.dim;
  const int32_t* extent = inputs[0].extent;
  if (!extent) return fail(SODA_HIP_ERR_INVALID, "run_host: NULL extent");
  int64_t cells = 1;
  for (int d = 0; d < dim; ++d) {
    if (extent[d] < 1) return fail(SODA_HIP_ERR_INVALID, "run_host: extent < 1");
    cells *= extent[d];
  }
  auto same_extent = [&](const soda_hip_host_tensor_t& t) {
    if (!t.ptr || !t.extent || !t.stride) return false;
    for (int d = 0; d < dim; ++d)
      if (t.extent[d] != extent[d]) return false;
    return true;
  };
  for (int i = 0; i < plan.num_inputs; ++i)
    if (!same_extent(inputs[i]))
      return fail(SODA_HIP_ERR_INVALID, "run_host: bad input tensor");
  for (int o = 0; o < plan.num_outputs; ++o)
    if (!same_extent(outputs[o]))
      return fail(SODA_HIP_ERR_INVALID, "run_host: bad output tensor");
  HIP_TRY(hipSetDevice(p->device));

  int32_t zero[SODA_HIP_MAX_DIM] = {0, 0, 0, 0};
  std::vector<char> staging;
  std::vector<const void*> in_ptrs(plan.num_inputs);
  std::vector<void*> out_ptrs(plan.num_outputs);
  for (int i = 0; i < plan.num_inputs; ++i) {
    size_t bytes = (size_t)cells * plan.elem_size[i];
    int rc = ensure(p->host_in[i], bytes);
    if (rc) return rc;
    const void* host = inputs[i].ptr;
    if (!is_dense(inputs[i], dim)) {
      staging.resize(bytes);
      copy_box(static_cast<char*>(inputs[i].ptr), inputs[i].stride,
               staging.data(), extent, zero, extent, dim, plan.elem_size[i],
               true);
      host = staging.data();
    }
    HIP_TRY(hipMemcpy(p->host_in[i].ptr, host, bytes, hipMemcpyHostToDevice));
    in_ptrs[i] = p->host_in[i].ptr;
  }
  for (int o = 0; o < plan.num_outputs; ++o) {
    size_t bytes = (size_t)cells * plan.elem_size[plan.num_inputs + o];
    int rc = ensure(p->host_out[o], bytes);
    if (rc) return rc;
    out_ptrs[o] = p->host_out[o].ptr;
  }
  const int prm0 = plan.num_inputs + plan.num_outputs + plan.num_locals;
  for (int k = 0; k < plan.num_params; ++k) {
    const soda_hip_host_tensor_t& t = inputs[plan.num_inputs + k];
    if (!t.ptr) return fail(SODA_HIP_ERR_INVALID, "run_host: NULL param");
    size_t bytes = (size_t)plan.param_elems[k] * plan.elem_size[prm0 + k];
    int rc = ensure(p->host_prm[k], bytes);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(p->host_prm[k].ptr, t.ptr, bytes, hipMemcpyHostToDevice));
    in_ptrs.push_back(p->host_prm[k].ptr);
  }
  int rc = soda_hip_run_device(p, out_ptrs.data(), in_ptrs.data(), extent,
                               iterate, nullptr);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(nullptr));
  for (int o = 0; o < plan.num_outputs; ++o) {
    int elem = plan.elem_size[plan.num_inputs + o];
    size_t bytes = (size_t)cells * elem;
    const int32_t* lo = valid_lo ? valid_lo + o * dim : zero;
    const int32_t* hi = valid_hi ? valid_hi + o * dim : extent;
    bool whole = true;
    for (int d = 0; d < dim; ++d)
      whole = whole && lo[d] == 0 && hi[d] == extent[d];
    if (whole && is_dense(outputs[o], dim)) {
      HIP_TRY(hipMemcpy(outputs[o].ptr, out_ptrs[o], bytes,
                        hipMemcpyDeviceToHost));
    } else {
      // only the valid box reaches the caller's array (frt/host.py:357-375)
      staging.resize(bytes);
      HIP_TRY(hipMemcpy(staging.data(), out_ptrs[o], bytes,
                        hipMemcpyDeviceToHost));
      copy_box(static_cast<char*>(outputs[o].ptr), outputs[o].stride,
               staging.data(), extent, lo, hi, dim, elem, false);
    }
  }
  return SODA_HIP_OK;
}

int soda_hip_run_host(soda_hip_program_t* p,
                      const soda_hip_host_tensor_t* inputs,
                      const soda_hip_host_tensor_t* outputs, int32_t iterate) {
  return soda_hip_run_host_box(p, inputs, outputs, iterate, nullptr, nullptr);
}

// -- memory / timing helpers ---------------------------------------------------

int soda_hip_malloc(int32_t device, size_t bytes, void** ptr) {
  if (!ptr) return fail(SODA_HIP_ERR_INVALID, "malloc: NULL ptr");
  *ptr = nullptr;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMalloc(ptr, bytes ? bytes : 1));
  return SODA_HIP_OK;
}

int soda_hip_free(int32_t device, void* ptr) {
  if (!ptr) return SODA_HIP_OK;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipFree(ptr));
  return SODA_HIP_OK;
}

int soda_hip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice,
                         static_cast<hipStream_t>(stream)));
  HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return SODA_HIP_OK;
}

int soda_hip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost,
                         static_cast<hipStream_t>(stream)));
  HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return SODA_HIP_OK;
}

int soda_hip_memset(void* dst, int value, size_t bytes, void* stream) {
  HIP_TRY(hipMemsetAsync(dst, value, bytes, static_cast<hipStream_t>(stream)));
  return SODA_HIP_OK;
}

int soda_hip_stream_synchronize(void* stream) {
  HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return SODA_HIP_OK;
}

int soda_hip_event_create(soda_hip_event_t** event) {
  if (!event) return fail(SODA_HIP_ERR_INVALID, "event_create: NULL");
  soda_hip_event* e = new (std::nothrow) soda_hip_event;
  if (!e) return fail(SODA_HIP_ERR_NOMEM, "new event");
  hipError_t err = hipEventCreate(&e->ev);
  if (err != hipSuccess) {
    delete e;
    return hip_fail(err, "hipEventCreate");
  }
  *event = e;
  return SODA_HIP_OK;
}

int soda_hip_event_record(soda_hip_event_t* event, void* stream) {
  if (!event) return fail(SODA_HIP_ERR_INVALID, "event_record: NULL");
  HIP_TRY(hipEventRecord(event->ev, static_cast<hipStream_t>(stream)));
  return SODA_HIP_OK;
}

int soda_hip_event_elapsed_ms(soda_hip_event_t* start, soda_hip_event_t* stop,
                              float* ms) {
  if (!start || !stop || !ms)
    return fail(SODA_HIP_ERR_INVALID, "event_elapsed: NULL");
  HIP_TRY(hipEventSynchronize(stop->ev));
  HIP_TRY(hipEventElapsedTime(ms, start->ev, stop->ev));
  return SODA_HIP_OK;
}

int soda_hip_event_destroy(soda_hip_event_t* event) {
  if (!event) return SODA_HIP_OK;
  (void)hipEventDestroy(event->ev);
  delete event;
  return SODA_HIP_OK;
}

}  // extern "C"
