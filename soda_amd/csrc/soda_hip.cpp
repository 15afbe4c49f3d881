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
// Build: hipcc -O2 -fPIC -shared soda_hip.cpp soda_group.cpp -o libsoda_hip.so
//        -lhiprtc   (__graft_entry__.build_library)
#include "soda_internal.h"

#include <hip/hiprtc.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

namespace soda_detail {

thread_local std::string g_error;

int fail(int status, const std::string& what) {
  g_error = what;
  return status;
}

const std::string& last_error_text() { return g_error; }

int hip_fail(hipError_t e, const char* what) {
  char buf[512];
  snprintf(buf, sizeof buf, "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
  g_error = buf;
  if (e == hipErrorOutOfMemory) return SODA_HIP_ERR_NOMEM;
  if (e == hipErrorNoDevice || e == hipErrorInvalidDevice)
    return SODA_HIP_ERR_NODEVICE;
  return SODA_HIP_ERR_RUNTIME;
}

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

}  // namespace soda_detail

using namespace soda_detail;

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
    case 5: return sizeof(soda_hip_stream_desc_t);
    case 6: return sizeof(soda_hip_slab_run_t);
    case 7: return sizeof(soda_hip_group_desc_t);
    case 8: return sizeof(soda_hip_slab_info_t);
    case 9: return sizeof(soda_hip_group_stats_t);
    case 10: return sizeof(soda_hip_launch_info_t);
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

int soda_hip_compiler_version(int32_t* major, int32_t* minor) {
  if (!major || !minor) return fail(SODA_HIP_ERR_INVALID, "compiler_version: NULL");
  int a = 0, b = 0;
  if (hiprtcVersion(&a, &b) != HIPRTC_SUCCESS)
    return fail(SODA_HIP_ERR_COMPILE, "hiprtcVersion failed");
  *major = a;
  *minor = b;
  return SODA_HIP_OK;
}

static int check_plan(const soda_hip_plan_t* p);
static int plan_geometry_c(const soda_hip_plan_t* plan, const int32_t* extent,
                           int32_t* tiles, float* pass_ns);
static int plan_schedule_c(const soda_hip_plan_t* plan, const int32_t* extent,
                           int32_t iterate, int32_t* count);

int soda_hip_plan_geometry(const soda_hip_plan_t* plan, const int32_t* extent,
                           int32_t* tiles, float* pass_ns) {
  return plan_geometry_c(plan, extent, tiles, pass_ns);
}

int soda_hip_plan_schedule(const soda_hip_plan_t* plan, const int32_t* extent,
                           int32_t iterate, int32_t* count) {
  return plan_schedule_c(plan, extent, iterate, count);
}

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
    if (d.vec < 0 || d.march_dim < 0 || d.march_dim > p->dim ||
        d.waves_along < 0 || d.warm < 0 || d.vgprs < 0 || d.pipe < 0 ||
        d.step_ns < 0 || d.bytes_per_cell < 0 || d.warm_saved < 0)
      return fail(SODA_HIP_ERR_INVALID, "plan: bad launch-geometry field");
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
  if (p->has_reach && (p->reach_lo < 0 || p->reach_hi < 0))
    return fail(SODA_HIP_ERR_INVALID, "plan: negative reach");
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
  p->auto_calibrate = !getenv("SODA_HIP_NO_CALIBRATE");
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
  for (int i = 0; i < 2; ++i) {
    if (p->ev_pre[i]) (void)hipEventDestroy(p->ev_pre[i]);
    if (p->ev_bnd[i]) (void)hipEventDestroy(p->ev_bnd[i]);
  }
  if (p->side) (void)hipStreamDestroy(p->side);
  for (auto& st : p->hstream)
    if (st) (void)hipStreamDestroy(st);
  p->ring_in.release();
  p->ring_out.release();
  for (auto& b : p->band_out)
    if (b.ptr) (void)hipFree(b.ptr);
  for (auto& e : p->hevents) (void)hipEventDestroy(e);
  if (p->module) (void)hipModuleUnload(p->module);
  delete p;
  return SODA_HIP_OK;
}

// ---- launch geometry --------------------------------------------------------
// MI355X: 256 CUs x 4 SIMDs; 512 VGPRs per lane per SIMD in granules of 8
// (MI355X_MICROARCH.md, Register files); sustained HBM rate of a streaming
// kernel of this family ~6.3 TB/s (profiles/).
namespace {

const int kSimds = 1024;
// Constants of the launch-time model, fitted (tools/fit_model.py, least squares
// on the log ratio) to 128 measured pass times -- jacobi2d T = 1..12 on eight
// extents from 8192 x 600 to 8192^2, with DPP and with mixed DPP / swizzle
// lane shifts, heat3d T = 1, 2 on four (profiles/r03_model_data.jsonl,
// r03_model_data2.jsonl): rms error 9 % (15-17 % with round 2's hand fits), and
// the schedules the model picks for 100 iterations cost at most 6 % more than
// the ones picked by the clock on every one of those extents (the hand fits:
// +19-20 % on the 1120- and 1224-row slabs of an 8-GPU run).
// tests/test_hip_parity.py::test_model_schedule_is_close_to_the_calibrated_one
// re-checks that on the GPU.
const double kHbmBytesPerNs = 6369.0;
const double kLaunchNs = 2547.0;
const double kWaveNs = 391.0;          // per wave per SIMD of the grid
const double kNormP = 2.708;           // time = (issue^p + memory^p)^(1/p)
const int64_t kBufWindowMax = 1ll << 30;      // SODA_BUF_WINDOW_MAX, soda_rt.h

int waves_per_simd(int vgprs) {
  int alloc = vgprs < 8 ? 8 : (vgprs + 7) / 8 * 8;
  int w = 512 / alloc;
  return w < 1 ? 1 : w > 8 ? 8 : w;
}

// issue time a wave costs its SIMD, relative to one of >= 3 resident waves
// (the chunk rule's own weights: they rank chunk lengths of ONE kernel, were
// tuned by kernel sweeps, and stay as they are)
double issue_share(int64_t k) { return k == 1 ? 2.0 : k == 2 ? 1.2 : 1.0; }
// the same for the time model, fitted together with the constants above
double time_share(int64_t k) {
  return k == 1 ? 1.312 : k == 2 ? 1.475 : k == 3 ? 1.365 : 1.046;
}

// Length (cells along the marched dimension) one wave should own.  The waves
// of a launch are dealt evenly over the SIMDs, a SIMD's waves share its issue
// slots, and every wave pays `warm` pipeline warm-up steps on top of its
// chunk, so  time ~ k * eff(k) * (chunk + warm),  k = waves per SIMD.  One
// wave alone on a SIMD issues at half rate, two still leave bubbles; a kernel
// that fuses few iterations is latency-bound and wants >= 4 waves per SIMD.
// A register-limited grid that fills the last 2-3 % of its wave slots runs
// slower than one a row longer that leaves them free (measured).
int32_t tuned_chunk(const soda_hip_kernel_desc_t& d, const int32_t* tile,
                    const int32_t* extent, int dim, int64_t others) {
  const int axis = d.march_dim - 1;
  const int32_t n = extent[axis];
  const int wpb = (d.block[0] * d.block[1] * d.block[2] + 63) / 64;
  const int along = d.waves_along > 0 ? d.waves_along : 1;
  // Peeled warm-up steps compute little but still LOAD their rows.  Where
  // peeling saves (nearly) ALL of a long warm-up -- a one-iteration kernel with
  // a tall window, whose stages all start at the end of it: xcorr, 18 of 18
  // steps -- "warm-up costs nothing" would shorten the chunks to 8 rows, each
  // row read 3.25 times (measured: 123 us against 59 at 64 rows, 72 at 128,
  // profiles/r03_sweep_xcorr_slide_chunk.json).  Such a kernel is bound by its
  // loads: its warm-up counts in full, and like the other load-bound kernels
  // it gets many short waves.
  const bool load_only_warm = d.warm > 12 && d.warm_saved >= 0.75 * d.warm;
  const double warm = load_only_warm ? d.warm : d.warm - d.warm_saved;
  (void)tile; (void)dim;
  const int cap = waves_per_simd(d.vgprs);
  if (d.pipe > 1) {
    // stage-pipelined blocks: the warm-up is paid once per block; ~3.5x the
    // warm-up is best on every grid -- but never a grid that needs a second,
    // nearly empty round of blocks (a block holds one wave slot on `pipe`
    // SIMDs; measured: 2592 blocks on 1792 slots cost 15 %)
    int32_t target = (int32_t)(3.5 * d.warm) > 64 ? (int32_t)(3.5 * d.warm) : 64;
    int64_t chunks = (n + target - 1) / target;
    if (chunks < 1) chunks = 1;
    const int64_t slots = (int64_t)kSimds * cap / d.pipe;
    int64_t blocks = others * chunks;
    if (blocks > slots * 0.975 && blocks < slots * 1.6) {
      chunks = (int64_t)(slots * 0.975) / others;
      if (chunks < 1) chunks = 1;
    }
    return (int32_t)((n + chunks - 1) / chunks);
  }
  const int k_min = d.warm <= 12 ? 4 : 2;
  double best_cost = 0;
  int32_t best = 0;
  // Candidates: the cost below grows with the chunk length while the number of
  // chunks stays the same, so only the shortest chunk of every chunk COUNT can
  // win -- O(sqrt n) candidates instead of n (a 2M-row stream took 59 ms of
  // host time per run this way, a launch must not)
  for (int32_t chunk = n < 8 ? n : 8; chunk <= n;) {
    int64_t chunks = (n + chunk - 1) / chunk;
    int64_t blocks = others * ((chunks + along - 1) / along);
    int64_t waves = blocks * wpb;
    double cost;
    const int64_t slots = (int64_t)cap * kSimds;
    if (waves <= slots) {
      int64_t k = (waves + kSimds - 1) / kSimds;
      // (one wave per SIMD has nothing to pack: a full grid is fine there --
      // heat3d T = 2 at 260 VGPRs, 1024 waves of 128 planes: 195 us, two
      // rounds of 64-plane waves 205)
      if (cap > 1) {
        if (k >= cap && waves > 0.975 * k * kSimds) ++k;
        if (k < k_min) k = k_min;
      }
      cost = k * issue_share(k) * (chunk + warm);
    } else {
      // more waves than fit at once: whole rounds of `cap` waves per SIMD,
      // then a ragged one whose few waves issue at a lone wave's rate
      // (heat3d T = 2, 2-wave blocks, 245 VGPRs: 19 chunks = 2.4 rounds took
      // 243 us, 16 chunks = 2 rounds 218 us)
      const int64_t full = waves / slots, rem = waves - full * slots;
      cost = full * cap * issue_share(cap) * (chunk + warm);
      if (rem > 0) {
        const int64_t rk = (rem + kSimds - 1) / kSimds;
        cost += rk * issue_share(rk) * (chunk + warm);
      }
    }
    if (!best || cost < best_cost || (cost == best_cost && chunk > best)) {
      best_cost = cost;
      best = chunk;
    }
    if (chunks <= 1) break;
    // the shortest chunk that makes one chunk fewer
    const int64_t next = (n + chunks - 2) / (chunks - 1);
    chunk = next > chunk ? (int32_t)next : chunk + 1;
  }
  // latency-bound: more, shorter waves (not for a kernel the registers allow
  // one wave per SIMD of: it has no second wave to hide anything behind)
  if ((d.warm <= 12 || load_only_warm) && best > 64 && cap > 1) best = 64;
  int64_t chunks = (n + best - 1) / best;
  return (int32_t)((n + chunks - 1) / chunks);   // same count, equal lengths
}

int kernel_geometry(const soda_hip_kernel_desc_t& d, const int32_t* extent,
                    int dim, Geometry* g) {
  char buf[384];
  int64_t cells = 1;
  for (int i = 0; i < SODA_HIP_MAX_DIM; ++i) {
    g->tile[i] = i < dim ? d.tile[i] : 1;
    if (i < dim) cells *= extent[i];
  }
  g->ns = 0;
  if (d.vec > 1 && extent[0] % d.vec) {
    snprintf(buf, sizeof buf,
             "%s was built for rows that are a multiple of %d cells; extent[0] "
             "= %d is not (rebuild the program for this extent)", d.name, d.vec,
             extent[0]);
    return fail(SODA_HIP_ERR_INVALID, buf);
  }
  if (d.max_extent0 > 0 && extent[0] > d.max_extent0) {
    snprintf(buf, sizeof buf,
             "%s was built for rows of at most %d cells (one block covers the "
             "row); extent[0] = %d (rebuild the program for this extent)",
             d.name, d.max_extent0, extent[0]);
    return fail(SODA_HIP_ERR_INVALID, buf);
  }
  double rows_factor = 1.0;
  double valu_ns = 0, wave_ns = 0;
  if (d.march_dim > 0) {
    if (d.march_dim > dim)
      return fail(SODA_HIP_ERR_INVALID, "plan: bad march_dim");
    const int axis = d.march_dim - 1;
    const int along = d.waves_along > 0 ? d.waves_along : 1;
    const int wpb = (d.block[0] * d.block[1] * d.block[2] + 63) / 64;
    int64_t others = 1;
    for (int i = 0; i < dim; ++i)
      if (i != axis) others *= (extent[i] + g->tile[i] - 1) / g->tile[i];
    int32_t per_wave = g->tile[axis] / along;
    if (per_wave < 1) per_wave = 1;
    if (!d.chunk_fixed && d.vgprs > 0)
      per_wave = tuned_chunk(d, g->tile, extent, dim, others);
    if (d.window_extra >= 0) {
      int64_t plane = d.max_elem > 0 ? d.max_elem : 8;
      for (int i = 0; i < axis; ++i) plane *= extent[i];
      int64_t limit = kBufWindowMax / plane - d.window_extra;
      if (limit < 1) {
        snprintf(buf, sizeof buf,
                 "%s: one plane of this extent (%lld bytes) exceeds the 1 GiB "
                 "buffer window of the marching kernels; use the `direct` "
                 "strategy", d.name, (long long)plane);
        return fail(SODA_HIP_ERR_INVALID, buf);
      }
      if (per_wave > limit) {
        if (d.chunk_fixed) {
          snprintf(buf, sizeof buf,
                   "%s: chunks of %d planes exceed the 1 GiB buffer window on "
                   "this extent (at most %lld)", d.name, per_wave,
                   (long long)limit);
          return fail(SODA_HIP_ERR_INVALID, buf);
        }
        per_wave = (int32_t)limit;
      }
    }
    g->tile[axis] = per_wave * along;
    const int32_t n = extent[axis];
    const int64_t chunks = (n + per_wave - 1) / per_wave;
    const int64_t waves = others * ((chunks + along - 1) / along) * wpb;
    const int pipe = d.pipe > 0 ? d.pipe : 1;
    double steps = per_wave + d.warm - d.warm_saved;
    if (steps < 1) steps = 1;
    // SIMD issue time in units of one wave's row step: whole rounds of as
    // many waves as the registers allow, then the ragged rest (tuned_chunk)
    const int64_t slots = (int64_t)(d.vgprs > 0 ? waves_per_simd(d.vgprs) : 8) *
                          kSimds;
    const int64_t full = waves / slots, rem = waves - full * slots;
    double share = full * (double)(slots / kSimds) * time_share(slots / kSimds);
    if (rem > 0 || full == 0) {
      int64_t rk = (rem + kSimds - 1) / kSimds;
      if (rk < 1) rk = 1;
      share += rk * time_share(rk);
    }
    valu_ns = share * steps * d.step_ns / pipe;
    rows_factor = (per_wave + (double)d.warm) / per_wave;
    wave_ns = kWaveNs * (double)waves / kSimds;
  }
  if (d.step_ns > 0 || d.bytes_per_cell > 0) {
    const double lanes = d.lane_redundancy > 1 ? d.lane_redundancy : 1.0;
    // reads are re-fetched for halo lanes and warm-up rows, writes are not
    const double bytes = cells * (double)d.bytes_per_cell *
                         (0.5 * lanes * rows_factor + 0.5);
    const double mem_ns = bytes / kHbmBytesPerNs;
    g->ns = pow(pow(valu_ns, kNormP) + pow(mem_ns, kNormP), 1.0 / kNormP) +
            kLaunchNs + wave_ns;
  }
  return SODA_HIP_OK;
}

// tiles of every kernel and time of every pass for `extent`
int plan_geometry(const soda_hip_plan_t& plan, const int32_t* extent,
                  std::vector<Geometry>* geo, std::vector<double>* pass_ns) {
  geo->resize(plan.num_kernels);
  for (int k = 0; k < plan.num_kernels; ++k)
    if (int rc = kernel_geometry(plan.kernels[k], extent, plan.dim, &(*geo)[k]))
      return rc;
  pass_ns->assign(plan.num_passes, 0.0);
  for (int i = 0; i < plan.num_passes; ++i) {
    double t = 0;
    bool modelled = true;
    for (int j = 0; j < plan.passes[i].num_kernels; ++j) {
      const Geometry& g = (*geo)[plan.passes[i].kernel[j]];
      if (g.ns <= 0) modelled = false;
      t += g.ns;
    }
    (*pass_ns)[i] = modelled ? t : 0.0;
  }
  return SODA_HIP_OK;
}

}  // namespace

}  // extern "C"

namespace {

// A launch may cover a marching kernel's chunks with a run of them left out
// (kargs skip_from / skip_count): `count` block tiles along dimension `ax`.
struct TileRange {
  int ax;
  int32_t skip_from, skip_count, count;
};

int launch(soda_hip_program* p, int k, const soda_hip_kargs_t& base,
           const int32_t* tile, hipStream_t stream,
           const TileRange* range = nullptr) {
  const soda_hip_kernel_desc_t& d = p->plan.kernels[k];
  soda_hip_kargs_t args = base;
  int64_t blocks = 1;
  for (int i = 0; i < SODA_HIP_MAX_DIM; ++i) {
    int32_t t = i < p->plan.dim ? tile[i] : 1;
    args.tile[i] = t;
    args.ntile[i] = (args.extent[i] + t - 1) / t;
    if (range && range->ax == i) args.ntile[i] = range->count;
    blocks *= args.ntile[i];
  }
  args.skip_from = range ? range->skip_from : 0;
  args.skip_count = range ? range->skip_count : 0;
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

int64_t ceil_div(int64_t a, int64_t b) { return a <= 0 ? 0 : (a + b - 1) / b; }
int64_t floor_div(int64_t a, int64_t b) { return a <= 0 ? 0 : a / b; }

}  // namespace

namespace soda_detail {

int schedule(const soda_hip_plan_t& plan, const std::vector<double>& pass_ns,
             int32_t iterate, int32_t* count, int32_t* total) {
  // modelled times for this extent where every pass has one, else the plan's
  // static relative costs, else greedy
  std::vector<double> cost(plan.num_passes, 0.0);
  bool modelled = true;
  for (int i = 0; i < plan.num_passes; ++i)
    modelled = modelled && pass_ns[i] > 0;
  bool costed = modelled;
  for (int i = 0; i < plan.num_passes; ++i) {
    cost[i] = modelled ? pass_ns[i] : plan.passes[i].cost;
    costed = costed || cost[i] > 0;
  }
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
      const double c = cost[i] > 0 ? cost[i] : 1e12;
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

int extent_plan(const soda_hip_plan_t& plan,
                std::map<ExtentKey, ExtentPlan>* cache, const int32_t* ext,
                const ExtentPlan** out) {
  ExtentKey key;
  for (int d = 0; d < SODA_HIP_MAX_DIM; ++d) key[d] = ext[d];
  auto it = cache->find(key);
  if (it == cache->end()) {
    ExtentPlan ep;
    if (int rc = plan_geometry(plan, ext, &ep.geo, &ep.model_ns)) return rc;
    if (cache->size() > 4096) cache->clear();
    it = cache->emplace(key, std::move(ep)).first;
  }
  *out = &it->second;
  return SODA_HIP_OK;
}

}  // namespace soda_detail

static int plan_geometry_c(const soda_hip_plan_t* plan, const int32_t* extent,
                           int32_t* tiles, float* pass_ns) {
  if (!plan || !extent) return fail(SODA_HIP_ERR_INVALID, "geometry: NULL argument");
  if (int rc = check_plan(plan)) return rc;
  int32_t ext[SODA_HIP_MAX_DIM];
  for (int d = 0; d < SODA_HIP_MAX_DIM; ++d) {
    ext[d] = d < plan->dim ? extent[d] : 1;
    if (ext[d] < 1) return fail(SODA_HIP_ERR_INVALID, "geometry: extent < 1");
  }
  std::vector<Geometry> geo;
  std::vector<double> ns;
  if (int rc = plan_geometry(*plan, ext, &geo, &ns)) return rc;
  if (tiles)
    for (int k = 0; k < plan->num_kernels; ++k)
      for (int d = 0; d < SODA_HIP_MAX_DIM; ++d)
        tiles[k * SODA_HIP_MAX_DIM + d] = geo[k].tile[d];
  if (pass_ns)
    for (int i = 0; i < plan->num_passes; ++i) pass_ns[i] = (float)ns[i];
  return SODA_HIP_OK;
}

static int plan_schedule_c(const soda_hip_plan_t* plan, const int32_t* extent,
                           int32_t iterate, int32_t* count) {
  if (!plan || !extent || !count)
    return fail(SODA_HIP_ERR_INVALID, "schedule: NULL argument");
  if (iterate < 1) return fail(SODA_HIP_ERR_INVALID, "cannot iterate < 1 times");
  if (int rc = check_plan(plan)) return rc;
  int32_t ext[SODA_HIP_MAX_DIM];
  for (int d = 0; d < SODA_HIP_MAX_DIM; ++d) {
    ext[d] = d < plan->dim ? extent[d] : 1;
    if (ext[d] < 1) return fail(SODA_HIP_ERR_INVALID, "schedule: extent < 1");
  }
  std::vector<Geometry> geo;
  std::vector<double> ns;
  if (int rc = plan_geometry(*plan, ext, &geo, &ns)) return rc;
  int32_t total = 0;
  return schedule(*plan, ns, iterate, count, &total);
}

namespace soda_detail {

// The launches of a run in order: which rows every pass covers (all of them,
// or the cone that can still reach the kept range) and, next to a halo
// exchange, how the first and the last pass are cut in two.
//
// A pass is launched in two parts when it is ONE marching kernel streaming
// along the last dimension: the chunks whose inputs include ghost rows still
// in flight (first pass of the run) or whose rows the neighbours fetch next
// (last pass) form the boundary [0, bnd_lo) U [bnd_hi, chunks), the rest the
// interior.  Any other pass waits, runs whole, and signals.
int plan_launches(const soda_hip_plan_t& plan,
                  std::map<ExtentKey, ExtentPlan>* cache, const int32_t* ext,
                  const int32_t* count, int32_t total, int32_t iterate,
                  const SlabRun* slab, std::vector<PassLaunch>* out) {
  const int ax = plan.dim - 1;
  const int32_t rows = ext[ax];
  const Cone* cone = slab ? &slab->cone : nullptr;
  out->clear();
  out->reserve(total);
  int done = 0, done_iters = 0;
  for (int i = 0; i < plan.num_passes; ++i) {
    for (int c = 0; c < count[i]; ++c, ++done) {
      PassLaunch L;
      memset(&L, 0, sizeof L);
      L.pass = i;
      // Rows this pass has to touch.  With a cone: the rows it must DELIVER
      // are those the iterations still to come can carry into the kept range;
      // it is launched on these plus the rows its own iterations read beyond
      // them (their results are written too, and are wrong where the launch
      // saw zeros instead of neighbours -- nothing reads them again).
      int32_t lo = 0, hi = rows;
      const int32_t fused = plan.passes[i].fused_iters;
      done_iters += fused;
      if (cone) {
        const int64_t after = iterate - done_iters;        // iterations to come
        const int64_t a = cone->keep_lo - (after + fused) * (int64_t)cone->reach_lo;
        const int64_t b = cone->keep_hi + (after + fused) * (int64_t)cone->reach_hi;
        if (cone->keep_lo > 0 && a > 0) lo = (int32_t)a;
        if (cone->keep_hi < rows && b < rows) hi = (int32_t)b;
      }
      L.lo = lo;
      L.hi = hi;
      const bool first = done == 0, last = done + 1 == total;
      L.wait = first && slab && slab->ghosts_ready;
      L.record = last && slab && slab->sendable;
      const int k0 = plan.passes[i].kernel[0];
      if ((L.wait || L.record) && plan.passes[i].num_kernels == 1 &&
          plan.kernels[k0].march_dim == plan.dim) {
        int32_t sub[SODA_HIP_MAX_DIM];
        for (int d = 0; d < SODA_HIP_MAX_DIM; ++d) sub[d] = ext[d];
        sub[ax] = hi - lo;
        const ExtentPlan* use = nullptr;
        if (int rc = extent_plan(plan, cache, sub, &use)) return rc;
        const int64_t lb = use->geo[k0].tile[ax];
        const int64_t nchunk = (hi - lo + lb - 1) / lb;
        int64_t bnd_lo = 0, bnd_hi = nchunk;
        if (L.wait) {
          // chunk c reads rows [c lb - fused reach_lo, (c + 1) lb + fused reach_hi)
          if (slab->ghost_lo > 0)
            bnd_lo = ceil_div(slab->ghost_lo - lo + (int64_t)fused * cone->reach_lo, lb);
          if (slab->ghost_hi > 0)
            bnd_hi = floor_div(rows - slab->ghost_hi - lo -
                               (int64_t)fused * cone->reach_hi, lb);
        }
        if (L.record) {
          // `sendable` promises two things to the exchange that is ordered
          // behind it: the rows the neighbours fetch are complete, and NOTHING
          // launched later in this run writes the result's ghost rows (the
          // exchange overwrites them next).  So the boundary holds the chunks
          // that deliver send rows ...
          if (slab->send_lo > 0) {
            const int64_t v = ceil_div(cone->keep_lo + slab->send_lo - lo, lb);
            if (v > bnd_lo) bnd_lo = v;
          }
          if (slab->send_hi > 0) {
            const int64_t v = floor_div(cone->keep_hi - slab->send_hi - lo, lb);
            if (v < bnd_hi) bnd_hi = v;
          }
          // ... AND every chunk that writes a row outside the kept range, on
          // either side, whether or not that side sends anything.  With a
          // symmetric reach the send rule above already covers them (a side
          // with ghosts is a side that sends); with a one-sided reach (upwind
          // taps: ghosts above, sends below) it does not, and when the run
          // also has nothing to wait for (fresh ghosts: the wait rule is off)
          // the top chunk -- which writes the upper ghost rows -- used to
          // count as interior and could overwrite the rows the next exchange
          // had just received (tools/flake_loop.py, round 4: 2 of 400 skewed
          // trials of the upwind case).
          if (cone->keep_lo > lo) {
            const int64_t v = ceil_div(cone->keep_lo - lo, lb);
            if (v > bnd_lo) bnd_lo = v;
          }
          if (cone->keep_hi < hi) {
            const int64_t v = floor_div(cone->keep_hi - lo, lb);
            if (v < bnd_hi) bnd_hi = v;
          }
        }
        if (bnd_lo > nchunk) bnd_lo = nchunk;
        if (bnd_hi > nchunk) bnd_hi = nchunk;
        L.chunk = (int32_t)lb;
        L.chunks = (int32_t)nchunk;
        L.bnd_lo = (int32_t)bnd_lo;
        L.bnd_hi = (int32_t)bnd_hi;
        L.split = bnd_lo < bnd_hi && (bnd_lo > 0 || bnd_hi < nchunk);
      }
      out->push_back(L);
    }
  }
  return SODA_HIP_OK;
}

// The two parts of a split pass go out on the caller's stream one after the
// other (the part that does not depend on the exchange first).  Measured on
// the middle slab of an 8-GPU run alone on an MI355X
// (profiles/r03_slab_overlap.jsonl): jacobi2d 8192 x 1224, 100 iterations,
// two split passes per step -- 0.333 ms unsplit, 0.367 ms this way, 0.384-0.390
// ms with the boundary part on a stream of its own (SODA_HIP_SPLIT=side: the
// parts then share the GPU, but two hand-overs between hardware queues cost
// more than that buys).
static bool split_in_order() {
  static const int mode = [] {
    const char* v = getenv("SODA_HIP_SPLIT");
    return v && !strcmp(v, "side") ? 0 : 1;
  }();
  return mode == 1;
}

// the stream and events of split passes, made on first use
static int side_stream(soda_hip_program* p) {
  if (p->side) return SODA_HIP_OK;
  HIP_TRY(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
  for (int i = 0; i < 2; ++i) {
    HIP_TRY(hipEventCreateWithFlags(&p->ev_pre[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&p->ev_bnd[i], hipEventDisableTiming));
  }
  return SODA_HIP_OK;
}

int run_core(soda_hip_program_t* p, void* const* outputs,
             const void* const* inputs, const int32_t* extent,
             const int32_t* origin, const int32_t* global_extent,
             int32_t iterate, void* stream_, int force_pass,
             const SlabRun* slab) {
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
  const Cone* cone = slab ? &slab->cone : nullptr;

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
  // next...  Remembered per (extent, iterate).
  const ExtentPlan* ep = nullptr;
  if (int rc = extent_plan(plan, &p->extents, base.extent, &ep)) return rc;
  int32_t count[SODA_HIP_MAX_PASSES];
  int32_t total = 0;
  if (force_pass >= 0) {
    if (force_pass >= plan.num_passes ||
        iterate % plan.passes[force_pass].fused_iters)
      return fail(SODA_HIP_ERR_INVALID, "run_core: bad forced pass");
    for (int i = 0; i < plan.num_passes; ++i) count[i] = 0;
    count[force_pass] = total = iterate / plan.passes[force_pass].fused_iters;
  } else {
    ExtentKey key;
    for (int d = 0; d < SODA_HIP_MAX_DIM; ++d) key[d] = base.extent[d];
    // Never next to a halo exchange (the caller's events say other streams
    // are live: a calibration allocates, frees and synchronises -- it would
    // stall the neighbours and make every rank's schedule depend on who ran
    // beside it), never on a capturing stream (synchronising is illegal
    // there): such runs are scheduled by the model unless the caller
    // calibrated the extent beforehand (soda_hip_program_calibrate).
    const bool beside_exchange = slab && (slab->ghosts_ready || slab->sendable);
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (p->auto_calibrate && !p->calibrating && !beside_exchange)
      if (hipStreamIsCapturing(stream, &capture) != hipSuccess) {
        (void)hipGetLastError();
        capture = hipStreamCaptureStatusNone;
      }
    if (p->auto_calibrate && !p->calibrating && !beside_exchange &&
        capture == hipStreamCaptureStatusNone && iterate > 1 &&
        plan.num_passes > 1 && plan.num_inputs == plan.num_outputs &&
        !p->measured.count(key)) {
      // first run on this extent: let the clock rank the passes
      p->calibrating = true;
      int rc = soda_hip_program_calibrate(p, base.extent, 4, stream_);
      p->calibrating = false;
      if (rc) return rc;
      if (int rc2 = extent_plan(plan, &p->extents, base.extent, &ep)) return rc2;
    }
    auto& memo = const_cast<ExtentPlan*>(ep)->sched;
    auto hit = memo.find(iterate);
    if (hit == memo.end()) {
      // measured launch times of this very extent (soda_hip_program_calibrate)
      // outrank the model
      auto it = p->measured.find(key);
      const std::vector<double>& pass_ns =
          it != p->measured.end() ? it->second : ep->model_ns;
      if (int rc = schedule(plan, pass_ns, iterate, count, &total)) return rc;
      std::vector<int32_t> row(count, count + plan.num_passes);
      row.push_back(total);
      if (memo.size() > 256) memo.clear();
      hit = memo.emplace(iterate, std::move(row)).first;
    }
    for (int i = 0; i < plan.num_passes; ++i) count[i] = hit->second[i];
    total = hit->second[plan.num_passes];
  }

  if (p->debug) base.buf[SODA_HIP_MAX_TENSORS - 1] = p->debug;
  const int in0 = 0, out0 = plan.num_inputs, loc0 = out0 + plan.num_outputs;
  const int prm0 = loc0 + plan.num_locals;
  for (int k = 0; k < plan.num_params; ++k)   // the same in every iteration
    base.buf[prm0 + k] = const_cast<void*>(inputs[plan.num_inputs + k]);
  // scratch for the local tensors -- only if a scheduled pass keeps them in
  // memory (marching kernels hold them in registers)
  bool need_locals = false;
  for (int i = 0; i < plan.num_passes; ++i)
    for (int k = 0; count[i] && k < plan.passes[i].num_kernels; ++k)
      need_locals = need_locals ||
                    plan.kernels[plan.passes[i].kernel[k]].march_dim == 0;
  for (int l = 0; need_locals && l < plan.num_locals; ++l) {
    int rc = ensure(p->locals[l], (size_t)cells * plan.elem_size[loc0 + l]);
    if (rc) return rc;
    base.buf[loc0 + l] = p->locals[l].ptr;
  }
  if (total > 1)
    for (int o = 0; o < plan.num_outputs; ++o) {
      int rc = ensure(p->temps[o], (size_t)cells * plan.elem_size[out0 + o]);
      if (rc) return rc;
    }

  std::vector<PassLaunch> launches;
  if (int rc = plan_launches(plan, &p->extents, base.extent, count, total,
                             iterate, slab, &launches))
    return rc;
  p->last_launches = 0;
  p->last_fused = 0;
  p->last_split = 0;
  p->last_rows = 0;
  std::vector<const void*> src(inputs, inputs + plan.num_inputs);
  const int ax = plan.dim - 1;
  const int32_t rows = base.extent[ax];
  for (size_t done = 0; done < launches.size(); ++done) {
    const PassLaunch& L = launches[done];
    const int i = L.pass;
    // the last pass writes the caller's outputs; before that alternate
    // between the program's temporaries and the caller's outputs
    bool to_out = ((total - 1 - (int)done) % 2) == 0;
    for (int j = 0; j < plan.num_inputs; ++j)
      base.buf[in0 + j] = const_cast<void*>(src[j]);
    for (int o = 0; o < plan.num_outputs; ++o)
      base.buf[out0 + o] = to_out ? outputs[o] : p->temps[o].ptr;
    soda_hip_kargs_t args = base;
    if (L.lo > 0 || L.hi < rows) {
      args.extent[ax] = L.hi - L.lo;
      args.origin[ax] = base.origin[ax] + L.lo;
      for (int t = 0; t < plan.num_inputs + plan.num_outputs; ++t)
        args.buf[t] = static_cast<char*>(base.buf[t]) +
                      (int64_t)L.lo * base.stride[ax] * plan.elem_size[t];
    }
    const ExtentPlan* use = nullptr;
    if (int rc = extent_plan(plan, &p->extents, args.extent, &use)) return rc;
    p->last_rows += L.hi - L.lo;
    const int k0 = plan.passes[i].kernel[0];
    if (L.split && split_in_order()) {
      // both parts on the caller's stream, the part that does not depend on
      // the exchange first: no hand-over between streams, but the two launches
      // do not share the GPU
      const int32_t hole = L.bnd_hi - L.bnd_lo;
      const TileRange boundary = {ax, L.bnd_lo, hole, L.chunks - hole};
      const TileRange interior = {ax, 0, L.bnd_lo, hole};
      if (!L.record) {       // first pass: interior, wait, boundary
        if (int rc = launch(p, k0, args, use->geo[k0].tile, stream, &interior))
          return rc;
        HIP_TRY(hipStreamWaitEvent(stream, slab->ghosts_ready, 0));
        if (int rc = launch(p, k0, args, use->geo[k0].tile, stream, &boundary))
          return rc;
      } else {               // last pass: (wait,) boundary, signal, interior
        if (L.wait) HIP_TRY(hipStreamWaitEvent(stream, slab->ghosts_ready, 0));
        if (int rc = launch(p, k0, args, use->geo[k0].tile, stream, &boundary))
          return rc;
        HIP_TRY(hipEventRecord(slab->sendable, stream));
        if (int rc = launch(p, k0, args, use->geo[k0].tile, stream, &interior))
          return rc;
      }
      p->last_launches += 2;
      if (i == 0) ++p->last_fused;
      ++p->last_split;
    } else if (L.split) {
      // the boundary chunks on the program's side stream, the interior on the
      // caller's: the two share the GPU, the copies run underneath
      if (int rc = side_stream(p)) return rc;
      const int turn = p->ev_turn;
      p->ev_turn ^= 1;
      HIP_TRY(hipEventRecord(p->ev_pre[turn], stream));
      HIP_TRY(hipStreamWaitEvent(p->side, p->ev_pre[turn], 0));
      if (L.wait) HIP_TRY(hipStreamWaitEvent(p->side, slab->ghosts_ready, 0));
      const int32_t hole = L.bnd_hi - L.bnd_lo;
      const TileRange boundary = {ax, L.bnd_lo, hole, L.chunks - hole};
      if (int rc = launch(p, k0, args, use->geo[k0].tile, p->side, &boundary))
        return rc;
      if (L.record) HIP_TRY(hipEventRecord(slab->sendable, p->side));
      HIP_TRY(hipEventRecord(p->ev_bnd[turn], p->side));
      const TileRange interior = {ax, 0, L.bnd_lo, hole};
      if (int rc = launch(p, k0, args, use->geo[k0].tile, stream, &interior))
        return rc;
      HIP_TRY(hipStreamWaitEvent(stream, p->ev_bnd[turn], 0));
      p->last_launches += 2;
      if (i == 0) ++p->last_fused;
      ++p->last_split;
    } else {
      if (L.wait) HIP_TRY(hipStreamWaitEvent(stream, slab->ghosts_ready, 0));
      for (int k = 0; k < plan.passes[i].num_kernels; ++k) {
        const int kk = plan.passes[i].kernel[k];
        int rc = launch(p, kk, args, use->geo[kk].tile, stream);
        if (rc) return rc;
        ++p->last_launches;
        if (i == 0) ++p->last_fused;
      }
      if (L.record) HIP_TRY(hipEventRecord(slab->sendable, stream));
    }
    if ((int)done + 1 < total)
      for (int j = 0; j < plan.num_inputs; ++j) src[j] = base.buf[out0 + j];
  }
  return SODA_HIP_OK;
}

}  // namespace soda_detail

extern "C" {

int soda_hip_run_device(soda_hip_program_t* p, void* const* outputs,
                        const void* const* inputs, const int32_t* extent,
                        int32_t iterate, void* stream_) {
  return run_core(p, outputs, inputs, extent, nullptr, nullptr, iterate,
                  stream_, -1, nullptr);
}

int soda_hip_run_device_window(soda_hip_program_t* p, void* const* outputs,
                               const void* const* inputs,
                               const int32_t* extent, const int32_t* origin,
                               const int32_t* global_extent, int32_t iterate,
                               void* stream_) {
  return run_core(p, outputs, inputs, extent, origin, global_extent, iterate,
                  stream_, -1, nullptr);
}

int soda_hip_run_device_cone(soda_hip_program_t* p, void* const* outputs,
                             const void* const* inputs, const int32_t* extent,
                             const int32_t* origin,
                             const int32_t* global_extent, int32_t iterate,
                             int32_t keep_lo, int32_t keep_hi,
                             int32_t reach_lo, int32_t reach_hi,
                             void* stream_) {
  soda_hip_slab_run_t run;
  memset(&run, 0, sizeof run);
  run.keep_lo = keep_lo;
  run.keep_hi = keep_hi;
  run.reach_lo = reach_lo;
  run.reach_hi = reach_hi;
  return soda_hip_run_device_slab(p, outputs, inputs, extent, origin,
                                  global_extent, iterate, &run, stream_);
}

static int slab_run(const soda_hip_slab_run_t* run, const int32_t* extent,
                    int ax, SlabRun* out);

int soda_hip_run_device_slab(soda_hip_program_t* p, void* const* outputs,
                             const void* const* inputs, const int32_t* extent,
                             const int32_t* origin,
                             const int32_t* global_extent, int32_t iterate,
                             const soda_hip_slab_run_t* run, void* stream_) {
  if (!p || !extent || !run)
    return fail(SODA_HIP_ERR_INVALID, "run_device_slab: NULL argument");
  SlabRun s;
  if (int rc = slab_run(run, extent, p->plan.dim - 1, &s)) return rc;
  return run_core(p, outputs, inputs, extent, origin, global_extent, iterate,
                  stream_, -1, &s);
}

int soda_hip_plan_launches(const soda_hip_plan_t* plan, const int32_t* extent,
                           int32_t iterate, const soda_hip_slab_run_t* run,
                           int32_t capacity, soda_hip_launch_info_t* launches,
                           int32_t* count) {
  if (!plan || !extent || !count || capacity < 0 || (capacity && !launches))
    return fail(SODA_HIP_ERR_INVALID, "plan_launches: bad argument");
  if (iterate < 1) return fail(SODA_HIP_ERR_INVALID, "cannot iterate < 1 times");
  if (int rc = check_plan(plan)) return rc;
  int32_t ext[SODA_HIP_MAX_DIM];
  for (int d = 0; d < SODA_HIP_MAX_DIM; ++d) {
    ext[d] = d < plan->dim ? extent[d] : 1;
    if (ext[d] < 1) return fail(SODA_HIP_ERR_INVALID, "plan_launches: extent < 1");
  }
  SlabRun s;
  if (run)
    if (int rc = slab_run(run, ext, plan->dim - 1, &s)) return rc;
  std::map<ExtentKey, ExtentPlan> cache;
  const ExtentPlan* ep = nullptr;
  if (int rc = extent_plan(*plan, &cache, ext, &ep)) return rc;
  int32_t per_pass[SODA_HIP_MAX_PASSES], total = 0;
  if (int rc = schedule(*plan, ep->model_ns, iterate, per_pass, &total)) return rc;
  std::vector<PassLaunch> list;
  if (int rc = plan_launches(*plan, &cache, ext, per_pass, total, iterate,
                             run ? &s : nullptr, &list))
    return rc;
  *count = (int32_t)list.size();
  for (int32_t i = 0; i < *count && i < capacity; ++i) {
    const PassLaunch& L = list[i];
    launches[i] = {plan->passes[L.pass].fused_iters, L.lo, L.hi, L.wait,
                   L.record, L.split, L.chunk, L.chunks, L.bnd_lo, L.bnd_hi};
  }
  return SODA_HIP_OK;
}

static int slab_run(const soda_hip_slab_run_t* run, const int32_t* extent,
                    int ax, SlabRun* out) {
  if (run->keep_lo < 0 || run->keep_hi > extent[ax] ||
      run->keep_lo >= run->keep_hi || run->reach_lo < 0 || run->reach_hi < 0)
    return fail(SODA_HIP_ERR_INVALID,
                "run_device_slab: [keep_lo, keep_hi) must be a non-empty range "
                "of the last dimension, the reaches non-negative");
  if (run->ghost_lo < 0 || run->ghost_hi < 0 || run->send_lo < 0 ||
      run->send_hi < 0 || run->ghost_lo + run->ghost_hi > extent[ax] ||
      run->send_lo > run->keep_hi - run->keep_lo ||
      run->send_hi > run->keep_hi - run->keep_lo)
    return fail(SODA_HIP_ERR_INVALID,
                "run_device_slab: ghost / send rows out of range");
  SlabRun& s = *out;
  s.cone = {run->keep_lo, run->keep_hi, run->reach_lo, run->reach_hi};
  s.ghost_lo = run->ghost_lo;
  s.ghost_hi = run->ghost_hi;
  s.send_lo = run->send_lo;
  s.send_hi = run->send_hi;
  s.ghosts_ready = static_cast<hipEvent_t>(run->ghosts_ready);
  s.sendable = static_cast<hipEvent_t>(run->sendable);
  return SODA_HIP_OK;
}

int soda_hip_program_calibrate(soda_hip_program_t* p, const int32_t* extent,
                               int32_t launches, void* stream_) {
  if (!p || !extent) return fail(SODA_HIP_ERR_INVALID, "calibrate: NULL argument");
  const soda_hip_plan_t& plan = p->plan;
  if (launches < 2) launches = 4;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  HIP_TRY(hipSetDevice(p->device));
  std::array<int32_t, SODA_HIP_MAX_DIM> key;
  int64_t cells = 1;
  for (int d = 0; d < SODA_HIP_MAX_DIM; ++d) {
    key[d] = d < plan.dim ? extent[d] : 1;
    if (key[d] < 1) return fail(SODA_HIP_ERR_INVALID, "calibrate: extent < 1");
    cells *= key[d];
  }
  if (plan.num_passes < 2 || plan.num_inputs != plan.num_outputs) {
    p->measured.erase(key);          // nothing to choose between
    auto it = p->extents.find(key);
    if (it != p->extents.end()) it->second.sched.clear();
    return SODA_HIP_OK;
  }
  // stand-in arrays: inputs (a byte pattern that reads as ~0.75 in fp32, small
  // positive integers otherwise), outputs, params
  std::vector<DeviceBuffer> bufs(plan.num_inputs + plan.num_outputs +
                                 plan.num_params);
  std::vector<const void*> ins;
  std::vector<void*> outs;
  int rc = SODA_HIP_OK;
  const int prm0 = plan.num_inputs + plan.num_outputs + plan.num_locals;
  for (size_t b = 0; b < bufs.size() && rc == SODA_HIP_OK; ++b) {
    size_t bytes;
    if ((int)b < plan.num_inputs + plan.num_outputs)
      bytes = (size_t)cells * plan.elem_size[b];
    else
      bytes = (size_t)plan.param_elems[b - plan.num_inputs - plan.num_outputs] *
              plan.elem_size[prm0 + (b - plan.num_inputs - plan.num_outputs)];
    rc = ensure(bufs[b], bytes);
    if (rc == SODA_HIP_OK &&
        hipMemsetAsync(bufs[b].ptr, 0x3f, bytes, stream) != hipSuccess)
      rc = fail(SODA_HIP_ERR_RUNTIME, "calibrate: hipMemsetAsync");
  }
  std::vector<double> ns(plan.num_passes, 0.0);
  // Two rounds over all the passes bring the clocks up (a burst of launches a
  // millisecond after idle reads 20-30 % slow, and not equally per pass: the
  // issue-bound deep passes follow the core clock, the memory-bound shallow
  // ones do not); then kTimedRounds rounds in which every pass runs once
  // untimed and `launches` times inside an event pair -- a schedule runs a
  // pass many times in a row, and the first launch behind a different kernel
  // is not representative (jacobi2d T = 12 right behind the memory-bound
  // T = 1: 167 us, sustained 146).  A pass's time is the SHORTEST of its
  // rounds: the clocks keep rising through the first milliseconds, and one
  // round alone priced T = 13 at 161-173 us (sustained 144), enough to flip
  // the schedule of 100 iterations between 4 x 13 + 4 x 12 and 7 x 12 + 2 x 8
  // from run to run.  Nothing synchronises in between, so the GPU never idles.
  constexpr int kTimedRounds = 4;
  std::vector<hipEvent_t> ev(2 * plan.num_passes * kTimedRounds, nullptr);
  for (auto& e : ev)
    if (rc == SODA_HIP_OK && hipEventCreate(&e) != hipSuccess)
      rc = fail(SODA_HIP_ERR_RUNTIME, "calibrate: hipEventCreate");
  if (rc == SODA_HIP_OK) {
    for (int i = 0; i < plan.num_inputs; ++i) ins.push_back(bufs[i].ptr);
    for (int k = 0; k < plan.num_params; ++k)
      ins.push_back(bufs[plan.num_inputs + plan.num_outputs + k].ptr);
    for (int o = 0; o < plan.num_outputs; ++o)
      outs.push_back(bufs[plan.num_inputs + o].ptr);
    for (int round = 0; round < 2 && rc == SODA_HIP_OK; ++round)
      for (int i = 0; i < plan.num_passes && rc == SODA_HIP_OK; ++i)
        rc = run_core(p, outs.data(), ins.data(), key.data(), nullptr, nullptr,
                      plan.passes[i].fused_iters * launches, stream_, i,
                      nullptr);
    for (int round = 0; round < kTimedRounds && rc == SODA_HIP_OK; ++round)
      for (int i = 0; i < plan.num_passes && rc == SODA_HIP_OK; ++i) {
        const int32_t one = plan.passes[i].fused_iters;
        hipEvent_t* pair = &ev[2 * (round * plan.num_passes + i)];
        rc = run_core(p, outs.data(), ins.data(), key.data(), nullptr, nullptr,
                      one, stream_, i, nullptr);
        if (rc != SODA_HIP_OK) break;
        if (hipEventRecord(pair[0], stream) != hipSuccess)
          rc = fail(SODA_HIP_ERR_RUNTIME, "calibrate: hipEventRecord");
        if (rc != SODA_HIP_OK) break;
        rc = run_core(p, outs.data(), ins.data(), key.data(), nullptr, nullptr,
                      one * launches, stream_, i, nullptr);
        if (rc == SODA_HIP_OK && hipEventRecord(pair[1], stream) != hipSuccess)
          rc = fail(SODA_HIP_ERR_RUNTIME, "calibrate: hipEventRecord");
      }
    if (rc == SODA_HIP_OK && hipStreamSynchronize(stream) != hipSuccess)
      rc = fail(SODA_HIP_ERR_RUNTIME, "calibrate: hipStreamSynchronize");
    for (int i = 0; i < plan.num_passes && rc == SODA_HIP_OK; ++i)
      for (int round = 0; round < kTimedRounds && rc == SODA_HIP_OK; ++round) {
        float ms = 0;
        hipEvent_t* pair = &ev[2 * (round * plan.num_passes + i)];
        if (hipEventElapsedTime(&ms, pair[0], pair[1]) != hipSuccess)
          rc = fail(SODA_HIP_ERR_RUNTIME, "calibrate: hipEventElapsedTime");
        const double t = ms * 1e6 / launches;
        if (round == 0 || t < ns[i]) ns[i] = t;
      }
  }
  for (auto& e : ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& b : bufs)
    if (b.ptr) (void)hipFree(b.ptr);
  if (rc == SODA_HIP_OK) {
    p->measured[key] = ns;
    auto it = p->extents.find(key);        // schedules made from the model
    if (it != p->extents.end()) it->second.sched.clear();
  }
  return rc;
}

int soda_hip_program_set_auto_calibrate(soda_hip_program_t* p, int on) {
  if (!p) return fail(SODA_HIP_ERR_INVALID, "NULL program");
  p->auto_calibrate = on != 0;
  return SODA_HIP_OK;
}

int soda_hip_program_pass_times(soda_hip_program_t* p, const int32_t* extent,
                                float* pass_ns, int32_t* measured) {
  if (!p || !extent || !pass_ns)
    return fail(SODA_HIP_ERR_INVALID, "pass_times: NULL argument");
  const soda_hip_plan_t& plan = p->plan;
  std::array<int32_t, SODA_HIP_MAX_DIM> key;
  for (int d = 0; d < SODA_HIP_MAX_DIM; ++d) key[d] = d < plan.dim ? extent[d] : 1;
  auto it = p->measured.find(key);
  if (measured) *measured = it != p->measured.end();
  if (it != p->measured.end()) {
    for (int i = 0; i < plan.num_passes; ++i) pass_ns[i] = (float)it->second[i];
    return SODA_HIP_OK;
  }
  return plan_geometry_c(&plan, key.data(), nullptr, pass_ns);
}

int soda_hip_program_schedule(soda_hip_program_t* p, const int32_t* extent,
                              int32_t iterate, int32_t* count) {
  if (!p || !extent || !count)
    return fail(SODA_HIP_ERR_INVALID, "program_schedule: NULL argument");
  if (iterate < 1) return fail(SODA_HIP_ERR_INVALID, "cannot iterate < 1 times");
  std::vector<float> ns(p->plan.num_passes);
  if (int rc = soda_hip_program_pass_times(p, extent, ns.data(), nullptr))
    return rc;
  std::vector<double> t(ns.begin(), ns.end());
  int32_t total = 0;
  return schedule(p->plan, t, iterate, count, &total);
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

int soda_hip_last_split(soda_hip_program_t* p, int32_t* passes) {
  if (!p || !passes) return fail(SODA_HIP_ERR_INVALID, "NULL argument");
  *passes = p->last_split;
  return SODA_HIP_OK;
}

int soda_hip_last_rows(soda_hip_program_t* p, int64_t* rows) {
  if (!p || !rows) return fail(SODA_HIP_ERR_INVALID, "NULL argument");
  *rows = p->last_rows;
  return SODA_HIP_OK;
}

// -- wire format: <app>_kernel on banked streams -------------------------------

struct soda_hip_stream {
  soda_hip_stream_desc_t desc;
  soda_hip_program* dense = nullptr;
  std::vector<soda_hip_program*> linear, unwire, wire;
  std::vector<DeviceBuffer> dense_in, dense_out;   // de-interleaved streams
  std::vector<DeviceBuffer> host_banks;            // run_host staging
  bool dense_failed = false;
  int last_mode = 0;
  // Device-resident banks narrower than this run the linear form even where
  // the dense view exists: narrow tiles leave most of a 64-lane x V-wide
  // marching strip idle (heat3d 32 x 32 tiles: 0.50 vs 0.79 ms).  Host banks
  // take the dense view whenever there is one -- there the copies dominate and
  // the dense view is what lets them overlap in bands.
  int device_dense_min_tile0 = 256;
};

int soda_hip_stream_create(const soda_hip_stream_desc_t* desc,
                           soda_hip_program_t* dense,
                           soda_hip_program_t* const* linear,
                           soda_hip_program_t* const* unwire,
                           soda_hip_program_t* const* wire,
                           soda_hip_stream_t** stream) {
  if (!desc || !linear || !wire || !stream)
    return fail(SODA_HIP_ERR_INVALID, "stream_create: NULL argument");
  *stream = nullptr;
  const int nt = desc->num_inputs + desc->num_outputs;
  if (desc->dim < 1 || desc->dim > SODA_HIP_MAX_DIM || desc->num_inputs < 1 ||
      desc->num_outputs < 1 || nt > SODA_HIP_MAX_TENSORS || desc->iterate < 1 ||
      desc->num_linear < 1 || desc->num_linear > 4 ||
      desc->linear_vec[desc->num_linear - 1] != 1)
    return fail(SODA_HIP_ERR_INVALID, "stream_create: bad description");
  for (int t = 0; t < nt; ++t)
    if (desc->banks[t] < 1 || desc->elem_size[t] < 1 ||
        desc->elems_per_cycle[t] < 1 || desc->shift[t] < 0)
      return fail(SODA_HIP_ERR_INVALID, "stream_create: bad tensor entry");
  for (int t = 1; t < nt; ++t)
    if (desc->elems_per_cycle[t] != desc->elems_per_cycle[0])
      return fail(SODA_HIP_ERR_UNSUPPORTED,
                  "stream mode needs every tensor to move the same number of "
                  "elements per cycle (burst width / element width x banks)");
  for (int i = 0; i < desc->num_inputs; ++i)
    if ((desc->banks[i] > 1 || desc->shift[i]) && (!unwire || !unwire[i]))
      return fail(SODA_HIP_ERR_INVALID, "stream_create: missing unwire kernel");
  for (int k = 0; k < desc->num_linear; ++k)
    if (!linear[k]) return fail(SODA_HIP_ERR_INVALID, "stream_create: NULL linear");
  // an output on one bank that the program stores at its wire position (shift
  // 0) needs no copy kernel: the program writes the caller's bank
  for (int o = 0; o < desc->num_outputs; ++o) {
    const int t = desc->num_inputs + o;
    if (!wire[o] && (desc->banks[t] > 1 || desc->shift[t]))
      return fail(SODA_HIP_ERR_INVALID, "stream_create: missing wire kernel");
  }
  soda_hip_stream* s = new (std::nothrow) soda_hip_stream;
  if (!s) return fail(SODA_HIP_ERR_NOMEM, "new stream");
  s->desc = *desc;
  s->dense = dense;
  s->dense_failed = dense == nullptr;
  s->linear.assign(linear, linear + desc->num_linear);
  s->unwire.resize(desc->num_inputs, nullptr);
  for (int i = 0; unwire && i < desc->num_inputs; ++i) s->unwire[i] = unwire[i];
  s->wire.assign(wire, wire + desc->num_outputs);
  s->dense_in.resize(desc->num_inputs);
  s->dense_out.resize(desc->num_outputs);
  *stream = s;
  return SODA_HIP_OK;
}

int soda_hip_stream_destroy(soda_hip_stream_t* s) {
  if (!s) return SODA_HIP_OK;
  for (auto* v : {&s->dense_in, &s->dense_out, &s->host_banks})
    for (auto& b : *v)
      if (b.ptr) (void)hipFree(b.ptr);
  delete s;
  return SODA_HIP_OK;
}

int soda_hip_stream_last_mode(soda_hip_stream_t* s) { return s ? s->last_mode : 0; }

int soda_hip_stream_set_device_dense_min_tile(soda_hip_stream_t* s,
                                              int32_t min_tile0) {
  if (!s || min_tile0 < 0)
    return fail(SODA_HIP_ERR_INVALID, "stream_set_device_dense_min_tile");
  s->device_dense_min_tile0 = min_tile0;
  return SODA_HIP_OK;
}

// Dense view: a stream of n elements is an array of extent (tile..., rows) iff
// every tile starts on a row-block boundary -- a tile occupies
// round_up(block * extent_last, epc) elements (frt/host.py:137-142), so for any
// extent iff block % epc == 0 -- and the void tail (kStencilDistance elements,
// :151-162) covers the partial last row the view drops.
static bool dense_view(const soda_hip_stream* s, int64_t n, int32_t* ext) {
  const soda_hip_stream_desc_t& d = s->desc;
  if (s->dense_failed || d.dim < 2) return false;
  int64_t block = 1;
  for (int t = 0; t < d.dim - 1; ++t) block *= d.tile[t];
  const int64_t rows = n / block;
  if (rows < 1 || d.stencil_distance < block ||
      block % d.elems_per_cycle[0] != 0)
    return false;
  for (int t = 0; t < SODA_HIP_MAX_DIM; ++t) ext[t] = 1;
  for (int t = 0; t < d.dim - 1; ++t) ext[t] = d.tile[t];
  ext[d.dim - 1] = (int32_t)rows;
  return true;
}

int soda_hip_stream_run_device(soda_hip_stream_t* s, void* const* out_banks,
                               const void* const* in_banks,
                               uint64_t coalesced_data_num, void* hip_stream) {
  if (!s || !out_banks || !in_banks)
    return fail(SODA_HIP_ERR_INVALID, "stream_run: NULL argument");
  const soda_hip_stream_desc_t& d = s->desc;
  const uint64_t n64 = coalesced_data_num * (uint64_t)d.elems_per_cycle[0];
  if (n64 < 1 || n64 >= (1ull << 31))
    return fail(SODA_HIP_ERR_INVALID, "stream_run: stream length out of range");
  const int32_t n = (int32_t)n64;
  const int32_t ext1[1] = {n};
  HIP_TRY(hipSetDevice(s->linear.back()->device));
  // 1. un-interleave (and un-delay) the inputs
  std::vector<const void*> din(d.num_inputs);
  int bank0 = 0;
  for (int i = 0; i < d.num_inputs; ++i) {
    const int nb = d.banks[i];
    for (int b = 0; b < nb; ++b)
      if (!in_banks[bank0 + b])
        return fail(SODA_HIP_ERR_INVALID, "stream_run: NULL input bank");
    if (nb == 1 && d.shift[i] == 0) {
      if (reinterpret_cast<uintptr_t>(in_banks[bank0]) & 15)
        return fail(SODA_HIP_ERR_INVALID,
                    "stream_run: a bank the program reads in place must be "
                    "16-byte aligned");
      din[i] = in_banks[bank0];     // the bank IS the dense stream
    } else {
      if (int rc = ensure(s->dense_in[i], (size_t)n * d.elem_size[i])) return rc;
      void* outs[1] = {s->dense_in[i].ptr};
      if (int rc = soda_hip_run_device(s->unwire[i], outs, in_banks + bank0, ext1,
                                       1, hip_stream))
        return rc;
      din[i] = s->dense_in[i].ptr;
    }
    bank0 += nb;
  }
  std::vector<void*> dout(d.num_outputs);
  bank0 = 0;
  for (int o = 0; o < d.num_outputs; bank0 += d.banks[d.num_inputs + o], ++o) {
    for (int b = 0; b < d.banks[d.num_inputs + o]; ++b)
      if (!out_banks[bank0 + b])
        return fail(SODA_HIP_ERR_INVALID, "stream_run: NULL output bank");
    if (!s->wire[o]) {
      if (reinterpret_cast<uintptr_t>(out_banks[bank0]) & 15)
        return fail(SODA_HIP_ERR_INVALID,
                    "stream_run: a bank the program writes in place must be "
                    "16-byte aligned");
      dout[o] = out_banks[bank0];   // born at its wire position, in place
      continue;
    }
    const size_t bytes = (size_t)n * d.elem_size[d.num_inputs + o];
    const void* before = s->dense_out[o].ptr;
    if (int rc = ensure(s->dense_out[o], bytes)) return rc;
    // new memory: the dense view drops the partial last row, whose bytes would
    // otherwise reach the caller's banks uninitialised through wire_<out>
    if (s->dense_out[o].ptr != before)
      HIP_TRY(hipMemsetAsync(s->dense_out[o].ptr, 0, s->dense_out[o].bytes,
                             static_cast<hipStream_t>(hip_stream)));
    dout[o] = s->dense_out[o].ptr;
  }
  // 2. the program, on the dense view where there is one
  bool done = false;
  {
    int32_t ext[SODA_HIP_MAX_DIM];
    if (d.tile[0] >= s->device_dense_min_tile0 && dense_view(s, n, ext)) {
      int rc = soda_hip_run_device(s->dense, dout.data(), din.data(), ext,
                                   d.iterate, hip_stream);
      if (rc == SODA_HIP_OK) {
        done = true;
        s->last_mode = 1;
      } else if (rc != SODA_HIP_ERR_INVALID) {
        return rc;
      }   // INVALID: this extent does not suit the dense kernels; go linear
    }
  }
  if (!done) {
    int pick = d.num_linear - 1;
    for (int k = 0; k < d.num_linear; ++k)
      if (n % d.linear_vec[k] == 0) { pick = k; break; }
    if (int rc = soda_hip_run_device(s->linear[pick], dout.data(), din.data(),
                                     ext1, d.iterate, hip_stream))
      return rc;
    s->last_mode = 2;
  }
  // 3. outputs: shifted by the stencil offset, re-interleaved
  bank0 = 0;
  for (int o = 0; o < d.num_outputs; bank0 += d.banks[d.num_inputs + o], ++o) {
    if (!s->wire[o]) continue;
    const void* ins[1] = {dout[o]};
    if (int rc = soda_hip_run_device(s->wire[o], out_banks + bank0, ins, ext1, 1,
                                     hip_stream))
      return rc;
  }
  return SODA_HIP_OK;
}

int soda_hip_stream_run_host(soda_hip_stream_t* s, void* const* out_banks,
                             const void* const* in_banks,
                             uint64_t coalesced_data_num) {
  if (!s || !out_banks || !in_banks)
    return fail(SODA_HIP_ERR_INVALID, "stream_run_host: NULL argument");
  const soda_hip_stream_desc_t& d = s->desc;
  // Every tensor on one bank that the program reads / writes in place, and the
  // stream a dense array of rows: the caller's banks ARE host arrays of the
  // n-D program, and the call is the host-array entry's (soda_host.cpp) --
  // copy-in, kernels and copy-out overlapped in bands along the rows.  The
  // elements of the partial last row stay as the caller left them (void tail).
  // Banked tensors ride along: their (de)interleave is done by the host
  // threads in the pack / unpack step instead of by copy kernels on the GPU.
  {
    bool in_place = !getenv("SODA_HIP_STREAM_NO_BANDS");
    // (a delayed input on ONE bank is un-delayed in the pack step as well)
    for (int t = 0; t < d.num_inputs + d.num_outputs; ++t)
      in_place = in_place && (d.shift[t] == 0 ||
                              (t < d.num_inputs && d.banks[t] == 1));
    const uint64_t n64 = coalesced_data_num * (uint64_t)d.elems_per_cycle[0];
    int32_t ext[SODA_HIP_MAX_DIM];
    if (in_place && n64 >= 1 && n64 < (1ull << 31) &&
        dense_view(s, (int64_t)n64, ext)) {
      int32_t stride[SODA_HIP_MAX_DIM];
      int64_t run = 1;
      for (int t = 0; t < SODA_HIP_MAX_DIM; ++t) {
        stride[t] = run > INT32_MAX ? INT32_MAX : (int32_t)run;   // (t >= dim: unread)
        run *= ext[t];
      }
      std::vector<soda_hip_host_tensor_t> tin(d.num_inputs), tout(d.num_outputs);
      bool null_bank = false;
      int bank = 0;
      for (int i = 0; i < d.num_inputs; bank += d.banks[i], ++i) {
        for (int b = 0; b < d.banks[i]; ++b)
          null_bank = null_bank || !in_banks[bank + b];
        // one bank: the array itself; several: the list of them
        tin[i] = {d.banks[i] == 1
                      ? const_cast<void*>(in_banks[bank])
                      : const_cast<void*>(static_cast<const void*>(in_banks + bank)),
                  ext, stride, nullptr};
      }
      bank = 0;
      for (int o = 0; o < d.num_outputs; bank += d.banks[d.num_inputs + o], ++o) {
        for (int b = 0; b < d.banks[d.num_inputs + o]; ++b)
          null_bank = null_bank || !out_banks[bank + b];
        tout[o] = {d.banks[d.num_inputs + o] == 1
                       ? out_banks[bank]
                       : const_cast<void*>(static_cast<const void*>(out_banks + bank)),
                   ext, stride, nullptr};
      }
      if (null_bank)
        return fail(SODA_HIP_ERR_INVALID, "stream_run_host: NULL bank");
      int rc = run_host_call(s->dense, tin.data(), tout.data(), d.iterate,
                             nullptr, nullptr, d.banks, d.shift, (int64_t)n64);
      if (rc == SODA_HIP_OK) {
        s->last_mode = 1;
        return rc;
      }
      if (rc != SODA_HIP_ERR_INVALID) return rc;
      // INVALID: this extent does not suit the dense kernels; the long way
    }
  }
  // Otherwise the banks -- the generated host's pageable `aligned_alloc`
  // buffers (ref frt/host.py:165-178) -- travel whole through the pinned rings
  // and the worker threads of the host-array entry, on one stream with the
  // kernels.
  soda_hip_program* owner = s->linear.back();
  hipStream_t stream = nullptr;
  if (int rc = host_stream(owner, &stream)) return rc;
  int total = 0;
  for (int t = 0; t < d.num_inputs + d.num_outputs; ++t) total += d.banks[t];
  s->host_banks.resize(total);
  std::vector<const void*> dev_in;
  std::vector<void*> dev_out;
  int slot = 0, bank = 0;
  for (int i = 0; i < d.num_inputs; ++i)
    for (int b = 0; b < d.banks[i]; ++b, ++slot, ++bank) {
      const size_t bytes = (size_t)coalesced_data_num * d.elems_per_cycle[i] /
                           d.banks[i] * d.elem_size[i];
      if (!in_banks[bank])
        return fail(SODA_HIP_ERR_INVALID, "stream_run_host: NULL input bank");
      if (int rc = ensure(s->host_banks[slot], bytes)) return rc;
      if (int rc = ring_send(owner, s->host_banks[slot].ptr, in_banks[bank],
                             bytes, stream))
        return rc;
      dev_in.push_back(s->host_banks[slot].ptr);
    }
  std::vector<size_t> out_bytes;
  for (int o = 0; o < d.num_outputs; ++o)
    for (int b = 0; b < d.banks[d.num_inputs + o]; ++b, ++slot) {
      const int t = d.num_inputs + o;
      const size_t bytes = (size_t)coalesced_data_num * d.elems_per_cycle[t] /
                           d.banks[t] * d.elem_size[t];
      const void* before = s->host_banks[slot].ptr;
      if (int rc = ensure(s->host_banks[slot], bytes)) return rc;
      // void positions a program writing in place never touches: the same
      // bytes for the caller run after run
      if (s->host_banks[slot].ptr != before)
        HIP_TRY(hipMemsetAsync(s->host_banks[slot].ptr, 0, bytes, stream));
      dev_out.push_back(s->host_banks[slot].ptr);
      out_bytes.push_back(bytes);
    }
  for (size_t k = 0; k < dev_out.size(); ++k)
    if (!out_banks[k])
      return fail(SODA_HIP_ERR_INVALID, "stream_run_host: NULL output bank");
  if (int rc = soda_hip_stream_run_device(s, dev_out.data(), dev_in.data(),
                                          coalesced_data_num, stream))
    return rc;
  for (size_t k = 0; k < dev_out.size(); ++k)
    if (int rc = ring_fetch(owner, out_banks[k], dev_out[k], out_bytes[k],
                            stream))
      return rc;
  HIP_TRY(hipStreamSynchronize(stream));
  return SODA_HIP_OK;
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

int soda_hip_memcpy_d2d(void* dst, int32_t dst_device, const void* src,
                        int32_t src_device, size_t bytes, void* stream) {
  if (dst_device == src_device)
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice,
                           static_cast<hipStream_t>(stream)));
  else
    HIP_TRY(hipMemcpyPeerAsync(dst, dst_device, src, src_device, bytes,
                               static_cast<hipStream_t>(stream)));
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

int soda_hip_event_handle(soda_hip_event_t* event, void** hip_event) {
  if (!event || !hip_event) return fail(SODA_HIP_ERR_INVALID, "event_handle: NULL");
  *hip_event = event->ev;
  return SODA_HIP_OK;
}

int soda_hip_hipstream_create(int32_t device, void** stream) {
  if (!stream) return fail(SODA_HIP_ERR_INVALID, "hipstream_create: NULL");
  *stream = nullptr;
  HIP_TRY(hipSetDevice(device));
  hipStream_t s = nullptr;
  HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = s;
  return SODA_HIP_OK;
}

int soda_hip_hipstream_destroy(void* stream) {
  if (stream) HIP_TRY(hipStreamDestroy(static_cast<hipStream_t>(stream)));
  return SODA_HIP_OK;
}

int soda_hip_hipstream_wait_event(void* stream, soda_hip_event_t* event) {
  if (!event) return fail(SODA_HIP_ERR_INVALID, "hipstream_wait_event: NULL");
  HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), event->ev, 0));
  return SODA_HIP_OK;
}

int soda_hip_event_destroy(soda_hip_event_t* event) {
  if (!event) return SODA_HIP_OK;
  (void)hipEventDestroy(event->ev);
  delete event;
  return SODA_HIP_OK;
}

}  // extern "C"
