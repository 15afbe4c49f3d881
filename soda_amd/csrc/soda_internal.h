// soda_internal.h -- what the translation units of libsoda_hip.so share
// (soda_hip.cpp: programs, launch geometry, runs; soda_group.cpp: slab groups).
// Nothing here crosses the C ABI (include/soda_hip.h).
#ifndef SODA_INTERNAL_H_
#define SODA_INTERNAL_H_

#include "soda_hip.h"

#include <hip/hip_runtime.h>

#include <array>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

namespace soda_detail {

int fail(int status, const std::string& what);
const std::string& last_error_text();   // of the calling thread
int hip_fail(hipError_t e, const char* what);

#define HIP_TRY(expr)                                              \
  do {                                                             \
    hipError_t e_ = (expr);                                        \
    if (e_ != hipSuccess) return soda_detail::hip_fail(e_, #expr); \
  } while (0)

struct DeviceBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
};
int ensure(DeviceBuffer& b, size_t bytes);

typedef std::array<int32_t, SODA_HIP_MAX_DIM> ExtentKey;

struct Geometry {
  int32_t tile[SODA_HIP_MAX_DIM];
  double ns;          // modelled time of one launch; 0: no model
};

// What a run on one extent uses.  Sizing the chunks of every kernel costs tens
// of microseconds of host time (more on tall grids) and the schedule a
// knapsack over the iteration count: both are remembered per extent, a launch
// must not pay for them again.
struct ExtentPlan {
  std::vector<Geometry> geo;      // per kernel
  std::vector<double> model_ns;   // per pass, from the descriptors' model
  // iterate -> launches of every pass, then their sum (num_passes + 1 values)
  std::map<int32_t, std::vector<int32_t>> sched;
};

// rows (cells along the last dimension) a run must deliver, and how far one
// iteration reaches beyond a row on either side: passes may then skip the rows
// whose results no later iteration of the run can carry into [keep_lo, keep_hi)
struct Cone {
  int32_t keep_lo, keep_hi, reach_lo, reach_hi;
};

// A run between two halo exchanges (include/soda_hip.h, soda_hip_slab_run_t)
struct SlabRun {
  Cone cone;
  int32_t ghost_lo, ghost_hi;   // input rows an exchange in flight refreshes
  int32_t send_lo, send_hi;     // result rows the neighbours fetch next
  hipEvent_t ghosts_ready;      // may be null
  hipEvent_t sendable;          // may be null
};

// Pinned staging slots of the host-array entry (soda_host.cpp): chunk i + 1
// is packed by worker threads while the DMA engine moves chunk i.
struct HostRing {
  static const int kMaxSlots = 4;
  char* base = nullptr;
  size_t slot_bytes = 0;
  int slots = 0;
  hipEvent_t ev[kMaxSlots] = {nullptr, nullptr, nullptr, nullptr};
  // a DMA out of / into the slot may still be in flight (its event recorded):
  // whoever fills the slot next waits for the event first
  bool busy[kMaxSlots] = {false, false, false, false};
  int ensure(size_t slot_bytes, int slots);
  int wait(int slot);      // until the slot's last DMA is done
  void release();
};

}  // namespace soda_detail

struct soda_hip_program {
  soda_hip_plan_t plan;
  int device = 0;
  hipModule_t module = nullptr;
  std::vector<hipFunction_t> functions;
  std::vector<soda_detail::DeviceBuffer> locals;   // one per local tensor
  std::vector<soda_detail::DeviceBuffer> temps;    // one per output: ping-pong
  std::vector<soda_detail::DeviceBuffer> host_in;  // run_host staging
  std::vector<soda_detail::DeviceBuffer> host_prm; // ... of the param arrays
  std::vector<soda_detail::DeviceBuffer> host_out;
  // ... its streams (copies in / kernels / copies out), pinned slots, the
  // band output arrays and events of the banded way (soda_host.cpp)
  hipStream_t hstream[3] = {nullptr, nullptr, nullptr};
  soda_detail::HostRing ring_in, ring_out;
  std::vector<soda_detail::DeviceBuffer> band_out;
  std::vector<hipEvent_t> hevents;
  bool hbuf_used[2] = {false, false};
  int32_t last_launches = 0;
  int32_t last_fused = 0;
  int32_t last_split = 0;    // passes of the last run launched in two parts
  int64_t last_rows = 0;     // cells along the last dimension, summed over passes
  void* debug = nullptr;     // time-stamp buffer of diagnostic builds
  // time every pass on an extent the first time it is run (a few ms, once):
  // on unless SODA_HIP_NO_CALIBRATE is set or the caller turns it off
  bool auto_calibrate = true;
  bool calibrating = false;
  // measured time of one launch of every pass, per extent (calibrate)
  std::map<soda_detail::ExtentKey, std::vector<double>> measured;
  std::map<soda_detail::ExtentKey, soda_detail::ExtentPlan> extents;
  // split passes: the chunks next to a slab's ghost rows run on a stream of
  // their own, beside the interior on the caller's stream
  hipStream_t side = nullptr;
  hipEvent_t ev_pre[2] = {nullptr, nullptr};
  hipEvent_t ev_bnd[2] = {nullptr, nullptr};
  int ev_turn = 0;
};

namespace soda_detail {

// force_pass >= 0: use only that pass (calibration)
int run_core(soda_hip_program* p, void* const* outputs,
             const void* const* inputs, const int32_t* extent,
             const int32_t* origin, const int32_t* global_extent,
             int32_t iterate, void* stream, int force_pass,
             const SlabRun* slab);

// one launch (or, split, one pair of launches) of a run
struct PassLaunch {
  int pass;                  // index into plan.passes
  int32_t lo, hi;            // rows of the last dimension it covers
  bool wait, record, split;  // behind ghosts_ready / ahead of sendable / in two
  int32_t chunk, chunks;     // split: rows per block tile, tiles along the axis
  int32_t bnd_lo, bnd_hi;    // boundary tiles: [0, bnd_lo) U [bnd_hi, chunks)
};
int plan_launches(const soda_hip_plan_t& plan,
                  std::map<ExtentKey, ExtentPlan>* cache, const int32_t* ext,
                  const int32_t* count, int32_t total, int32_t iterate,
                  const SlabRun* slab, std::vector<PassLaunch>* out);

// tiles, modelled pass times and schedules of `plan` on `ext` (all
// SODA_HIP_MAX_DIM entries filled), remembered in `cache`
int extent_plan(const soda_hip_plan_t& plan,
                std::map<ExtentKey, ExtentPlan>* cache, const int32_t* ext,
                const ExtentPlan** out);

// launches of every pass for `iterate` iterations by the given pass times
int schedule(const soda_hip_plan_t& plan, const std::vector<double>& pass_ns,
             int32_t iterate, int32_t* count, int32_t* total);

void copy_box(char* strided, const int32_t* stride, char* dense,
              const int32_t* extent, const int32_t* lo, const int32_t* hi,
              int dim, int elem, bool to_dense);
bool is_dense(const soda_hip_host_tensor_t& t, int dim);
// the same for a dense array that starts at index `row0` of the last
// dimension, on up to `threads` threads (0: the pool's; soda_host.cpp)
// soda_hip_run_host_box, with tensors as the wire format's host streams hold
// them: nbanks (NULL: none) holds, per tensor, inputs then outputs, the number
// of DRAM banks -- where it is > 1 the tensor's `ptr` is the list of its bank
// pointers; lead (NULL: none), per input, the elements the generated host
// delayed a single-bank stream of stream_elems elements by.
int run_host_call(soda_hip_program* p, const soda_hip_host_tensor_t* inputs,
                  const soda_hip_host_tensor_t* outputs, int32_t iterate,
                  const int32_t* valid_lo, const int32_t* valid_hi,
                  const int32_t* nbanks, const int32_t* lead,
                  int64_t stream_elems);
void weave_banks(char* const* banks, int nb, char* dense, int64_t first,
                 int64_t count, int elem, bool to_dense, int threads);
bool host_pinned(const void* ptr, size_t bytes);
void copy_rows(char* strided, const int32_t* stride, char* dense,
               const int32_t* extent, const int32_t* lo, const int32_t* hi,
               int dim, int elem, bool to_dense, int32_t row0, int threads);

// contiguous host <-> device copies through a program's pinned rings and the
// worker pool (soda_host.cpp); ring_fetch returns when the bytes are delivered
int ring_send(soda_hip_program* p, void* dev, const void* host, size_t bytes,
              hipStream_t stream);
int ring_fetch(soda_hip_program* p, void* host, const void* dev, size_t bytes,
               hipStream_t stream);
// rows [a, b) of a host tensor -> device / the part of box [lo, hi) in rows
// [a, b) of a device array -> a host tensor, through the rings (soda_host.cpp);
// `dev` holds the dense array from row `dev_row0` of the last dimension on
int send_rows(soda_hip_program* p, const soda_hip_host_tensor_t& t,
              const int32_t* extent, int dim, int elem, int64_t a, int64_t b,
              void* dev, int64_t dev_row0, hipStream_t stream);
int fetch_rows(soda_hip_program* p, const soda_hip_host_tensor_t& t,
               const int32_t* extent, const int32_t* lo, const int32_t* hi,
               int dim, int elem, int64_t a, int64_t b, const void* dev,
               int64_t dev_row0, hipStream_t stream);
// the program's stream of host-array runs, made on first use
int host_stream(soda_hip_program* p, hipStream_t* stream);

}  // namespace soda_detail

#endif  // SODA_INTERNAL_H_
