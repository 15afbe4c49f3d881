// soda_host.cpp -- the operator-level entry on HOST arrays,
// soda_hip_run_host_box: what `soda::app::<app>()` is in the reference's
// generated host (reference src/soda/codegen/frt/host.py:62-88; scatter
// :181-249, WriteToDevice / Exec / ReadFromDevice / Finish :319-322, gather of
// the valid box only :340-427).
//
// Rounds 1-4 copied with synchronous hipMemcpy from and to the caller's
// pageable arrays and gathered the valid box through a std::vector: 82 ms for
// the headline step (2 x 256 MiB moved, 1.1 ms of kernels).  Measured on the
// pool's MI355X boxes (tools/experiments/r05_hostcopy_bench*.cpp,
// profiles/r05_hostcopy*.jsonl): DMA from pinned memory 57 GB/s each way (48
// each with both directions busy); hipHostRegister on a caller's fresh 4 KiB
// pages 42 us/MiB and serialised between threads -- no faster than the
// driver's own pageable path (11 ms per 256 MiB); host memcpy between pageable
// and pinned memory 31 GB/s on one thread, 100 GB/s on eight.  So the copies
// go through a ring of pinned staging slots the program owns: worker threads
// pack chunk i+1 (the reference packs with `#pragma omp parallel for`,
// frt/host.py:193) while the DMA engine moves chunk i, and the other way round
// on the way out -- where only the valid box is written, straight from the
// slot into the caller's (possibly strided) array.
#include "soda_internal.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>

#include <pthread.h>
#include <sched.h>
#include <unistd.h>

namespace soda_detail {

// ---- the GPU's NUMA node ------------------------------------------------------
// The pool's hosts have two sockets with four GPUs each.  Measured
// (tools/experiments/r05_numa_try.py, the headline step through host arrays):
// everything on the GPU's node 8.8-9.0 ms, everything on the other node
// 11.4-11.8 ms, left to the scheduler 8.3-13 ms from run to run.  What the
// library controls: where its pinned staging slots live (the DMA then stays
// on the GPU's socket) and where its worker threads run.  The node comes from
// sysfs (no libnuma in the image); SODA_HIP_HOST_NUMA=0 switches it off; any
// failure to find out leaves everything unbound.
struct NumaCpus {
  bool known = false;
  cpu_set_t cpus;
};

const NumaCpus& cpus_near_device(int device) {
  static std::mutex mu;
  static std::map<int, NumaCpus> memo;
  std::lock_guard<std::mutex> hold(mu);
  auto it = memo.find(device);
  if (it != memo.end()) return it->second;
  NumaCpus& out = memo[device];
  CPU_ZERO(&out.cpus);
  const char* env = getenv("SODA_HIP_HOST_NUMA");
  if (env && !strcmp(env, "0")) return out;
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, sizeof bus, device) != hipSuccess) {
    (void)hipGetLastError();
    return out;
  }
  for (char* c = bus; *c; ++c)
    if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
  char path[160];
  snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
  int node = -1;
  if (FILE* f = fopen(path, "r")) {
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
  }
  if (node < 0) return out;
  snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
  FILE* f = fopen(path, "r");
  if (!f) return out;
  char list[4096] = {0};
  const bool got = fgets(list, sizeof list, f) != nullptr;
  fclose(f);
  if (!got) return out;
  // "0-63,128-191": only CPUs this process may run on anyway
  cpu_set_t allowed;
  CPU_ZERO(&allowed);
  if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return out;
  int count = 0;
  for (char* tok = strtok(list, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
    int a = 0, b = 0;
    const int n = sscanf(tok, "%d-%d", &a, &b);
    if (n < 1) continue;
    if (n == 1) b = a;
    for (int c = a; c <= b && c < CPU_SETSIZE; ++c)
      if (CPU_ISSET(c, &allowed)) {
        CPU_SET(c, &out.cpus);
        ++count;
      }
  }
  out.known = count > 0;
  return out;
}

// the calling thread on the GPU's node for as long as the object lives: what
// it allocates (and the driver pins) in between lands in that node's memory
class RunNear {
 public:
  explicit RunNear(int device) {
    const NumaCpus& near = cpus_near_device(device);
    if (!near.known) return;
    if (sched_getaffinity(0, sizeof before_, &before_) != 0) return;
    moved_ = sched_setaffinity(0, sizeof near.cpus, &near.cpus) == 0;
  }
  ~RunNear() {
    if (moved_) (void)sched_setaffinity(0, sizeof before_, &before_);
  }

 private:
  cpu_set_t before_;
  bool moved_ = false;
};

// ---- worker threads -----------------------------------------------------------
// One process-wide pool, made on first use, never destroyed (its threads sleep
// on a condition variable between jobs and die with the process).  One job at
// a time: callers from several threads take turns.  A forked child finds the
// parent's pid and starts its own.
class CopyPool {
 public:
  static CopyPool& get() {
    static std::mutex make;
    static CopyPool* pool = nullptr;
    std::lock_guard<std::mutex> hold(make);
    if (!pool || pool->pid_ != getpid()) pool = new CopyPool;   // (leaked)
    return *pool;
  }

  // the workers onto the CPUs of the node of the GPU the FIRST host-array run
  // of the process talks to (a process that drives GPUs on both sockets keeps
  // that choice: its other GPUs' copies cross the socket link as they would
  // unbound half of the time)
  void run_near(int device) {
    std::lock_guard<std::mutex> hold(m_);
    if (bound_) return;
    bound_ = true;
    const NumaCpus& near = cpus_near_device(device);
    if (!near.known) return;
    for (pthread_t t : threads_)
      (void)pthread_setaffinity_np(t, sizeof near.cpus, &near.cpus);
  }

  int threads() const { return (int)workers_ + 1; }

  // job(i) for every i in [0, count), on the workers and the caller
  void run(size_t count, const std::function<void(size_t)>& job) {
    if (count == 0) return;
    if (count == 1 || workers_ == 0) {
      for (size_t i = 0; i < count; ++i) job(i);
      return;
    }
    std::lock_guard<std::mutex> turn(turn_);
    {
      std::lock_guard<std::mutex> hold(m_);
      job_ = &job;
      count_ = count;
      next_.store(0, std::memory_order_relaxed);
      active_ = workers_;
      ++generation_;
    }
    wake_.notify_all();
    work();
    std::unique_lock<std::mutex> hold(m_);
    done_.wait(hold, [this] { return active_ == 0; });
    job_ = nullptr;
  }

 private:
  CopyPool() : pid_(getpid()) {
    long want = 8;
    if (const char* v = getenv("SODA_HIP_HOST_THREADS")) want = atol(v);
    const long have = (long)std::thread::hardware_concurrency();
    if (have > 0 && want > have) want = have;
    if (want < 1) want = 1;
    workers_ = (size_t)want - 1;
    for (size_t i = 0; i < workers_; ++i) {
      std::thread t([this] { loop(); });
      threads_.push_back(t.native_handle());
      t.detach();
    }
  }

  void work() {
    for (;;) {
      const size_t i = next_.fetch_add(1, std::memory_order_relaxed);
      if (i >= count_) return;
      (*job_)(i);
    }
  }

  void loop() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> hold(m_);
        wake_.wait(hold, [&] { return generation_ != seen; });
        seen = generation_;
      }
      work();
      std::lock_guard<std::mutex> hold(m_);
      if (--active_ == 0) done_.notify_one();
    }
  }

  const pid_t pid_;
  size_t workers_ = 0;
  std::vector<pthread_t> threads_;
  bool bound_ = false;
  std::mutex turn_, m_;
  std::condition_variable wake_, done_;
  const std::function<void(size_t)>* job_ = nullptr;
  size_t count_ = 0, active_ = 0;
  uint64_t generation_ = 0;
  std::atomic<size_t> next_{0};
};

bool is_dense(const soda_hip_host_tensor_t& t, int dim) {
  int64_t s = 1;
  for (int d = 0; d < dim; ++d) {
    if (t.stride[d] != s) return false;
    s *= t.extent[d];
  }
  return true;
}

// Copies box [lo, hi) between a strided host array and a dense staging array
// that holds the cells from index `row0` of the LAST dimension on (row0 = 0:
// the whole array), with up to `threads` threads (0: the pool's).  A "line" is
// a run along dimension 0; long lines are cut into segments so that a 1-D
// array or a box of a few long rows still spreads over the threads.
void copy_rows(char* strided, const int32_t* stride, char* dense,
               const int32_t* extent, const int32_t* lo, const int32_t* hi,
               int dim, int elem, bool to_dense, int32_t row0, int threads) {
  int32_t l[SODA_HIP_MAX_DIM], h[SODA_HIP_MAX_DIM];
  int64_t dstride[SODA_HIP_MAX_DIM], sstride[SODA_HIP_MAX_DIM], s = 1;
  int64_t lines = 1;
  for (int d = 0; d < SODA_HIP_MAX_DIM; ++d) {
    l[d] = d < dim ? lo[d] : 0;
    h[d] = d < dim ? hi[d] : 1;
    dstride[d] = s;
    sstride[d] = d < dim ? stride[d] : 0;
    s *= d < dim ? extent[d] : 1;
    if (h[d] <= l[d]) return;
    if (d > 0) lines *= h[d] - l[d];
  }
  const int64_t dense_shift = dim > 0 ? (int64_t)row0 * dstride[dim - 1] : 0;
  const int64_t line_bytes = (int64_t)(h[0] - l[0]) * elem;
  const bool unit = sstride[0] == 1;
  const int64_t kSeg = 256 << 10;
  const int64_t segs = unit ? (line_bytes + kSeg - 1) / kSeg : 1;
  const int64_t pieces = lines * segs;
  // ~1 MiB of copying per task, at least one piece
  const int64_t piece_bytes = unit && line_bytes > kSeg ? kSeg : line_bytes;
  int64_t per_task = ((unit ? 1 << 20 : 1 << 18) + piece_bytes - 1) /
                     (piece_bytes > 0 ? piece_bytes : 1);
  if (per_task < 1) per_task = 1;
  const int64_t tasks = (pieces + per_task - 1) / per_task;
  const int32_t n1 = h[1] - l[1], n2 = h[2] - l[2];
  auto body = [&](size_t task) {
    const int64_t first = (int64_t)task * per_task;
    const int64_t last = first + per_task < pieces ? first + per_task : pieces;
    for (int64_t p = first; p < last; ++p) {
      const int64_t line = p / segs, seg = p % segs;
      const int32_t i1 = l[1] + (int32_t)(line % n1);
      const int32_t i2 = l[2] + (int32_t)((line / n1) % n2);
      const int32_t i3 = l[3] + (int32_t)(line / ((int64_t)n1 * n2));
      const int64_t so = i1 * sstride[1] + i2 * sstride[2] + i3 * sstride[3];
      const int64_t dof = i1 * dstride[1] + i2 * dstride[2] + i3 * dstride[3] -
                          dense_shift;
      if (unit) {
        const int64_t b0 = seg * kSeg;
        const int64_t b1 = b0 + kSeg < line_bytes ? b0 + kSeg : line_bytes;
        char* a = strided + (so + l[0]) * elem + b0;
        char* b = dense + (dof + l[0]) * elem + b0;
        if (to_dense) memcpy(b, a, (size_t)(b1 - b0));
        else memcpy(a, b, (size_t)(b1 - b0));
      } else {
        for (int32_t x = l[0]; x < h[0]; ++x) {
          char* a = strided + (so + (int64_t)x * sstride[0]) * elem;
          char* b = dense + (dof + x) * elem;
          if (to_dense) memcpy(b, a, elem);
          else memcpy(a, b, elem);
        }
      }
    }
  };
  const int64_t bytes = lines * line_bytes;
  if (threads == 1 || tasks <= 1 || bytes < (2 << 20)) {
    for (int64_t t = 0; t < tasks; ++t) body((size_t)t);
    return;
  }
  CopyPool::get().run((size_t)tasks, body);
}

void copy_box(char* strided, const int32_t* stride, char* dense,
              const int32_t* extent, const int32_t* lo, const int32_t* hi,
              int dim, int elem, bool to_dense) {
  copy_rows(strided, stride, dense, extent, lo, hi, dim, elem, to_dense, 0, 0);
}

// Stream elements [first, first + count) of a tensor dealt cyclically over
// `nb` banks -- element k in bank k % nb at index k / nb (reference
// docs/data-layout.md "Multi-Bank"; frt/host.py:241-246, 422-424) -- <-> a
// dense run that starts at `dense`.  first and count are multiples of nb.
// (the bank count as a template constant: the compiler turns the loops into
// unpack / shuffle sequences; with a run-time count they stay scalar, 1-6 GB/s
// a thread instead of 10-20)
template <typename T, int NB>
static void weave_fixed(T* const* banks, T* __restrict__ dense, int64_t j0,
                        int64_t n, bool to_dense) {
  T* __restrict__ bp[NB];
  for (int b = 0; b < NB; ++b) bp[b] = banks[b] + j0;
  if (to_dense) {
    for (int64_t j = 0; j < n; ++j)
      for (int b = 0; b < NB; ++b) dense[j * NB + b] = bp[b][j];
  } else {
    for (int64_t j = 0; j < n; ++j)
      for (int b = 0; b < NB; ++b) bp[b][j] = dense[j * NB + b];
  }
}

template <typename T>
static void weave_run(T* const* banks, int nb, T* dense, int64_t j0, int64_t j1,
                      int64_t jbase, bool to_dense) {
  T* d0 = dense + (j0 - jbase) * nb;
  switch (nb) {
    case 1: weave_fixed<T, 1>(banks, d0, j0, j1 - j0, to_dense); return;
    case 2: weave_fixed<T, 2>(banks, d0, j0, j1 - j0, to_dense); return;
    case 3: weave_fixed<T, 3>(banks, d0, j0, j1 - j0, to_dense); return;
    case 4: weave_fixed<T, 4>(banks, d0, j0, j1 - j0, to_dense); return;
    case 8: weave_fixed<T, 8>(banks, d0, j0, j1 - j0, to_dense); return;
    default: break;
  }
  if (to_dense) {
    for (int64_t j = j0; j < j1; ++j) {
      T* d = dense + (j - jbase) * nb;
      for (int b = 0; b < nb; ++b) d[b] = banks[b][j];
    }
  } else {
    for (int64_t j = j0; j < j1; ++j) {
      const T* d = dense + (j - jbase) * nb;
      for (int b = 0; b < nb; ++b) banks[b][j] = d[b];
    }
  }
}

void weave_banks(char* const* banks, int nb, char* dense, int64_t first,
                 int64_t count, int elem, bool to_dense, int threads) {
  const int64_t jbase = first / nb, groups = count / nb;
  if (groups < 1) return;
  // ~1 MiB of the dense side per task
  int64_t per_task = (1 << 20) / ((int64_t)nb * elem);
  if (per_task < 1) per_task = 1;
  const int64_t tasks = (groups + per_task - 1) / per_task;
  auto body = [&](size_t task) {
    const int64_t j0 = jbase + (int64_t)task * per_task;
    const int64_t j1 = j0 + per_task < jbase + groups ? j0 + per_task
                                                        : jbase + groups;
    switch (elem) {
      case 1: weave_run((uint8_t* const*)banks, nb, (uint8_t*)dense, j0, j1, jbase, to_dense); break;
      case 2: weave_run((uint16_t* const*)banks, nb, (uint16_t*)dense, j0, j1, jbase, to_dense); break;
      case 4: weave_run((uint32_t* const*)banks, nb, (uint32_t*)dense, j0, j1, jbase, to_dense); break;
      case 8: weave_run((uint64_t* const*)banks, nb, (uint64_t*)dense, j0, j1, jbase, to_dense); break;
      default:
        for (int64_t j = j0; j < j1; ++j)
          for (int b = 0; b < nb; ++b) {
            char* d = dense + ((j - jbase) * nb + b) * elem;
            char* s = banks[b] + j * elem;
            if (to_dense) memcpy(d, s, elem);
            else memcpy(s, d, elem);
          }
    }
  };
  if (threads == 1 || tasks <= 1 || count * elem < (2 << 20)) {
    for (int64_t t = 0; t < tasks; ++t) body((size_t)t);
    return;
  }
  CopyPool::get().run((size_t)tasks, body);
}

// ---- the ring of pinned staging slots -------------------------------------------
// Ranges pinned through soda_hip_host_register, start -> bytes.
struct Registered {
  std::mutex mu;
  std::map<uintptr_t, size_t> ranges;
};
Registered& registered() {
  static Registered* r = new Registered;     // (leaked: alive at exit)
  return *r;
}

// SODA_HIP_HOST_DIRECT=0: everything through the staging slots
bool direct_allowed() {
  const char* env = getenv("SODA_HIP_HOST_DIRECT");
  return !env || strcmp(env, "0");
}

// Is [ptr, ptr + bytes) host memory the DMA engine may be pointed at?  Yes for
// what lies inside ONE range registered through soda_hip_host_register.  What
// the runtime says about other memory (hipPointerGetAttributes: hipHostMalloc,
// somebody's hipHostRegister) is only taken on request
// (SODA_HIP_HOST_DIRECT=attributes): a registration the library did not make
// may be stale, partial or not writable, and a DMA into such a page is a GPU
// fault, not an error code.
bool host_pinned(const void* ptr, size_t bytes) {
  if (!ptr || !bytes) return false;
  {
    Registered& r = registered();
    std::lock_guard<std::mutex> hold(r.mu);
    const uintptr_t p = reinterpret_cast<uintptr_t>(ptr);
    auto it = r.ranges.upper_bound(p);
    if (it != r.ranges.begin()) {
      --it;
      if (p >= it->first && p + bytes <= it->first + it->second) return true;
    }
  }
  const char* env = getenv("SODA_HIP_HOST_DIRECT");
  if (!env || strcmp(env, "attributes")) return false;
  const char* ends[2] = {static_cast<const char*>(ptr),
                         static_cast<const char*>(ptr) + bytes - 1};
  for (const char* q : ends) {
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof at);
    if (hipPointerGetAttributes(&at, q) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    if (at.type != hipMemoryTypeHost) return false;
  }
  return true;
}

int HostRing::wait(int slot) {
  if (busy[slot]) {
    HIP_TRY(hipEventSynchronize(ev[slot]));
    busy[slot] = false;
  }
  return SODA_HIP_OK;
}

int HostRing::ensure(size_t want_slot_bytes, int want_slots) {
  if (base && slot_bytes >= want_slot_bytes && slots >= want_slots)
    return SODA_HIP_OK;
  for (int i = 0; i < slots; ++i)      // nothing in flight on memory we free
    if (int rc = wait(i)) return rc;
  release();
  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&base),
                        want_slot_bytes * want_slots, hipHostMallocDefault));
  slot_bytes = want_slot_bytes;
  slots = want_slots;
  for (int i = 0; i < slots; ++i)
    HIP_TRY(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
  return SODA_HIP_OK;
}

void HostRing::release() {
  for (int i = 0; i < kMaxSlots; ++i)
    if (ev[i]) {
      (void)hipEventDestroy(ev[i]);
      ev[i] = nullptr;
    }
  if (base) (void)hipHostFree(base);
  base = nullptr;
  slot_bytes = 0;
  slots = 0;
  for (bool& b : busy) b = false;
}

// A program's ring, big enough, its memory on the GPU's NUMA node (the thread
// that allocates runs there while it does) and the worker threads with it.
int ensure_ring(soda_hip_program* p, HostRing* ring, size_t slot_bytes,
                int slots) {
  CopyPool::get().run_near(p->device);
  if (ring->base && ring->slot_bytes >= slot_bytes && ring->slots >= slots)
    return SODA_HIP_OK;
  RunNear near(p->device);
  return ring->ensure(slot_bytes, slots);
}

namespace {

double now_ms() {
  return std::chrono::duration<double, std::milli>(
             std::chrono::steady_clock::now().time_since_epoch())
      .count();
}

size_t chunk_target_bytes() {
  // 16-32 MiB chunks sustain the DMA rate, 4 MiB halve it
  // (profiles/r05_hostcopy.jsonl).  _KB: tests that want many chunks of a
  // small array
  if (const char* v = getenv("SODA_HIP_HOST_CHUNK_KB"))
    if (atol(v) > 0) return (size_t)atol(v) << 10;
  long mb = 16;
  if (const char* v = getenv("SODA_HIP_HOST_CHUNK_MB")) mb = atol(v);
  if (mb < 1) mb = 1;
  return (size_t)mb << 20;
}

int make_events(std::vector<hipEvent_t>* pool, size_t count) {
  while (pool->size() < count) {
    hipEvent_t e = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    pool->push_back(e);
  }
  return SODA_HIP_OK;
}

// How box [lo, hi) of a dense array in pinned memory can be filled by the DMA
// engine: 1 = the box holds whole rows (plain copies), 2 = 2-D, a column range
// of the rows (one strided copy per run of rows), 3 = 3-D, cut in dimension
// 0 / 1 (one 3-D copy), 0 = not at all (dim > 3 with a cut box: the slots).
int direct_mode(const int32_t* extent, const int32_t* lo, const int32_t* hi,
                int dim) {
  bool whole_rows = true, whole_above_0 = true;
  for (int d = 0; d < dim - 1; ++d) {
    const bool full = lo[d] == 0 && hi[d] == extent[d];
    whole_rows = whole_rows && full;
    if (d > 0) whole_above_0 = whole_above_0 && full;
  }
  if (whole_rows) return 1;
  if (dim == 2 && whole_above_0) return 2;
  return dim == 3 ? 3 : 0;
}

// Rows [a, b) of the box, from a device array that is dense from row
// `dev_row0` on, straight into the caller's dense pinned array `host`.
int dma_rows_home(int mode, void* host, const void* dev, int64_t dev_row0,
                  const int32_t* extent, const int32_t* lo, const int32_t* hi,
                  int dim, int elem, int64_t a, int64_t b, hipStream_t stream) {
  int64_t plane = 1;
  for (int d = 0; d < dim - 1; ++d) plane *= extent[d];
  const char* src = static_cast<const char*>(dev) +
                    (size_t)(a - dev_row0) * plane * elem;
  if (mode == 1) {
    HIP_TRY(hipMemcpyAsync(static_cast<char*>(host) + (size_t)a * plane * elem,
                           src, (size_t)(b - a) * plane * elem,
                           hipMemcpyDeviceToHost, stream));
  } else if (mode == 2) {
    const size_t pitch = (size_t)plane * elem, x0 = (size_t)lo[0] * elem;
    HIP_TRY(hipMemcpy2DAsync(static_cast<char*>(host) + (size_t)a * pitch + x0,
                             pitch, src + x0, pitch,
                             (size_t)(hi[0] - lo[0]) * elem, (size_t)(b - a),
                             hipMemcpyDeviceToHost, stream));
  } else {
    const size_t pitch = (size_t)extent[0] * elem;
    hipMemcpy3DParms q;
    memset(&q, 0, sizeof q);
    q.srcPtr = make_hipPitchedPtr(
        const_cast<char*>(static_cast<const char*>(dev)), pitch, pitch,
        (size_t)extent[1]);
    q.srcPos = make_hipPos((size_t)lo[0] * elem, (size_t)lo[1],
                           (size_t)(a - dev_row0));
    q.dstPtr = make_hipPitchedPtr(host, pitch, pitch, (size_t)extent[1]);
    q.dstPos = make_hipPos((size_t)lo[0] * elem, (size_t)lo[1], (size_t)a);
    q.extent = make_hipExtent((size_t)(hi[0] - lo[0]) * elem,
                              (size_t)(hi[1] - lo[1]), (size_t)(b - a));
    q.kind = hipMemcpyDeviceToHost;
    HIP_TRY(hipMemcpy3DAsync(&q, stream));
  }
  return SODA_HIP_OK;
}

// What both ways of running share: the validated call, sizes, the rings.
struct HostCall {
  soda_hip_program* p;
  const soda_hip_host_tensor_t* inputs;
  const soda_hip_host_tensor_t* outputs;
  int32_t iterate;
  const int32_t* valid_lo;
  const int32_t* valid_hi;
  int dim, ax;
  const int32_t* extent;
  int32_t rows;            // extent of the last dimension
  int64_t cells, plane;    // plane: cells per index of the last dimension
  int64_t chunk_rows;      // rows per staged chunk
  int slots;
  int32_t zero[SODA_HIP_MAX_DIM] = {0, 0, 0, 0};
  std::vector<const void*> in_ptrs;     // device: inputs, then params
  int in_turn = 0;
  // Tensors the DMA engine can reach where they are: dense arrays in pinned
  // (hipHostMalloc'ed / registered) memory.  Their rows are neither packed nor
  // unpacked and use no staging slot; an output whose box does not hold whole
  // rows goes home by one strided copy per chunk (2-D: a column range of the
  // rows; 3-D: the box's part of the planes, 15.7 -> 13.9 ms for C4).
  std::vector<char> in_direct, out_direct;
  // Tensors dealt over DRAM banks (the wire format's streams on the host,
  // soda_hip_stream_run_host): per tensor, inputs then outputs, the number of
  // banks; where it is > 1 the tensor's `ptr` is the list of bank pointers and
  // the interleave happens in the pack / unpack step the host pays anyway.
  const int32_t* nbanks = nullptr;
  // ... and single-bank inputs the generated host DELAYED (produce_offset,
  // frt/host.py:241-246): dense element k is element k + lead[i] of the stream
  // of stream_elems elements, zero beyond it -- undone in the pack step too.
  const int32_t* lead = nullptr;
  int64_t stream_elems = 0;
  int64_t lead_in(int i) const { return lead ? lead[i] : 0; }
  int banks_in(int i) const { return nbanks ? nbanks[i] : 1; }
  int banks_out(int o) const { return nbanks ? nbanks[p->plan.num_inputs + o] : 1; }

  const int32_t* lo(int o) const { return valid_lo ? valid_lo + o * dim : zero; }
  const int32_t* hi(int o) const { return valid_hi ? valid_hi + o * dim : extent; }

  // is the slot the next send() of input i packs into free (its last DMA done)?
  bool can_send(int i) {
    if (in_direct[i]) return true;
    const int sl = in_turn % slots;
    if (!p->ring_in.busy[sl]) return true;
    if (hipEventQuery(p->ring_in.ev[sl]) == hipSuccess) {
      p->ring_in.busy[sl] = false;
      return true;
    }
    (void)hipGetLastError();       // (hipErrorNotReady is not an error)
    return false;
  }

  // rows [a, b) of input i: packed into a pinned slot, sent on `stream`
  int send(int i, int64_t a, int64_t b, hipStream_t stream) {
    const int elem = p->plan.elem_size[i];
    if (in_direct[i]) {
      const size_t off = (size_t)a * plane * elem;
      HIP_TRY(hipMemcpyAsync(static_cast<char*>(p->host_in[i].ptr) + off,
                             static_cast<const char*>(inputs[i].ptr) + off,
                             (size_t)(b - a) * plane * elem,
                             hipMemcpyHostToDevice, stream));
      return SODA_HIP_OK;
    }
    const int sl = in_turn++ % slots;
    if (int rc = p->ring_in.wait(sl)) return rc;
    char* slot = p->ring_in.base + (size_t)sl * p->ring_in.slot_bytes;
    int32_t l[SODA_HIP_MAX_DIM], h[SODA_HIP_MAX_DIM];
    for (int d = 0; d < SODA_HIP_MAX_DIM; ++d) {
      l[d] = 0;
      h[d] = d < dim ? extent[d] : 1;
    }
    l[ax] = (int32_t)a;
    h[ax] = (int32_t)b;
    if (lead_in(i) > 0) {
      const int64_t k0 = a * plane + lead_in(i), want = (b - a) * plane;
      int64_t have = stream_elems - k0;
      if (have > want) have = want;
      if (have < 0) have = 0;
      char* bank = static_cast<char*>(inputs[i].ptr);
      weave_banks(&bank, 1, slot, k0, have, elem, true, 0);
      memset(slot + (size_t)have * elem, 0, (size_t)(want - have) * elem);
    } else if (banks_in(i) > 1)
      weave_banks(static_cast<char* const*>(inputs[i].ptr), banks_in(i), slot,
                  a * plane, (b - a) * plane, elem, true, 0);
    else
      copy_rows(static_cast<char*>(inputs[i].ptr), inputs[i].stride, slot,
                extent, l, h, dim, elem, true, (int32_t)a, 0);
    HIP_TRY(hipMemcpyAsync(static_cast<char*>(p->host_in[i].ptr) +
                               (size_t)a * plane * elem,
                           slot, (size_t)(b - a) * plane * elem,
                           hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(p->ring_in.ev[sl], stream));
    p->ring_in.busy[sl] = true;
    return SODA_HIP_OK;
  }

  // rows [a, b) of output o, `dev` = where row `dev_row0` of it lives on the
  // device: fetched into slot `sl` on `stream`
  int fetch(int o, int64_t a, int64_t b, const void* dev, int64_t dev_row0,
            int sl, hipStream_t stream) {
    const int elem = p->plan.elem_size[p->plan.num_inputs + o];
    char* slot = p->ring_out.base + (size_t)sl * p->ring_out.slot_bytes;
    const char* src = static_cast<const char*>(dev) +
                      (size_t)(a - dev_row0) * plane * elem;
    if (out_direct[o]) {
      if (int rc = dma_rows_home(out_direct[o], outputs[o].ptr, dev, dev_row0,
                                 extent, lo(o), hi(o), dim, elem, a, b, stream))
        return rc;
    } else {
      HIP_TRY(hipMemcpyAsync(slot, src, (size_t)(b - a) * plane * elem,
                             hipMemcpyDeviceToHost, stream));
    }
    // (a direct chunk keeps its turn in the ring's bookkeeping, not its memory)
    HIP_TRY(hipEventRecord(p->ring_out.ev[sl], stream));
    p->ring_out.busy[sl] = true;
    return SODA_HIP_OK;
  }

  // the valid box's part of rows [a, b) of output o: slot -> caller's array
  // (frt/host.py:357-375: nothing outside the box is touched)
  void deliver(int o, int64_t a, int64_t b, int sl) {
    if (out_direct[o]) return;           // the DMA engine wrote it home
    const int elem = p->plan.elem_size[p->plan.num_inputs + o];
    char* slot = p->ring_out.base + (size_t)sl * p->ring_out.slot_bytes;
    int32_t l[SODA_HIP_MAX_DIM], h[SODA_HIP_MAX_DIM];
    for (int k = 0; k < SODA_HIP_MAX_DIM; ++k) {
      l[k] = k < dim ? lo(o)[k] : 0;
      h[k] = k < dim ? hi(o)[k] : 1;
    }
    l[ax] = (int32_t)a;
    h[ax] = (int32_t)b;
    if (banks_out(o) > 1)       // (whole rows: checked where the call starts)
      weave_banks(static_cast<char* const*>(outputs[o].ptr), banks_out(o), slot,
                  a * plane, (b - a) * plane, elem, false, 0);
    else
      copy_rows(static_cast<char*>(outputs[o].ptr), outputs[o].stride, slot,
                extent, l, h, dim, elem, false, (int32_t)a, 0);
  }

  bool empty(int o) const {
    for (int d = 0; d < dim; ++d)
      if (hi(o)[d] <= lo(o)[d]) return true;
    return false;
  }
};

// Copy in, run, copy out, on one stream; packing overlaps the DMA of the
// previous chunk, the DMA runs `slots` chunks ahead of the unpacking.
int run_whole(HostCall& c) {
  soda_hip_program* p = c.p;
  const soda_hip_plan_t& plan = p->plan;
  hipStream_t stream = p->hstream[0];
  for (int i = 0; i < plan.num_inputs; ++i)
    for (int64_t a = 0; a < c.rows; a += c.chunk_rows) {
      const int64_t b = a + c.chunk_rows < c.rows ? a + c.chunk_rows : c.rows;
      if (int rc = c.send(i, a, b, stream)) return rc;
    }
  std::vector<void*> out_ptrs(plan.num_outputs);
  for (int o = 0; o < plan.num_outputs; ++o) {
    size_t bytes = (size_t)c.cells * plan.elem_size[plan.num_inputs + o];
    if (int rc = ensure(p->host_out[o], bytes)) return rc;
    out_ptrs[o] = p->host_out[o].ptr;
  }
  if (int rc = soda_hip_run_device(p, out_ptrs.data(), c.in_ptrs.data(),
                                   c.extent, c.iterate, stream))
    return rc;
  for (int o = 0; o < plan.num_outputs; ++o) {
    if (c.empty(o)) continue;
    const int64_t r0 = c.lo(o)[c.ax], r1 = c.hi(o)[c.ax];
    const int64_t nchunk = (r1 - r0 + c.chunk_rows - 1) / c.chunk_rows;
    for (int64_t k = 0; k < nchunk + c.slots - 1; ++k) {
      if (k < nchunk) {
        const int64_t a = r0 + k * c.chunk_rows;
        const int64_t b = a + c.chunk_rows < r1 ? a + c.chunk_rows : r1;
        if (int rc = c.fetch(o, a, b, out_ptrs[o], 0, (int)(k % c.slots), stream))
          return rc;
      }
      const int64_t d = k - (c.slots - 1);       // the chunk to unpack now
      if (d >= 0) {
        const int64_t a = r0 + d * c.chunk_rows;
        const int64_t b = a + c.chunk_rows < r1 ? a + c.chunk_rows : r1;
        if (int rc = p->ring_out.wait((int)(d % c.slots))) return rc;
        c.deliver(o, a, b, (int)(d % c.slots));
      }
    }
  }
  HIP_TRY(hipStreamSynchronize(stream));   // (the param copies, an empty box)
  return SODA_HIP_OK;
}

// Bands along the last dimension, each a window run with iterate x reach ghost
// rows (the slab mechanism of a multi-GPU run, soda_hip_run_device_slab, with
// the PCIe link in the place of xGMI): three streams -- copies in, kernels,
// copies out -- so that the link is busy in both directions while band i
// computes.  One host thread drives all of it; whatever it finds to do next,
// in this order: deliver a fetched chunk, fetch one, launch a band whose
// input has been sent, send the next input rows if their slot is free, else
// look again in 20 us (it never blocks on one of the rings while the other has
// work for it: blocking sends made C2 10.3 instead of 9.0 ms).
int run_banded(HostCall& c, int64_t band_rows, int64_t g_lo, int64_t g_hi) {
  soda_hip_program* p = c.p;
  const soda_hip_plan_t& plan = p->plan;
  hipStream_t s_in = p->hstream[0], s_run = p->hstream[1], s_out = p->hstream[2];
  const int ax = c.ax;
  const int nb = (int)((c.rows + band_rows - 1) / band_rows);
  const int64_t groups = (c.rows + c.chunk_rows - 1) / c.chunk_rows;
  const int kBufs = 2;      // band output arrays per output, used in turn
  // events: one per input row group, then per band buffer "computed" / "free"
  if (int rc = make_events(&p->hevents, (size_t)groups + 2 * kBufs)) return rc;
  hipEvent_t* ev_rows = p->hevents.data();
  hipEvent_t* ev_run = ev_rows + groups;
  hipEvent_t* ev_free = ev_run + kBufs;
  if ((int)p->band_out.size() < kBufs * plan.num_outputs)
    p->band_out.resize(kBufs * plan.num_outputs);
  const int64_t buf_rows = band_rows + g_lo + g_hi;
  for (int o = 0; o < plan.num_outputs; ++o)
    for (int k = 0; k < kBufs; ++k)
      if (int rc = ensure(p->band_out[o * kBufs + k],
                          (size_t)buf_rows * c.plane *
                              plan.elem_size[plan.num_inputs + o]))
        return rc;

  struct OutChunk {
    int o, band;
    int64_t a, b, origin;
    bool first, last;      // of its band
  };
  std::vector<OutChunk> outq;
  std::vector<int64_t> band_last(nb, -1);   // index of a band's last out chunk
  int last_user[kBufs];                     // band that wrote a buffer last
  for (int k = 0; k < kBufs; ++k) last_user[k] = -1;
  int64_t gi = 0;          // next (input row group, input) unit to send
  int bl = 0;              // next band to launch
  size_t oi = 0, od = 0;   // out chunks: next to fetch, next to deliver

  auto launch = [&](int b) -> int {
    const int64_t r0 = (int64_t)b * band_rows;
    const int64_t r1 = r0 + band_rows < c.rows ? r0 + band_rows : c.rows;
    // rows of the band some output's valid box holds
    const size_t before = outq.size();
    for (int o = 0; o < plan.num_outputs; ++o) {
      if (c.empty(o)) continue;
      const int64_t a0 = r0 > c.lo(o)[ax] ? r0 : c.lo(o)[ax];
      const int64_t b0 = r1 < c.hi(o)[ax] ? r1 : c.hi(o)[ax];
      for (int64_t a = a0; a < b0; a += c.chunk_rows)
        outq.push_back({o, b, a, a + c.chunk_rows < b0 ? a + c.chunk_rows : b0,
                        0, false, false});
    }
    if (outq.size() == before) return SODA_HIP_OK;   // nothing to deliver
    const int64_t gl = g_lo < r0 ? g_lo : r0;
    const int64_t gh = g_hi < c.rows - r1 ? g_hi : c.rows - r1;
    const int64_t origin_row = r0 - gl;
    for (size_t k = before; k < outq.size(); ++k) outq[k].origin = origin_row;
    outq[before].first = true;
    outq.back().last = true;
    band_last[b] = (int64_t)outq.size() - 1;
    int32_t lext[SODA_HIP_MAX_DIM], origin[SODA_HIP_MAX_DIM],
        gext[SODA_HIP_MAX_DIM];
    for (int d = 0; d < SODA_HIP_MAX_DIM; ++d) {
      lext[d] = d < c.dim ? c.extent[d] : 1;
      gext[d] = lext[d];
      origin[d] = 0;
    }
    lext[ax] = (int32_t)(r1 - r0 + gl + gh);
    origin[ax] = (int32_t)origin_row;
    std::vector<const void*> ins(c.in_ptrs);
    for (int i = 0; i < plan.num_inputs; ++i)
      ins[i] = static_cast<const char*>(c.in_ptrs[i]) +
               (size_t)origin_row * c.plane * plan.elem_size[i];
    std::vector<void*> outs(plan.num_outputs);
    const int buf = b % kBufs;
    for (int o = 0; o < plan.num_outputs; ++o)
      outs[o] = p->band_out[o * kBufs + buf].ptr;
    const int64_t need = (r1 + gh + c.chunk_rows - 1) / c.chunk_rows;
    HIP_TRY(hipStreamWaitEvent(s_run, ev_rows[need - 1], 0));
    if (p->hbuf_used[buf]) HIP_TRY(hipStreamWaitEvent(s_run, ev_free[buf], 0));
    SlabRun slab;
    memset(&slab, 0, sizeof slab);
    slab.cone = {(int32_t)gl, (int32_t)(gl + r1 - r0), plan.reach_lo,
                 plan.reach_hi};
    // (scheduled by the model: a calibration per band extent would cost more
    // than the bands save, and synchronise in the middle of the pipeline)
    const bool was = p->calibrating;
    p->calibrating = true;
    const int rc = run_core(p, outs.data(), ins.data(), lext, origin, gext,
                            c.iterate, s_run, -1, &slab);
    p->calibrating = was;
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ev_run[buf], s_run));
    p->hbuf_used[buf] = true;
    return SODA_HIP_OK;
  };

  for (int k = 0; k < kBufs; ++k) p->hbuf_used[k] = false;
  // SODA_HIP_HOST_TRACE=1: where the host thread's time went, on stderr
  const bool trace = getenv("SODA_HIP_HOST_TRACE") != nullptr;
  double t_begin = now_ms(), t_send = 0, t_deliver = 0, t_launch = 0, t_wait = 0;
  double t0 = 0;
  for (;;) {
    if (bl == nb && od == outq.size()) break;
    // 1. a fetched chunk is there: write its part of the box
    if (od < oi &&
        hipEventQuery(p->ring_out.ev[od % c.slots]) == hipSuccess) {
      p->ring_out.busy[od % c.slots] = false;
      const OutChunk& q = outq[od];
      t0 = now_ms();
      c.deliver(q.o, q.a, q.b, (int)(od % c.slots));
      t_deliver += now_ms() - t0;
      ++od;
      continue;
    }
    (void)hipGetLastError();       // (hipErrorNotReady is not an error)
    // 2. a slot is free: fetch the next chunk of a launched band
    if (oi < outq.size() && oi - od < (size_t)c.slots) {
      const OutChunk& q = outq[oi];
      const int buf = q.band % kBufs;
      if (q.first) HIP_TRY(hipStreamWaitEvent(s_out, ev_run[buf], 0));
      if (int rc = c.fetch(q.o, q.a, q.b, p->band_out[q.o * kBufs + buf].ptr,
                           q.origin, (int)(oi % c.slots), s_out))
        return rc;
      if (q.last) HIP_TRY(hipEventRecord(ev_free[buf], s_out));
      ++oi;
      continue;
    }
    // 3. the next band's input rows have been sent and its output array is
    //    free (every fetch from its previous user enqueued): launch it
    if (bl < nb) {
      const int64_t r1 = (int64_t)(bl + 1) * band_rows < c.rows
                             ? (int64_t)(bl + 1) * band_rows : c.rows;
      const int64_t hi_row = r1 + g_hi < c.rows ? r1 + g_hi : c.rows;
      const int64_t need = (hi_row + c.chunk_rows - 1) / c.chunk_rows;
      const int user = last_user[bl % kBufs];
      const bool buffer_free = user < 0 || band_last[user] < (int64_t)oi;
      if (gi >= need * plan.num_inputs && buffer_free) {
        t0 = now_ms();
        if (int rc = launch(bl)) return rc;
        t_launch += now_ms() - t0;
        if (band_last[bl] >= 0) last_user[bl % kBufs] = bl;
        ++bl;
        continue;
      }
    }
    // 4. send the next rows of the next input, if their slot is free (gi
    //    counts (row group, input) units)
    if (gi < groups * plan.num_inputs &&
        c.can_send((int)(gi % plan.num_inputs))) {
      const int64_t g = gi / plan.num_inputs;
      const int i = (int)(gi % plan.num_inputs);
      const int64_t a = g * c.chunk_rows;
      const int64_t b = a + c.chunk_rows < c.rows ? a + c.chunk_rows : c.rows;
      t0 = now_ms();
      if (int rc = c.send(i, a, b, s_in)) return rc;
      if (i == plan.num_inputs - 1) HIP_TRY(hipEventRecord(ev_rows[g], s_in));
      t_send += now_ms() - t0;
      ++gi;
      continue;
    }
    // 5. everything the host could do waits for a DMA (a slot of the input
    //    ring, a fetch): look again in a moment
    if (od < oi || gi < groups * plan.num_inputs) {
      t0 = now_ms();
      std::this_thread::sleep_for(std::chrono::microseconds(20));
      t_wait += now_ms() - t0;
      continue;
    }
    return fail(SODA_HIP_ERR_RUNTIME, "run_host: the band pipeline stalled");
  }
  if (trace)
    fprintf(stderr, "soda_hip_run_host_box: %d bands of %lld rows, %lld-row "
            "chunks: %.2f ms = send %.2f (pack + enqueue) + deliver %.2f + "
            "launch %.2f + wait %.2f + other\n", nb, (long long)band_rows,
            (long long)c.chunk_rows, now_ms() - t_begin, t_send, t_deliver,
            t_launch, t_wait);
  HIP_TRY(hipStreamSynchronize(s_in));
  HIP_TRY(hipStreamSynchronize(s_run));
  HIP_TRY(hipStreamSynchronize(s_out));
  return SODA_HIP_OK;
}

// Rows per band by a timeline estimate, 0 = do not band, -1 = the plan has no
// time model.  Three resources in sequence per band -- the link in (rows
// arrive in order), the kernels (a band waits for its last ghost row and for
// the band before it), the link out -- at the rates measured on the pool's
// hosts (profiles/r05_hostcopy.jsonl: 57 GB/s one way alone, 48 each with both
// busy; the host threads pack and deliver 60-85 GB/s in total), kernel time from
// the library's own pass times (calibrated if this extent was, else the
// model), inflated 15 % for the thinner grids of bands plus 20 us per launch.
// Bands must win by 10 % to be chosen.
int64_t choose_bands(HostCall& c, int64_t g_lo, int64_t g_hi) {
  soda_hip_program* p = c.p;
  const soda_hip_plan_t& plan = p->plan;
  int32_t ext[SODA_HIP_MAX_DIM];
  ExtentKey key;
  for (int d = 0; d < SODA_HIP_MAX_DIM; ++d) {
    ext[d] = d < c.dim ? c.extent[d] : 1;
    key[d] = ext[d];
  }
  const ExtentPlan* ep = nullptr;
  if (extent_plan(plan, &p->extents, ext, &ep)) return -1;
  auto it = p->measured.find(key);
  const std::vector<double>& pass_ns =
      it != p->measured.end() ? it->second : ep->model_ns;
  for (int i = 0; i < plan.num_passes; ++i)
    if (!(pass_ns[i] > 0)) return -1;
  int32_t count[SODA_HIP_MAX_PASSES], total = 0;
  if (schedule(plan, pass_ns, c.iterate, count, &total)) return -1;
  double compute_ns = 0;
  for (int i = 0; i < plan.num_passes; ++i) compute_ns += count[i] * pass_ns[i];
  double in_row = 0, out_row = 0;        // bytes per index of the last dimension
  double in_packed = 0, out_packed = 0;  // ... of them through the host threads
  for (int i = 0; i < plan.num_inputs; ++i) {
    in_row += (double)c.plane * plan.elem_size[i];
    if (!c.in_direct[i]) in_packed += (double)c.plane * plan.elem_size[i];
  }
  int64_t out_lo = c.rows, out_hi = 0;   // rows some valid box holds
  for (int o = 0; o < plan.num_outputs; ++o) {
    if (c.empty(o)) continue;
    out_row += (double)c.plane * plan.elem_size[plan.num_inputs + o];
    if (!c.out_direct[o])
      out_packed += (double)c.plane * plan.elem_size[plan.num_inputs + o];
    if (c.lo(o)[c.ax] < out_lo) out_lo = c.lo(o)[c.ax];
    if (c.hi(o)[c.ax] > out_hi) out_hi = c.hi(o)[c.ax];
  }
  if (out_hi <= out_lo) return 0;
  const double kAlone = 57.0, kBoth = 48.0, kHost = 85.0;   // bytes per ns
  const double whole = c.rows * in_row / kAlone + compute_ns +
                       (out_hi - out_lo) * out_row / kAlone;
  const double host_floor =
      (c.rows * in_packed + (out_hi - out_lo) * out_packed) / kHost;
  double best = whole * 0.9;
  int64_t best_rows = 0;
  for (int64_t per_band = 1; per_band * c.chunk_rows < c.rows; ++per_band) {
    const int64_t band = per_band * c.chunk_rows;
    const int64_t nb = (c.rows + band - 1) / band;
    if (nb < 2) break;
    if (nb > 64) continue;               // (thinner bands never won)
    double run_end = 0, out_end = 0;
    for (int64_t b = 0; b < nb; ++b) {
      const int64_t r0 = b * band, r1 = r0 + band < c.rows ? r0 + band : c.rows;
      const int64_t gl = g_lo < r0 ? g_lo : r0;
      const int64_t gh = g_hi < c.rows - r1 ? g_hi : c.rows - r1;
      // (whole chunks arrive: the band's last row rounded up to a chunk)
      int64_t need = (r1 + gh + c.chunk_rows - 1) / c.chunk_rows * c.chunk_rows;
      if (need > c.rows) need = c.rows;
      const double arrived = need * in_row / kBoth;
      const double run = compute_ns * (double)(r1 - r0 + gl + gh) / c.rows * 1.15 +
                         20000.0 * total;
      const double start = arrived > run_end ? arrived : run_end;
      run_end = start + run;
      const int64_t a = r0 > out_lo ? r0 : out_lo, e = r1 < out_hi ? r1 : out_hi;
      if (e > a) {
        const double s0 = run_end > out_end ? run_end : out_end;
        out_end = s0 + (e - a) * out_row / kBoth;
      }
    }
    double t = out_end > run_end ? out_end : run_end;
    if (t < host_floor) t = host_floor;
    if (t < best) {
      best = t;
      best_rows = band;
    }
  }
  if (getenv("SODA_HIP_HOST_TRACE"))
    fprintf(stderr, "soda_hip_run_host_box: estimate whole %.2f ms (kernels "
            "%.2f), best bands of %lld rows %.2f ms\n", whole / 1e6,
            compute_ns / 1e6, (long long)best_rows,
            best_rows ? best / 1e6 : 0.0);
  return best_rows;
}

}  // namespace

// Contiguous host bytes -> device through the program's input ring: chunk i + 1
// is copied into its pinned slot by the worker threads while chunk i is on the
// link.  Returns with the last DMAs in flight on `stream`.
int ring_send(soda_hip_program* p, void* dev, const void* host, size_t bytes,
              hipStream_t stream) {
  if (!bytes) return SODA_HIP_OK;
  const size_t target = chunk_target_bytes();
  const size_t chunk = bytes < target ? bytes : target;
  const int slots = chunk < bytes ? HostRing::kMaxSlots : 1;
  if (int rc = ensure_ring(p, &p->ring_in, chunk, slots)) return rc;
  const int32_t one = 1;
  int turn = 0;
  for (size_t off = 0; off < bytes; off += chunk, ++turn) {
    const size_t n = off + chunk < bytes ? chunk : bytes - off;
    const int sl = turn % slots;
    if (int rc = p->ring_in.wait(sl)) return rc;   // (also a previous call's DMA)
    char* slot = p->ring_in.base + (size_t)sl * p->ring_in.slot_bytes;
    const int32_t ext = (int32_t)n, lo = 0;
    copy_rows(const_cast<char*>(static_cast<const char*>(host)) + off, &one,
              slot, &ext, &lo, &ext, 1, 1, true, 0, 0);
    HIP_TRY(hipMemcpyAsync(static_cast<char*>(dev) + off, slot, n,
                           hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(p->ring_in.ev[sl], stream));
    p->ring_in.busy[sl] = true;
  }
  return SODA_HIP_OK;
}

// Device bytes -> contiguous host memory through the output ring; returns
// when every byte has been delivered.
int ring_fetch(soda_hip_program* p, void* host, const void* dev, size_t bytes,
               hipStream_t stream) {
  if (!bytes) return SODA_HIP_OK;
  const size_t target = chunk_target_bytes();
  const size_t chunk = bytes < target ? bytes : target;
  const int slots = chunk < bytes ? HostRing::kMaxSlots : 1;
  if (int rc = ensure_ring(p, &p->ring_out, chunk, slots)) return rc;
  const int32_t one = 1;
  const int64_t nchunk = (int64_t)((bytes + chunk - 1) / chunk);
  for (int64_t c = 0; c < nchunk + slots - 1; ++c) {
    if (c < nchunk) {
      const size_t off = (size_t)c * chunk;
      const size_t n = off + chunk < bytes ? chunk : bytes - off;
      const int sl = (int)(c % slots);
      HIP_TRY(hipMemcpyAsync(p->ring_out.base + (size_t)sl * p->ring_out.slot_bytes,
                             static_cast<const char*>(dev) + off, n,
                             hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipEventRecord(p->ring_out.ev[sl], stream));
      p->ring_out.busy[sl] = true;
    }
    const int64_t d = c - (slots - 1);
    if (d >= 0) {
      const size_t off = (size_t)d * chunk;
      const size_t n = off + chunk < bytes ? chunk : bytes - off;
      const int sl = (int)(d % slots);
      if (int rc = p->ring_out.wait(sl)) return rc;
      const int32_t ext = (int32_t)n, lo = 0;
      copy_rows(static_cast<char*>(host) + off, &one,
                p->ring_out.base + (size_t)sl * p->ring_out.slot_bytes, &ext,
                &lo, &ext, 1, 1, false, 0, 0);
    }
  }
  return SODA_HIP_OK;
}

namespace {

int64_t rows_per_chunk(int64_t plane_bytes, int64_t rows) {
  int64_t n = (int64_t)(chunk_target_bytes() / (size_t)(plane_bytes > 0 ? plane_bytes : 1));
  if (n < 1) n = 1;
  return n > rows ? rows : n;
}

}  // namespace

// Rows [a, b) (last dimension) of a host tensor of `extent`, dense or strided
// -> device memory holding the dense array from row `dev_row0` on: packed chunk
// by chunk into p's input ring by the worker threads while the previous chunk
// is on the link.  Returns with the last DMAs in flight on `stream`.
int send_rows(soda_hip_program* p, const soda_hip_host_tensor_t& t,
              const int32_t* extent, int dim, int elem, int64_t a, int64_t b,
              void* dev, int64_t dev_row0, hipStream_t stream) {
  if (b <= a) return SODA_HIP_OK;
  const int ax = dim - 1;
  int64_t plane = 1;
  for (int d = 0; d < ax; ++d) plane *= extent[d];
  // a dense array in memory pinned with soda_hip_host_register: its rows go by
  // DMA from where they are (one copy; N slabs on N GPUs: N links at once,
  // no host thread in between)
  if (direct_allowed() && is_dense(t, dim) &&
      host_pinned(t.ptr, (size_t)plane * extent[ax] * elem)) {
    HIP_TRY(hipMemcpyAsync(static_cast<char*>(dev) +
                               (size_t)(a - dev_row0) * plane * elem,
                           static_cast<const char*>(t.ptr) +
                               (size_t)a * plane * elem,
                           (size_t)(b - a) * plane * elem,
                           hipMemcpyHostToDevice, stream));
    return SODA_HIP_OK;
  }
  const int64_t step = rows_per_chunk(plane * elem, b - a);
  const int slots = step < b - a ? HostRing::kMaxSlots : 1;
  if (int rc = ensure_ring(p, &p->ring_in, (size_t)step * plane * elem, slots))
    return rc;
  int turn = 0;
  for (int64_t r = a; r < b; r += step, ++turn) {
    const int64_t e = r + step < b ? r + step : b;
    const int sl = turn % slots;
    if (int rc = p->ring_in.wait(sl)) return rc;
    char* slot = p->ring_in.base + (size_t)sl * p->ring_in.slot_bytes;
    int32_t l[SODA_HIP_MAX_DIM], h[SODA_HIP_MAX_DIM];
    for (int d = 0; d < SODA_HIP_MAX_DIM; ++d) {
      l[d] = 0;
      h[d] = d < dim ? extent[d] : 1;
    }
    l[ax] = (int32_t)r;
    h[ax] = (int32_t)e;
    copy_rows(static_cast<char*>(t.ptr), t.stride, slot, extent, l, h, dim, elem,
              true, (int32_t)r, 0);
    HIP_TRY(hipMemcpyAsync(static_cast<char*>(dev) +
                               (size_t)(r - dev_row0) * plane * elem,
                           slot, (size_t)(e - r) * plane * elem,
                           hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(p->ring_in.ev[sl], stream));
    p->ring_in.busy[sl] = true;
  }
  return SODA_HIP_OK;
}

// The part of box [lo, hi) that lies in rows [a, b): device array (dense, from
// row `dev_row0` on) -> the host tensor, nothing outside the box written; the
// DMA runs up to four chunks ahead of the threads that unpack.  Returns when
// everything is delivered.
int fetch_rows(soda_hip_program* p, const soda_hip_host_tensor_t& t,
               const int32_t* extent, const int32_t* lo, const int32_t* hi,
               int dim, int elem, int64_t a, int64_t b, const void* dev,
               int64_t dev_row0, hipStream_t stream) {
  const int ax = dim - 1;
  if (a < lo[ax]) a = lo[ax];
  if (b > hi[ax]) b = hi[ax];
  for (int d = 0; d < dim; ++d)
    if (hi[d] <= lo[d]) return SODA_HIP_OK;
  if (b <= a) return SODA_HIP_OK;
  int64_t plane = 1;
  for (int d = 0; d < ax; ++d) plane *= extent[d];
  // (as send_rows; this one returns with the copy IN FLIGHT on `stream` -- the
  // caller synchronises the streams it used)
  if (direct_allowed() && is_dense(t, dim) &&
      host_pinned(t.ptr, (size_t)plane * extent[ax] * elem)) {
    if (const int mode = direct_mode(extent, lo, hi, dim))
      return dma_rows_home(mode, t.ptr, dev, dev_row0, extent, lo, hi, dim, elem,
                           a, b, stream);
  }
  const int64_t step = rows_per_chunk(plane * elem, b - a);
  const int slots = step < b - a ? HostRing::kMaxSlots : 1;
  if (int rc = ensure_ring(p, &p->ring_out, (size_t)step * plane * elem, slots))
    return rc;
  const int64_t nchunk = (b - a + step - 1) / step;
  for (int64_t c = 0; c < nchunk + slots - 1; ++c) {
    if (c < nchunk) {
      const int64_t r = a + c * step, e = r + step < b ? r + step : b;
      const int sl = (int)(c % slots);
      HIP_TRY(hipMemcpyAsync(p->ring_out.base + (size_t)sl * p->ring_out.slot_bytes,
                             static_cast<const char*>(dev) +
                                 (size_t)(r - dev_row0) * plane * elem,
                             (size_t)(e - r) * plane * elem,
                             hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipEventRecord(p->ring_out.ev[sl], stream));
      p->ring_out.busy[sl] = true;
    }
    const int64_t d = c - (slots - 1);
    if (d >= 0) {
      const int64_t r = a + d * step, e = r + step < b ? r + step : b;
      const int sl = (int)(d % slots);
      if (int rc = p->ring_out.wait(sl)) return rc;
      int32_t l[SODA_HIP_MAX_DIM], h[SODA_HIP_MAX_DIM];
      for (int k = 0; k < SODA_HIP_MAX_DIM; ++k) {
        l[k] = k < dim ? lo[k] : 0;
        h[k] = k < dim ? hi[k] : 1;
      }
      l[ax] = (int32_t)r;
      h[ax] = (int32_t)e;
      copy_rows(static_cast<char*>(t.ptr), t.stride,
                p->ring_out.base + (size_t)sl * p->ring_out.slot_bytes, extent,
                l, h, dim, elem, false, (int32_t)r, 0);
    }
  }
  return SODA_HIP_OK;
}

int host_stream(soda_hip_program* p, hipStream_t* stream) {
  HIP_TRY(hipSetDevice(p->device));
  if (!p->hstream[0])
    HIP_TRY(hipStreamCreateWithFlags(&p->hstream[0], hipStreamNonBlocking));
  *stream = p->hstream[0];
  return SODA_HIP_OK;
}

}  // namespace soda_detail

using namespace soda_detail;

extern "C" {

int soda_hip_host_copy_box(void* strided, const int32_t* stride, void* dense,
                           const int32_t* extent, const int32_t* lo,
                           const int32_t* hi, int32_t dim, int32_t elem,
                           int32_t to_dense, int32_t row0, int32_t threads) {
  if (!strided || !stride || !dense || !extent || !lo || !hi || dim < 1 ||
      dim > SODA_HIP_MAX_DIM || elem < 1 || row0 < 0 || threads < 0)
    return fail(SODA_HIP_ERR_INVALID, "host_copy_box: bad argument");
  for (int d = 0; d < dim; ++d)
    if (extent[d] < 1 || lo[d] < 0 || hi[d] > extent[d])
      return fail(SODA_HIP_ERR_INVALID, "host_copy_box: box outside the array");
  if (lo[dim - 1] < row0 && lo[dim - 1] < hi[dim - 1])
    return fail(SODA_HIP_ERR_INVALID, "host_copy_box: box starts below row0");
  copy_rows(static_cast<char*>(strided), stride, static_cast<char*>(dense),
            extent, lo, hi, dim, elem, to_dense != 0, row0, threads);
  return SODA_HIP_OK;
}

int soda_hip_host_register(void* ptr, size_t bytes) {
  if (!ptr || !bytes) return fail(SODA_HIP_ERR_INVALID, "host_register: nothing");
  // Whole pages of the caller's own: a range that starts inside a heap page
  // shares it with whatever malloc puts next to it, and registrations coming
  // and going over such pages beside the runtime's own pinning of pageable
  // copies ended in GPU memory faults (tools/experiments/r05_host_soak.py).
  if (reinterpret_cast<uintptr_t>(ptr) & 4095)
    return fail(SODA_HIP_ERR_INVALID,
                "host_register: the range must start on a page boundary "
                "(aligned_alloc(4096, ...) as the reference host allocates, "
                "frt/host.py:165-178; mmap)");
  HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  Registered& r = registered();
  std::lock_guard<std::mutex> hold(r.mu);
  r.ranges[reinterpret_cast<uintptr_t>(ptr)] = bytes;
  return SODA_HIP_OK;
}

int soda_hip_host_unregister(void* ptr) {
  if (!ptr) return SODA_HIP_OK;
  {
    Registered& r = registered();
    std::lock_guard<std::mutex> hold(r.mu);
    if (!r.ranges.erase(reinterpret_cast<uintptr_t>(ptr)))
      return fail(SODA_HIP_ERR_INVALID,
                  "host_unregister: not a range soda_hip_host_register pinned");
  }
  HIP_TRY(hipHostUnregister(ptr));
  return SODA_HIP_OK;
}

int soda_hip_host_weave_banks(void* const* banks, int32_t num_banks, void* dense,
                              int64_t first, int64_t count, int32_t elem,
                              int32_t to_dense, int32_t threads) {
  if (!banks || !dense || num_banks < 1 || elem < 1 || first < 0 || count < 0 ||
      first % num_banks || count % num_banks || threads < 0)
    return fail(SODA_HIP_ERR_INVALID, "host_weave_banks: bad argument");
  for (int b = 0; b < num_banks; ++b)
    if (!banks[b]) return fail(SODA_HIP_ERR_INVALID, "host_weave_banks: NULL bank");
  weave_banks(reinterpret_cast<char* const*>(banks), num_banks,
              static_cast<char*>(dense), first, count, elem, to_dense != 0,
              threads);
  return SODA_HIP_OK;
}

int soda_hip_run_host_box(soda_hip_program_t* p,
                          const soda_hip_host_tensor_t* inputs,
                          const soda_hip_host_tensor_t* outputs,
                          int32_t iterate, const int32_t* valid_lo,
                          const int32_t* valid_hi) {
  return run_host_call(p, inputs, outputs, iterate, valid_lo, valid_hi, nullptr,
                       nullptr, 0);
}

}  // extern "C"

namespace soda_detail {

int run_host_call(soda_hip_program* p, const soda_hip_host_tensor_t* inputs,
                  const soda_hip_host_tensor_t* outputs, int32_t iterate,
                  const int32_t* valid_lo, const int32_t* valid_hi,
                  const int32_t* nbanks, const int32_t* lead,
                  int64_t stream_elems) {
  if (!p || !inputs || !outputs)
    return fail(SODA_HIP_ERR_INVALID, "run_host: NULL argument");
  const soda_hip_plan_t& plan = p->plan;
  if (iterate < 1) return fail(SODA_HIP_ERR_INVALID, "cannot iterate < 1 times");
  HostCall c;
  c.nbanks = nbanks;
  c.lead = lead;
  c.stream_elems = stream_elems;
  c.p = p;
  c.inputs = inputs;
  c.outputs = outputs;
  c.iterate = iterate;
  c.valid_lo = valid_lo;
  c.valid_hi = valid_hi;
  c.dim = plan.dim;
  c.ax = plan.dim - 1;
  c.extent = inputs[0].extent;
  if (!c.extent) return fail(SODA_HIP_ERR_INVALID, "run_host: NULL extent");
  c.cells = 1;
  for (int d = 0; d < c.dim; ++d) {
    if (c.extent[d] < 1)
      return fail(SODA_HIP_ERR_INVALID, "run_host: extent < 1");
    c.cells *= c.extent[d];
  }
  auto same_extent = [&](const soda_hip_host_tensor_t& t) {
    if (!t.ptr || !t.extent || !t.stride) return false;
    for (int d = 0; d < c.dim; ++d)
      if (t.extent[d] != c.extent[d]) return false;
    return true;
  };
  for (int i = 0; i < plan.num_inputs; ++i)
    if (!same_extent(inputs[i]))
      return fail(SODA_HIP_ERR_INVALID, "run_host: bad input tensor");
  for (int o = 0; o < plan.num_outputs; ++o) {
    if (!same_extent(outputs[o]))
      return fail(SODA_HIP_ERR_INVALID, "run_host: bad output tensor");
    for (int d = 0; d < c.dim; ++d)
      if (c.lo(o)[d] < 0 || c.hi(o)[d] > c.extent[d])
        return fail(SODA_HIP_ERR_INVALID, "run_host: box outside the array");
  }
  c.rows = c.extent[c.ax];
  c.plane = c.cells / c.rows;
  for (int t = 0; nbanks && t < plan.num_inputs + plan.num_outputs; ++t) {
    if (nbanks[t] < 1 || c.plane % nbanks[t])
      return fail(SODA_HIP_ERR_INVALID, "run_host: rows do not divide over the banks");
    if (nbanks[t] > 1 && t >= plan.num_inputs && (valid_lo || valid_hi))
      return fail(SODA_HIP_ERR_INVALID, "run_host: a banked output has no box");
  }
  for (int i = 0; lead && i < plan.num_inputs; ++i)
    if (lead[i] < 0 || (lead[i] > 0 && (c.banks_in(i) != 1 || stream_elems < 1)))
      return fail(SODA_HIP_ERR_INVALID, "run_host: bad input delay");
  HIP_TRY(hipSetDevice(p->device));
  for (int k = 0; k < 3; ++k)
    if (!p->hstream[k])
      HIP_TRY(hipStreamCreateWithFlags(&p->hstream[k], hipStreamNonBlocking));
  // dense arrays in pinned memory go by DMA from / to where they are
  // (SODA_HIP_HOST_DIRECT=0: everything through the staging slots)
  c.in_direct.assign(plan.num_inputs, 0);
  c.out_direct.assign(plan.num_outputs, 0);
  if (direct_allowed()) {
    for (int i = 0; i < plan.num_inputs; ++i)
      c.in_direct[i] =
          c.banks_in(i) == 1 && c.lead_in(i) == 0 && is_dense(inputs[i], c.dim) &&
          host_pinned(inputs[i].ptr, (size_t)c.cells * plan.elem_size[i]);
    for (int o = 0; o < plan.num_outputs; ++o) {
      const int elem = plan.elem_size[plan.num_inputs + o];
      if (c.banks_out(o) > 1 || c.empty(o) || !is_dense(outputs[o], c.dim) ||
          !host_pinned(outputs[o].ptr, (size_t)c.cells * elem))
        continue;
      c.out_direct[o] = (char)direct_mode(c.extent, c.lo(o), c.hi(o), c.dim);
    }
  }
  if (getenv("SODA_HIP_HOST_TRACE")) {
    int ni = 0, no = 0;
    for (char f : c.in_direct) ni += f != 0;
    for (char f : c.out_direct) no += f != 0;
    fprintf(stderr, "soda_hip_run_host_box: by DMA in place: %d of %d inputs, "
            "%d of %d outputs\n", ni, plan.num_inputs, no, plan.num_outputs);
  }
  int max_elem = 1;
  for (int t = 0; t < plan.num_inputs + plan.num_outputs; ++t)
    if (plan.elem_size[t] > max_elem) max_elem = plan.elem_size[t];
  // rows per staged chunk: ~16 MiB of the widest tensor, the whole array if
  // it is smaller
  c.chunk_rows =
      (int64_t)(chunk_target_bytes() / ((size_t)c.plane * max_elem));
  if (c.chunk_rows < 1) c.chunk_rows = 1;
  if (c.chunk_rows > c.rows) c.chunk_rows = c.rows;
  c.slots = c.chunk_rows < c.rows ? HostRing::kMaxSlots : 1;
  const size_t slot_bytes = (size_t)c.chunk_rows * c.plane * max_elem;
  if (int rc = ensure_ring(p, &p->ring_in, slot_bytes, c.slots)) return rc;
  if (int rc = ensure_ring(p, &p->ring_out, slot_bytes, c.slots)) return rc;
  c.in_ptrs.resize(plan.num_inputs);
  for (int i = 0; i < plan.num_inputs; ++i) {
    if (int rc = ensure(p->host_in[i], (size_t)c.cells * plan.elem_size[i]))
      return rc;
    c.in_ptrs[i] = p->host_in[i].ptr;
  }
  const int prm0 = plan.num_inputs + plan.num_outputs + plan.num_locals;
  for (int k = 0; k < plan.num_params; ++k) {
    const soda_hip_host_tensor_t& t = inputs[plan.num_inputs + k];
    if (!t.ptr) return fail(SODA_HIP_ERR_INVALID, "run_host: NULL param");
    size_t bytes = (size_t)plan.param_elems[k] * plan.elem_size[prm0 + k];
    if (int rc = ensure(p->host_prm[k], bytes)) return rc;
    // (a few hundred bytes, pageable: staged by the time the call returns;
    // first on the stream the input rows follow on)
    HIP_TRY(hipMemcpyAsync(p->host_prm[k].ptr, t.ptr, bytes,
                           hipMemcpyHostToDevice, p->hstream[0]));
    c.in_ptrs.push_back(p->host_prm[k].ptr);
  }
  // Bands or not, and how thick.  SODA_HIP_HOST_BANDS=0: never; =1: by the
  // rule of the first version (ghost rows at most half a band, at least four
  // bands of two or more staged chunks -- what the tests force onto small
  // grids); unset: by a timeline estimate of both ways (choose_bands), which
  // also bands runs the rule leaves whole -- heat3d 512^3 x 50: two bands of
  // 256 + 100 ghost planes do 1.4x the arithmetic and still finish earlier,
  // because the first band computes while the second half of the input is on
  // the link and its result leaves while the second band computes.
  const char* bands_env = getenv("SODA_HIP_HOST_BANDS");
  const bool never = bands_env && !strcmp(bands_env, "0");
  const bool by_rule = bands_env && !strcmp(bands_env, "1");
  // (not in 1-D: a band would start at an arbitrary cell, and the kernels are
  // built for rows that start on 16-byte boundaries and are whole vectors long)
  if (plan.has_reach && c.dim >= 2 && !never) {
    const int64_t g_lo = (int64_t)iterate * plan.reach_lo;
    const int64_t g_hi = (int64_t)iterate * plan.reach_hi;
    int64_t band_rows = 0;
    if (!by_rule) band_rows = choose_bands(c, g_lo, g_hi);
    if (by_rule || band_rows < 0) {        // (< 0: no time model for this plan)
      int64_t per_band = (2 * (g_lo + g_hi) + c.chunk_rows - 1) / c.chunk_rows;
      if (per_band < 2) per_band = 2;
      band_rows = per_band * c.chunk_rows;
      if (band_rows * 4 > c.rows + band_rows - 1) band_rows = 0;   // < 4 bands
    }
    if (band_rows > 0 && band_rows < c.rows)
      return run_banded(c, band_rows, g_lo, g_hi);
  }
  return run_whole(c);
}

}  // namespace soda_detail

extern "C" {

int soda_hip_run_host(soda_hip_program_t* p,
                      const soda_hip_host_tensor_t* inputs,
                      const soda_hip_host_tensor_t* outputs, int32_t iterate) {
  return soda_hip_run_host_box(p, inputs, outputs, iterate, nullptr, nullptr);
}

}  // extern "C"
