// soda_group.cpp -- one host thread, N GPUs: a grid cut into slabs along the
// streamed dimension, halo exchange by peer-to-peer copies hidden under the
// compute (interface: include/soda_hip.h, "one host thread, N GPUs").
//
// What the reference's generated host does for ONE device -- scatter the
// input, launch, gather the valid box (reference
// src/soda/codegen/frt/host.py:181-249, 282-322, 340-427) -- with its
// replicated-halo tiling (frt/host.py:124-128) turned into slabs that stay
// resident on their GPUs and trade ghost rows as the iterations advance.
//
// Ordering is by events only; the host never waits inside a run:
//
//   slab s, stream `main`:  I(first pass) ......... passes ........ I(last pass)
//   slab s, stream `side`:    wait G_s -> B(first)        B(last) -> record S_s
//   slab s, stream `comm`:  wait S_s, S_s-1, S_s+1 (previous interval) ->
//                           copy neighbours' send rows into my ghost rows ->
//                           record G_s
//
// I = chunks whose inputs stay clear of the ghost rows / that deliver no send
// row, B = the others (soda_hip_run_device_slab splits the pass).  S_s also
// says "nothing of that interval writes my result's ghost rows any more", so
// the copy into them may start.  State ping-pongs between two arrays per
// tensor; that a neighbour has finished reading an array before its owner
// overwrites it follows from the chain  copy -> neighbour's B(first) ->
// neighbour's B(last) -> S -> my next copy -> my B(first) -> my last pass --
// where I fetch from that neighbour.  With a one-sided reach I do not, so
// `main` also waits, at the head of interval j, for G(j - 1) of every
// neighbour that fetched from me (enqueue_interval).
#include "soda_internal.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

using namespace soda_detail;

namespace {

struct Slab {
  int device = 0;
  soda_hip_program* prog = nullptr;
  int32_t own_begin = 0, own_end = 0, ghost_lo = 0, ghost_hi = 0;
  int32_t begin = 0, end = 0;
  int32_t extent[SODA_HIP_MAX_DIM];
  int32_t origin[SODA_HIP_MAX_DIM];
  std::vector<DeviceBuffer> a, b, params;
  // the two array sets the state ping-pongs between: a program that iterates
  // reads side[j & 1] and writes the other in interval j (counted since the
  // group was made, the same on every slab); any other reads a, writes b
  std::vector<void*> side[2];
  hipStream_t main = nullptr, comm = nullptr;
  hipEvent_t ghosts_ready[2] = {nullptr, nullptr};   // by interval parity
  hipEvent_t sendable[2] = {nullptr, nullptr};
  // the interval whose exchange recorded ghosts_ready[parity] last (-1: none)
  int64_t exchanged_at[2] = {-1, -1};
  // intervals whose launches (and `sendable` record) are enqueued: what a
  // neighbour's enqueueing thread waits for before it orders a copy behind them
  std::atomic<int64_t> enqueued{0};
  // of the last run (summed over the slabs for soda_hip_group_last_stats)
  int32_t copies = 0, launches = 0, split_passes = 0;
  int64_t copy_bytes = 0;
  int rc = 0;
  std::string error;
};

// Transfer model behind the choice of the exchange interval.  A peer copy
// over one xGMI link (MI355X_MICROARCH.md: ~153 GB/s per link and direction
// peak; copies of a few MB reach about a third of it); what an exchange costs
// even when its bytes hide under the compute: the event hand-overs between
// three streams and two devices, and two passes launched in two parts each.
const double kXferLatencyNs = 20000.0;
const double kXferBytesPerNs = 50.0;
// (measured: a pass launched in two parts costs ~17 us more than whole, two
// per exchange -- profiles/r03_slab_overlap.jsonl)
const double kSplitNs = 34000.0;

}  // namespace

struct soda_hip_group {
  soda_hip_plan_t plan;
  soda_hip_group_desc_t desc;
  int32_t every = 1;           // iterations between exchanges
  bool iterable = false;       // outputs feed the inputs of the next run
  bool loaded = false;
  bool fresh = false;          // ghost rows of the state hold neighbour data
  int64_t interval = 0;        // intervals run since the group was made
  int64_t row_cells = 1;
  std::vector<std::unique_ptr<Slab>> slabs;
  soda_hip_group_stats_t stats;
  // SODA_HIP_GROUP_THREADS: one enqueueing thread per slab
  std::vector<std::thread> workers;
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  int64_t job = 0;             // number of the run the workers should enqueue
  int32_t job_iterate = 0;
  int finished = 0;
  bool stop = false;
  std::atomic<bool> failed{false};

  const std::vector<void*>& inputs_of(const Slab& s, int64_t j) const {
    return s.side[iterable ? (int)(j & 1) : 0];
  }
  const std::vector<void*>& outputs_of(const Slab& s, int64_t j) const {
    return s.side[iterable ? (int)((j & 1) ^ 1) : 1];
  }
};

namespace {

void slab_rows(const soda_hip_group_desc_t& d, int dim, int32_t every, int s,
               Slab* out) {
  const int n = d.num_slabs;
  const int32_t rows = d.extent[dim - 1];
  const int32_t base = rows / n, extra = rows % n;
  out->own_begin = s * base + (s < extra ? s : extra);
  out->own_end = out->own_begin + base + (s < extra ? 1 : 0);
  out->ghost_lo = s > 0 ? d.reach_lo * every : 0;
  out->ghost_hi = s < n - 1 ? d.reach_hi * every : 0;
  out->begin = out->own_begin - out->ghost_lo;
  out->end = out->own_end + out->ghost_hi;
  for (int i = 0; i < SODA_HIP_MAX_DIM; ++i) {
    out->extent[i] = i < dim ? d.extent[i] : 1;
    out->origin[i] = 0;
  }
  out->extent[dim - 1] = out->end - out->begin;
  out->origin[dim - 1] = out->begin;
}

int check_desc(const soda_hip_plan_t& plan, const soda_hip_group_desc_t& d) {
  if (d.num_slabs < 1 || d.num_slabs > SODA_HIP_MAX_SLABS)
    return fail(SODA_HIP_ERR_INVALID, "group: bad number of slabs");
  if (d.reach_lo < 0 || d.reach_hi < 0 || d.iterate < 1 || d.exchange_every < 0)
    return fail(SODA_HIP_ERR_INVALID,
                "group: reaches must be >= 0, iterate >= 1, exchange_every >= 0");
  for (int i = 0; i < plan.dim; ++i)
    if (d.extent[i] < 1) return fail(SODA_HIP_ERR_INVALID, "group: extent < 1");
  if (d.iterate > 1 && plan.num_inputs != plan.num_outputs)
    return fail(SODA_HIP_ERR_INVALID,
                "number of input tensors must be the same as output if "
                "iterate > 1 times");
  return SODA_HIP_OK;
}

// every rank of a decomposition must hold the rows its neighbours fetch:
// the thinnest slab decides
int check_interval(const soda_hip_plan_t& plan, const soda_hip_group_desc_t& d,
                   int32_t every) {
  const int32_t base = d.extent[plan.dim - 1] / d.num_slabs;
  const int32_t reach = d.reach_lo > d.reach_hi ? d.reach_lo : d.reach_hi;
  if (d.num_slabs > 1 && (int64_t)reach * every > base) {
    char buf[256];
    snprintf(buf, sizeof buf,
             "slabs of %d rows (%d rows over %d slabs) are thinner than the "
             "%lld-row halo; use fewer GPUs or a smaller exchange interval",
             base, d.extent[plan.dim - 1], d.num_slabs,
             (long long)reach * every);
    return fail(SODA_HIP_ERR_INVALID, buf);
  }
  return SODA_HIP_OK;
}

// time of one launch of every pass on an extent: the model's, or the clock's
typedef int (*PassTimes)(void* ctx, const soda_hip_plan_t& plan,
                         const int32_t* ext, std::vector<double>* ns);

int model_times(void* ctx, const soda_hip_plan_t& plan, const int32_t* ext,
                std::vector<double>* ns) {
  auto* cache = static_cast<std::map<ExtentKey, ExtentPlan>*>(ctx);
  const ExtentPlan* ep = nullptr;
  if (int rc = extent_plan(plan, cache, ext, &ep)) return rc;
  *ns = ep->model_ns;
  return SODA_HIP_OK;
}

int measured_times(void* ctx, const soda_hip_plan_t& plan, const int32_t* ext,
                   std::vector<double>* ns) {
  soda_hip_program* prog = static_cast<soda_hip_program*>(ctx);
  if (int rc = soda_hip_program_calibrate(prog, ext, 4, nullptr)) return rc;
  std::vector<float> f(plan.num_passes);
  if (int rc = soda_hip_program_pass_times(prog, ext, f.data(), nullptr)) return rc;
  ns->assign(f.begin(), f.end());
  return SODA_HIP_OK;
}

// time of `iters` iterations by the given pass times, and of the first + last pass
int interval_ns(const soda_hip_plan_t& plan, const std::vector<double>& ns,
                int32_t iters, double* total, double* edge) {
  int32_t count[SODA_HIP_MAX_PASSES], n = 0;
  if (int rc = schedule(plan, ns, iters, count, &n)) return rc;
  *total = 0;
  int first = -1, last = -1;
  for (int i = 0; i < plan.num_passes; ++i) {
    *total += count[i] * ns[i];
    if (count[i]) {
      if (first < 0) first = i;
      last = i;
    }
  }
  *edge = n > 1 ? ns[first] + ns[last] : (first >= 0 ? ns[first] : 0.0);
  return SODA_HIP_OK;
}

struct Candidate {
  int32_t every;
  double ns;
};

// Time of a run of desc.iterate iterations with `k` iterations per interval,
// seen from a middle slab.  Longer intervals mean fewer exchanges but more
// ghost rows to compute.
int interval_cost(const soda_hip_plan_t& plan, const soda_hip_group_desc_t& d,
                  int32_t k, PassTimes times, void* ctx, double* cost) {
  const int ax = plan.dim - 1;
  const int n = d.num_slabs;
  const int32_t reach = d.reach_lo > d.reach_hi ? d.reach_lo : d.reach_hi;
  const int32_t own = d.extent[ax] / n;
  const int32_t rounds = (d.iterate + k - 1) / k;
  const int32_t tail = d.iterate - (rounds - 1) * k;
  // ghosts on both sides (one side when there are two slabs)
  const int32_t ghosts = (n > 2 ? d.reach_lo + d.reach_hi : reach) * k;
  int32_t ext[SODA_HIP_MAX_DIM];
  for (int i = 0; i < SODA_HIP_MAX_DIM; ++i) ext[i] = i < plan.dim ? d.extent[i] : 1;
  ext[ax] = own + ghosts;
  std::vector<double> ns;
  if (int rc = times(ctx, plan, ext, &ns)) return rc;
  double t_full = 0, e_full = 0, t_tail = 0, e_tail = 0;
  if (int rc = interval_ns(plan, ns, k, &t_full, &e_full)) return rc;
  if (int rc = interval_ns(plan, ns, tail, &t_tail, &e_tail)) return rc;
  // passes trim the ghost rows nothing can carry into own rows any more: on
  // average little more than half of them are computed
  const double trim = (own + 0.55 * ghosts) / (double)(own + ghosts);
  int64_t row_bytes = 0, cells = 1;
  for (int i = 0; i < ax; ++i) cells *= d.extent[i];
  for (int t = 0; t < plan.num_inputs; ++t) row_bytes += cells * plan.elem_size[t];
  const bool overlap = !(d.flags & SODA_HIP_GROUP_NO_OVERLAP);
  const double bytes_ns = (double)reach * k * row_bytes / kXferBytesPerNs;
  double exposed = overlap ? bytes_ns - 0.6 * e_full : bytes_ns;
  if (exposed < 0) exposed = 0;
  // chained runs: every run opens with an exchange
  *cost = trim * ((rounds - 1) * t_full + t_tail) +
          rounds * (kXferLatencyNs + (overlap ? kSplitNs : 0.0) + exposed);
  return SODA_HIP_OK;
}

// The interval of least time for a run of desc.iterate iterations.
// Parametrised by the number of intervals r: K = ceil(iterate / r), rounded up
// to a whole number of the deepest pass.  All candidates are ranked by the
// model; with a program to measure on (`prog`), the best few are then timed on
// the GPU -- each on the slab extent it implies -- and the clock decides.
int pick_interval(const soda_hip_plan_t& plan, const soda_hip_group_desc_t& d,
                  soda_hip_program* prog, int32_t* every) {
  const int ax = plan.dim - 1;
  const int n = d.num_slabs;
  const int32_t reach = d.reach_lo > d.reach_hi ? d.reach_lo : d.reach_hi;
  if (n == 1 || reach == 0 || plan.num_inputs != plan.num_outputs) {
    *every = d.iterate;
    return SODA_HIP_OK;
  }
  const int32_t own = d.extent[ax] / n;
  const int32_t kmax = own / reach < d.iterate ? own / reach : d.iterate;
  if (kmax < 1) return check_interval(plan, d, 1);
  const int32_t deepest = plan.passes[0].fused_iters;
  bool modelled = true;
  for (int i = 0; i < plan.num_passes; ++i)
    for (int k = 0; k < plan.passes[i].num_kernels; ++k) {
      const soda_hip_kernel_desc_t& kd = plan.kernels[plan.passes[i].kernel[k]];
      modelled = modelled && (kd.step_ns > 0 || kd.bytes_per_cell > 0);
    }
  if (!modelled) {
    // no time model: ghost rows (both sides together) at most a quarter slab
    int32_t k = own / (4 * (d.reach_lo + d.reach_hi));
    k = k < 1 ? 1 : k > kmax ? kmax : k;
    if (deepest > 1 && k >= deepest) k = k / deepest * deepest;
    *every = k;
    return SODA_HIP_OK;
  }
  std::map<ExtentKey, ExtentPlan> cache;
  std::vector<Candidate> ranked;
  const int32_t r0 = (d.iterate + kmax - 1) / kmax;
  for (int32_t r = r0; r < r0 + 48; ++r) {
    int32_t k = (d.iterate + r - 1) / r;
    if (deepest > 1 && k > deepest) k = (k + deepest - 1) / deepest * deepest;
    if (k > kmax) k = kmax;
    if (k < 1) break;
    bool seen = false;
    for (auto& c : ranked) seen = seen || c.every == k;
    if (!seen) {
      double cost = 0;
      if (int rc = interval_cost(plan, d, k, model_times, &cache, &cost)) return rc;
      ranked.push_back({k, cost});
    }
    if (k <= deepest || k == 1) break;
  }
  for (size_t i = 1; i < ranked.size(); ++i)       // by modelled time
    for (size_t j = i; j > 0 && ranked[j].ns < ranked[j - 1].ns; --j)
      std::swap(ranked[j], ranked[j - 1]);
  if (prog) {
    if (ranked.size() > 4) ranked.resize(4);
    for (auto& c : ranked)
      if (int rc = interval_cost(plan, d, c.every, measured_times, prog, &c.ns))
        return rc;
    for (size_t i = 1; i < ranked.size(); ++i)
      if (ranked[i].ns < ranked[0].ns) std::swap(ranked[i], ranked[0]);
  }
  *every = ranked[0].every;
  return SODA_HIP_OK;
}

int copy_rows(const Slab& to, void* dst, const Slab& from, const void* src,
              size_t bytes) {
  if (to.device == from.device)
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, to.comm));
  else
    HIP_TRY(hipMemcpyPeerAsync(dst, to.device, src, from.device, bytes, to.comm));
  return SODA_HIP_OK;
}

// refreshes slab s's ghost rows of the state interval j starts from, from the
// neighbours' own rows
int enqueue_exchange(soda_hip_group* g, int s, int64_t j) {
  Slab& me = *g->slabs[s];
  const int prev = (int)((j - 1) & 1);
  const int n = (int)g->slabs.size();
  HIP_TRY(hipSetDevice(me.device));
  HIP_TRY(hipStreamWaitEvent(me.comm, me.sendable[prev], 0));
  for (int side = 0; side < 2; ++side) {
    const int peer = side == 0 ? s - 1 : s + 1;
    const int32_t ghost = side == 0 ? me.ghost_lo : me.ghost_hi;
    if (peer < 0 || peer >= n || ghost == 0) continue;
    Slab& nb = *g->slabs[peer];
    HIP_TRY(hipStreamWaitEvent(me.comm, nb.sendable[prev], 0));
    // global rows fetched, then local rows on either side
    const int32_t g0 = side == 0 ? me.begin : me.own_end;
    for (int t = 0; t < g->plan.num_inputs; ++t) {
      const int64_t row_bytes = g->row_cells * g->plan.elem_size[t];
      char* dst = static_cast<char*>(g->inputs_of(me, j)[t]) +
                  (int64_t)(g0 - me.begin) * row_bytes;
      const char* src = static_cast<const char*>(g->inputs_of(nb, j)[t]) +
                        (int64_t)(g0 - nb.begin) * row_bytes;
      if (int rc = copy_rows(me, dst, nb, src, (size_t)ghost * row_bytes))
        return rc;
      ++me.copies;
      me.copy_bytes += (int64_t)ghost * row_bytes;
    }
  }
  HIP_TRY(hipEventRecord(me.ghosts_ready[j & 1], me.comm));
  me.exchanged_at[j & 1] = j;
  return SODA_HIP_OK;
}

int enqueue_interval(soda_hip_group* g, int s, int64_t j, int32_t iters,
                     bool exchanged) {
  Slab& me = *g->slabs[s];
  const int n = (int)g->slabs.size();
  const bool overlap = !(g->desc.flags & SODA_HIP_GROUP_NO_OVERLAP);
  SlabRun run;
  run.cone = {me.ghost_lo, me.ghost_lo + (me.own_end - me.own_begin),
              g->desc.reach_lo, g->desc.reach_hi};
  run.ghost_lo = exchanged ? me.ghost_lo : 0;
  run.ghost_hi = exchanged ? me.ghost_hi : 0;
  // what the neighbours fetch: their ghost rows on the side that faces me
  run.send_lo = s > 0 ? g->slabs[s - 1]->ghost_hi : 0;
  run.send_hi = s < n - 1 ? g->slabs[s + 1]->ghost_lo : 0;
  run.ghosts_ready = exchanged ? me.ghosts_ready[j & 1] : nullptr;
  run.sendable = n > 1 ? me.sendable[j & 1] : nullptr;
  HIP_TRY(hipSetDevice(me.device));
  hipEvent_t after = nullptr;
  if (!overlap) {     // exchange, compute, signal -- nothing runs underneath
    if (run.ghosts_ready) HIP_TRY(hipStreamWaitEvent(me.main, run.ghosts_ready, 0));
    after = run.sendable;
    run.ghosts_ready = run.sendable = nullptr;
    run.ghost_lo = run.ghost_hi = run.send_lo = run.send_hi = 0;
  }
  // This interval writes the arrays interval j - 1 read -- and exchange j - 1
  // copied the neighbours' ghost rows OUT of.  Those copies run on the
  // neighbours' streams; when the reach is two-sided they are ordered ahead of
  // this interval by the chain copy -> the neighbour's interval j - 1 -> its
  // `sendable` -> my exchange j, but a neighbour I never fetch from (one-sided
  // reach: the program taps upward only, say) puts nothing in my way, and I
  // could overwrite rows it has not fetched yet -- seen as two wrong planes
  // once in ~300 random groups (tools/fuzz_scan.py group).  So: behind them.
  for (int peer = s - 1; j >= 1 && peer <= s + 1; peer += 2) {
    if (peer < 0 || peer >= n) continue;
    Slab& nb = *g->slabs[peer];
    const int32_t fetched = peer < s ? nb.ghost_hi : nb.ghost_lo;
    if (fetched > 0 && nb.exchanged_at[(j - 1) & 1] == j - 1)
      HIP_TRY(hipStreamWaitEvent(me.main, nb.ghosts_ready[(j - 1) & 1], 0));
  }
  const std::vector<void*>& cur = g->inputs_of(me, j);
  std::vector<const void*> ins(cur.begin(), cur.end());
  for (auto& prm : me.params) ins.push_back(prm.ptr);
  std::vector<void*> outs = g->outputs_of(me, j);
  if (int rc = run_core(me.prog, outs.data(), ins.data(), me.extent, me.origin,
                        g->desc.extent, iters, me.main, -1, &run))
    return rc;
  if (after) HIP_TRY(hipEventRecord(after, me.main));
  me.launches += me.prog->last_launches;
  me.split_passes += me.prog->last_split;
  return SODA_HIP_OK;
}

// Everything slab s has to enqueue for a run of `iterate` iterations that
// starts at interval j0.  Threaded: before a copy is ordered behind a
// neighbour's `sendable`, that neighbour's thread must have RECORDED it.
int enqueue_slab_run(soda_hip_group* g, int s, int64_t j0, int32_t iterate,
                     bool fresh) {
  const int n = (int)g->slabs.size();
  Slab& me = *g->slabs[s];
  int32_t done = 0;
  for (int64_t j = j0; done < iterate; ++j) {
    const int32_t k = iterate - done < g->every ? iterate - done : g->every;
    const bool exchange = !fresh && n > 1;
    if (exchange) {
      for (int peer = s - 1; peer <= s + 1; peer += 2)
        if (peer >= 0 && peer < n)
          while (g->slabs[peer]->enqueued.load(std::memory_order_acquire) < j) {
            if (g->failed.load()) return fail(SODA_HIP_ERR_RUNTIME,
                                              "another slab's thread failed");
            std::this_thread::yield();
          }
      if (int rc = enqueue_exchange(g, s, j)) return rc;
    }
    if (int rc = enqueue_interval(g, s, j, k, exchange)) return rc;
    me.enqueued.store(j + 1, std::memory_order_release);
    fresh = !g->iterable;
    done += k;
  }
  return SODA_HIP_OK;
}

void worker_main(soda_hip_group* g, int s) {
  int64_t seen = 0;
  for (;;) {
    int32_t iterate;
    {
      std::unique_lock<std::mutex> lock(g->mu);
      g->cv_job.wait(lock, [&] { return g->stop || g->job > seen; });
      if (g->stop) return;
      seen = g->job;
      iterate = g->job_iterate;
    }
    Slab& me = *g->slabs[s];
    me.rc = enqueue_slab_run(g, s, g->interval, iterate, g->fresh);
    if (me.rc) {
      me.error = last_error_text();
      g->failed.store(true);
    }
    {
      std::lock_guard<std::mutex> lock(g->mu);
      ++g->finished;
    }
    g->cv_done.notify_one();
  }
}

void destroy_slab(Slab& s) {
  (void)hipSetDevice(s.device);
  for (auto* v : {&s.a, &s.b, &s.params})
    for (auto& buf : *v)
      if (buf.ptr) (void)hipFree(buf.ptr);
  for (int i = 0; i < 2; ++i) {
    if (s.ghosts_ready[i]) (void)hipEventDestroy(s.ghosts_ready[i]);
    if (s.sendable[i]) (void)hipEventDestroy(s.sendable[i]);
  }
  if (s.main) (void)hipStreamDestroy(s.main);
  if (s.comm) (void)hipStreamDestroy(s.comm);
  if (s.prog) (void)soda_hip_program_destroy(s.prog);
}

int create_slab(soda_hip_group* g, int s, const void* code, size_t code_size) {
  Slab& me = *g->slabs[s];
  const soda_hip_plan_t& plan = g->plan;
  if (int rc = soda_hip_program_create(code, code_size, &plan, me.device, &me.prog))
    return rc;
  HIP_TRY(hipSetDevice(me.device));
  HIP_TRY(hipStreamCreateWithFlags(&me.main, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&me.comm, hipStreamNonBlocking));
  for (int i = 0; i < 2; ++i) {
    HIP_TRY(hipEventCreateWithFlags(&me.ghosts_ready[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&me.sendable[i], hipEventDisableTiming));
  }
  int64_t cells = g->row_cells * me.extent[plan.dim - 1];
  me.a.resize(plan.num_inputs);
  me.b.resize(plan.num_outputs);
  me.params.resize(plan.num_params);
  for (int t = 0; t < plan.num_inputs; ++t) {
    if (int rc = ensure(me.a[t], (size_t)cells * plan.elem_size[t])) return rc;
    me.side[0].push_back(me.a[t].ptr);
  }
  for (int o = 0; o < plan.num_outputs; ++o) {
    if (int rc = ensure(me.b[o], (size_t)cells * plan.elem_size[plan.num_inputs + o]))
      return rc;
    me.side[1].push_back(me.b[o].ptr);
  }
  const int prm0 = plan.num_inputs + plan.num_outputs + plan.num_locals;
  for (int k = 0; k < plan.num_params; ++k)
    if (int rc = ensure(me.params[k],
                        (size_t)plan.param_elems[k] * plan.elem_size[prm0 + k]))
      return rc;
  return SODA_HIP_OK;
}

}  // namespace

extern "C" {

int soda_hip_group_plan(const soda_hip_plan_t* plan,
                        const soda_hip_group_desc_t* desc,
                        int32_t* exchange_every) {
  if (!plan || !desc || !exchange_every)
    return fail(SODA_HIP_ERR_INVALID, "group_plan: NULL argument");
  // (also checks the plan and that its kernels can run this extent at all)
  if (int rc = soda_hip_plan_geometry(plan, desc->extent, nullptr, nullptr))
    return rc;
  if (int rc = check_desc(*plan, *desc)) return rc;
  int32_t every = desc->exchange_every;
  if (every == 0)
    if (int rc = pick_interval(*plan, *desc, nullptr, &every)) return rc;
  if (every > desc->iterate) every = desc->iterate;
  if (int rc = check_interval(*plan, *desc, every)) return rc;
  *exchange_every = every;
  return SODA_HIP_OK;
}

int soda_hip_group_create(const void* code, size_t code_size,
                          const soda_hip_plan_t* plan,
                          const soda_hip_group_desc_t* desc,
                          soda_hip_group_t** group) {
  if (!code || !code_size || !plan || !desc || !group)
    return fail(SODA_HIP_ERR_INVALID, "group_create: NULL argument");
  *group = nullptr;
  int32_t every = 0;
  if (int rc = soda_hip_group_plan(plan, desc, &every)) return rc;
  int ndev = 0;
  if (int rc = soda_hip_device_count(&ndev)) return rc;
  if (ndev < 1) return fail(SODA_HIP_ERR_NODEVICE, "no GPU visible");
  for (int s = 0; s < desc->num_slabs; ++s)
    if (desc->device[s] < 0 || desc->device[s] >= ndev)
      return fail(SODA_HIP_ERR_INVALID, "group_create: no such device");
  if (desc->exchange_every == 0 && (desc->flags & SODA_HIP_GROUP_CALIBRATE) &&
      desc->num_slabs > 1) {
    // the clock picks among the model's best intervals, on a middle slab's GPU
    soda_hip_program* probe = nullptr;
    if (int rc = soda_hip_program_create(code, code_size, plan,
                                         desc->device[desc->num_slabs / 2],
                                         &probe))
      return rc;
    int rc = pick_interval(*plan, *desc, probe, &every);
    const std::string why = last_error_text();
    soda_hip_program_destroy(probe);
    if (rc) return fail(rc, why);
    if (every > desc->iterate) every = desc->iterate;
    if (int rc2 = check_interval(*plan, *desc, every)) return rc2;
  }
  soda_hip_group* g = new (std::nothrow) soda_hip_group;
  if (!g) return fail(SODA_HIP_ERR_NOMEM, "new group");
  g->plan = *plan;
  g->desc = *desc;
  g->every = every;
  g->iterable = plan->num_inputs == plan->num_outputs;
  for (int t = 0; g->iterable && t < plan->num_inputs; ++t)
    g->iterable = plan->elem_size[t] == plan->elem_size[plan->num_inputs + t];
  memset(&g->stats, 0, sizeof g->stats);
  g->stats.exchange_every = every;
  for (int i = 0; i < plan->dim - 1; ++i) g->row_cells *= desc->extent[i];
  int rc = SODA_HIP_OK;
  for (int s = 0; s < desc->num_slabs && rc == SODA_HIP_OK; ++s) {
    g->slabs.emplace_back(new Slab);
    g->slabs[s]->device = desc->device[s];
    slab_rows(*desc, plan->dim, every, s, g->slabs[s].get());
    rc = create_slab(g, s, code, code_size);
  }
  // neighbours on different GPUs copy peer to peer over xGMI
  for (int s = 0; s + 1 < desc->num_slabs && rc == SODA_HIP_OK; ++s) {
    const int a = desc->device[s], b = desc->device[s + 1];
    if (a == b) continue;
    for (int dir = 0; dir < 2; ++dir) {
      const int from = dir ? b : a, to = dir ? a : b;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, from, to) != hipSuccess || !can) continue;
      if (hipSetDevice(from) != hipSuccess) continue;
      hipError_t e = hipDeviceEnablePeerAccess(to, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
        (void)hipGetLastError();     // copies then stage through the host
    }
  }
  if (rc == SODA_HIP_OK && (desc->flags & SODA_HIP_GROUP_CALIBRATE))
    for (auto& sl : g->slabs) {
      rc = soda_hip_program_calibrate(sl->prog, sl->extent, 4, sl->main);
      if (rc) break;
    }
  if (rc) {
    const std::string why = last_error_text();
    soda_hip_group_destroy(g);
    return fail(rc, why);
  }
  if ((desc->flags & SODA_HIP_GROUP_THREADS) && desc->num_slabs > 1)
    for (int s = 0; s < desc->num_slabs; ++s)
      g->workers.emplace_back(worker_main, g, s);
  *group = g;
  return SODA_HIP_OK;
}

int soda_hip_group_destroy(soda_hip_group_t* g) {
  if (!g) return SODA_HIP_OK;
  if (!g->workers.empty()) {
    {
      std::lock_guard<std::mutex> lock(g->mu);
      g->stop = true;
    }
    g->cv_job.notify_all();
    for (auto& w : g->workers) w.join();
  }
  for (auto& s : g->slabs) {
    if (s->main) {
      (void)hipSetDevice(s->device);
      (void)hipStreamSynchronize(s->main);
      (void)hipStreamSynchronize(s->comm);
    }
  }
  for (auto& s : g->slabs) destroy_slab(*s);
  delete g;
  return SODA_HIP_OK;
}

int soda_hip_group_slab(soda_hip_group_t* g, int32_t slab,
                        soda_hip_slab_info_t* info) {
  if (!g || !info) return fail(SODA_HIP_ERR_INVALID, "group_slab: NULL argument");
  if (slab < 0 || slab >= (int32_t)g->slabs.size())
    return fail(SODA_HIP_ERR_INVALID, "group_slab: no such slab");
  const Slab& s = *g->slabs[slab];
  memset(info, 0, sizeof *info);
  info->device = s.device;
  info->begin = s.begin;
  info->end = s.end;
  info->own_begin = s.own_begin;
  info->own_end = s.own_end;
  info->ghost_lo = s.ghost_lo;
  info->ghost_hi = s.ghost_hi;
  for (int i = 0; i < SODA_HIP_MAX_DIM; ++i) info->extent[i] = s.extent[i];
  // the state the next interval reads; the results of the last one (for a
  // program that iterates the same arrays)
  const std::vector<void*>& in = g->inputs_of(s, g->interval);
  const std::vector<void*>& out = g->iterable ? in : s.side[1];
  for (size_t t = 0; t < in.size(); ++t) info->inputs[t] = in[t];
  for (size_t o = 0; o < out.size(); ++o) info->outputs[o] = out[o];
  return SODA_HIP_OK;
}

int soda_hip_group_load(soda_hip_group_t* g,
                        const soda_hip_host_tensor_t* inputs) {
  if (!g || !inputs) return fail(SODA_HIP_ERR_INVALID, "group_load: NULL argument");
  const soda_hip_plan_t& plan = g->plan;
  const int dim = plan.dim;
  if (int rc = soda_hip_group_synchronize(g)) return rc;
  // every slab's rows (ghost rows included) through that slab's pinned ring,
  // packed by the worker threads while the previous chunk -- or the previous
  // slab's last chunks, on another GPU's link -- is in flight (soda_host.cpp)
  for (int t = 0; t < plan.num_inputs; ++t) {
    const soda_hip_host_tensor_t& h = inputs[t];
    if (!h.ptr || !h.extent || !h.stride)
      return fail(SODA_HIP_ERR_INVALID, "group_load: bad input tensor");
    for (int i = 0; i < dim; ++i)
      if (h.extent[i] != g->desc.extent[i])
        return fail(SODA_HIP_ERR_INVALID,
                    "group_load: a tensor's extent is not the group's");
    for (auto& s : g->slabs) {
      hipStream_t stream = nullptr;
      if (int rc = host_stream(s->prog, &stream)) return rc;
      if (int rc = send_rows(s->prog, h, g->desc.extent, dim, plan.elem_size[t],
                             s->begin, s->end,
                             g->inputs_of(*s, g->interval)[t], s->begin, stream))
        return rc;
    }
  }
  for (auto& s : g->slabs) {
    hipStream_t stream = nullptr;
    if (int rc = host_stream(s->prog, &stream)) return rc;
    HIP_TRY(hipStreamSynchronize(stream));
  }
  const int prm0 = plan.num_inputs + plan.num_outputs + plan.num_locals;
  for (int k = 0; k < plan.num_params; ++k) {
    const void* host = inputs[plan.num_inputs + k].ptr;
    if (!host) return fail(SODA_HIP_ERR_INVALID, "group_load: NULL param");
    const size_t bytes = (size_t)plan.param_elems[k] * plan.elem_size[prm0 + k];
    for (auto& s : g->slabs) {
      HIP_TRY(hipSetDevice(s->device));
      HIP_TRY(hipMemcpy(s->params[k].ptr, host, bytes, hipMemcpyHostToDevice));
    }
  }
  g->loaded = true;
  g->fresh = true;
  return SODA_HIP_OK;
}

int soda_hip_group_loaded(soda_hip_group_t* g) {
  if (!g) return fail(SODA_HIP_ERR_INVALID, "group_loaded: NULL group");
  if (int rc = soda_hip_group_synchronize(g)) return rc;
  g->loaded = true;
  g->fresh = true;
  return SODA_HIP_OK;
}

int soda_hip_group_run(soda_hip_group_t* g, int32_t iterate) {
  if (!g) return fail(SODA_HIP_ERR_INVALID, "group_run: NULL group");
  if (!g->loaded)
    return fail(SODA_HIP_ERR_INVALID,
                "group_run: no state (soda_hip_group_load first)");
  if (iterate < 1) return fail(SODA_HIP_ERR_INVALID, "cannot iterate < 1 times");
  if (iterate > 1 && !g->iterable)
    return fail(SODA_HIP_ERR_INVALID,
                "number of input tensors must be the same as output if "
                "iterate > 1 times");
  const auto t0 = std::chrono::steady_clock::now();
  const int n = (int)g->slabs.size();
  const int32_t intervals = (iterate + g->every - 1) / g->every;
  for (auto& s : g->slabs) {
    s->copies = s->launches = s->split_passes = 0;
    s->copy_bytes = 0;
    s->rc = 0;
  }
  int rc = SODA_HIP_OK;
  std::string why;
  if (!g->workers.empty()) {
    g->failed.store(false);
    {
      std::lock_guard<std::mutex> lock(g->mu);
      g->job_iterate = iterate;
      g->finished = 0;
      ++g->job;
    }
    g->cv_job.notify_all();
    {
      std::unique_lock<std::mutex> lock(g->mu);
      g->cv_done.wait(lock, [&] { return g->finished == n; });
    }
    for (auto& s : g->slabs)
      if (s->rc && !rc) {
        rc = s->rc;
        why = s->error;
      }
  } else {
    // one thread: interval by interval, so that every `sendable` is recorded
    // before a neighbour's copy is ordered behind it
    int32_t done = 0;
    bool fresh = g->fresh;
    for (int64_t j = g->interval; done < iterate && !rc; ++j) {
      const int32_t k = iterate - done < g->every ? iterate - done : g->every;
      const bool exchange = !fresh && n > 1;
      for (int s = 0; exchange && s < n && !rc; ++s) rc = enqueue_exchange(g, s, j);
      for (int s = 0; s < n && !rc; ++s) {
        rc = enqueue_interval(g, s, j, k, exchange);
        g->slabs[s]->enqueued.store(j + 1, std::memory_order_release);
      }
      fresh = !g->iterable;
      done += k;
    }
    if (rc) why = last_error_text();
  }
  soda_hip_group_stats_t& st = g->stats;
  memset(&st, 0, sizeof st);
  st.exchange_every = g->every;
  st.intervals = intervals;
  st.exchanges = n > 1 ? intervals - (g->fresh ? 1 : 0) : 0;
  for (auto& s : g->slabs) {
    st.copies += s->copies;
    st.copy_bytes += s->copy_bytes;
    st.launches += s->launches;
    st.split_passes += s->split_passes;
  }
  // (the arrays a program that iterates reads next follow from the count)
  g->interval += intervals;
  for (auto& s : g->slabs) s->enqueued.store(g->interval);
  g->fresh = !g->iterable;      // a result's ghost rows are stale
  st.enqueue_ms = std::chrono::duration<float, std::milli>(
                      std::chrono::steady_clock::now() - t0).count();
  if (rc) {
    g->loaded = false;          // the state is not what any caller expects
    return fail(rc, why);
  }
  return SODA_HIP_OK;
}

int soda_hip_group_synchronize(soda_hip_group_t* g) {
  if (!g) return fail(SODA_HIP_ERR_INVALID, "group_synchronize: NULL group");
  for (auto& s : g->slabs) {
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->main));
    HIP_TRY(hipStreamSynchronize(s->comm));
    if (s->prog->side) HIP_TRY(hipStreamSynchronize(s->prog->side));
  }
  return SODA_HIP_OK;
}

int soda_hip_group_store(soda_hip_group_t* g,
                         const soda_hip_host_tensor_t* outputs,
                         const int32_t* valid_lo, const int32_t* valid_hi) {
  if (!g || !outputs)
    return fail(SODA_HIP_ERR_INVALID, "group_store: NULL argument");
  const soda_hip_plan_t& plan = g->plan;
  const int dim = plan.dim;
  if (int rc = soda_hip_group_synchronize(g)) return rc;
  int32_t zero[SODA_HIP_MAX_DIM] = {0, 0, 0, 0};
  for (int o = 0; o < plan.num_outputs; ++o) {
    const soda_hip_host_tensor_t& h = outputs[o];
    if (!h.ptr || !h.extent || !h.stride)
      return fail(SODA_HIP_ERR_INVALID, "group_store: bad output tensor");
    for (int i = 0; i < dim; ++i)
      if (h.extent[i] != g->desc.extent[i])
        return fail(SODA_HIP_ERR_INVALID,
                    "group_store: a tensor's extent is not the group's");
    const int elem = plan.elem_size[plan.num_inputs + o];
    const int32_t* lo = valid_lo ? valid_lo + o * dim : zero;
    const int32_t* hi = valid_hi ? valid_hi + o * dim : g->desc.extent;
    // only the valid box reaches the caller's array (frt/host.py:357-375):
    // every slab's own rows of it, through the slab's pinned ring
    for (auto& s : g->slabs) {
      const void* result = g->iterable ? g->inputs_of(*s, g->interval)[o]
                                       : s->side[1][o];
      hipStream_t stream = nullptr;
      if (int rc = host_stream(s->prog, &stream)) return rc;
      if (int rc = fetch_rows(s->prog, h, g->desc.extent, lo, hi, dim, elem,
                              s->own_begin, s->own_end, result, s->begin,
                              stream))
        return rc;
    }
  }
  // (rows that went home by DMA directly are still on their way)
  for (auto& s : g->slabs) {
    hipStream_t stream = nullptr;
    if (int rc = host_stream(s->prog, &stream)) return rc;
    HIP_TRY(hipStreamSynchronize(stream));
  }
  return SODA_HIP_OK;
}

int soda_hip_group_run_host(soda_hip_group_t* g,
                            const soda_hip_host_tensor_t* inputs,
                            const soda_hip_host_tensor_t* outputs,
                            int32_t iterate, const int32_t* valid_lo,
                            const int32_t* valid_hi) {
  if (int rc = soda_hip_group_load(g, inputs)) return rc;
  if (int rc = soda_hip_group_run(g, iterate)) return rc;
  return soda_hip_group_store(g, outputs, valid_lo, valid_hi);
}

int soda_hip_group_last_stats(soda_hip_group_t* g,
                              soda_hip_group_stats_t* stats) {
  if (!g || !stats) return fail(SODA_HIP_ERR_INVALID, "group_stats: NULL argument");
  *stats = g->stats;
  return SODA_HIP_OK;
}

}  // extern "C"
