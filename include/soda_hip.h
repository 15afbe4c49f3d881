/* soda_hip.h -- C ABI of the MI355X stencil execution backend (libsoda_hip.so).
 *
 * This is the drop-in boundary for the hot path "execute the stencil a .soda
 * file describes".  In the reference that path is GENERATED code reached
 * through two interfaces, both replaced here:
 *
 *   (1) the kernel C ABI the generated host calls,
 *         extern "C" void <app>_kernel(ap_uint<BW>* bank_i_<out>...,
 *                                      ap_uint<BW>* bank_i_<in>...,
 *                                      uint64_t coalesced_data_num);
 *       reference src/soda/codegen/frt/host.py:44-59 (declaration),
 *       :282-289 (call), src/soda/codegen/xilinx/hls_kernel.py:62-66
 *       (definition).  Argument order there is OUTPUTS FIRST, THEN INPUTS;
 *       soda_hip_run_device() keeps that order.  The reference kernel
 *       returns void and has no error channel; every function here returns
 *       a status instead (0 = success, cf. `return 0;` frt/host.py:429).
 *
 *   (2) the operator-level host entry,
 *         int soda::app::<app>(const T* var_<in>_ptr, const int32_t extent[],
 *                              const int32_t stride[], const int32_t min[],
 *                              ... same four per output ..., const char*
 *                              bitstream, int burst_width, int tile_size_d...,
 *                              int unroll_factor);
 *       reference src/soda/codegen/frt/host.py:62-88 (signature), :181-249
 *       (tile + scatter), :282-322 (launch), :340-427 (gather of the valid
 *       box).  soda_hip_run_host() takes the same (ptr, extent, stride, min)
 *       quadruple per tensor; `min` is accepted and ignored exactly as the
 *       reference body ignores it; strides are honoured on the caller's
 *       arrays only (frt/host.py:236-246,395-424).
 *
 * The FPGA bitstream argument is replaced by a program handle made from HIP
 * source text (what `sodac --hip-kernel` prints) JIT-compiled for gfx950 plus
 * a launch plan.  Plain C types only; no C++ or torch types cross this ABI.
 * The library is synchronous per call unless a stream is given; it is not
 * re-entrant per program handle (the reference promises none either: single
 * host thread, frt/host.py:319-322).
 *
 * Result contract (reference docs/data-layout.md:12-25, frt/host.py:357-375):
 * only cells inside the valid box of each output are defined; everything else
 * in an output array is unspecified.
 */
#ifndef SODA_HIP_H_
#define SODA_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped when a struct that crosses the boundary changes layout (7: the plan
 * carries the program's per-iteration reach).  Entry points added since, under
 * the same number (no struct changed): soda_hip_host_register / _unregister,
 * soda_hip_host_weave_banks, soda_hip_stream_set_device_dense_min_tile; and
 * soda_hip_stream_create accepts wire[o] == NULL for an output on one bank
 * that the program stores at its wire position itself. */
#define SODA_HIP_ABI_VERSION 7
#define SODA_HIP_MAX_DIM 4
#define SODA_HIP_MAX_TENSORS 16
#define SODA_HIP_MAX_KERNELS 32
#define SODA_HIP_MAX_PASSES 8
#define SODA_HIP_MAX_PASS_KERNELS 16
#define SODA_HIP_MAX_PARAMS 8
#define SODA_HIP_NAME_LEN 96

enum soda_hip_status {
  SODA_HIP_OK = 0,
  SODA_HIP_ERR_INVALID = 1,     /* bad argument / plan / extent */
  SODA_HIP_ERR_COMPILE = 2,     /* hiprtc rejected the source; see last_error */
  SODA_HIP_ERR_RUNTIME = 3,     /* a HIP call failed; see last_error */
  SODA_HIP_ERR_NOMEM = 4,       /* host or device allocation failed */
  SODA_HIP_ERR_NODEVICE = 5,    /* no usable GPU */
  SODA_HIP_ERR_UNSUPPORTED = 6  /* valid request this build cannot serve */
};

/* Argument block every generated kernel receives by value
 * (`extern "C" __global__ void k(soda_hip_kargs_t a)`).  Tensor slots are
 * program-wide: inputs first, then outputs, then locals. */
typedef struct soda_hip_kargs {
  void* buf[SODA_HIP_MAX_TENSORS];     /* device pointers by slot */
  int64_t stride[SODA_HIP_MAX_DIM];    /* in elements; dimension 0 fastest */
  int32_t extent[SODA_HIP_MAX_DIM];    /* cells per dimension */
  int32_t ntile[SODA_HIP_MAX_DIM];     /* blocks per dimension; grid.x = product */
  int32_t tile[SODA_HIP_MAX_DIM];      /* cells one block owns (the kernel
                                          descriptor's tile: chunk lengths are
                                          run-time values, tuned per GPU/extent
                                          without recompiling) */
  int32_t origin[SODA_HIP_MAX_DIM];    /* position of the arrays' cell 0 in ... */
  int32_t gextent[SODA_HIP_MAX_DIM];   /* ... the global grid (a slab of a
                                          multi-GPU run; = 0 / extent else):
                                          `border: preserve` is about the
                                          GLOBAL border */
  int32_t skip_from;                   /* a launch may cover a marching kernel's
                                          chunks with a run of them left out:
                                          block tile t along the streamed
                                          dimension stands for tile
                                          t + (t >= skip_from ? skip_count : 0)
                                          (ntile there = tiles launched).  The
                                          chunks next to a slab's ghost rows
                                          and the rest then run as two
                                          launches, around a halo exchange
                                          (soda_hip_run_device_slab); 0, 0
                                          everywhere else */
  int32_t skip_count;
  int32_t reserved[2];
} soda_hip_kargs_t;

typedef struct soda_hip_kernel_desc {
  char name[SODA_HIP_NAME_LEN];        /* extern "C" symbol in the code object */
  int32_t block[3];                    /* threads per block */
  int32_t tile[SODA_HIP_MAX_DIM];      /* output cells one block owns, per dim */
  int32_t lds_bytes;                   /* dynamic LDS */
  /* What the library needs to check a launch and to shape it for the extent
   * it is given at RUN time (the reference bakes tile sizes into its kernel
   * at compile time, hls_kernel.py:30-202; here only the code is fixed, the
   * launch geometry is not).  All zero: a kernel with a fixed tile and no
   * constraint. */
  int32_t vec;                         /* cells per lane: extent[0] must be a
                                          multiple (0 / 1: any) */
  int32_t march_dim;                   /* 1 + the dimension a marching kernel
                                          streams along (its tile there is the
                                          chunk, a run-time value); 0: none */
  int32_t waves_along;                 /* waves of a block along that dimension */
  int32_t warm;                        /* row steps of pipeline warm-up a chunk
                                          pays on top of its own rows */
  int32_t window_extra;                /* buffer addressing: planes beyond the
                                          chunk one wave's window spans; the
                                          window must stay <= 1 GiB.  -1: plain
                                          pointers, no limit */
  int32_t max_elem;                    /* bytes of the widest element so addressed */
  int32_t vgprs;                       /* registers per lane of the compiled
                                          kernel (occupancy); 0: unknown, the
                                          tile is used as given */
  int32_t pipe;                        /* waves sharing the fused iterations */
  int32_t chunk_fixed;                 /* 1: never re-size the chunk */
  /* time model of one launch, used to pick the chunk and, per pass, the
   * multiset of passes that advances `iterate` iterations fastest on THIS
   * extent:  max(k * (chunk + warm - warm_saved) * step_ns, bytes / HBM rate)
   * with k = waves per SIMD the grid needs */
  float step_ns;                       /* one row step of one wave, ns of SIMD
                                          issue time; 0: memory time only */
  float warm_saved;                    /* row steps' worth of work the peeled
                                          warm-up skips */
  float bytes_per_cell;                /* HBM bytes per cell per launch (inputs
                                          + outputs), before halo re-reads */
  float lane_redundancy;               /* lanes of a strip / lanes that store */
  int32_t max_extent0;                 /* > 0: the kernel's block covers the
                                          whole row (its waves hand x-halos
                                          over through LDS): extent[0] must not
                                          exceed this, else ERR_INVALID */
} soda_hip_kernel_desc_t;

/* One way of advancing the program by `fused_iters` iterations: the listed
 * kernels launched in order (one fused kernel, or one kernel per stage). */
typedef struct soda_hip_pass_desc {
  int32_t fused_iters;
  int32_t num_kernels;
  int32_t kernel[SODA_HIP_MAX_PASS_KERNELS];
  float cost;                          /* relative time of one such pass (any
                                          unit) where the kernels carry no time
                                          model (step_ns and bytes_per_cell all
                                          zero); <= 0 in every pass: schedule
                                          greedily, deepest first */
} soda_hip_pass_desc_t;

typedef struct soda_hip_plan {
  int32_t abi_version;                 /* SODA_HIP_ABI_VERSION */
  int32_t dim;
  int32_t num_inputs;
  int32_t num_outputs;
  int32_t num_locals;                  /* scratch tensors the library owns */
  int32_t num_params;                  /* `param` arrays (ref grammar.py:41-45,
                                          frt/host.py:72-78): small read-only
                                          arrays, C order, the same for every
                                          iteration */
  int32_t param_elems[SODA_HIP_MAX_PARAMS];  /* elements per param array */
  /* slots: inputs, outputs, locals, params -- bytes per element of each */
  int32_t elem_size[SODA_HIP_MAX_TENSORS];
  int32_t num_kernels;
  soda_hip_kernel_desc_t kernels[SODA_HIP_MAX_KERNELS];
  int32_t num_passes;                  /* sorted by fused_iters, largest first;
                                          the last one must have fused_iters 1
                                          when the program iterates.  A run of
                                          N iterations uses the multiset of
                                          passes of least total cost that adds
                                          up to N, deepest first */
  soda_hip_pass_desc_t passes[SODA_HIP_MAX_PASSES];
  /* ABI 7: cells ONE iteration of the program reads below / above a cell
   * along the last dimension (the per-iteration growth of the outputs'
   * windows, reference core.py:858-919), valid if has_reach != 0.  Lets the
   * host-array entries cut a run into bands along that dimension -- each band
   * a window run with iterate x reach ghost rows -- so that the copy-in of
   * band i + 1 and the copy-out of band i - 1 run beside the kernels of band
   * i; without it they copy in, run, copy out. */
  int32_t has_reach;
  int32_t reach_lo, reach_hi;
} soda_hip_plan_t;

typedef struct soda_hip_program soda_hip_program_t;   /* opaque */
typedef struct soda_hip_event soda_hip_event_t;       /* opaque */

/* -- library ------------------------------------------------------------ */
int soda_hip_abi_version(void);
const char* soda_hip_status_string(int status);
/* Copies the calling thread's last error text (NUL-terminated) into buf;
 * returns the full length. */
size_t soda_hip_last_error(char* buf, size_t cap);
int soda_hip_device_count(int* count);
/* sizeof() of the ABI structs as this library was compiled, for bindings to
 * check their mirrors: 0 kargs, 1 kernel_desc, 2 pass_desc, 3 plan,
 * 4 host_tensor, 5 stream_desc, 6 slab_run, 7 group_desc, 8 slab_info,
 * 9 group_stats, 10 launch_info; 0 for anything else. */
size_t soda_hip_sizeof(int which);

/* -- JIT: HIP source text -> gfx950 code object (hiprtc; needs no GPU) ---- */
int soda_hip_compile(const char* source, const char* name,
                     const char* const* options, int32_t num_options,
                     void** code, size_t* code_size);
void soda_hip_free_code(void* code);
/* Version of the hiprtc that compiles (kernels built by another version may
 * differ in registers and code size, which the launch geometry and the
 * choice of peeled warm-up depend on: caches are keyed by it). */
int soda_hip_compiler_version(int32_t* major, int32_t* minor);

/* -- launch geometry (pure functions of the plan: no GPU, no program) ------- */
/* What a run on `extent` would use: the tile of every kernel (num_kernels x
 * SODA_HIP_MAX_DIM values; chunk lengths sized so the grid fills the 1024
 * SIMDs of an MI355X in whole rounds of equally loaded waves) and the modelled
 * time of every pass in nanoseconds (num_passes values; 0 where the kernels
 * carry no model).  Fails with SODA_HIP_ERR_INVALID -- text in last_error --
 * if a kernel cannot run this extent: rows that are not a multiple of its
 * vector width, a plane too large for its 1 GiB buffer window.  Either output
 * may be NULL. */
int soda_hip_plan_geometry(const soda_hip_plan_t* plan, const int32_t* extent,
                           int32_t* tiles, float* pass_ns);
/* How many times each pass runs to advance `iterate` iterations on `extent`
 * (num_passes values): the multiset of least total modelled time. */
int soda_hip_plan_schedule(const soda_hip_plan_t* plan, const int32_t* extent,
                           int32_t iterate, int32_t* count);

/* -- program = code object + plan, bound to one device -------------------- */
int soda_hip_program_create(const void* code, size_t code_size,
                            const soda_hip_plan_t* plan, int32_t device,
                            soda_hip_program_t** program);
int soda_hip_program_destroy(soda_hip_program_t* program);

/* Replaces <app>_kernel(): runs `iterate` iterations on device-resident,
 * dense (dim-0-fastest) arrays of `extent`.  outputs first, then inputs, as in
 * the reference ABI.  Inputs are never written.  Asynchronous on `stream`
 * (a hipStream_t, or NULL for the default stream); scratch buffers are owned
 * by the program and reused across calls.  `inputs` holds the num_inputs
 * tensors followed by the num_params param arrays (device pointers to
 * param_elems[k] elements each), the order of the reference's operator
 * signature (frt/host.py:72-78: inputs, outputs, then params). */
/* Same, for arrays that are a window of a larger grid (one GPU's slab):
 * `origin` = global position of cell 0 of the arrays, `global_extent` = size
 * of the whole grid (NULL, NULL = the arrays are the grid).  Only programs
 * with `border: preserve` look at them. */
int soda_hip_run_device_window(soda_hip_program_t* program,
                               void* const* outputs,
                               const void* const* inputs,
                               const int32_t* extent, const int32_t* origin,
                               const int32_t* global_extent, int32_t iterate,
                               void* stream);

/* Same, when the caller needs only part of the result: the cells [keep_lo,
 * keep_hi) along the LAST dimension (a slab's own rows -- its ghost rows are
 * refreshed by the next halo exchange anyway).  One iteration reads `reach_lo`
 * cells below and `reach_hi` above a cell along that dimension.  Every pass is
 * then launched only on the rows the iterations still to come can carry into
 * the kept range (a cone that narrows pass by pass); outside it the outputs
 * are unspecified -- NOT "unchanged": the last pass still stores up to
 * fused_iters x reach rows on either side of [keep_lo, keep_hi), computed with
 * zeros where the launch ended, and leaves the rows beyond those as they
 * were, so a caller that reuses the output arrays sees a mix of stale and
 * wrong rows there.  A side with keep_lo = 0 (keep_hi = extent) is never
 * trimmed.  The reference has no counterpart: its host tiles with a
 * replicated halo and recomputes all of it (frt/host.py:124-128). */
int soda_hip_run_device_cone(soda_hip_program_t* program,
                             void* const* outputs, const void* const* inputs,
                             const int32_t* extent, const int32_t* origin,
                             const int32_t* global_extent, int32_t iterate,
                             int32_t keep_lo, int32_t keep_hi,
                             int32_t reach_lo, int32_t reach_hi, void* stream);
/* Same, for a slab whose ghost rows are refreshed by a halo exchange the
 * caller runs on another stream -- RCCL send/recv, peer copies; the library
 * only sees the two events -- so that the exchange hides under the compute:
 *   ghost_lo / ghost_hi  rows [0, ghost_lo) and [extent - ghost_hi, extent) of
 *                        the INPUT arrays are being written by an exchange
 *                        that is complete when `ghosts_ready` (a hipEvent_t)
 *                        fires.  The chunks of the first pass whose inputs
 *                        stay clear of these rows are launched at once, the
 *                        others behind the event.  NULL: the ghosts are fresh;
 *   send_lo / send_hi    rows [keep_lo, keep_lo + send_lo) and [keep_hi -
 *                        send_hi, keep_hi) of the RESULT are what the
 *                        neighbours fetch next.  The chunks of the last pass
 *                        that deliver them, AND every chunk that writes a row
 *                        outside [keep_lo, keep_hi) -- the result's ghost
 *                        rows, which the next exchange overwrites -- whether
 *                        or not that side sends anything (a one-sided reach
 *                        has ghosts above and sends below), go first;
 *                        `sendable` (a hipEvent_t, recorded exactly once per
 *                        call; NULL: none) fires behind them; the rest of
 *                        the pass follows and writes kept rows only -- the
 *                        next exchange may start while it computes.
 * A split pass is two launches on `stream`, the part that does not depend on
 * the exchange first (environment SODA_HIP_SPLIT=side: the boundary chunks on
 * a stream the program owns, ordered against `stream` by events, so the parts
 * share the GPU -- measured slower).  Passes that are not one
 * marching kernel along the last dimension run whole: wait, compute, signal.
 * What the library cannot see is who still READS the arrays a call writes: a
 * neighbour that fetches rows of this slab's state on ITS stream must have
 * finished before a later call overwrites that state.  With a two-sided reach
 * the event chain implies it (its fetch -> its next pass -> its `sendable` ->
 * my next exchange -> my `ghosts_ready`); with a one-sided reach -- a slab
 * that sends to a neighbour it never receives from -- the caller has to order
 * `stream` behind that neighbour's fetch itself (soda_group.cpp does; a
 * transport that batches a rank's sends with its receives, like
 * dist.StreamOverlap, is covered by `ghosts_ready`).
 * The reference has no counterpart (one
 * device, frt/host.py:319-322); SURVEY.md 8(e): "compute boundary planes
 * first, send, compute interior". */
typedef struct soda_hip_slab_run {
  int32_t keep_lo, keep_hi;            /* as in soda_hip_run_device_cone */
  int32_t reach_lo, reach_hi;
  int32_t ghost_lo, ghost_hi;
  int32_t send_lo, send_hi;
  void* ghosts_ready;
  void* sendable;
} soda_hip_slab_run_t;
int soda_hip_run_device_slab(soda_hip_program_t* program, void* const* outputs,
                             const void* const* inputs, const int32_t* extent,
                             const int32_t* origin,
                             const int32_t* global_extent, int32_t iterate,
                             const soda_hip_slab_run_t* run, void* stream);
/* The launches a soda_hip_run_device_slab call on `extent` would issue, in
 * order, by the plan's time model (no GPU, no program; `run` may be NULL, its
 * two events only count as "given" or not): which rows of the last dimension
 * every pass covers and how a pass next to the exchange is cut -- block tiles
 * of `chunk` rows, `chunks` of them, the boundary [0, bnd_lo) U [bnd_hi,
 * chunks) on the side stream, the interior beside it.  At most `capacity`
 * entries are written; *count is the number of launches. */
typedef struct soda_hip_launch_info {
  int32_t fused_iters;
  int32_t lo, hi;
  int32_t wait, record, split;
  int32_t chunk, chunks, bnd_lo, bnd_hi;
} soda_hip_launch_info_t;
int soda_hip_plan_launches(const soda_hip_plan_t* plan, const int32_t* extent,
                           int32_t iterate, const soda_hip_slab_run_t* run,
                           int32_t capacity, soda_hip_launch_info_t* launches,
                           int32_t* count);
/* Cells along the last dimension the passes of the last run covered, summed
 * over the passes (a run that trims reports fewer than passes x extent). */
int soda_hip_last_rows(soda_hip_program_t* program, int64_t* rows);
int soda_hip_run_device(soda_hip_program_t* program, void* const* outputs,
                        const void* const* inputs, const int32_t* extent,
                        int32_t iterate, void* stream);

/* Replaces soda::app::<app>(): host arrays described by (ptr, extent, stride,
 * min) per tensor, inputs then outputs; copies in, runs, copies the outputs
 * back, synchronises.  Strides are in elements; every tensor must share one
 * extent. */
typedef struct soda_hip_host_tensor {
  void* ptr;
  const int32_t* extent;
  const int32_t* stride;
  const int32_t* min;      /* accepted, unused (as in the reference) */
} soda_hip_host_tensor_t;
/* `inputs`: the num_inputs tensors, then one entry per param array of which
 * only `ptr` is read (param_elems[k] contiguous elements). */
int soda_hip_run_host(soda_hip_program_t* program,
                      const soda_hip_host_tensor_t* inputs,
                      const soda_hip_host_tensor_t* outputs, int32_t iterate);
/* Same, but only box [valid_lo, valid_hi) of each output (num_outputs x dim
 * values each, NULL = whole array) is written to the caller's array, which is
 * what the reference's gather loop does with its compiled-in stencil offsets
 * (frt/host.py:357-375). */
int soda_hip_run_host_box(soda_hip_program_t* program,
                          const soda_hip_host_tensor_t* inputs,
                          const soda_hip_host_tensor_t* outputs,
                          int32_t iterate, const int32_t* valid_lo,
                          const int32_t* valid_hi);
/* How the two entries above move data (soda_host.cpp): through pinned staging
 * slots the program owns, ~16 MiB chunks along the last dimension
 * (SODA_HIP_HOST_CHUNK_MB), packed / unpacked by a process-wide pool of worker
 * threads (SODA_HIP_HOST_THREADS, default 8 -- the reference's pack and unpack
 * loops carry `#pragma omp parallel for`, frt/host.py:193,357) while the DMA
 * engine moves the neighbouring chunk.  With the program's per-iteration reach
 * in the plan (has_reach) a 2-D / 3-D run is cut into bands along the last
 * dimension -- window runs with iterate x reach ghost rows -- where a timeline
 * estimate of copy-in / kernels / copy-out says the overlap wins
 * (SODA_HIP_HOST_BANDS=0: never, =1: by a fixed rule; SODA_HIP_HOST_TRACE=1
 * prints the estimates and where the host thread's time went).  Slots and
 * worker threads live on the GPU's NUMA node (SODA_HIP_HOST_NUMA=0: wherever
 * the scheduler puts them).  A dense tensor inside a range pinned with
 * soda_hip_host_register skips the slots and the threads: its rows go by DMA
 * from / to the caller's array (an output whose box does not hold whole rows
 * by one strided copy per chunk).  SODA_HIP_HOST_DIRECT=0: never;
 * =attributes: also memory the HIP runtime reports as host-pinned
 * (hipHostMalloc, a hipHostRegister of the caller's) -- at the caller's risk:
 * a registration that is stale or not writable faults the GPU.  Results are
 * the same bits either way.
 * The pack / unpack step alone, exported
 * for callers that stage their own transfers and for tests without a GPU:
 * copies box [lo, hi) between a strided host array (strides in elements) and
 * a dense array that starts at index `row0` of the last dimension (0: holds
 * the whole array); to_dense != 0: strided -> dense.  threads: 1 = the calling
 * thread only, 0 = the pool. */
int soda_hip_host_copy_box(void* strided, const int32_t* stride, void* dense,
                           const int32_t* extent, const int32_t* lo,
                           const int32_t* hi, int32_t dim, int32_t elem,
                           int32_t to_dense, int32_t row0, int32_t threads);
/* The pack / unpack step of a tensor dealt over DRAM banks (the wire format's
 * streams, below: element k in bank k % num_banks at index k / num_banks;
 * reference docs/data-layout.md "Multi-Bank", frt/host.py:241-246,422-424):
 * stream elements [first, first + count) <-> a dense run starting at `dense`;
 * first and count multiples of num_banks.  What soda_hip_stream_run_host does
 * with banked tensors on its way through the staging slots; exported for
 * tests without a GPU.  threads as soda_hip_host_copy_box. */
int soda_hip_host_weave_banks(void* const* banks, int32_t num_banks, void* dense,
                              int64_t first, int64_t count, int32_t elem,
                              int32_t to_dense, int32_t threads);
/* Pins [ptr, ptr + bytes) of the caller's memory (hipHostRegister) so that
 * soda_hip_run_host / soda_hip_run_host_box reach it by DMA where it is -- for
 * hosts that keep their arrays across calls and do not link HIP themselves.
 * `ptr` must start a page (INVALID otherwise): the reference host's buffers
 * do (aligned_alloc(4096, ...), frt/host.py:165-178); ranges inside malloc's
 * heap share pages with other objects and are not accepted.  Meant for
 * memory that lives long: registering costs about as much as one copy of the
 * range -- once per array, not per call.  Unregister before freeing. */
int soda_hip_host_register(void* ptr, size_t bytes);
int soda_hip_host_unregister(void* ptr);

/* Measures one launch of every pass on `extent` (stand-in arrays; two warming
 * rounds, then four rounds of `launches` back-to-back launches per pass inside
 * HIP events on `stream`, the shortest round counts; synchronises) and
 * remembers the times: later runs on exactly this extent schedule their passes
 * by the clock instead of the model.  Costs milliseconds, once; programs with
 * a single pass have nothing to choose and return at once. */
int soda_hip_program_calibrate(soda_hip_program_t* program,
                               const int32_t* extent, int32_t launches,
                               void* stream);
/* Programs calibrate by themselves: the first run of more than one iteration
 * on an extent (any run entry; only the extent the call names, not the
 * trimmed sub-extents of cone runs) first times the passes on it -- it
 * synchronises `stream` once and costs a few milliseconds -- so that every
 * caller, the generated C++ host included, is scheduled by the clock, not by
 * the model (whose error is ~9 % per pass).  on = 0 turns that off for a
 * program; the environment variable SODA_HIP_NO_CALIBRATE=1 for all.
 * Two kinds of run never calibrate by themselves and stay asynchronous on
 * `stream` as documented: a soda_hip_run_device_slab call that is handed an
 * event (other streams are live beside it: the calibration's allocations and
 * its synchronisation would stall the neighbours' exchange), and any run on a
 * stream that is being captured into a graph.  Calibrate such extents
 * beforehand with soda_hip_program_calibrate, or they run by the model. */
int soda_hip_program_set_auto_calibrate(soda_hip_program_t* program, int on);
/* Time of one launch of every pass on `extent` in ns (num_passes values):
 * measured if calibrated (*measured = 1), else the model's. */
int soda_hip_program_pass_times(soda_hip_program_t* program,
                                const int32_t* extent, float* pass_ns,
                                int32_t* measured);

/* The schedule a run of `iterate` iterations on `extent` uses (num_passes
 * counts), from measured times where calibrated. */
int soda_hip_program_schedule(soda_hip_program_t* program,
                              const int32_t* extent, int32_t iterate,
                              int32_t* count);

/* Diagnostics: kernels generated with time stamps (sodac --hip-stamps) write
 * four 64-bit words per wavefront -- s_memtime at entry and exit, HW_ID,
 * XCC_ID -- to this device buffer (slot SODA_HIP_MAX_TENSORS - 1 of the kernel
 * arguments; at least 32 bytes x wavefronts of the largest launch).  NULL
 * (the default) for kernels without stamps. */
int soda_hip_program_set_debug_buffer(soda_hip_program_t* program, void* buf);

/* Number of kernel launches the last run_device/run_host call issued and how
 * many of them used the pass with the largest fused_iters. */
int soda_hip_last_launches(soda_hip_program_t* program, int32_t* launches,
                           int32_t* fused_launches);
/* Passes of the last run that were launched in two parts (boundary chunks /
 * interior) around a halo exchange. */
int soda_hip_last_split(soda_hip_program_t* program, int32_t* passes);

/* -- one host thread, N GPUs: a grid cut into slabs -------------------------
 * The reference's host is one blocking C++ sequence on one device
 * (frt/host.py:319-322: WriteToDevice, Exec, ReadFromDevice, Finish); its only
 * scale-out is host-side tiling with a replicated halo that is recomputed
 * (frt/host.py:124-128,181-249).  A group is the MI355X form of that host
 * (SURVEY.md 8b "one host thread drives N GPUs (one stream each)", 8e): the
 * LAST dimension -- the one SODA streams -- is cut into one contiguous slab
 * per device; every slab keeps `exchange_every x reach` ghost rows per side,
 * runs that many iterations without communication, then fetches its ghosts
 * from the neighbours' own rows by peer-to-peer copies (xGMI) on a stream of
 * its own while the rows that need no fresh ghost keep computing
 * (soda_hip_run_device_slab).  No collective anywhere.  The result on the
 * global valid box is the single-device result bit for bit.
 *
 * `device[]` may name one GPU several times: such slabs are "virtual devices"
 * that share it (copies between them are plain device copies).  This is how a
 * one-GPU box runs the whole N-slab schedule -- streams, events, overlap. */
#define SODA_HIP_MAX_SLABS 64
#define SODA_HIP_GROUP_NO_OVERLAP 1    /* exchange, then compute: for A/B runs */
#define SODA_HIP_GROUP_CALIBRATE 2     /* time every pass on every slab extent
                                          once (soda_hip_program_calibrate) */

#define SODA_HIP_GROUP_THREADS 4       /* one enqueueing thread per slab inside
                                          the library: the caller still sees one
                                          blocking sequence, but a step of
                                          9 launches x 8 GPUs no longer costs
                                          0.3 ms of ONE thread's time, as long
                                          as the GPUs take to run it */

typedef struct soda_hip_group_desc {
  int32_t num_slabs;
  int32_t device[SODA_HIP_MAX_SLABS];
  int32_t extent[SODA_HIP_MAX_DIM];    /* the whole grid */
  int32_t reach_lo, reach_hi;          /* cells along the last dimension one
                                          iteration reads below / above a cell
                                          (the stencil window's extent there:
                                          ref core.py:876-926) */
  int32_t iterate;                     /* iterations of a typical run: what the
                                          exchange interval is chosen for */
  int32_t exchange_every;              /* iterations between exchanges; 0: the
                                          library picks the interval of least
                                          modelled time from its per-extent
                                          pass times and a transfer model */
  int32_t flags;                       /* SODA_HIP_GROUP_* */
} soda_hip_group_desc_t;

typedef struct soda_hip_slab_info {
  int32_t device;
  int32_t begin, end;                  /* global rows held: own + ghosts */
  int32_t own_begin, own_end;
  int32_t ghost_lo, ghost_hi;
  int32_t extent[SODA_HIP_MAX_DIM];    /* of the slab's arrays */
  void* inputs[SODA_HIP_MAX_TENSORS];  /* device arrays holding the state the
                                          next run starts from (num_inputs) */
  void* outputs[SODA_HIP_MAX_TENSORS]; /* ... the last run's results
                                          (num_outputs; for a program that
                                          iterates these ARE `inputs`) */
} soda_hip_slab_info_t;

typedef struct soda_hip_group_stats {   /* of the last soda_hip_group_run */
  int32_t exchange_every;
  int32_t intervals;                   /* runs of <= exchange_every iterations */
  int32_t exchanges;                   /* of them opened by a halo exchange */
  int32_t copies;                      /* peer copies enqueued, all slabs */
  int64_t copy_bytes;
  int32_t launches;                    /* kernel launches, all slabs */
  int32_t split_passes;                /* passes launched in two parts */
  float enqueue_ms;                    /* host time the run took to enqueue */
} soda_hip_group_stats_t;

typedef struct soda_hip_group soda_hip_group_t;        /* opaque */

/* Loads the code object on every device of the group and lays out the slabs.
 * Fails with SODA_HIP_ERR_INVALID if a slab would be thinner than its
 * neighbour's ghost rows. */
int soda_hip_group_create(const void* code, size_t code_size,
                          const soda_hip_plan_t* plan,
                          const soda_hip_group_desc_t* desc,
                          soda_hip_group_t** group);
int soda_hip_group_destroy(soda_hip_group_t* group);
int soda_hip_group_slab(soda_hip_group_t* group, int32_t slab,
                        soda_hip_slab_info_t* info);
/* The exchange interval a group of this description would use (no GPU). */
int soda_hip_group_plan(const soda_hip_plan_t* plan,
                        const soda_hip_group_desc_t* desc,
                        int32_t* exchange_every);
/* Scatters host arrays (inputs, then one entry per param array of which only
 * `ptr` is read -- as soda_hip_run_host) over the slabs, ghost rows included:
 * the state the next run starts from.  Synchronous. */
int soda_hip_group_load(soda_hip_group_t* group,
                        const soda_hip_host_tensor_t* inputs);
/* Tells the group that the caller filled the slabs' `inputs` arrays on the
 * devices itself (soda_hip_slab_info_t), ghost rows included. */
int soda_hip_group_loaded(soda_hip_group_t* group);
/* Advances the state `iterate` iterations.  A program that iterates continues
 * from the previous run's result (its ghost rows are refreshed first).
 * Asynchronous: returns when everything is enqueued. */
int soda_hip_group_run(soda_hip_group_t* group, int32_t iterate);
int soda_hip_group_synchronize(soda_hip_group_t* group);
/* Gathers every slab's own rows of the results into host arrays; only box
 * [valid_lo, valid_hi) of each output is written (NULL: everything), as
 * soda_hip_run_host_box.  Synchronises first. */
int soda_hip_group_store(soda_hip_group_t* group,
                         const soda_hip_host_tensor_t* outputs,
                         const int32_t* valid_lo, const int32_t* valid_hi);
/* Replaces soda::app::<app>() on N GPUs: load, run, store. */
int soda_hip_group_run_host(soda_hip_group_t* group,
                            const soda_hip_host_tensor_t* inputs,
                            const soda_hip_host_tensor_t* outputs,
                            int32_t iterate, const int32_t* valid_lo,
                            const int32_t* valid_hi);
int soda_hip_group_last_stats(soda_hip_group_t* group,
                              soda_hip_group_stats_t* stats);

/* -- the reference kernel's WIRE format: <app>_kernel on banked streams -----
 * The reference's generated host hands its kernel one linear stream per
 * tensor -- tiles end to end, each padded to whole bursts, elements dealt
 * cyclically over the tensor's DRAM banks, kStencilDistance void elements
 * appended (src/soda/codegen/frt/host.py:124-249, docs/data-layout.md) -- and
 * calls
 *     <app>_kernel(bank_0_<out>..., bank_0_<in>..., coalesced_data_num)
 * (frt/host.py:44-59 declaration, :282-289 call; outputs first).  A stream
 * object runs exactly that contract on the GPU, so the unmodified generated
 * host can link against it under SODA_CPP_BINDING (`sodac --hip-wire-kernel`
 * prints the extern "C" <app>_kernel definition that calls it).  Per call:
 * un-interleave the inputs (skipped for single-bank inputs, read in place),
 * run the program on the de-interleaved stream -- as the original n-D program
 * with the marching kernels when the stream is a dense array of rows, else as
 * the linearised 1-D program -- and write the outputs back shifted by their
 * stencil offset (frt/host.py:400-424) and re-interleaved.  A program built
 * to store its outputs at their wire positions (`sodac --hip-wire-kernel`
 * does: store index moved by the window point of largest linear offset)
 * declares shift 0 for them; such an output on ONE bank is written by the
 * program straight into the caller's bank and has no copy kernel. */
typedef struct soda_hip_stream_desc {
  int32_t dim;                         /* of the original program */
  int32_t num_inputs;
  int32_t num_outputs;
  int32_t iterate;
  int32_t tile[SODA_HIP_MAX_DIM];      /* tile size of dimensions 0..dim-2 */
  int32_t stencil_distance;            /* kStencilDistance (frt/host.py:683-696) =
                                        * max(window distance, stencil offset),
                                        * reference core.py:620-625 */
  /* per tensor, inputs first, then outputs */
  int32_t banks[SODA_HIP_MAX_TENSORS];
  int32_t elem_size[SODA_HIP_MAX_TENSORS];
  int32_t elems_per_cycle[SODA_HIP_MAX_TENSORS];  /* burst width / element
                                          width x banks (frt/host.py:120-122) */
  int32_t shift[SODA_HIP_MAX_TENSORS]; /* inputs: produce offset the host delays
                                          the tensor by (frt/host.py:241-246);
                                          outputs: what is LEFT of the stencil
                                          offset the kernel emits a cell late
                                          by (:401-408) for the copy kernel to
                                          apply; 0 = the program did it */
  int32_t num_linear;                  /* 1-D programs offered, widest first */
  int32_t linear_vec[4];               /* their cells per thread; the last is 1 */
} soda_hip_stream_desc_t;

typedef struct soda_hip_stream soda_hip_stream_t;      /* opaque */

/* `dense`: the original program (NULL: always the linear form); `linear`:
 * num_linear programs of the linearised 1-D form; `unwire[i]` / `wire[o]`: the
 * copy kernels of input i / output o as one-kernel programs (unwire[i] /
 * wire[o] may be NULL for a tensor on one bank with shift 0).  The stream
 * borrows the programs; the caller destroys them after the stream. */
int soda_hip_stream_create(const soda_hip_stream_desc_t* desc,
                           soda_hip_program_t* dense,
                           soda_hip_program_t* const* linear,
                           soda_hip_program_t* const* unwire,
                           soda_hip_program_t* const* wire,
                           soda_hip_stream_t** stream);
int soda_hip_stream_destroy(soda_hip_stream_t* stream);
/* Bank pointers in the order of the reference kernel's ports: all banks of
 * output 0, output 1, ...; then all banks of input 0, input 1, ....  Device
 * pointers, asynchronous on `hip_stream`.  A tensor on ONE bank is read /
 * written in place by the program and must be 16-byte aligned (INVALID
 * otherwise; hipMalloc and the reference host's aligned_alloc(4096) are);
 * banks of a tensor on several may start anywhere (the copy kernels move 16
 * bytes per bank per thread when all are aligned, single elements if not). */
int soda_hip_stream_run_device(soda_hip_stream_t* stream,
                               void* const* out_banks,
                               const void* const* in_banks,
                               uint64_t coalesced_data_num, void* hip_stream);
/* The same on host buffers sized as the reference host allocates them
 * (coalesced_data_num x elems_per_cycle / banks elements per bank): what
 * <app>_kernel receives under SODA_CPP_BINDING.  Synchronous.  Where the
 * program stores its outputs late itself and the stream is a dense array of
 * rows, the banks go through the host-array entry on the n-D program (bands:
 * copy-in, kernels and copy-out overlapped; a banked tensor is
 * (de)interleaved, a delayed single-bank input un-delayed, by the host
 * threads on its way through the staging slots, no copy kernel runs;
 * SODA_HIP_STREAM_NO_BANDS=1: whole banks in, copy kernels, run, copy kernels,
 * whole banks out, as for every other stream). */
int soda_hip_stream_run_host(soda_hip_stream_t* stream, void* const* out_banks,
                             const void* const* in_banks,
                             uint64_t coalesced_data_num);
/* 1: the last run used the dense n-D form, 2: the linear form, 0: none yet */
int soda_hip_stream_last_mode(soda_hip_stream_t* stream);
/* soda_hip_stream_run_device takes the dense view only for tiles at least this
 * wide in dimension 0 (default 256: narrower ones leave most of a marching
 * strip idle and the linear form is faster on the GPU; 0: whenever there is a
 * dense view).  soda_hip_stream_run_host always prefers the dense view: there
 * the copies dominate, and the dense view lets them overlap in bands. */
int soda_hip_stream_set_device_dense_min_tile(soda_hip_stream_t* stream,
                                              int32_t min_tile0);

/* -- device memory and timing helpers for hosts without their own -------- */
int soda_hip_malloc(int32_t device, size_t bytes, void** ptr);
int soda_hip_free(int32_t device, void* ptr);
int soda_hip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream);
int soda_hip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);
int soda_hip_memcpy_d2d(void* dst, int32_t dst_device, const void* src,
                        int32_t src_device, size_t bytes, void* stream);
int soda_hip_memset(void* dst, int value, size_t bytes, void* stream);
int soda_hip_stream_synchronize(void* stream);
int soda_hip_event_create(soda_hip_event_t** event);
int soda_hip_event_record(soda_hip_event_t* event, void* stream);
int soda_hip_event_elapsed_ms(soda_hip_event_t* start, soda_hip_event_t* stop,
                              float* ms);   /* synchronises on `stop` */
int soda_hip_event_destroy(soda_hip_event_t* event);
/* For callers that run their own halo exchange beside
 * soda_hip_run_device_slab: the hipEvent_t inside an event (what
 * soda_hip_slab_run_t takes), a HIP stream of their own, and ordering a
 * stream behind an event. */
int soda_hip_event_handle(soda_hip_event_t* event, void** hip_event);
int soda_hip_hipstream_create(int32_t device, void** stream);
int soda_hip_hipstream_destroy(void* stream);
int soda_hip_hipstream_wait_event(void* stream, soda_hip_event_t* event);

#ifdef __cplusplus
}
#endif
#endif  /* SODA_HIP_H_ */
