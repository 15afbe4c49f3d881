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

#define SODA_HIP_ABI_VERSION 4
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
} soda_hip_kargs_t;

typedef struct soda_hip_kernel_desc {
  char name[SODA_HIP_NAME_LEN];        /* extern "C" symbol in the code object */
  int32_t block[3];                    /* threads per block */
  int32_t tile[SODA_HIP_MAX_DIM];      /* output cells one block owns, per dim */
  int32_t lds_bytes;                   /* dynamic LDS */
  int32_t reserved;
} soda_hip_kernel_desc_t;

/* One way of advancing the program by `fused_iters` iterations: the listed
 * kernels launched in order (one fused kernel, or one kernel per stage). */
typedef struct soda_hip_pass_desc {
  int32_t fused_iters;
  int32_t num_kernels;
  int32_t kernel[SODA_HIP_MAX_PASS_KERNELS];
  float cost;                          /* relative time of one such pass (any
                                          unit); <= 0 in every pass: schedule
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
 * 4 host_tensor; 0 for anything else. */
size_t soda_hip_sizeof(int which);

/* -- JIT: HIP source text -> gfx950 code object (hiprtc; needs no GPU) ---- */
int soda_hip_compile(const char* source, const char* name,
                     const char* const* options, int32_t num_options,
                     void** code, size_t* code_size);
void soda_hip_free_code(void* code);

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

/* -- device memory and timing helpers for hosts without their own -------- */
int soda_hip_malloc(int32_t device, size_t bytes, void** ptr);
int soda_hip_free(int32_t device, void* ptr);
int soda_hip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream);
int soda_hip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);
int soda_hip_memset(void* dst, int value, size_t bytes, void* stream);
int soda_hip_stream_synchronize(void* stream);
int soda_hip_event_create(soda_hip_event_t** event);
int soda_hip_event_record(soda_hip_event_t* event, void* stream);
int soda_hip_event_elapsed_ms(soda_hip_event_t* start, soda_hip_event_t* stop,
                              float* ms);   /* synchronises on `stop` */
int soda_hip_event_destroy(soda_hip_event_t* event);

#ifdef __cplusplus
}
#endif
#endif  /* SODA_HIP_H_ */
