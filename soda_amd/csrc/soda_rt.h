// soda_rt.h -- device-side runtime of the generated gfx950 stencil kernels.
//
// The code generator (soda_amd/codegen/hip) pastes this text in front of every
// program's kernels before JIT compilation (hiprtc has no include path into
// this tree).  It plays the part of the fixed helper text the reference's HLS
// emitter prints ahead of its modules (reference
// src/soda/codegen/xilinx/hls_kernel.py:238-336: BurstRead/BurstWrite,
// ReadData/WriteData): moving rows between memory and on-chip storage.  Here
// "on-chip" is the register file of a 64-lane wavefront; neighbours along
// dimension 0 come from adjacent lanes through DPP whole-wave shifts instead
// of the FPGA's FIFO-chained line buffer.
//
// gfx950 only: wave64, DPP wave_shr/wave_shl (GFX9 encodings).

// hiprtc (the JIT path) has no <stdint.h>; offline hipcc builds of this text do
#ifdef __HIPCC_RTC__
typedef __INT8_TYPE__ int8_t;
typedef __UINT8_TYPE__ uint8_t;
typedef __INT16_TYPE__ int16_t;
typedef __UINT16_TYPE__ uint16_t;
typedef __INT32_TYPE__ int32_t;
typedef __UINT32_TYPE__ uint32_t;
typedef __INT64_TYPE__ int64_t;
typedef __UINT64_TYPE__ uint64_t;
#else
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

// must match soda_hip_kargs_t in include/soda_hip.h
struct soda_hip_kargs_t {
  void* buf[16];
  int64_t stride[4];
  int32_t extent[4];
  int32_t ntile[4];
  int32_t tile[4];
  int32_t origin[4];
  int32_t gextent[4];
  int32_t skip_from;     // block tile t along a marching kernel's streamed
  int32_t skip_count;    // dimension stands for t + (t >= skip_from ? skip_count : 0)
  int32_t reserved[2];
};

#define SODA_DEV static __device__ inline __attribute__((always_inline))

// min/max/abs of the DSL: arguments evaluated once, result type given by the
// usual arithmetic conversions (same as the CPU oracle's macros in ir.py)
template <class A, class B>
SODA_DEV auto soda_min(A a, B b) -> decltype(a + b) {
  const decltype(a + b) x = a, y = b;
  return y < x ? y : x;
}
template <class A, class B>
SODA_DEV auto soda_max(A a, B b) -> decltype(a + b) {
  const decltype(a + b) x = a, y = b;
  return x < y ? y : x;
}
template <class A>
SODA_DEV auto soda_abs(A a) -> decltype(+a) {
  const decltype(+a) x = a;
  return x < 0 ? -x : x;
}
#define SODA_MIN(a, b) soda_min((a), (b))
#define SODA_MAX(a, b) soda_max((a), (b))
#define SODA_ABS(a) soda_abs((a))

// ---- whole-wave lane shifts (DPP) -----------------------------------------
// soda_lane_dn(v): lane i receives lane i-1's v (lane 0 receives 0).
// soda_lane_up(v): lane i receives lane i+1's v (lane 63 receives 0).
// bound_ctrl:1 = lanes without a source lane read 0, so no `old` register has
// to be materialised (saves one v_mov per shift; the shift itself fuses into
// the consuming v_add_f32 as a DPP operand)
SODA_DEV int soda_dpp_shr1(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x138 /* wave_shr:1 */, 0xf, 0xf,
                                     true);
}
SODA_DEV int soda_dpp_shl1(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x130 /* wave_shl:1 */, 0xf, 0xf,
                                     true);
}

// same, but the lane with no source (0 resp. 63) receives ITS OWN `edge`
// value: DPP leaves `old` in lanes whose source lane does not exist
SODA_DEV int soda_dpp_shr1_or(int v, int edge) {
  return __builtin_amdgcn_update_dpp(edge, v, 0x138, 0xf, 0xf, false);
}
SODA_DEV int soda_dpp_shl1_or(int v, int edge) {
  return __builtin_amdgcn_update_dpp(edge, v, 0x130, 0xf, 0xf, false);
}

// The DPP result of INTEGER and 64-bit shifts is passed through an empty asm
// so that LLVM's DPP-combine pass cannot fold the shift into the consuming
// instruction: on ROCm 7.x that folding produced wrong results for fused
// integer stencils (v_add_u32_dpp / v_subrev_u32_dpp; found by tests/
// test_fuzz.py, fixed by -amdgpu-dpp-combine=false) and illegal encodings for
// double ("DP ALU dpp only support row_newbcast", also for an fp32 shift folded
// into v_cvt_f64_f32).  Pure 32-bit-float programs keep the folding
// (v_add_f32_dpp; the generator defines SODA_FOLD_F32_DPP for them): it is what
// the hot jacobi/heat kernels rely on, and every such program of the suites is
// bit-exact with it.
//
// SODA_UNGUARDED_OPAQUE / _WIDE / _OWN switch ONE of the three compiler-fault
// workarounds of this file off (tests/test_compiler_pins.py builds each
// fault's reproducer that way to record whether the installed compiler still
// needs it; never defined in a product build).
SODA_DEV int soda_opaque(int v) {
#ifndef SODA_UNGUARDED_OPAQUE
  asm volatile("" : "+v"(v));
#endif
  return v;
}

// A one-byte cell as the int C promotes it to, its value range HIDDEN from the
// compiler (an empty asm).  hipcc (ROCm 7.2, every level above -O0) turns
// expressions over bytes it knows to be bytes into packed-byte instructions --
// v_dot4_u32_u8 over v_perm_b32-assembled operands, SDWA byte selects -- and
// gets some of them wrong next to a min / max and lane-shifted copies (round
// 4, tools/fuzz_scan.py deep seed 159 and its reduction: a quarter of the
// cells of a uint8 program, exact at -O0; round 3, seed 613: v_min_i32_sdwa).
// An operand that is "some int" takes the plain 32-bit instructions.
template <class T>
SODA_DEV int soda_wide(T v) {
  int w = (int)v;
#ifndef SODA_UNGUARDED_WIDE
  asm volatile("" : "+v"(w));
#endif
  return w;
}

// every element of a fragment in a 32-bit register of its own (no instruction:
// the value passes through an empty asm the compiler cannot see through)
template <class T, int V>
SODA_DEV void soda_own_register(T (&v)[V]) {
#ifndef SODA_UNGUARDED_OWN
#pragma unroll
  for (int e = 0; e < V; ++e) {
    int w = (int)v[e];
    asm volatile("" : "+v"(w));
    v[e] = (T)w;
  }
#endif
}

template <class T, int kSize = sizeof(T), bool kFloat = __is_floating_point(T)>
struct soda_lane_shift;

#ifdef SODA_FOLD_F32_DPP   // set by the generator for programs without 64-bit types
template <class T>
struct soda_lane_shift<T, 4, true> {
  SODA_DEV T dn(T v) {
    return __builtin_bit_cast(T, soda_dpp_shr1(__builtin_bit_cast(int, v)));
  }
  SODA_DEV T up(T v) {
    return __builtin_bit_cast(T, soda_dpp_shl1(__builtin_bit_cast(int, v)));
  }
  SODA_DEV T dn_or(T v, T edge) {
    return __builtin_bit_cast(T, soda_dpp_shr1_or(__builtin_bit_cast(int, v),
                                                  __builtin_bit_cast(int, edge)));
  }
  SODA_DEV T up_or(T v, T edge) {
    return __builtin_bit_cast(T, soda_dpp_shl1_or(__builtin_bit_cast(int, v),
                                                  __builtin_bit_cast(int, edge)));
  }
};

#endif

template <class T>
struct soda_lane_shift<T, 4, false> {
  SODA_DEV T dn(T v) {
    return __builtin_bit_cast(T, soda_opaque(soda_dpp_shr1(__builtin_bit_cast(int, v))));
  }
  SODA_DEV T up(T v) {
    return __builtin_bit_cast(T, soda_opaque(soda_dpp_shl1(__builtin_bit_cast(int, v))));
  }
  SODA_DEV T dn_or(T v, T edge) {
    return __builtin_bit_cast(T, soda_opaque(soda_dpp_shr1_or(
        __builtin_bit_cast(int, v), __builtin_bit_cast(int, edge))));
  }
  SODA_DEV T up_or(T v, T edge) {
    return __builtin_bit_cast(T, soda_opaque(soda_dpp_shl1_or(
        __builtin_bit_cast(int, v), __builtin_bit_cast(int, edge))));
  }
};

#ifndef SODA_FOLD_F32_DPP
// mixed-precision programs: an fp32 shift folded into v_cvt_f64_f32 is illegal
template <class T>
struct soda_lane_shift<T, 4, true> : soda_lane_shift<int, 4, false> {
  typedef soda_lane_shift<int, 4, false> I;
  SODA_DEV T dn(T v) { return __builtin_bit_cast(T, I::dn(__builtin_bit_cast(int, v))); }
  SODA_DEV T up(T v) { return __builtin_bit_cast(T, I::up(__builtin_bit_cast(int, v))); }
  SODA_DEV T dn_or(T v, T e) {
    return __builtin_bit_cast(T, I::dn_or(__builtin_bit_cast(int, v), __builtin_bit_cast(int, e)));
  }
  SODA_DEV T up_or(T v, T e) {
    return __builtin_bit_cast(T, I::up_or(__builtin_bit_cast(int, v), __builtin_bit_cast(int, e)));
  }
};
#endif

template <class T, bool kFloat>
struct soda_lane_shift<T, 8, kFloat> {
  struct pair { int lo, hi; };
  SODA_DEV T dn(T v) {
    pair p = __builtin_bit_cast(pair, v);
    p.lo = soda_opaque(soda_dpp_shr1(p.lo));
    p.hi = soda_opaque(soda_dpp_shr1(p.hi));
    return __builtin_bit_cast(T, p);
  }
  SODA_DEV T up(T v) {
    pair p = __builtin_bit_cast(pair, v);
    p.lo = soda_opaque(soda_dpp_shl1(p.lo));
    p.hi = soda_opaque(soda_dpp_shl1(p.hi));
    return __builtin_bit_cast(T, p);
  }
  SODA_DEV T dn_or(T v, T edge) {
    pair p = __builtin_bit_cast(pair, v), q = __builtin_bit_cast(pair, edge);
    p.lo = soda_opaque(soda_dpp_shr1_or(p.lo, q.lo));
    p.hi = soda_opaque(soda_dpp_shr1_or(p.hi, q.hi));
    return __builtin_bit_cast(T, p);
  }
  SODA_DEV T up_or(T v, T edge) {
    pair p = __builtin_bit_cast(pair, v), q = __builtin_bit_cast(pair, edge);
    p.lo = soda_opaque(soda_dpp_shl1_or(p.lo, q.lo));
    p.hi = soda_opaque(soda_dpp_shl1_or(p.hi, q.hi));
    return __builtin_bit_cast(T, p);
  }
};

template <class T, bool kFloat>
struct soda_lane_shift<T, 2, kFloat> {  // widened: one VGPR per element
  SODA_DEV T dn(T v) { return (T)soda_opaque(soda_dpp_shr1((int)v)); }
  SODA_DEV T up(T v) { return (T)soda_opaque(soda_dpp_shl1((int)v)); }
  SODA_DEV T dn_or(T v, T edge) { return (T)soda_opaque(soda_dpp_shr1_or((int)v, (int)edge)); }
  SODA_DEV T up_or(T v, T edge) { return (T)soda_opaque(soda_dpp_shl1_or((int)v, (int)edge)); }
};

template <class T, bool kFloat>
struct soda_lane_shift<T, 1, kFloat> {
  SODA_DEV T dn(T v) { return (T)soda_opaque(soda_dpp_shr1((int)v)); }
  SODA_DEV T up(T v) { return (T)soda_opaque(soda_dpp_shl1((int)v)); }
  SODA_DEV T dn_or(T v, T edge) { return (T)soda_opaque(soda_dpp_shr1_or((int)v, (int)edge)); }
  SODA_DEV T up_or(T v, T edge) { return (T)soda_opaque(soda_dpp_shl1_or((int)v, (int)edge)); }
};

template <class T> SODA_DEV T soda_lane_dn(T v) { return soda_lane_shift<T>::dn(v); }
template <class T> SODA_DEV T soda_lane_up(T v) { return soda_lane_shift<T>::up(v); }
template <class T> SODA_DEV T soda_lane_dn_or(T v, T edge) { return soda_lane_shift<T>::dn_or(v, edge); }
template <class T> SODA_DEV T soda_lane_up_or(T v, T edge) { return soda_lane_shift<T>::up_or(v, edge); }

// ---- lane shifts through the LDS crossbar (ds_bpermute_b32) -----------------
// Measured on gfx950 (tools/valubench.py): a VALU op carrying a DPP shift
// costs ~15 issue cycles against 2 for a plain one, whatever the DPP pattern;
// ds_bpermute runs in the LDS pipe beside the VALU and, issued a stage ahead of
// its use, costs only its issue slot.  Lane 0 (63) receives lane 63's (0's)
// value instead of 0: only halo lanes see the difference.
template <class T, int kSize = sizeof(T)>
struct soda_bperm;
template <class T>
struct soda_bperm<T, 4> {
  SODA_DEV T get(int byte_addr, T v) {
    return __builtin_bit_cast(
        T, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
  }
};
template <class T>
struct soda_bperm<T, 8> {
  struct pair { int lo, hi; };
  SODA_DEV T get(int byte_addr, T v) {
    pair p = __builtin_bit_cast(pair, v);
    p.lo = __builtin_amdgcn_ds_bpermute(byte_addr, p.lo);
    p.hi = __builtin_amdgcn_ds_bpermute(byte_addr, p.hi);
    return __builtin_bit_cast(T, p);
  }
};
template <class T>
struct soda_bperm<T, 2> {
  SODA_DEV T get(int byte_addr, T v) {
    return (T)__builtin_amdgcn_ds_bpermute(byte_addr, (int)v);
  }
};
template <class T>
struct soda_bperm<T, 1> {
  SODA_DEV T get(int byte_addr, T v) {
    return (T)__builtin_amdgcn_ds_bpermute(byte_addr, (int)v);
  }
};
template <class T> SODA_DEV T soda_lane_from(int byte_addr, T v) {
  return soda_bperm<T>::get(byte_addr, v);
}

// ---- lane shifts through ds_swizzle (LDS crossbar, no LDS memory) -----------
// A DPP-carrying VALU op stalls the vector issue of gfx950 far beyond its own
// slot (tools/dppbench.py, tools/tickbench.py: the row step of the fused
// jacobi kernel sustains 3.7-4.0 cycles per instruction with its 6 DPP
// modifiers, 2.1-2.8 without).  ds_swizzle_b32 in rotate mode moves a register
// by one lane through the LDS crossbar instead: it issues in the LDS pipe beside
// the VALU (one issue slot, result after ~50 cycles, counted by lgkmcnt) and,
// unlike ds_bpermute, needs no address register and a quarter of the LDS time.
// The rotation is WITHIN each 32-lane half of the wave:
//   *32 forms: lanes 0/32 (dn) resp. 31/63 (up) receive the value that wrapped
//              around their half -- for strips laid out as two independent
//              32-lane halves, whose end lanes are halo lanes anyway;
//   *64 forms: the one lane that must cross the halves is patched with
//              v_readlane / v_writelane (two plain VALU ops); lanes 0 (dn) and
//              63 (up) receive a wrapped value instead of 0: they are halo
//              lanes of a 64-lane strip.
// offset = 0xC000 | direction << 10 | amount << 5 (GFX9 rotate mode)
#define SODA_SWZ_ROT_UP 0xC020   /* lane i <- lane (i + 1) % 32 of its half */
#define SODA_SWZ_ROT_DN 0xC420   /* lane i <- lane (i - 1) % 32 of its half */
SODA_DEV int soda_swz_dn32(int v) {
  return __builtin_amdgcn_ds_swizzle(v, SODA_SWZ_ROT_DN);
}
SODA_DEV int soda_swz_up32(int v) {
  return __builtin_amdgcn_ds_swizzle(v, SODA_SWZ_ROT_UP);
}
SODA_DEV int soda_swz_dn64(int v) {
  int r = __builtin_amdgcn_ds_swizzle(v, SODA_SWZ_ROT_DN);
  const int s = __builtin_amdgcn_readlane(v, 31);
  asm("v_writelane_b32 %0, %1, 32" : "+v"(r) : "s"(s));   // no builtin in ROCm 7.2
  return r;
}
SODA_DEV int soda_swz_up64(int v) {
  int r = __builtin_amdgcn_ds_swizzle(v, SODA_SWZ_ROT_UP);
  const int s = __builtin_amdgcn_readlane(v, 32);
  asm("v_writelane_b32 %0, %1, 31" : "+v"(r) : "s"(s));
  return r;
}

template <class T, int kSize = sizeof(T)>
struct soda_swz;
template <class T>
struct soda_swz<T, 4> {
  SODA_DEV T dn32(T v) { return __builtin_bit_cast(T, soda_swz_dn32(__builtin_bit_cast(int, v))); }
  SODA_DEV T up32(T v) { return __builtin_bit_cast(T, soda_swz_up32(__builtin_bit_cast(int, v))); }
  SODA_DEV T dn64(T v) { return __builtin_bit_cast(T, soda_swz_dn64(__builtin_bit_cast(int, v))); }
  SODA_DEV T up64(T v) { return __builtin_bit_cast(T, soda_swz_up64(__builtin_bit_cast(int, v))); }
};
template <class T>
struct soda_swz<T, 8> {
  struct pair { int lo, hi; };
#define SODA_SWZ_PAIR(fn)                          \
  SODA_DEV T fn(T v) {                             \
    pair p = __builtin_bit_cast(pair, v);          \
    p.lo = soda_swz_##fn(p.lo);                    \
    p.hi = soda_swz_##fn(p.hi);                    \
    return __builtin_bit_cast(T, p);               \
  }
  SODA_SWZ_PAIR(dn32) SODA_SWZ_PAIR(up32) SODA_SWZ_PAIR(dn64) SODA_SWZ_PAIR(up64)
#undef SODA_SWZ_PAIR
};
template <class T>
struct soda_swz<T, 2> {   // widened: one VGPR per element
  SODA_DEV T dn32(T v) { return (T)soda_swz_dn32((int)v); }
  SODA_DEV T up32(T v) { return (T)soda_swz_up32((int)v); }
  SODA_DEV T dn64(T v) { return (T)soda_swz_dn64((int)v); }
  SODA_DEV T up64(T v) { return (T)soda_swz_up64((int)v); }
};
template <class T>
struct soda_swz<T, 1> : soda_swz<T, 2> {};
// up-shift of a whole 64-lane strip with the crossing lane patched by ONE DPP
// move: wave_shl:1 restricted to row 1, bank 3 (lanes 28-31) -- lanes 28-30
// get what the rotation gave them anyway, lane 31 gets lane 32's value
SODA_DEV int soda_swz_up64d(int v) {
  const int r = __builtin_amdgcn_ds_swizzle(v, SODA_SWZ_ROT_UP);
  return __builtin_amdgcn_update_dpp(r, v, 0x130, 0x2, 0x8, false);
}
template <class T> SODA_DEV T soda_lane_up64d(T v) {
  static_assert(sizeof(T) == 4, "mix64d: 4-byte cells");
  return __builtin_bit_cast(T, soda_swz_up64d(__builtin_bit_cast(int, v)));
}
template <class T> SODA_DEV T soda_lane_dn32(T v) { return soda_swz<T>::dn32(v); }
template <class T> SODA_DEV T soda_lane_up32(T v) { return soda_swz<T>::up32(v); }
template <class T> SODA_DEV T soda_lane_dn64(T v) { return soda_swz<T>::dn64(v); }
template <class T> SODA_DEV T soda_lane_up64(T v) { return soda_swz<T>::up64(v); }

// ---- row fragments: V consecutive cells of one row per lane ----------------
template <class T, int V>
struct soda_vec {
  typedef T type __attribute__((ext_vector_type(V)));
};

template <class T, int V, bool kNonTemporal = false>
SODA_DEV void soda_load_frag(T (&dst)[V], const T* __restrict__ p) {
  if constexpr (V == 1) {
    dst[0] = kNonTemporal ? __builtin_nontemporal_load(p) : *p;
  } else {
    typedef typename soda_vec<T, V>::type vec_t;
    const vec_t* vp = reinterpret_cast<const vec_t*>(p);
    const vec_t t = kNonTemporal ? __builtin_nontemporal_load(vp) : *vp;
#pragma unroll
    for (int e = 0; e < V; ++e) dst[e] = t[e];
  }
}

template <class T, int V>
SODA_DEV void soda_zero_frag(T (&dst)[V]) {
#pragma unroll
  for (int e = 0; e < V; ++e) dst[e] = (T)0;
}

template <class T, int V, bool kNonTemporal = false>
SODA_DEV void soda_store_frag(T* __restrict__ p, const T (&src)[V]) {
  if constexpr (V == 1) {
    if constexpr (kNonTemporal) __builtin_nontemporal_store(src[0], p);
    else *p = src[0];
  } else {
    typedef typename soda_vec<T, V>::type vec_t;
    vec_t t;
#pragma unroll
    for (int e = 0; e < V; ++e) t[e] = src[e];
    if constexpr (kNonTemporal)
      __builtin_nontemporal_store(t, reinterpret_cast<vec_t*>(p));
    else
      *reinterpret_cast<vec_t*>(p) = t;
  }
}

// ---- the same fragments through buffer resources ---------------------------
// A raw buffer access whose offset is >= num_records is dropped by the memory
// pipeline (loads return 0, stores vanish): rows above/below the grid, lanes
// beyond the row end and pipeline warm-up need NO branch, so every vector-memory
// instruction of the marching loop is straight-line code and the compiler's
// s_waitcnt insertion can count them exactly.  (With `if (row_ok) load` the
// counters merge pessimistically at each join and the loop got an
// `s_waitcnt vmcnt(0)` right behind its prefetch loads.)
//
// Offsets are unsigned bytes from the start of the wave's window of a tensor;
// the host keeps windows <= SODA_BUF_WINDOW_MAX bytes so that
//   valid + valid < 2^30,  valid + SODA_OOB_X in [2^30, 2^31),
//   SODA_OOB_ROW + anything >= 2^31 without wrapping;
// 3-D kernels add three parts (plane, row, lane), each in range or
// SODA_OOB_X: in range together they stay below 2^30, with one to three
// sentinels the sum lies in [2^30, 2^32) -- out of range, no wrap.
typedef unsigned soda_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned soda_u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t soda_rsrc_t;
#define SODA_OOB_X 0x40000000u
#define SODA_OOB_ROW 0x80000000u
#define SODA_BUF_WINDOW_MAX 0x40000000ll

SODA_DEV soda_rsrc_t soda_make_rsrc(const void* base, int64_t bytes) {
  // word 3 = 0x00020000: raw buffer, 32-bit data format (gfx9 encoding)
  return __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(base), 0,
      (int)(bytes < 0 ? 0 : bytes > SODA_BUF_WINDOW_MAX ? SODA_BUF_WINDOW_MAX
                                                        : bytes),
      0x00020000);
}

template <class T, int V, bool kNonTemporal = false>
SODA_DEV void soda_buf_load_frag(T (&dst)[V], soda_rsrc_t r, unsigned off) {
  constexpr int kBytes = V * (int)sizeof(T);
  constexpr int kAux = kNonTemporal ? 2 : 0;   // gfx94x/95x: bit 1 = nt
  if constexpr (kBytes % 16 == 0) {
#pragma unroll
    for (int i = 0; i < kBytes / 16; ++i) {
      const soda_u32x4 t =
          __builtin_amdgcn_raw_buffer_load_b128(r, off + 16 * i, 0, kAux);
      __builtin_memcpy((char*)dst + 16 * i, &t, 16);
    }
  } else if constexpr (kBytes == 8) {
    const soda_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, kAux);
    __builtin_memcpy(dst, &t, 8);
  } else if constexpr (kBytes == 4) {
    const unsigned t = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, kAux);
    __builtin_memcpy(dst, &t, 4);
  } else if constexpr (kBytes == 2) {
    const unsigned short t =
        __builtin_amdgcn_raw_buffer_load_b16(r, off, 0, kAux);
    __builtin_memcpy(dst, &t, 2);
  } else {
    static_assert(kBytes == 1, "fragment size");
    const unsigned char t = __builtin_amdgcn_raw_buffer_load_b8(r, off, 0, kAux);
    __builtin_memcpy(dst, &t, 1);
  }
}

template <class T, int V, bool kNonTemporal = false>
SODA_DEV void soda_buf_store_frag(soda_rsrc_t r, unsigned off,
                                  const T (&src)[V]) {
  constexpr int kBytes = V * (int)sizeof(T);
  constexpr int kAux = kNonTemporal ? 2 : 0;
  if constexpr (kBytes % 16 == 0) {
#pragma unroll
    for (int i = 0; i < kBytes / 16; ++i) {
      soda_u32x4 t;
      __builtin_memcpy(&t, (const char*)src + 16 * i, 16);
      __builtin_amdgcn_raw_buffer_store_b128(t, r, off + 16 * i, 0, kAux);
    }
  } else if constexpr (kBytes == 8) {
    soda_u32x2 t;
    __builtin_memcpy(&t, src, 8);
    __builtin_amdgcn_raw_buffer_store_b64(t, r, off, 0, kAux);
  } else if constexpr (kBytes == 4) {
    unsigned t;
    __builtin_memcpy(&t, src, 4);
    __builtin_amdgcn_raw_buffer_store_b32(t, r, off, 0, kAux);
  } else if constexpr (kBytes == 2) {
    unsigned short t;
    __builtin_memcpy(&t, src, 2);
    __builtin_amdgcn_raw_buffer_store_b16(t, r, off, 0, kAux);
  } else {
    static_assert(kBytes == 1, "fragment size");
    unsigned char t;
    __builtin_memcpy(&t, src, 1);
    __builtin_amdgcn_raw_buffer_store_b8(t, r, off, 0, kAux);
  }
}

// ---- stage-pipelined blocks ------------------------------------------------
// One barrier per row step: every LDS access this wave has issued has
// completed (lgkmcnt(0)) before any wave of the block starts the next step.
// Deliberately NOT __syncthreads(): a workgroup fence would also drain vmcnt,
// i.e. wait for the global prefetch loads the first wave keeps in flight.
// value of lane `from` in every lane (a scalar register): the halo cell a
// strip's neighbour handed over, fed to a DPP shift as its `old` operand
template <class T>
SODA_DEV T soda_bcast(T v, int from) {
  static_assert(sizeof(T) <= 8, "one or two registers");
  unsigned u[2] = {0u, 0u};
  __builtin_memcpy(u, &v, sizeof(T));
  u[0] = (unsigned)__builtin_amdgcn_readlane((int)u[0], from);
  if (sizeof(T) > 4) u[1] = (unsigned)__builtin_amdgcn_readlane((int)u[1], from);
  T r;
  __builtin_memcpy(&r, u, sizeof(T));
  return r;
}

SODA_DEV void soda_pipe_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
