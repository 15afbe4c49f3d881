/* CPU ORACLE, hand-written known-answer kernels -- TEST INFRASTRUCTURE ONLY.
 *
 * Loop nests written by hand, straight from the DSL text of the reference's
 * own test programs (reference tests/src/jacobi2d.soda:5-6, blur.soda:5-8,
 * heat3d.soda:5-12) following the semantics of the reference's self-check
 * loop nest (src/soda/codegen/frt/host.py:558-624): tensor zero outside its
 * valid box, loads at x + idx - st_idx, C arithmetic, iterations chained.
 * They do NOT go through this repo's parser or code generators, so they pin
 * the front-end + generated oracle + HIP code generator against an independent
 * reading of the same programs.  Layout: dimension 0 fastest.
 * Build: gcc -O2 -ffp-contract=off (see oracle/Makefile).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* t0(0,0) = (t1(0,1) + t1(1,0) + t1(0,0) + t1(0,-1) + t1(-1,0)) * 0.2f */
int kat_jacobi2d(const float* in, float* out, int32_t w, int32_t h,
                 int32_t iterate) {
  size_t cells = (size_t)w * h;
  float* tmp[2] = {calloc(cells, sizeof(float)), calloc(cells, sizeof(float))};
  if (!tmp[0] || !tmp[1]) return 1;
  const float* src = in;
  memset(out, 0, cells * sizeof(float));
  for (int32_t it = 0; it < iterate; ++it) {
    float* dst = it == iterate - 1 ? out : tmp[it & 1];
    int32_t r = it + 1;
    for (int32_t q = r; q < h - r; ++q)
      for (int32_t p = r; p < w - r; ++p) {
        const float* c = src + (size_t)q * w + p;
        dst[(size_t)q * w + p] = (c[w] + c[1] + c[0] + c[-w] + c[-1]) * 0.2f;
      }
    src = dst;
  }
  free(tmp[0]);
  free(tmp[1]);
  return 0;
}

/* blur_x(0,0) = (input(0,0) + input(0,1) + input(0,2)) / 3
 * blur_y(0,0) = (blur_x(0,0) + blur_x(1,0) + blur_x(2,0)) / 3     (uint16) */
int kat_blur(const uint16_t* in, uint16_t* out, int32_t w, int32_t h) {
  size_t cells = (size_t)w * h;
  uint16_t* bx = calloc(cells, sizeof(uint16_t));
  if (!bx) return 1;
  memset(out, 0, cells * sizeof(uint16_t));
  for (int32_t q = 0; q < h - 2; ++q)
    for (int32_t p = 0; p < w; ++p) {
      const uint16_t* c = in + (size_t)q * w + p;
      bx[(size_t)q * w + p] = (uint16_t)((c[0] + c[w] + c[2 * (size_t)w]) / 3);
    }
  for (int32_t q = 0; q < h - 2; ++q)
    for (int32_t p = 0; p < w - 2; ++p) {
      const uint16_t* c = bx + (size_t)q * w + p;
      out[(size_t)q * w + p] = (uint16_t)((c[0] + c[1] + c[2]) / 3);
    }
  free(bx);
  return 0;
}

/* out(0,0,0) = .125f*in(1,0,0) + .125f*in(-1,0,0) + .125f*in(0,1,0)
 *            + .125f*in(0,-1,0) + .125f*in(0,0,1) + .125f*in(0,0,-1)
 *            + .25f*in(0,0,0) */
int kat_heat3d(const float* in, float* out, int32_t nx, int32_t ny, int32_t nz,
               int32_t iterate) {
  size_t sy = (size_t)nx, sz = (size_t)nx * ny, cells = sz * nz;
  float* tmp[2] = {calloc(cells, sizeof(float)), calloc(cells, sizeof(float))};
  if (!tmp[0] || !tmp[1]) return 1;
  const float* src = in;
  memset(out, 0, cells * sizeof(float));
  for (int32_t it = 0; it < iterate; ++it) {
    float* dst = it == iterate - 1 ? out : tmp[it & 1];
    int32_t r = it + 1;
    for (int32_t z = r; z < nz - r; ++z)
      for (int32_t y = r; y < ny - r; ++y)
        for (int32_t x = r; x < nx - r; ++x) {
          const float* c = src + z * sz + y * sy + x;
          dst[z * sz + y * sy + x] =
              .125f * c[1] + .125f * c[-1] + .125f * c[sy] +
              .125f * c[-(ptrdiff_t)sy] + .125f * c[sz] +
              .125f * c[-(ptrdiff_t)sz] + .25f * c[0];
        }
    src = dst;
  }
  free(tmp[0]);
  free(tmp[1]);
  return 0;
}

/* An asymmetric two-stage program with a non-zero store index, written for
 * tests/golden/skew2d.soda (this repo's own test program):
 *   local float:  b(0, 0) = a(-1, 0) * 2.0f + a(2, 1) * 3.0f - a(0, -2)
 *   output float: c(1, 0) = b(0, 0) - b(1, 1) * 0.5f + a(0, 0)
 * so c[p,q] = b[p-1,q] - b[p,q+1]*0.5f + a[p-1,q],
 *    b[u,v] = a[u-1,v]*2.0f + a[u+2,v+1]*3.0f - a[u,v-2];
 * boxes: b on [1,w-2)x[2,h-1), c on [2,w-2)x[2,h-2). */
int kat_skew2d(const float* a, float* c, int32_t w, int32_t h) {
  size_t cells = (size_t)w * h;
  float* b = calloc(cells, sizeof(float));
  if (!b) return 1;
  memset(c, 0, cells * sizeof(float));
  for (int32_t v = 2; v < h - 1; ++v)
    for (int32_t u = 1; u < w - 2; ++u)
      b[(size_t)v * w + u] = a[(size_t)v * w + u - 1] * 2.0f +
                             a[(size_t)(v + 1) * w + u + 2] * 3.0f -
                             a[(size_t)(v - 2) * w + u];
  for (int32_t q = 2; q < h - 2; ++q)
    for (int32_t p = 2; p < w - 2; ++p)
      c[(size_t)q * w + p] = b[(size_t)q * w + p - 1] -
                             b[(size_t)(q + 1) * w + p] * 0.5f +
                             a[(size_t)q * w + p - 1];
  free(b);
  return 0;
}

/* ---- the rest of the reference's 2-D corpus, by hand from the DSL text ------
 * Boxes follow frt/host.py:565-577 with its always-true `tensor.is_output`:
 * EVERY tensor is defined on the box of its window relative to the program
 * inputs (Minkowski closure of the taps along the producer chain). */

/* reference tests/src/sobel2d.soda:5-14
 *   local int16:  mag_x = (img(1,-1) - img(-1,-1)) + (img(1,0) - img(-1,0)) * 3
 *                       + (img(1,1) - img(-1,1))
 *   local uint16: mag_y = (img(-1,1) - img(-1,-1)) + (img(0,1) - img(0,-1)) * 3
 *                       + (img(1,1) - img(1,-1))
 *   output uint16: mag = 65535 - (mag_x * mag_x + mag_y * mag_y)
 * all three on [1,w-1) x [1,h-1); int arithmetic, wrap on every store. */
int kat_sobel2d(const int16_t* img, uint16_t* mag, int32_t w, int32_t h) {
  size_t cells = (size_t)w * h;
  int16_t* mx = calloc(cells, sizeof(int16_t));
  uint16_t* my = calloc(cells, sizeof(uint16_t));
  if (!mx || !my) return 1;
  memset(mag, 0, cells * sizeof(uint16_t));
  for (int32_t q = 1; q < h - 1; ++q)
    for (int32_t p = 1; p < w - 1; ++p) {
      const int16_t* c = img + (size_t)q * w + p;
      mx[(size_t)q * w + p] = (int16_t)((c[1 - w] - c[-1 - w]) +
                                        (c[1] - c[-1]) * 3 +
                                        (c[1 + w] - c[-1 + w]));
      my[(size_t)q * w + p] = (uint16_t)((c[-1 + w] - c[-1 - w]) +
                                         (c[w] - c[-w]) * 3 +
                                         (c[1 + w] - c[1 - w]));
    }
  for (int32_t q = 1; q < h - 1; ++q)
    for (int32_t p = 1; p < w - 1; ++p) {
      size_t i = (size_t)q * w + p;
      mag[i] = (uint16_t)(65535 - (mx[i] * mx[i] + my[i] * my[i]));
    }
  free(mx);
  free(my);
  return 0;
}

/* reference tests/src/seidel2d.soda:9-14 (iterate: 2 as shipped)
 *   output = (input(-1,-1) + input(-1,0) + input(-1,1) + input(0,-1) +
 *             input(0,0) + input(0,1) + input(1,-1) + input(1,0) + input(1,1))
 *            * .1111111f
 * note the order: the FIRST index is dimension 0 (the fastest). */
int kat_seidel2d(const float* in, float* out, int32_t w, int32_t h,
                 int32_t iterate) {
  size_t cells = (size_t)w * h;
  float* tmp[2] = {calloc(cells, sizeof(float)), calloc(cells, sizeof(float))};
  if (!tmp[0] || !tmp[1]) return 1;
  const float* src = in;
  memset(out, 0, cells * sizeof(float));
  for (int32_t it = 0; it < iterate; ++it) {
    float* dst = it == iterate - 1 ? out : tmp[it & 1];
    int32_t r = it + 1;
    for (int32_t q = r; q < h - r; ++q)
      for (int32_t p = r; p < w - r; ++p) {
        const float* c = src + (size_t)q * w + p;
        dst[(size_t)q * w + p] =
            (c[-1 - w] + c[-1] + c[-1 + w] + c[-w] + c[0] + c[w] + c[1 - w] +
             c[1] + c[1 + w]) * .1111111f;
      }
    src = dst;
  }
  free(tmp[0]);
  free(tmp[1]);
  return 0;
}

/* reference tests/src/denoise2d.soda:8-33: two inputs f, u; locals diff_u,
 * diff_d, diff_l, diff_r, g, r0, r1.  Boxes: diff_u y>=1; diff_d y<h-1;
 * diff_l x>=1; diff_r x<w-1; g [1,w-1)x[1,h-1); r0, r1 everywhere; output
 * [2,w-2)x[2,h-2).  `sqrt` of a float is the float overload. */
int kat_denoise2d(const float* f, const float* u, float* out, int32_t w,
                  int32_t h) {
  size_t cells = (size_t)w * h;
  float* g = calloc(cells, sizeof(float));
  float* r1 = calloc(cells, sizeof(float));
  if (!g || !r1) return 1;
  memset(out, 0, cells * sizeof(float));
  for (int32_t q = 1; q < h - 1; ++q)
    for (int32_t p = 1; p < w - 1; ++p) {
      size_t i = (size_t)q * w + p;
      float du = u[i] - u[i - w], dd = u[i] - u[i + w];
      float dl = u[i] - u[i - 1], dr = u[i] - u[i + 1];
      g[i] = 1.0f / sqrtf(1.0f + du * du + dd * dd + dl * dl + dr * dr);
    }
  for (size_t i = 0; i < cells; ++i) {
    float r0 = u[i] * f[i] * 4.9f;
    r1[i] = (r0 * (2.5f + r0 * (10.2f + r0))) *
            (4.3f + r0 * (5.4f + r0 * (6.3f + r0)));
  }
  for (int32_t q = 2; q < h - 2; ++q)
    for (int32_t p = 2; p < w - 2; ++p) {
      size_t i = (size_t)q * w + p;
      out[i] = (u[i] + 7.7f * (u[i + w] * g[i + w] + u[i - w] * g[i - w] +
                               u[i - 1] * g[i - 1] + u[i + 1] * g[i + 1] +
                               5.7f * f[i] * r1[i])) *
               (11.1f + 7.7f * (g[i + w] + g[i - w] + g[i - 1] + g[i + 1] +
                                5.7f));
    }
  free(g);
  free(r1);
  return 0;
}

/* reference tests/src/erosion.soda:5-15
 *   local int16:  tmp(0, 9)    = min(input(0, 0) ... input(0, 18))
 *   output int16: output(9, 0) = min(tmp(0, 0) ... tmp(18, 0))
 * tmp on all x, y in [9,h-9); output on [9,w-9) x [9,h-9). */
int kat_erosion(const int16_t* in, int16_t* out, int32_t w, int32_t h) {
  size_t cells = (size_t)w * h;
  int16_t* tmp = calloc(cells, sizeof(int16_t));
  if (!tmp) return 1;
  memset(out, 0, cells * sizeof(int16_t));
  for (int32_t q = 9; q < h - 9; ++q)
    for (int32_t p = 0; p < w; ++p) {
      int m = in[(size_t)(q - 9) * w + p];
      for (int j = 1; j <= 18; ++j) {
        int v = in[(size_t)(q - 9 + j) * w + p];
        if (v < m) m = v;
      }
      tmp[(size_t)q * w + p] = (int16_t)m;
    }
  for (int32_t q = 9; q < h - 9; ++q)
    for (int32_t p = 9; p < w - 9; ++p) {
      int m = tmp[(size_t)q * w + p - 9];
      for (int i = 1; i <= 18; ++i) {
        int v = tmp[(size_t)q * w + p - 9 + i];
        if (v < m) m = v;
      }
      out[(size_t)q * w + p] = (int16_t)m;
    }
  free(tmp);
  return 0;
}

/* reference tests/src/xcorr.soda:5-15
 *   local int16:  tmp1(0, 9) = input(0, 0) + ... + input(0, 18)
 *   local int16:  tmp2(9, 0) = tmp1(0, 0) + ... + tmp1(18, 0)
 *   output int16: tmp3(0, 0) = (int32(tmp2(0,0)) - input(0,0)) * input(0,0) / 256
 * int arithmetic, wrap to int16 on every store, `/` truncates. */
int kat_xcorr(const int16_t* in, int16_t* out, int32_t w, int32_t h) {
  size_t cells = (size_t)w * h;
  int16_t* t1 = calloc(cells, sizeof(int16_t));
  int16_t* t2 = calloc(cells, sizeof(int16_t));
  if (!t1 || !t2) return 1;
  memset(out, 0, cells * sizeof(int16_t));
  for (int32_t q = 9; q < h - 9; ++q)
    for (int32_t p = 0; p < w; ++p) {
      int s = 0;
      for (int j = 0; j <= 18; ++j) s += in[(size_t)(q - 9 + j) * w + p];
      t1[(size_t)q * w + p] = (int16_t)s;
    }
  for (int32_t q = 9; q < h - 9; ++q)
    for (int32_t p = 9; p < w - 9; ++p) {
      int s = 0;
      for (int i = 0; i <= 18; ++i) s += t1[(size_t)q * w + p - 9 + i];
      t2[(size_t)q * w + p] = (int16_t)s;
    }
  for (int32_t q = 9; q < h - 9; ++q)
    for (int32_t p = 9; p < w - 9; ++p) {
      size_t i = (size_t)q * w + p;
      out[i] = (int16_t)(((int32_t)t2[i] - in[i]) * in[i] / 256);
    }
  free(t1);
  free(t2);
  return 0;
}

/* A long fp32 weighted sum as the reference REALLY evaluates it (contrast.soda,
 * 197 terms `input(dx, dy) * coef`): `inline.rebalance` (reference
 * src/soda/optimization/inline.py:175-262, always run from core.py:138) cuts
 * the terms, in textual order, into groups of 32; every group but the last is a
 * local tensor of its own (rounded to float on its own), and the statement
 * becomes  last group + local_0 + local_1 + ...  left to right.  The taps come
 * in as tables (the test reads them from the DSL text with a regular
 * expression, not with this repo's parser).  Output on [0, w - max dx) x
 * [0, h - max dy) for taps with non-negative offsets. */
int kat_rebalanced_sum(const float* in, float* out, int32_t w, int32_t h,
                       int32_t terms, const int32_t* dx, const int32_t* dy,
                       const int32_t* coef, int32_t group) {
  size_t cells = (size_t)w * h;
  int32_t mx = 0, my = 0;
  for (int32_t k = 0; k < terms; ++k) {
    if (dx[k] < 0 || dy[k] < 0) return 2;
    if (dx[k] > mx) mx = dx[k];
    if (dy[k] > my) my = dy[k];
  }
  int32_t ngroups = (terms + group - 1) / group;
  float* part = malloc(sizeof(float) * (ngroups > 0 ? ngroups : 1));
  if (!part) return 1;
  memset(out, 0, cells * sizeof(float));
  for (int32_t q = 0; q < h - my; ++q)
    for (int32_t p = 0; p < w - mx; ++p) {
      for (int32_t g = 0; g < ngroups; ++g) {
        int32_t k0 = g * group, k1 = k0 + group < terms ? k0 + group : terms;
        float s = in[(size_t)(q + dy[k0]) * w + p + dx[k0]] * coef[k0];
        for (int32_t k = k0 + 1; k < k1; ++k)
          s = s + in[(size_t)(q + dy[k]) * w + p + dx[k]] * coef[k];
        part[g] = s;
      }
      float s = part[ngroups - 1];
      for (int32_t g = 0; g < ngroups - 1; ++g) s = s + part[g];
      out[(size_t)q * w + p] = s;
    }
  free(part);
  return 0;
}

/* ---- the reference's two remaining 3-D programs, by hand from the DSL text --
 * Same box rule as above (frt/host.py:565-577): every tensor lives on the box
 * of its window relative to the program inputs and is zero outside it.  With
 * these two, every program of the reference's corpus has a pin that never
 * touches this repo's parser, IR or generators. */

/* reference tests/src/jacobi3d.soda:5-11 (iterate: 2 as shipped)
 *   t0(0,0,0) = (t1(0,0,0) + t1(1,0,0) + t1(-1,0,0) + t1(0,1,0) + t1(0,-1,0)
 *                + t1(0,0,1) + t1(0,0,-1)) * 0.142857142f
 * the sum left to right as written; iteration r lives on [r, n - r)^3. */
int kat_jacobi3d(const float* in, float* out, int32_t nx, int32_t ny,
                 int32_t nz, int32_t iterate) {
  const ptrdiff_t sy = nx, sz = (ptrdiff_t)nx * ny;
  const size_t cells = (size_t)sz * nz;
  float* tmp[2] = {calloc(cells, sizeof(float)), calloc(cells, sizeof(float))};
  if (!tmp[0] || !tmp[1]) return 1;
  const float* src = in;
  memset(out, 0, cells * sizeof(float));
  for (int32_t it = 0; it < iterate; ++it) {
    float* dst = it == iterate - 1 ? out : tmp[it & 1];
    const int32_t r = it + 1;
    for (int32_t z = r; z < nz - r; ++z)
      for (int32_t y = r; y < ny - r; ++y)
        for (int32_t x = r; x < nx - r; ++x) {
          const float* c = src + z * sz + y * sy + x;
          dst[z * sz + y * sy + x] =
              (c[0] + c[1] + c[-1] + c[sy] + c[-sy] + c[sz] + c[-sz]) *
              0.142857142f;
        }
    src = dst;
  }
  free(tmp[0]);
  free(tmp[1]);
  return 0;
}

/* reference tests/src/denoise3d.soda:8-29: two inputs f, u; locals diff_u
 * (u - u(0,-1,0), box y >= 1), diff_d (y < ny-1), diff_l (x >= 1), diff_r
 * (x < nx-1), diff_i (u - u(0,0,-1), z >= 1), diff_o (z < nz-1); g on
 * [1, n-1)^3 (it reads all six where each is defined); r0, r1 everywhere;
 * output on [2, n-2)^3.  `1.0f/0.03f` is one float constant; `sqrt` of a float
 * is the float overload; sums and products associate left to right as written:
 *   g  = 1.0f / sqrt(0.00005f + du*du + dd*dd + dl*dl + dr*dr + di*di + do*do)
 *   r0 = u * f * (1.0f/0.03f)
 *   r1 = (r0*(2.38944f + r0*(0.950037f + r0)))
 *        / (4.65314f + r0*(2.57541f + r0*(1.48937f + r0)))
 *   output = (u + 5.0f*(u(1,0,0)*g(1,0,0) + u(-1,0,0)*g(-1,0,0)
 *             + u(0,1,0)*g(0,1,0) + u(0,-1,0)*g(0,-1,0) + u(0,0,1)*g(0,0,1)
 *             + u(0,0,-1)*g(0,0,-1) + (1.0f/0.03f)*f*r1))
 *          / (1.0f + 5.0f*(g(1,0,0) + g(-1,0,0) + g(0,1,0) + g(0,-1,0)
 *             + g(0,0,1) + g(0,0,-1) + (1.0f/0.03f))) */
int kat_denoise3d(const float* f, const float* u, float* out, int32_t nx,
                  int32_t ny, int32_t nz) {
  const ptrdiff_t sy = nx, sz = (ptrdiff_t)nx * ny;
  const size_t cells = (size_t)sz * nz;
  float* g = calloc(cells, sizeof(float));
  float* r1 = calloc(cells, sizeof(float));
  if (!g || !r1) return 1;
  const float k = 1.0f / 0.03f;
  memset(out, 0, cells * sizeof(float));
  for (int32_t z = 1; z < nz - 1; ++z)
    for (int32_t y = 1; y < ny - 1; ++y)
      for (int32_t x = 1; x < nx - 1; ++x) {
        const ptrdiff_t i = z * sz + y * sy + x;
        const float du = u[i] - u[i - sy], dd = u[i] - u[i + sy];
        const float dl = u[i] - u[i - 1], dr = u[i] - u[i + 1];
        const float di = u[i] - u[i - sz], dq = u[i] - u[i + sz];
        g[i] = 1.0f / sqrtf(0.00005f + du * du + dd * dd + dl * dl + dr * dr +
                            di * di + dq * dq);
      }
  for (size_t i = 0; i < cells; ++i) {
    const float r0 = u[i] * f[i] * k;
    r1[i] = (r0 * (2.38944f + r0 * (0.950037f + r0))) /
            (4.65314f + r0 * (2.57541f + r0 * (1.48937f + r0)));
  }
  for (int32_t z = 2; z < nz - 2; ++z)
    for (int32_t y = 2; y < ny - 2; ++y)
      for (int32_t x = 2; x < nx - 2; ++x) {
        const ptrdiff_t i = z * sz + y * sy + x;
        out[i] = (u[i] + 5.0f * (u[i + 1] * g[i + 1] + u[i - 1] * g[i - 1] +
                                 u[i + sy] * g[i + sy] + u[i - sy] * g[i - sy] +
                                 u[i + sz] * g[i + sz] + u[i - sz] * g[i - sz] +
                                 k * f[i] * r1[i])) /
                 (1.0f + 5.0f * (g[i + 1] + g[i - 1] + g[i + sy] + g[i - sy] +
                                 g[i + sz] + g[i - sz] + k));
      }
  free(g);
  free(r1);
  return 0;
}
