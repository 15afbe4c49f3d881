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
