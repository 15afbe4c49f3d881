// jacobi2d.soda AS SHIPPED (float, burst width 64, tile 32, iterate 2, one
// DRAM bank per tensor) behind the reference's host logic (tests/host/
// frt_host.h, transcribed from reference src/soda/codegen/frt/host.py) on a
// grid of FOUR tiles.  Constants as the generator would print them (reference
// src/tests/test_core.py numbers): overall window of two iterations 5 x 5,
// offset (2, 2), kStencilDistance 130.  Input p + q: two Jacobi sweeps keep a
// linear field within rounding, so cell (x, y) of tile t must read
// (x - t) + y -- tile t's stream holds the input shifted by t columns, the
// upstream scatter / gather step mismatch frt_host.h keeps (tile 0: p + q
// itself) -- by the reference's own comparison rule (frt/host.py:634-657,
// 1e-5 relative); every cell the gather does not name keeps the caller's
// value.
#include <cmath>
#include <cstdio>

#include "frt_host.h"

extern "C" void jacobi2d_kernel(void* bank_0_t0, void* bank_0_t1,
                                uint64_t coalesced_data_num);

int main() {
  FrtHost<float, 2> host;
  host.extent[0] = 100;
  host.extent[1] = 37;
  host.tile_size[0] = 32;
  host.stencil_dim[0] = host.stencil_dim[1] = 5;
  host.window_offset[0] = host.window_offset[1] = 2;
  host.stencil_distance = 130;
  host.burst_width = 64;
  host.bank_count_in = host.bank_count_out = 1;
  const int n0 = host.extent[0], n1 = host.extent[1];
  std::vector<float> in((size_t)n0 * n1), out((size_t)n0 * n1, -7.0f);
  for (int q = 0; q < n1; ++q)
    for (int p = 0; p < n0; ++p) in[(size_t)q * n0 + p] = (float)(p + q);
  int rc = host.Run(in.data(), out.data(),
                    [](const std::vector<float*>& o, const std::vector<float*>& i,
                       uint64_t cycles) { jacobi2d_kernel(o[0], i[0], cycles); },
                    0.0f);
  if (rc) return rc;
  if (host.tile_count != 4) return 3;
  long bad = 0, checked = 0;
  for (int y = 0; y < n1; ++y)
    for (int x = 0; x < n0; ++x) {
      const float got = out[(size_t)y * n0 + x];
      // which tile's gather names (x, y), if any (tiles step by 32 - 5 + 1)
      int tile = -1;
      if (y >= 2 && y < n1 - 2)
        for (int t = 0; t < 4; ++t) {
          const int actual = t == 3 ? n0 - 28 * t : 32;
          const int i = x - 28 * t;
          if (i >= 2 && i < actual - 2) tile = t;
        }
      if (tile < 0) {
        bad += got != -7.0f;
        continue;
      }
      const double want = (double)(x - tile) + y;
      const double d2 = (got - want) * (got - want);
      ++checked;
      if (d2 > 1e-10 && d2 / (want * want) > 1e-10) ++bad;       // :634-657
    }
  printf(bad ? "FAIL %ld cells\n" : "OK %ld cells in 4 tiles\n",
         bad ? bad : checked);
  return bad != 0;
}
