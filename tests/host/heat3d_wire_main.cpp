// heat3d.soda as shipped (float, burst width 64, tiles 32 x 32, iterate 2)
// behind the reference's host logic (tests/host/frt_host.h) on 2 x 2 tiles of
// a 40 x 36 x 12 grid.  Constants as the generator would print them (reference
// src/tests/test_core.py numbers): window 5 x 5 x 5, offset (2, 2, 2),
// kStencilDistance 4162.  Input p + q + r: heat3d's coefficients are powers of
// two that add up to one, so the field is a fixed point BIT FOR BIT; cell
// (x, y, z) gathered from tile (t0, t1) must read (x - t0) + (y - t1) + z (the
// upstream scatter / gather step mismatch, frt_host.h), every other cell the
// caller's value.
#include <cstdio>

#include "frt_host.h"

extern "C" void heat3d_kernel(void* bank_0_out, void* bank_0_in,
                              uint64_t coalesced_data_num);

int main() {
  FrtHost<float, 3> host;
  host.extent[0] = 40;
  host.extent[1] = 36;
  host.extent[2] = 12;
  host.tile_size[0] = host.tile_size[1] = 32;
  for (int d = 0; d < 3; ++d) {
    host.stencil_dim[d] = 5;
    host.window_offset[d] = 2;
  }
  host.stencil_distance = 4162;
  host.burst_width = 64;
  host.bank_count_in = host.bank_count_out = 1;
  const int n0 = 40, n1 = 36, n2 = 12;
  std::vector<float> in((size_t)n0 * n1 * n2), out((size_t)n0 * n1 * n2, -7.0f);
  for (int r = 0; r < n2; ++r)
    for (int q = 0; q < n1; ++q)
      for (int p = 0; p < n0; ++p)
        in[((size_t)r * n1 + q) * n0 + p] = (float)(p + q + r);
  int rc = host.Run(in.data(), out.data(),
                    [](const std::vector<float*>& o, const std::vector<float*>& i,
                       uint64_t cycles) { heat3d_kernel(o[0], i[0], cycles); },
                    0.0f);
  if (rc) return rc;
  if (host.tile_count != 4) return 3;
  auto tile_of = [](int x, int n) {      // tiles step by 32 - 5 + 1 = 28
    for (int t = 0; t < 2; ++t) {
      const int actual = t == 1 ? n - 28 : 32;
      const int i = x - 28 * t;
      if (i >= 2 && i < actual - 2) return t;
    }
    return -1;
  };
  long bad = 0, checked = 0;
  for (int z = 0; z < n2; ++z)
    for (int y = 0; y < n1; ++y)
      for (int x = 0; x < n0; ++x) {
        const float got = out[((size_t)z * n1 + y) * n0 + x];
        const int t0 = tile_of(x, n0), t1 = tile_of(y, n1);
        if (t0 < 0 || t1 < 0 || z < 2 || z >= n2 - 2) {
          bad += got != -7.0f;
          continue;
        }
        ++checked;
        bad += got != (float)((x - t0) + (y - t1) + z);
      }
  printf(bad ? "FAIL %ld cells\n" : "OK %ld cells in 2 x 2 tiles\n",
         bad ? bad : checked);
  return bad != 0;
}
