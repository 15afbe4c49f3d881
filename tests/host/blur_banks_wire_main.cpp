// blur with every tensor spread over TWO DRAM banks (`input dram 0.1 uint16`,
// `output dram 2.3 uint16`; burst width 256, tile 2000): the cyclic bank
// partition of reference docs/data-layout.md:62-127 and the port order of
// frt/host.py:44-59 -- all banks of the output, then all banks of the input.
// Host logic: tests/host/frt_host.h.  Constants: window 3 x 3, offset (0, 0),
// kStencilDistance 4002.  Input p + q (the harness's integer init, :519): the
// closed form on the valid box is p + q + 2; one tile, so no shift.
#include <cstdio>

#include "frt_host.h"

extern "C" void blur_kernel(void* bank_0_blur_y, void* bank_1_blur_y,
                            void* bank_0_input, void* bank_1_input,
                            uint64_t coalesced_data_num);

int main() {
  FrtHost<uint16_t, 2> host;
  host.extent[0] = 2000;
  host.extent[1] = 40;
  host.tile_size[0] = 2000;
  host.stencil_dim[0] = host.stencil_dim[1] = 3;
  host.window_offset[0] = host.window_offset[1] = 0;
  host.stencil_distance = 4002;
  host.burst_width = 256;
  host.bank_count_in = host.bank_count_out = 2;
  const int n0 = 2000, n1 = 40;
  std::vector<uint16_t> in((size_t)n0 * n1), out((size_t)n0 * n1, 0x1234);
  for (int q = 0; q < n1; ++q)
    for (int p = 0; p < n0; ++p) in[(size_t)q * n0 + p] = (uint16_t)(p + q);
  int rc = host.Run(
      in.data(), out.data(),
      [](const std::vector<uint16_t*>& o, const std::vector<uint16_t*>& i,
         uint64_t cycles) { blur_kernel(o[0], o[1], i[0], i[1], cycles); },
      (uint16_t)0);
  if (rc) return rc;
  if (host.tile_count != 1) return 3;
  long bad = 0;
  for (int q = 0; q < n1; ++q)
    for (int p = 0; p < n0; ++p) {
      const bool valid = p < n0 - 2 && q < n1 - 2;
      const uint16_t want = valid ? (uint16_t)(p + q + 2) : (uint16_t)0x1234;
      bad += out[(size_t)q * n0 + p] != want;
    }
  printf(bad ? "FAIL %ld cells\n" : "OK two banks per tensor %ld\n", bad);
  return bad != 0;
}
