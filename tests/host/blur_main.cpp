// Drives the generated C++ host (sodac --hip-host) exactly as a user of the
// reference's --frt-host would: soda::app::blur(ptr, extent, stride, min, ...).
// Input p + q (the reference harness's integer init, frt/host.py:519); the
// closed form on the valid box is p + q + 2; outside it the caller's array
// must be untouched (frt/host.py:357-375).
#include <cstdint>
#include <cstdio>
#include <vector>

namespace soda { namespace app {
int blur(const uint16_t* var_input_ptr, const int32_t var_input_extent[2],
         const int32_t var_input_stride[2], const int32_t var_input_min[2],
         uint16_t* var_blur_y_ptr, const int32_t var_blur_y_extent[2],
         const int32_t var_blur_y_stride[2], const int32_t var_blur_y_min[2],
         const char* bitstream, const int burst_width, const int tile_size_0,
         const int unroll_factor);
}}

int main() {
  const int32_t extent[2] = {2000, 64}, stride[2] = {1, 2000}, mn[2] = {0, 0};
  std::vector<uint16_t> in(2000 * 64), out(2000 * 64, 0xabcd);
  for (int q = 0; q < 64; ++q)
    for (int p = 0; p < 2000; ++p) in[q * 2000 + p] = (uint16_t)(p + q);
  int rc = soda::app::blur(in.data(), extent, stride, mn, out.data(), extent,
                           stride, mn, nullptr, 256, 2000, 8);
  if (rc) { printf("FAIL rc=%d\n", rc); return 1; }
  long bad = 0;
  for (int q = 0; q < 64; ++q)
    for (int p = 0; p < 2000; ++p) {
      const bool valid = p < 1998 && q < 62;
      const uint16_t want = valid ? (uint16_t)(p + q + 2) : (uint16_t)0xabcd;
      bad += out[q * 2000 + p] != want;
    }
  printf(bad ? "FAIL %ld cells\n" : "OK %ld\n", bad);
  return bad != 0;
}
