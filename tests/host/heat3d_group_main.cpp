// Drives the generated C++ host of a multi-GPU build (sodac --hip-host
// --hip-gpus 4 --iterate 9) as a user of the reference's --frt-host would:
// one blocking call, soda::app::heat3d(ptr, extent, stride, min, ...), from one
// host thread (reference frt/host.py:62-88, 319-322).  Behind it the grid is
// cut into four slabs with halo exchanges (soda_hip_group_*); with
// SODA_HIP_VIRTUAL_GPUS=1 all four share device 0.
// Input p + q + r (the reference harness's init, frt/host.py:519, as fp32):
// heat3d's coefficients are powers of two that sum to 1, so the field is a
// fixed point bit for bit on the valid box [9, N - 9)^3; outside it the
// caller's array must be untouched (frt/host.py:357-375).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace soda { namespace app {
int heat3d(const float* var_in_ptr, const int32_t var_in_extent[3],
           const int32_t var_in_stride[3], const int32_t var_in_min[3],
           float* var_out_ptr, const int32_t var_out_extent[3],
           const int32_t var_out_stride[3], const int32_t var_out_min[3],
           const char* bitstream, const int burst_width, const int tile_size_0,
           const int tile_size_1, const int unroll_factor);
}}

int main() {
  const int nx = 64, ny = 48, nz = 96, it = 9;
  const int32_t extent[3] = {nx, ny, nz}, stride[3] = {1, nx, nx * ny},
                mn[3] = {0, 0, 0};
  std::vector<float> in((size_t)nx * ny * nz), out(in.size());
  const float mark = -12345.5f;
  for (int r = 0; r < nz; ++r)
    for (int q = 0; q < ny; ++q)
      for (int p = 0; p < nx; ++p) {
        in[((size_t)r * ny + q) * nx + p] = (float)(p + q + r);
        out[((size_t)r * ny + q) * nx + p] = mark;
      }
  for (int round = 0; round < 2; ++round) {      // the second call reuses the group
    int rc = soda::app::heat3d(in.data(), extent, stride, mn, out.data(), extent,
                               stride, mn, nullptr, 512, 32, 32, 2);
    if (rc) { printf("FAIL rc=%d\n", rc); return 1; }
  }
  long bad = 0;
  for (int r = 0; r < nz; ++r)
    for (int q = 0; q < ny; ++q)
      for (int p = 0; p < nx; ++p) {
        const bool valid = p >= it && p < nx - it && q >= it && q < ny - it &&
                           r >= it && r < nz - it;
        const float want = valid ? (float)(p + q + r) : mark;
        const float got = out[((size_t)r * ny + q) * nx + p];
        bad += memcmp(&got, &want, sizeof got) != 0;
      }
  printf(bad ? "FAIL %ld cells\n" : "OK %ld\n", bad);
  return bad != 0;
}
