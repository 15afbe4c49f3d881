// CPU stand-ins of `<app>_kernel` for the three independent wire-format hosts
// (tests/host/*_wire_main.cpp), so that the hosts' own logic -- sizes,
// scatter, gather, their closed-form expectations -- is checked on the CPU
// before a GPU sees them.  TEST INFRASTRUCTURE ONLY: the kernel contract of
// reference docs/data-layout.md:12-25 written down in the plainest way -- the
// FPGA kernel sees ONE long stream per tensor (banks interleaved element by
// element), applies the program's taps as offsets in that stream (dimension d
// of a tap weighs prod(tile_size[:d])), and emits the result for stream cell c
// at position c + kStencilDistance - serialize(window offset).
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

template <class T>
std::vector<T> Unbank(const std::vector<const T*>& banks, int64_t n) {
  std::vector<T> s((size_t)n);
  const int nb = (int)banks.size();
  for (int64_t e = 0; e < n; ++e) s[e] = banks[e % nb][e / nb];
  return s;
}

template <class T>
void Bank(const std::vector<T>& s, const std::vector<T*>& banks) {
  const int nb = (int)banks.size();
  for (int64_t e = 0; e < (int64_t)s.size(); ++e) banks[e % nb][e / nb] = s[e];
}

}  // namespace

extern "C" void jacobi2d_kernel(void* bank_0_t0, void* bank_0_t1,
                                uint64_t coalesced_data_num) {
  const int64_t n = (int64_t)coalesced_data_num * 2;     // 64 bits / float
  const float* in = static_cast<const float*>(bank_0_t1);
  float* out = static_cast<float*>(bank_0_t0);
  auto sweep = [n](const std::vector<float>& a) {
    std::vector<float> b((size_t)n, 0.0f);
    for (int64_t c = 32; c + 32 < n; ++c)
      b[c] = (a[c + 32] + a[c + 1] + a[c] + a[c - 1] + a[c - 32]) * 0.2f;
    return b;
  };
  std::vector<float> a(in, in + n);
  a = sweep(sweep(a));
  // kStencilDistance 130 - serialize((2, 2), tile 32) = 64
  for (int64_t c = 0; c + 64 < n; ++c) out[c + 64] = a[c];
}

extern "C" void heat3d_kernel(void* bank_0_out, void* bank_0_in,
                              uint64_t coalesced_data_num) {
  const int64_t n = (int64_t)coalesced_data_num * 2;
  const float* in = static_cast<const float*>(bank_0_in);
  float* out = static_cast<float*>(bank_0_out);
  auto sweep = [n](const std::vector<float>& a) {
    std::vector<float> b((size_t)n, 0.0f);
    for (int64_t c = 1024; c + 1024 < n; ++c)
      b[c] = .125f * a[c + 1] + .125f * a[c - 1] + .125f * a[c + 32] +
             .125f * a[c - 32] + .125f * a[c + 1024] + .125f * a[c - 1024] +
             .25f * a[c];
    return b;
  };
  std::vector<float> a(in, in + n);
  a = sweep(sweep(a));
  // kStencilDistance 4162 - serialize((2, 2, 2), tiles 32 x 32) = 4162 - 2114
  for (int64_t c = 0; c + 2048 < n; ++c) out[c + 2048] = a[c];
}

extern "C" void blur_kernel(void* bank_0_blur_y, void* bank_1_blur_y,
                            void* bank_0_input, void* bank_1_input,
                            uint64_t coalesced_data_num) {
  const int64_t n = (int64_t)coalesced_data_num * 32;    // 2 x 256 bits / u16
  std::vector<uint16_t> a = Unbank<uint16_t>(
      {static_cast<const uint16_t*>(bank_0_input),
       static_cast<const uint16_t*>(bank_1_input)}, n);
  std::vector<uint16_t> x((size_t)n, 0), y((size_t)n, 0), o((size_t)n, 0);
  for (int64_t c = 0; c + 4000 < n; ++c)
    x[c] = (uint16_t)((a[c] + a[c + 2000] + a[c + 4000]) / 3);
  for (int64_t c = 0; c + 2 < n; ++c)
    y[c] = (uint16_t)((x[c] + x[c + 1] + x[c + 2]) / 3);
  for (int64_t c = 0; c + 4002 < n; ++c) o[c + 4002] = y[c];   // offset (0, 0)
  Bank<uint16_t>(o, {static_cast<uint16_t*>(bank_0_blur_y),
                     static_cast<uint16_t*>(bank_1_blur_y)});
}
