// Calls `blur_kernel` the way the reference's generated host does when built
// with -DSODA_CPP_BINDING (reference src/soda/codegen/frt/host.py): sizes
// (:124-178), scatter of the caller's array into the burst-aligned stream
// (:181-249), the call itself with the OUTPUT bank first (:44-59, :282-289),
// gather of the valid region (:340-427).  Everything below is written from
// that text for blur.soda (uint16, burst width 256, tile 2000, one DRAM bank
// per tensor); nothing here knows about the GPU.  Input p + q, the reference
// harness's integer init (:519): the closed form on the valid box is p + q + 2.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

// frt/host.py:44-59: ap_uint<256>* ports; C linkage carries no types
extern "C" void blur_kernel(uint64_t* bank_0_blur_y, uint64_t* bank_0_input,
                            uint64_t coalesced_data_num);

int main() {
  const int32_t kStencilDim0 = 3, kStencilDim1 = 3, kStencilDistance = 4002;
  const int32_t tile_size_0 = 2000, burst_width = 256, kWidth = 16;
  const int32_t extent[2] = {2000, 40};
  const int32_t elem_count_per_cycle = burst_width / kWidth * 1;      // :120
  const int32_t tile_count_dim_0 =
      (extent[0] - kStencilDim0) / (tile_size_0 - kStencilDim0 + 1) + 1;  // :124
  const int64_t tile_count = tile_count_dim_0;
  const int64_t elem_count_per_tile = (int64_t)tile_size_0 * extent[1];  // :137
  const int64_t cycle_count_per_tile =
      (elem_count_per_tile - 1) / elem_count_per_cycle + 1;
  const int64_t elem_count_aligned_per_tile =
      cycle_count_per_tile * elem_count_per_cycle;                     // :142
  const int64_t tail = ((kStencilDistance - 1) / elem_count_per_cycle + 1) *
                       elem_count_per_cycle;
  const int64_t buf_elems = tile_count * elem_count_aligned_per_tile + tail;
  uint16_t* buf_in = (uint16_t*)aligned_alloc(4096, (buf_elems * 2 + 4095) / 4096 * 4096);
  uint16_t* buf_out = (uint16_t*)aligned_alloc(4096, (buf_elems * 2 + 4095) / 4096 * 4096);
  if (tile_count != 1 || !buf_in || !buf_out) return 2;
  for (int64_t i = 0; i < buf_elems; ++i) { buf_in[i] = 0; buf_out[i] = 0xabcd; }
  std::vector<uint16_t> in((size_t)extent[0] * extent[1]);
  std::vector<uint16_t> out((size_t)extent[0] * extent[1], 0x1234);
  for (int q = 0; q < extent[1]; ++q)
    for (int p = 0; p < extent[0]; ++p) in[(size_t)q * extent[0] + p] = (uint16_t)(p + q);
  // scatter (:181-249), one tile, produce offset 0
  for (int32_t j = 0; j < extent[1]; ++j)
    for (int32_t i = 0; i < extent[0]; ++i) {
      const int64_t off = i + (int64_t)j * tile_size_0;
      const int64_t tiled = off / elem_count_per_cycle * elem_count_per_cycle +
                            off % elem_count_per_cycle;
      buf_in[tiled] = in[(size_t)j * extent[0] + i];
    }
  const uint64_t cycle_count =
      ((elem_count_per_tile * tile_count + kStencilDistance - 1) /
           elem_count_per_cycle + 1);                                  // :272-276
  blur_kernel((uint64_t*)buf_out, (uint64_t*)buf_in, cycle_count);    // :282-289
  // gather (:340-427): stencil offset = kStencilDistance - serialize(offset) = 4002
  const int32_t stencil_offset = 4002;
  for (int32_t j = 0; j < extent[1] - (kStencilDim1 - 1); ++j)
    for (int32_t i = 0; i < extent[0] - (kStencilDim0 - 1); ++i)
      out[(size_t)j * extent[0] + i] =
          buf_out[i + (int64_t)j * tile_size_0 + stencil_offset];
  long bad = 0;
  for (int q = 0; q < extent[1]; ++q)
    for (int p = 0; p < extent[0]; ++p) {
      const bool valid = p < extent[0] - 2 && q < extent[1] - 2;
      const uint16_t want = valid ? (uint16_t)(p + q + 2) : (uint16_t)0x1234;
      bad += out[(size_t)q * extent[0] + p] != want;
    }
  printf(bad ? "FAIL %ld cells\n" : "OK %ld\n", bad);
  free(buf_in);
  free(buf_out);
  return bad != 0;
}
