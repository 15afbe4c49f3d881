// The body of the reference's generated host function, `soda::app::<app>()`
// built with -DSODA_CPP_BINDING, transcribed ONCE for any program from the
// text of reference src/soda/codegen/frt/host.py (line numbers below) and
// docs/data-layout.md:62-127 -- sizes, the scatter of the caller's array into
// tiled, burst-aligned, bank-interleaved streams, the <app>_kernel call with
// all output banks first, the gather of the valid region.  Nothing here knows
// about the GPU, libsoda_hip.so, soda_amd.stream.WireLayout or the oracle: the
// programs that include it (tests/host/*_wire_main.cpp) supply the numbers the
// generator would print as constants (kStencilDim, the window offset,
// kStencilDistance: the reference's own known answers, src/tests/test_core.py
// and SURVEY.md 8c) and check a closed form.
//
// Kept as upstream has it, on purpose: the scatter places tile t at original
// coordinate t * (tile_size - kStencilDim) (:224-228) while the tile count
// (:124-128) and the gather (:389-393) step by tile_size - kStencilDim + 1.
// With more than one tile the stream of tile t therefore holds the input
// shifted by t cells against where the gather files its results; the mains
// state their expectations accordingly.
#ifndef TESTS_HOST_FRT_HOST_H_
#define TESTS_HOST_FRT_HOST_H_

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <vector>

template <class T, int D>
struct FrtHost {
  int32_t extent[D];          // var_<in>_extent
  int32_t tile_size[D];       // tile_size_<d>, d < D - 1
  int32_t stencil_dim[D];     // kStencilDim<d>
  int32_t window_offset[D];   // get_stencil_window_offset of the overall window
  int64_t stencil_distance;   // kStencilDistance
  int burst_width;            // bits
  int bank_count_in, bank_count_out;

  static int64_t round_up(int64_t a, int64_t b) { return ((a - 1) / b + 1) * b; }

  // out_banks / in_banks: one pointer per bank; the callee is <app>_kernel
  typedef std::function<void(const std::vector<T*>& out_banks,
                             const std::vector<T*>& in_banks,
                             uint64_t cycle_count)> Kernel;

  int64_t tile_count_dim[D];
  int64_t tile_count, aligned_i, aligned_o;

  int Run(const T* in, T* out, const Kernel& kernel, T fill) {
    const int width = 8 * (int)sizeof(T);
    const int epc_in = burst_width / width * bank_count_in;      // :118-121
    const int epc_out = burst_width / width * bank_count_out;
    tile_count = 1;
    for (int d = 0; d < D - 1; ++d) {                             // :122-126
      tile_count_dim[d] = (extent[d] - stencil_dim[d] + 1 - 1) /
                              (tile_size[d] - stencil_dim[d] + 1) + 1;
      tile_count *= tile_count_dim[d];
    }
    int64_t elem_count_per_tile = extent[D - 1];                  // :134-137
    for (int d = 0; d < D - 1; ++d) elem_count_per_tile *= tile_size[d];
    const int64_t cycle_count_per_tile = (elem_count_per_tile - 1) / epc_in + 1;
    aligned_i = cycle_count_per_tile * epc_in;                    // :140-143
    aligned_o = cycle_count_per_tile * epc_out;
    const int64_t elems_in =                                      // :147-160
        tile_count * aligned_i + round_up(stencil_distance, epc_in);
    const int64_t elems_out =
        tile_count * aligned_o + round_up(stencil_distance, epc_out);
    std::vector<T*> buf_in, buf_out;                              // :163-178
    auto alloc = [&](int64_t elems, int banks, std::vector<T*>* bufs) {
      const int64_t bytes = elems / banks * (int64_t)sizeof(T);
      for (int b = 0; b < banks; ++b) {
        T* p = static_cast<T*>(aligned_alloc(4096, round_up(bytes, 4096)));
        if (!p) return false;
        for (int64_t i = 0; i < round_up(bytes, 4096) / (int64_t)sizeof(T); ++i)
          p[i] = fill;
        bufs->push_back(p);
      }
      return true;
    };
    if (!alloc(elems_in, bank_count_in, &buf_in) ||
        !alloc(elems_out, bank_count_out, &buf_out))
      return 2;
    int64_t stride[D];
    stride[0] = 1;
    for (int d = 1; d < D; ++d) stride[d] = stride[d - 1] * extent[d - 1];

    // tiling + scatter (:181-249)
    int64_t tile_index[D] = {0};
    std::function<void(int)> tiles_in = [&](int d) {
      if (d < 0) {
        int32_t actual[D];
        for (int e = 0; e < D - 1; ++e)
          actual[e] = tile_index[e] == tile_count_dim[e] - 1
                          ? extent[e] - (tile_size[e] - stencil_dim[e] + 1) *
                                            (int32_t)tile_index[e]
                          : tile_size[e];
        actual[D - 1] = extent[D - 1];
        int32_t c[D] = {0};          // (i, j, k): coordinates in a tile
        std::function<void(int)> cells = [&](int e) {
          if (e < 0) {
            int64_t offset_in_tile = 0, mul = 1;
            for (int x = 0; x < D; ++x) {
              offset_in_tile += c[x] * mul;
              if (x < D - 1) mul *= tile_size[x];
            }
            const int64_t burst_index = offset_in_tile / epc_in;
            const int64_t burst_residue = offset_in_tile % epc_in;
            int64_t tile_linear = 0, tmul = 1;
            for (int x = 0; x < D - 1; ++x) {
              tile_linear += tmul * tile_index[x];
              tmul *= tile_count_dim[x];
            }
            const int64_t tiled_offset =
                tile_linear * aligned_i + burst_index * epc_in + burst_residue;
            int64_t original_offset = 0;
            for (int x = 0; x < D; ++x) {
              const int64_t p =                                  // :224-231
                  x < D - 1 ? tile_index[x] * (tile_size[x] - stencil_dim[x]) +
                                  c[x]
                            : c[x];
              original_offset += p * stride[x];
            }
            buf_in[tiled_offset % bank_count_in][tiled_offset / bank_count_in] =
                in[std::max<int64_t>(0, original_offset)];       // :244-247
            return;
          }
          for (c[e] = 0; c[e] < actual[e]; ++c[e]) cells(e - 1);
        };
        cells(D - 1);
        return;
      }
      for (tile_index[d] = 0; tile_index[d] < tile_count_dim[d]; ++tile_index[d])
        tiles_in(d - 1);
    };
    tiles_in(D - 2);

    int64_t per_plane = extent[D - 1];                            // :266-276
    for (int d = 0; d < D - 1; ++d) per_plane *= tile_size[d];
    const uint64_t cycle_count =
        (uint64_t)((per_plane * tile_count + stencil_distance - 1) / epc_in + 1);
    kernel(buf_out, buf_in, cycle_count);                         // :278-289

    // gather (:340-427)
    int64_t serialized_offset = window_offset[0], smul = 1;
    for (int d = 1; d < D; ++d) {
      smul *= tile_size[d - 1];
      serialized_offset += window_offset[d] * smul;
    }
    const int64_t stencil_offset = stencil_distance - serialized_offset;
    std::function<void(int)> tiles_out = [&](int d) {
      if (d < 0) {
        int32_t actual[D];
        for (int e = 0; e < D - 1; ++e)
          actual[e] = tile_index[e] == tile_count_dim[e] - 1
                          ? extent[e] - (tile_size[e] - stencil_dim[e] + 1) *
                                            (int32_t)tile_index[e]
                          : tile_size[e];
        actual[D - 1] = extent[D - 1];
        int32_t c[D] = {0};
        std::function<void(int)> cells = [&](int e) {
          if (e < 0) {
            int64_t offset_in_tile = 0, mul = 1;
            for (int x = 0; x < D; ++x) {
              offset_in_tile += c[x] * mul;
              if (x < D - 1) mul *= tile_size[x];
            }
            int64_t original_offset = 0;
            for (int x = 0; x < D; ++x) {
              const int64_t p =                                  // :389-396
                  x < D - 1
                      ? tile_index[x] * (tile_size[x] - stencil_dim[x] + 1) +
                            c[x]
                      : c[x];
              original_offset += p * stride[x];
            }
            const int64_t shifted = offset_in_tile + stencil_offset;
            const int64_t burst_index = shifted / epc_out;
            const int64_t burst_residue = shifted % epc_out;
            int64_t tile_linear = 0, tmul = 1;
            for (int x = 0; x < D - 1; ++x) {
              tile_linear += tmul * tile_index[x];
              tmul *= tile_count_dim[x];
            }
            const int64_t tiled_offset =
                tile_linear * aligned_o + burst_index * epc_out + burst_residue;
            out[original_offset] =                               // :421-424
                buf_out[tiled_offset % bank_count_out]
                       [tiled_offset / bank_count_out];
            return;
          }
          const int32_t lo = std::max(0, window_offset[e]);
          const int32_t hi =
              actual[e] - std::max(0, stencil_dim[e] - 1 - window_offset[e]);
          for (c[e] = lo; c[e] < hi; ++c[e]) cells(e - 1);
        };
        cells(D - 1);
        return;
      }
      for (tile_index[d] = 0; tile_index[d] < tile_count_dim[d]; ++tile_index[d])
        tiles_out(d - 1);
    };
    tiles_out(D - 2);
    for (T* p : buf_in) free(p);
    for (T* p : buf_out) free(p);
    return 0;
  }
};

#endif  // TESTS_HOST_FRT_HOST_H_
