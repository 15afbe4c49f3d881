// The reference host's scatter and gather (tests/host/frt_host.h, transcribed
// from reference src/soda/codegen/frt/host.py) run on a layout given on the
// command line, with a "kernel" that only records: the input banks as the
// host laid them out go to stdout, the output banks are filled with their own
// stream positions, and the gathered array (cells the host never writes stay
// -1) follows.  tests/test_host.py holds oracle/frt_layout.py -- the numpy
// restatement every wire test on the GPU is driven by -- against it.
//   frt_dump D burst_width banks_in banks_out distance  then per dimension:
//   extent tile stencil_dim window_offset
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "frt_host.h"

template <int D>
static int run(char** a) {
  FrtHost<int32_t, D> host;
  host.burst_width = atoi(a[0]);
  host.bank_count_in = atoi(a[1]);
  host.bank_count_out = atoi(a[2]);
  host.stencil_distance = atoll(a[3]);
  int64_t cells = 1;
  for (int d = 0; d < D; ++d) {
    host.extent[d] = atoi(a[4 + 4 * d]);
    host.tile_size[d] = atoi(a[5 + 4 * d]);
    host.stencil_dim[d] = atoi(a[6 + 4 * d]);
    host.window_offset[d] = atoi(a[7 + 4 * d]);
    cells *= host.extent[d];
  }
  std::vector<int32_t> in(cells), out(cells, -1);
  for (int64_t i = 0; i < cells; ++i) in[i] = (int32_t)(i * 7 + 3);
  const int width = 32;
  auto kernel = [&](const std::vector<int32_t*>& out_banks,
                    const std::vector<int32_t*>& in_banks, uint64_t cycles) {
    const int epc_in = host.burst_width / width * host.bank_count_in;
    const int epc_out = host.burst_width / width * host.bank_count_out;
    const int64_t elems_in = host.tile_count * host.aligned_i +
        FrtHost<int32_t, D>::round_up(host.stencil_distance, epc_in);
    const int64_t elems_out = host.tile_count * host.aligned_o +
        FrtHost<int32_t, D>::round_up(host.stencil_distance, epc_out);
    printf("cycles %llu\n", (unsigned long long)cycles);
    for (int b = 0; b < host.bank_count_in; ++b) {
      printf("in %d", b);
      for (int64_t j = 0; j < elems_in / host.bank_count_in; ++j)
        printf(" %d", in_banks[b][j]);
      printf("\n");
    }
    for (int b = 0; b < host.bank_count_out; ++b)
      for (int64_t j = 0; j < elems_out / host.bank_count_out; ++j)
        out_banks[b][j] = (int32_t)(1000000 + j * host.bank_count_out + b);
  };
  const int rc = host.Run(in.data(), out.data(), kernel, 0);
  if (rc) return rc;
  printf("out");
  for (int64_t i = 0; i < cells; ++i) printf(" %d", out[i]);
  printf("\n");
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const int D = atoi(argv[1]);
  if (argc != 6 + 4 * D) return 2;
  if (D == 2) return run<2>(argv + 2);
  if (D == 3) return run<3>(argv + 2);
  return 2;
}
