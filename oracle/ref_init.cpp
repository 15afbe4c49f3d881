// CPU ORACLE, input initialisation -- TEST INFRASTRUCTURE ONLY.
//
// The reference's generated test harness fills float inputs with
//   std::default_random_engine generator;                 (default seed)
//   std::uniform_real_distribution<double> distribution(0.0, 1.0);
// drawing one value per cell with the LAST dimension outermost, i.e. in memory
// order, and converting to the tensor type on store; integer inputs get
// p + q (+ r) (reference src/soda/codegen/frt/host.py:503-528).  Using
// <random> here IS that recipe (libstdc++), so inputs equal what the reference
// harness would have used.
#include <cstdint>
#include <random>

extern "C" {

void ref_init_float(float* data, int64_t cells) {
  std::default_random_engine generator;
  std::uniform_real_distribution<double> distribution(0.0, 1.0);
  for (int64_t i = 0; i < cells; ++i) data[i] = distribution(generator);
}

void ref_init_double(double* data, int64_t cells) {
  std::default_random_engine generator;
  std::uniform_real_distribution<double> distribution(0.0, 1.0);
  for (int64_t i = 0; i < cells; ++i) data[i] = distribution(generator);
}

}  // extern "C"
