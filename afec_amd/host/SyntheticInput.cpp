// afec_amd/host/SyntheticInput.cpp -- the synthetic input of BASELINE.json configs[1] ("x = U(-1,1) from
// std::mt19937(1234), same generator on both sides"): the CPU-baseline driver built from the reference's own objects (its `time` mode)
// draws its frames from std::mt19937 through std::uniform_real_distribution<float>(-1, 1); bench.py and the C2 parity
// test fill their buffers with the same stream through this entry point, so "MT19937(1234)" in a workload string means
// this generator and not another library's seeding of the same twister.
#include <cstdint>
#include <random>

#include "Crawler.h"

extern "C" void afec_fill_uniform_mt19937(float* dst, int64_t n, uint32_t seed) {
  std::mt19937 Generator(seed);
  std::uniform_real_distribution<float> Uniform(-1.0f, 1.0f);
  for (int64_t i = 0; i < n; ++i) dst[i] = Uniform(Generator);
}
