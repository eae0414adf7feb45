// Sampling helpers for RegressionMatcher.sample — reference: romatch/models/matcher.py:468-495.
// Both draws of the reference are torch.multinomial(weights, k, replacement=False); sampling k items without replacement
// with probabilities proportional to w is an "exponential race": key_i = w_i / E_i, E_i ~ Exp(1) i.i.d., keep the k largest
// keys (this is also what ATen's multinomial does internally).  The kernel fuses the certainty threshold of the "threshold"
// sample modes (matcher.py:474-477), a counter-based uniform (a pure function of (seed, index): reproducible on a CPU
// oracle, independent of launch geometry) and the key; the selection itself is a library top-k.
#include "common.h"

namespace roma {
namespace {

__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}

__global__ __launch_bounds__(256) void race_keys_kernel(const float* __restrict__ p, const long* __restrict__ counter,
                                                        float* __restrict__ keys, long N, float thresh, uint32_t seed) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
    float w = p[i];
    if (thresh >= 0.f && w > thresh) w = 1.f;
    const uint32_t ctr = (uint32_t)(counter ? counter[i] : i);             // the item's identity, not its position
    const uint32_t h = fmix32(ctr * 0x9E3779B1u + seed * 0x85EBCA77u + 0x165667B1u);
    const float u = ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);        // (0, 1), 24 bits, exact in fp32
    const float e = -logf(u);
    keys[i] = (w > 0.f) ? w / e : 0.f;
  }
}

}  // namespace
}  // namespace roma

using namespace roma;

extern "C" int roma_race_keys(const float* p, const long* counter, float* keys, long N, float thresh, unsigned seed, void* stream) {
  ROMA_REQUIRE(p && keys, ROMA_E_ARG, "roma_race_keys: null pointer");
  ROMA_REQUIRE(N > 0, ROMA_E_SHAPE, "roma_race_keys: N must be positive");
  long g = (N + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(race_keys_kernel, dim3((unsigned)g), dim3(256), 0, static_cast<hipStream_t>(stream), p, counter, keys, N, thresh, (uint32_t)seed);
  ROMA_CHECK_LAUNCH();
}
