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
                                                        float* __restrict__ keys, long N, float thresh, uint32_t seed, uint32_t stage) {
  // the stream of one (seed, stage) pair: mixed non-linearly, so that neither seed + 1 nor the second draw of the same seed is a
  // shifted copy of this one (round 2 hashed ctr * A + seed * B + C and ran the second draw at seed + 1)
  const uint32_t stream = fmix32(seed ^ (stage * 0x9E3779B9u));
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
    float w = p[i];
    if (thresh >= 0.f && w > thresh) w = 1.f;
    const uint32_t ctr = (uint32_t)(counter ? counter[i] : i);             // the item's identity, not its position
    const uint32_t h = fmix32(stream + ctr * 0x9E3779B1u);
    const float u = ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);        // (0, 1), 24 bits, exact in fp32
    const float e = -logf(u);
    keys[i] = (w > 0.f) ? w / e : 0.f;
  }
}

// nearest reference point of every query point (2-D, squared Euclidean distance, LOWEST index on exact ties) —
// RegressionMatcher.match_keypoints (matcher.py:576-591) builds the full |x_A| x |x_B| cdist matrix and compares it with its
// row / column minima; the mutual-nearest-neighbour test only needs the two arg-min vectors.
constexpr int NN_TILE = 1024;
__global__ __launch_bounds__(256) void nn_argmin_kernel(const float* __restrict__ q, const float* __restrict__ r, int* __restrict__ idx,
                                                        int NQ, int NR) {
  __shared__ float sx[NN_TILE], sy[NN_TILE];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const float qx = i < NQ ? q[2 * i] : 0.f, qy = i < NQ ? q[2 * i + 1] : 0.f;
  float best = INFINITY;
  int bi = 0;
  for (int j0 = 0; j0 < NR; j0 += NN_TILE) {
    __syncthreads();
    for (int t = threadIdx.x; t < NN_TILE && j0 + t < NR; t += blockDim.x) {
      sx[t] = r[2 * (size_t)(j0 + t)];
      sy[t] = r[2 * (size_t)(j0 + t) + 1];
    }
    __syncthreads();
    const int cnt = min(NN_TILE, NR - j0);
    for (int t = 0; t < cnt; ++t) {
      const float dx = qx - sx[t], dy = qy - sy[t];
      const float d = __builtin_fmaf(dy, dy, dx * dx);
      if (d < best) { best = d; bi = j0 + t; }          // strict: the first (lowest) index wins ties
    }
  }
  if (i < NQ) idx[i] = bi;
}

}  // namespace
}  // namespace roma

using namespace roma;

extern "C" int roma_nn_argmin(const float* q, const float* r, int* idx, int NQ, int NR, void* stream) {
  ROMA_REQUIRE(q && r && idx, ROMA_E_ARG, "roma_nn_argmin: null pointer");
  ROMA_REQUIRE(NQ > 0 && NR > 0, ROMA_E_SHAPE, "roma_nn_argmin: empty point set");
  hipLaunchKernelGGL(nn_argmin_kernel, dim3((NQ + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), q, r, idx, NQ, NR);
  ROMA_CHECK_LAUNCH();
}

extern "C" int roma_race_keys(const float* p, const long* counter, float* keys, long N, float thresh, unsigned seed, unsigned stage,
                              void* stream) {
  ROMA_REQUIRE(p && keys, ROMA_E_ARG, "roma_race_keys: null pointer");
  ROMA_REQUIRE(N > 0, ROMA_E_SHAPE, "roma_race_keys: N must be positive");
  long g = (N + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(race_keys_kernel, dim3((unsigned)g), dim3(256), 0, static_cast<hipStream_t>(stream), p, counter, keys, N, thresh, (uint32_t)seed,
                     (uint32_t)stage);
  ROMA_CHECK_LAUNCH();
}
