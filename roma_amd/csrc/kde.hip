// Gaussian kernel density for sample() — reference: romatch/utils/kde.py:4-12 (called at matcher.py:489):
//   density[i] = sum_j exp(-|x_i - x_j|^2 / (2 std^2)),  x: (N,4) match coordinates, ref points every `down`-th row.
// The reference materialises the N x N distance matrix (3.2 GB in half for N = 40 000); here nothing is
// materialised: 320 KB in, 160 KB out, exp-bound.  A workgroup owns 64 query points x 4 interleaved quarters of each
// staged 256-point reference tile; the four partial sums are combined through LDS in a fixed order, so the result is
// bitwise reproducible (no atomics).
#include "common.h"

namespace roma {
namespace {

constexpr int QP = 64, SPLIT = 4, TILE = 256;

__global__ __launch_bounds__(256) void kde_kernel(const float4_t* __restrict__ x, float* __restrict__ density, int N, int down,
                                                  float neg_scale_log2) {
  __shared__ float4_t tile[TILE];
  __shared__ float part[SPLIT][QP];
  const int tid = threadIdx.x, qi = tid & (QP - 1), s = tid >> 6;
  const int i = blockIdx.x * QP + qi;
  const float4_t xi = i < N ? x[i] : float4_t{0, 0, 0, 0};
  const int nref = (N + down - 1) / down;
  float sum = 0.f;
  for (int j0 = 0; j0 < nref; j0 += TILE) {
    __syncthreads();
    const int j = j0 + tid;
    if (j < nref) tile[tid] = x[(size_t)j * down];
    __syncthreads();
    const int cnt = min(TILE, nref - j0);
    for (int t = s; t < cnt; t += SPLIT) {
      const float4_t r = tile[t];
      const float d0 = xi[0] - r[0], d1 = xi[1] - r[1], d2 = xi[2] - r[2], d3 = xi[3] - r[3];
      const float d2sum = __builtin_fmaf(d3, d3, __builtin_fmaf(d2, d2, __builtin_fmaf(d1, d1, d0 * d0)));
      sum += exp2f(d2sum * neg_scale_log2);
    }
  }
  part[s][qi] = sum;
  __syncthreads();
  if (s == 0 && i < N) density[i] = ((part[0][qi] + part[1][qi]) + part[2][qi]) + part[3][qi];
}

}  // namespace
}  // namespace roma

using namespace roma;

extern "C" int roma_kde_density(const float* x, float* density, int N, int down, float std, void* stream) {
  ROMA_REQUIRE(x && density, ROMA_E_ARG, "roma_kde_density: null pointer");
  ROMA_REQUIRE(N > 0 && down >= 1 && std > 0.f, ROMA_E_SHAPE, "roma_kde_density: bad arguments N=%d down=%d std=%g", N, down, std);
  ROMA_REQUIRE(aligned16(x), ROMA_E_ALIGN, "roma_kde_density: x must be 16-byte aligned");
  const float neg_scale_log2 = -1.4426950408889634f / (2.f * std * std);
  hipLaunchKernelGGL(kde_kernel, dim3((N + QP - 1) / QP), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4_t*>(x), density, N, down, neg_scale_log2);
  ROMA_CHECK_LAUNCH();
}
