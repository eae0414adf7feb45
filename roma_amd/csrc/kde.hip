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

__device__ __forceinline__ float r16(float v) { return (float)(half_t)v; }      // round to fp16, keep as float

// One term of the reference's half=True arithmetic (kde.py:6-12 with x.half()): torch.cdist on fp16 takes its matmul route
// (_euclidean_dist: [-2x, |x|^2, 1] . [y, 1, |y|^2] in an fp16 GEMM with fp32 accumulation, rounded to fp16, clamped, sqrt),
// then `** 2`, `/ (2 std^2)`, `exp` are fp16 element-wise ops (fp32 math, fp16 result) — every rounding point reproduced:
//   s = fp16(-2 x.y + |x|^2 + |y|^2);  d = fp16(sqrt(max(s,0)));  q = fp16(d*d);  t = fp16(-q / (2 std^2));  term = fp16(exp(t))
// The cancellation in s is rounded at fp16 resolution (~1e-3 .. 4e-3 absolute), which moves a term by up to ~20 %: a property
// of the reference that sample()'s `density < 10` cut sees, so the product path reproduces it instead of "improving" it.
__device__ __forceinline__ float term_half(const float4_t& xi, float xn, const float4_t& r, float rn, float neg_inv_two_var) {
  float s = (-2.f * xi[0]) * r[0];
  s = __builtin_fmaf(-2.f * xi[1], r[1], s);
  s = __builtin_fmaf(-2.f * xi[2], r[2], s);
  s = __builtin_fmaf(-2.f * xi[3], r[3], s);
  s = (s + xn) + rn;
  // hardware sqrt / exp2 (1-2 fp32 ulp) and a multiplication instead of the division: every intermediate is rounded to fp16
  // right after, so the fp16 value differs from the IEEE-exact evaluation only when the fp32 value sits within ~1e-7 of an fp16
  // rounding boundary (about 1 term in 4 000, each then off by one fp16 ulp OF THAT TERM); 3.2 -> 1.6 ms at N = 40 000
  const float d = r16(__builtin_amdgcn_sqrtf(fmaxf(r16(s), 0.f)));
  const float q = r16(d * d);
  const float t = r16(q * neg_inv_two_var);
  return r16(__builtin_amdgcn_exp2f(t * 1.4426950408889634f));
}

__device__ __forceinline__ float norm_half(const float4_t& v) {              // x.pow(2).sum(-1) in fp16
  return r16(((r16(v[0] * v[0]) + r16(v[1] * v[1])) + r16(v[2] * v[2])) + r16(v[3] * v[3]));
}

template <bool HALF>
__global__ __launch_bounds__(256) void kde_kernel(const float4_t* __restrict__ x, float* __restrict__ density, int N, int down,
                                                  float neg_scale_log2, float two_var) {
  __shared__ float4_t tile[TILE];
  __shared__ float tnorm[TILE];
  __shared__ float part[SPLIT][QP];
  const int tid = threadIdx.x, qi = tid & (QP - 1), s = tid >> 6;
  const int i = blockIdx.x * QP + qi;
  const float4_t xi = i < N ? x[i] : float4_t{0, 0, 0, 0};
  const float xn = HALF ? norm_half(xi) : 0.f;
  const int nref = (N + down - 1) / down;
  float sum = 0.f;
  for (int j0 = 0; j0 < nref; j0 += TILE) {
    __syncthreads();
    const int j = j0 + tid;
    if (j < nref) {
      tile[tid] = x[(size_t)j * down];
      if (HALF) tnorm[tid] = norm_half(tile[tid]);
    }
    __syncthreads();
    const int cnt = min(TILE, nref - j0);
    for (int t = s; t < cnt; t += SPLIT) {
      const float4_t r = tile[t];
      if (HALF) {
        sum += term_half(xi, xn, r, tnorm[t], -1.0f / two_var);
        continue;
      }
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

extern "C" int roma_kde_density(const float* x, float* density, int N, int down, float std, int half_mode, void* stream) {
  ROMA_REQUIRE(x && density, ROMA_E_ARG, "roma_kde_density: null pointer");
  ROMA_REQUIRE(N > 0 && down >= 1 && std > 0.f, ROMA_E_SHAPE, "roma_kde_density: bad arguments N=%d down=%d std=%g", N, down, std);
  ROMA_REQUIRE(aligned16(x), ROMA_E_ALIGN, "roma_kde_density: x must be 16-byte aligned");
  const float neg_scale_log2 = -1.4426950408889634f / (2.f * std * std);
  const float two_var = (float)(2.0 * (double)std * (double)std);
  if (half_mode)
    hipLaunchKernelGGL(kde_kernel<true>, dim3((N + QP - 1) / QP), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4_t*>(x), density, N, down, neg_scale_log2, two_var);
  else
    hipLaunchKernelGGL(kde_kernel<false>, dim3((N + QP - 1) / QP), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4_t*>(x), density, N, down, neg_scale_log2, two_var);
  ROMA_CHECK_LAUNCH();
}
