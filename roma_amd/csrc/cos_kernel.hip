// CosKernel on the CDNA4 matrix cores — reference: romatch/models/matcher.py:154-163 (used by GP.forward, :255-257).
//   K[b,n,m] = exp((<x_n, y_m> / (|x_n| |y_m| + eps) - 1) / T)          x: (B,N,D), y: (B,M,D) fp32 row-major
// This is the one dense C x C feature contraction of the path ("global correlation"), so it runs on MFMA.  The GP
// is fp32 in the reference on every device (matcher.py:254), so the exact-fp32 form v_mfma_f32_32x32x2_f32 is used:
// bit-for-bit an fmaf chain, 1/16 of the bf16 rate but still the full fp32 vector rate with the VALU left free.
// 64x64 output tile per 256-thread workgroup (2x2 wavefronts of 32x32), K staged 16 deep through padded LDS tiles,
// row/column norms accumulated from the same staged tiles, normalise + exp fused into the accumulator epilogue.
//
// Accuracy (round 2).  cond(K_yy + sigma I) is ~1e4 on real features and |Z| = |K_yy^-1 f| reaches 16, so the posterior
// mu = K_xy Z sees every relative error of a K entry amplified: with all 512 products of a dot product summed in one fp32
// accumulator (error ~1e-6 of |x||y|, x 1/T = 5 through the exponential: entries good to 1e-5) mu is 1.3e-3 off the fp64 answer —
// as is the reference's own fp32 einsum, each with its own rounding (tools/gp_error_budget.py).  With entries good to 1e-7 it is
// 2e-4 (3e-5 after spd_solve's refinement step).  So the MFMA accumulator is flushed into an fp64 sum every 64 channels, the norms
// are summed in fp64, and the epilogue (normalise, subtract 1, exp) runs in fp64; the MFMA work is unchanged.
#include "common.h"

namespace roma {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BT = 64, KT = 16, LD = KT + 1;

// 4 consecutive channels of a row as fp32 (16-bit storage converts exactly, so the result equals the fp32 kernel on x.float())
template <typename T> __device__ __forceinline__ float4_t ld4(const T* p) {
  if constexpr (sizeof(T) == 4) {
    return *reinterpret_cast<const float4_t*>(p);
  } else {
    typedef T v4 __attribute__((ext_vector_type(4)));
    const v4 h = *reinterpret_cast<const v4*>(p);
    return float4_t{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cos_kernel_mfma(const T* __restrict__ x, const T* __restrict__ y, float* __restrict__ K,
                                                       int N, int M, int D, int x_pitch, int y_pitch, int y_shift, float invT,
                                                       float eps, float diag_add) {
  __shared__ float sA[BT * LD], sB[BT * LD];
  __shared__ double sN[2 * BT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int b = blockIdx.z, n0 = blockIdx.y * BT, m0 = blockIdx.x * BT;
  const T* xb = x + (size_t)b * N * x_pitch;
  const T* yb = y + (size_t)((b + y_shift) % (int)gridDim.z) * M * y_pitch;
  f32x16 acc;
  double accd[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc[i] = 0.f; accd[i] = 0.0; }
  double nrm = 0.0;                                  // threads 0..63: |x_row|^2, 64..127: |y_row|^2
  const int lr = tid >> 2, lc = (tid & 3) * 4;       // loader: row 0..63, 4 consecutive k
  for (int k0 = 0; k0 < D; k0 += KT) {
    float4_t va{0, 0, 0, 0}, vb{0, 0, 0, 0};
    if (n0 + lr < N) va = ld4<T>(xb + (size_t)(n0 + lr) * x_pitch + k0 + lc);
    if (m0 + lr < M) vb = ld4<T>(yb + (size_t)(m0 + lr) * y_pitch + k0 + lc);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sA[lr * LD + lc + i] = va[i];
      sB[lr * LD + lc + i] = vb[i];
    }
    __syncthreads();
    if (tid < 2 * BT) {
      const float* r = (tid < BT ? sA : sB) + (tid & (BT - 1)) * LD;
#pragma unroll
      for (int i = 0; i < KT; ++i) nrm = __builtin_fma((double)r[i], (double)r[i], nrm);
    }
    // lane l feeds A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31] of each 32x32x2 step
    const float* pa = sA + (wr * 32 + (lane & 31)) * LD + (lane >> 5);
    const float* pb = sB + (wc * 32 + (lane & 31)) * LD + (lane >> 5);
#pragma unroll
    for (int kk = 0; kk < KT; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[kk], pb[kk], acc, 0, 0, 0);
    if (((k0 / KT) & 3) == 3 || k0 + KT >= D) {      // every 64 channels: fp32 partial sums -> fp64 totals
#pragma unroll
      for (int i = 0; i < 16; ++i) { accd[i] += (double)acc[i]; acc[i] = 0.f; }
    }
  }
  if (tid < 2 * BT) sN[tid] = sqrt(nrm);
  __syncthreads();
  const int col = lane & 31;
  const int m = m0 + wc * 32 + col;
  const double ny = sN[BT + wc * 32 + col];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int n = n0 + wr * 32 + row;
    if (n < N && m < M) {
      const double c = accd[r] / (sN[wr * 32 + row] * ny + (double)eps);
      float v = (float)exp((c - 1.0) * (double)invT);
      if (n == m) v += diag_add;
      K[((size_t)b * N + n) * M + m] = v;
    }
  }
}

}  // namespace
}  // namespace roma

using namespace roma;

extern "C" int roma_cos_kernel(const void* x, const void* y, float* K, int B, int N, int M, int D, int dtype, int x_pitch,
                               int y_pitch, int y_batch_shift, float T, float eps, float diag_add, void* stream) {
  ROMA_REQUIRE(x && y && K, ROMA_E_ARG, "roma_cos_kernel: null pointer");
  ROMA_REQUIRE(B > 0 && N > 0 && M > 0 && D > 0, ROMA_E_SHAPE, "roma_cos_kernel: bad shape");
  ROMA_REQUIRE(D % KT == 0, ROMA_E_SHAPE, "roma_cos_kernel: D=%d must be a multiple of %d", D, KT);
  ROMA_REQUIRE(x_pitch >= D && y_pitch >= D, ROMA_E_SHAPE, "roma_cos_kernel: row pitch smaller than D");
  ROMA_REQUIRE(dtype >= ROMA_F32 && dtype <= ROMA_BF16, ROMA_E_DTYPE, "roma_cos_kernel: unknown dtype %d", dtype);
  const int e16 = dtype == ROMA_F32 ? 4 : 8;
  ROMA_REQUIRE(aligned16(x) && aligned16(y) && x_pitch % e16 == 0 && y_pitch % e16 == 0, ROMA_E_ALIGN,
               "roma_cos_kernel: x and y must be 16-byte aligned with row pitches that keep every row 16-byte aligned");
  ROMA_REQUIRE(y_batch_shift >= 0 && y_batch_shift < B, ROMA_E_ARG, "roma_cos_kernel: y_batch_shift %d outside [0, B)", y_batch_shift);
  ROMA_REQUIRE(T > 0.f, ROMA_E_ARG, "roma_cos_kernel: temperature must be positive");
  dim3 grid((M + BT - 1) / BT, (N + BT - 1) / BT, B);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define ROMA_COS(TT)                                                                                                         \
  hipLaunchKernelGGL((cos_kernel_mfma<TT>), grid, dim3(256), 0, s, (const TT*)x, (const TT*)y, K, N, M, D, x_pitch, y_pitch, \
                     y_batch_shift, 1.f / T, eps, diag_add)
  if (dtype == ROMA_F32) ROMA_COS(float); else if (dtype == ROMA_F16) ROMA_COS(half_t); else ROMA_COS(bf16_t);
#undef ROMA_COS
  ROMA_CHECK_LAUNCH();
}
