// Diagonal-block step of a blocked Cholesky solve for GP.forward — reference: romatch/models/matcher.py:259-263
// (K_yy_inv = inv(K_yy + sigma I); mu = K_xy @ (K_yy_inv @ f)).  K_yy + sigma I is SPD (1600 x 1600 per direction at
// 560 x 560), so the product with the inverse is computed as a Cholesky solve; the vendor batched potrf takes ~9 ms for
// this size on MI355X (a thousand tiny launches), so the factorisation is blocked by hand: this kernel factors one
// nb x nb diagonal block in LDS (one workgroup per batch item) and also emits the inverse of its triangular factor, which
// turns every panel / substitution step into a plain GEMM (rocBLAS through PyTorch).  fp32 throughout, like the reference.
#include "common.h"

namespace roma {
namespace {

constexpr int NBMAX = 64;

// One 64-lane wavefront per matrix, everything in registers: lane r owns ROW r of the (padded) 64 x 64 block.
//  1. Cholesky, right-looking, column by column: the pivot row's entries are broadcast with v_readlane (static lane
//     index, fully unrolled), so there is no LDS traffic and no barrier at all in the factorisation.
//  2. L is written to LDS once; V = L^-1 by forward substitution with lane j owning COLUMN j of V (the columns are
//     independent): v_i = (delta_ij - sum_{c<i} L[i][c] v_c) / L[i][i], L[i][c] being a wave-uniform broadcast read.
// The previous 256-thread LDS version spent 83 us in ~260 barriers; this one is a straight-line wave program.
__device__ __forceinline__ float lane_bcast(float v, int src_lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}

__global__ __launch_bounds__(64) void chol_diag_kernel(float* __restrict__ A, int lda, long strideA, float* __restrict__ W, int ldw,
                                                       long strideW, int nb, int* __restrict__ info, int info_base) {
  __shared__ float Ls[NBMAX * NBMAX];
  const int r = threadIdx.x, b = blockIdx.x;
  float* Ab = A + (size_t)b * strideA;
  float* Wb = W + (size_t)b * strideW;
  float a[NBMAX];                                   // row r of the block (lower part), identity-padded beyond nb
#pragma unroll
  for (int c = 0; c < NBMAX; ++c) a[c] = (r < nb && c < nb && c <= r) ? Ab[(size_t)r * lda + c] : (r == c ? 1.f : 0.f);
  int bad = 0;
#pragma unroll
  for (int k = 0; k < NBMAX; ++k) {
    const float akk = lane_bcast(a[k], k);
    if (!(akk > 0.f) && bad == 0) bad = k + 1;
    const float inv = 1.f / sqrtf(fmaxf(akk, 1e-30f));
    a[k] = (r >= k) ? a[k] * inv : a[k];            // column k scaled (row k: akk * inv = sqrt(akk))
    const float lrk = a[k];
#pragma unroll
    for (int c = k + 1; c < NBMAX; ++c) {
      const float lck = lane_bcast(a[k], c);        // L[c][k]
      a[c] = (r >= c) ? a[c] - lrk * lck : a[c];
    }
  }
  if (r == 0 && bad != 0 && bad <= nb && info[b] == 0) info[b] = info_base + bad;   // the FIRST failing pivot of the whole solve
#pragma unroll
  for (int c = 0; c < NBMAX; ++c) Ls[r * NBMAX + c] = a[c];
  __syncthreads();
  // V = L^-1: lane j = column j
  float v[NBMAX];
#pragma unroll
  for (int i = 0; i < NBMAX; ++i) {
    float sacc = (i == r) ? 1.f : 0.f;
#pragma unroll
    for (int c = 0; c < i; ++c) sacc = __builtin_fmaf(-Ls[i * NBMAX + c], v[c], sacc);   // v[c] == 0 for c < j
    v[i] = (i >= r) ? sacc / Ls[i * NBMAX + i] : 0.f;
  }
  if (r < nb) {
#pragma unroll
    for (int c = 0; c < NBMAX; ++c)
      if (c < nb && c <= r) Ab[(size_t)r * lda + c] = a[c];
  }
  // lane j holds column j of V: W[i][j] = v[i]
#pragma unroll
  for (int i = 0; i < NBMAX; ++i)
    if (i < nb && r < nb) Wb[(size_t)i * ldw + r] = v[i];
}

}  // namespace
}  // namespace roma

using namespace roma;

extern "C" int roma_chol_diag_block(float* A, int lda, long strideA, float* W, int ldw, long strideW, int nb, int B, int* info,
                                    int info_base, void* stream) {
  ROMA_REQUIRE(A && W && info, ROMA_E_ARG, "roma_chol_diag_block: null pointer");
  ROMA_REQUIRE(nb >= 1 && nb <= NBMAX && B >= 1 && lda >= nb && ldw >= nb, ROMA_E_SHAPE, "roma_chol_diag_block: bad shape nb=%d B=%d", nb, B);
  hipLaunchKernelGGL(chol_diag_kernel, dim3(B), dim3(64), 0, static_cast<hipStream_t>(stream), A, lda, strideA, W, ldw, strideW, nb, info, info_base);
  ROMA_CHECK_LAUNCH();
}
