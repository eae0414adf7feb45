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

// A: lower triangle of the nb x nb block is read, L is written back to the lower triangle (upper untouched).
// W: (nb x nb) row-major, receives L^-1 (lower triangular, zeros above the diagonal).
// 256 threads: thread (r = tid & 63, part = tid >> 6) owns row r and a quarter of the columns / of the dot products.
__global__ __launch_bounds__(256) void chol_diag_kernel(float* __restrict__ A, int lda, long strideA, float* __restrict__ W, int ldw,
                                                        long strideW, int nb, int* __restrict__ info) {
  constexpr int LD = NBMAX + 1;
  __shared__ float L[NBMAX * LD], V[NBMAX * LD], Pp[4][NBMAX];
  const int tid = threadIdx.x, b = blockIdx.x, r = tid & 63, part = tid >> 6;
  float* Ab = A + (size_t)b * strideA;
  float* Wb = W + (size_t)b * strideW;
  for (int c = part; c < NBMAX; c += 4) {
    L[r * LD + c] = (r < nb && c <= r) ? Ab[(size_t)r * lda + c] : (r == c ? 1.f : 0.f);   // identity padding beyond nb
    V[r * LD + c] = 0.f;
  }
  __syncthreads();
  // right-looking Cholesky, one column per step; row r updates its columns c = part, part+4, ...
  for (int k = 0; k < nb; ++k) {
    const float akk = L[k * LD + k];
    if (tid == 0 && !(akk > 0.f)) info[b] = k + 1;
    const float inv = 1.f / sqrtf(fmaxf(akk, 1e-30f));
    __syncthreads();
    if (part == 0 && r >= k) L[r * LD + k] = (r == k) ? akk * inv : L[r * LD + k] * inv;
    __syncthreads();
    if (r > k) {
      const float lrk = L[r * LD + k];
      for (int c = k + 1 + part; c <= r; c += 4) L[r * LD + c] -= lrk * L[c * LD + k];
    }
  }
  __syncthreads();
  // V = L^-1, one row per step: V[i][j] = (delta_ij - sum_{c=j}^{i-1} L[i][c] V[c][j]) / L[i][i], thread (j = r, part)
  for (int i = 0; i < nb; ++i) {
    const int j = r;
    float sp = 0.f;
    if (j < i)
      for (int c = j + part; c < i; c += 4) sp += L[i * LD + c] * V[c * LD + j];
    Pp[part][j] = sp;
    __syncthreads();
    if (part == 0 && j <= i) {
      const float ssum = (Pp[0][j] + Pp[1][j]) + (Pp[2][j] + Pp[3][j]);
      V[i * LD + j] = ((j == i ? 1.f : 0.f) - ssum) / L[i * LD + i];
    }
    __syncthreads();
  }
  if (r < nb)
    for (int c = part; c < nb; c += 4) {
      if (c <= r) Ab[(size_t)r * lda + c] = L[r * LD + c];
      Wb[(size_t)r * ldw + c] = (c <= r) ? V[r * LD + c] : 0.f;
    }
}

}  // namespace
}  // namespace roma

using namespace roma;

extern "C" int roma_chol_diag_block(float* A, int lda, long strideA, float* W, int ldw, long strideW, int nb, int B, int* info,
                                    void* stream) {
  ROMA_REQUIRE(A && W && info, ROMA_E_ARG, "roma_chol_diag_block: null pointer");
  ROMA_REQUIRE(nb >= 1 && nb <= NBMAX && B >= 1 && lda >= nb && ldw >= nb, ROMA_E_SHAPE, "roma_chol_diag_block: bad shape nb=%d B=%d", nb, B);
  hipLaunchKernelGGL(chol_diag_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), A, lda, strideA, W, ldw, strideW, nb, info);
  ROMA_CHECK_LAUNCH();
}
