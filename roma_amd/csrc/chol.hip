// Diagonal-block step of a blocked Cholesky solve for GP.forward — reference: romatch/models/matcher.py:259-263
// (K_yy_inv = inv(K_yy + sigma I); mu = K_xy @ (K_yy_inv @ f)).  K_yy + sigma I is SPD (1600 x 1600 per direction at
// 560 x 560), so the product with the inverse is computed as a Cholesky solve; the vendor batched potrf takes ~9 ms for
// this size on MI355X (a thousand tiny launches), so the factorisation is blocked by hand: this kernel factors one
// nb x nb diagonal block (nb <= 64, one workgroup per batch item) and also emits the inverse of its triangular factor, which
// turns every panel / substitution step into a plain GEMM (rocBLAS through PyTorch).  fp32 throughout, like the reference.
//
// The 25 diagonal blocks of a 1600 x 1600 solve are a DEPENDENT chain on the critical path of the coarse pass, so what counts
// is the latency of one block.  Round 1 factored the block with ONE wavefront, lane = row, every pivot-row entry broadcast
// with v_readlane: 2016 broadcasts, and hipcc turned the triangular predication into 23 000 instructions (4 054 v_readlane,
// 3 261 s_nop hazard pads, 1 796 v_writelane SGPR spills): 65 us per block.  This version keeps the block in LDS and works in
// 16-column panels with the whole 256-thread workgroup:
//   per panel p:  (1) wave 0 factors the 16 x 16 diagonal block in registers (lane = row, 120 broadcasts, no predication: the
//                     strict upper triangle is allowed to hold garbage, nothing reads it);
//                 (2) one thread per row below solves its 16 entries of the panel against that block (L11 is a broadcast read);
//                 (3) all threads apply the rank-16 update to the trailing lower triangle;
//   inverse:      (4) the four 16 x 16 diagonal blocks of L^-1 by forward substitution (one thread per column), then the
//                     off-diagonal blocks as small products, W21 = -W22 (L21 W11), first inside each 32 x 32 half, then the
//                     32 x 32 block below the diagonal.
#include <type_traits>
#include "common.h"

namespace roma {
namespace {

constexpr int NBMAX = 64, PB = 16, LDS_LD = NBMAX + 1;           // row stride 65: a column walk hits 64 different banks

__device__ __forceinline__ float lane_bcast(float v, int src_lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}

__global__ __launch_bounds__(256) void chol_diag_kernel(float* __restrict__ A, int lda, long strideA, float* __restrict__ W, int ldw,
                                                        long strideW, int nb, int* __restrict__ info, int info_base) {
  __shared__ float Ls[NBMAX * LDS_LD];                           // the block, then its factor L (lower; strict upper = garbage)
  __shared__ float Ws[NBMAX * LDS_LD];                           // L^-1
  __shared__ float Ts[32 * 33];                                  // product scratch of the inverse
  __shared__ float s_inv[NBMAX];                                 // 1 / L[i][i]
  __shared__ int s_bad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  float* Ab = A + (size_t)b * strideA;
  float* Wb = W + (size_t)b * strideW;
  if (tid == 0) s_bad = 0;
  // load, identity-padded beyond nb (so partial blocks factor like full ones)
  for (int e = tid; e < NBMAX * NBMAX; e += 256) {
    const int r = e >> 6, c = e & 63;
    Ls[r * LDS_LD + c] = (r < nb && c < nb) ? Ab[(size_t)r * lda + c] : (r == c ? 1.f : 0.f);
    Ws[r * LDS_LD + c] = 0.f;
  }
  __syncthreads();

#pragma unroll
  for (int p = 0; p < NBMAX / PB; ++p) {
    const int o = p * PB;                                        // first row / column of the panel
    // (1) 16 x 16 diagonal block, wave 0, lane = row (lanes >= 16 replicate rows 0..15 and are ignored)
    if (wave == 0) {
      const int i = lane & 15;
      float d[PB];
#pragma unroll
      for (int c = 0; c < PB; ++c) d[c] = Ls[(o + i) * LDS_LD + o + c];
      int bad = 0;
#pragma unroll
      for (int k = 0; k < PB; ++k) {
        const float akk = lane_bcast(d[k], k);
        if (!(akk > 0.f) && bad == 0) bad = o + k + 1;           // non-positive or NaN pivot: clamped, recorded
        const float inv = 1.f / sqrtf(fmaxf(akk, 1e-30f));
        d[k] *= inv;                                             // rows i >= k: L[i][k] (row k: sqrt(akk)); rows i < k: unused garbage
        const float lik = d[k];
#pragma unroll
        for (int c = k + 1; c < PB; ++c) d[c] -= lik * lane_bcast(d[k], c);   // lane c >= k holds the valid L[c][k]
        if (lane == 0) s_inv[o + k] = inv;
      }
      if (lane < PB) {
#pragma unroll
        for (int c = 0; c < PB; ++c) Ls[(o + i) * LDS_LD + o + c] = d[c];
      }
      if (lane == 0 && bad != 0 && s_bad == 0) s_bad = bad;
    }
    __syncthreads();
    // (2) rows below the diagonal block: x L11^T = a, one thread per row
    const int below = NBMAX - o - PB;
    if (tid < below) {
      const int r = o + PB + tid;
      float x[PB];
#pragma unroll
      for (int c = 0; c < PB; ++c) {
        float acc = Ls[r * LDS_LD + o + c];
#pragma unroll
        for (int k = 0; k < c; ++k) acc = __builtin_fmaf(-x[k], Ls[(o + c) * LDS_LD + o + k], acc);
        x[c] = acc * s_inv[o + c];
      }
#pragma unroll
      for (int c = 0; c < PB; ++c) Ls[r * LDS_LD + o + c] = x[c];
    }
    __syncthreads();
    // (3) trailing lower triangle -= panel panel^T
    if (below > 0) {
      for (int e = tid; e < below * below; e += 256) {
        const int ii = e / below, jj = e - ii * below;
        if (jj <= ii) {
          const float* pi = Ls + (o + PB + ii) * LDS_LD + o;
          const float* pj = Ls + (o + PB + jj) * LDS_LD + o;
          float acc = 0.f;
#pragma unroll
          for (int k = 0; k < PB; ++k) acc = __builtin_fmaf(pi[k], pj[k], acc);
          Ls[(o + PB + ii) * LDS_LD + o + PB + jj] -= acc;
        }
      }
      __syncthreads();
    }
  }

  // (4a) diagonal blocks of W = L^-1: thread (q, j) owns column j of block q; w_i = (delta_ij - sum_{c=j}^{i-1} L[i][c] w_c) / L[i][i]
  if (tid < NBMAX) {
    const int q = tid >> 4, j = tid & 15, o = q * PB;
    float w[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      float acc = (i == j) ? 1.f : 0.f;
#pragma unroll
      for (int c = 0; c < i; ++c) acc = __builtin_fmaf(-Ls[(o + i) * LDS_LD + o + c], (c >= j) ? w[c] : 0.f, acc);
      w[i] = (i >= j) ? acc * s_inv[o + i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) Ws[(o + i) * LDS_LD + o + j] = w[i];
  }
  __syncthreads();
  // off-diagonal block (rows r0.., cols c0.., size m) of W = -W22 (L21 W11), W11 / W22 the already finished m x m diagonal
  // blocks at c0 / r0.  Two products through the scratch tile Ts; every thread owns m*m/256 (or fewer) elements.
  // (fixed-length, fully unrolled sums: W11 / W22 are stored with exact zeros above their diagonals, so the triangular bounds are
  // not needed, and without run-time trip counts hipcc pipelines the LDS reads instead of paying one round trip per term)
  auto offdiag = [&](int r0, int c0, auto mconst) {
    constexpr int m = decltype(mconst)::value;
    for (int e = tid; e < m * m; e += 256) {                     // T = L21 W11
      const int i = e / m, j = e - i * m;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < m; ++k) acc = __builtin_fmaf(Ls[(r0 + i) * LDS_LD + c0 + k], Ws[(c0 + k) * LDS_LD + c0 + j], acc);
      Ts[i * 33 + j] = acc;
    }
    __syncthreads();
    for (int e = tid; e < m * m; e += 256) {                     // W21 = -W22 T
      const int i = e / m, j = e - i * m;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < m; ++k) acc = __builtin_fmaf(Ws[(r0 + i) * LDS_LD + r0 + k], Ts[k * 33 + j], acc);
      Ws[(r0 + i) * LDS_LD + c0 + j] = -acc;
    }
    __syncthreads();
  };
  offdiag(PB, 0, std::integral_constant<int, PB>{});            // (4b) inside the upper-left 32 x 32 half
  offdiag(3 * PB, 2 * PB, std::integral_constant<int, PB>{});   //      inside the lower-right 32 x 32 half
  offdiag(2 * PB, 0, std::integral_constant<int, 2 * PB>{});    // (4c) the 32 x 32 block below the diagonal

  // store: L over the lower triangle of the block, W in full (zero above its diagonal)
  for (int e = tid; e < NBMAX * NBMAX; e += 256) {
    const int r = e >> 6, c = e & 63;
    if (r < nb && c < nb) {
      if (c <= r) Ab[(size_t)r * lda + c] = Ls[r * LDS_LD + c];
      Wb[(size_t)r * ldw + c] = Ws[r * LDS_LD + c];
    }
  }
  if (tid == 0 && s_bad != 0 && s_bad <= nb && info[b] == 0) info[b] = info_base + s_bad;   // the FIRST failing pivot of the whole solve
}

}  // namespace
}  // namespace roma

using namespace roma;

extern "C" int roma_chol_diag_block(float* A, int lda, long strideA, float* W, int ldw, long strideW, int nb, int B, int* info,
                                    int info_base, void* stream) {
  ROMA_REQUIRE(A && W && info, ROMA_E_ARG, "roma_chol_diag_block: null pointer");
  ROMA_REQUIRE(nb >= 1 && nb <= NBMAX && B >= 1 && lda >= nb && ldw >= nb, ROMA_E_SHAPE, "roma_chol_diag_block: bad shape nb=%d B=%d", nb, B);
  hipLaunchKernelGGL(chol_diag_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), A, lda, strideA, W, ldw, strideW, nb, info, info_base);
  ROMA_CHECK_LAUNCH();
}
