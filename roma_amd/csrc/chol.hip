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
//                     strict upper triangle is allowed to hold garbage, nothing reads it; one v_rcp per pivot on the dependent
//                     chain, the 1 / sqrt of the stored column off it);
//                 (2) one thread per row below solves its 16 entries of the panel against that block (L11 is a broadcast read);
//                 (3) the rank-16 update of the trailing lower triangle, one 4 x 4 register tile per thread;
//   inverse:      (4) the four 16 x 16 diagonal blocks of L^-1 by forward substitution (one thread per column), then the
//                     off-diagonal blocks as small products, W21 = -W22 (L21 W11), first inside each 32 x 32 half, then the
//                     32 x 32 block below the diagonal.
// Round 3 (tools/prof/chol_prof: wall-clock stamps inside the diagonal workgroup of a blocked step): of 44 us per step, 10 us were
// 32 SERIALISED global loads (hipcc had put a vmcnt(0) into every iteration of the tile-load loops), 13 us the four diagonal blocks
// (IEEE sqrt + IEEE division per pivot), 6 us the trailing updates; with the loads batched, rcp / rsq pivots and register-tiled updates a
// step's workgroup takes 31 us, the 25-step chain 0.96 ms instead of 1.25 ms, ops.spd_solve 1.17 ms instead of 1.49 ms.
#include <type_traits>
#include "common.h"

namespace roma {
namespace {

constexpr int NBMAX = 64, PB = 16, LDS_LD = NBMAX + 1;           // row stride 65: a column walk hits 64 different banks

__device__ __forceinline__ float lane_bcast(float v, int src_lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}

// Factor the NBMAX x NBMAX block in Ls (row stride LDS_LD, identity-padded beyond the caller's nb) in place and leave L^-1 in Ws
// (zero-initialised by the caller).  Whole 256-thread workgroup; s_bad receives the first failing pivot + 1 (0: none).
__device__ __forceinline__ void chol_factor_lds(float* __restrict__ Ls, float* __restrict__ Ws, float* __restrict__ Ts,
                                                float* __restrict__ s_inv, int* __restrict__ s_bad_p, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  int& s_bad = *s_bad_p;
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
        float akk = lane_bcast(d[k], k);
        if (!(akk > 0.f) && bad == 0) bad = o + k + 1;           // non-positive or NaN pivot: clamped, recorded
        akk = fmaxf(akk, 1e-30f);
        // The 64 pivots of a block are one dependent chain, and what the next pivot waits for is the update of ITS column only.  On that
        // chain: one hardware reciprocal and the unnormalised column, A[i][c] -= a_i a_c / akk (1 ulp, like any other rounding of the
        // update).  Off it: 1 / sqrt(akk) (v_rsq + one Newton step) for the column that is stored.  `1.f / sqrtf()` was ~25 dependent
        // instructions per pivot (IEEE sqrt, then IEEE division): 3.2 us per 16 x 16 block, 13 of the 44 us of a blocked step.
        const float t = d[k] * __builtin_amdgcn_rcpf(akk);
#pragma unroll
        for (int c = k + 1; c < PB; ++c) d[c] = __builtin_fmaf(-t, lane_bcast(d[k], c), d[c]);   // lane c >= k holds the valid a_c
        float inv = __builtin_amdgcn_rsqf(akk);
        inv = inv * __builtin_fmaf(-0.5f * akk * inv, inv, 1.5f);
        d[k] *= inv;                                             // rows i >= k: L[i][k] (row k: sqrt(akk)); rows i < k: unused garbage
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
    // (one 4 x 4 tile of the lower triangle per thread — 78 / 36 / 10 tiles: 128 LDS reads for 256 multiply-adds; the element-per-
    // thread form read 32 values per multiply-add chain and walked the whole square, 9 rounds at 48 rows)
    if (below > 0) {
      constexpr int nt = (NBMAX - PB) / 4;                       // tiles per side at the first panel (upper bound of the search below)
      const int ntp = below >> 2;
      int t = tid, ti = 0;
#pragma unroll
      for (int q = 0; q < nt; ++q)
        if (t >= ti + 1) { t -= ti + 1; ++ti; }
      if (ti < ntp) {                                            // tile (ti, t), t <= ti
        const float* pi = Ls + (o + PB + 4 * ti) * LDS_LD + o;
        const float* pj = Ls + (o + PB + 4 * t) * LDS_LD + o;
        float acc[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;
#pragma unroll
        for (int k = 0; k < PB; ++k) {
          float a[4], b[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) { a[r] = pi[r * LDS_LD + k]; b[r] = pj[r * LDS_LD + k]; }
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = __builtin_fmaf(a[r], b[c], acc[r][c]);
        }
        float* cc = Ls + (o + PB + 4 * ti) * LDS_LD + o + PB + 4 * t;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) cc[r * LDS_LD + c] -= acc[r][c];   // diagonal tiles also touch the strict upper triangle: unused
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

}

__global__ __launch_bounds__(256) void chol_diag_kernel(float* __restrict__ A, int lda, long strideA, float* __restrict__ W, int ldw,
                                                        long strideW, int nb, int* __restrict__ info, int info_base) {
  __shared__ float Ls[NBMAX * LDS_LD];                           // the block, then its factor L (lower; strict upper = garbage)
  __shared__ float Ws[NBMAX * LDS_LD];                           // L^-1
  __shared__ float Ts[32 * 33];                                  // product scratch of the inverse
  __shared__ float s_inv[NBMAX];                                 // 1 / L[i][i]
  __shared__ int s_bad;
  const int tid = threadIdx.x, b = blockIdx.x;
  float* Ab = A + (size_t)b * strideA;
  float* Wb = W + (size_t)b * strideW;
  if (tid == 0) s_bad = 0;
  // load, identity-padded beyond nb (so partial blocks factor like full ones)
  {
    const int r0 = tid >> 6, c = tid & 63;
    float av[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) av[u] = Ab[(size_t)min(r0 + 4 * u, nb - 1) * lda + min(c, nb - 1)];   // all in flight at once
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int r = r0 + 4 * u;
      Ls[r * LDS_LD + c] = (r < nb && c < nb) ? av[u] : (r == c ? 1.f : 0.f);
      Ws[r * LDS_LD + c] = 0.f;
    }
  }
  __syncthreads();

  chol_factor_lds(Ls, Ws, Ts, s_inv, &s_bad, tid);

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


typedef float f2_t __attribute__((ext_vector_type(2)));

// acc (4 x 4 outputs as 4 x 2 packed pairs) += sum_k a[k][ra..ra+3] (x) b[k][cb..cb+3] over the 64 rows of two LDS tiles
// (row stride TLD): v_pk_fma_f32, two outputs per instruction.
__device__ __forceinline__ void tile_product(const float* __restrict__ sa, const float* __restrict__ sb, int ra, int cb, f2_t (&acc)[4][2]) {
#pragma unroll 8
  for (int k = 0; k < 64; ++k) {
    const float4_t a4 = *reinterpret_cast<const float4_t*>(sa + k * 68 + ra);
    const float4_t b4 = *reinterpret_cast<const float4_t*>(sb + k * 68 + cb);
    const f2_t b01 = __builtin_shufflevector(b4, b4, 0, 1), b23 = __builtin_shufflevector(b4, b4, 2, 3);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f2_t ar{a4[r], a4[r]};
      acc[r][0] = __builtin_elementwise_fma(ar, b01, acc[r][0]);
      acc[r][1] = __builtin_elementwise_fma(ar, b23, acc[r][1]);
    }
  }
}

// ---- one fused forward step of the blocked solve (round 2) ----------------------------------------------------------------------
// ops.spd_solve ran, per 64-row block s of the augmented matrix A = [K | F]: this file's diagonal-block kernel, a GEMM
// r = W_s A[j:e, e:] and a GEMM A[e:, e:] -= r[:, :n-e]^T r — 75 dependent launches of 5-25 us kernels on the critical path of the
// coarse pass.  This kernel is one launch per block: every 64 x 64 tile (I, J >= I) of the trailing matrix recomputes the two row
// panels it needs, r_I = W_s A[j:e, I] and r_J (64^3 products out of LDS; redundant across tiles, but no tile waits for another),
// applies A[I, J] -= r_I^T r_J, and the workgroup that owns the NEXT diagonal tile factors it in place (chol_factor_lds) and emits
// W_{s+1}.  Only tiles J >= I are kept up to date: the next step reads the row block A[e:e+64, e+64:] and nothing below the diagonal.
// "R tiles" (one per column tile) write the panel r itself, which the backward substitution needs.
struct StepParams {
  float* A; int lda; long strideA; int n, ncols, j, nb;
  const float* W; int ldw; long strideW;
  float* R; int ldr; long strideR;
  float* Wn; int ldwn; long strideWn;
  int* info; int info_base;
  int ntr, ntc;                                                   // row / column tiles of the trailing region
};

constexpr int TS = 64, TLD = 68;                                  // tile size; LDS row stride (16-byte aligned rows, 4-bank skew)

__global__ __launch_bounds__(256) void chol_step_kernel(StepParams p) {
  __shared__ __attribute__((aligned(16))) float smem[3 * TS * TLD + NBMAX + 4];
  float* sWT = smem;                                              // W^T
  float* sI = smem + TS * TLD;                                    // A[j:e, I-range], then r_I
  float* sJ = smem + 2 * TS * TLD;                                // A[j:e, J-range], then r_J
  const int tid = threadIdx.x, b = blockIdx.y;
  const int e = p.j + p.nb;
  // tile id -> kind: the first ntc ids are R tiles (J = id); then the update tiles row by row, J = I .. ntc-1
  int id = blockIdx.x, I = -1, J = id;
  if (id >= p.ntc) {
    id -= p.ntc;
    I = 0;
    while (id >= p.ntc - I) { id -= p.ntc - I; ++I; }
    J = I + id;
  }
  float* Ab = p.A + (size_t)b * p.strideA;
  const float* Wb = p.W + (size_t)b * p.strideW;
  // W_s and the row block's columns of this tile: element u of a thread is (row (tid >> 6) + 4 u, column tid & 63).  ALL loads of the
  // three tiles are issued (clamped addresses, no branches) before the first one is consumed: as `for (idx ...) lds[..] = cond ?
  // global[..] : 0` hipcc emitted one load + s_waitcnt vmcnt(0) per iteration — 32 dependent round trips, 10 of the 44 us of a step.
  const int J0 = e + J * TS, I0 = e + (I < 0 ? 0 : I) * TS;
  const bool two = I >= 0 && I != J;
  const int t4r = (tid >> 4) * 4, t4c = (tid & 15) * 4;           // 4 x 4 outputs per thread
  float aold[4][4];                                                // this tile of A: requested with the rest, needed after the three products
  {
    const int r0 = tid >> 6, c = tid & 63;
    float wv[16], jv[16], iv[16];
    const float* wp = Wb + min(c, p.nb - 1);
    const float* jp = Ab + (size_t)p.j * p.lda + min(J0 + c, p.ncols - 1);
    const float* ip = Ab + (size_t)p.j * p.lda + min(I0 + c, p.n - 1);
#pragma unroll
    for (int u = 0; u < 16; ++u) wv[u] = wp[(size_t)min(r0 + 4 * u, p.nb - 1) * p.ldw];
#pragma unroll
    for (int u = 0; u < 16; ++u) jv[u] = jp[(size_t)min(r0 + 4 * u, p.nb - 1) * p.lda];
    if (two) {
#pragma unroll
      for (int u = 0; u < 16; ++u) iv[u] = ip[(size_t)min(r0 + 4 * u, p.nb - 1) * p.lda];
    }
    if (I >= 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)                               // clamped: out-of-range entries are never stored
          aold[r][cc] = Ab[(size_t)min(I0 + t4r + r, p.n - 1) * p.lda + min(J0 + t4c + cc, p.ncols - 1)];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int k = r0 + 4 * u;
      sWT[c * TLD + k] = (k < p.nb && c < p.nb) ? wv[u] : 0.f;       // W^T, zero-padded beyond nb
      sJ[k * TLD + c] = (k < p.nb && J0 + c < p.ncols) ? jv[u] : 0.f;
    }
    if (two) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int k = r0 + 4 * u;
        sI[k * TLD + c] = (k < p.nb && I0 + c < p.n) ? iv[u] : 0.f;
      }
    }
  }
  __syncthreads();
  auto panel = [&](const float* sX, float (&out)[4][4]) {          // out = (W A_X)[t4r.., t4c..]
    f2_t acc[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r][0] = acc[r][1] = f2_t{0.f, 0.f};
    tile_product(sWT, sX, t4r, t4c, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) { out[r][0] = acc[r][0][0]; out[r][1] = acc[r][0][1]; out[r][2] = acc[r][1][0]; out[r][3] = acc[r][1][1]; }
  };
  float rj[4][4], ri[4][4];
  panel(sJ, rj);
  if (I >= 0 && I != J) panel(sI, ri);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    *reinterpret_cast<float4_t*>(sJ + (t4r + r) * TLD + t4c) = float4_t{rj[r][0], rj[r][1], rj[r][2], rj[r][3]};
    if (I >= 0 && I != J) *reinterpret_cast<float4_t*>(sI + (t4r + r) * TLD + t4c) = float4_t{ri[r][0], ri[r][1], ri[r][2], ri[r][3]};
  }
  __syncthreads();
  if (I < 0) {                                                     // R tile: store the panel
    float* Rb = p.R + (size_t)b * p.strideR;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = t4r + r;
      if (k >= p.nb) continue;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (J0 + t4c + c < p.ncols) Rb[(size_t)k * p.ldr + (J0 - e) + t4c + c] = rj[r][c];
    }
    return;
  }
  const float* sL = (I == J) ? sJ : sI;                            // r_I
  f2_t acc2[4][2];                                                 // (r_I^T r_J)[t4r.., t4c..]; rows k >= nb of both panels are zero
#pragma unroll
  for (int r = 0; r < 4; ++r) acc2[r][0] = acc2[r][1] = f2_t{0.f, 0.f};
  tile_product(sL, sJ, t4r, t4c, acc2);
  float acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { acc[r][0] = acc2[r][0][0]; acc[r][1] = acc2[r][0][1]; acc[r][2] = acc2[r][1][0]; acc[r][3] = acc2[r][1][1]; }
  const bool next_diag = (I == 0 && J == 0 && p.Wn != nullptr);
  float outv[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = I0 + t4r + r;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int col = J0 + t4c + c;
      const bool in = row < p.n && col < p.ncols;
      float v = in ? aold[r][c] - acc[r][c] : 0.f;
      outv[r][c] = v;
      if (in) Ab[(size_t)row * p.lda + col] = v;
    }
  }
  if (!next_diag) return;
  // ---- the next diagonal block: factor it here, in the workgroup that just produced it ----
  __syncthreads();                                                 // the panels in LDS are dead from here
  float* Ls = smem;                                                // NBMAX x LDS_LD
  float* Ws = smem + NBMAX * LDS_LD;
  float* Ts = smem + 2 * NBMAX * LDS_LD;                           // 32 x 33
  float* s_inv = smem + 3 * TS * TLD;
  int* s_bad = reinterpret_cast<int*>(smem + 3 * TS * TLD + NBMAX);
  static_assert(2 * NBMAX * LDS_LD + 32 * 33 <= 3 * TS * TLD, "factor scratch must fit the panel buffers");
  const int nbn = min(NBMAX, p.n - e);
  if (tid == 0) *s_bad = 0;
  for (int idx = tid; idx < NBMAX * NBMAX; idx += 256) Ws[(idx >> 6) * LDS_LD + (idx & 63)] = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int rr = t4r + r, cc = t4c + c;
      Ls[rr * LDS_LD + cc] = (rr < nbn && cc < nbn) ? outv[r][c] : (rr == cc ? 1.f : 0.f);
    }
  __syncthreads();
  chol_factor_lds(Ls, Ws, Ts, s_inv, s_bad, tid);
  float* Wnb = p.Wn + (size_t)b * p.strideWn;
  for (int idx = tid; idx < NBMAX * NBMAX; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    if (r < nbn && c < nbn) {
      if (c <= r) Ab[(size_t)(e + r) * p.lda + e + c] = Ls[r * LDS_LD + c];
      Wnb[(size_t)r * p.ldwn + c] = Ws[r * LDS_LD + c];
    }
  }
  if (tid == 0 && *s_bad != 0 && *s_bad <= nbn && p.info[b] == 0) p.info[b] = p.info_base + *s_bad;
}


// ---- one fused step of a triangular substitution with the finished factor (round 2) -----------------------------------------------
// dir < 0, back substitution L^T X = T: block row s (from the last to the first): X_s = W_s^T T_s, then every block row i < s takes
//   T_i -= L[s, i]^T X_s.
// dir > 0, forward substitution L Y = T (a NEW right-hand side against the finished factor: iterative refinement): block row s
//   (from the first to the last): Y_s = W_s T_s, then every block row i > s takes T_i -= L[i, s] Y_s.
// L comes from the panels roma_chol_step wrote: panel k is L[e_k:, block k]^T (64 x (n - e_k)).  Right-looking, so one launch per
// block row: every (i, 64-column tile c) workgroup recomputes the X_s / Y_s tile it needs (one 64^3 product; no workgroup waits for
// another) and applies its update; "X tiles" (one per c) store X_s / Y_s.
struct SubstParams {
  const float* W; int ldw; long strideW;                          // W_s = inverse factor of diagonal block s
  const float* R; long strideRb, strideRs; int ldr;               // panels: panel k of matrix b at R + b*strideRb + k*strideRs
  float* T; long strideTb, strideTs; int ldt, t_in_panel;         // T_i at T + b*strideTb + i*strideTs (+ n - e_i columns if t_in_panel)
  float* X; int ldx; long strideX;
  int n, m, nb, s, dir, nblk, ctiles;
};

__global__ __launch_bounds__(256) void chol_subst_kernel(SubstParams p) {
  __shared__ __attribute__((aligned(16))) float smem[3 * TS * TLD];
  float* sW = smem;                                               // W_s (back) / W_s^T (forward), then the L piece
  float* sT = smem + TS * TLD;                                    // T_s tile
  float* sX = smem + 2 * TS * TLD;                                // X_s / Y_s tile
  const int tid = threadIdx.x, b = blockIdx.y;
  int id = blockIdx.x, i = -1, c;
  if (id < p.ctiles) c = id;
  else {
    id -= p.ctiles;
    i = id / p.ctiles;
    c = id - i * p.ctiles;
    if (p.dir > 0) i += p.s + 1;                                   // forward: the block rows below s
  }
  const int js = p.s * p.nb, es = min(js + p.nb, p.n), ws = es - js;
  const float* Wb = p.W + (size_t)b * p.strideW;
  const float* Rb = p.R + (size_t)b * p.strideRb;
  float* Tb = p.T + (size_t)b * p.strideTb;
  const float* Ts = Tb + (size_t)p.s * p.strideTs + (p.t_in_panel ? p.n - es : 0);
  const int c0 = c * TS;
  const int t4r = (tid >> 4) * 4, t4c = (tid & 15) * 4;
  // every global load of the workgroup is issued (clamped addresses, no branches) before the first one is consumed — see chol_step_kernel
  const int ji = (i < 0 ? 0 : i) * p.nb, ei = min(ji + p.nb, p.n), wi = ei - ji;
  float* Ti = Tb + (size_t)(i < 0 ? 0 : i) * p.strideTs + (p.t_in_panel ? p.n - ei : 0);
  float told[4][4], lpiece[16];                                    // this workgroup's T_i tile and its piece of L: consumed after the first product
  {
    const int r0 = tid >> 6, q = tid & 63;
    float wv[16], tv[16];
    const float* wp = Wb + min(q, ws - 1);
    const float* tp = Ts + min(c0 + q, p.m - 1);
#pragma unroll
    for (int u = 0; u < 16; ++u) wv[u] = wp[(size_t)min(r0 + 4 * u, ws - 1) * p.ldw];
#pragma unroll
    for (int u = 0; u < 16; ++u) tv[u] = tp[(size_t)min(r0 + 4 * u, ws - 1) * p.ldt];
    if (i >= 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) told[r][cc] = Ti[(size_t)min(t4r + r, wi - 1) * p.ldt + min(c0 + t4c + cc, p.m - 1)];
      if (p.dir < 0) {                                             // L[s, i][q][r] = panel_i[r][(js - e_i) + q]
        const float* Pi = Rb + (size_t)i * p.strideRs + (js - ei) + min(q, ws - 1);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const float v = Pi[(size_t)min(r0 + 4 * u, wi - 1) * p.ldr];
          lpiece[u] = (r0 + 4 * u < wi && q < ws) ? v : 0.f;
        }
      } else {                                                     // L[i, s][r][q] = panel_s[q][(ji - e_s) + r]
        const float* Ps = Rb + (size_t)p.s * p.strideRs + (ji - es) + min(q, wi - 1);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const float v = Ps[(size_t)min(r0 + 4 * u, ws - 1) * p.ldr];
          lpiece[u] = (q < wi && r0 + 4 * u < ws) ? v : 0.f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int k = r0 + 4 * u;
      const float w = (k < ws && q < ws) ? wv[u] : 0.f;
      if (p.dir < 0) sW[k * TLD + q] = w; else sW[q * TLD + k] = w;   // product below sums over the FIRST index
      sT[k * TLD + q] = (k < ws && c0 + q < p.m) ? tv[u] : 0.f;
    }
  }
  __syncthreads();
  f2_t x2[4][2];                                                   // X_s[q][col] = sum_k sW[k][q] T_s[k][col]
#pragma unroll
  for (int r = 0; r < 4; ++r) x2[r][0] = x2[r][1] = f2_t{0.f, 0.f};
  tile_product(sW, sT, t4r, t4c, x2);
  float xs[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { xs[r][0] = x2[r][0][0]; xs[r][1] = x2[r][0][1]; xs[r][2] = x2[r][1][0]; xs[r][3] = x2[r][1][1]; }
  if (i < 0) {                                                     // X tile: store
    float* Xb = p.X + (size_t)b * p.strideX;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc)
        if (t4r + r < ws && c0 + t4c + cc < p.m) Xb[(size_t)(js + t4r + r) * p.ldx + c0 + t4c + cc] = xs[r][cc];
    return;
  }
  __syncthreads();                                                 // every thread is done with sW
#pragma unroll
  for (int r = 0; r < 4; ++r) *reinterpret_cast<float4_t*>(sX + (t4r + r) * TLD + t4c) = float4_t{xs[r][0], xs[r][1], xs[r][2], xs[r][3]};
#pragma unroll
  for (int u = 0; u < 16; ++u) {                                   // sW[q][r] = (the L piece)[q][r]: the product sums over q
    const int idx = tid + 256 * u;
    if (p.dir < 0) sW[(idx & 63) * TLD + (idx >> 6)] = lpiece[u];  // loaded r-major
    else sW[(idx >> 6) * TLD + (idx & 63)] = lpiece[u];            // loaded q-major
  }
  __syncthreads();
  f2_t acc2[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r) acc2[r][0] = acc2[r][1] = f2_t{0.f, 0.f};
  tile_product(sW, sX, t4r, t4c, acc2);
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
      if (t4r + r < wi && c0 + t4c + cc < p.m) Ti[(size_t)(t4r + r) * p.ldt + c0 + t4c + cc] = told[r][cc] - acc2[r][cc >> 1][cc & 1];
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

extern "C" int roma_chol_step(float* A, int lda, long strideA, int n, int ncols, int j, int nb, const float* W, int ldw, long strideW,
                              float* R, int ldr, long strideR, float* Wn, int ldwn, long strideWn, int* info, int info_base, int B,
                              void* stream) {
  ROMA_REQUIRE(A && W && R && info, ROMA_E_ARG, "roma_chol_step: null pointer");
  ROMA_REQUIRE(nb == NBMAX || j + nb >= n, ROMA_E_UNSUPPORTED, "roma_chol_step: block size %d with blocks to follow (the next diagonal block is "
               "factored at %d rows: only the last block may be narrower)", nb, NBMAX);
  ROMA_REQUIRE(nb >= 1 && nb <= NBMAX && B >= 1 && n >= 1 && ncols >= n && j >= 0 && j + nb <= n && lda >= ncols && ldw >= nb &&
                   ldr >= ncols - (j + nb) && (Wn == nullptr || ldwn >= 1),
               ROMA_E_SHAPE, "roma_chol_step: bad shape n=%d ncols=%d j=%d nb=%d B=%d", n, ncols, j, nb, B);
  const int e = j + nb;
  StepParams p{A, lda, strideA, n, ncols, j, nb, W, ldw, strideW, R, ldr, strideR, e < n ? Wn : nullptr, ldwn, strideWn, info, info_base, 0, 0};
  p.ntr = (n - e + TS - 1) / TS;
  p.ntc = (ncols - e + TS - 1) / TS;
  if (p.ntc == 0) return 0;                                        // nothing to the right of the block
  int tiles = p.ntc;                                               // R tiles
  for (int I = 0; I < p.ntr; ++I) tiles += p.ntc - I;
  hipLaunchKernelGGL(chol_step_kernel, dim3(tiles, B), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  ROMA_CHECK_LAUNCH();
}

extern "C" int roma_chol_subst_step(int dir, const float* W, int ldw, long strideW, const float* R, long strideRb, long strideRs, int ldr,
                                    float* T, long strideTb, long strideTs, int ldt, int t_in_panel, float* X, int ldx, long strideX, int n,
                                    int m, int nb, int s, int B, void* stream) {
  ROMA_REQUIRE(W && R && T && X, ROMA_E_ARG, "roma_chol_subst_step: null pointer");
  ROMA_REQUIRE(nb >= 1 && nb <= NBMAX && B >= 1 && n >= 1 && m >= 1 && s >= 0 && s * nb < n && ldw >= 1 && ldr >= 1 && ldt >= 1 &&
                   ldx >= m && dir != 0,
               ROMA_E_SHAPE, "roma_chol_subst_step: bad shape n=%d m=%d nb=%d s=%d B=%d dir=%d", n, m, nb, s, B, dir);
  const int nblk = (n + nb - 1) / nb;
  ROMA_REQUIRE(nb == NBMAX || nblk == 1, ROMA_E_UNSUPPORTED, "roma_chol_subst_step: block size %d (the update tiles assume 64-row blocks)", nb);
  SubstParams p{W, ldw, strideW, R, strideRb, strideRs, ldr, T, strideTb, strideTs, ldt, t_in_panel, X, ldx, strideX, n, m, nb, s, dir, nblk,
                (m + TS - 1) / TS};
  const int others = dir < 0 ? s : nblk - 1 - s;
  hipLaunchKernelGGL(chol_subst_kernel, dim3(p.ctiles * (1 + others), B), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  ROMA_CHECK_LAUNCH();
}
