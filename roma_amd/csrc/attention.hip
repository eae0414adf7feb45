// Multi-head self-attention forward for the two transformers of the path — the 24 blocks of DINOv2 ViT-L/14
// (romatch/models/transformer/layers/attention.py:48-60: softmax(q k^T / sqrt(d)) v, 16 heads x 64) and the 5 blocks of the decoder
// transformer (romatch/models/transformer/__init__.py:30-46 through the same Block) — fp16 / bf16, head dimension 64, no mask other
// than "the first nk tokens are keys" (the sequence is row-padded to the GEMM tile; padded tokens query but are never attended to).
// Round 1/2 ran torch's scaled_dot_product_attention (aotriton flash attention): 62 us per layer at B = 2, 1 664 x 1 601 tokens.
//
// Flash-style: one 256-thread workgroup = 64 queries (one 16-query slab per wavefront) of one (batch, head); K / V tiles of 64 keys
// are double-buffered through LDS; the S x S score matrix never exists.  Everything is arranged so that a query lives on ONE lane
// column of every MFMA result, which makes the online softmax lane-local:
//   S^T = K Q^T     (A = K tile rows from LDS, B = the wavefront's Q fragments from registers): lane (n16, kg) holds the scores of
//                   query n16 against keys 16 j + 4 kg + r;  the row maximum / sum of a query need 2 cross-lane steps (kg);
//   O^T = V^T P^T   (B = P^T: the exponentiated scores ARE the B fragment, straight from the S^T accumulators — no LDS round trip;
//                   A = V^T read from the row-major V tile with gfx950's transposing LDS read, ds_read_b64_tr_b16: 4 keys x 16
//                   columns per 16-lane group, cdna_hip_programming.md T10): lane (n16, kg) holds O of query n16 at d = 16 nb + 4 kg + r,
//                   so the running rescale exp2(m_old - m_new) is a per-lane scalar.
// The k-slot order of the second product is permuted the same way on both operands (slots 0..3 = keys 32 ks + 4 kg + 0..3, slots 4..7
// = keys 32 ks + 16 + 4 kg + 0..3): a contraction does not care, and it is exactly what the S^T accumulators hold.
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace roma {
namespace {

typedef _Float16 ah8 __attribute__((ext_vector_type(8)));
typedef __bf16 ab8 __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4_t mfma_att(const u32x4& a, const u32x4& b, float4_t c, half_t) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ah8, a), __builtin_bit_cast(ah8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float4_t mfma_att(const u32x4& a, const u32x4& b, float4_t c, bf16_t) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ab8, a), __builtin_bit_cast(ab8, b), c, 0, 0, 0);
}
template <typename T> __device__ __forceinline__ uint32_t pack2h(float a, float b) {
  if constexpr (std::is_same<T, half_t>::value) return __builtin_bit_cast(uint32_t, half2_t{(half_t)a, (half_t)b});
  else return __builtin_bit_cast(uint32_t, bf162_t{(bf16_t)a, (bf16_t)b});
}

struct AttnParams {
  const void* q; const void* k; const void* v; void* o;
  long q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn, o_sh;   // element strides: batch, token, head
  int B, H, Nq, Nk;
  float scale_log2e;
};

constexpr int BQ = 64, BK = 64, HD = 64, LDT = 72;              // LDS tile row stride in elements: 144 B, conflict-free b128 rows

// QS = 16-query slabs per wavefront (1: 64 queries per workgroup; 2: 128 — every K / V^T fragment read from LDS then feeds two
// slabs, for launches large enough to fill the chip with half as many workgroups)
template <typename T, int QS>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) T sK[2][BK * LDT];
  __shared__ __attribute__((aligned(16))) T sV[2][BK * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, kg = lane >> 4;
  const int b = blockIdx.y / p.H, h = blockIdx.y - b * p.H;
  const int q0 = blockIdx.x * (BQ * QS) + wave * (16 * QS);
  const T* qp = static_cast<const T*>(p.q) + (size_t)b * p.q_sb + (size_t)h * p.q_sh;
  const T* kp = static_cast<const T*>(p.k) + (size_t)b * p.k_sb + (size_t)h * p.k_sh;
  const T* vp = static_cast<const T*>(p.v) + (size_t)b * p.v_sb + (size_t)h * p.v_sh;
  // Q fragments: lane (query n16, kg) holds Q[q0 + n16][32 ks + 8 kg .. + 8]
  u32x4 qf[QS][2];
#pragma unroll
  for (int z = 0; z < QS; ++z) {
    const T* qr = qp + (size_t)min(q0 + 16 * z + n16, p.Nq - 1) * p.q_sn;
    qf[z][0] = *reinterpret_cast<const u32x4*>(qr + 8 * kg);
    qf[z][1] = *reinterpret_cast<const u32x4*>(qr + 32 + 8 * kg);
  }
  // tile loader: 512 packets of 16 bytes per tensor and tile, two per thread: key = idx / 8, packet = idx % 8
  const int lk0 = tid >> 3, lk1 = (tid + 256) >> 3, lpk = (tid & 7) * 8;
  u32x4 rk[2], rv[2];
  const int ntiles = (p.Nk + BK - 1) / BK;
  // running row pointers of this thread's two packets (advanced by one tile per load; only the last tile can run past nk and is
  // clamped to a valid row — its scores are masked below)
  const T* kq0 = kp + (size_t)lk0 * p.k_sn + lpk;
  const T* kq1 = kp + (size_t)lk1 * p.k_sn + lpk;
  const T* vq0 = vp + (size_t)lk0 * p.v_sn + lpk;
  const T* vq1 = vp + (size_t)lk1 * p.v_sn + lpk;
  const size_t kstep = (size_t)BK * p.k_sn, vstep = (size_t)BK * p.v_sn;
  auto gload = [&](int t) {
    if (t + 1 < ntiles || p.Nk % BK == 0) {
      rk[0] = *reinterpret_cast<const u32x4*>(kq0);
      rk[1] = *reinterpret_cast<const u32x4*>(kq1);
      rv[0] = *reinterpret_cast<const u32x4*>(vq0);
      rv[1] = *reinterpret_cast<const u32x4*>(vq1);
    } else {
      const int k0 = min(t * BK + lk0, p.Nk - 1), k1 = min(t * BK + lk1, p.Nk - 1);
      rk[0] = *reinterpret_cast<const u32x4*>(kp + (size_t)k0 * p.k_sn + lpk);
      rk[1] = *reinterpret_cast<const u32x4*>(kp + (size_t)k1 * p.k_sn + lpk);
      rv[0] = *reinterpret_cast<const u32x4*>(vp + (size_t)k0 * p.v_sn + lpk);
      rv[1] = *reinterpret_cast<const u32x4*>(vp + (size_t)k1 * p.v_sn + lpk);
    }
    kq0 += kstep; kq1 += kstep; vq0 += vstep; vq1 += vstep;
  };
  auto lstore = [&](int buf) {
    *reinterpret_cast<u32x4*>(&sK[buf][lk0 * LDT + lpk]) = rk[0];
    *reinterpret_cast<u32x4*>(&sK[buf][lk1 * LDT + lpk]) = rk[1];
    *reinterpret_cast<u32x4*>(&sV[buf][lk0 * LDT + lpk]) = rv[0];
    *reinterpret_cast<u32x4*>(&sV[buf][lk1 * LDT + lpk]) = rv[1];
  };
  gload(0);
  lstore(0);
  __syncthreads();
  float m_run[QS], l_run[QS];
  float4_t oacc[QS][4];
#pragma unroll
  for (int z = 0; z < QS; ++z) {
    m_run[z] = -INFINITY;
    l_run[z] = 0.f;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) oacc[z][nb] = float4_t{0.f, 0.f, 0.f, 0.f};
  }
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) gload(t + 1);
    // ---- S^T = K Q^T: block j = keys 16 j .. 16 j + 15 ----
    float4_t sacc[QS][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int z = 0; z < QS; ++z) sacc[z][j] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(&sK[buf][(16 * j + n16) * LDT + 32 * ks + 8 * kg]);
#pragma unroll
        for (int z = 0; z < QS; ++z) sacc[z][j] = mfma_att(a, qf[z][ks], sacc[z][j], T{});
      }
    }
    // ---- online softmax of query n16 of each slab (its scores: this lane's 16 values and the same lanes of the other three kg groups).
    //      m_run lives in the scaled domain (score * scale * log2 e); the scale is folded into the exponent's fused multiply-add ----
    u32x4 pf[QS][2];
#pragma unroll
    for (int z = 0; z < QS; ++z) {
      if (t + 1 == ntiles && p.Nk % BK != 0) {                      // only the last tile can hold keys past nk
        const int kbase = t * BK + 4 * kg;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kbase + 16 * j + r >= p.Nk) sacc[z][j][r] = -INFINITY;
      }
      float mt = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) mt = fmaxf(mt, sacc[z][j][r]);
      mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float m_new = fmaxf(m_run[z], mt * p.scale_log2e);     // finite: tile 0 always holds a real key (scale > 0)
      const bool grew = __any(m_new > m_run[z]);                    // wave-uniform: after the first tiles the running maxima rarely move
      const float alpha = grew ? __builtin_amdgcn_exp2f(m_run[z] - m_new) : 1.f;
      m_run[z] = m_new;
      float ls = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[z][j][r], p.scale_log2e, -m_new));
          sacc[z][j][r] = e;
          ls += e;
        }
      l_run[z] = l_run[z] * alpha + ls;                            // per-lane partial sum; the kg groups are added at the end
      if (grew) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) oacc[z][nb][r] *= alpha;
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        pf[z][ks] = u32x4{pack2h<T>(sacc[z][2 * ks][0], sacc[z][2 * ks][1]), pack2h<T>(sacc[z][2 * ks][2], sacc[z][2 * ks][3]),
                          pack2h<T>(sacc[z][2 * ks + 1][0], sacc[z][2 * ks + 1][1]), pack2h<T>(sacc[z][2 * ks + 1][2], sacc[z][2 * ks + 1][3])};
    }
    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        // transposing read: lane i = 4 q + pp of the 16-lane group addresses row q (key), columns 4 pp .. 4 pp + 3 of the 4 x 16 block and
        // receives column i (d = 16 nb + i) of its four rows
        const T* va = &sV[buf][(32 * ks + 4 * kg + (n16 >> 2)) * LDT + 16 * nb + 4 * (n16 & 3)];
        const s4v v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(va));
        const s4v v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(va + 16 * LDT));
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
        const u2 w0 = __builtin_bit_cast(u2, v0), w1 = __builtin_bit_cast(u2, v1);
        const u32x4 af{w0[0], w0[1], w1[0], w1[1]};
#pragma unroll
        for (int z = 0; z < QS; ++z) oacc[z][nb] = mfma_att(af, pf[z][ks], oacc[z][nb], T{});
      }
    }
    if (t + 1 < ntiles) lstore(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int z = 0; z < QS; ++z) {
    float l = l_run[z];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    const int qi = q0 + 16 * z + n16;
    if (qi < p.Nq) {
      T* op = static_cast<T*>(p.o) + (size_t)b * p.o_sb + (size_t)qi * p.o_sn + (size_t)h * p.o_sh;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
        const u2 w{pack2h<T>(oacc[z][nb][0] * inv, oacc[z][nb][1] * inv), pack2h<T>(oacc[z][nb][2] * inv, oacc[z][nb][3] * inv)};
        *reinterpret_cast<u2*>(op + 16 * nb + 4 * kg) = w;
      }
    }
  }
}

}  // namespace
}  // namespace roma

using namespace roma;

extern "C" int roma_attention_fwd(const void* q, const void* k, const void* v, void* o, int B, int H, int Nq, int Nk, int head_dim,
                                  long q_sb, long q_sn, long q_sh, long k_sb, long k_sn, long k_sh, long v_sb, long v_sn, long v_sh,
                                  long o_sb, long o_sn, long o_sh, float scale, int dtype, void* stream) {
  ROMA_REQUIRE(q && k && v && o, ROMA_E_ARG, "roma_attention_fwd: null pointer");
  ROMA_REQUIRE(B > 0 && H > 0 && Nq > 0 && Nk > 0, ROMA_E_SHAPE, "roma_attention_fwd: bad shape");
  ROMA_REQUIRE(head_dim == HD, ROMA_E_UNSUPPORTED, "roma_attention_fwd: head dimension %d (64 only)", head_dim);
  ROMA_REQUIRE(dtype == ROMA_F16 || dtype == ROMA_BF16, ROMA_E_DTYPE, "roma_attention_fwd: fp16 / bf16 only");
  ROMA_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o) && q_sn % 8 == 0 && k_sn % 8 == 0 && v_sn % 8 == 0 && o_sn % 8 == 0 &&
                   q_sh % 8 == 0 && k_sh % 8 == 0 && v_sh % 8 == 0 && o_sh % 8 == 0 && q_sb % 8 == 0 && k_sb % 8 == 0 && v_sb % 8 == 0 && o_sb % 8 == 0,
               ROMA_E_ALIGN, "roma_attention_fwd: bases must be 16-byte aligned and every stride a multiple of 8 elements");
  AttnParams p{q, k, v, o, q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn, o_sh, B, H, Nq, Nk, scale * 1.44269504088896341f};
  hipStream_t s = static_cast<hipStream_t>(stream);
  // 128 queries per workgroup once that still gives every CU several workgroups; 64 otherwise (the B = 2 shape of one image pair)
  static int force = -1;                                            // ROMA_ATT_QS=1|2 (A/B aid)
  if (force < 0) { const char* ev = getenv("ROMA_ATT_QS"); force = ev ? atoi(ev) : 0; }
  const long wg128 = (long)((Nq + 2 * BQ - 1) / (2 * BQ)) * B * H;
  const bool two = force ? force == 2 : wg128 >= num_cus();     // measured: better from 1.6 workgroups per CU up (B = 2, one pair)
  if (two) {
    dim3 grid((Nq + 2 * BQ - 1) / (2 * BQ), B * H);
    if (dtype == ROMA_F16) hipLaunchKernelGGL((attn_fwd_kernel<half_t, 2>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<bf16_t, 2>), grid, dim3(256), 0, s, p);
  } else {
    dim3 grid((Nq + BQ - 1) / BQ, B * H);
    if (dtype == ROMA_F16) hipLaunchKernelGGL((attn_fwd_kernel<half_t, 1>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<bf16_t, 1>), grid, dim3(256), 0, s, p);
  }
  ROMA_CHECK_LAUNCH();
}
