// Residual add (+ LayerScale) fused with the NEXT LayerNorm, for gfx950 — the seam between two transformer half-blocks:
//   reference  romatch/models/transformer/layers/block.py:87-107  (x = x + ls1(attn(norm1(x))); x = x + ls2(mlp(norm2(x))))
//   and        romatch/models/transformer/__init__.py:30-46       (decoder transformer under autocast: the token / residual
//              stream is fp32 because cat(gp_posterior[fp32], features[fp16]) promotes, LayerNorm runs in fp32, the
//              GEMMs / attention consume 16-bit casts).
//   x  <- x + ls * y          in x's own dtype (fp32 decoder stream, or the 16-bit DINOv2 stream), written back in place
//   out = LN(x) * gamma + beta  cast to out's dtype           (gamma == NULL: out = cast(x), the cast in front of `to_out`)
// One wavefront per row, the row kept in registers between the two passes (mean, then centred second moment: fp32 like
// ATen's kernel), 16-byte accesses; HBM-bound: one read of x and y, one write of x and out.
#include "common.h"

namespace roma {
namespace {

template <typename T> __device__ __forceinline__ void load8(const T* p, float* f);
template <> __device__ __forceinline__ void load8<float>(const float* p, float* f) {
  const float4_t a = *reinterpret_cast<const float4_t*>(p), b = *reinterpret_cast<const float4_t*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[i] = a[i]; f[4 + i] = b[i]; }
}
template <> __device__ __forceinline__ void load8<half_t>(const half_t* p, float* f) { unpack16<half_t>(*reinterpret_cast<const u32x4*>(p), f); }
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float* f) { unpack16<bf16_t>(*reinterpret_cast<const u32x4*>(p), f); }

template <typename T> __device__ __forceinline__ void store8(T* p, const float* f);
template <> __device__ __forceinline__ void store8<float>(float* p, const float* f) {
  *reinterpret_cast<float4_t*>(p) = float4_t{f[0], f[1], f[2], f[3]};
  *reinterpret_cast<float4_t*>(p + 4) = float4_t{f[4], f[5], f[6], f[7]};
}
template <> __device__ __forceinline__ void store8<half_t>(half_t* p, const float* f) { *reinterpret_cast<u32x4*>(p) = pack16<half_t>(f); }
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float* f) { *reinterpret_cast<u32x4*>(p) = pack16<bf16_t>(f); }

// round-trip through the stream dtype: the statistics must see the values the stream really holds
template <typename T> __device__ __forceinline__ float rnd(float v) { return to_f32(from_f32<T>(v)); }

// NCH chunks of 512 elements (64 lanes x 8) cover the row
template <typename TX, typename TY, typename TO, int NCH>
__global__ __launch_bounds__(256) void add_layernorm_kernel(TX* __restrict__ x, const TY* __restrict__ y, const float* __restrict__ ls,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            TO* __restrict__ out, long rows, int C, long x_stride, long y_stride,
                                                            long out_stride, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  TX* xr = x + row * x_stride;
  float v[NCH][8];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e0 = c * 512 + lane * 8;
    if (e0 < C) {
      load8<TX>(xr + e0, v[c]);
      if (y) {
        float t[8];
        load8<TY>(y + row * y_stride + e0, t);
        if (ls) {
          float g[8];
          load8<float>(ls + e0, g);
#pragma unroll
          for (int i = 0; i < 8; ++i) v[c][i] = rnd<TX>(__builtin_fmaf(g[i], t[i], v[c][i]));
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[c][i] = rnd<TX>(v[c][i] + t[i]);
        }
        store8<TX>(xr + e0, v[c]);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) sum += v[c][i];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[c][i] = 0.f;
    }
  }
  TO* orow = out + row * out_stride;
  if (!gamma) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int e0 = c * 512 + lane * 8;
      if (e0 < C) store8<TO>(orow + e0, v[c]);
    }
    return;
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e0 = c * 512 + lane * 8;
    if (e0 < C) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float d = v[c][i] - mean;
        sq = __builtin_fmaf(d, d, sq);
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e0 = c * 512 + lane * 8;
    if (e0 < C) {
      float g[8], bt[8], o[8];
      load8<float>(gamma + e0, g);
      load8<float>(beta + e0, bt);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = __builtin_fmaf((v[c][i] - mean) * rstd, g[i], bt[i]);
      store8<TO>(orow + e0, o);
    }
  }
}

template <typename TX, typename TY, typename TO>
int launch_aln(void* x, const void* y, const float* ls, const float* gamma, const float* beta, void* out, long rows, int C,
               long xs, long ys, long os, float eps, hipStream_t s) {
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  const int nch = (C + 511) / 512;
#define ROMA_ALN(N)                                                                                                           \
  hipLaunchKernelGGL((add_layernorm_kernel<TX, TY, TO, N>), grid, block, 0, s, (TX*)x, (const TY*)y, ls, gamma, beta, (TO*)out, \
                     rows, C, xs, ys, os, eps)
  switch (nch) {
    case 1: ROMA_ALN(1); break;
    case 2: ROMA_ALN(2); break;
    case 3: ROMA_ALN(3); break;
    default: ROMA_ALN(4); break;
  }
#undef ROMA_ALN
  ROMA_CHECK_LAUNCH();
}

template <typename TX, typename TY>
int dispatch_out(int out_dtype, void* x, const void* y, const float* ls, const float* gamma, const float* beta, void* out, long rows,
                 int C, long xs, long ys, long os, float eps, hipStream_t s) {
  switch (out_dtype) {
    case ROMA_F32: return launch_aln<TX, TY, float>(x, y, ls, gamma, beta, out, rows, C, xs, ys, os, eps, s);
    case ROMA_F16: return launch_aln<TX, TY, half_t>(x, y, ls, gamma, beta, out, rows, C, xs, ys, os, eps, s);
    case ROMA_BF16: return launch_aln<TX, TY, bf16_t>(x, y, ls, gamma, beta, out, rows, C, xs, ys, os, eps, s);
  }
  return ROMA_E_DTYPE;
}

}  // namespace
}  // namespace roma

using namespace roma;

extern "C" int roma_add_layernorm(void* x, int x_dtype, long x_stride, const void* y, int y_dtype, long y_stride, const float* ls,
                                  const float* gamma, const float* beta, void* out, int out_dtype, long out_stride, long rows, int C,
                                  float eps, void* stream) {
  ROMA_REQUIRE(x && out, ROMA_E_ARG, "roma_add_layernorm: null pointer");
  ROMA_REQUIRE((gamma == nullptr) == (beta == nullptr), ROMA_E_ARG, "roma_add_layernorm: gamma and beta go together");
  ROMA_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= 2048, ROMA_E_SHAPE, "roma_add_layernorm: C=%d must be a multiple of 8, <= 2048", C);
  ROMA_REQUIRE(x_stride >= C && out_stride >= C && (!y || y_stride >= C) && x_stride % 8 == 0 && out_stride % 8 == 0 && y_stride % 8 == 0,
               ROMA_E_SHAPE, "roma_add_layernorm: row strides must be >= C and multiples of 8");
  ROMA_REQUIRE(aligned16(x) && aligned16(out) && aligned16(y) && aligned16(ls) && aligned16(gamma) && aligned16(beta), ROMA_E_ALIGN,
               "roma_add_layernorm: all bases must be 16-byte aligned");
  ROMA_REQUIRE(x_dtype >= ROMA_F32 && x_dtype <= ROMA_BF16 && out_dtype >= ROMA_F32 && out_dtype <= ROMA_BF16 &&
                   (!y || (y_dtype >= ROMA_F32 && y_dtype <= ROMA_BF16)),
               ROMA_E_DTYPE, "roma_add_layernorm: unknown dtype");
  // supported streams: fp32 stream with a 16-bit or fp32 branch (decoder transformer), 16-bit stream with a branch of the same
  // dtype (DINOv2 trunk)
  if (!y) y_dtype = x_dtype;
  hipStream_t s = static_cast<hipStream_t>(stream);
#define ROMA_GO(TX, TY) return dispatch_out<TX, TY>(out_dtype, x, y, ls, gamma, beta, out, rows, C, x_stride, y_stride, out_stride, eps, s)
  if (x_dtype == ROMA_F32 && y_dtype == ROMA_F32) ROMA_GO(float, float);
  if (x_dtype == ROMA_F32 && y_dtype == ROMA_F16) ROMA_GO(float, half_t);
  if (x_dtype == ROMA_F32 && y_dtype == ROMA_BF16) ROMA_GO(float, bf16_t);
  if (x_dtype == ROMA_F16 && y_dtype == ROMA_F16) ROMA_GO(half_t, half_t);
  if (x_dtype == ROMA_BF16 && y_dtype == ROMA_BF16) ROMA_GO(bf16_t, bf16_t);
#undef ROMA_GO
  set_error("roma_add_layernorm: unsupported dtype combination x=%d y=%d", x_dtype, y_dtype);
  return ROMA_E_UNSUPPORTED;
}
