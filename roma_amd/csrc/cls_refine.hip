// cls_to_flow_refine for gfx950 — reference: romatch/utils/utils.py:301-323 (called at matcher.py:383-385).
// softmax over C = res^2 anchor classes, arg-max, then a 5-point {mode-1, mode, mode+1, mode-res, mode+res} weighted
// mean of anchor coordinates, neighbour indices CLAMPED to [0, C-1] (so mode±1 may cross anchor rows and clamped
// duplicates are counted twice, exactly as the reference's gather does).  The softmax normaliser cancels in the
// ratio, so one pass for the max, one for exp/arg-max is enough; fp32 throughout.
// Row-major logits (class stride 1, the layout `to_out` produces): one wavefront per pixel, 64 lanes stride the
// classes (coalesced), wave reductions for max / arg-max.  Planar logits (the reference's (B,C,H,W)): one lane per
// pixel, lanes run along pixels (coalesced), a serial class loop.
#include <type_traits>
#include "common.h"

namespace roma {
namespace {

template <typename T> __device__ __forceinline__ float ld(const T* p, long i) { return to_f32(p[i]); }

__device__ __forceinline__ float anchor(int i, int res) { return -1.f + (2.f * i + 1.f) / res; }

__device__ __forceinline__ void refine5(float e[5], int mode, int C, int res, float& fx, float& fy) {
  const int idx[5] = {mode - 1, mode, mode + 1, mode - res, mode + res};
  float nx = 0.f, ny = 0.f, den = 0.f;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int k = min(max(idx[j], 0), C - 1);
    nx += e[j] * anchor(k % res, res);
    ny += e[j] * anchor(k / res, res);
    den += e[j];
  }
  fx = nx / den;
  fy = ny / den;
}

template <typename T>
__global__ __launch_bounds__(256) void cls_rows_kernel(const T* __restrict__ logits, float* __restrict__ flow,
                                                       float* __restrict__ cert, int B, int C, int HW, int res, long sb, long sp) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwave = (gridDim.x * blockDim.x) >> 6;
  for (int pix = wave; pix < B * HW; pix += nwave) {
    const int b = pix / HW, p = pix - b * HW;
    const T* row = logits + b * sb + p * sp;
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, ld(row, c));
    m = wave_max(m);
    // arg-max of exp(l - m) with the lowest index winning ties (what torch.max(dim) returns on CPU)
    float best = -1.f;
    int bi = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
      const float e = expf(ld(row, c) - m);
      if (e > best) { best = e; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) {
      float e[5];
      const int idx[5] = {bi - 1, bi, bi + 1, bi - res, bi + res};
#pragma unroll
      for (int j = 0; j < 5; ++j) e[j] = expf(ld(row, min(max(idx[j], 0), C - 1)) - m);
      float fx, fy;
      refine5(e, bi, C, res, fx, fy);
      flow[((size_t)b * 2 + 0) * HW + p] = fx;
      flow[((size_t)b * 2 + 1) * HW + p] = fy;
      if (cert) cert[(size_t)b * HW + p] = ld(row, C);
    }
  }
}

// 16-bit rows, C <= kVecMaxC: ONE pass over HBM with 16-byte loads.  A row of C+1 two-byte logits starts at an arbitrary even
// address (4097 elements per pixel: 8194-byte pitch), so the wave reads the 16-byte-ALIGNED packets that cover the row —
// packet pk holds elements 8 pk - h .. 8 pk - h + 7, h = elements between the aligned address below the row and the row — keeps
// them in registers (9 packets per lane cover 4608 elements), and takes max, exp and arg-max from there.  The two boundary
// packets are read element by element under a guard: nothing outside [row, row + C) is touched.
constexpr int kVecPk = 9, kVecMaxC = kVecPk * 64 * 8 - 8;

template <typename T>
__global__ __launch_bounds__(256) void cls_rows_vec_kernel(const T* __restrict__ logits, float* __restrict__ flow,
                                                           float* __restrict__ cert, int B, int C, int HW, int res, long sb, long sp) {
  static_assert(sizeof(T) == 2, "16-bit rows only");
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwave = (gridDim.x * blockDim.x) >> 6;
  const unsigned short ninf = std::is_same<T, half_t>::value ? 0xFC00u : 0xFF80u;      // -inf in fp16 / bf16
  for (int pix = wave; pix < B * HW; pix += nwave) {
    const int b = pix / HW, p = pix - b * HW;
    const T* row = logits + b * sb + p * sp;
    const int h = (int)((reinterpret_cast<uintptr_t>(row) & 15) >> 1);
    const unsigned short* a0 = reinterpret_cast<const unsigned short*>(row) - h;       // 16-byte aligned
    const int npk = (C + h + 7) >> 3;
    union Pk { uint4 q; unsigned short u[8]; } v[kVecPk];
#pragma unroll
    for (int k = 0; k < kVecPk; ++k) {
      const int pk = lane + 64 * k, e0 = pk * 8 - h;
      if (pk < npk && e0 >= 0 && e0 + 8 <= C) {
        v[k].q = *reinterpret_cast<const uint4*>(a0 + (size_t)pk * 8);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = e0 + j;
          v[k].u[j] = (pk < npk && c >= 0 && c < C) ? a0[(size_t)pk * 8 + j] : ninf;
        }
      }
    }
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < kVecPk; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) m = fmaxf(m, to_f32(__builtin_bit_cast(T, v[k].u[j])));
    m = wave_max(m);
    float best = -1.f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < kVecPk; ++k) {
      const int e0 = (lane + 64 * k) * 8 - h;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = e0 + j;
        const float e = expf(to_f32(__builtin_bit_cast(T, v[k].u[j])) - m);
        if (c >= 0 && c < C && e > best) { best = e; bi = c; }                          // ascending c within the lane
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) {
      float e[5];
      const int idx[5] = {bi - 1, bi, bi + 1, bi - res, bi + res};
#pragma unroll
      for (int j = 0; j < 5; ++j) e[j] = expf(ld(row, min(max(idx[j], 0), C - 1)) - m);
      float fx, fy;
      refine5(e, bi, C, res, fx, fy);
      flow[((size_t)b * 2 + 0) * HW + p] = fx;
      flow[((size_t)b * 2 + 1) * HW + p] = fy;
      if (cert) cert[(size_t)b * HW + p] = ld(row, C);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cls_planar_kernel(const T* __restrict__ logits, float* __restrict__ flow,
                                                         float* __restrict__ cert, int B, int C, int HW, int res, long sb, long sc,
                                                         long sp) {
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= B * HW) return;
  const int b = pix / HW, p = pix - b * HW;
  const T* base = logits + b * sb + p * sp;
  float m = -INFINITY;
  for (int c = 0; c < C; ++c) m = fmaxf(m, ld(base, c * sc));
  float best = -1.f;
  int bi = 0;
  for (int c = 0; c < C; ++c) {
    const float e = expf(ld(base, c * sc) - m);
    if (e > best) { best = e; bi = c; }
  }
  float e[5];
  const int idx[5] = {bi - 1, bi, bi + 1, bi - res, bi + res};
#pragma unroll
  for (int j = 0; j < 5; ++j) e[j] = expf(ld(base, (long)min(max(idx[j], 0), C - 1) * sc) - m);
  float fx, fy;
  refine5(e, bi, C, res, fx, fy);
  flow[((size_t)b * 2 + 0) * HW + p] = fx;
  flow[((size_t)b * 2 + 1) * HW + p] = fy;
  if (cert) cert[(size_t)b * HW + p] = ld(base, (long)C * sc);
}

}  // namespace
}  // namespace roma

using namespace roma;

extern "C" int roma_cls_to_flow_refine(const void* logits, float* flow_out, float* cert_out, int B, int C, int HW, long stride_b,
                                       long stride_c, long stride_p, int dtype, void* stream) {
  ROMA_REQUIRE(logits && flow_out, ROMA_E_ARG, "roma_cls_to_flow_refine: null pointer");
  ROMA_REQUIRE(B > 0 && C > 0 && HW > 0, ROMA_E_SHAPE, "roma_cls_to_flow_refine: bad shape");
  int res = (int)lroundf(sqrtf((float)C));
  ROMA_REQUIRE(res * res == C, ROMA_E_SHAPE, "roma_cls_to_flow_refine: C=%d is not a square number of anchors", C);
  ROMA_REQUIRE(dtype >= ROMA_F32 && dtype <= ROMA_BF16, ROMA_E_DTYPE, "roma_cls_to_flow_refine: unknown dtype %d", dtype);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int npix = B * HW;
  const dim3 rgrid((npix + 3) / 4 > 4096 ? 4096 : (npix + 3) / 4);
  if (stride_c == 1 && dtype != ROMA_F32 && C <= kVecMaxC) {        // 16-bit rows: one pass, 16-byte loads
    if (dtype == ROMA_F16)
      hipLaunchKernelGGL((cls_rows_vec_kernel<half_t>), rgrid, dim3(256), 0, s, (const half_t*)logits, flow_out, cert_out, B, C, HW, res,
                         stride_b, stride_p);
    else
      hipLaunchKernelGGL((cls_rows_vec_kernel<bf16_t>), rgrid, dim3(256), 0, s, (const bf16_t*)logits, flow_out, cert_out, B, C, HW, res,
                         stride_b, stride_p);
    ROMA_CHECK_LAUNCH();
  }
#define ROMA_CLS(T)                                                                                                            \
  if (stride_c == 1)                                                                                                           \
    hipLaunchKernelGGL((cls_rows_kernel<T>), dim3((npix + 3) / 4 > 4096 ? 4096 : (npix + 3) / 4), dim3(256), 0, s, (const T*)logits, \
                       flow_out, cert_out, B, C, HW, res, stride_b, stride_p);                                                 \
  else                                                                                                                         \
    hipLaunchKernelGGL((cls_planar_kernel<T>), dim3((npix + 255) / 256), dim3(256), 0, s, (const T*)logits, flow_out, cert_out, B, \
                       C, HW, res, stride_b, stride_c, stride_p);
  if (dtype == ROMA_F32) { ROMA_CLS(float) } else if (dtype == ROMA_F16) { ROMA_CLS(half_t) } else { ROMA_CLS(bf16_t) }
#undef ROMA_CLS
  ROMA_CHECK_LAUNCH();
}
