// Bilinear sampling family for gfx950: grid_sample warp, planar bilinear resize, displacement embedding, flow update.
// References: matcher.py:109 (F.grid_sample in ConvRefiner.forward), matcher.py:349-360,408-417,657-659 (F.interpolate),
// matcher.py:111-120 (disp_emb), matcher.py:397-402 (flow / certainty update).
// All of these are HBM-bound elementwise/gather kernels: 16-byte packets along the channel axis of channels-last
// features, fp32 arithmetic, one pass.
#include "common.h"

namespace roma {
namespace {

struct Corner {
  int x0, y0;
  float w00, w01, w10, w11;  // wYX
  bool v00, v01, v10, v11;
};

__device__ __forceinline__ Corner corners(float fx, float fy, int Hs, int Ws) {
  float px = ((fx + 1.f) * Ws - 1.f) * 0.5f, py = ((fy + 1.f) * Hs - 1.f) * 0.5f;
  if (!(px > -1e6f && px < 1e6f)) px = -1e6f;
  if (!(py > -1e6f && py < 1e6f)) py = -1e6f;
  const float fx0 = floorf(px), fy0 = floorf(py);
  const float ax = px - fx0, ay = py - fy0;
  Corner c;
  c.x0 = (int)fx0;
  c.y0 = (int)fy0;
  c.w00 = (1.f - ay) * (1.f - ax);
  c.w01 = (1.f - ay) * ax;
  c.w10 = ay * (1.f - ax);
  c.w11 = ay * ax;
  const bool xa = c.x0 >= 0 && c.x0 < Ws, xb = c.x0 + 1 >= 0 && c.x0 + 1 < Ws;
  const bool ya = c.y0 >= 0 && c.y0 < Hs, yb = c.y0 + 1 >= 0 && c.y0 + 1 < Hs;
  c.v00 = ya && xa;
  c.v01 = ya && xb;
  c.v10 = yb && xa;
  c.v11 = yb && xb;
  return c;
}

// channels-last source and destination, C a multiple of the packet width
template <typename T>
__global__ __launch_bounds__(256) void warp_nhwc_kernel(const T* __restrict__ src, const float* __restrict__ flow,
                                                        T* __restrict__ dst, int B, int C, int Hs, int Ws, int H, int W,
                                                        int src_pitch, int dst_pitch, int src_shift) {
  constexpr int E16 = ElemTraits<T>::kPer16B;
  const int PK = C / E16;
  const size_t total = (size_t)B * H * W * PK;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % PK);
    const size_t pix = i / PK;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const int b = (int)(pix / ((size_t)W * H));
    const float fx = flow[((size_t)(b * 2 + 0) * H + y) * W + x];
    const float fy = flow[((size_t)(b * 2 + 1) * H + y) * W + x];
    const Corner c = corners(fx, fy, Hs, Ws);
    // all four taps are loaded unconditionally from clamped coordinates and masked through their weights: a guarded
    // load would make hipcc branch and wait vmcnt(0) per tap (four serial L2 round trips)
    const T* base = src + (size_t)((b + src_shift) % B) * Hs * Ws * src_pitch + (size_t)k * E16;
    const int xa = min(max(c.x0, 0), Ws - 1), xb = min(max(c.x0 + 1, 0), Ws - 1);
    const int ya = min(max(c.y0, 0), Hs - 1), yb = min(max(c.y0 + 1, 0), Hs - 1);
    const u32x4 v00 = *reinterpret_cast<const u32x4*>(base + ((size_t)ya * Ws + xa) * src_pitch);
    const u32x4 v01 = *reinterpret_cast<const u32x4*>(base + ((size_t)ya * Ws + xb) * src_pitch);
    const u32x4 v10 = *reinterpret_cast<const u32x4*>(base + ((size_t)yb * Ws + xa) * src_pitch);
    const u32x4 v11 = *reinterpret_cast<const u32x4*>(base + ((size_t)yb * Ws + xb) * src_pitch);
    const float w00 = c.v00 ? c.w00 : 0.f, w01 = c.v01 ? c.w01 : 0.f, w10 = c.v10 ? c.w10 : 0.f, w11 = c.v11 ? c.w11 : 0.f;
    float f00[E16], f01[E16], f10[E16], f11[E16], acc[E16];
    unpack16<T>(v00, f00);
    unpack16<T>(v01, f01);
    unpack16<T>(v10, f10);
    unpack16<T>(v11, f11);
#pragma unroll
    for (int e = 0; e < E16; ++e)
      acc[e] = __builtin_fmaf(w11, f11[e], __builtin_fmaf(w10, f10[e], __builtin_fmaf(w01, f01[e], w00 * f00[e])));
    *reinterpret_cast<u32x4*>(dst + pix * dst_pitch + (size_t)k * E16) = pack16<T>(acc);
  }
}

// channels-last, 16-bit, FEW channels that are not a whole packet (the scale-1 refiner warps 9 channels inside its 24-channel pixel:
// source = channels [0, 9) of the concat buffer, destination = channels [9, 18) of the same pixels): one thread per PIXEL — flow and
// corner arithmetic once instead of once per channel, every tap one or two 16-byte loads of the pixel's first packets (the bytes behind
// channel C are inside the pixel's pitch and ignored), 32-bit stores where the destination slice allows.  The element-per-thread kernel
// below ran this call at 0.65 TB/s (2-byte gathers, the corner arithmetic nine times per pixel): 100 us at 864 x 864 x 2.
template <typename T, int NPK>
__global__ __launch_bounds__(256) void warp_small_kernel(const T* __restrict__ src, const float* __restrict__ flow, T* __restrict__ dst,
                                                         int B, int C, int Hs, int Ws, int H, int W, int src_pitch, int dst_pitch,
                                                         int src_shift) {
  static_assert(sizeof(T) == 2, "16-bit elements");
  const size_t total = (size_t)B * H * W;
  const bool odd = (reinterpret_cast<uintptr_t>(dst) & 2) != 0;  // the slice starts in the upper half of a 32-bit word (dst_pitch is even)
  for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const int b = (int)(pix / ((size_t)W * H));
    const float fx = flow[((size_t)(b * 2 + 0) * H + y) * W + x];
    const float fy = flow[((size_t)(b * 2 + 1) * H + y) * W + x];
    const Corner c = corners(fx, fy, Hs, Ws);
    const T* base = src + (size_t)((b + src_shift) % B) * Hs * Ws * src_pitch;
    const int xa = min(max(c.x0, 0), Ws - 1), xb = min(max(c.x0 + 1, 0), Ws - 1);
    const int ya = min(max(c.y0, 0), Hs - 1), yb = min(max(c.y0 + 1, 0), Hs - 1);
    const T* p00 = base + ((size_t)ya * Ws + xa) * src_pitch;
    const T* p01 = base + ((size_t)ya * Ws + xb) * src_pitch;
    const T* p10 = base + ((size_t)yb * Ws + xa) * src_pitch;
    const T* p11 = base + ((size_t)yb * Ws + xb) * src_pitch;
    u32x4 v[4][NPK];
#pragma unroll
    for (int k = 0; k < NPK; ++k) {                              // unconditional, clamped: masked through the weights (see above)
      v[0][k] = *reinterpret_cast<const u32x4*>(p00 + k * 8);
      v[1][k] = *reinterpret_cast<const u32x4*>(p01 + k * 8);
      v[2][k] = *reinterpret_cast<const u32x4*>(p10 + k * 8);
      v[3][k] = *reinterpret_cast<const u32x4*>(p11 + k * 8);
    }
    const float w00 = c.v00 ? c.w00 : 0.f, w01 = c.v01 ? c.w01 : 0.f, w10 = c.v10 ? c.w10 : 0.f, w11 = c.v11 ? c.w11 : 0.f;
    T o[8 * NPK + 2];
#pragma unroll
    for (int k = 0; k < NPK; ++k) {
      float f00[8], f01[8], f10[8], f11[8];
      unpack16<T>(v[0][k], f00);
      unpack16<T>(v[1][k], f01);
      unpack16<T>(v[2][k], f10);
      unpack16<T>(v[3][k], f11);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        o[8 * k + e] = from_f32<T>(__builtin_fmaf(w11, f11[e], __builtin_fmaf(w10, f10[e], __builtin_fmaf(w01, f01[e], w00 * f00[e]))));
    }
    T* d = dst + pix * dst_pitch;
    int e = 0;
    if (odd) { d[0] = o[0]; e = 1; }
#pragma unroll
    for (int j = 0; j < 4 * NPK; ++j) {                          // pairs on 32-bit boundaries
      const int ee = e + 2 * j;
      if (ee + 1 < C) {
        const uint32_t lo = __builtin_bit_cast(uint16_t, odd ? o[2 * j + 1] : o[2 * j]), hi = __builtin_bit_cast(uint16_t, odd ? o[2 * j + 2] : o[2 * j + 1]);
        *reinterpret_cast<uint32_t*>(d + ee) = lo | (hi << 16);
      } else if (ee < C) {
        d[ee] = odd ? o[2 * j + 1] : o[2 * j];
      }
    }
  }
}

// any layout combination, scalar accesses; x runs fastest over lanes for planar data, c for channels-last
template <typename T>
__global__ __launch_bounds__(256) void warp_generic_kernel(const T* __restrict__ src, const float* __restrict__ flow,
                                                           T* __restrict__ dst, int B, int C, int Hs, int Ws, int H, int W,
                                                           int src_nhwc, int src_pitch, int dst_nhwc, int dst_pitch, int src_shift) {
  const size_t total = (size_t)B * C * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int b, ch, y, x;
    if (dst_nhwc) {
      ch = (int)(i % C);
      size_t r = i / C;
      x = (int)(r % W); r /= W;
      y = (int)(r % H);
      b = (int)(r / H);
    } else {
      x = (int)(i % W);
      size_t r = i / W;
      y = (int)(r % H); r /= H;
      ch = (int)(r % C);
      b = (int)(r / C);
    }
    const float fx = flow[((size_t)(b * 2 + 0) * H + y) * W + x];
    const float fy = flow[((size_t)(b * 2 + 1) * H + y) * W + x];
    const Corner c = corners(fx, fy, Hs, Ws);
    const int bs = (b + src_shift) % B;
    auto at = [&](int yy, int xx) -> float {
      const size_t o = src_nhwc ? (((size_t)bs * Hs + yy) * Ws + xx) * src_pitch + ch
                                : (((size_t)bs * src_pitch + ch) * Hs + yy) * Ws + xx;
      return to_f32(src[o]);
    };
    float v = 0.f;
    if (c.v00) v = __builtin_fmaf(c.w00, at(c.y0, c.x0), v);
    if (c.v01) v = __builtin_fmaf(c.w01, at(c.y0, c.x0 + 1), v);
    if (c.v10) v = __builtin_fmaf(c.w10, at(c.y0 + 1, c.x0), v);
    if (c.v11) v = __builtin_fmaf(c.w11, at(c.y0 + 1, c.x0 + 1), v);
    const size_t o = dst_nhwc ? (((size_t)b * H + y) * W + x) * dst_pitch + ch : (((size_t)b * dst_pitch + ch) * H + y) * W + x;
    dst[o] = from_f32<T>(v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void disp_emb_kernel(const float* __restrict__ flow, const float* __restrict__ weight,
                                                       const float* __restrict__ bias, T* __restrict__ dst, int B, int E, int H,
                                                       int W, float gain, int dst_nhwc, int dst_pitch) {
  const size_t total = (size_t)B * E * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int b, e, y, x;
    if (dst_nhwc) {
      e = (int)(i % E);
      size_t r = i / E;
      x = (int)(r % W); r /= W;
      y = (int)(r % H);
      b = (int)(r / H);
    } else {
      x = (int)(i % W);
      size_t r = i / W;
      y = (int)(r % H); r /= H;
      e = (int)(r % E);
      b = (int)(r / E);
    }
    const float dx = gain * (flow[((size_t)(b * 2 + 0) * H + y) * W + x] - pix_center(x, W));
    const float dy = gain * (flow[((size_t)(b * 2 + 1) * H + y) * W + x] - pix_center(y, H));
    const float v = __builtin_fmaf(weight[2 * e + 1], dy, __builtin_fmaf(weight[2 * e], dx, bias[e]));
    const size_t o = dst_nhwc ? (((size_t)b * H + y) * W + x) * dst_pitch + e : (((size_t)b * dst_pitch + e) * H + y) * W + x;
    dst[o] = from_f32<T>(v);
  }
}

__global__ __launch_bounds__(256) void interp_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int Hi, int Wi,
                                                     int Ho, int Wo) {
  const size_t total = (size_t)N * Ho * Wo;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int xo = (int)(i % Wo);
    const int yo = (int)((i / Wo) % Ho);
    const int n = (int)(i / ((size_t)Wo * Ho));
    int y0, y1, x0, x1;
    float ly, lx;
    interp_src(yo, Hi, Ho, y0, y1, ly);
    interp_src(xo, Wi, Wo, x0, x1, lx);
    const float* p = x + (size_t)n * Hi * Wi;
    const float top = (1.f - lx) * p[(size_t)y0 * Wi + x0] + lx * p[(size_t)y0 * Wi + x1];
    const float bot = (1.f - lx) * p[(size_t)y1 * Wi + x0] + lx * p[(size_t)y1 * Wi + x1];
    y[i] = (1.f - ly) * top + ly * bot;
  }
}

__global__ __launch_bounds__(256) void flow_update_kernel(float* __restrict__ flow, float* __restrict__ cert,
                                                          const float* __restrict__ cert_in, const float* __restrict__ delta, int B,
                                                          int HW, float sx, float sy) {
  const size_t total = (size_t)B * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW);
    const int p = (int)(i - (size_t)b * HW);
    const float* d = delta + (size_t)b * 3 * HW + p;
    float* f = flow + (size_t)b * 2 * HW + p;
    f[0] += sx * d[0];
    f[HW] += sy * d[HW];
    cert[i] = (cert_in ? cert_in[i] : 0.f) + d[2 * (size_t)HW];
  }
}

inline int grid_for(size_t total) {
  size_t g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace
}  // namespace roma

using namespace roma;

extern "C" int roma_warp_bilinear(const void* src, const float* flow, void* dst, int B, int C, int Hs, int Ws, int H, int W,
                                  int dtype, int layout, int src_pitch, int dst_layout, int dst_pitch, int src_batch_shift, void* stream) {
  ROMA_REQUIRE(src_batch_shift >= 0 && src_batch_shift < (B > 0 ? B : 1), ROMA_E_ARG, "roma_warp_bilinear: src_batch_shift %d outside [0, B)", src_batch_shift);
  ROMA_REQUIRE(src && flow && dst, ROMA_E_ARG, "roma_warp_bilinear: null pointer");
  ROMA_REQUIRE(B > 0 && C > 0 && Hs > 0 && Ws > 0 && H > 0 && W > 0, ROMA_E_SHAPE, "roma_warp_bilinear: bad shape");
  ROMA_REQUIRE(src_pitch >= C && dst_pitch >= C, ROMA_E_SHAPE, "roma_warp_bilinear: pitch smaller than channel count");
  ROMA_REQUIRE(dtype >= ROMA_F32 && dtype <= ROMA_BF16, ROMA_E_DTYPE, "roma_warp_bilinear: unknown dtype %d", dtype);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int e16 = dtype == ROMA_F32 ? 4 : 8;
  const bool vec = layout == ROMA_NHWC && dst_layout == ROMA_NHWC && C % e16 == 0 && src_pitch % e16 == 0 &&
                   dst_pitch % e16 == 0 && aligned16(src) && aligned16(dst);
  const int npk = (C + 7) / 8;
  const bool small = !vec && dtype != ROMA_F32 && layout == ROMA_NHWC && dst_layout == ROMA_NHWC && C <= 16 && src_pitch % 8 == 0 &&
                     src_pitch >= 8 * npk && dst_pitch % 2 == 0 && aligned16(src);
  if (small) {
    const size_t px = (size_t)B * H * W;
#define ROMA_WARP_S(T)                                                                                                                 \
  if (npk == 1)                                                                                                                        \
    hipLaunchKernelGGL((warp_small_kernel<T, 1>), dim3(grid_for(px)), dim3(256), 0, s, (const T*)src, flow, (T*)dst, B, C, Hs, Ws, H, W, \
                       src_pitch, dst_pitch, src_batch_shift);                                                                         \
  else                                                                                                                                 \
    hipLaunchKernelGGL((warp_small_kernel<T, 2>), dim3(grid_for(px)), dim3(256), 0, s, (const T*)src, flow, (T*)dst, B, C, Hs, Ws, H, W, \
                       src_pitch, dst_pitch, src_batch_shift);
    if (dtype == ROMA_F16) { ROMA_WARP_S(half_t) } else { ROMA_WARP_S(bf16_t) }
#undef ROMA_WARP_S
    ROMA_CHECK_LAUNCH();
  }
  const size_t total = vec ? (size_t)B * H * W * (C / e16) : (size_t)B * H * W * C;
#define ROMA_WARP(T)                                                                                                     \
  if (vec)                                                                                                               \
    hipLaunchKernelGGL((warp_nhwc_kernel<T>), dim3(grid_for(total)), dim3(256), 0, s, (const T*)src, flow, (T*)dst, B, C, Hs, \
                       Ws, H, W, src_pitch, dst_pitch, src_batch_shift);                                                 \
  else                                                                                                                   \
    hipLaunchKernelGGL((warp_generic_kernel<T>), dim3(grid_for(total)), dim3(256), 0, s, (const T*)src, flow, (T*)dst, B, C, \
                       Hs, Ws, H, W, layout == ROMA_NHWC, src_pitch, dst_layout == ROMA_NHWC, dst_pitch, src_batch_shift);
  if (dtype == ROMA_F32) { ROMA_WARP(float) } else if (dtype == ROMA_F16) { ROMA_WARP(half_t) } else { ROMA_WARP(bf16_t) }
#undef ROMA_WARP
  ROMA_CHECK_LAUNCH();
}

extern "C" int roma_disp_emb(const float* flow, const float* weight, const float* bias, void* dst, int B, int E, int H, int W,
                             float gain, int dtype, int dst_layout, int dst_pitch, void* stream) {
  ROMA_REQUIRE(flow && weight && bias && dst, ROMA_E_ARG, "roma_disp_emb: null pointer");
  ROMA_REQUIRE(B > 0 && E > 0 && H > 0 && W > 0 && dst_pitch >= E, ROMA_E_SHAPE, "roma_disp_emb: bad shape");
  ROMA_REQUIRE(dtype >= ROMA_F32 && dtype <= ROMA_BF16, ROMA_E_DTYPE, "roma_disp_emb: unknown dtype %d", dtype);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t total = (size_t)B * E * H * W;
  const int nhwc = dst_layout == ROMA_NHWC;
  if (dtype == ROMA_F32)
    hipLaunchKernelGGL((disp_emb_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s, flow, weight, bias, (float*)dst, B, E, H, W, gain, nhwc, dst_pitch);
  else if (dtype == ROMA_F16)
    hipLaunchKernelGGL((disp_emb_kernel<half_t>), dim3(grid_for(total)), dim3(256), 0, s, flow, weight, bias, (half_t*)dst, B, E, H, W, gain, nhwc, dst_pitch);
  else
    hipLaunchKernelGGL((disp_emb_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, s, flow, weight, bias, (bf16_t*)dst, B, E, H, W, gain, nhwc, dst_pitch);
  ROMA_CHECK_LAUNCH();
}

extern "C" int roma_interp_bilinear(const float* x, float* y, int N, int Hi, int Wi, int Ho, int Wo, void* stream) {
  ROMA_REQUIRE(x && y, ROMA_E_ARG, "roma_interp_bilinear: null pointer");
  ROMA_REQUIRE(N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, ROMA_E_SHAPE, "roma_interp_bilinear: bad shape");
  hipLaunchKernelGGL(interp_kernel, dim3(grid_for((size_t)N * Ho * Wo)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, N, Hi, Wi, Ho, Wo);
  ROMA_CHECK_LAUNCH();
}

extern "C" int roma_flow_update(float* flow, float* cert, const float* cert_in, const float* delta, int B, int H, int W, float sx,
                                float sy, void* stream) {
  ROMA_REQUIRE(flow && cert && delta, ROMA_E_ARG, "roma_flow_update: null pointer");
  ROMA_REQUIRE(B > 0 && H > 0 && W > 0, ROMA_E_SHAPE, "roma_flow_update: bad shape");
  hipLaunchKernelGGL(flow_update_kernel, dim3(grid_for((size_t)B * H * W)), dim3(256), 0, static_cast<hipStream_t>(stream), flow, cert, cert_in, delta, B, H * W, sx, sy);
  ROMA_CHECK_LAUNCH();
}
