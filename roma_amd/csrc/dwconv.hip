// Depthwise 5x5 convolution + BatchNorm(eval, folded) + ReLU, channels-last, for gfx950 — the front half of every
// ConvRefiner block: romatch/models/matcher.py:77-103 (create_block: Conv2d(groups=C, k=5, pad=2) -> BatchNorm2d ->
// ReLU), applied 9x per refiner at matcher.py:139-140.
//   y[b,h,w,c] = relu(scale[c] * sum_{dy,dx} w[dy*5+dx][c] * x[b,h+dy-2,w+dx-2,c] + shift[c])
// HBM-bound (one read + one write of the activation): each thread owns one 16-byte channel packet and a strip of
// XS output pixels along W, streams the (XS+4) input columns of each of the 5 rows through registers once and
// reuses every loaded packet for up to 5 taps x XS outputs; fp32 accumulation.  Lanes run along the channel
// packets, so a wavefront's loads are 1 KiB contiguous for C >= 512 (fp16).
#include "common.h"

namespace roma {
namespace {

constexpr int XS = 4;

template <typename T>
__global__ __launch_bounds__(256) void dwconv5x5_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        T* __restrict__ y, int B, int C, int H, int W, int x_pitch, int y_pitch) {
  constexpr int E = ElemTraits<T>::kPer16B;
  const int PK = C / E;
  const int WS = (W + XS - 1) / XS;
  const size_t total = (size_t)B * H * WS * PK;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % PK);
    size_t r = i / PK;
    const int xs = (int)(r % WS) * XS; r /= WS;
    const int yo = (int)(r % H);
    const int b = (int)(r / H);
    const int c0 = k * E;
    float acc[XS][E];
#pragma unroll
    for (int o = 0; o < XS; ++o)
#pragma unroll
      for (int e = 0; e < E; ++e) acc[o][e] = 0.f;
    const T* xb = x + (size_t)b * H * W * x_pitch + c0;
    // interior strips (all but a 2-pixel frame) run with unconditional loads; a guarded load makes hipcc branch and
    // wait vmcnt(0) per element.  Border strips clamp the address and zero the tap through its weight.
    const bool interior = yo >= 2 && yo + 2 < H && xs >= 2 && xs + XS + 2 <= W;
    if (interior) {
#pragma unroll
      for (int dy = 0; dy < 5; ++dy) {
        float wr[5][E];
#pragma unroll
        for (int dx = 0; dx < 5; ++dx)
#pragma unroll
          for (int e = 0; e < E; e += 4)
            *reinterpret_cast<float4_t*>(&wr[dx][e]) = *reinterpret_cast<const float4_t*>(w + (size_t)(dy * 5 + dx) * C + c0 + e);
        const T* row = xb + ((size_t)(yo + dy - 2) * W + (xs - 2)) * x_pitch;
        u32x4 raw[XS + 4];
#pragma unroll
        for (int cx = 0; cx < XS + 4; ++cx) raw[cx] = *reinterpret_cast<const u32x4*>(row + (size_t)cx * x_pitch);
#pragma unroll
        for (int cx = 0; cx < XS + 4; ++cx) {
          float f[E];
          unpack16<T>(raw[cx], f);
#pragma unroll
          for (int dx = 0; dx < 5; ++dx) {
            const int o = cx - dx;
            if (o >= 0 && o < XS) {
#pragma unroll
              for (int e = 0; e < E; ++e) acc[o][e] = __builtin_fmaf(wr[dx][e], f[e], acc[o][e]);
            }
          }
        }
      }
    } else {
#pragma unroll 1
      for (int dy = 0; dy < 5; ++dy) {
        const int yi = yo + dy - 2;
        const float my = (yi >= 0 && yi < H) ? 1.f : 0.f;
        const int yc = min(max(yi, 0), H - 1);
        float wr[5][E];
#pragma unroll
        for (int dx = 0; dx < 5; ++dx)
#pragma unroll
          for (int e = 0; e < E; e += 4)
            *reinterpret_cast<float4_t*>(&wr[dx][e]) = *reinterpret_cast<const float4_t*>(w + (size_t)(dy * 5 + dx) * C + c0 + e);
        const T* row = xb + (size_t)yc * W * x_pitch;
#pragma unroll
        for (int cx = 0; cx < XS + 4; ++cx) {
          const int xi = xs + cx - 2;
          const float m = (xi >= 0 && xi < W) ? my : 0.f;
          const int xc = min(max(xi, 0), W - 1);
          float f[E];
          unpack16<T>(*reinterpret_cast<const u32x4*>(row + (size_t)xc * x_pitch), f);
#pragma unroll
          for (int dx = 0; dx < 5; ++dx) {
            const int o = cx - dx;
            if (o >= 0 && o < XS) {
#pragma unroll
              for (int e = 0; e < E; ++e) acc[o][e] = __builtin_fmaf(wr[dx][e] * m, f[e], acc[o][e]);
            }
          }
        }
      }
    }
    float sc[E], sh[E];
#pragma unroll
    for (int e = 0; e < E; e += 4) {
      *reinterpret_cast<float4_t*>(&sc[e]) = *reinterpret_cast<const float4_t*>(scale + c0 + e);
      *reinterpret_cast<float4_t*>(&sh[e]) = *reinterpret_cast<const float4_t*>(shift + c0 + e);
    }
    T* yrow = y + (((size_t)b * H + yo) * W) * y_pitch + c0;
#pragma unroll
    for (int o = 0; o < XS; ++o) {
      if (xs + o >= W) break;
      float v[E];
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = fmaxf(__builtin_fmaf(acc[o][e], sc[e], sh[e]), 0.f);
      *reinterpret_cast<u32x4*>(yrow + (size_t)(xs + o) * y_pitch) = pack16<T>(v);
    }
  }
}

}  // namespace
}  // namespace roma

using namespace roma;

extern "C" int roma_dwconv5x5_bn_relu(const void* x, const float* w, const float* scale, const float* shift, void* y, int B, int C,
                                      int H, int W, int dtype, int x_pitch, int y_pitch, void* stream) {
  ROMA_REQUIRE(x && w && scale && shift && y, ROMA_E_ARG, "roma_dwconv5x5_bn_relu: null pointer");
  ROMA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && x_pitch >= C && y_pitch >= C, ROMA_E_SHAPE, "roma_dwconv5x5_bn_relu: bad shape");
  ROMA_REQUIRE(dtype >= ROMA_F32 && dtype <= ROMA_BF16, ROMA_E_DTYPE, "roma_dwconv5x5_bn_relu: unknown dtype %d", dtype);
  const int e = dtype == ROMA_F32 ? 4 : 8;
  ROMA_REQUIRE(C % e == 0 && x_pitch % e == 0 && y_pitch % e == 0 && aligned16(x) && aligned16(y) && aligned16(w) &&
                   aligned16(scale) && aligned16(shift),
               ROMA_E_ALIGN, "roma_dwconv5x5_bn_relu: C and pitches must be multiples of %d, all bases 16-byte aligned", e);
  const size_t total = (size_t)B * H * ((W + XS - 1) / XS) * (C / e);
  size_t g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == ROMA_F32)
    hipLaunchKernelGGL((dwconv5x5_kernel<float>), dim3((int)g), dim3(256), 0, s, (const float*)x, w, scale, shift, (float*)y, B, C, H, W, x_pitch, y_pitch);
  else if (dtype == ROMA_F16)
    hipLaunchKernelGGL((dwconv5x5_kernel<half_t>), dim3((int)g), dim3(256), 0, s, (const half_t*)x, w, scale, shift, (half_t*)y, B, C, H, W, x_pitch, y_pitch);
  else
    hipLaunchKernelGGL((dwconv5x5_kernel<bf16_t>), dim3((int)g), dim3(256), 0, s, (const bf16_t*)x, w, scale, shift, (bf16_t*)y, B, C, H, W, x_pitch, y_pitch);
  ROMA_CHECK_LAUNCH();
}
