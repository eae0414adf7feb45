// Depthwise 5x5 convolution + BatchNorm(eval, folded) + ReLU, channels-last, for gfx950 — the front half of every
// ConvRefiner block: romatch/models/matcher.py:77-103 (create_block: Conv2d(groups=C, k=5, pad=2) -> BatchNorm2d ->
// ReLU), applied 9x per refiner at matcher.py:139-140.
//   y[b,h,w,c] = relu(scale[c] * sum_{dy,dx} w[dy*5+dx][c] * x[b,h+dy-2,w+dx-2,c] + shift[c])
// HBM-bound (one read + one write of the activation): each thread owns one 16-byte channel packet and a strip of
// XS output pixels along W, streams the (XS+4) input columns of each of the 5 rows through registers once and
// reuses every loaded packet for up to 5 taps x XS outputs; fp32 accumulation.  Lanes run along the channel
// packets, so a wavefront's loads are 1 KiB contiguous for C >= 512 (fp16).
#include <cstdlib>
#include "common.h"

namespace roma {
namespace {

// Register tile per thread: XS x YS output pixels of one 16-byte channel packet; the (XS+4) x (YS+4) input patch is
// streamed row by row through registers.  Measured on the nine refiner shapes (fp16, B=2, sum of times): 4x1 562 us,
// 4x2 496 us, 8x1 467 us (default), 8x2 574 us, 4x4 607 us; capping registers for occupancy (launch bounds 3-4 waves
// per SIMD) spills and is 4x slower, so MINW stays 1.
template <typename T, int XS, int YS, int MINW>
__global__ __launch_bounds__(256, MINW) void dwconv5x5_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        T* __restrict__ y, int B, int C, int H, int W, int x_pitch, int y_pitch) {
  constexpr int E = ElemTraits<T>::kPer16B;
  const int PK = C / E;
  const int WS = (W + XS - 1) / XS, HS = (H + YS - 1) / YS;
  const size_t total = (size_t)B * HS * WS * PK;
  // XCD-aware block order: the dispatcher deals consecutive workgroups round-robin to the 8 XCDs, so in launch order the strip
  // below a strip (same columns, next row: 4 of its 5 input rows are the same) lands on another XCD's L2 and the rows are
  // fetched over the fabric again — rocprofv3 FETCH_SIZE: 2.8x the input at C = 576, 216 x 216.  With the remap every XCD walks
  // a contiguous band of rows and the halo rows stay in its own L2 (reuse distance: one image row, 0.25 MB).
  const size_t wid = (size_t)xcd_remap(blockIdx.x, gridDim.x);
  for (size_t i = wid * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % PK);
    size_t r = i / PK;
    const int xs = (int)(r % WS) * XS; r /= WS;
    const int ys = (int)(r % HS) * YS;
    const int b = (int)(r / HS);
    const int c0 = k * E;
    float acc[YS][XS][E];
#pragma unroll
    for (int oy = 0; oy < YS; ++oy)
#pragma unroll
      for (int o = 0; o < XS; ++o)
#pragma unroll
        for (int e = 0; e < E; ++e) acc[oy][o][e] = 0.f;
    const T* xb = x + (size_t)b * H * W * x_pitch + c0;
    // interior patches (all but a 2-pixel frame) run with unconditional loads; a guarded load makes hipcc branch and
    // wait vmcnt(0) per element.  Border patches clamp the address and zero the tap through a mask.
    const bool interior = ys >= 2 && ys + YS + 2 <= H && xs >= 2 && xs + XS + 2 <= W;
#pragma unroll 1
    for (int iy = 0; iy < YS + 4; ++iy) {                           // rolled: one input row in registers at a time
      const int yi = ys + iy - 2;
      const float my = (yi >= 0 && yi < H) ? 1.f : 0.f;
      const T* row = xb + (size_t)min(max(yi, 0), H - 1) * W * x_pitch;
      u32x4 raw[XS + 4];
      if (interior) {
#pragma unroll
        for (int cx = 0; cx < XS + 4; ++cx) raw[cx] = *reinterpret_cast<const u32x4*>(row + (size_t)(xs - 2 + cx) * x_pitch);
      } else {
#pragma unroll
        for (int cx = 0; cx < XS + 4; ++cx) raw[cx] = *reinterpret_cast<const u32x4*>(row + (size_t)min(max(xs - 2 + cx, 0), W - 1) * x_pitch);
      }
#pragma unroll
      for (int oy = 0; oy < YS; ++oy) {
        const int dy = iy - oy;                                   // tap row of output row oy fed by input row iy
        if (dy < 0 || dy > 4) continue;
        float wr[5][E];
#pragma unroll
        for (int dx = 0; dx < 5; ++dx)
#pragma unroll
          for (int e = 0; e < E; e += 4)
            *reinterpret_cast<float4_t*>(&wr[dx][e]) = *reinterpret_cast<const float4_t*>(w + (size_t)(dy * 5 + dx) * C + c0 + e);
#pragma unroll
        for (int cx = 0; cx < XS + 4; ++cx) {
          float f[E];
          unpack16<T>(raw[cx], f);
          if (!interior) {
            const int xi = xs - 2 + cx;
            const float m = (xi >= 0 && xi < W) ? my : 0.f;
#pragma unroll
            for (int e = 0; e < E; ++e) f[e] *= m;
          }
#pragma unroll
          for (int dx = 0; dx < 5; ++dx) {
            const int o = cx - dx;
            if (o >= 0 && o < XS) {
#pragma unroll
              for (int e = 0; e < E; ++e) acc[oy][o][e] = __builtin_fmaf(wr[dx][e], f[e], acc[oy][o][e]);
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
#pragma unroll
    for (int oy = 0; oy < YS; ++oy) {
      if (ys + oy >= H) break;
      T* yrow = y + (((size_t)b * H + ys + oy) * W) * y_pitch + c0;
#pragma unroll
      for (int o = 0; o < XS; ++o) {
        if (xs + o >= W) break;
        float v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = fmaxf(__builtin_fmaf(acc[oy][o][e], sc[e], sh[e]), 0.f);
        *reinterpret_cast<u32x4*>(yrow + (size_t)(xs + o) * y_pitch) = pack16<T>(v);
      }
    }
  }
}

template <typename T, int XS, int YS, int MINW = 1>
void launch_dw(const void* x, const float* w, const float* scale, const float* shift, void* y, int B, int C, int H, int W,
               int x_pitch, int y_pitch, hipStream_t s) {
  constexpr int E = ElemTraits<T>::kPer16B;
  const size_t total = (size_t)B * ((H + YS - 1) / YS) * ((W + XS - 1) / XS) * (C / E);
  size_t g = (total + 255) / 256;
  if (g > 32768) g = 32768;
  hipLaunchKernelGGL((dwconv5x5_kernel<T, XS, YS, MINW>), dim3((int)g), dim3(256), 0, s, (const T*)x, w, scale, shift, (T*)y, B, C, H, W, x_pitch, y_pitch);
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
  hipStream_t s = static_cast<hipStream_t>(stream);
  // tile shape: ROMA_DW_TILE=XSxYS overrides (tuning aid); default 8x1
  static int tile = -1;
  if (tile < 0) {
    const char* ev = getenv("ROMA_DW_TILE");
    tile = ev ? atoi(ev) : 81;
  }
#define ROMA_DW(T)                                                                                                    \
  switch (tile) {                                                                                                     \
    case 41: launch_dw<T, 4, 1>(x, w, scale, shift, y, B, C, H, W, x_pitch, y_pitch, s); break;                       \
    case 42: launch_dw<T, 4, 2>(x, w, scale, shift, y, B, C, H, W, x_pitch, y_pitch, s); break;                       \
    case 82: launch_dw<T, 8, 2>(x, w, scale, shift, y, B, C, H, W, x_pitch, y_pitch, s); break;                       \
    case 44: launch_dw<T, 4, 4>(x, w, scale, shift, y, B, C, H, W, x_pitch, y_pitch, s); break;                       \
    default: launch_dw<T, 8, 1>(x, w, scale, shift, y, B, C, H, W, x_pitch, y_pitch, s); break;                       \
  }
  if (dtype == ROMA_F32) { ROMA_DW(float) } else if (dtype == ROMA_F16) { ROMA_DW(half_t) } else { ROMA_DW(bf16_t) }
#undef ROMA_DW
  ROMA_CHECK_LAUNCH();
}
