// In-place bias + ReLU on planar (NCHW) activations — the epilogue of every VGG19-BN layer once BatchNorm is folded into
// the convolution (reference: romatch/models/encoders.py:68-78 runs conv -> BN -> ReLU; the folded bias add and the ReLU
// are two separate full passes in the library path, and MIOpen's fused conv+bias+activation plans measured 20-37 % slower
// than its plain 3x3 solvers on gfx950).  HBM-bound: one read + one write of the map.
#include "common.h"

namespace roma {
namespace {

template <typename T, int VEC>
__global__ __launch_bounds__(256) void bias_relu_kernel(T* __restrict__ x, const T* __restrict__ bias, int C, int HW) {
  const int plane = blockIdx.y;                               // b * C + c
  const float bv = to_f32(bias[plane % C]);
  T* p = x + (size_t)plane * HW;
  const int nvec = HW / VEC;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nvec; i += gridDim.x * 256) {
    if constexpr (VEC == 8) {
      u32x4 v = reinterpret_cast<u32x4*>(p)[i];
      float f[8];
      unpack16<T>(v, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e] + bv, 0.f);
      reinterpret_cast<u32x4*>(p)[i] = pack16<T>(f);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) p[i * VEC + e] = from_f32<T>(fmaxf(to_f32(p[i * VEC + e]) + bv, 0.f));
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void bias_relu_kernel_f32(float* __restrict__ x, const float* __restrict__ bias, int C, int HW) {
  const int plane = blockIdx.y;
  const float bv = bias[plane % C];
  float* p = x + (size_t)plane * HW;
  const int nvec = HW / VEC;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nvec; i += gridDim.x * 256) {
    if constexpr (VEC == 4) {
      float4_t v = reinterpret_cast<float4_t*>(p)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] + bv, 0.f);
      reinterpret_cast<float4_t*>(p)[i] = v;
    } else {
      p[i] = fmaxf(p[i] + bv, 0.f);
    }
  }
}

// bias + ReLU in place AND the 2x2 / stride-2 max-pool of the result in the same pass — the VGG19-BN layers that are followed
// by a pool (encoders.py:68-78: the feature is captured BEFORE each pool, so the full-resolution map is still written; what the
// fusion removes is the pool kernel's second read of it: 191 MB at 864 x 864 x 64).  One thread = two rows x VEC columns.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void bias_relu_pool_kernel(T* __restrict__ x, const T* __restrict__ bias, T* __restrict__ pooled,
                                                            int C, int H, int W) {
  const int plane = blockIdx.y;                               // b * C + c
  const float bv = to_f32(bias[plane % C]);
  T* p = x + (size_t)plane * H * W;
  T* q = pooled + (size_t)plane * (H / 2) * (W / 2);
  const int wv = W / VEC, n = (H / 2) * wv;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int yo = i / wv, xv = i - yo * wv;
    T* r0 = p + (size_t)(2 * yo) * W + xv * VEC;
    T* r1 = r0 + W;
    float a[VEC], b[VEC];
    if constexpr (VEC == 8 && sizeof(T) == 2) {
      unpack16<T>(*reinterpret_cast<const u32x4*>(r0), a);
      unpack16<T>(*reinterpret_cast<const u32x4*>(r1), b);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) { a[e] = to_f32(r0[e]); b[e] = to_f32(r1[e]); }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) { a[e] = fmaxf(a[e] + bv, 0.f); b[e] = fmaxf(b[e] + bv, 0.f); }
    if constexpr (VEC == 8 && sizeof(T) == 2) {
      *reinterpret_cast<u32x4*>(r0) = pack16<T>(a);
      *reinterpret_cast<u32x4*>(r1) = pack16<T>(b);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) { r0[e] = from_f32<T>(a[e]); r1[e] = from_f32<T>(b[e]); }
    }
    T* o = q + (size_t)yo * (W / 2) + xv * (VEC / 2);
#pragma unroll
    for (int e = 0; e < VEC / 2; ++e)   // max of values already rounded to T would equal the rounded max: rounding is monotone
      o[e] = from_f32<T>(fmaxf(fmaxf(a[2 * e], a[2 * e + 1]), fmaxf(b[2 * e], b[2 * e + 1])));
  }
}

// ---- channels-last (NHWC) variants: the layout the VGG19 stack runs in since round 3 (with MIOpen's solver search on, the CK
// NHWC implicit-GEMM kernels are the fastest 3x3 solvers on gfx950 — and on planar activations MIOpen wraps them in layout
// transposes: 1.3 ms of `batched_transpose` per match).  One thread = one 16-byte channel packet of one pixel.
template <typename T>
__global__ __launch_bounds__(256) void bias_relu_nhwc_kernel(T* __restrict__ x, const T* __restrict__ bias, long npack, int cpack) {
  constexpr int E = ElemTraits<T>::kPer16B;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npack; i += (long)gridDim.x * 256) {
    float f[E], bb[E];
    unpack16<T>(reinterpret_cast<const u32x4*>(x)[i], f);
    unpack16<T>(reinterpret_cast<const u32x4*>(bias)[i % cpack], bb);
#pragma unroll
    for (int e = 0; e < E; ++e) f[e] = fmaxf(f[e] + bb[e], 0.f);
    reinterpret_cast<u32x4*>(x)[i] = pack16<T>(f);
  }
}

// bias + ReLU in place and the 2x2 / stride-2 max of the result: one thread = one channel packet of one POOLED pixel
template <typename T>
__global__ __launch_bounds__(256) void bias_relu_pool_nhwc_kernel(T* __restrict__ x, const T* __restrict__ bias, T* __restrict__ pooled,
                                                                 int B, int H, int W, int cpack) {
  constexpr int E = ElemTraits<T>::kPer16B;
  const int Ho = H / 2, Wo = W / 2;
  const long n = (long)B * Ho * Wo * cpack;
  u32x4* xv = reinterpret_cast<u32x4*>(x);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int cp = (int)(i % cpack);
    long r = i / cpack;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho), b = (int)(r / Ho);
    float bb[E], m[E];
    unpack16<T>(reinterpret_cast<const u32x4*>(bias)[cp], bb);
#pragma unroll
    for (int e = 0; e < E; ++e) m[e] = 0.f;                     // ReLU output is >= 0
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const long idx = (((long)b * H + 2 * yo + dy) * W + 2 * xo + dx) * cpack + cp;
        float f[E];
        unpack16<T>(xv[idx], f);
#pragma unroll
        for (int e = 0; e < E; ++e) { f[e] = fmaxf(f[e] + bb[e], 0.f); }
        const u32x4 pk = pack16<T>(f);
        xv[idx] = pk;
        unpack16<T>(pk, f);                                       // the max of the ROUNDED values, as a separate pool pass would see them
#pragma unroll
        for (int e = 0; e < E; ++e) m[e] = fmaxf(m[e], f[e]);
      }
    reinterpret_cast<u32x4*>(pooled)[i] = pack16<T>(m);
  }
}

}  // namespace
}  // namespace roma

using namespace roma;

extern "C" int roma_bias_relu_nhwc(void* x, const void* bias, long npix, int C, int dtype, void* stream) {
  ROMA_REQUIRE(x && bias, ROMA_E_ARG, "roma_bias_relu_nhwc: null pointer");
  const int e16 = dtype == ROMA_F32 ? 4 : 8;
  ROMA_REQUIRE(npix > 0 && C > 0 && C % e16 == 0 && aligned16(x) && aligned16(bias), ROMA_E_ALIGN,
               "roma_bias_relu_nhwc: C = %d must be a multiple of %d and the buffers 16-byte aligned", C, e16);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long npack = npix * (C / e16);
  const unsigned grid = (unsigned)((npack + 255) / 256 < 65536 ? (npack + 255) / 256 : 65536);
  switch (dtype) {
    case ROMA_F32: hipLaunchKernelGGL((bias_relu_nhwc_kernel<float>), dim3(grid), dim3(256), 0, s, (float*)x, (const float*)bias, npack, C / e16); break;
    case ROMA_F16: hipLaunchKernelGGL((bias_relu_nhwc_kernel<half_t>), dim3(grid), dim3(256), 0, s, (half_t*)x, (const half_t*)bias, npack, C / e16); break;
    case ROMA_BF16: hipLaunchKernelGGL((bias_relu_nhwc_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, (bf16_t*)x, (const bf16_t*)bias, npack, C / e16); break;
    default: ROMA_REQUIRE(false, ROMA_E_DTYPE, "roma_bias_relu_nhwc: dtype %d", dtype);
  }
  ROMA_CHECK_LAUNCH();
}

extern "C" int roma_bias_relu_pool2_nhwc(void* x, const void* bias, void* pooled, int B, int C, int H, int W, int dtype, void* stream) {
  ROMA_REQUIRE(x && bias && pooled, ROMA_E_ARG, "roma_bias_relu_pool2_nhwc: null pointer");
  const int e16 = dtype == ROMA_F32 ? 4 : 8;
  ROMA_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, ROMA_E_SHAPE, "roma_bias_relu_pool2_nhwc: bad shape B=%d H=%d W=%d (even H, W)", B, H, W);
  ROMA_REQUIRE(C > 0 && C % e16 == 0 && aligned16(x) && aligned16(bias) && aligned16(pooled), ROMA_E_ALIGN,
               "roma_bias_relu_pool2_nhwc: C = %d must be a multiple of %d and the buffers 16-byte aligned", C, e16);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long n = (long)B * (H / 2) * (W / 2) * (C / e16);
  const unsigned grid = (unsigned)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
  switch (dtype) {
    case ROMA_F32: hipLaunchKernelGGL((bias_relu_pool_nhwc_kernel<float>), dim3(grid), dim3(256), 0, s, (float*)x, (const float*)bias, (float*)pooled, B, H, W, C / e16); break;
    case ROMA_F16: hipLaunchKernelGGL((bias_relu_pool_nhwc_kernel<half_t>), dim3(grid), dim3(256), 0, s, (half_t*)x, (const half_t*)bias, (half_t*)pooled, B, H, W, C / e16); break;
    case ROMA_BF16: hipLaunchKernelGGL((bias_relu_pool_nhwc_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, (bf16_t*)x, (const bf16_t*)bias, (bf16_t*)pooled, B, H, W, C / e16); break;
    default: ROMA_REQUIRE(false, ROMA_E_DTYPE, "roma_bias_relu_pool2_nhwc: dtype %d", dtype);
  }
  ROMA_CHECK_LAUNCH();
}

extern "C" int roma_bias_relu_pool2_nchw(void* x, const void* bias, void* pooled, int B, int C, int H, int W, int dtype, void* stream) {
  ROMA_REQUIRE(x && bias && pooled, ROMA_E_ARG, "roma_bias_relu_pool2_nchw: null pointer");
  ROMA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && (long)B * C <= 65535 && H % 2 == 0 && W % 2 == 0, ROMA_E_SHAPE,
               "roma_bias_relu_pool2_nchw: bad shape B=%d C=%d H=%d W=%d (even H, W; B*C <= 65535)", B, C, H, W);
  ROMA_REQUIRE(dtype >= ROMA_F32 && dtype <= ROMA_BF16, ROMA_E_DTYPE, "roma_bias_relu_pool2_nchw: dtype %d", dtype);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int planes = B * C;
  auto gridx = [&](int vec) { const int n = ((H / 2) * (W / vec) + 255) / 256; return n < 64 ? n : 64; };
  const bool v8 = dtype != ROMA_F32 && W % 8 == 0 && aligned16(x);
#define ROMA_BRP(T, V) hipLaunchKernelGGL((bias_relu_pool_kernel<T, V>), dim3(gridx(V), planes), dim3(256), 0, s, (T*)x, (const T*)bias, (T*)pooled, C, H, W)
  if (dtype == ROMA_F32) ROMA_BRP(float, 2);
  else if (dtype == ROMA_F16) { if (v8) ROMA_BRP(half_t, 8); else ROMA_BRP(half_t, 2); }
  else { if (v8) ROMA_BRP(bf16_t, 8); else ROMA_BRP(bf16_t, 2); }
#undef ROMA_BRP
  ROMA_CHECK_LAUNCH();
}

extern "C" int roma_bias_relu_nchw(void* x, const void* bias, int B, int C, int HW, int dtype, void* stream) {
  ROMA_REQUIRE(x && bias, ROMA_E_ARG, "roma_bias_relu_nchw: null pointer");
  ROMA_REQUIRE(B > 0 && C > 0 && HW > 0 && (long)B * C <= 65535, ROMA_E_SHAPE, "roma_bias_relu_nchw: bad shape B=%d C=%d HW=%d", B, C, HW);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int planes = B * C;
  auto gridx = [&](int vec) { const int n = (HW / vec + 255) / 256; return n < 64 ? n : 64; };
  if (dtype == ROMA_F32) {
    if (HW % 4 == 0 && aligned16(x)) hipLaunchKernelGGL((bias_relu_kernel_f32<4>), dim3(gridx(4), planes), dim3(256), 0, s, (float*)x, (const float*)bias, C, HW);
    else hipLaunchKernelGGL((bias_relu_kernel_f32<1>), dim3(gridx(1), planes), dim3(256), 0, s, (float*)x, (const float*)bias, C, HW);
  } else if (dtype == ROMA_F16 || dtype == ROMA_BF16) {
    const bool v8 = HW % 8 == 0 && aligned16(x);
    if (dtype == ROMA_F16) {
      if (v8) hipLaunchKernelGGL((bias_relu_kernel<half_t, 8>), dim3(gridx(8), planes), dim3(256), 0, s, (half_t*)x, (const half_t*)bias, C, HW);
      else hipLaunchKernelGGL((bias_relu_kernel<half_t, 1>), dim3(gridx(1), planes), dim3(256), 0, s, (half_t*)x, (const half_t*)bias, C, HW);
    } else {
      if (v8) hipLaunchKernelGGL((bias_relu_kernel<bf16_t, 8>), dim3(gridx(8), planes), dim3(256), 0, s, (bf16_t*)x, (const bf16_t*)bias, C, HW);
      else hipLaunchKernelGGL((bias_relu_kernel<bf16_t, 1>), dim3(gridx(1), planes), dim3(256), 0, s, (bf16_t*)x, (const bf16_t*)bias, C, HW);
    }
  } else {
    ROMA_REQUIRE(false, ROMA_E_DTYPE, "roma_bias_relu_nchw: dtype %d", dtype);
  }
  ROMA_CHECK_LAUNCH();
}
