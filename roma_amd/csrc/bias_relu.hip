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

}  // namespace
}  // namespace roma

using namespace roma;

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
