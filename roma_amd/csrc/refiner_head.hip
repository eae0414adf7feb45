// ConvRefiner head fused with the Decoder's update step — reference: romatch/models/matcher.py:141 (out_conv on d.float(),
// a 1x1 convolution D -> 3 in fp32) and :397-402 (flow += ins * (dx/(4W), dy/(4H)); certainty += dcert).
//   d[m][0..2] = bo + sum_k x[m][k] * wo[k][0..2]           x: (M, pitch) channels-last rows of the last block, any dtype
//   flow[b,0,p] += sx * d0;  flow[b,1,p] += sy * d1;  cert_out[b,p] = (cert_in ? cert_in[b,p] : 0) + d2
// One pass over the activation instead of: fp32 copy, a 3-column GEMM, a planar transpose and five elementwise kernels.
// L lanes share a pixel (L = 64 / 16 / 4 by channel count), each striding 16-byte packets, shuffle-reduced.
#include "common.h"

namespace roma {
namespace {

template <typename T, int L>
__global__ __launch_bounds__(256) void refiner_head_kernel(const T* __restrict__ x, const float* __restrict__ wo,
                                                          const float* __restrict__ bo, float* __restrict__ flow,
                                                          const float* __restrict__ cert_in, float* __restrict__ cert_out,
                                                          float* __restrict__ delta_out, int B, int HW, int C, int pitch, float sx, float sy) {
  constexpr int E = ElemTraits<T>::kPer16B;
  const int PK = C / E;
  const size_t M = (size_t)B * HW;
  const int sub = threadIdx.x % L;
  for (size_t m = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / L; m < M; m += (size_t)gridDim.x * blockDim.x / L) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int pk = sub; pk < PK; pk += L) {
      float f[E];
      unpack16<T>(*reinterpret_cast<const u32x4*>(x + m * pitch + pk * E), f);
      const float* w = wo + (size_t)pk * E * 3;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        a0 = __builtin_fmaf(f[e], w[e * 3 + 0], a0);
        a1 = __builtin_fmaf(f[e], w[e * 3 + 1], a1);
        a2 = __builtin_fmaf(f[e], w[e * 3 + 2], a2);
      }
    }
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) {
      a0 += __shfl_xor(a0, o, 64);
      a1 += __shfl_xor(a1, o, 64);
      a2 += __shfl_xor(a2, o, 64);
    }
    if (sub == 0) {
      a0 += bo[0]; a1 += bo[1]; a2 += bo[2];
      const int b = (int)(m / HW);
      const size_t p = m - (size_t)b * HW;
      float* fl = flow + (size_t)b * 2 * HW + p;
      fl[0] += sx * a0;
      fl[HW] += sy * a1;
      cert_out[m] = (cert_in ? cert_in[m] : 0.f) + a2;
      if (delta_out) {
        delta_out[(size_t)b * 3 * HW + p] = a0;
        delta_out[(size_t)b * 3 * HW + HW + p] = a1;
        delta_out[(size_t)b * 3 * HW + 2 * (size_t)HW + p] = a2;
      }
    }
  }
}

}  // namespace
}  // namespace roma

using namespace roma;

extern "C" int roma_refiner_head(const void* x, const float* wo, const float* bo, float* flow, const float* cert_in, float* cert_out,
                                 float* delta_out, int B, int H, int W, int C, int pitch, int dtype, float sx, float sy, void* stream) {
  ROMA_REQUIRE(x && wo && bo && flow && cert_out, ROMA_E_ARG, "roma_refiner_head: null pointer");
  ROMA_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && pitch >= C, ROMA_E_SHAPE, "roma_refiner_head: bad shape");
  ROMA_REQUIRE(dtype >= ROMA_F32 && dtype <= ROMA_BF16, ROMA_E_DTYPE, "roma_refiner_head: unknown dtype %d", dtype);
  const int e = dtype == ROMA_F32 ? 4 : 8;
  ROMA_REQUIRE(C % e == 0 && pitch % e == 0 && aligned16(x), ROMA_E_ALIGN, "roma_refiner_head: C and pitch must be multiples of %d, x 16-byte aligned", e);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int pk = C / e;
  const int L = pk >= 64 ? 64 : (pk >= 16 ? 16 : 4);
  const size_t threads = (size_t)B * H * W * L;
  size_t g = (threads + 255) / 256;
  if (g > 32768) g = 32768;
#define ROMA_RH(T, LL) \
  hipLaunchKernelGGL((refiner_head_kernel<T, LL>), dim3((int)g), dim3(256), 0, s, (const T*)x, wo, bo, flow, cert_in, cert_out, delta_out, B, H * W, C, pitch, sx, sy)
#define ROMA_RH_T(T) \
  if (L == 64) { ROMA_RH(T, 64); } else if (L == 16) { ROMA_RH(T, 16); } else { ROMA_RH(T, 4); }
  if (dtype == ROMA_F32) { ROMA_RH_T(float) } else if (dtype == ROMA_F16) { ROMA_RH_T(half_t) } else { ROMA_RH_T(bf16_t) }
#undef ROMA_RH_T
#undef ROMA_RH
  ROMA_CHECK_LAUNCH();
}
