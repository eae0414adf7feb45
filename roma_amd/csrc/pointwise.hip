// 1x1 convolution for NARROW channels-last activations (C <= 32) — the back half of a ConvRefiner block
// (romatch/models/matcher.py:102, Conv2d(D, D, 1)) at the two finest scales, where D = 24 and M = h*w*B is 0.6-1.5 M rows.
// hipBLASLt runs this skinny GEMM (N = K = 24) at 0.65 TB/s (212 us at 864x864); it is a pure streaming op:
//   y[m][n] = b[n] + sum_k x[m][k] * wt[k][n]
// One thread per pixel row: the row's packets are loaded once (16-byte loads), the K x N fp32 weights are wave-uniform
// (scalar loads through the constant cache), fp32 accumulate, 16-byte stores.
#include "common.h"

namespace roma {
namespace {

template <typename T, int NP>   // NP = number of 16-byte packets per row (C = NP * elements-per-packet)
__global__ __launch_bounds__(256) void pointwise_small_kernel(const T* __restrict__ x, const float* __restrict__ wt,
                                                             const float* __restrict__ bias, T* __restrict__ y, size_t M,
                                                             int x_pitch, int y_pitch) {
  constexpr int E = ElemTraits<T>::kPer16B;
  constexpr int C = NP * E;
  for (size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (size_t)gridDim.x * blockDim.x) {
    float xin[C];
#pragma unroll
    for (int pk = 0; pk < NP; ++pk) unpack16<T>(*reinterpret_cast<const u32x4*>(x + m * x_pitch + pk * E), xin + pk * E);
    float acc[C];
#pragma unroll
    for (int n = 0; n < C; ++n) acc[n] = bias[n];
#pragma unroll
    for (int k = 0; k < C; ++k)
#pragma unroll
      for (int n = 0; n < C; ++n) acc[n] = __builtin_fmaf(xin[k], wt[k * C + n], acc[n]);
#pragma unroll
    for (int pk = 0; pk < NP; ++pk) *reinterpret_cast<u32x4*>(y + m * y_pitch + pk * E) = pack16<T>(acc + pk * E);
  }
}

}  // namespace
}  // namespace roma

using namespace roma;

extern "C" int roma_pointwise_small(const void* x, const float* wt, const float* bias, void* y, long M, int C, int dtype, int x_pitch,
                                    int y_pitch, void* stream) {
  ROMA_REQUIRE(x && wt && bias && y, ROMA_E_ARG, "roma_pointwise_small: null pointer");
  ROMA_REQUIRE(M > 0 && C > 0 && x_pitch >= C && y_pitch >= C, ROMA_E_SHAPE, "roma_pointwise_small: bad shape");
  ROMA_REQUIRE(dtype >= ROMA_F32 && dtype <= ROMA_BF16, ROMA_E_DTYPE, "roma_pointwise_small: unknown dtype %d", dtype);
  const int e = dtype == ROMA_F32 ? 4 : 8;
  ROMA_REQUIRE(C % e == 0 && C <= 32 && x_pitch % e == 0 && y_pitch % e == 0 && aligned16(x) && aligned16(y), ROMA_E_ALIGN,
               "roma_pointwise_small: C must be a multiple of %d and <= 32, pitches multiples of %d, bases 16-byte aligned", e, e);
  hipStream_t s = static_cast<hipStream_t>(stream);
  size_t g = ((size_t)M + 255) / 256;
  if (g > 16384) g = 16384;
  const int np = C / e;
#define ROMA_PW(T, NP)                                                                                                        \
  hipLaunchKernelGGL((pointwise_small_kernel<T, NP>), dim3((int)g), dim3(256), 0, s, (const T*)x, wt, bias, (T*)y, (size_t)M, \
                     x_pitch, y_pitch)
#define ROMA_PW_T(T)                                                              \
  switch (np) {                                                                   \
    case 1: ROMA_PW(T, 1); break;                                                 \
    case 2: ROMA_PW(T, 2); break;                                                 \
    case 3: ROMA_PW(T, 3); break;                                                 \
    case 4: ROMA_PW(T, 4); break;                                                 \
    case 5: if constexpr (sizeof(T) == 4) { ROMA_PW(T, 5); } break;               \
    case 6: if constexpr (sizeof(T) == 4) { ROMA_PW(T, 6); } break;               \
    case 7: if constexpr (sizeof(T) == 4) { ROMA_PW(T, 7); } break;               \
    case 8: if constexpr (sizeof(T) == 4) { ROMA_PW(T, 8); } break;               \
  }
  if (dtype == ROMA_F32) { ROMA_PW_T(float) } else if (dtype == ROMA_F16) { ROMA_PW_T(half_t) } else { ROMA_PW_T(bf16_t) }
#undef ROMA_PW_T
#undef ROMA_PW
  ROMA_CHECK_LAUNCH();
}
