// Device-side image pre-processing — reference: romatch/utils/utils.py:165-261 (get_tuple_transform_ops: TupleResize =
// PIL bicubic resize, ToTensorScaled = /255, TupleNormalize = ImageNet mean/std), called from matcher.py:606-637, 667-676.
// The reference resizes every image twice on the host with PIL.  PIL's 8-bit resampling is integer arithmetic
// (Pillow src/libImaging/Resample.c, ImagingResampleHorizontal_8bpc / Vertical_8bpc): per output sample
//     ss = 1 << 21;  ss += in[xmin + k] * coef[k]  (k < xmax);  out = clip8(ss >> 22)
// with 22-bit fixed-point coefficients, a horizontal pass to 8 bits followed by a vertical pass.  The tables (bounds,
// coefficients) are tiny and computed on the host in float64 exactly like precompute_coeffs / normalize_coeffs_8bpc
// (roma_amd/preproc.py); these kernels do the two integer passes and the final scale + normalise, so the result is
// BIT-IDENTICAL to the host path while the photograph crosses PCIe once, as uint8.
#include "common.h"

namespace roma {
namespace {

__device__ __forceinline__ unsigned char clip8(int ss) {
  const int v = ss >> 22;                                       // arithmetic shift, like the C code's table lookup index
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// axis = 1: horizontal (out (H, OUT, C) from in (H, W, C));  axis = 0: vertical (out (OUT, W, C) from in (H, W, C))
__global__ __launch_bounds__(256) void resample_u8_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out,
                                                          int H, int W, int C, int OUT, int axis, const int* __restrict__ bounds,
                                                          const int* __restrict__ coef, int ksize) {
  const long total = axis ? (long)H * OUT * C : (long)OUT * W * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long r = i / C;
    int ss = 1 << 21;
    if (axis) {
      const int xx = (int)(r % OUT), yy = (int)(r / OUT);
      const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
      const int* k = coef + (size_t)xx * ksize;
      const unsigned char* row = in + ((size_t)yy * W + xmin) * C + c;
      for (int x = 0; x < xmax; ++x) ss += (int)row[(size_t)x * C] * k[x];
    } else {
      const int xx = (int)(r % W), yy = (int)(r / W);
      const int ymin = bounds[2 * yy], ymax = bounds[2 * yy + 1];
      const int* k = coef + (size_t)yy * ksize;
      const unsigned char* col = in + ((size_t)ymin * W + xx) * C + c;
      for (int y = 0; y < ymax; ++y) ss += (int)col[(size_t)y * W * C] * k[y];
    }
    out[i] = clip8(ss);
  }
}

// uint8 (H, W, 3) -> float32 (3, H, W): ((v / 255) - mean[c]) / std[c], the reference's operation order (two IEEE divisions)
__global__ __launch_bounds__(256) void normalize_u8_kernel(const unsigned char* __restrict__ in, float* __restrict__ out, int HW,
                                                           float m0, float m1, float m2, float s0, float s1, float s2) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
    const float r = (float)in[3 * i] / 255.0f, g = (float)in[3 * i + 1] / 255.0f, b = (float)in[3 * i + 2] / 255.0f;
    out[i] = (r - m0) / s0;
    out[HW + i] = (g - m1) / s1;
    out[2 * HW + i] = (b - m2) / s2;
  }
}

}  // namespace
}  // namespace roma

using namespace roma;

extern "C" int roma_resample_u8(const void* in, void* out, int H, int W, int C, int out_size, int axis, const int* bounds,
                                const int* coef, int ksize, void* stream) {
  ROMA_REQUIRE(in && out && bounds && coef, ROMA_E_ARG, "roma_resample_u8: null pointer");
  ROMA_REQUIRE(H > 0 && W > 0 && C > 0 && out_size > 0 && ksize > 0 && (axis == 0 || axis == 1), ROMA_E_SHAPE, "roma_resample_u8: bad shape");
  const long total = axis ? (long)H * out_size * C : (long)out_size * W * C;
  long g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(resample_u8_kernel, dim3((int)g), dim3(256), 0, static_cast<hipStream_t>(stream), (const unsigned char*)in,
                     (unsigned char*)out, H, W, C, out_size, axis, bounds, coef, ksize);
  ROMA_CHECK_LAUNCH();
}

extern "C" int roma_normalize_u8(const void* in, float* out, int H, int W, const float* mean3, const float* std3, void* stream) {
  ROMA_REQUIRE(in && out && mean3 && std3, ROMA_E_ARG, "roma_normalize_u8: null pointer");
  ROMA_REQUIRE(H > 0 && W > 0, ROMA_E_SHAPE, "roma_normalize_u8: bad shape");
  const int HW = H * W;
  int g = (HW + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(normalize_u8_kernel, dim3(g), dim3(256), 0, static_cast<hipStream_t>(stream), (const unsigned char*)in, out, HW,
                     mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  ROMA_CHECK_LAUNCH();
}
