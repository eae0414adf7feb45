// match() post-processing for gfx950 — reference: romatch/models/matcher.py:656-662 (certainty attenuation by the
// bilinearly up-sampled coarse scale-16 certainty), :684-718 (sigmoid, certainty = 0 where |flow| > 1, clamp to
// [-1,1], identity-grid concat, symmetric A->B | B->A side-by-side layout).  One thread per output pixel, one pass.
#include "common.h"

namespace roma {
namespace {

__global__ __launch_bounds__(256) void finalize_kernel(const float* __restrict__ flow, const float* __restrict__ cert,
                                                       const float* __restrict__ cert16, float* __restrict__ warp,
                                                       float* __restrict__ certainty, int P, int H, int W, int H16, int W16,
                                                       int symmetric) {
  const int WW = symmetric ? 2 * W : W;
  const size_t total = (size_t)P * H * WW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % WW);
    const int y = (int)((i / WW) % H);
    const int p = (int)(i / ((size_t)WW * H));
    const bool right = xx >= W;
    const int x = right ? xx - W : xx;
    const int b = right ? P + p : p;
    const size_t HW = (size_t)H * W;
    const float fx = flow[((size_t)b * 2 + 0) * HW + (size_t)y * W + x];
    const float fy = flow[((size_t)b * 2 + 1) * HW + (size_t)y * W + x];
    float c = cert[(size_t)b * HW + (size_t)y * W + x];
    if (cert16) {
      int y0, y1, x0, x1;
      float ly, lx;
      interp_src(y, H16, H, y0, y1, ly);
      interp_src(x, W16, W, x0, x1, lx);
      const float* q = cert16 + (size_t)b * H16 * W16;
      const float top = (1.f - lx) * q[y0 * W16 + x0] + lx * q[y0 * W16 + x1];
      const float bot = (1.f - lx) * q[y1 * W16 + x0] + lx * q[y1 * W16 + x1];
      const float low = (1.f - ly) * top + ly * bot;
      c -= 0.5f * low * (low < 0.f ? 1.f : 0.f);
    }
    float pr = 1.f / (1.f + expf(-c));
    if (fabsf(fx) > 1.f || fabsf(fy) > 1.f) pr = 0.f;
    const float cx = fminf(fmaxf(fx, -1.f), 1.f), cy = fminf(fmaxf(fy, -1.f), 1.f);
    const float gx = pix_center(x, W), gy = pix_center(y, H);
    float4_t o = right ? float4_t{cx, cy, gx, gy} : float4_t{gx, gy, cx, cy};
    *reinterpret_cast<float4_t*>(warp + i * 4) = o;
    certainty[i] = pr;
  }
}

}  // namespace
}  // namespace roma

using namespace roma;

extern "C" int roma_match_finalize(const float* flow, const float* cert, const float* cert16, float* warp, float* certainty, int P,
                                   int H, int W, int H16, int W16, int symmetric, void* stream) {
  ROMA_REQUIRE(flow && cert && warp && certainty, ROMA_E_ARG, "roma_match_finalize: null pointer");
  ROMA_REQUIRE(P > 0 && H > 0 && W > 0, ROMA_E_SHAPE, "roma_match_finalize: bad shape");
  ROMA_REQUIRE(!cert16 || (H16 > 0 && W16 > 0), ROMA_E_SHAPE, "roma_match_finalize: bad coarse certainty shape");
  ROMA_REQUIRE(aligned16(warp), ROMA_E_ALIGN, "roma_match_finalize: warp must be 16-byte aligned");
  const size_t total = (size_t)P * H * (symmetric ? 2 * W : W);
  size_t g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(finalize_kernel, dim3((int)g), dim3(256), 0, static_cast<hipStream_t>(stream), flow, cert, cert16, warp, certainty,
                     P, H, W, H16, W16, symmetric);
  ROMA_CHECK_LAUNCH();
}
