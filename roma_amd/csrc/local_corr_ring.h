// Interface between local_corr.hip (dispatch) and local_corr_ring.hip (persistent loader/consumer kernel).
#pragma once
#include <hip/hip_runtime.h>

namespace roma {

struct LCRingParams {
  const void* f0;
  const void* f1;
  const float* flow;
  void* out;
  int B, C, H, W;
  int f0_pitch, f1_pitch, out_pitch;
  int out_nhwc;
  int tiles_x, tiles_y;
  int f1_shift;
  float scale;
  int dbg;       // ablation switches (ROMA_LC_DBG, diagnostics only): 1 = no MFMA/epilogue, 2 = no DMA (stale LDS is read), 4 = no epilogue
};

// 16-bit channels-last inputs, r in 1..3, C a multiple of 32 and >= kRingMinC
constexpr int kRingMinC = 224;
int local_corr_ring(const LCRingParams& p, int r, int dtype, hipStream_t stream);
// one 8x8 tile per workgroup (local_corr_t8.hip): 16-bit channels-last inputs, r in 1..3, C a multiple of 32
int local_corr_t8(const LCRingParams& p, int r, int dtype, hipStream_t stream);

}  // namespace roma
