// Interface between local_corr.hip (dispatch) and the tile kernels for 16-bit channels-last inputs with r <= 3
// (local_corr_t8.hip: 8x8 tiles, 32-channel chunks; local_corr_rows.hip: box rows streamed whole, f0 in registers).
#pragma once
#include <hip/hip_runtime.h>

namespace roma {

struct LCTileParams {
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
  float inv_tiles_per_img, inv_tiles_x;   // 1 / (tiles_x tiles_y bands), 1 / tiles_x (row-streaming kernel: block id -> tile without integer division)
  int bands;                              // row-streaming kernel: workgroups per tile (each streams a band of the tile's box rows); set by its launcher
  float inv_bands;
};

// one 8x8 tile per workgroup (local_corr_t8.hip): 16-bit channels-last inputs, r in 1..3, C a multiple of 32
int local_corr_t8(const LCTileParams& p, int r, int dtype, hipStream_t stream);
// row-streaming kernel (local_corr_rows.hip), 8x8 tiles: 16-bit channels-last inputs, C = 256 or 512, r in 1..3, or r = 7 on maps up to
// 47 pixels wide (no patch path there: every tile streams)
bool local_corr_rows_supports(int C, int r, int H, int W);
int local_corr_rows(const LCTileParams& p, int r, int dtype, hipStream_t stream);

}  // namespace roma
