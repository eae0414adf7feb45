// Device helpers shared by the 8x8-tile local_correlation kernels (local_corr_t8.hip, local_corr_rows.hip).
#pragma once
#include "common.h"

namespace roma {
namespace lc {

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef __bf16 b8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float4_t mfma16r(const u32x4& a, const u32x4& b, float4_t c, half_t) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, a), __builtin_bit_cast(h8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ float4_t mfma16r(const u32x4& a, const u32x4& b, float4_t c, bf16_t) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8_t, a), __builtin_bit_cast(b8_t, b), c, 0, 0, 0);
}

// packet slot of (row, k-slice): rows are 64 bytes (4 packets), packet index XOR-swizzled by bit 2 of the row
__device__ __forceinline__ int swzr(int row, int k) { return row * 4 + (k ^ ((row >> 2) & 1)); }

// one LDS-DMA wave-instruction: lane i copies 16 bytes from ITS global address to lds_wave_base + 16 i.  Inline asm on
// purpose: hipcc does not count it, so every wait for it below is an explicit counted s_waitcnt (cdna_hip_programming.md §5.7).
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_wave_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_wave_base) : "memory");
}
// the same with the global address split into a wave-uniform base (SGPR pair) and a 32-bit per-lane byte offset: half the address
// registers, and a per-chunk advance of the base is scalar work
__device__ __forceinline__ void dma16_so(const void* sbase, uint32_t voff, uint32_t lds_wave_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_wave_base) : "memory");
}
__device__ __forceinline__ void dma4(const void* gsrc, uint32_t lds_wave_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_wave_base) : "memory");
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate): a binary tree of scalar branches
// (hipcc turned a 32-way switch into a chain of 31 compares).  A smaller count than needed only waits longer, so counts
// beyond 31 clamp down to 31.
#define ROMA_VM1(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define ROMA_VM2(n, A, B) do { if ((n) & 1) ROMA_VM1(B); else ROMA_VM1(A); } while (0)
__device__ __forceinline__ void wait_vmcnt(int n) {
  n = n > 31 ? 31 : n;
  if (n & 16) {
    if (n & 8) { if (n & 4) { if (n & 2) ROMA_VM2(n, 30, 31); else ROMA_VM2(n, 28, 29); } else { if (n & 2) ROMA_VM2(n, 26, 27); else ROMA_VM2(n, 24, 25); } }
    else       { if (n & 4) { if (n & 2) ROMA_VM2(n, 22, 23); else ROMA_VM2(n, 20, 21); } else { if (n & 2) ROMA_VM2(n, 18, 19); else ROMA_VM2(n, 16, 17); } }
  } else {
    if (n & 8) { if (n & 4) { if (n & 2) ROMA_VM2(n, 14, 15); else ROMA_VM2(n, 12, 13); } else { if (n & 2) ROMA_VM2(n, 10, 11); else ROMA_VM2(n, 8, 9); } }
    else       { if (n & 4) { if (n & 2) ROMA_VM2(n, 6, 7); else ROMA_VM2(n, 4, 5); } else { if (n & 2) ROMA_VM2(n, 2, 3); else ROMA_VM2(n, 0, 1); } }
  }
}

__device__ __forceinline__ void raw_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}


constexpr int kTP = 64, kTW = 8, kTH = 8, kNG = 4;

// group-major f0 row of tile pixel (py, px): group = 2*(py>>2) + (px>>2), inside the group row-major 4x4
__device__ __forceinline__ int frow(int py, int px) { return ((py >> 2) * 2 + (px >> 2)) * 16 + (py & 3) * 4 + (px & 3); }
__device__ __forceinline__ void fpix(int row, int& py, int& px) {
  const int g = row >> 4, m = row & 15;
  py = (g >> 1) * 4 + (m >> 2);
  px = (g & 1) * 4 + (m & 3);
}

__device__ __forceinline__ void flow_target(float fx, float fy, int H, int W, int& x0, int& y0, float& ax, float& ay) {
  float px = ((fx + 1.f) * W - 1.f) * 0.5f, py = ((fy + 1.f) * H - 1.f) * 0.5f;
  if (!(px > -1e6f && px < 1e6f)) px = -1e6f;                  // wild / non-finite flow: fully outside, integer conversion defined
  if (!(py > -1e6f && py < 1e6f)) py = -1e6f;
  const float fx0 = floorf(px), fy0 = floorf(py);
  ax = px - fx0;
  ay = py - fy0;
  x0 = (int)fx0;
  y0 = (int)fy0;
}

// lane i <- lane i+1 inside each row of 16 lanes (DPP row_shl:1; the last lane of a row reads 0)
__device__ __forceinline__ float right_neighbour(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));
}

}  // namespace lc
}  // namespace roma
