// Micro-benchmark of the L2 -> LDS staging path that bounds roma_local_corr (profiles/r02_local_corr_pmc.md): the same
// 16x16-pixel box of a channels-last fp16 map (C = 256: 512-byte pixel rows) per 8x8-pixel tile, staged by LDS-DMA
//   mode 0: 32-channel chunks, 64-byte pieces per pixel        (what local_corr_t8 does), wait-all + barrier per chunk
//   mode 1: 64-channel chunks, 128-byte pieces
//   mode 2: one box ROW per stage, whole 512-byte pixel rows     (the row-streaming design), wait-all + barrier per row
//   mode 3: mode 2 with a 4-slot ring, two rows in flight behind a counted vmcnt
//   mode 4: mode 0 with two chunk buffers, the next chunk in flight
//   mode 5: mode 2's addresses by global_load_dwordx4 into registers (no LDS)
// Build: hipcc -O3 --offload-arch=gfx950 tools/stage_micro.hip -o tools/scratch/stage_micro ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_wave_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_wave_base) : "memory");
}
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
#define VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

struct P { const unsigned char* f1; uint32_t* sink; int H, W, tiles_x, tiles_y, pad_lds; };

template <int MODE>
__global__ __launch_bounds__(256) void k(P p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tpi = p.tiles_x * p.tiles_y;
  const int wid = xcd_remap(blockIdx.x, gridDim.x);
  const int b = wid / tpi, t = wid - b * tpi;
  const int ty0 = (t / p.tiles_x) * 8, tx0 = (t % p.tiles_x) * 8;
  const int bx0 = min(max(tx0 - 4, 0), p.W - 16), by0 = min(max(ty0 - 4, 0), p.H - 16);
  const unsigned char* base = p.f1 + (size_t)b * p.H * p.W * 512;
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  u32x4* rows = reinterpret_cast<u32x4*>(smem);
  u32x4 acc = {0, 0, 0, 0};
  if (MODE == 0 || MODE == 4) {
    const unsigned char* src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int Pp = (i * 4 + wave) * 64 + lane, px = Pp >> 2, pc = Pp & 3;
      src[i] = base + ((size_t)(by0 + (px >> 4)) * p.W + bx0 + (px & 15)) * 512 + pc * 16;
    }
    if (MODE == 0) {
      for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(src[i] + c * 64, lds0 + (i * 4 + wave) * 1024);
        VM(0); BAR();
#pragma unroll
        for (int i = 0; i < 4; ++i) acc ^= rows[(i * 256 + tid + 17) & 1023];
        BAR();
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) dma16(src[i], lds0 + (i * 4 + wave) * 1024);
      for (int c = 0; c < 8; ++c) {
        const int nb = (c + 1) & 1;
        if (c + 1 < 8) {
#pragma unroll
          for (int i = 0; i < 4; ++i) dma16(src[i] + (c + 1) * 64, lds0 + nb * 16384 + (i * 4 + wave) * 1024);
          VM(4);
        } else VM(0);
        BAR();
#pragma unroll
        for (int i = 0; i < 4; ++i) acc ^= rows[(c & 1) * 1024 + ((i * 256 + tid + 17) & 1023)];
        BAR();
      }
    }
  } else if (MODE == 1) {
    const unsigned char* src[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int Pp = (i * 4 + wave) * 64 + lane, px = Pp >> 3, pc = Pp & 7;
      src[i] = base + ((size_t)(by0 + (px >> 4)) * p.W + bx0 + (px & 15)) * 512 + pc * 16;
    }
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int i = 0; i < 8; ++i) dma16(src[i] + c * 128, lds0 + (i * 4 + wave) * 1024);
      VM(0); BAR();
#pragma unroll
      for (int i = 0; i < 8; ++i) acc ^= rows[(i * 256 + tid + 17) & 2047];
      BAR();
    }
  } else if (MODE == 2 || MODE == 3 || MODE == 5) {
    uint32_t off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int Pp = (i * 4 + wave) * 64 + lane, px = Pp >> 5, pc = Pp & 31;
      off[i] = (uint32_t)((bx0 + px) * 512 + (pc ^ (px & 7)) * 16);
    }
    const unsigned char* rb = base + (size_t)by0 * p.W * 512;
    const size_t rstride = (size_t)p.W * 512;
    if (MODE == 2) {
      for (int j = 0; j < 16; ++j) {
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16(rb + j * rstride + off[i], lds0 + (i * 4 + wave) * 1024);
        VM(0); BAR();
#pragma unroll
        for (int i = 0; i < 2; ++i) acc ^= rows[(i * 256 + tid + 17) & 511];
        BAR();
      }
    } else if (MODE == 3) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16(rb + j * rstride + off[i], lds0 + j * 8192 + (i * 4 + wave) * 1024);
      for (int j = 0; j < 16; ++j) {
        if (j + 2 < 16) {
#pragma unroll
          for (int i = 0; i < 2; ++i) dma16(rb + (j + 2) * rstride + off[i], lds0 + ((j + 2) & 3) * 8192 + (i * 4 + wave) * 1024);
          VM(4);
        } else if (j + 1 < 16) VM(2);
        else VM(0);
        BAR();
#pragma unroll
        for (int i = 0; i < 2; ++i) acc ^= rows[(j & 3) * 512 + ((i * 256 + tid + 17) & 511)];
      }
    } else {
      for (int j = 0; j < 16; j += 4) {
        u32x4 v[8];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int i = 0; i < 2; ++i) v[jj * 2 + i] = *reinterpret_cast<const u32x4*>(rb + (j + jj) * rstride + off[i]);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc ^= v[q];
      }
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) p.sink[tid] = acc[0];
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, H = 216, W = 216, iters = 20;
  const size_t bytes = (size_t)B * H * W * 512;
  unsigned char* d; uint32_t* sink;
  hipMalloc(&d, bytes); hipMalloc(&sink, 4096);
  std::vector<uint32_t> h(bytes / 4);
  uint32_t s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
  hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
  P p{d, sink, H, W, W / 8, H / 8, 0};
  const int grid = B * p.tiles_x * p.tiles_y;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double staged = (double)grid * 256 * 512;
  const int lds_sizes[] = {32768, 40960, 53248, 80000};
  for (int mode = 0; mode < 6; ++mode)
    for (int li = 0; li < 4; ++li) {
      const int lds = lds_sizes[li];
      auto launch = [&]() {
        switch (mode) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), lds, 0, p); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), lds, 0, p); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), lds, 0, p); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), lds, 0, p); break;
          case 4: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), lds, 0, p); break;
          case 5: hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), lds, 0, p); break;
        }
      };
      hipFuncSetAttribute(reinterpret_cast<const void*>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
      hipFuncSetAttribute(reinterpret_cast<const void*>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
      hipFuncSetAttribute(reinterpret_cast<const void*>(k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
      hipFuncSetAttribute(reinterpret_cast<const void*>(k<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
      hipFuncSetAttribute(reinterpret_cast<const void*>(k<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
      hipFuncSetAttribute(reinterpret_cast<const void*>(k<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
      for (int i = 0; i < 3; ++i) launch();
      hipEventRecord(e0);
      for (int i = 0; i < iters; ++i) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1000.0 / iters;
      hipError_t err = hipGetLastError();
      printf("mode %d lds %6d (%d WG/CU): %8.1f us  staged %.2f TB/s  map %.2f TB/s %s\n", mode, lds, 163840 / lds, us, staged / us * 1e-6,
             (double)bytes / us * 1e-6, err == hipSuccess ? "" : hipGetErrorString(err));
    }
  return 0;
}
