// Shared device/host helpers for libroma_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/roma_hip.h"

namespace roma {

void set_error(const char* fmt, ...);

#define ROMA_REQUIRE(cond, code, ...)            \
  do {                                           \
    if (!(cond)) {                               \
      ::roma::set_error(__VA_ARGS__);            \
      return (code);                             \
    }                                            \
  } while (0)

#define ROMA_CHECK_LAUNCH()                                           \
  do {                                                                \
    hipError_t e_ = hipGetLastError();                                \
    if (e_ != hipSuccess) {                                           \
      ::roma::set_error("%s: %s", __func__, hipGetErrorString(e_));   \
      return (int)e_;                                                 \
    }                                                                 \
    return 0;                                                         \
  } while (0)

typedef _Float16 half_t;
typedef __bf16 bf16_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf162_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> { static constexpr int kPer16B = 4; };
template <> struct ElemTraits<half_t> { static constexpr int kPer16B = 8; };
template <> struct ElemTraits<bf16_t> { static constexpr int kPer16B = 8; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(half_t v) { return (float)v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ half_t from_f32<half_t>(float v) { return (half_t)v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// dot of one 16-byte packet (4 f32 / 8 f16 / 8 bf16) accumulated in fp32
template <typename T> __device__ __forceinline__ float dot16(const u32x4& a, const u32x4& b, float acc);
template <> __device__ __forceinline__ float dot16<float>(const u32x4& a, const u32x4& b, float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i) acc = __builtin_fmaf(__uint_as_float(a[i]), __uint_as_float(b[i]), acc);
  return acc;
}
template <> __device__ __forceinline__ float dot16<half_t>(const u32x4& a, const u32x4& b, float acc) {
#ifdef ROMA_NO_DOT2
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t ua = a[i], ub = b[i];
    const half2_t x = __builtin_bit_cast(half2_t, ua), y = __builtin_bit_cast(half2_t, ub);
    acc = __builtin_fmaf((float)x[0], (float)y[0], acc);
    acc = __builtin_fmaf((float)x[1], (float)y[1], acc);
  }
  return acc;
#endif
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t ua = a[i], ub = b[i];   // copy the lane first: bit_cast straight from a vector element reads element 0 (hipcc 7.2)
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, ua), __builtin_bit_cast(half2_t, ub), acc, false);
  }
  return acc;
}
template <> __device__ __forceinline__ float dot16<bf16_t>(const u32x4& a, const u32x4& b, float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t ua = a[i], ub = b[i];
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf162_t, ua), __builtin_bit_cast(bf162_t, ub), acc, false);
  }
  return acc;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// torch.linspace(-1+1/n, 1-1/n, n)[i], evaluated the way ATen does (symmetric around the midpoint)
__device__ __forceinline__ float pix_center(int i, int n) {
  const float start = -1.f + 1.f / n, end = 1.f - 1.f / n;
  if (n == 1) return start;
  const float step = (end - start) / (float)(n - 1);
  return (i < n / 2) ? start + step * i : end - step * (n - 1 - i);
}

template <typename T> __device__ __forceinline__ void unpack16(const u32x4& v, float* f);
template <> __device__ __forceinline__ void unpack16<float>(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v[i]);
}
template <> __device__ __forceinline__ void unpack16<half_t>(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t u = v[i];
    const half2_t h = __builtin_bit_cast(half2_t, u);
    f[2 * i] = (float)h[0];
    f[2 * i + 1] = (float)h[1];
  }
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(v[i] << 16);
    f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
  }
}
template <typename T> __device__ __forceinline__ u32x4 pack16(const float* f);
template <> __device__ __forceinline__ u32x4 pack16<float>(const float* f) {
  return u32x4{__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
}
template <> __device__ __forceinline__ u32x4 pack16<half_t>(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    half2_t h{(half_t)f[2 * i], (half_t)f[2 * i + 1]};
    v[i] = __builtin_bit_cast(uint32_t, h);
  }
  return v;
}
template <> __device__ __forceinline__ u32x4 pack16<bf16_t>(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bf162_t h{(bf16_t)f[2 * i], (bf16_t)f[2 * i + 1]};
    v[i] = __builtin_bit_cast(uint32_t, h);
  }
  return v;
}

// 16-byte store that does NOT keep its line in the XCD's L2 (sc1: write-through, line dropped — MI355X_MICROARCH.md, store
// flavours).  For streaming outputs of kernels whose INPUT reuse lives in the L2: a plain store leaves every output line resident
// and halves the capacity left for the halo rows.
__device__ __forceinline__ void store16_stream(void* p, const u32x4& v) {
  // the s_nop: a VALU write to the data registers of a > 64-bit VMEM store needs wait states after it; hipcc inserts them behind its
  // own stores but cannot see into an asm statement (without it: dword 0 of ~2 % of the packets was the NEXT packet's)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

// ATen upsample_bilinear2d, align_corners=False: src = max((dst+0.5)*in/out-0.5, 0)
__device__ __forceinline__ void interp_src(int d, int in, int out, int& i0, int& i1, float& l1) {
  const float scale = (float)in / (float)out;
  float s = scale * (d + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

// XCD-aware bijective remap of a linear block id: blocks that the dispatcher deals round-robin to one XCD get a
// contiguous range of work items, so neighbouring tiles (which share halo rows) hit the same L2.  Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE property of a kernel: opt it in once per (kernel, device).
// `mask` is one static per call site (bit d = done on device d); racing threads at worst set the attribute twice.
inline int ensure_dyn_smem(const void* fn, int bytes, std::atomic<uint64_t>& mask, const char* what) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) { set_error("%s: hipGetDevice: %s", what, hipGetErrorString(e)); return (int)e; }
  const uint64_t bit = 1ull << (dev & 63);
  if (mask.load(std::memory_order_acquire) & bit) return 0;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute(%d bytes of LDS): %s", what, bytes, hipGetErrorString(e)); return (int)e; }
  mask.fetch_or(bit, std::memory_order_release);
  return 0;
}

// compute units of the current device (cached per device)
inline int num_cus() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  int v = cache[dev & 63].load(std::memory_order_relaxed);
  if (v > 0) return v;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
  cache[dev & 63].store(v, std::memory_order_relaxed);
  return v;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace roma
