// Depthwise 5x5 convolution + BatchNorm(eval, folded) + ReLU, channels-last, for gfx950 — the front half of every
// ConvRefiner block: romatch/models/matcher.py:77-103 (create_block: Conv2d(groups=C, k=5, pad=2) -> BatchNorm2d ->
// ReLU), applied 9x per refiner at matcher.py:139-140.
//   y[b,h,w,c] = relu(scale[c] * sum_{dy,dx} w[dy*5+dx][c] * x[b,h+dy-2,w+dx-2,c] + shift[c])
// HBM-bound (one read + one write of the activation): each thread owns one 16-byte channel packet and a strip of
// XS output pixels along W, streams the (XS+4) input columns of each of the 5 rows through registers once and
// reuses every loaded packet for up to 5 taps x XS outputs; fp32 accumulation.  Lanes run along the channel
// packets, so a wavefront's loads are 1 KiB contiguous for C >= 512 (fp16).
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace roma {
namespace {

// Register tile per thread: XS x YS output pixels of one 16-byte channel packet; the (XS+4) x (YS+4) input patch is
// streamed row by row through registers.  Measured on the nine refiner shapes (fp16, B=2, sum of times): 4x1 562 us,
// 4x2 496 us, 8x1 467 us (default), 8x2 574 us, 4x4 607 us; capping registers for occupancy (launch bounds 3-4 waves
// per SIMD) spills and is 4x slower, so MINW stays 1.
template <typename T, int XS, int YS, int MINW>
__global__ __launch_bounds__(256, MINW) void dwconv5x5_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        T* __restrict__ y, int B, int C, int H, int W, int x_pitch, int y_pitch) {
  constexpr int E = ElemTraits<T>::kPer16B;
  const int PK = C / E;
  const int WS = (W + XS - 1) / XS, HS = (H + YS - 1) / YS;
  const size_t total = (size_t)B * HS * WS * PK;
  // XCD-aware block order: the dispatcher deals consecutive workgroups round-robin to the 8 XCDs, so in launch order the strip
  // below a strip (same columns, next row: 4 of its 5 input rows are the same) lands on another XCD's L2 and the rows are
  // fetched over the fabric again — rocprofv3 FETCH_SIZE: 2.8x the input at C = 576, 216 x 216.  With the remap every XCD walks
  // a contiguous band of rows and the halo rows stay in its own L2 (reuse distance: one image row, 0.25 MB).
  const size_t wid = (size_t)xcd_remap(blockIdx.x, gridDim.x);
  for (size_t i = wid * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % PK);
    size_t r = i / PK;
    const int xs = (int)(r % WS) * XS; r /= WS;
    const int ys = (int)(r % HS) * YS;
    const int b = (int)(r / HS);
    const int c0 = k * E;
    float acc[YS][XS][E];
#pragma unroll
    for (int oy = 0; oy < YS; ++oy)
#pragma unroll
      for (int o = 0; o < XS; ++o)
#pragma unroll
        for (int e = 0; e < E; ++e) acc[oy][o][e] = 0.f;
    const T* xb = x + (size_t)b * H * W * x_pitch + c0;
    // interior patches (all but a 2-pixel frame) run with unconditional loads; a guarded load makes hipcc branch and
    // wait vmcnt(0) per element.  Border patches clamp the address and zero the tap through a mask.
    const bool interior = ys >= 2 && ys + YS + 2 <= H && xs >= 2 && xs + XS + 2 <= W;
#pragma unroll 1
    for (int iy = 0; iy < YS + 4; ++iy) {                           // rolled: one input row in registers at a time
      const int yi = ys + iy - 2;
      const float my = (yi >= 0 && yi < H) ? 1.f : 0.f;
      const T* row = xb + (size_t)min(max(yi, 0), H - 1) * W * x_pitch;
      u32x4 raw[XS + 4];
      if (interior) {
#pragma unroll
        for (int cx = 0; cx < XS + 4; ++cx) raw[cx] = *reinterpret_cast<const u32x4*>(row + (size_t)(xs - 2 + cx) * x_pitch);
      } else {
#pragma unroll
        for (int cx = 0; cx < XS + 4; ++cx) raw[cx] = *reinterpret_cast<const u32x4*>(row + (size_t)min(max(xs - 2 + cx, 0), W - 1) * x_pitch);
      }
#pragma unroll
      for (int oy = 0; oy < YS; ++oy) {
        const int dy = iy - oy;                                   // tap row of output row oy fed by input row iy
        if (dy < 0 || dy > 4) continue;
        float wr[5][E];
#pragma unroll
        for (int dx = 0; dx < 5; ++dx)
#pragma unroll
          for (int e = 0; e < E; e += 4)
            *reinterpret_cast<float4_t*>(&wr[dx][e]) = *reinterpret_cast<const float4_t*>(w + (size_t)(dy * 5 + dx) * C + c0 + e);
#pragma unroll
        for (int cx = 0; cx < XS + 4; ++cx) {
          float f[E];
          unpack16<T>(raw[cx], f);
          if (!interior) {
            const int xi = xs - 2 + cx;
            const float m = (xi >= 0 && xi < W) ? my : 0.f;
#pragma unroll
            for (int e = 0; e < E; ++e) f[e] *= m;
          }
#pragma unroll
          for (int dx = 0; dx < 5; ++dx) {
            const int o = cx - dx;
            if (o >= 0 && o < XS) {
#pragma unroll
              for (int e = 0; e < E; ++e) acc[oy][o][e] = __builtin_fmaf(wr[dx][e], f[e], acc[oy][o][e]);
            }
          }
        }
      }
    }
    float sc[E], sh[E];
#pragma unroll
    for (int e = 0; e < E; e += 4) {
      *reinterpret_cast<float4_t*>(&sc[e]) = *reinterpret_cast<const float4_t*>(scale + c0 + e);
      *reinterpret_cast<float4_t*>(&sh[e]) = *reinterpret_cast<const float4_t*>(shift + c0 + e);
    }
#pragma unroll
    for (int oy = 0; oy < YS; ++oy) {
      if (ys + oy >= H) break;
      T* yrow = y + (((size_t)b * H + ys + oy) * W) * y_pitch + c0;
#pragma unroll
      for (int o = 0; o < XS; ++o) {
        if (xs + o >= W) break;
        float v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = fmaxf(__builtin_fmaf(acc[oy][o][e], sc[e], sh[e]), 0.f);
        store16_stream(yrow + (size_t)(xs + o) * y_pitch, pack16<T>(v));
      }
    }
  }
}

template <typename T, int XS, int YS, int MINW = 1>
void launch_dw(const void* x, const float* w, const float* scale, const float* shift, void* y, int B, int C, int H, int W,
               int x_pitch, int y_pitch, hipStream_t s) {
  constexpr int E = ElemTraits<T>::kPer16B;
  const size_t total = (size_t)B * ((H + YS - 1) / YS) * ((W + XS - 1) / XS) * (C / E);
  size_t g = (total + 255) / 256;
  if (g > 32768) g = 32768;
  hipLaunchKernelGGL((dwconv5x5_kernel<T, XS, YS, MINW>), dim3((int)g), dim3(256), 0, s, (const T*)x, w, scale, shift, (T*)y, B, C, H, W, x_pitch, y_pitch);
}


// ---- persistent variant, 16-bit activations (round 2) ------------------------------------------------------------------------
// What rocprofv3 says about the kernel above at C = 576, 216 x 216, B = 2 (profiles/r02_dwconv_pmc.md): the vector ALU is busy 54 %
// of the time, the L1 moves 1.44 GB (37 us of its 64 B/clk) of which 45 % are the 25 x C fp32 TAPS re-read by every thread for every
// row, the index decode is 64-bit division, and a wave issues a row's loads and then waits for them (1.5 waves per SIMD on average,
// no load in flight while it computes).  This variant: the taps live in LDS (one copy per persistent workgroup, read with
// ds_read_b128 beside the L1 traffic), 32-bit index arithmetic, and the NEXT input row is in flight while the current one is
// multiplied (a column's registers are refilled with the next row as soon as the column is converted; the row loop stays rolled so
// hipcc cannot hoist all five rows' loads).  Every
// XCD walks its own contiguous band of the (b, y, strip, packet) index space, so the halo rows of vertically adjacent strips stay
// in one L2, and the outputs are stored write-through (sc1) so that they do not occupy that L2.
template <typename T, int NT>
__global__ __launch_bounds__(NT, NT == 256 ? 2 : 1) void dwconv5x5_lds_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           T* __restrict__ y, int B, int C, int H, int W, int x_pitch, int y_pitch,
                                                           unsigned total, unsigned band) {
  constexpr int E = 8, XS = 8, NC = XS + 4;
  static_assert(sizeof(T) == 2, "16-bit activations");
  extern __shared__ __attribute__((aligned(16))) float s_w[];     // [25][C]
  const int tid = threadIdx.x;
  {                                                               // taps -> LDS, 8 loads in flight per thread (one load per round
    constexpr int SB = 8;                                         // trip would pay a full L2 latency 14-36 times)
    const int n4 = 25 * C / 4;
    for (int e0 = tid; e0 < n4; e0 += NT * SB) {
      float4_t t[SB];
#pragma unroll
      for (int j = 0; j < SB; ++j) t[j] = reinterpret_cast<const float4_t*>(w)[min(e0 + j * NT, n4 - 1)];
#pragma unroll
      for (int j = 0; j < SB; ++j)
        if (e0 + j * NT < n4) reinterpret_cast<float4_t*>(s_w)[e0 + j * NT] = t[j];
    }
  }
  __syncthreads();
  const unsigned PK = (unsigned)C / E, WS = (unsigned)(W + XS - 1) / XS;
  // work order: XCD x (the dispatcher deals workgroup b to XCD b % 8) owns the contiguous eighth `band` of the index space, and its
  // gridDim/8 workgroups walk that band together, NT items each per step — 8 rows of a 216-pixel map per step, so the halo rows
  // of a step are the L2-resident rows of the step before.  (Contiguous per-workgroup chunks instead had every workgroup of an
  // XCD miss on the same halo rows at the same time: FETCH_SIZE 5.8x the input.)
  const unsigned nx = gridDim.x >> 3;
  const unsigned beg = (blockIdx.x & 7u) * band;
  const unsigned end = min(beg + band, total);
  for (unsigned i = beg + (blockIdx.x >> 3) * NT + tid; i < end; i += nx * NT) {
    const unsigned k = i % PK;
    unsigned r = i / PK;
    const int xs = (int)(r % WS) * XS;
    r /= WS;
    const int ys = (int)(r % (unsigned)H), b = (int)(r / (unsigned)H);
    const int c0 = (int)k * E;
    float acc[XS][E];
#pragma unroll
    for (int o = 0; o < XS; ++o)
#pragma unroll
      for (int e = 0; e < E; ++e) acc[o][e] = 0.f;
    const bool interior = ys >= 2 && ys + 3 <= H && xs >= 2 && xs + XS + 2 <= W;
    // 5 taps x 8 channels of fused multiply-adds for every column of one input row (taps of tap-row iy from LDS)
    auto fma_row = [&](const u32x4 (&src)[NC], int iy, auto masked_c) {
      const int yi = ys + iy - 2;
      const float my = (yi >= 0 && yi < H) ? 1.f : 0.f;
      const float* wr = s_w + (size_t)iy * 5 * C + c0;
      float wv[5][E];
#pragma unroll
      for (int dx = 0; dx < 5; ++dx)
#pragma unroll
        for (int e = 0; e < E; e += 4) *reinterpret_cast<float4_t*>(&wv[dx][e]) = *reinterpret_cast<const float4_t*>(wr + dx * C + e);
#pragma unroll
      for (int cx = 0; cx < NC; ++cx) {
        float f[E];
        unpack16<T>(src[cx], f);
        if constexpr (decltype(masked_c)::value) {
          const int xi = xs - 2 + cx;
          const float m = (xi >= 0 && xi < W) ? my : 0.f;
#pragma unroll
          for (int e = 0; e < E; ++e) f[e] *= m;
        }
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
          const int o = cx - dx;
          if (o >= 0 && o < XS) {
#pragma unroll
            for (int e = 0; e < E; ++e) acc[o][e] = __builtin_fmaf(wv[dx][e], f[e], acc[o][e]);
          }
        }
      }
    };
    if (__all(interior)) {                                         // wave-uniform: a wave with one border strip takes the border path whole
      // all but a 2-pixel frame.  Byte offset of (b, row, column xs - 2, packet k) from x in 32 bits (the host checks the tensor is
      // < 4 GB); the 12 columns of a row are the uniform distances cx * x_pitch * 2 from it: a load is scalar base + one vector
      // offset.  Software pipeline: row iy + 1 is requested before row iy is multiplied.  (hipcc converts the whole row to fp32
      // first — 96 registers — and then issues the 12 loads back to back; pinning a finer interleave needs more registers, not fewer.)
      unsigned voff = (unsigned)((((size_t)b * H + ys - 2) * W + xs - 2) * x_pitch + c0) * 2u;
      const unsigned vstep = (unsigned)W * x_pitch * 2u;
      auto load_row = [&](u32x4 (&dst)[NC]) {
#pragma unroll
        for (int cx = 0; cx < NC; ++cx)
          dst[cx] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(x) + (size_t)cx * x_pitch * 2 + voff);
        voff += vstep;
      };
      u32x4 cur[NC], nxt[NC];
      load_row(nxt);
#pragma unroll 1
      for (int iy = 0; iy < 4; ++iy) {
#pragma unroll
        for (int cx = 0; cx < NC; ++cx) cur[cx] = nxt[cx];
        load_row(nxt);
        fma_row(cur, iy, std::false_type{});
      }
      fma_row(nxt, 4, std::false_type{});
    } else {
      // border strips: clamped addresses, taps zeroed through a mask, one row at a time
#pragma unroll 1
      for (int iy = 0; iy < 5; ++iy) {
        const T* row = x + ((size_t)b * H + min(max(ys + iy - 2, 0), H - 1)) * W * x_pitch + c0;
        u32x4 cur[NC];
#pragma unroll
        for (int cx = 0; cx < NC; ++cx) cur[cx] = *reinterpret_cast<const u32x4*>(row + (size_t)min(max(xs - 2 + cx, 0), W - 1) * x_pitch);
        fma_row(cur, iy, std::true_type{});
      }
    }
    float sc[E], sh[E];
#pragma unroll
    for (int e = 0; e < E; e += 4) {
      *reinterpret_cast<float4_t*>(&sc[e]) = *reinterpret_cast<const float4_t*>(scale + c0 + e);
      *reinterpret_cast<float4_t*>(&sh[e]) = *reinterpret_cast<const float4_t*>(shift + c0 + e);
    }
    T* yrow = y + (((size_t)b * H + ys) * W) * y_pitch + c0;
#pragma unroll
    for (int o = 0; o < XS; ++o) {
      if (xs + o >= W) break;
      float v[E];
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = fmaxf(__builtin_fmaf(acc[o][e], sc[e], sh[e]), 0.f);
      store16_stream(yrow + (size_t)(xs + o) * y_pitch, pack16<T>(v));
    }
  }
}

// Launch the persistent variant when it pays: 16-bit activations, the taps fit LDS, and every workgroup gets several rounds of
// work to amortise staging the taps (25 x C x 4 bytes per workgroup out of the L2).  Returns false when the caller should use
// the one-item-per-thread kernel.
template <typename T>
bool launch_dw_lds(const void* x, const float* w, const float* scale, const float* shift, void* y, int B, int C, int H, int W,
                   int x_pitch, int y_pitch, hipStream_t s, int min_rounds, int* rc) {
  const size_t smem = (size_t)25 * C * sizeof(float);
  const size_t total = (size_t)B * H * ((W + 7) / 8) * (C / 8);
  if (smem > 144 * 1024 || total >= (1u << 31) || (size_t)B * H * W * x_pitch * 2 >= (1ull << 32)) return false;
  const int cus = num_cus();
  const bool big = smem > 76 * 1024;                                // one 512-thread workgroup per CU instead of two of 256
  const int nt = big ? 512 : 256, grid = (big ? cus : 2 * cus) & ~7;   // 8 XCD bands: a multiple of 8 workgroups
  if (grid < 8) return false;
  if (total < (size_t)(min_rounds > 1 ? min_rounds : 2) * grid * nt) return false;   // measured break-even (profiles/r02_dwconv_pmc.md)
  const unsigned band = (unsigned)((total + 7) / 8);           // grid is a multiple of 8
  static std::atomic<uint64_t> done_a{0}, done_b{0};
  if (big) {
    if ((*rc = ensure_dyn_smem(reinterpret_cast<const void*>(dwconv5x5_lds_kernel<T, 512>), (int)smem, done_a, "roma_dwconv5x5_bn_relu"))) return true;
    hipLaunchKernelGGL((dwconv5x5_lds_kernel<T, 512>), dim3(grid), dim3(512), smem, s, (const T*)x, w, scale, shift, (T*)y, B, C, H, W,
                       x_pitch, y_pitch, (unsigned)total, band);
  } else {
    if ((*rc = ensure_dyn_smem(reinterpret_cast<const void*>(dwconv5x5_lds_kernel<T, 256>), (int)smem, done_b, "roma_dwconv5x5_bn_relu"))) return true;
    hipLaunchKernelGGL((dwconv5x5_lds_kernel<T, 256>), dim3(grid), dim3(256), smem, s, (const T*)x, w, scale, shift, (T*)y, B, C, H, W,
                       x_pitch, y_pitch, (unsigned)total, band);
  }
  return true;
}


// A third design was built and measured in round 2 (commit 2e370cf, "column-marching"): 4 channels x a 4-pixel strip per thread
// marching down a row segment with the last five input rows as fp32 in a 160-register ring, so that every input is converted
// once instead of five times (19 % fewer VALU instructions, FETCH_SIZE 1.23 x the input).  Parity-green, but the ring leaves one
// wave per SIMD and every LDS / HBM wait is exposed: 86.7 us against this file's 67.8 us at C = 576, 216 x 216.  Removed again.

}  // namespace
}  // namespace roma

using namespace roma;

extern "C" int roma_dwconv5x5_bn_relu(const void* x, const float* w, const float* scale, const float* shift, void* y, int B, int C,
                                      int H, int W, int dtype, int x_pitch, int y_pitch, void* stream) {
  ROMA_REQUIRE(x && w && scale && shift && y, ROMA_E_ARG, "roma_dwconv5x5_bn_relu: null pointer");
  ROMA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && x_pitch >= C && y_pitch >= C, ROMA_E_SHAPE, "roma_dwconv5x5_bn_relu: bad shape");
  ROMA_REQUIRE(dtype >= ROMA_F32 && dtype <= ROMA_BF16, ROMA_E_DTYPE, "roma_dwconv5x5_bn_relu: unknown dtype %d", dtype);
  const int e = dtype == ROMA_F32 ? 4 : 8;
  ROMA_REQUIRE(C % e == 0 && x_pitch % e == 0 && y_pitch % e == 0 && aligned16(x) && aligned16(y) && aligned16(w) &&
                   aligned16(scale) && aligned16(shift),
               ROMA_E_ALIGN, "roma_dwconv5x5_bn_relu: C and pitches must be multiples of %d, all bases 16-byte aligned", e);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // tile shape: ROMA_DW_TILE=XSxYS overrides (tuning aid); default 8x1
  static int tile = -1;
  if (tile < 0) {
    const char* ev = getenv("ROMA_DW_TILE");
    tile = ev ? atoi(ev) : 81;
  }
  static int use_lds = -1;                                          // ROMA_DW_LDS=0: one-item-per-thread kernel everywhere; N >= 2: persistent from N rounds of work (A/B aids)
  if (use_lds < 0) {
    const char* ev = getenv("ROMA_DW_LDS");
    use_lds = ev ? atoi(ev) : 1;
  }
  if (use_lds && dtype != ROMA_F32) {
    int rc = 0;
    const bool took = dtype == ROMA_F16 ? launch_dw_lds<half_t>(x, w, scale, shift, y, B, C, H, W, x_pitch, y_pitch, s, use_lds, &rc)
                                        : launch_dw_lds<bf16_t>(x, w, scale, shift, y, B, C, H, W, x_pitch, y_pitch, s, use_lds, &rc);
    if (took) {
      if (rc) return rc;
      ROMA_CHECK_LAUNCH();
    }
  }
#define ROMA_DW(T)                                                                                                    \
  switch (tile) {                                                                                                     \
    case 41: launch_dw<T, 4, 1>(x, w, scale, shift, y, B, C, H, W, x_pitch, y_pitch, s); break;                       \
    case 42: launch_dw<T, 4, 2>(x, w, scale, shift, y, B, C, H, W, x_pitch, y_pitch, s); break;                       \
    case 82: launch_dw<T, 8, 2>(x, w, scale, shift, y, B, C, H, W, x_pitch, y_pitch, s); break;                       \
    case 44: launch_dw<T, 4, 4>(x, w, scale, shift, y, B, C, H, W, x_pitch, y_pitch, s); break;                       \
    default: launch_dw<T, 8, 1>(x, w, scale, shift, y, B, C, H, W, x_pitch, y_pitch, s); break;                       \
  }
  if (dtype == ROMA_F32) { ROMA_DW(float) } else if (dtype == ROMA_F16) { ROMA_DW(half_t) } else { ROMA_DW(bf16_t) }
#undef ROMA_DW
  ROMA_CHECK_LAUNCH();
}
