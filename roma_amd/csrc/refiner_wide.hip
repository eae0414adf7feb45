// One whole ConvRefiner block at D = 576 (the scale-4 refiner: 140^2 and 216^2 pixels x 2 images, 9 blocks each), fused —
// reference: romatch/models/matcher.py:77-103 (create_block: depthwise 5x5 -> BatchNorm(eval) -> ReLU -> 1x1 conv), :139-140:
//   t[m][k]   = relu(scale[k] * sum_{dy,dx} w25[dy*5+dx][k] * x[m + (dy-2, dx-2)][k] + shift[k])      (rounded to T, as the
//   out[m][n] = bias[n] + sum_k t[m][k] * wt[k][n]                                                     two-kernel path stores it)
// Until round 3 this was the depthwise kernel (68 us at 216^2, VALU-bound) + a hipBLASLt GEMM (85 us), with the (B,h,w,D)
// intermediate written and re-read.  What makes the fusion possible at this width: a 128-pixel x 576-channel output tile is
// 144 accumulator registers on each of 8 waves, so ONE workgroup owns all output channels of its pixels — the depthwise result is
// computed once per pixel (a standard 256 x 256 GEMM tiling would recompute it 2.25x, and it is the VALU-bound half) — and the
// 1x1 weights (648 KB, L2-resident) stream through LDS once per 128 pixels, pre-tiled on the host so that a K-panel is one
// contiguous 36 KB LDS-DMA.  (At D = 1152 the same tile is 64 pixels and the weight stream, 2.6 MB per tile, becomes the bound:
// not built.)
//
// Workgroup = 512 threads, tile = 16 x 8 pixels, K-panels of 32 channels.  Per panel:
//   depthwise (VALU): wave w owns channel packet kq = w & 3 of 64 pixels; the panel's input halo lives in LDS as four packet
//     PLANES [kq][12 rows][32-pixel pitch] (pitch 32: the 64 consecutive-pixel ds_read_b128 of a tap are conflict-free), the taps
//     are wave-uniform (scalar loads), fp32 accumulate (v_fma_mix_f32), BN + ReLU, the packed result goes to t[pixel][64 B];
//   1x1 (MFMA, operands swapped so that a lane ends up with 4 consecutive output channels of one pixel): wave (mh, nq) multiplies
//     N-blocks 9 nq .. 9 nq + 8 (A = weight rows, read from the LDS panel) with M-blocks 4 mh .. 4 mh + 3 (B = t rows);
//   t and the weight panel are [row][64 B] with the packet index XOR-ed by (row >> 1) & 3: conflict-free fragment reads.
// One barrier per panel: between two barriers every wave does the 1x1 of panel kp and the depthwise of panel kp + 1, waves 0-3 in
// that order and waves 4-7 (their SIMD partners) in the opposite one; the weights of panel kp + 1 (LDS-DMA) and the input halo of
// panel kp + 3 (registers) are in flight meanwhile.
#include "common.h"
#include "lc_device.h"

namespace roma {
namespace {

using lc::mfma16r;

struct RWParams {
  const void* x;
  void* y;
  const void* w25p;    // depthwise taps, T, panel-major: [D/32][25 taps][4 packets][8 channels]
  const void* wp;      // panel-major 1x1 weights: [D/32][D out rows][64 B], packets swizzled (see roma_refiner_wide_pack)
  const float* scale;  // (D) folded BN
  const float* shift;  // (D)
  const float* bias;   // (D)
  int B, H, W, x_pitch, y_pitch, tiles_x, tiles_y;
};

// the wave-uniform BN constants are read through the constant address space: scalar loads (s_load_dwordx8), which neither occupy
// vector registers nor count in vmcnt (a vector load inside the panel loop made hipcc wait for vmcnt(0) there — and with it for the
// weight DMA that is meant to stay in flight).  The 25 x 8 taps of a (panel, packet) do NOT come that way: 25 dependent scalar-cache
// misses per panel (the 58 KB table does not fit the 16 KB scalar cache) cost 12 800 cycles per panel, 5x everything else; they are
// staged in LDS with the input halo (16-bit, like the reference's autocast convolution weights) and read as broadcasts.
typedef const __attribute__((address_space(4))) float cfloat;
__device__ __forceinline__ cfloat* as_const(const float* p) { return (cfloat*)(uintptr_t)p; }

constexpr int TW = 16, TH = 8, HXW = TW + 4, HYH = TH + 4, HP = 32;
constexpr int PLANE = HYH * HP * 16 + 16;                       // bytes of one packet plane (+16: the four planes start on different banks)

template <typename T, int D>
__global__ __launch_bounds__(512, 1) void refiner_wide_kernel(RWParams p) {
  constexpr int NKP = D / 32, NB = D / 16, NBW = NB / 4, MBW = 4;
  constexpr int WBUF = D * 64, XBUF = 4 * PLANE, TBUF = TW * TH * 64, DWBUF = 25 * 4 * 16;
  static_assert(NB % 4 == 0 && 2 * (WBUF + XBUF + TBUF + DWBUF) <= 160 * 1024, "tile does not fit");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* s_w = smem;                                    // [2][D][64 B]
  unsigned char* s_x = s_w + 2 * WBUF;                          // [2][4 planes]
  unsigned char* s_t = s_x + 2 * XBUF;                          // [2][128 pixels][64 B]
  unsigned char* s_dw = s_t + 2 * TBUF;                         // [2][25 taps][4 packets][16 B]
  const uint32_t lds_w = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)s_w;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles = p.tiles_x * p.tiles_y;
  const int b = blockIdx.x / tiles, tt = blockIdx.x - b * tiles;
  const int ty0 = (tt / p.tiles_x) * TH, tx0 = (tt % p.tiles_x) * TW;
  const T* xb = static_cast<const T*>(p.x) + (size_t)b * p.H * p.W * p.x_pitch;
  T* yb = static_cast<T*>(p.y) + (size_t)b * p.H * p.W * p.y_pitch;

  // ---- loader role: two (halo pixel, packet) items per thread ----
  uint32_t xsrc[2], xdst[2];
  bool xok[2], xuse[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int j = tid + 512 * s;
    const int pos = min(j >> 2, HXW * HYH - 1), kq = j & 3;
    const int hy = pos / HXW, hx = pos - hy * HXW;
    const int yy = ty0 - 2 + hy, xx = tx0 - 2 + hx;
    xuse[s] = j < 4 * HXW * HYH;
    xok[s] = xuse[s] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
    xsrc[s] = (uint32_t)((min(max(yy, 0), p.H - 1) * p.W + min(max(xx, 0), p.W - 1)) * p.x_pitch + kq * 8);
    xdst[s] = (uint32_t)(kq * PLANE + (hy * HP + hx) * 16);
  }
  // the halo (and taps) of panel kp + 2 are requested at the start of phase kp and written to LDS at its end
  u32x4 xreg[2], dwreg;
  auto load_x = [&](int kp) {
#pragma unroll
    for (int s = 0; s < 2; ++s) xreg[s] = *reinterpret_cast<const u32x4*>(xb + xsrc[s] + kp * 32);
    if (tid < 100) dwreg = reinterpret_cast<const u32x4*>(p.w25p)[kp * 100 + tid];     // that panel's taps
  };
  // (the zero padding of out-of-image halo pixels is applied HERE, a phase after the request: a select right behind the load made
  // every wave wait for its HBM miss at the start of the phase — 32 us of the first version's 191)
  auto store_x = [&](int buf) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
      if (xuse[s]) *reinterpret_cast<u32x4*>(s_x + buf * XBUF + xdst[s]) = xok[s] ? xreg[s] : u32x4{0, 0, 0, 0};
    if (tid < 100) *reinterpret_cast<u32x4*>(s_dw + buf * DWBUF + tid * 16) = dwreg;
  };
  // the weight panel: WBUF / 1024 = 36 wave-instructions, instruction ii = 8 k + wave
  auto dma_w = [&](int kp, int buf) {
    const unsigned char* src = static_cast<const unsigned char*>(p.wp) + (size_t)kp * WBUF;
#pragma unroll
    for (int k = 0; k < (WBUF / 1024 + 7) / 8; ++k) {
      const int ii = k * 8 + wave;
      if (ii < WBUF / 1024) lc::dma16_so(src, (uint32_t)(ii * 1024 + lane * 16), lds_w + (uint32_t)(buf * WBUF + ii * 1024));
    }
  };

  // ---- depthwise role ----
  const int kq = wave & 3;
  const int pl = (wave >> 2) * 64 + lane;                        // pixel of the tile: row pl >> 4, column pl & 15
  const uint32_t xrd = (uint32_t)(kq * PLANE + ((pl >> 4) * HP + (pl & 15)) * 16);
  const uint32_t twr = (uint32_t)(pl * 64 + ((kq ^ ((pl >> 1) & 3)) << 4));
  // ---- MFMA role ----
  const int n16 = lane & 15, kg = lane >> 4;
  const int mh = wave & 1, nq = wave >> 1;
  const uint32_t fo = (uint32_t)(n16 * 64 + ((kg ^ ((n16 >> 1) & 3)) << 4));
  float4_t acc[NBW][MBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) acc[nb][mb] = float4_t{0.f, 0.f, 0.f, 0.f};

  // depthwise 5x5 + BN + ReLU of panel kp: this wave's packet, 64 pixels; X / taps buffer `buf` -> t buffer `buf`
  auto dw_panel = [&](int kp, int buf) {
    const int c0 = __builtin_amdgcn_readfirstlane(kp * 32 + kq * 8);
    const unsigned char* wts = s_dw + buf * DWBUF + kq * 16;
    float a8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a8[e] = 0.f;
    const unsigned char* xs = s_x + buf * XBUF + xrd;
    // one row of taps at a time: its ten LDS reads are all issued before the row's 40 multiply-adds (the SIMD partner runs MFMAs
    // meanwhile, so nothing else hides this wave's LDS latency: one exposed LDS round trip per row instead of one per tap)
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) {
      u32x4 xv[5], wv[5];
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) {
        xv[dx] = *reinterpret_cast<const u32x4*>(xs + (dy * HP + dx) * 16);
        wv[dx] = *reinterpret_cast<const u32x4*>(wts + (dy * 5 + dx) * 64);           // one address per wave: an LDS broadcast
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dx = 0; dx < 5; ++dx)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t u = xv[dx][e], uw = wv[dx][e];
          const half2_t h = __builtin_bit_cast(half2_t, u), hw = __builtin_bit_cast(half2_t, uw);
          a8[2 * e] = __builtin_fmaf((float)h[0], (float)hw[0], a8[2 * e]);
          a8[2 * e + 1] = __builtin_fmaf((float)h[1], (float)hw[1], a8[2 * e + 1]);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    cfloat* sc = as_const(p.scale) + c0;
    cfloat* sh = as_const(p.shift) + c0;
    float r8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) r8[e] = fmaxf(__builtin_fmaf(a8[e], sc[e], sh[e]), 0.f);
    *reinterpret_cast<u32x4*>(s_t + buf * TBUF + twr) = pack16<T>(r8);
  };
  // 1x1 of panel kp on the matrix cores: 9 weight fragments x 4 pixel fragments out of t / weight buffers `buf`
  auto mfma_panel = [&](int buf) {
    const unsigned char* tb = s_t + buf * TBUF + mh * (MBW * 1024) + fo;
    const unsigned char* wb = s_w + buf * WBUF + nq * (NBW * 1024) + fo;
    u32x4 bt[MBW];
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) bt[mb] = *reinterpret_cast<const u32x4*>(tb + mb * 1024);
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      const u32x4 aw = *reinterpret_cast<const u32x4*>(wb + nb * 1024);
#pragma unroll
      for (int mb = 0; mb < MBW; ++mb) acc[nb][mb] = mfma16r(aw, bt[mb], acc[nb][mb], T{});
    }
  };

  // ---- prologue: weights 0, halo 0 -> t 0; halo 1 in LDS; halo 2 requested ----
  dma_w(0, 0);
  load_x(0);
  store_x(0);
  if (NKP > 1) load_x(1);
  __syncthreads();
  dw_panel(0, 0);
  if (NKP > 1) store_x(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // weights 0 landed
  __syncthreads();
  // ---- phase kp: the 1x1 of panel kp AND the depthwise of panel kp + 1, ONE barrier.  The two waves that share a SIMD (w and w + 4)
  // take the two halves in opposite order, so that one's MFMAs run beside the other's VALU work (ablation of the two-barrier version:
  // depthwise 63 us + 1x1 39 us + loads 41 us + skeleton 56 us added up exactly: nothing overlapped).
  auto phase = [&](int kp, auto buf_c) {
    constexpr int buf = decltype(buf_c)::value;                 // t / weight buffer of panel kp; panel kp + 1 uses the other ones
    if (kp + 1 < NKP) dma_w(kp + 1, buf ^ 1);
    if (kp + 2 < NKP) load_x(kp + 2);
    // (one copy of the 1x1 half — 144 accumulator registers pass through it — and the depthwise half before OR after it)
    const bool dw_first = wave >= 4;
    if (dw_first && kp + 1 < NKP) dw_panel(kp + 1, buf ^ 1);
    mfma_panel(buf);
    if (!dw_first && kp + 1 < NKP) dw_panel(kp + 1, buf ^ 1);
    if (kp + 2 < NKP) store_x(buf);                             // halo + taps of panel kp + 2 -> X buffer buf (panel kp's: read a phase ago)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the next panel's weights have landed
    __syncthreads();
  };
  for (int kp = 0; kp < NKP; kp += 2) {
    phase(kp, std::integral_constant<int, 0>{});
    if (kp + 1 < NKP) phase(kp + 1, std::integral_constant<int, 1>{});
  }
  // ---- epilogue: lane (n16, kg) holds output channels 4 kg .. 4 kg + 3 of block nb for pixel (row 4 mh + mb, column n16).  Storing
  // from there (8 bytes per lane, 32-byte runs) is store-ISSUE bound: 288 wave-instructions per tile cost 18 us of the first version's
  // 67 us per tile.  The tile goes through LDS instead (all staging buffers are free now): [128 pixels][OPITCH] 16-bit, then
  // whole 16-byte packets, 1 KB contiguous per wave-instruction.
  constexpr int OPITCH = D * 2 + 16;                            // bytes per pixel row in LDS (+16: rows start on different banks)
  static_assert(TW * TH * OPITCH <= 160 * 1024, "output tile fits LDS");
  unsigned char* s_o = smem;
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const int ch = (nq * NBW + nb) * 16 + 4 * kg;
    const float4_t bv = *reinterpret_cast<const float4_t*>(p.bias + ch);
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) {
      const float4_t v = acc[nb][mb] + bv;
      const half2_t lo{(half_t)v[0], (half_t)v[1]}, hi{(half_t)v[2], (half_t)v[3]};
      uint2 pk;
      pk.x = __builtin_bit_cast(uint32_t, lo);
      pk.y = __builtin_bit_cast(uint32_t, hi);
      *reinterpret_cast<uint2*>(s_o + ((mh * MBW + mb) * 16 + n16) * OPITCH + ch * 2) = pk;
    }
  }
  __syncthreads();
  constexpr int PKR = D / 8;                                    // 16-byte packets per pixel
  for (int id = tid; id < TW * TH * PKR; id += 512) {
    const int px = id / PKR, pk = id - px * PKR;
    const int oy = ty0 + (px >> 4), ox = tx0 + (px & 15);
    if (oy < p.H && ox < p.W)
      *reinterpret_cast<u32x4*>(yb + (size_t)(oy * p.W + ox) * p.y_pitch + pk * 8) = *reinterpret_cast<const u32x4*>(s_o + px * OPITCH + pk * 16);
  }
}

}  // namespace
}  // namespace roma

using namespace roma;

// Host-side helper for the caller's prepare step: nothing on the device.  wp[kp][n][slot] (16-byte packets) <- wt[n][32 kp + 8 (slot ^
// ((n >> 1) & 3)) .. + 8], wt = (D out, D in) row-major 16-bit: the layout roma_refiner_block_wide DMAs panel by panel.
extern "C" int roma_refiner_wide_pack(const void* wt, void* wp, int D) {
  ROMA_REQUIRE(wt && wp, ROMA_E_ARG, "roma_refiner_wide_pack: null pointer");
  ROMA_REQUIRE(D > 0 && D % 64 == 0, ROMA_E_SHAPE, "roma_refiner_wide_pack: D = %d must be a multiple of 64", D);
  const uint16_t* src = static_cast<const uint16_t*>(wt);
  uint16_t* dst = static_cast<uint16_t*>(wp);
  for (int kp = 0; kp < D / 32; ++kp)
    for (int n = 0; n < D; ++n)
      for (int slot = 0; slot < 4; ++slot) {
        const int kgl = slot ^ ((n >> 1) & 3);
        for (int e = 0; e < 8; ++e) dst[((size_t)kp * D + n) * 32 + slot * 8 + e] = src[(size_t)n * D + kp * 32 + kgl * 8 + e];
      }
  return 0;
}

extern "C" int roma_refiner_block_wide(const void* x, const void* w25p, const float* scale, const float* shift, const void* wp,
                                       const float* bias, void* y, int B, int H, int W, int D, int x_pitch, int y_pitch, int dtype,
                                       void* stream) {
  ROMA_REQUIRE(x && w25p && scale && shift && wp && bias && y, ROMA_E_ARG, "roma_refiner_block_wide: null pointer");
  ROMA_REQUIRE(B > 0 && H > 0 && W > 0, ROMA_E_SHAPE, "roma_refiner_block_wide: bad shape");
  ROMA_REQUIRE(D == 576 && dtype == ROMA_F16, ROMA_E_UNSUPPORTED, "roma_refiner_block_wide: built for D = 576, fp16 (got D = %d, dtype %d)", D, dtype);
  ROMA_REQUIRE(x_pitch >= D && y_pitch >= D && x_pitch % 8 == 0 && y_pitch % 4 == 0 && aligned16(x) && aligned16(wp) && aligned16(w25p) &&
               aligned16(y) && y_pitch % 8 == 0 && aligned16(bias), ROMA_E_ALIGN, "roma_refiner_block_wide: pitch / alignment");
  ROMA_REQUIRE((size_t)H * W * x_pitch < (1ull << 31), ROMA_E_SHAPE, "roma_refiner_block_wide: map too large for 32-bit offsets");
  RWParams p{x, y, w25p, wp, scale, shift, bias, B, H, W, x_pitch, y_pitch, (W + TW - 1) / TW, (H + TH - 1) / TH};
  constexpr int smem_loop = 2 * (576 * 64 + 4 * PLANE + TW * TH * 64 + 25 * 4 * 16), smem_out = TW * TH * (576 * 2 + 16);
  constexpr int smem = smem_loop > smem_out ? smem_loop : smem_out;
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = ensure_dyn_smem(reinterpret_cast<const void*>(refiner_wide_kernel<half_t, 576>), smem, attr_done, "roma_refiner_block_wide")) return rc;
  hipLaunchKernelGGL((refiner_wide_kernel<half_t, 576>), dim3(B * p.tiles_x * p.tiles_y), dim3(512), smem, static_cast<hipStream_t>(stream), p);
  ROMA_CHECK_LAUNCH();
}
