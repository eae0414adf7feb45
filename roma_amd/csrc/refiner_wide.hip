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
//     PLANES [kq][12 rows][16 pixel PAIRS][2 half-packets][16 B], a word = one channel of two horizontally adjacent pixels, so that
//     v_dot2_f32_f16 takes two taps per instruction: a lane owns an (even, odd) pixel pair x 4 channels, and both outputs use the same
//     three pair words per tap row — even: (w0,w1).E0 + (w2,w3).E1 + (w4,0).E2, odd: (0,w0).E0 + (w1,w2).E1 + (w3,w4).E2 — 15
//     instructions per output instead of the 25 v_fma_mix of the first version (both forms issue at ~4.2 cycles per SIMD:
//     profiles/r03_valu_rates.txt).  The interleave is two v_perm per word when the loader writes the halo; the six tap pairs per
//     channel come pre-packed from the host (roma_refiner_wide_taps), staged in LDS with the halo; BN + ReLU, the packed result
//     goes to t[pixel][64 B];
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
  const void* w25p;    // depthwise taps, T, as pairs: [D/32][5 tap rows][4 packets][2 half-packets][6 pair sets][4 channels][2]: sets (w0,w1) (w2,w3) (w4,0) (0,w0) (w1,w2) (w3,w4) of tap row (w0..w4)
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

constexpr int TW = 16, TH = 8, HXW = TW + 4, HYH = TH + 4, HPR = 16;   // HPR: pixel pairs per plane row (10 used; 512-byte rows)
constexpr int PLANE = HYH * HPR * 32 + 16;                      // bytes of one packet plane (+16: the four planes start on different banks)
constexpr int NTAPV = 5 * 4 * 2 * 6;                            // 16-byte tap vectors per panel

template <typename T, int D>
__global__ __launch_bounds__(512, 1) void refiner_wide_kernel(RWParams p) {
  constexpr int NKP = D / 32, NB = D / 16, NBW = NB / 4, MBW = 4;
  constexpr int WBUF = D * 64, XBUF = 4 * PLANE, TBUF = TW * TH * 64, DWBUF = NTAPV * 16;
  static_assert(NB % 4 == 0 && 2 * (WBUF + XBUF + TBUF + DWBUF) <= 160 * 1024, "tile does not fit");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* s_w = smem;                                    // [2][D][64 B]
  unsigned char* s_x = s_w + 2 * WBUF;                          // [2][4 planes]
  unsigned char* s_t = s_x + 2 * XBUF;                          // [2][128 pixels][64 B]
  unsigned char* s_dw = s_t + 2 * TBUF;                         // [2][5 tap rows][4 packets][2 halves][6 sets][16 B]
  const uint32_t lds_w = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)s_w;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles = p.tiles_x * p.tiles_y;
  const int b = blockIdx.x / tiles, tt = blockIdx.x - b * tiles;
  const int ty0 = (tt / p.tiles_x) * TH, tx0 = (tt % p.tiles_x) * TW;
  const T* xb = static_cast<const T*>(p.x) + (size_t)b * p.H * p.W * p.x_pitch;
  T* yb = static_cast<T*>(p.y) + (size_t)b * p.H * p.W * p.y_pitch;

  // ---- loader role: one (halo pixel PAIR, packet) item per thread: 12 rows x 10 pairs x 4 packets = 480 of the 512 threads ----
  uint32_t xsrc[2], xdst;
  bool xok[2], xuse;
  {
    const int pr = min(tid >> 2, (HXW / 2) * HYH - 1), kql = tid & 3;
    const int hy = pr / (HXW / 2), jj = pr - hy * (HXW / 2);
    const int yy = ty0 - 2 + hy;
    xuse = tid < 4 * (HXW / 2) * HYH;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int xx = tx0 - 2 + 2 * jj + s;
      xok[s] = xuse && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
      xsrc[s] = (uint32_t)((min(max(yy, 0), p.H - 1) * p.W + min(max(xx, 0), p.W - 1)) * p.x_pitch + kql * 8);
    }
    xdst = (uint32_t)(kql * PLANE + (hy * HPR + jj) * 32);
  }
  // the halo (and taps) of panel kp + 2 are requested at the start of phase kp and written to LDS at its end
  u32x4 xreg[2], dwreg;
  auto load_x = [&](int kp) {
#pragma unroll
    for (int s = 0; s < 2; ++s) xreg[s] = *reinterpret_cast<const u32x4*>(xb + xsrc[s] + kp * 32);
    if (tid < NTAPV) dwreg = reinterpret_cast<const u32x4*>(p.w25p)[kp * NTAPV + tid];     // that panel's tap pairs
  };
  // (the zero padding of out-of-image halo pixels is applied HERE, a phase after the request: a select right behind the load made
  // every wave wait for its HBM miss at the start of the phase — 32 us of the first version's 191)
  auto store_x = [&](int buf) {
    if (xuse) {
      const u32x4 a = xok[0] ? xreg[0] : u32x4{0, 0, 0, 0}, c = xok[1] ? xreg[1] : u32x4{0, 0, 0, 0};
      u32x4 lo, hi;                                             // word = (channel @ even pixel, channel @ odd pixel)
      lo[0] = __builtin_amdgcn_perm(c[0], a[0], 0x05040100u); lo[1] = __builtin_amdgcn_perm(c[0], a[0], 0x07060302u);
      lo[2] = __builtin_amdgcn_perm(c[1], a[1], 0x05040100u); lo[3] = __builtin_amdgcn_perm(c[1], a[1], 0x07060302u);
      hi[0] = __builtin_amdgcn_perm(c[2], a[2], 0x05040100u); hi[1] = __builtin_amdgcn_perm(c[2], a[2], 0x07060302u);
      hi[2] = __builtin_amdgcn_perm(c[3], a[3], 0x05040100u); hi[3] = __builtin_amdgcn_perm(c[3], a[3], 0x07060302u);
      *reinterpret_cast<u32x4*>(s_x + buf * XBUF + xdst) = lo;
      *reinterpret_cast<u32x4*>(s_x + buf * XBUF + xdst + 16) = hi;
    }
    if (tid < NTAPV) *reinterpret_cast<u32x4*>(s_dw + buf * DWBUF + tid * 16) = dwreg;
  };
  // the weight panel: WBUF / 1024 = 36 wave-instructions (1 KB each).  Prologue: all eight waves share them.  Inside a phase they are
  // issued by waves 0-3 ONE AT A TIME between the nine MFMA blocks of their 1x1 half (mfma_panel): issued together at the start of
  // the phase, 36 KB of DMA + the halo loads kept the vector-memory issue of all eight waves blocked for ~900 of the phase's 5800
  // cycles (tools/prof/rw_prof: two plain loads took a depthwise-first wave 1000 cycles to issue), and those waves — whose LDS
  // reads then ran against the landing weights — were the critical path while waves 0-3 idled ~1100 cycles at the barrier.
  auto dma_w_all = [&](int kp, int buf) {
    const unsigned char* src = static_cast<const unsigned char*>(p.wp) + (size_t)kp * WBUF;
#pragma unroll
    for (int k = 0; k < (WBUF / 1024 + 7) / 8; ++k) {
      const int ii = k * 8 + wave;
      if (ii < WBUF / 1024) lc::dma16_so(src, (uint32_t)(ii * 1024 + lane * 16), lds_w + (uint32_t)(buf * WBUF + ii * 1024));
    }
  };
  // ---- depthwise role: lane = (tile row, pixel pair, half-packet) ----
  const int kq = wave & 3;
  const int hsel = lane & 1, jp = (lane >> 1) & 7;
  const int prow = (wave >> 2) * 4 + (lane >> 4);                // tile row
  const int ple = prow * 16 + 2 * jp;                            // the even pixel of the pair (the odd one is ple + 1: same swizzle)
  const uint32_t xrd = (uint32_t)(kq * PLANE + ((prow * HPR + jp) * 2 + hsel) * 16);
  const uint32_t twr = (uint32_t)(ple * 64 + ((kq ^ ((ple >> 1) & 3)) << 4) + hsel * 8);
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
    const unsigned char* wts = s_dw + buf * DWBUF + (kq * 2 + hsel) * 96;
    float ae[4], ao[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) ae[e] = ao[e] = 0.f;
    const unsigned char* xs = s_x + buf * XBUF + xrd;
    // one row of taps at a time: its nine LDS reads are all issued before the row's 24 dot products (the SIMD partner runs MFMAs
    // meanwhile, so nothing else hides this wave's LDS latency: one exposed LDS round trip per row)
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) {
      u32x4 ev[3], tv[6];
#pragma unroll
      for (int k = 0; k < 3; ++k) ev[k] = *reinterpret_cast<const u32x4*>(xs + (dy * HPR + k) * 32);
#pragma unroll
      for (int q = 0; q < 6; ++q) tv[q] = *reinterpret_cast<const u32x4*>(wts + dy * (4 * 2 * 96) + q * 16);   // two addresses per wave
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const half2_t xe = __builtin_bit_cast(half2_t, (uint32_t)ev[k][e]);
          ae[e] = __builtin_amdgcn_fdot2(xe, __builtin_bit_cast(half2_t, (uint32_t)tv[k][e]), ae[e], false);
          ao[e] = __builtin_amdgcn_fdot2(xe, __builtin_bit_cast(half2_t, (uint32_t)tv[3 + k][e]), ao[e], false);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    cfloat* sc = as_const(p.scale) + c0;
    cfloat* sh = as_const(p.shift) + c0;
    uint32_t oe[2], oo[2];
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
      float re[2], ro[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float scv = hsel ? sc[4 + e + i] : sc[e + i], shv = hsel ? sh[4 + e + i] : sh[e + i];
        re[i] = fmaxf(__builtin_fmaf(ae[e + i], scv, shv), 0.f);
        ro[i] = fmaxf(__builtin_fmaf(ao[e + i], scv, shv), 0.f);
      }
      oe[e >> 1] = __builtin_bit_cast(uint32_t, half2_t{(half_t)re[0], (half_t)re[1]});
      oo[e >> 1] = __builtin_bit_cast(uint32_t, half2_t{(half_t)ro[0], (half_t)ro[1]});
    }
    *reinterpret_cast<uint2*>(s_t + buf * TBUF + twr) = uint2{oe[0], oe[1]};
    *reinterpret_cast<uint2*>(s_t + buf * TBUF + twr + 64) = uint2{oo[0], oo[1]};
  };
  // 1x1 of panel kp on the matrix cores: 9 weight fragments x 4 pixel fragments out of t / weight buffers `buf`
  // (kpn >= 0: this wave also streams its share of panel kpn's weights into the other weight buffer, one DMA per block)
  static_assert(4 * NBW == WBUF / 1024, "one weight DMA per wave 0-3 and MFMA block");
  auto mfma_panel = [&](int buf, int kpn) {
    const unsigned char* tb = s_t + buf * TBUF + mh * (MBW * 1024) + fo;
    const unsigned char* wb = s_w + buf * WBUF + nq * (NBW * 1024) + fo;
    const unsigned char* src = static_cast<const unsigned char*>(p.wp) + (size_t)max(kpn, 0) * WBUF;
    u32x4 bt[MBW];
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) bt[mb] = *reinterpret_cast<const u32x4*>(tb + mb * 1024);
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      if (kpn >= 0) {
        const int ii = nb * 4 + wave;
        lc::dma16_so(src, (uint32_t)(ii * 1024 + lane * 16), lds_w + (uint32_t)((buf ^ 1) * WBUF + ii * 1024));
      }
      const u32x4 aw = *reinterpret_cast<const u32x4*>(wb + nb * 1024);
#pragma unroll
      for (int mb = 0; mb < MBW; ++mb) acc[nb][mb] = mfma16r(aw, bt[mb], acc[nb][mb], T{});
    }
  };

  // ---- prologue: weights 0, halo 0 -> t 0; halo 1 in LDS; halo 2 requested ----
  dma_w_all(0, 0);
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
    if (kp + 2 < NKP) load_x(kp + 2);
    // (one copy of the 1x1 half — 144 accumulator registers pass through it — and the depthwise half before OR after it)
    const bool dw_first = wave >= 4;
    if (dw_first && kp + 1 < NKP) dw_panel(kp + 1, buf ^ 1);
    mfma_panel(buf, !dw_first && kp + 1 < NKP ? kp + 1 : -1);
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

// Host-side helper: w25 (25, D) tap-major 16-bit -> the pair layout of RWParams::w25p (60 D elements).  Both pointers in host memory.
extern "C" int roma_refiner_wide_taps(const void* w25, void* w25p, int D) {
  ROMA_REQUIRE(w25 && w25p, ROMA_E_ARG, "roma_refiner_wide_taps: null pointer");
  ROMA_REQUIRE(D > 0 && D % 32 == 0, ROMA_E_SHAPE, "roma_refiner_wide_taps: D = %d must be a multiple of 32", D);
  const uint16_t* src = static_cast<const uint16_t*>(w25);
  uint16_t* dst = static_cast<uint16_t*>(w25p);
  // pair set -> (tap column of the low half, of the high half); -1: zero.  Sets 0-2: even output column, 3-5: odd output column.
  static const int lo[6] = {0, 2, 4, -1, 1, 3}, hi[6] = {1, 3, -1, 0, 2, 4};
  size_t o = 0;
  for (int kp = 0; kp < D / 32; ++kp)
    for (int dy = 0; dy < 5; ++dy)
      for (int kq = 0; kq < 4; ++kq)
        for (int h = 0; h < 2; ++h)
          for (int st = 0; st < 6; ++st)
            for (int ch = 0; ch < 4; ++ch) {
              const int c = kp * 32 + kq * 8 + h * 4 + ch;
              dst[o++] = lo[st] < 0 ? (uint16_t)0 : src[(size_t)(dy * 5 + lo[st]) * D + c];
              dst[o++] = hi[st] < 0 ? (uint16_t)0 : src[(size_t)(dy * 5 + hi[st]) * D + c];
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
  constexpr int smem_loop = 2 * (576 * 64 + 4 * PLANE + TW * TH * 64 + NTAPV * 16), smem_out = TW * TH * (576 * 2 + 16);
  constexpr int smem = smem_loop > smem_out ? smem_loop : smem_out;
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = ensure_dyn_smem(reinterpret_cast<const void*>(refiner_wide_kernel<half_t, 576>), smem, attr_done, "roma_refiner_block_wide")) return rc;
  hipLaunchKernelGGL((refiner_wide_kernel<half_t, 576>), dim3(B * p.tiles_x * p.tiles_y), dim3(512), smem, static_cast<hipStream_t>(stream), p);
  ROMA_CHECK_LAUNCH();
}
