// One whole ConvRefiner block for MID/NARROW widths, fused — reference: romatch/models/matcher.py:77-103 (create_block:
// depthwise 5x5 conv -> BatchNorm(eval) -> ReLU -> 1x1 conv), applied 9x per refiner (matcher.py:139-140):
//   t[m][k]   = relu(scale[k] * sum_{dy,dx} w25[dy*5+dx][k] * x[m + (dy-2, dx-2)][k] + shift[k])
//   out[m][n] = bias[n] + sum_k t[m][k] * wt[k][n]
// for channel counts C <= 160 (the D = 144 and D = 24 refiners of the two finest scales, 0.3-1.5 M pixels per map), where
// the separate depthwise kernel + hipBLASLt GEMM spend 2.3-4x the bandwidth bound (the skinny GEMM alone runs at
// 0.65-2 TB/s) and write + re-read the intermediate.  Here the intermediate never leaves the CU:
//   * persistent 256-thread workgroups (one per CU) walk 16x8-pixel tiles; the 1x1 weights, depthwise weights and folded BN
//     constants are staged in LDS once per workgroup; the input tile + 2-pixel halo once per tile;
//   * depthwise: lane (m, kq) of a wavefront owns pixel column m of two tile rows and every 4th 8-channel packet; each
//     LDS packet read feeds both rows; fp32 accumulate (v_fma_mix_f32 / cvt + fma);
//   * the BN+ReLU result, packed to fp16/bf16, IS the A fragment of v_mfma_f32_16x16x32 (lane (m, kq) holds channels
//     32 ks + 8 kq .. +8 of pixel m), so the 1x1 conv runs on the matrix cores straight out of registers;
//   * results go through LDS for 16-byte coalesced stores.
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "lc_device.h"

namespace roma {
namespace {

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef __bf16 b8v __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float4_t mfma_blk(const u32x4& a, const u32x4& b, float4_t c, half_t) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
}
__device__ __forceinline__ float4_t mfma_blk(const u32x4& a, const u32x4& b, float4_t c, bf16_t) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8v, a), __builtin_bit_cast(b8v, b), c, 0, 0, 0);
}

struct RBParams {
  const void* x;
  void* y;
  const void* w25;    // (25, Kpad) T, zero-padded
  const void* wt;     // (Npad, Kpad) T: wt[n][k] = weight(in k -> out n), zero-padded
  const float* scale; // (Kpad)
  const float* shift; // (Kpad)
  const float* bias;  // (Npad)
  int B, H, W, C, x_pitch, y_pitch;
  int tiles_x, tiles_y;
  float inv_tiles_per_img, inv_tiles_x;   // tile id -> (b, ty, tx) without integer division (exact below 2^21 tiles: host check)
};

constexpr int XT = 16, YT = 8, HX = XT + 4, HY = YT + 4, NPOS = HX * HY;

// PKC: the real 16-byte packets per pixel as a compile-time constant (0: run-time p.C / 8).  The tile staging and the store loop turn
// a linear index into (position, packet) and (row, column) for every 16-byte access; with a run-time divisor each of those is a ~35-
// instruction sequence, and at D = 24 (PKC = 3, the two finest refiners, 18 launches per match) those divisions were as many
// instructions as the 400 multiply-adds of a tile.
template <typename T, int KP, int PKC>
__global__ __launch_bounds__(256, 1) void refiner_block_kernel(RBParams p) {
  constexpr int KPAD = 32 * KP;            // padded channel count (inputs of the 1x1)
  constexpr int NT = KPAD / 16;            // 16-wide output tiles (outputs are padded to KPAD as well)
  constexpr int RS = KPAD / 8 + 1;         // LDS row stride in 16-byte packets (odd: conflict-free column reads)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* s_in = reinterpret_cast<u32x4*>(smem);                 // [NPOS][RS]   input tile + halo
  u32x4* s_pw = s_in + NPOS * RS;                               // [KPAD][RS]   1x1 weights, row = output channel
  u32x4* s_dw = s_pw + KPAD * RS;                               // [25][KPAD/8] depthwise taps
  float* s_cs = reinterpret_cast<float*>(s_dw + 25 * (KPAD / 8));   // scale[KPAD], shift[KPAD], bias[KPAD]
  T* s_out = reinterpret_cast<T*>(s_in);                        // [XT*YT][OS] after the depthwise phase
  constexpr int OS = KPAD + 8;                                  // out row stride in elements

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int m = lane & 15, kq = lane >> 4;
  const int PKT = PKC ? PKC : p.C / 8;                          // real packets per pixel
  const T* x = static_cast<const T*>(p.x);
  T* y = static_cast<T*>(p.y);
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  auto tile_origin = [&](int tile, int& b, int& ty0, int& tx0) {
    b = (int)(((float)tile + 0.5f) * p.inv_tiles_per_img);
    const int tt = tile - b * tiles_per_img;
    const int tyi = (int)(((float)tt + 0.5f) * p.inv_tiles_x);
    ty0 = tyi * YT;
    tx0 = (tt - tyi * p.tiles_x) * XT;
  };

  // ---- once per workgroup: weights and constants ----
  for (int i = tid; i < KPAD * (KPAD / 8); i += 256) {
    const int n = i / (KPAD / 8), k = i % (KPAD / 8);
    s_pw[n * RS + k] = reinterpret_cast<const u32x4*>(p.wt)[i];
  }
  for (int i = tid; i < 25 * (KPAD / 8); i += 256) s_dw[i] = reinterpret_cast<const u32x4*>(p.w25)[i];
  for (int i = tid; i < KPAD; i += 256) {
    s_cs[i] = p.scale[i];
    s_cs[KPAD + i] = p.shift[i];
    s_cs[2 * KPAD + i] = p.bias[i];
  }
  // channel padding of the input tile (packets PKT .. KPAD/8-1) stays zero for the whole kernel
  const int ntile = p.tiles_x * p.tiles_y * p.B;
  constexpr int NLD = (NPOS * (PKC ? PKC : KPAD / 8) + 255) / 256;   // 16-byte loads per thread and tile (upper bound)
  const int nload = NPOS * PKT;
  u32x4 pre[NLD];
  // The next tile's input is fetched into registers while the current one is computed (one workgroup per CU at
  // KPAD = 160: nothing else would hide the HBM latency).  Out-of-image taps become zeros (the conv's padding).
  auto fetch = [&](int tile) {
    int b, ty0, tx0;
    tile_origin(tile, b, ty0, tx0);
    const T* xb = x + (size_t)b * p.H * p.W * p.x_pitch;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int i = min(tid + j * 256, nload - 1);
      const int pos = i / PKT, k = i - pos * PKT;
      const int yy = ty0 - 2 + pos / HX, xx = tx0 - 2 + pos % HX;
      const bool ok = yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
      const int yc = min(max(yy, 0), p.H - 1), xc = min(max(xx, 0), p.W - 1);
      u32x4 v = *reinterpret_cast<const u32x4*>(xb + ((size_t)yc * p.W + xc) * p.x_pitch + k * 8);
      if (!ok) v = u32x4{0, 0, 0, 0};
      pre[j] = v;
    }
  };
  // channel padding of the input tile (packets PKT .. KPAD/8-1) is written once and stays zero
  for (int i = tid; i < NPOS * (KPAD / 8 - PKT); i += 256) {
    const int pos = i / (KPAD / 8 - PKT), k = PKT + i % (KPAD / 8 - PKT);
    s_in[pos * RS + k] = u32x4{0, 0, 0, 0};
  }
  if ((int)blockIdx.x < ntile) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    int b, ty0, tx0;
    tile_origin(tile, b, ty0, tx0);
    __syncthreads();                                            // previous tile's stores have read s_out
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int i = tid + j * 256;
      if (i < nload) {
        const int pos = i / PKT, k = i - pos * PKT;
        s_in[pos * RS + k] = pre[j];
      }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntile) fetch(tile + gridDim.x);
    // ---- depthwise 5x5 + BN + ReLU for pixel column m, tile rows 2wv and 2wv+1 ----
    u32x4 afrag[2][KP];
#pragma unroll
    for (int ks = 0; ks < KP; ++ks) {
      const int pk = 4 * ks + kq;
      float acc0[8], acc1[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc0[e] = acc1[e] = 0.f;
#pragma unroll
      for (int dy = 0; dy < 6; ++dy) {
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
          float fi[8];
          unpack16<T>(s_in[((2 * wv + dy) * HX + m + dx) * RS + pk], fi);
          if (dy < 5) {
            float fw[8];
            unpack16<T>(s_dw[(dy * 5 + dx) * (KPAD / 8) + pk], fw);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc0[e] = __builtin_fmaf(fi[e], fw[e], acc0[e]);
          }
          if (dy >= 1) {
            float fw[8];
            unpack16<T>(s_dw[((dy - 1) * 5 + dx) * (KPAD / 8) + pk], fw);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc1[e] = __builtin_fmaf(fi[e], fw[e], acc1[e]);
          }
        }
      }
      float v0[8], v1[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float sc = s_cs[pk * 8 + e], sh = s_cs[KPAD + pk * 8 + e];
        v0[e] = fmaxf(__builtin_fmaf(acc0[e], sc, sh), 0.f);
        v1[e] = fmaxf(__builtin_fmaf(acc1[e], sc, sh), 0.f);
      }
      afrag[0][ks] = pack16<T>(v0);
      afrag[1][ks] = pack16<T>(v1);
    }
    // ---- 1x1 conv on the matrix cores ----
    float4_t acc[2][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[0][nt] = acc[1][nt] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int ks = 0; ks < KP; ++ks) {
        const u32x4 bfrag = s_pw[(nt * 16 + m) * RS + 4 * ks + kq];
        acc[0][nt] = mfma_blk(afrag[0][ks], bfrag, acc[0][nt], T{});
        acc[1][nt] = mfma_blk(afrag[1][ks], bfrag, acc[1][nt], T{});
      }
    }
    __syncthreads();                                            // every wavefront is done with s_in
    // ---- accumulators (+bias) -> LDS [pixel][channel]; lane (n = m, mq = kq) holds pixels 4mq..4mq+3 of each row ----
#pragma unroll
    for (int row = 0; row < 2; ++row)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int ch = nt * 16 + m;
        const float bsv = s_cs[2 * KPAD + ch];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
          s_out[((2 * wv + row) * XT + 4 * kq + r4) * OS + ch] = from_f32<T>(acc[row][nt][r4] + bsv);
      }
    __syncthreads();
    T* yb = y + (size_t)b * p.H * p.W * p.y_pitch;
    for (int i = tid; i < XT * YT * PKT; i += 256) {
      const int pix = i / PKT, k = i - pix * PKT;
      const int yy = ty0 + pix / XT, xx = tx0 + pix % XT;
      if (yy < p.H && xx < p.W)
        *reinterpret_cast<u32x4*>(yb + ((size_t)yy * p.W + xx) * p.y_pitch + k * 8) = *reinterpret_cast<const u32x4*>(s_out + pix * OS + k * 8);
    }
  }
}

template <typename T, int KP, int PKC>
int launch_rb(RBParams p, hipStream_t s) {
  constexpr int KPAD = 32 * KP, RS = KPAD / 8 + 1;
  const size_t smem = (size_t)(NPOS * RS + KPAD * RS + 25 * (KPAD / 8)) * 16 + 3 * KPAD * 4;
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = ensure_dyn_smem(reinterpret_cast<const void*>(refiner_block_kernel<T, KP, PKC>), (int)smem, attr_done, "roma_refiner_block")) return rc;
  const long ntile = (long)p.B * p.tiles_x * p.tiles_y;
  ROMA_REQUIRE(ntile < (1 << 21), ROMA_E_SHAPE, "roma_refiner_block: %ld tiles in one launch (< 2^21)", ntile);
  p.inv_tiles_per_img = 1.0f / (float)(p.tiles_x * p.tiles_y);
  p.inv_tiles_x = 1.0f / (float)p.tiles_x;
  const int per_cu = KP == 1 ? 4 : 1;                          // LDS footprint: ~25 KB (KP=1) or ~140 KB (KP=5)
  const int grid = ntile < 256 * per_cu ? (int)ntile : 256 * per_cu;
  hipLaunchKernelGGL((refiner_block_kernel<T, KP, PKC>), dim3(grid), dim3(256), smem, s, p);
  ROMA_CHECK_LAUNCH();
}

template <typename T> int launch_rb_toep(RBParams p, hipStream_t s);   // the D = 24 block with the depthwise on the matrix cores (below)

template <typename T>
int launch_rb_any(const RBParams& p, int kpad, hipStream_t s) {
  if (kpad == 32 && p.C == 24) {
    static int toep = -1;                                          // ROMA_RB_TOEP=0: the VALU depthwise (A/B aid)
    if (toep < 0) { const char* e = getenv("ROMA_RB_TOEP"); toep = e ? atoi(e) : 1; }
    if (toep) return launch_rb_toep<T>(p, s);
  }
  if (kpad == 32) return p.C == 24 ? launch_rb<T, 1, 3>(p, s) : launch_rb<T, 1, 0>(p, s);
  return launch_rb<T, 5, 0>(p, s);
}


// ---- The D = 24 block with the DEPTHWISE ON THE MATRIX CORES (late round 3) -------------------------------------------------------
// refiner_block_kernel<T, 1> is bound by its 400 v_fma_mix per tile-lane: the multiply-add forms a 16-bit depthwise can use all issue
// at ~4.2 cycles per SIMD (profiles/r03_valu_rates.txt), 38-76 TMAC/s chip-wide.  A 5-tap row filter is a banded Toeplitz product,
//   out[c][row][x] = sum_dy sum_x' T_{c,dy}[x][x'] in[c][row + dy][x'],   T_{c,dy}[x][x'] = w[c][dy][x' - x] for 0 <= x' - x <= 4,
// i.e. ONE v_mfma_f32_16x16x32 per (channel, tap row) for a 16 x 16-pixel tile: M = output column x, N = tile row, K = halo column x'
// (20 used of 32).  16 % of its multiply-adds are useful — and that is still ~2.6x the dot2 rate and ~5x v_fma_mix.  The price is
// layout: K is the pixel column, so the input halo lives in LDS as channel PLANES [c][20 rows][32 columns] (16-bit scatter writes from
// the channels-last packets) and the BN + ReLU result goes back to channels-last [pixel][32 channels] for the 1x1 (whole 16-byte
// packets: a wave owns the 8 channels of one packet).  The 120 Toeplitz fragments (24 channels x 5 tap rows, 768 bytes each: lane
// quarters 0-2; quarter 3 = columns 24-31 is all zero and never read) are built once per persistent workgroup and stay in LDS.
// Tile = 16 x 16 pixels, 256 threads; per tile: waves 0-2 do the depthwise of packet 0-2 (40 MFMAs each) while the next tile's halo is
// in flight in registers; then every wave 8 MFMAs of the 1x1; accumulators -> LDS -> 16-byte stores; halo scatter of the next tile.
constexpr int TPX = 16, TPY = 16, TPHX = TPX + 4, TPHY = TPY + 4;
constexpr int TP_PR = 64;                                        // bytes per plane row: 32 columns (20 used; the rest stays zero)
constexpr int TP_PLB = TPHY * TP_PR;                             // bytes per channel plane
constexpr int TP_NCH = 24, TP_FRAG = 13 * 16;                    // channels; bytes per Toeplitz fragment: 12 distinct 16-byte windows + a zero one
constexpr int TP_TROW = TPX * 64 + 16;                           // bytes per tile row of t (+16: rows start 4 banks apart: conflict-free packet writes)
constexpr int TP_OS = 40;                                        // out staging: 16-bit elements per pixel (80 bytes)
constexpr int TP_TOEP = TP_NCH * 5 * TP_FRAG;
constexpr int TP_PL = TP_NCH * TP_PLB, TP_TB = TPX * TPY * TP_OS * 2, TP_PW = 32 * 5 * 16;
constexpr int TP_SMEM = TP_TOEP + TP_PL + TP_TB + TP_PW + 3 * 32 * 4;
static_assert(TPY * TP_TROW <= TP_TB && 2 * TP_SMEM <= 160 * 1024, "LDS budget of the Toeplitz block: two workgroups per CU");

template <typename T>
__global__ __launch_bounds__(256, 2) void refiner_block_toep_kernel(RBParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* s_toep = smem;                                  // [24 c][5 dy][13 windows][16 B]
  unsigned char* s_pl = s_toep + TP_TOEP;                        // [24 c][20 rows][64 B]
  unsigned char* s_t = s_pl + TP_PL;                             // t: [16 rows][TP_TROW]; then out staging [256 pixels][80 B]
  u32x4* s_pw = reinterpret_cast<u32x4*>(s_t + TP_TB);           // [32 out channels][5 packets] (4 used)
  float* s_cs = reinterpret_cast<float*>(s_pw + 32 * 5);         // scale[32], shift[32], bias[32]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, q = lane >> 4;
  const T* x = static_cast<const T*>(p.x);
  T* y = static_cast<T*>(p.y);
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  auto tile_origin = [&](int tile, int& b, int& ty0, int& tx0) {
    b = (int)(((float)tile + 0.5f) * p.inv_tiles_per_img);
    const int tt = tile - b * tiles_per_img;
    const int tyi = (int)(((float)tt + 0.5f) * p.inv_tiles_x);
    ty0 = tyi * TPY;
    tx0 = (tt - tyi * p.tiles_x) * TPX;
  };

  // ---- once per workgroup ----
  {
    const T* w25 = static_cast<const T*>(p.w25);                 // (25, 32)
    // Toeplitz fragments.  Lane (mm, qq) of fragment (c, dy) holds T[mm][8 qq + j] = w[c][dy][s + j], s = 8 qq - mm, zero outside taps
    // 0..4: a 16-byte WINDOW of the zero-padded tap row that depends on s alone, and only s = -7 .. 4 gives a non-zero one.  So a
    // fragment is stored as those 12 windows + a zero window (208 bytes instead of 1 KB: the whole table 25 KB instead of 92 KB, which
    // is what lets two workgroups share a CU) and a lane reads window min(s + 7, 12) — equal addresses broadcast, the 13 distinct ones
    // sit in different banks.
    for (int i = tid; i < TP_NCH * 5 * 13; i += 256) {
      const int f = i / 13, wi = i - f * 13, c = f / 5, dy = f - c * 5;
      T v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int dx = wi - 7 + j;
        v[j] = (wi < 12 && dx >= 0 && dx <= 4) ? w25[(dy * 5 + dx) * 32 + c] : from_f32<T>(0.f);
      }
      *reinterpret_cast<u32x4*>(s_toep + (size_t)i * 16) = *reinterpret_cast<const u32x4*>(v);
    }
    for (int i = tid; i < 32 * 4; i += 256) s_pw[(i >> 2) * 5 + (i & 3)] = reinterpret_cast<const u32x4*>(p.wt)[i];
    if (tid < 32) { s_cs[tid] = p.scale[tid]; s_cs[32 + tid] = p.shift[tid]; s_cs[64 + tid] = p.bias[tid]; }
    for (int i = tid; i < (TP_PL + TP_TB) / 16; i += 256) reinterpret_cast<u32x4*>(s_pl)[i] = u32x4{0, 0, 0, 0};   // plane padding, t's fourth packet
  }
  const int ntile = tiles_per_img * p.B;
  // halo of a tile: 400 positions x 3 packets, item idx = k * 400 + position (a wave-instruction = 64 consecutive positions of one packet)
  constexpr int NLD = (TPHX * TPHY * 3 + 255) / 256;
  u32x4 pre[NLD];
  uint32_t okm = 0;                                              // bit j: item j of the fetched tile lies inside the image
  uint32_t pdst[NLD];                                            // plane byte offset of element 0 of the item (0xffffffff: no item)
  int ihy[NLD], ihx[NLD], ik8[NLD];                              // the item's halo row / column / first channel (tile-independent)
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int idx = tid + 256 * j, idc = min(idx, TPHX * TPHY * 3 - 1);
    const int k = idc / (TPHX * TPHY), pos = idc - k * (TPHX * TPHY), hy = pos / TPHX, hx = pos - hy * TPHX;
    pdst[j] = idx < TPHX * TPHY * 3 ? (uint32_t)(k * 8 * TP_PLB + hy * TP_PR + hx * 2) : 0xffffffffu;
    ihy[j] = hy - 2; ihx[j] = hx - 2; ik8[j] = k * 8;
  }
  auto fetch = [&](int tile) {
    int b, ty0, tx0;
    tile_origin(tile, b, ty0, tx0);
    const T* xb = x + (size_t)b * p.H * p.W * p.x_pitch;
    okm = 0xffffffffu;                                           // (the zero padding is applied by scatter(): a select right behind a load waits for it)
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int yy = ty0 + ihy[j], xx = tx0 + ihx[j];
      if (!(yy >= 0 && yy < p.H && xx >= 0 && xx < p.W)) okm &= ~(1u << j);
      pre[j] = *reinterpret_cast<const u32x4*>(xb + (uint32_t)((min(max(yy, 0), p.H - 1) * p.W + min(max(xx, 0), p.W - 1)) * p.x_pitch + ik8[j]));
    }
  };
  auto scatter = [&]() {                                         // registers -> channel planes, one 16-bit write per element
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      if (pdst[j] == 0xffffffffu) continue;
      const bool ok = (okm >> j) & 1u;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t w = ok ? pre[j][e >> 1] : 0u;
        *reinterpret_cast<uint16_t*>(s_pl + pdst[j] + e * TP_PLB) = (uint16_t)((e & 1) ? (w >> 16) : (w & 0xffffu));
      }
    }
  };
  // Tile order: the dispatcher deals workgroup b to XCD b % 8, and every XCD has its own L2.  XCD x owns the contiguous eighth `band` of
  // the tile sequence and its gridDim / 8 workgroups walk that band side by side, so the halo a tile shares with its neighbours (56 % more
  // input than the tile itself) is in that XCD's L2: FETCH_SIZE was 1.65 x the input with tiles dealt round-robin over all workgroups.
  const bool xcd_order = (gridDim.x & 7) == 0 && ntile >= 64;
  const int nx = xcd_order ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int band = xcd_order ? (ntile + 7) >> 3 : ntile;
  const int t_beg = xcd_order ? (int)(blockIdx.x & 7) * band + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int t_end = xcd_order ? min(((int)(blockIdx.x & 7) + 1) * band, ntile) : ntile;
  if (t_beg < t_end) fetch(t_beg);
  __syncthreads();                                               // tables and zero fill done
  if (t_beg < t_end) scatter();
  if (t_beg + nx < t_end) fetch(t_beg + nx);                     // `pre` always holds the tile after the one in the planes
  __syncthreads();

  const int swz = (m >> 1) & 3;                                  // packet swizzle of t rows (pixel column m), as in refiner_wide.hip
  const int wsh = 8 * q - m;                                     // this lane's Toeplitz window: start tap s = 8 q - m, non-zero for -7 <= s <= 4
  const int woff = (wsh >= -7 && wsh <= 4 ? wsh + 7 : 12) * 16;
  // BN constants of this wave's channel packet and the bias of this lane's output channels: in registers for the whole kernel (read
  // from LDS per channel, every read also waited for the ten operand reads issued ahead of it: 3 100 cycles of depthwise per tile)
  float scv[8], shv[8], bsv4[2][4];
#pragma unroll
  for (int e = 0; e < 8; ++e) { scv[e] = s_cs[min(wv, 2) * 8 + e]; shv[e] = s_cs[32 + min(wv, 2) * 8 + e]; }
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int i = 0; i < 4; ++i) bsv4[nb][i] = s_cs[64 + nb * 16 + 4 * q + i];
  for (int tile = t_beg; tile < t_end; tile += nx) {
    int b, ty0, tx0;
    tile_origin(tile, b, ty0, tx0);
    const bool more = tile + nx < t_end, more2 = tile + 2 * nx < t_end;
    // ---- depthwise + BN + ReLU: wave k < 3 owns channel packet k; lane (m = tile row for B / output column for A, q) ----
    if (wv < 3) {
      float vals[8][4];
      // the ten operand reads of channel e + 1 are issued before the five MFMAs of channel e (left to itself hipcc put every pair of
      // reads and an lgkmcnt(0) directly in front of its MFMA: 40 exposed LDS round trips per tile, 103 us at 864^2)
      // hipcc put every pair of operand reads and an lgkmcnt(0) directly in front of its MFMA (40 exposed LDS round trips per tile);
      // with all ten reads of the next channel issued ahead, the 4-bit lgkmcnt still made every channel wait for reads just issued.
      // So: the two reads for tap row dy of channel e + 1 go out right behind the MFMA of tap row dy of channel e (eight reads younger
      // than the one an MFMA waits for).
      u32x4 av[5], bv[5];
      auto load_op = [&](int c, int dy, u32x4& a, u32x4& bq) {
        a = *reinterpret_cast<const u32x4*>(s_toep + (c * 5 + dy) * TP_FRAG + woff);
        bq = *reinterpret_cast<const u32x4*>(s_pl + c * TP_PLB + (m + dy) * TP_PR + q * 16);
      };
#pragma unroll
      for (int dy = 0; dy < 5; ++dy) load_op(wv * 8, dy, av[dy], bv[dy]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = wv * 8 + e;
        u32x4 an[5], bn[5];
        float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 5; ++dy) {
          __builtin_amdgcn_sched_barrier(0);
          acc = lc::mfma16r(av[dy], bv[dy], acc, T{});
          if (e < 7) load_op(c + 1, dy, an[dy], bn[dy]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) vals[e][i] = fmaxf(__builtin_fmaf(acc[i], scv[e], shv[e]), 0.f);
        if (e < 7) {
#pragma unroll
          for (int dy = 0; dy < 5; ++dy) { av[dy] = an[dy]; bv[dy] = bn[dy]; }
        }
      }
      // lane (q, m) holds output columns 4 q + i of tile row m, 8 channels: one 16-byte packet per pixel
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float f8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f8[e] = vals[e][i];
        const int xo = 4 * q + i;
        *reinterpret_cast<u32x4*>(s_t + m * TP_TROW + xo * 64 + ((wv ^ ((xo >> 1) & 3)) << 4)) = pack16<T>(f8);
      }
    }
    __syncthreads();
    // ---- 1x1 on the matrix cores: wave w takes tile rows 4 w .. 4 w + 3, both 16-channel output blocks.  Operands swapped (A = weight
    // rows, B = t rows) so that lane (q, m) ends up with output channels 16 nb + 4 q .. + 3 of pixel column m: 8-byte staging writes ----
    float4_t acc2[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const u32x4 tb = *reinterpret_cast<const u32x4*>(s_t + (wv * 4 + r) * TP_TROW + m * 64 + ((q ^ swz) << 4));
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) acc2[r][nb] = lc::mfma16r(s_pw[(nb * 16 + m) * 5 + q], tb, float4_t{0.f, 0.f, 0.f, 0.f}, T{});
    }
    __syncthreads();                                             // every wave is done reading t
    T* s_out = reinterpret_cast<T*>(s_t);                        // [256 pixels][TP_OS]
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        T o4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o4[i] = from_f32<T>(acc2[r][nb][i] + bsv4[nb][i]);
        *reinterpret_cast<uint2*>(s_out + ((wv * 4 + r) * TPX + m) * TP_OS + nb * 16 + 4 * q) = *reinterpret_cast<const uint2*>(o4);
      }
    // The halo pipeline is two tiles deep: the registers hold tile N + 1 (requested a whole iteration ago), they go to the planes (free
    // since the first barrier of this tile), and tile N + 2 is requested IN FRONT OF this tile's global stores: requested behind them
    // (at the top of the loop) the loads sat behind an s_waitcnt vmcnt(0) for the stores — register reuse — 2 400 cycles per tile.
    if (more) scatter();
    if (more2) fetch(tile + 2 * nx);
    __syncthreads();
    T* yb = y + (size_t)b * p.H * p.W * p.y_pitch;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = tid + 256 * j, pix = i / 3, k = i - pix * 3;
      const int yy = ty0 + (pix >> 4), xx = tx0 + (pix & 15);
      if (yy < p.H && xx < p.W)
        *reinterpret_cast<u32x4*>(yb + ((size_t)yy * p.W + xx) * p.y_pitch + k * 8) = *reinterpret_cast<const u32x4*>(s_out + pix * TP_OS + k * 8);
    }
    __syncthreads();                                             // staging read; t's fourth packet must be zero again before the next depthwise
    for (int i = tid; i < TPX * TPY; i += 256) {
      const int row = i >> 4, xo = i & 15;
      *reinterpret_cast<u32x4*>(s_t + row * TP_TROW + xo * 64 + ((3 ^ ((xo >> 1) & 3)) << 4)) = u32x4{0, 0, 0, 0};
    }
    // (no barrier needed here: the depthwise writes other slots of t, and the next reader of the zero slots is behind the next barrier)
  }
}

template <typename T>
int launch_rb_toep(RBParams p, hipStream_t s) {
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = ensure_dyn_smem(reinterpret_cast<const void*>(refiner_block_toep_kernel<T>), TP_SMEM, attr_done, "roma_refiner_block")) return rc;
  p.tiles_x = (p.W + TPX - 1) / TPX;
  p.tiles_y = (p.H + TPY - 1) / TPY;
  const long ntile = (long)p.B * p.tiles_x * p.tiles_y;
  ROMA_REQUIRE(ntile < (1 << 21), ROMA_E_SHAPE, "roma_refiner_block: %ld tiles in one launch (< 2^21)", ntile);
  p.inv_tiles_per_img = 1.0f / (float)(p.tiles_x * p.tiles_y);
  p.inv_tiles_x = 1.0f / (float)p.tiles_x;
  const int grid = ntile < 2 * num_cus() ? (int)ntile : 2 * num_cus();      // persistent, two workgroups per CU
  hipLaunchKernelGGL((refiner_block_toep_kernel<T>), dim3(grid), dim3(256), TP_SMEM, s, p);
  ROMA_CHECK_LAUNCH();
}


// (The same depthwise as a STAND-ALONE kernel for the wide levels — one workgroup per group of 24 channels, output packets straight to
// global — was built and measured late in round 3 and removed again: 1.5-1.8 TB/s at D = 144 .. 1384 against the 2.3-3.2 TB/s of
// dwconv.hip's VALU kernel; the 48-byte pieces per pixel, the halo re-read per group and the 16-bit plane scatter cost more than the
// matrix cores save.  It wins only where a whole pixel is one group, i.e. here.  Numbers in DESIGN.md §3.3.)

// ---- 1x1 convolution alone at mid widths (32 < C <= 160), for the refiner whose depthwise half stays a separate kernel:
// y[m][n] = bias[n] + sum_k x[m][k] * wt[n][k].  hipBLASLt runs this skinny GEMM (N = K = 144, M = 10^5..10^6) at ~2 TB/s of
// in+out; here it is a streaming kernel: the weights sit in LDS once per (persistent) workgroup, every wavefront takes 16
// rows at a time, its A fragments are the 16-byte channel packets straight from global memory, the products go
// through a wave-private LDS patch for 16-byte stores.  No workgroup barrier after the weight load. ----
struct PWParams {
  const void* x;
  void* y;
  const void* wt;     // (kpad, kpad) T: wt[n][k], zero-padded
  const float* bias;  // (kpad)
  long M;
  int C, x_pitch, y_pitch;
};

template <typename T, int KP>
__global__ __launch_bounds__(256, 2) void pointwise_mfma_kernel(PWParams p) {
  constexpr int KPAD = 32 * KP, NT = KPAD / 16, RS = KPAD / 8 + 1, OS = KPAD + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* s_pw = reinterpret_cast<u32x4*>(smem);                         // [KPAD][RS]
  float* s_b = reinterpret_cast<float*>(s_pw + KPAD * RS);              // [KPAD]
  T* s_out = reinterpret_cast<T*>(s_b + KPAD);                          // [4 waves][16][OS]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int m = lane & 15, kq = lane >> 4;
  const int PKT = p.C / 8;
  for (int i = tid; i < KPAD * (KPAD / 8); i += 256) {
    const int n = i / (KPAD / 8), k = i % (KPAD / 8);
    s_pw[n * RS + k] = reinterpret_cast<const u32x4*>(p.wt)[i];
  }
  for (int i = tid; i < KPAD; i += 256) s_b[i] = p.bias[i];
  __syncthreads();
  const T* x = static_cast<const T*>(p.x);
  T* y = static_cast<T*>(p.y);
  T* so = s_out + wv * 16 * OS;
  const long ntile = (p.M + 15) / 16;
  for (long tile = (long)blockIdx.x * 4 + wv; tile < ntile; tile += (long)gridDim.x * 4) {
    const long row0 = tile * 16;
    const long r = row0 + m < p.M ? row0 + m : p.M - 1;
    u32x4 a[KP];
#pragma unroll
    for (int ks = 0; ks < KP; ++ks) {
      const int pk = 4 * ks + kq;
      u32x4 v = *reinterpret_cast<const u32x4*>(x + r * p.x_pitch + (pk < PKT ? pk : 0) * 8);
      if (pk >= PKT) v = u32x4{0, 0, 0, 0};
      a[ks] = v;
    }
    float4_t acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      acc[nt] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KP; ++ks) acc[nt] = mfma_blk(a[ks], s_pw[(nt * 16 + m) * RS + 4 * ks + kq], acc[nt], T{});
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float bv = s_b[nt * 16 + m];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) so[(4 * kq + r4) * OS + nt * 16 + m] = from_f32<T>(acc[nt][r4] + bv);
    }
    for (int i = lane; i < 16 * PKT; i += 64) {
      const int row = i / PKT, k = i - row * PKT;
      if (row0 + row < p.M) *reinterpret_cast<u32x4*>(y + (row0 + row) * p.y_pitch + k * 8) = *reinterpret_cast<const u32x4*>(so + row * OS + k * 8);
    }
  }
}

template <typename T, int KP>
int launch_pw(const PWParams& p, hipStream_t s) {
  constexpr int KPAD = 32 * KP, RS = KPAD / 8 + 1, OS = KPAD + 8;
  const size_t smem = (size_t)KPAD * RS * 16 + KPAD * 4 + (size_t)4 * 16 * OS * sizeof(T);
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = ensure_dyn_smem(reinterpret_cast<const void*>(pointwise_mfma_kernel<T, KP>), (int)smem, attr_done, "roma_pointwise_mfma")) return rc;
  const long need = (p.M + 63) / 64;
  const int grid = (int)(need < 512 ? need : 512);             // two persistent workgroups per CU
  hipLaunchKernelGGL((pointwise_mfma_kernel<T, KP>), dim3(grid), dim3(256), smem, s, p);
  ROMA_CHECK_LAUNCH();
}

}  // namespace
}  // namespace roma

using namespace roma;

extern "C" int roma_refiner_block(const void* x, const void* w25, const float* scale, const float* shift, const void* wt,
                                  const float* bias, void* y, int B, int C, int H, int W, int kpad, int dtype, int x_pitch, int y_pitch,
                                  void* stream) {
  ROMA_REQUIRE(x && w25 && scale && shift && wt && bias && y, ROMA_E_ARG, "roma_refiner_block: null pointer");
  ROMA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && x_pitch >= C && y_pitch >= C, ROMA_E_SHAPE, "roma_refiner_block: bad shape");
  ROMA_REQUIRE(dtype == ROMA_F16 || dtype == ROMA_BF16, ROMA_E_DTYPE, "roma_refiner_block: fp16 / bf16 only (the fp32 parity mode uses the separate kernels)");
  ROMA_REQUIRE((kpad == 32 || kpad == 160) && C <= kpad && C % 8 == 0, ROMA_E_UNSUPPORTED, "roma_refiner_block: C=%d, kpad=%d (kpad must be 32 or 160, C a multiple of 8)", C, kpad);
  ROMA_REQUIRE(x_pitch % 8 == 0 && y_pitch % 8 == 0 && aligned16(x) && aligned16(y) && aligned16(w25) && aligned16(wt), ROMA_E_ALIGN,
               "roma_refiner_block: pitches must be multiples of 8 and bases 16-byte aligned");
  RBParams p{x, y, w25, wt, scale, shift, bias, B, H, W, C, x_pitch, y_pitch, (W + XT - 1) / XT, (H + YT - 1) / YT, 0.f, 0.f};
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == ROMA_F16 ? launch_rb_any<half_t>(p, kpad, s) : launch_rb_any<bf16_t>(p, kpad, s);
}

extern "C" int roma_pointwise_mfma(const void* x, const void* wt, const float* bias, void* y, long M, int C, int kpad, int dtype,
                                   int x_pitch, int y_pitch, void* stream) {
  ROMA_REQUIRE(x && wt && bias && y, ROMA_E_ARG, "roma_pointwise_mfma: null pointer");
  ROMA_REQUIRE(M > 0 && C > 0 && x_pitch >= C && y_pitch >= C, ROMA_E_SHAPE, "roma_pointwise_mfma: bad shape");
  ROMA_REQUIRE(dtype == ROMA_F16 || dtype == ROMA_BF16, ROMA_E_DTYPE, "roma_pointwise_mfma: fp16 / bf16 only");
  ROMA_REQUIRE((kpad == 32 || kpad == 160) && C <= kpad && C % 8 == 0, ROMA_E_UNSUPPORTED, "roma_pointwise_mfma: C=%d, kpad=%d", C, kpad);
  ROMA_REQUIRE(x_pitch % 8 == 0 && y_pitch % 8 == 0 && aligned16(x) && aligned16(y) && aligned16(wt), ROMA_E_ALIGN,
               "roma_pointwise_mfma: pitches must be multiples of 8 and bases 16-byte aligned");
  PWParams p{x, y, wt, bias, M, C, x_pitch, y_pitch};
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == ROMA_F16) return kpad == 32 ? launch_pw<half_t, 1>(p, s) : launch_pw<half_t, 5>(p, s);
  return kpad == 32 ? launch_pw<bf16_t, 1>(p, s) : launch_pw<bf16_t, 5>(p, s);
}
