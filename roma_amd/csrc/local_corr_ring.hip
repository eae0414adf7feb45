// local_correlation, fp16 / bf16 channels-last, r <= 3: persistent loader / consumer kernel with an LDS ring.
// Reference: romatch/utils/local_correlation.py:4-48 (called at matcher.py:121-125); formulation as in local_corr.hip
// (shared bilinear fraction: (2r+2)^2 integer-grid dot products per pixel, 4-tap blend).
//
// Why this shape (DESIGN.md §3.1, round-1 ablation): the one-tile-per-workgroup kernel was bound by per-tile FIXED latencies —
// flow -> box -> first staged chunk is two dependent global round trips, then ~12 barriers that each drained vmcnt(0) — not by
// HBM, LDS or MFMA.  Here one 512-thread workgroup per CU walks a list of 8x8-pixel tiles and the fixed costs leave the
// critical path:
//   * waves 4-7 are LOADERS: they only issue LDS-DMA (global_load_lds_dwordx4, per-lane source address, no VGPR round trip)
//     into a ring of RING slots and wait with COUNTED s_waitcnt vmcnt(N), so RING-1 channel chunks (tile t's last chunks and
//     tile t+1's first ones alike) stay in flight across the one raw s_barrier per chunk;
//   * waves 0-3 are CONSUMERS: wave g owns the 4x4-pixel group g of the tile: A = the group's 16 f0 rows, B = 16 consecutive
//     target columns of one row of the group's union window, v_mfma_f32_16x16x32_{f16,bf16}, K = 32 channels = one 64-byte LDS
//     row chunk (same conflict-free swizzle as local_corr.hip); the epilogue (scatter into per-pixel (2r+2)^2 images, 4-tap
//     blend, store) is wave-private: no barrier;
//   * the NEXT tile's flow arrives by LDS-DMA too (so nothing in the loop is an ordinary global load that would make hipcc
//     drain the ring), its boxes are computed by the consumers R-1 chunks into the current tile, and its DMA plan by the
//     loaders just before the ring runs into it.
// Tiles whose targets are not compact (incoherent flow: the box or a group's union window does not fit) are skipped by the
// ring and handled afterwards by the consumer waves with per-pixel patches on the VALU, 32 pixels per pass through the
// whole ring area (the L2->LDS gather rate bounds that regime, not this kernel's structure).
#include <cstdlib>
#include "common.h"
#include "lc_device.h"
#include "local_corr_ring.h"

namespace roma {


namespace {

using namespace lc;

template <int R> struct RingGeom {
  static constexpr int N2 = 2 * R + 2;
  static constexpr int MAXR = R <= 2 ? 400 : 464;              // staged box rows (20 x 20 / 21 x 22: scale 1.1 + 3-sigma jitter fits)
  static constexpr int NRUNMAX = 1;                            // 16-column runs per union row (2 would double the accumulators: spills at r = 3)
  static constexpr int GHMAX = N2 + 8;                         // union rows a 4x4 group may span
  static constexpr int NB = GHMAX * NRUNMAX;                   // accumulator blocks per consumer wave
  static constexpr int RING = 4;
};

constexpr int kRingThreads = 512, kLoaders = 256;

template <typename T, int R>
__global__ __launch_bounds__(kRingThreads, 1) void local_corr_ring_kernel(LCRingParams p) {
  using G = RingGeom<R>;
  constexpr int N1 = 2 * R + 1, N2 = G::N2, Q = N2 * N2, K = N1 * N1;
  constexpr int MAXR = G::MAXR, GHMAX = G::GHMAX, NB = G::NB, RING = G::RING;
  constexpr int SLOT_ROWS = kTP + MAXR + 16;                   // + 16: a B fragment reads 16 rows from any box row
  constexpr int SLOT16 = SLOT_ROWS * 4;                        // slot size in 16-byte packets
  constexpr int NL = ((kTP + MAXR) * 4 + kLoaders - 1) / kLoaders;   // DMA rounds per chunk (64 rows each)
  constexpr int E16 = 8, CC = 32;
  // slow path (per-pixel patches): the whole ring area is one buffer
  constexpr int NIT = (Q + 15) / 16, QP = NIT * 16;
  constexpr int SROWS = RING * SLOT_ROWS;                      // rows available
  constexpr int SB = 32;                                       // pixels per pass
  static_assert(kTP + SB * QP + 16 <= SROWS, "slow-path patches must fit the ring area");
  constexpr int SNL = ((kTP + SB * QP) * 4 + kRingThreads - 1) / kRingThreads;   // DMA rounds of a slow pass (all 512 threads)
  constexpr int UB = (SB * NIT + 31) / 32;                     // work items per 16-lane group (32 of them) and pass
  constexpr int MAXT = 64;                                     // tiles per workgroup the slow list can hold (else processed inline... see host)

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* ring = reinterpret_cast<u32x4*>(smem);
  u32x4* s_zero = ring + RING * SLOT16;                        // 16 all-zero rows (slow path)
  float* s_D = reinterpret_cast<float*>(s_zero + 64);          // [kTP][Q+1]: per-wave shares (fallback epilogue images / blended outputs)
  int* s_x0 = reinterpret_cast<int*>(s_D + kTP * (Q + 1));     // [2][kTP]  (indexed by group-major row)
  int* s_y0 = s_x0 + 2 * kTP;
  float* s_ax = reinterpret_cast<float*>(s_y0 + 2 * kTP);
  float* s_ay = s_ax + 2 * kTP;
  int* s_gbox = reinterpret_cast<int*>(s_ay + 2 * kTP);        // [2][kNG][8]: x0,y0,x1,y1,ok,inner
  float* s_flow = reinterpret_cast<float*>(s_gbox + 2 * kNG * 8);   // [2][kTP]: x comps, y comps of the NEXT tile (tile-local order)
  int* s_slow = reinterpret_cast<int*>(s_flow + 2 * kTP);      // [MAXT] + count at [MAXT]
  int* s_org = s_slow + MAXT + 1;                              // [MAXT][4]: b, ty0, tx0 of this workgroup's tiles
  int* s_epi = s_org + MAXT * 4;                               // [kNG]: the group's blended outputs wait in s_D for the loaders to store

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 4;
  const int H = p.H, W = p.W;
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const int NT = p.B * tiles_per_img;
  // Tile list of this workgroup.  Blocks are dealt round-robin to the 8 XCDs (blockIdx % 8 names the group that shares an L2;
  // speed only), so XCD x gets ONE contiguous band of the tile raster for the whole launch and its workgroups sweep that band
  // together, nwx consecutive tiles at a time: vertically adjacent tiles (one raster row apart) then meet in the same L2 within
  // an iteration or two and their shared halo rows are L2 hits.  (Striding the whole raster by gridDim.x instead put vertical
  // neighbours on different XCDs: every f1 row was pulled into 2-3 L2s.)
  const int GW = gridDim.x;
  const int xq = GW >> 3, xr = GW & 7, xcd = blockIdx.x & 7, wi = blockIdx.x >> 3;
  const int nwx = xq + (xcd < xr ? 1 : 0);                                     // workgroups on this XCD
  const int cw = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;   // workgroups on the XCDs before it
  const int band_lo = (int)((long long)NT * cw / GW), band_hi = (int)((long long)NT * (cw + nwx) / GW);
  const int first = band_lo + wi;
  if (first >= band_hi) return;
  const int n_mine = (band_hi - first + nwx - 1) / nwx;         // <= MAXT (the host sizes the grid for it)
  const int NCH = p.C / CC;
  const int total_units = n_mine * NCH;
  const T* f0base = static_cast<const T*>(p.f0);
  const T* f1base = static_cast<const T*>(p.f1);
  T* out = static_cast<T*>(p.out);
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;   // LDS byte offset of the dynamic segment

  // per-tile origins once (the only integer divisions of the kernel)
  for (int k = tid; k < n_mine; k += kRingThreads) {
    const int id = first + k * nwx;
    const int b = id / tiles_per_img;
    const int t = id - b * tiles_per_img;
    const int ty = t / p.tiles_x;
    s_org[k * 4 + 0] = b;
    s_org[k * 4 + 1] = ty * kTH;
    s_org[k * 4 + 2] = (t - ty * p.tiles_x) * kTW;
  }
  if (tid == 0) s_slow[MAXT] = 0;
  if (tid < kNG) s_epi[tid] = 0;
  for (int i = tid; i < 64; i += kRingThreads) s_zero[i] = u32x4{0, 0, 0, 0};
  __syncthreads();

  auto tile_origin = [&](int k, int& b, int& ty0, int& tx0) {   // wave-uniform
    b = __builtin_amdgcn_readfirstlane(s_org[k * 4 + 0]);
    ty0 = __builtin_amdgcn_readfirstlane(s_org[k * 4 + 1]);
    tx0 = __builtin_amdgcn_readfirstlane(s_org[k * 4 + 2]);
  };
  // tile box + eligibility from the four group boxes of parity `par`
  auto tile_box = [&](int par, int& bx0, int& by0, int& bw, int& bh, bool& ok) {
    int lx = 0x3fffffff, ly = 0x3fffffff, hx = -0x3fffffff, hy = -0x3fffffff;
    ok = true;
#pragma unroll
    for (int g = 0; g < kNG; ++g) {
      const int* gb = s_gbox + (par * kNG + g) * 8;
      if (gb[2] >= gb[0] && gb[3] >= gb[1]) {
        lx = min(lx, gb[0]); ly = min(ly, gb[1]); hx = max(hx, gb[2]); hy = max(hy, gb[3]);
      }
      ok = ok && gb[4];
    }
    const bool empty = hx < lx || hy < ly;
    bx0 = __builtin_amdgcn_readfirstlane(empty ? 0 : lx);
    by0 = __builtin_amdgcn_readfirstlane(empty ? 0 : ly);
    bw = __builtin_amdgcn_readfirstlane(empty ? 0 : hx - lx + 1);
    bh = __builtin_amdgcn_readfirstlane(empty ? 0 : hy - ly + 1);
    ok = ok && (bw * bh <= MAXR);
  };
  constexpr int KP = K + 1;                                     // blended outputs of the register epilogue: fp32 [16][KP] per group, inside
                                                                // the group's own share of s_D
  constexpr int NSTORE = (16 * K + 63) / 64;                    // store instructions of one group's copy-out

  // =====================================================================================================================
  // The two roles run SEPARATE loops with the same barrier sequence (2 in the prologue, one per channel chunk, one at the
  // end), so that each role's registers hold only its own state.  Everything that is not MFMA work sits with the loaders,
  // whose waves are otherwise waiting: flow -> targets -> boxes of the NEXT tile, its DMA plan, and the global stores of the
  // PREVIOUS tile's blended outputs.
  // =====================================================================================================================
  if (loader) {
    const int g = wave - 4;                                     // the group this loader wave serves (boxes, copy-out)
    const int ltid = tid - kLoaders;
    // per-thread constants of the DMA plan: LDS packet slot l*256 + ltid <-> row l*64 + (ltid >> 2), k-slice swizzled by row bit 2
    const int prow = ltid >> 2;
    const int kk8 = (((ltid & 3) ^ ((ltid >> 4) & 1))) * E16;
    int fpy, fpx;
    fpix(prow, fpy, fpx);
    uint32_t off[NL];                                           // element offset of every slot inside its batch item
    const T* f0b = f0base;
    const T* f1b = f1base;
    int nl_used = 0;
    int hist[RING - 2];
#pragma unroll
    for (int i = 0; i < RING - 2; ++i) hist[i] = 0;

    auto issue_flow = [&](int k) {                              // wave 4 only: tile k's flow -> s_flow
      int b, ty0, tx0;
      tile_origin(k, b, ty0, tx0);
      const int y = min(ty0 + (lane >> 3), H - 1), x = min(tx0 + (lane & 7), W - 1);
      const float* fp = p.flow + ((size_t)(b * 2) * H + y) * W + x;
      dma4(fp, lds0 + (uint32_t)((unsigned char*)s_flow - smem));
      dma4(fp + (size_t)H * W, lds0 + (uint32_t)((unsigned char*)(s_flow + kTP) - smem));
    };
    auto make_boxes = [&](int k) {                              // targets + group box of tile k from s_flow (or the identity grid)
      int b, ty0, tx0;
      tile_origin(k, b, ty0, tx0);
      const int m = lane & 15;
      const int py = (g >> 1) * 4 + (m >> 2), px = (g & 1) * 4 + (m & 3);
      const int y = ty0 + py, x = tx0 + px;
      int x0 = 0, y0 = 0;
      float ax = 0.f, ay = 0.f;
      int lox = 0x3fffffff, loy = 0x3fffffff, hix = -0x3fffffff, hiy = -0x3fffffff;
      bool inner = false;                                       // pixel inside the image AND its whole (2r+2)^2 window too
      if (y < H && x < W) {
        float fx, fy;
        if (p.flow) {
          fx = s_flow[py * kTW + px];
          fy = s_flow[kTP + py * kTW + px];
        } else {
          fx = -1.f + (2.f * x + 1.f) / W;
          fy = -1.f + (2.f * y + 1.f) / H;
        }
        flow_target(fx, fy, H, W, x0, y0, ax, ay);
        const int a0 = max(x0 - R, 0), a1 = min(x0 + R + 1, W - 1);
        const int b0 = max(y0 - R, 0), b1 = min(y0 + R + 1, H - 1);
        if (a0 <= a1 && b0 <= b1) { lox = a0; hix = a1; loy = b0; hiy = b1; }
        inner = (x0 - R >= 0) && (y0 - R >= 0) && (x0 + R + 1 <= W - 1) && (y0 + R + 1 <= H - 1);
      }
      const bool all_inner = __ballot(inner) == ~0ull;
      const int par = k & 1;
      if (lane < 16) {
        const int r = par * kTP + g * 16 + m;
        s_x0[r] = x0; s_y0[r] = y0; s_ax[r] = ax; s_ay[r] = ay;
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        lox = min(lox, __shfl_xor(lox, o, 64)); loy = min(loy, __shfl_xor(loy, o, 64));
        hix = max(hix, __shfl_xor(hix, o, 64)); hiy = max(hiy, __shfl_xor(hiy, o, 64));
      }
      if (lane == 0) {
        int* gb = s_gbox + (par * kNG + g) * 8;
        const bool empty = hix < lox || hiy < loy;
        gb[0] = lox; gb[1] = loy; gb[2] = hix; gb[3] = hiy;
        gb[4] = empty || ((hix - lox + 1 <= 16) && (hiy - loy + 1 <= GHMAX));
        gb[5] = all_inner;                                      // the register epilogue applies (no zero padding anywhere in the group)
      }
    };
    auto make_plan = [&](int k) {                               // per-lane source offset of every DMA slot of tile k
      int b, ty0, tx0, bx0, by0, bw, bh;
      bool ok;
      tile_origin(k, b, ty0, tx0);
      tile_box(k & 1, bx0, by0, bw, bh, ok);
      f0b = f0base + (size_t)b * H * W * p.f0_pitch;
      f1b = f1base + (size_t)((b + p.f1_shift) % p.B) * H * W * p.f1_pitch;
      const int nrows = bw * bh;
      nl_used = __builtin_amdgcn_readfirstlane(ok ? (kTP + nrows + 63) >> 6 : 0);
      const float inv_bw = 1.0f / (float)max(bw, 1);
      off[0] = (uint32_t)((min(ty0 + fpy, H - 1) * W + min(tx0 + fpx, W - 1)) * p.f0_pitch + kk8);
#pragma unroll
      for (int l = 1; l < NL; ++l) {
        const int rr = (l - 1) * 64 + prow;                     // box row of this slot
        const int ry = min((int)(((float)rr + 0.5f) * inv_bw), max(bh - 1, 0));   // rr / bw, exact for these small integers
        const int y = by0 + ry, x = bx0 + rr - ry * bw;
        off[l] = rr < nrows ? (uint32_t)((y * W + x) * p.f1_pitch + kk8) : 0u;    // slots past the box read a valid dummy
      }
    };
    auto issue_unit = [&](int cv, int slot) -> int {            // chunk cv of the planned tile -> ring slot
      if (p.dbg & 2) return 0;
      const uint32_t base = lds0 + (uint32_t)(slot * SLOT16 * 16) + (uint32_t)g * 1024u;
      const T* a0 = f0b + cv * CC;
      const T* a1 = f1b + cv * CC;
      if (nl_used > 0) dma16(a0 + off[0], base);
#pragma unroll
      for (int l = 1; l < NL; ++l)
        if (l < nl_used) dma16(a1 + off[l], base + (uint32_t)(l * kLoaders * 16));
      return nl_used;
    };
    auto copy_out = [&](int k) -> int {                         // group g's blended outputs of tile k: scale, round, store
      if (!__builtin_amdgcn_readfirstlane(s_epi[g])) return 0;
      int b, ty0, tx0;
      tile_origin(k, b, ty0, tx0);
      const float* s_V = s_D + g * 16 * (Q + 1);
      for (int e = lane; e < 16 * K; e += 64) {
        int m, kk;
        if (p.out_nhwc) { m = e / K; kk = e - m * K; } else { kk = e >> 4; m = e & 15; }
        const int y = ty0 + (g >> 1) * 4 + (m >> 2), x = tx0 + (g & 1) * 4 + (m & 3);
        const size_t o = p.out_nhwc ? (((size_t)b * H + y) * W + x) * p.out_pitch + kk
                                    : (((size_t)b * p.out_pitch + kk) * H + y) * W + x;
        out[o] = from_f32<T>(s_V[m * KP + kk] * p.scale);
      }
      return NSTORE;
    };

    // ---- prologue: tile 0's flow, boxes, plan; the first RING-1 chunks in flight ----
    if (g == 0 && p.flow) {
      issue_flow(0);
      wait_vmcnt(0);
    }
    raw_barrier();
    make_boxes(0);
    raw_barrier();
    make_plan(0);
    int islot = 0;                                              // ring slot of the next unit to issue
    for (int v = 0; v < RING - 1; ++v) {
      int n = 0;
      if (v == 0 && g == 0 && p.flow && n_mine > 1) { issue_flow(1); n += 2; }   // consumed at chunk RING-1 of tile 0
      n += issue_unit(v, islot);                                                  // RING-1 <= NCH: all in tile 0
      islot = islot + 1 == RING ? 0 : islot + 1;
#pragma unroll
      for (int i = RING - 3; i > 0; --i) hist[i] = hist[i - 1];
      hist[0] = n;
    }
    // ---- main loop: one raw barrier per channel chunk ----
    int k = 0, c = 0, kv = 0, cv = RING - 1;
    for (int u = 0; u < total_units; ++u) {
      // everything this wave issued AFTER unit u's DMA: the last RING-2 issue steps
      int n = 0;
#pragma unroll
      for (int i = 0; i < RING - 2; ++i) n += hist[i];
      wait_vmcnt(__builtin_amdgcn_readfirstlane(n));
      raw_barrier();          // unit u has landed for every loader wave; every consumer is done with unit u-1
      n = 0;
      if (c == 0 && k > 0) n += copy_out(k - 1);                // the consumers finished blending tile k-1 before this barrier
      if (c == 0 && u > 0 && g == 0 && p.flow && k + 1 < n_mine) { issue_flow(k + 1); n += 2; }
      if (c == RING - 1 && k + 1 < n_mine) make_boxes(k + 1);   // its flow landed with unit (k, RING-1)
      if (u + RING - 1 < total_units) {
        if (cv == 0) make_plan(kv);
        n += issue_unit(cv, islot);
        islot = islot + 1 == RING ? 0 : islot + 1;
      }
#pragma unroll
      for (int i = RING - 3; i > 0; --i) hist[i] = hist[i - 1];
      hist[0] = n;
      if (++c == NCH) { c = 0; ++k; }
      if (++cv == NCH) { cv = 0; ++kv; }
    }
    raw_barrier();            // the last tile's blended outputs are in s_D
    copy_out(n_mine - 1);
  } else {
    // ---- consumers: wave g owns the 4x4-pixel group g of every tile ----
    const int g = wave;
    const int n16 = lane & 15, kg = lane >> 4;
    const int ks = (kg == 1) ? 2 : (kg == 2 ? 1 : kg);          // K-slice of this lane quarter: (0,2,1,3)
    const int aidx = swzr(g * 16 + n16, ks);
    float4_t acc[NB];
    int bidx[NB];
    int cur_ok = 0, c_inner = 0, nblk = 0;
    int c_b = 0, c_ty0 = 0, c_tx0 = 0, c_gx0 = 0, c_gy0 = 0, c_gw = 0;

    auto setup_tile = [&](int k) {                              // fragment addresses of tile k
      int bx0, by0, bw, bh;
      bool ok;
      tile_origin(k, c_b, c_ty0, c_tx0);
      tile_box(k & 1, bx0, by0, bw, bh, ok);
      cur_ok = __builtin_amdgcn_readfirstlane((int)ok);
      const int* gb = s_gbox + ((k & 1) * kNG + g) * 8;
      const bool gempty = gb[2] < gb[0] || gb[3] < gb[1];
      c_gx0 = __builtin_amdgcn_readfirstlane(gb[0]);
      c_gy0 = __builtin_amdgcn_readfirstlane(gb[1]);
      c_gw = __builtin_amdgcn_readfirstlane(gempty ? 0 : gb[2] - gb[0] + 1);
      const int gh = __builtin_amdgcn_readfirstlane(gempty ? 0 : gb[3] - gb[1] + 1);
      c_inner = __builtin_amdgcn_readfirstlane(gb[5]);
      nblk = cur_ok ? gh : 0;
      const int row0 = kTP + (c_gy0 - by0) * bw + (c_gx0 - bx0) + n16;
#pragma unroll
      for (int j = 0; j < NB; ++j) bidx[j] = (j < nblk) ? swzr(row0 + j * bw, ks) : 0;   // unused blocks read row 0; never looked at
      if (!ok && lane == 0 && g == 0) {
        const int n = s_slow[MAXT];
        if (n < MAXT) { s_slow[n] = k; s_slow[MAXT] = n + 1; }
      }
    };
    auto epilogue = [&](int k) {                                // wave-private: 4-tap blend; the loaders store it after the next barrier
      const int par = k & 1;
      if (c_inner) {
        // Register epilogue (every window of the group lies inside the image): lane (n, kg) holds, for the pixels
        // m = 4 kg + r4, D(m; row gy0 + j, column gx0 + n) in acc[j][r4].  The 4-tap blend is a DPP shift along n (the
        // right-hand tap), an FMA, and an FMA with the next block (the lower tap) — no scatter of the (2r+2)^2 images.
        float* s_V = s_D + g * 16 * (Q + 1);
        float axv[4], ayv[4];
        int iy0[4], ixv[4];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int r = par * kTP + g * 16 + 4 * kg + r4;
          axv[r4] = s_ax[r];
          ayv[r4] = s_ay[r];
          ixv[r4] = c_gx0 + n16 - (s_x0[r] - R);                // window column of this lane
          // window row of union row 0; a lane outside the window columns gets a value that fails the row test for every j
          iy0[r4] = ((unsigned)ixv[r4] < (unsigned)N1) ? c_gy0 - (s_y0[r] - R) : -1000;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const float d = acc[j][r4];
            acc[j][r4] = d + axv[r4] * (right_neighbour(d) - d);
          }
#pragma unroll
        for (int j = 0; j + 1 < NB; ++j)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const int iy = j + iy0[r4];
            if ((unsigned)iy < (unsigned)N1)
              s_V[(4 * kg + r4) * KP + iy * N1 + ixv[r4]] = acc[j][r4] + ayv[r4] * (acc[j + 1][r4] - acc[j][r4]);
          }
        if (lane == 0) s_epi[g] = 1;
        return;
      }
      if (lane == 0) s_epi[g] = 0;
      int wx[4], wy[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        wx[r4] = s_x0[par * kTP + g * 16 + 4 * kg + r4] - R;
        wy[r4] = s_y0[par * kTP + g * 16 + 4 * kg + r4] - R;
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        if (j < nblk && n16 < c_gw) {                           // inside the group's (image-clamped) box
          const int ya = c_gy0 + j, xa = c_gx0 + n16;
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const int dy = ya - wy[r4], dx = xa - wx[r4];
            if (dy >= 0 && dy < N2 && dx >= 0 && dx < N2) s_D[(g * 16 + 4 * kg + r4) * (Q + 1) + dy * N2 + dx] = acc[j][r4] * p.scale;
          }
        }
      }
      // wave-private LDS hand-off: the wave's own ds_write / ds_read execute in order
      for (int e = lane; e < 16 * K; e += 64) {
        int m, kk;
        if (p.out_nhwc) { m = e / K; kk = e - m * K; } else { kk = e >> 4; m = e & 15; }
        const int py = (g >> 1) * 4 + (m >> 2), px = (g & 1) * 4 + (m & 3);
        const int y = c_ty0 + py, x = c_tx0 + px;
        if (y >= H || x >= W) continue;
        const int iy = kk / N1, ix = kk - iy * N1;
        const int r = par * kTP + g * 16 + m;
        const float ax = s_ax[r], ay = s_ay[r];
        const int yy = s_y0[r] - R + iy, xx = s_x0[r] - R + ix;
        const float* d = s_D + (g * 16 + m) * (Q + 1) + iy * N2 + ix;
        const bool y0in = yy >= 0 && yy < H, y1in = yy + 1 >= 0 && yy + 1 < H;
        const bool x0in = xx >= 0 && xx < W, x1in = xx + 1 >= 0 && xx + 1 < W;
        const float d00 = (y0in && x0in) ? d[0] : 0.f, d01 = (y0in && x1in) ? d[1] : 0.f;
        const float d10 = (y1in && x0in) ? d[N2] : 0.f, d11 = (y1in && x1in) ? d[N2 + 1] : 0.f;
        const float top = d00 + ax * (d01 - d00);
        const float bot = d10 + ax * (d11 - d10);
        const size_t o = p.out_nhwc ? (((size_t)c_b * H + y) * W + x) * p.out_pitch + kk
                                    : (((size_t)c_b * p.out_pitch + kk) * H + y) * W + x;
        out[o] = from_f32<T>(top + ay * (bot - top));
      }
    };

    raw_barrier();             // tile 0's flow has landed
    raw_barrier();             // loaders: make_boxes(0)
    int k = 0, c = 0, cslot = 0;
    for (int u = 0; u < total_units; ++u) {
      raw_barrier();
      if (c == 0) setup_tile(k);
      if (cur_ok && !(p.dbg & 1)) {
        const u32x4* slot = ring + cslot * SLOT16;
        const u32x4 a = slot[aidx];
        u32x4 bf[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) bf[j] = slot[bidx[j]];
        __builtin_amdgcn_sched_barrier(0);                      // every fragment read is in flight before the first MFMA waits
        if (c == 0) {                                           // first chunk: C = 0 (no separate zeroing pass)
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[j] = mfma16r(a, bf[j], float4_t{0.f, 0.f, 0.f, 0.f}, T{});
        } else {
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[j] = mfma16r(a, bf[j], acc[j], T{});
        }
        if (c == NCH - 1 && !(p.dbg & 4)) epilogue(k);
      } else if (c == NCH - 1 && lane == 0) {
        s_epi[g] = 0;
      }
      cslot = cslot + 1 == RING ? 0 : cslot + 1;
      if (++c == NCH) { c = 0; ++k; }
    }
    raw_barrier();             // blended outputs of the last tile are visible to the loaders
  }

  // =====================================================================================================================
  // incoherent tiles: per-pixel (2r+2)^2 patches on the VALU, SB pixels per pass through the whole ring area; all 512
  // threads (every DMA of the ring has been waited for by now)
  // =====================================================================================================================
  raw_barrier();
  const int nslow = __builtin_amdgcn_readfirstlane(s_slow[MAXT]);
  if (nslow == 0) return;
  int g16, idx;                                                 // 16-lane ds_read_b128 service groups of a wavefront
  {
    const int l5 = lane & 31;
    int hg;
    if (l5 < 4) { hg = 0; idx = l5; }
    else if (l5 < 12) { hg = 1; idx = l5 - 4; }
    else if (l5 < 16) { hg = 0; idx = l5 - 8; }
    else if (l5 < 20) { hg = 1; idx = l5 - 8; }
    else if (l5 < 28) { hg = 0; idx = l5 - 12; }
    else { hg = 1; idx = l5 - 16; }
    g16 = wave * 4 + (lane >> 5) * 2 + hg;
  }
  constexpr int ZROW = RING * SLOT_ROWS;                        // s_zero starts right after the ring: row index of the zero rows
  for (int si = 0; si < nslow; ++si) {
    const int k = s_slow[si];
    int b, ty0, tx0;
    tile_origin(k, b, ty0, tx0);
    const T* f0 = f0base + (size_t)b * H * W * p.f0_pitch;
    const T* f1 = f1base + (size_t)((b + p.f1_shift) % p.B) * H * W * p.f1_pitch;
    __syncthreads();
    if (tid < kTP) {                                            // per-pixel targets again (meta of parity 0), flow by ordinary loads
      int py, px;
      fpix(tid, py, px);
      const int y = ty0 + py, x = tx0 + px;
      int x0 = 0, y0 = 0;
      float ax = 0.f, ay = 0.f;
      if (y < H && x < W) {
        float fx, fy;
        if (p.flow) {
          fx = p.flow[((size_t)(b * 2 + 0) * H + y) * W + x];
          fy = p.flow[((size_t)(b * 2 + 1) * H + y) * W + x];
        } else {
          fx = -1.f + (2.f * x + 1.f) / W;
          fy = -1.f + (2.f * y + 1.f) / H;
        }
        flow_target(fx, fy, H, W, x0, y0, ax, ay);
      }
      s_x0[tid] = x0; s_y0[tid] = y0; s_ax[tid] = ax; s_ay[tid] = ay;
    }
    for (int i = tid; i < kTP * (Q + 1); i += kRingThreads) s_D[i] = 0.f;
    __syncthreads();
    for (int pass = 0; pass < kTP / SB; ++pass) {
      int pixw[UB], qw[UB], rowidx[UB];
      float sacc[UB];
#pragma unroll
      for (int w = 0; w < UB; ++w) {
        const int uu = g16 + 32 * w;
        const int sl = uu / NIT, it = uu - sl * NIT;
        int pix = (uu < SB * NIT) ? pass * SB + sl : kTP;       // group-major row index of the pixel
        const bool pact = pix < kTP;
        pix = pact ? pix : 0;
        int py, px;
        fpix(pix, py, px);
        const bool pvalid = pact && (ty0 + py < H) && (tx0 + px < W);
        const int q = it * 16 + idx;
        const int yy = s_y0[pix] - R + q / N2, xx = s_x0[pix] - R + q % N2;
        const bool ok = pvalid && (q < Q) && (yy >= 0) && (yy < H) && (xx >= 0) && (xx < W);
        pixw[w] = pvalid ? pix : -1;
        qw[w] = q;
        rowidx[w] = ok ? kTP + sl * QP + q : ZROW;
        sacc[w] = 0.f;
      }
      const T* ssrc[SNL];
      constexpr int used_rows = kTP + SB * QP;
#pragma unroll
      for (int l = 0; l < SNL; ++l) {
        const int slot = l * kRingThreads + tid;
        const int row = slot >> 2;
        const int kk = (slot & 3) ^ ((row >> 2) & 1);
        int y = ty0, x = tx0, pitch = p.f0_pitch;
        const T* base = f0;
        if (row < kTP) {
          int py, px;
          fpix(row, py, px);
          y = min(ty0 + py, H - 1);
          x = min(tx0 + px, W - 1);
        } else if (row < used_rows) {
          const int rr = row - kTP;
          const int sl = rr / QP, q = rr - sl * QP;
          const int pix = pass * SB + sl;
          if (q < Q) {
            y = min(max(s_y0[pix] - R + q / N2, 0), H - 1);
            x = min(max(s_x0[pix] - R + q % N2, 0), W - 1);
            base = f1;
            pitch = p.f1_pitch;
          }
        }
        ssrc[l] = base + ((size_t)y * W + x) * pitch + kk * E16;
      }
      for (int c0 = 0; c0 < p.C; c0 += CC) {
#pragma unroll
        for (int l = 0; l < SNL; ++l)
          if (l * 128 < used_rows) dma16(ssrc[l] + c0, lds0 + (uint32_t)(l * kRingThreads + wave * 64) * 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int w = 0; w < UB; ++w) {
          const int prow = pixw[w] < 0 ? 0 : pixw[w];
          const int r0 = rowidx[w];
          float s = sacc[w];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) s = dot16<T>(ring[swzr(prow, kk)], ring[swzr(r0, kk)], s);
          sacc[w] = s;
        }
        __syncthreads();
      }
#pragma unroll
      for (int w = 0; w < UB; ++w)
        if (qw[w] < Q && pixw[w] >= 0) s_D[pixw[w] * (Q + 1) + qw[w]] = sacc[w] * p.scale;
    }
    __syncthreads();
    for (int e = tid; e < kTP * K; e += kRingThreads) {
      int row, kk;
      if (p.out_nhwc) { row = e / K; kk = e - row * K; } else { kk = e / kTP; row = e - kk * kTP; }
      int py, px;
      fpix(row, py, px);
      const int y = ty0 + py, x = tx0 + px;
      if (y >= H || x >= W) continue;
      const int iy = kk / N1, ix = kk - iy * N1;
      const float ax = s_ax[row], ay = s_ay[row];
      const float* d = s_D + row * (Q + 1) + iy * N2 + ix;
      const float top = d[0] + ax * (d[1] - d[0]);
      const float bot = d[N2] + ax * (d[N2 + 1] - d[N2]);
      const size_t o = p.out_nhwc ? (((size_t)b * H + y) * W + x) * p.out_pitch + kk
                                  : (((size_t)b * p.out_pitch + kk) * H + y) * W + x;
      out[o] = from_f32<T>(top + ay * (bot - top));
    }
  }
}

template <int R> constexpr size_t ring_smem_bytes() {
  using G = RingGeom<R>;
  constexpr int Q = G::N2 * G::N2;
  constexpr int SLOT_ROWS = kTP + G::MAXR + 16;
  return (size_t)G::RING * SLOT_ROWS * 64 + 16 * 64 + (size_t)kTP * (Q + 1) * 4 + 8 * kTP * 4 + 2 * kNG * 8 * 4 + 2 * kTP * 4 +
         (64 + 1) * 4 + 64 * 4 * 4 + kNG * 4;
}

template <typename T, int R>
int launch_ring(LCRingParams p, hipStream_t stream) {
  constexpr size_t smem = ring_smem_bytes<R>();
  static_assert(smem <= 160 * 1024, "LDS budget");
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = ensure_dyn_smem(reinterpret_cast<const void*>(local_corr_ring_kernel<T, R>), (int)smem, attr_done, "roma_local_corr")) return rc;
  p.tiles_x = (p.W + kTW - 1) / kTW;
  p.tiles_y = (p.H + kTH - 1) / kTH;
  { const char* e = getenv("ROMA_LC_DBG"); p.dbg = e ? atoi(e) : 0; }
  const int NT = p.B * p.tiles_x * p.tiles_y;
  int ncu = num_cus();
  // one workgroup per CU (its LDS footprint admits no second); every workgroup's slow list holds 64 tiles
  int grid = NT < ncu ? NT : ncu;
  if ((NT + grid - 1) / grid > 60) grid = ((NT + 59) / 60 + 7) & ~7;   // more workgroups than CUs: the surplus queues behind the first wave
                                                                       // (bands are rounded per XCD: keep a margin under the 64-tile list)
  hipLaunchKernelGGL((local_corr_ring_kernel<T, R>), dim3(grid), dim3(kRingThreads), smem, stream, p);
  ROMA_CHECK_LAUNCH();
}

}  // namespace

// entry used by roma_local_corr (local_corr.hip) for 16-bit channels-last inputs with r <= 3 and C a multiple of 32
int local_corr_ring(const LCRingParams& p, int r, int dtype, hipStream_t stream) {
  if (dtype == ROMA_F16) {
    switch (r) {
      case 1: return launch_ring<half_t, 1>(p, stream);
      case 2: return launch_ring<half_t, 2>(p, stream);
      case 3: return launch_ring<half_t, 3>(p, stream);
    }
  } else if (dtype == ROMA_BF16) {
    switch (r) {
      case 1: return launch_ring<bf16_t, 1>(p, stream);
      case 2: return launch_ring<bf16_t, 2>(p, stream);
      case 3: return launch_ring<bf16_t, 3>(p, stream);
    }
  }
  set_error("local_corr_ring: unsupported r=%d dtype=%d", r, dtype);
  return ROMA_E_UNSUPPORTED;
}

}  // namespace roma
