// local_correlation, fp16 / bf16 channels-last, r <= 3, C = 256 or 512: the ROW-STREAMING kernel (round 3).
// Reference: romatch/utils/local_correlation.py:4-48 (called at matcher.py:121-125); formulation as in local_corr.hip:
// corr[p][iy][ix] = 4-tap blend of D[p][j][i] = <f0[p], f1[y0-r+j][x0-r+i]> on the (2r+2)^2 integer patch.
//
// Why another kernel (profiles/r02_local_corr_pmc.md, gpurun_out/r3_stage_micro.txt): the 32-channel-chunk kernels are bound by the
// L2 -> LDS staging path, and tools/stage_micro.hip measured what that path can do on the same box of the same map: 64-byte pieces
// per pixel (what the chunk kernels request) 10.4-12.1 TB/s, whole 512-byte pixel rows 14.1 TB/s, and 15.6-16.0 TB/s once a ring
// keeps two stages in flight across the barrier — against the 7.2 TB/s the chunk kernels reach, whose every chunk is
// issue -> wait for all of it -> barrier -> compute -> barrier.  So here
//   * a stage is one ROW of the tile's f1 box: bw pixels x 256 channels, every pixel a contiguous 512-byte request, DMA'd into a
//     4-slot LDS ring, two stages in flight behind a counted vmcnt, ONE barrier per stage;
//   * f0 never touches LDS: wave g owns the 4x4-pixel group g and holds its 16 x C operand in registers (32 / 64 VGPRs), loaded
//     straight from global memory before the flow is even read.  Because A comes from registers the K order is free, and lane
//     quarter kg takes channels [64 kg, 64 kg + 64): with the 16-byte pieces of a staged pixel XOR-swizzled by (pixel & 7) inside
//     each 128-byte quarter, the B-fragment ds_read_b128 of 16 consecutive pixels is conflict-free for any first pixel;
//   * the K loop runs over the whole channel block inside a stage, so one (group, row) accumulator is complete when its stage is
//     done: x-blend by a DPP shift, y-blend against the previous row kept in 4 registers, and the result goes straight into the
//     group's [16 pixels][K] output image in LDS (a dump slot takes what falls outside a pixel's window: no exec juggling) —
//     8 accumulator registers instead of 48-64, and the box may be as tall as it likes;
//   * image borders need no second code path: out-of-image box columns are staged from a clamped address and zeroed by a select,
//     out-of-image rows are not staged at all (the previous-row registers start at zero; one virtual row below the image).
// Tiles whose targets are not compact (box wider than BWMAX, a group wider than 16 columns) gather one patch per pixel through the same
// ring (patch path below).  r = 7 on small maps (the 40 x 40 scale-16 level): up to three column blocks per group and a box that may be
// the whole map, so every tile streams.  Tile = 8 x 8 pixels, 4 waves (an 8 x 16 / 8-wave variant measured 1.5-2x slower: its waves
// idle through half of the 26 stages and 150 VGPRs leave one workgroup per CU at C = 512; gpurun_out/r3_lcb3.txt).
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "lc_device.h"
#include "lc_variants.h"

namespace roma {
namespace {

using namespace lc;

template <int R, int NW> struct RowsGeom {
  static constexpr int N1 = 2 * R + 1, K = N1 * N1;
  // r <= 3: one 16-column MFMA block per group, boxes up to 20 pixels wide, patch path for what does not fit.
  // r >= 4 (the 40 x 40 / r = 7 level): up to three blocks per group and a box as wide as the whole map (W <= 47, host check), so
  // EVERY tile streams, however incoherent the flow — at that size a box that is the whole image (1.6 MB per tile) is a tenth of what
  // 64 private 16 x 16 patches cost.
  static constexpr int NBLK = R <= 3 ? 1 : 3;
  static constexpr int BWMAX = R <= 3 ? 20 : 48;               // widest staged box (pixels); a multiple of 2 NW: whole DMA rounds
  static constexpr int SLOTB = BWMAX * 512;                    // one stage: BWMAX pixels x 256 channels
  static constexpr int NS = 4;                                 // ring slots; NS - 2 stages in flight beside the one being read
  static constexpr bool kPatchPath = R <= 3;
  static constexpr int PF = NS - 2;
  static constexpr int RINGB = NS * SLOTB;
  static constexpr int NIT = SLOTB / 1024;                     // DMA wave-instructions per stage (two pixels each)
  static constexpr int NIW = (NIT + NW - 1) / NW;
  // Output image of one group: [17 pixel slots][N1 + 2 rows][N1] elements.  A lane writes one element per accumulator row for EVERY
  // box row the group walks: window row tt = ya - (y0 - r) - 1 clamped to [-1, N1], so rows 0 and N1 + 1 of a slot collect what
  // lies above / below the pixel's window, and slot 16 takes the lanes whose box column is outside the pixel's window.
  static constexpr int TR = N1 + 2;
  static constexpr int PPB = TR * N1 * 2;                      // bytes per pixel slot
  static constexpr int OTB = ((17 * PPB + 15) / 16) * 16;
  static constexpr int NPIX = 16 * NW;
  static constexpr int REGA = RINGB;                           // the ring (streaming path and patch path alike)
  static constexpr int REGB = NW * OTB;                        // output images
  static constexpr int SMEM = REGA + REGB + 4 * NPIX * 4 + NW * 8 * 4;
  static constexpr int MAXROWS = 96;                           // tallest box that still streams (beyond: patches)
};

// one LDS-DMA wave-instruction without the M0 save / restore of lc::dma16_so: nothing else in this kernel uses M0
__device__ __forceinline__ void dma_row(const void* sbase, uint32_t voff, uint32_t lds_wave_base) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_wave_base) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

typedef __attribute__((address_space(3))) unsigned char lds_u8;

// min / max over each row of 16 lanes, result in every lane: four DPP steps on the VALU (quad xor 1, quad xor 2, half-row mirror, row
// mirror) instead of four ds_bpermute round trips through the LDS pipe
// lane i <- lane (i + 1) mod 16 inside each row of 16 lanes (DPP row_ror:15)
__device__ __forceinline__ float rot_left(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x12F, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ int row16_min(int v) {
  v = min(v, dpp_i<0xB1>(v)); v = min(v, dpp_i<0x4E>(v)); v = min(v, dpp_i<0x141>(v)); return min(v, dpp_i<0x140>(v));
}
__device__ __forceinline__ int row16_max(int v) {
  v = max(v, dpp_i<0xB1>(v)); v = max(v, dpp_i<0x4E>(v)); v = max(v, dpp_i<0x141>(v)); return max(v, dpp_i<0x140>(v));
}

template <typename T, int R, int NCB, int NW>
__global__ __launch_bounds__(64 * NW) void local_corr_rows_kernel(LCTileParams p) {
  using G = RowsGeom<R, NW>;
  constexpr int N1 = G::N1, K = G::K, TH = 2 * NW, NIW = G::NIW, NPIX = G::NPIX, SLOTB = G::SLOTB, BWMAX = G::BWMAX;
  constexpr int NS = G::NS, PF = G::PF, PPB = G::PPB, NBLK = G::NBLK;
  constexpr int UNR = NS * NCB;                                  // stages per unrolled block: slot and channel block are compile-time
  constexpr int BIG = 0x3fffffff;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ot_all = smem + G::REGA;
  int* s_x0 = reinterpret_cast<int*>(ot_all + G::REGB);          // [NPIX], index = group * 16 + pixel of the group
  int* s_y0 = s_x0 + NPIX;
  float* s_ax = reinterpret_cast<float*>(s_y0 + NPIX);
  float* s_ay = s_ax + NPIX;
  int* s_gbox = reinterpret_cast<int*>(s_ay + NPIX);             // [NW][8]: gx0, gx1, ga0, ga1, ok, empty

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kg = lane >> 4;
  const int H = p.H, W = p.W;
  // block id -> (batch item, tile) by float reciprocals (exact for grids below 2^21 tiles, checked by the host): an integer division
  // costs ~35 instructions each, and there were four of them per wave
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const int wid = xcd_remap(blockIdx.x, gridDim.x);
  const int b = (int)(((float)wid + 0.5f) * p.inv_tiles_per_img);
  const int tb = wid - b * tiles_per_img * p.bands;
  const int t = (int)(((float)tb + 0.5f) * p.inv_bands);
  const int band = __builtin_amdgcn_readfirstlane(tb - t * p.bands);   // this workgroup's share of the tile (see "bands" below)
  const int tyi = (int)(((float)t + 0.5f) * p.inv_tiles_x);
  const int ty0 = tyi * TH, tx0 = (t - tyi * p.tiles_x) * kTW;
  int b1 = b + p.f1_shift;
  b1 -= b1 >= p.B ? p.B : 0;
  const T* f0 = static_cast<const T*>(p.f0) + (size_t)b * H * W * p.f0_pitch;
  const T* f1 = static_cast<const T*>(p.f1) + (size_t)b1 * H * W * p.f1_pitch;
  T* out = static_cast<T*>(p.out);
  const uint32_t lds0 = (uint32_t)(size_t)(lds_u8*)smem;

  const int g = wave, gxi = g & 1, gyi = g >> 1;
  const int gpy = ty0 + gyi * 4, gpx = tx0 + gxi * 4;            // first pixel of this wave's 4x4 group
  const int y = gpy + (n16 >> 2), x = gpx + (n16 & 3);
  const bool pvalid = y < H && x < W;

  // ---- the flow of this lane's pixel is requested FIRST: the targets, the boxes and the first stages depend on it, the A operand is
  // not needed before the first MFMA — the wait below lets the NCB * 8 A loads stay in flight behind it (a one-round launch is a latency
  // chain; with the A loads in front every tile waited for them before it could even compute its box) ----
  float fx_raw = 0.f, fy_raw = 0.f;
  if (pvalid && p.flow) {
    const float* fb = p.flow + (size_t)b * 2 * H * W;
    fx_raw = fb[(uint32_t)(y * W + x)];
    fy_raw = fb[(uint32_t)(H * W + y * W + x)];
  }
  // ---- A operand: pixel n16 of the group, channels [64 kg, 64 kg + 64) of every 256-channel block, straight into registers ----
  u32x4 a[NCB * 8];
  {
    const T* pa = f0 + (uint32_t)((min(y, H - 1) * W + min(x, W - 1)) * p.f0_pitch + kg * 64);   // one map < 4 GB (host check)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int s = 0; s < 8; ++s) a[cb * 8 + s] = *reinterpret_cast<const u32x4*>(pa + cb * 256 + s * 8);
  }
  // the group's output image starts at zero (what no window position reaches stays zero)
  {
    u32x4* ot4 = reinterpret_cast<u32x4*>(ot_all + g * G::OTB);
    for (int i = lane; i < G::OTB / 16; i += 64) ot4[i] = u32x4{0, 0, 0, 0};
  }
  // ---- targets of the group's 16 pixels (replicated over the four lane quarters), group box by shuffles ----
  int x0 = 0, y0 = 0;
  float ax = 0.f, ay = 0.f;
  int lox = BIG, loy = BIG, hix = -BIG, hiy = -BIG;
  // the two flow loads are older than the A loads: wait until only those NCB * 8 are outstanding (vmcnt is in order)
  __builtin_amdgcn_s_waitcnt(((NCB * 8) & 15) | (((NCB * 8) >> 4) << 14) | 0x0F70);
  if (pvalid) {
    float fx, fy;
    if (p.flow) {
      fx = fx_raw;
      fy = fy_raw;
    } else {
      fx = -1.f + (2.f * x + 1.f) / W;
      fy = -1.f + (2.f * y + 1.f) / H;
    }
    flow_target(fx, fy, H, W, x0, y0, ax, ay);
    // a window that meets the image contributes its UNCLIPPED extent (the box is clipped below, keeping one virtual column / row)
    if (max(x0 - R, 0) <= min(x0 + R + 1, W - 1) && max(y0 - R, 0) <= min(y0 + R + 1, H - 1)) {
      lox = x0 - R; hix = x0 + R + 1; loy = y0 - R; hiy = y0 + R + 1;
    }
  }
  if (kg == 0) { s_x0[g * 16 + n16] = x0; s_y0[g * 16 + n16] = y0; s_ax[g * 16 + n16] = ax; s_ay[g * 16 + n16] = ay; }
  lox = row16_min(lox); loy = row16_min(loy); hix = row16_max(hix); hiy = row16_max(hiy);
  const bool gempty = __builtin_amdgcn_readfirstlane((int)(hix < lox));
  // columns: [max(lox, -1), min(hix, W-1)] (column -1 is virtual: staged from a clamped address, zeroed by the mask);
  // rows: [max(loy, 0), min(hiy, H-1)] are staged; row -1 is the zero start of the previous-row registers, row H one virtual step
  const int gx0 = __builtin_amdgcn_readfirstlane(gempty ? 0 : max(lox, -1));
  const int gx1 = __builtin_amdgcn_readfirstlane(gempty ? -1 : min(hix, W - 1));
  const int ga0 = __builtin_amdgcn_readfirstlane(gempty ? 0 : max(loy, 0));
  const int ga1 = __builtin_amdgcn_readfirstlane(gempty ? -1 : min(hiy, H - 1));
  const bool gvirt = __builtin_amdgcn_readfirstlane((int)(!gempty && hiy > H - 1));
  if (lane == 0) {
    int* gb = s_gbox + g * 8;
    gb[0] = gx0; gb[1] = gx1; gb[2] = ga0; gb[3] = ga1;
    gb[4] = gempty || gx1 - gx0 + 1 <= 16 * NBLK;
    gb[5] = gempty;
  }
  __syncthreads();
  int bx0 = BIG, bx1 = -BIG, ba0 = BIG, ba1 = -BIG;
  bool ok = true;
#pragma unroll
  for (int gg = 0; gg < NW; ++gg) {
    const int* gb = s_gbox + gg * 8;
    if (!gb[5]) { bx0 = min(bx0, gb[0]); bx1 = max(bx1, gb[1]); ba0 = min(ba0, gb[2]); ba1 = max(ba1, gb[3]); }
    ok = ok && gb[4];
  }
  const bool tempty = bx1 < bx0;
  bx0 = __builtin_amdgcn_readfirstlane(tempty ? 0 : bx0);
  ba0 = __builtin_amdgcn_readfirstlane(tempty ? 0 : ba0);
  const int bw = __builtin_amdgcn_readfirstlane(tempty ? 0 : bx1 - bx0 + 1);
  const int nrows = __builtin_amdgcn_readfirstlane(tempty ? 0 : max(ba1 - ba0 + 1, 0));
  ok = ok && bw <= BWMAX && nrows <= G::MAXROWS;
  // Bands: a launch with fewer tiles than the chip has room for (B = 2 at the 40 x 40 level: 50 tiles for 256 CUs, each streaming up to
  // the whole map) gets `bands` workgroups per tile from the launcher.  The emit positions j = 0 .. nrows (box row j blended with row
  // j - 1; j = nrows is the virtual row) are cut into runs of `per`; band k stages rows [k per - 1, (k + 1) per) and owns the output
  // rows whose emit position, clamped to [0, nrows], falls into its run: a partition, so every output element is written exactly once
  // and no workgroup waits for another (the first staged row of a band only fills the previous-row registers: what it emits lands in
  // a row of this workgroup's LDS image that another band owns and is never stored).  Patch path: the bands share out the pixels.
  const int per = (int)(((float)(nrows + p.bands) + 0.5f) * p.inv_bands);      // ceil((nrows + 1) / bands)
  const int c_lo = band * per;
  const int js = max(c_lo - 1, 0), je = min(c_lo + per, nrows);                // staged box rows [js, je)

  if constexpr (G::kPatchPath) {
    if (!__builtin_amdgcn_readfirstlane((int)ok)) {
      // =========================== incoherent tile: every pixel gathers its own (2r+2)^2 patch ===========================
      // WAVE-PRIVATE streaming: wave g walks the 16 pixels of its own group, one after the other; a stage is ONE ROW of the pixel's
      // patch (2r+2 positions x 256 channels, every position a whole 512-byte request) DMA'd by this wave alone into its own 2-3
      // slot ring, so there is no workgroup barrier anywhere — the four waves drift apart freely and a wave's MFMAs overlap the
      // other waves' DMA waits.  The 16 x 16 MFMA multiplies the row's positions with all 16 pixels of the group (A is already in
      // registers); the pixel's own result row goes into a 256-byte LDS patch image, and after the patch's last row the wave blends
      // and stores the K outputs.  Pixels outside the image or whose window misses the map are not staged (outputs zero).
      // (The 32-channel-chunk kernels gather 64-byte pieces, 10-12 TB/s at best — tools/stage_micro.hip; a first version of this
      // path with workgroup-wide stages and one owner wave per stage was barrier-bound and slower than they are.)
      constexpr int N2 = 2 * R + 2;
      constexpr int NIP = N2 / 2;                                 // DMA wave-instructions per patch row (two positions each)
      constexpr int SLOTW = N2 * 512;                             // one patch row of one channel block
      constexpr int NSW = R <= 2 ? 3 : 2;                         // wave-private ring slots; NSW - 1 rows in flight beside the one in use
      constexpr int PFW = NSW - 1;
      constexpr int UNRW = (NSW % NCB == 0) ? NSW : NSW * NCB;    // unroll: slot and channel block compile-time
      static_assert(NSW * SLOTW * NW <= G::REGA && K <= 64 && G::OTB >= N2 * N2 * 4, "patch path geometry");
      // this wave's pixels: record m = pixel of the group (lane m holds it; lanes >= 16 replicate)
      const int mrec = lane & 15;
      const int qx0 = s_x0[g * 16 + mrec], qy0 = s_y0[g * 16 + mrec];
      const float qax = s_ax[g * 16 + mrec], qay = s_ay[g * 16 + mrec];
      const bool hit = gpy + (mrec >> 2) < H && gpx + (mrec & 3) < W && max(qx0 - R, 0) <= min(qx0 + R + 1, W - 1) &&
                       max(qy0 - R, 0) <= min(qy0 + R + 1, H - 1);
      uint32_t own = 0xffffu;                                     // this band's pixels of the group: m = band (mod bands)
      if (p.bands > 1) {
        own = 0;
        for (int mm = band; mm < 16; mm += p.bands) own |= 1u << mm;
      }
      const uint32_t act = (uint32_t)(__ballot(hit) & 0xffffull) & own;
      T* ob = p.out_nhwc ? out + (((size_t)b * H + gpy) * W + gpx) * p.out_pitch : out + ((size_t)b * p.out_pitch * H + gpy) * W + gpx;
      const uint32_t plane = (uint32_t)(H * W);
      auto out_off = [&](int mm, int kk) -> uint32_t {
        const int dy = mm >> 2, dx = mm & 3;
        return p.out_nhwc ? (uint32_t)((dy * W + dx) * p.out_pitch + kk) : (uint32_t)kk * plane + (uint32_t)(dy * W + dx);
      };
      for (int mm = 0; mm < 16; ++mm) {                           // zeros for pixels inside the image that gather nothing
        if (((act >> mm) & 1u) || !((own >> mm) & 1u) || gpy + (mm >> 2) >= H || gpx + (mm & 3) >= W) continue;
        if (lane < K) ob[out_off(mm, lane)] = from_f32<T>(0.f);
      }
      const int nstw = __builtin_popcount(act) * N2 * NCB;
      // DMA plan of one patch row: instruction k covers positions 2k, 2k + 1
      uint32_t coff[NIP], vconst[NIP];
      int dxk[NIP];
#pragma unroll
      for (int k = 0; k < NIP; ++k) {
        const int pos = 2 * k + (lane >> 5), qp = lane & 31;
        dxk[k] = pos;
        coff[k] = (uint32_t)(((qp & 24) | ((qp ^ pos) & 7)) * 16);
        vconst[k] = (uint32_t)(pos * p.f1_pitch) * 2u + coff[k];
      }
      const uint32_t ring0 = lds0 + (uint32_t)(wave * NSW * SLOTW);
      uint32_t bo2[8];
      {
        const int pos = min(n16, N2 - 1);
        const uint32_t base = ring0 + (uint32_t)(pos * 512 + kg * 128), c = (uint32_t)((pos & 7) << 4);
#pragma unroll
        for (int s = 0; s < 8; ++s) bo2[s] = base + (c ^ (uint32_t)(s << 4));
      }
      float* pd = reinterpret_cast<float*>(ot_all + g * G::OTB);  // this wave's patch image: Q products of the pixel in hand

      // issue side and compute side each walk (pixel, patch row); the pixel cursor is a bit mask
      uint32_t mi = act, mc = act;
      int rowi = 0, rowc = 0;
      auto issue = [&](auto slot_c, auto cb_c) {
        constexpr int SLOT = decltype(slot_c)::value, CB = decltype(cb_c)::value;
        const int mm = __builtin_ctz(mi);
        const int x0j = __builtin_amdgcn_readlane(qx0, mm) - R, yj = __builtin_amdgcn_readlane(qy0, mm) - R + rowi;
        const uint32_t dst = ring0 + (uint32_t)(SLOT * SLOTW);
        if (x0j >= 0 && x0j + N2 - 1 <= W - 1 && yj >= 0 && yj <= H - 1) {                  // wave-uniform: the common case
          const T* sb = f1 + CB * 256 + (size_t)(yj * W + x0j) * p.f1_pitch;                 // the row's first position rides on the scalar base
#pragma unroll
          for (int k = 0; k < NIP; ++k) dma_row(sb, vconst[k], dst + (uint32_t)(k * 1024));
        } else {
          const int yy = min(max(yj, 0), H - 1);
#pragma unroll
          for (int k = 0; k < NIP; ++k) {
            const int xx = min(max(x0j + dxk[k], 0), W - 1);
            dma_row(f1 + CB * 256, (uint32_t)((yy * W + xx) * p.f1_pitch) * 2u + coff[k], dst + (uint32_t)(k * 1024));
          }
        }
        if (CB == NCB - 1 && ++rowi == N2) { rowi = 0; mi &= mi - 1; }
      };
      float4_t acc2 = {0.f, 0.f, 0.f, 0.f};
      if (nstw > 0) issue(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      if (PFW >= 2 && nstw > 1) issue(std::integral_constant<int, 1 % NSW>{}, std::integral_constant<int, 1 % NCB>{});
      // every global load of the prologue (the A operand) and the stages just issued have landed; said with the builtin so that hipcc
      // KNOWS its vmcnt scoreboard is empty and puts no wait of its own into the loop (it cannot count the asm LDS-DMAs: a vmcnt(0) it
      // placed in front of an MFMA that reads `a` drained the ring every stage)
      __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0), expcnt / lgkmcnt untouched
      for (int st0 = 0; st0 < nstw; st0 += UNRW) {
#pragma unroll
        for (int u = 0; u < UNRW; ++u) {
          const int st = st0 + u;
          if (st < nstw) {
            if (st + PFW < nstw) {
              if (u == 0) issue(std::integral_constant<int, (0 + PFW) % NSW>{}, std::integral_constant<int, (0 + PFW) % NCB>{});
              if (u == 1) issue(std::integral_constant<int, (1 + PFW) % NSW>{}, std::integral_constant<int, (1 + PFW) % NCB>{});
              if (u == 2) issue(std::integral_constant<int, (2 + PFW) % NSW>{}, std::integral_constant<int, (2 + PFW) % NCB>{});
              if (u == 3) issue(std::integral_constant<int, (3 + PFW) % NSW>{}, std::integral_constant<int, (3 + PFW) % NCB>{});
              if (u == 4) issue(std::integral_constant<int, (4 + PFW) % NSW>{}, std::integral_constant<int, (4 + PFW) % NCB>{});
              if (u == 5) issue(std::integral_constant<int, (5 + PFW) % NSW>{}, std::integral_constant<int, (5 + PFW) % NCB>{});
              vm_wait<PFW * NIP>();
            } else if (PFW >= 2 && st + 1 < nstw) {
              vm_wait<(PFW - 1) * NIP>();
            } else {
              vm_wait<0>();
            }
            const int cb = u % NCB;
            if (cb == 0) acc2 = float4_t{0.f, 0.f, 0.f, 0.f};
            u32x4 bq[8];
#pragma unroll
            for (int s = 0; s < 8; ++s)
              bq[s] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>((lds_u8*)(size_t)(bo2[s]) + (u % NSW) * SLOTW);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 8; ++s) acc2 = mfma16r(a[cb * 8 + s], bq[s], acc2, T{});
            if (cb == NCB - 1) {
              // result row jm (the pixel in hand): lanes of quarter jm / 4, accumulator element jm % 4; column n16 = position in the row
              const int jm = __builtin_ctz(mc);
              const int x0j = __builtin_amdgcn_readlane(qx0, jm) - R, yj = __builtin_amdgcn_readlane(qy0, jm) - R + rowc;
              const int sel = jm & 3;
              float v = sel == 0 ? acc2[0] : sel == 1 ? acc2[1] : sel == 2 ? acc2[2] : acc2[3];
              const int xx = x0j + n16;
              if (!(yj >= 0 && yj < H && xx >= 0 && xx < W)) v = 0.f;
              if (kg == (jm >> 2) && n16 < N2) pd[rowc * N2 + n16] = v * p.scale;
              if (++rowc == N2) {
                rowc = 0;
                mc &= mc - 1;
                const float axj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qax), jm));
                const float ayj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qay), jm));
                if (lane < K) {
                  const int iy = lane / N1, ix = lane - iy * N1;
                  const float* d = pd + iy * N2 + ix;
                  const float top = d[0] + axj * (d[1] - d[0]);
                  const float bot = d[N2] + axj * (d[N2 + 1] - d[N2]);
                  ob[out_off(jm, lane)] = from_f32<T>(top + ayj * (bot - top));
                }
              }
            }
          }
        }
      }
      return;
    }
  }
  (void)ok;                                                      // r >= 4: the host admits only maps every box of which fits (W <= BWMAX - 1)

  // =========================== streaming path ===========================
  const int nst = max(je - js, 0) * NCB;
  // DMA plan: wave-instruction ii = k NW + wave of a stage covers LDS bytes [1024 ii, 1024 ii + 1024) = pixels 2 ii, 2 ii + 1
  uint32_t voff[NIW];
  const int ni = (bw + 1) >> 1;
  const int n_w = __builtin_amdgcn_readfirstlane(ni > wave ? (ni - wave + NW - 1) / NW : 0);
#pragma unroll
  for (int k = 0; k < NIW; ++k) {
    const int P = (k * NW + wave) * 64 + lane;
    const int pxb = P >> 5, qp = P & 31;
    const int q = (qp & 24) | ((qp ^ pxb) & 7);                  // the slot holds channel piece q of its 256-channel block
    const int xs = min(max(bx0 + min(pxb, max(bw - 1, 0)), 0), W - 1);
    voff[k] = (uint32_t)(xs * p.f1_pitch + q * 8) * 2u;
  }
  // B fragment addresses: lane (n16, kg) reads piece s of quarter kg of staged pixel col0 + n16
  uint32_t bo[NBLK][8];
#pragma unroll
  for (int k = 0; k < NBLK; ++k) {
    const int pxb = min(max(gx0 - bx0 + 16 * k + n16, 0), BWMAX - 1);
    const uint32_t base = lds0 + (uint32_t)(pxb * 512 + kg * 128), c = (uint32_t)((pxb & 7) << 4);
#pragma unroll
    for (int s = 0; s < 8; ++s) bo[k][s] = base + (c ^ (uint32_t)(s << 4));
  }
  // per accumulator row r4 (pixel 4 kg + r4 of the group): blend weights (the C^-1/2 scale folded into the y weights), the LDS
  // address of row 0 of the image column this lane produces for that pixel, and (in bytes of image rows) where its window starts
  float axv[4], ay0[4], ay1[4];
  uint32_t wbase[NBLK][4];
  int wyb[4];
  constexpr int ROWB = N1 * 2;                                   // bytes per image row
  const uint32_t otb = lds0 + (uint32_t)(G::REGA + g * G::OTB);
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const int pp = 4 * kg + r4, r = g * 16 + pp;
    axv[r4] = s_ax[r];
    const float ayp = s_ay[r];
    ay1[r4] = ayp * p.scale;
    ay0[r4] = p.scale - ay1[r4];
#pragma unroll
    for (int k = 0; k < NBLK; ++k) {
      const int ix = gx0 + 16 * k + n16 - (s_x0[r] - R);         // window column this lane produces for that pixel in block k
      const bool colvalid = (unsigned)ix < (unsigned)N1;
      wbase[k][r4] = otb + (uint32_t)(colvalid ? pp * PPB + ix * 2 : 16 * PPB);
    }
    wyb[r4] = (s_y0[r] - R - ba0) * ROWB;                        // box row ya = ba0 + j is image row j - wyb / ROWB (window row + 1)
  }
  bool dvalid[NBLK];
#pragma unroll
  for (int k = 0; k < NBLK; ++k) { const int xa = gx0 + 16 * k + n16; dvalid[k] = xa >= 0 && xa <= W - 1; }
  const int gw = gx1 - gx0 + 1;
  const int nblk_g = __builtin_amdgcn_readfirstlane(NBLK == 1 ? 1 : (gw + 15) >> 4);   // column blocks this group needs
  const bool inner = __builtin_amdgcn_readfirstlane((int)(gx0 >= 0 && gx0 + 16 * nblk_g - 1 <= W - 1));
  const bool last_col = n16 == 15;
  const int gs0 = ga0 - ba0, gs1 = ga1 - ba0;
  float tprev[NBLK][4];
  float4_t acc[NBLK];
#pragma unroll
  for (int k = 0; k < NBLK; ++k) {
    acc[k] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) tprev[k][r4] = 0.f;
  }

  // box row ya = ba0 + j, jb = j ROWB: x-blend (DPP: the right-hand neighbour column; across a block boundary it is lane 0 of the
  // next block), y-blend against the previous row, store at the clamped image row
  auto emit_row = [&](const float4_t* d, int jb) {
    int rowb[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) rowb[r4] = min(max(jb - wyb[r4], 0), (N1 + 1) * ROWB);
    float dm[NBLK + 1][4];
#pragma unroll
    for (int k = 0; k < NBLK; ++k)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        float v = d[k][r4];
        if (NBLK > 1 && k >= nblk_g) v = 0.f;
        else if (!inner) v = dvalid[k] ? v : 0.f;
        dm[k][r4] = v;
      }
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) dm[NBLK][r4] = 0.f;
#pragma unroll
    for (int k = 0; k < NBLK; ++k) {
      if (NBLK > 1 && k >= nblk_g) break;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const float dcur = dm[k][r4];
        float rgt;
        if (NBLK == 1) {
          rgt = right_neighbour(dcur);
        } else {
          const float a0 = rot_left(dcur), a1 = rot_left(dm[k + 1][r4]);
          rgt = last_col ? a1 : a0;
        }
        const float tc = dcur + axv[r4] * (rgt - dcur);
        const float v = ay0[r4] * tprev[k][r4] + ay1[r4] * tc;
        tprev[k][r4] = tc;
        *reinterpret_cast<__attribute__((address_space(3))) T*>((lds_u8*)(size_t)(wbase[k][r4] + (uint32_t)rowb[r4])) = from_f32<T>(v);
      }
    }
  };

  auto run = [&](auto nwi_c) {
    constexpr int NWI = decltype(nwi_c)::value;
    const T* rb = f1 + (size_t)(ba0 + js) * W * p.f1_pitch;      // next stage to issue: row pointer (channel block 0)
    const size_t rstride = (size_t)W * p.f1_pitch;
    const uint32_t dst0 = lds0 + (uint32_t)(wave * 1024);
    auto issue = [&](auto slot_c, auto cb_c) {
      constexpr int SLOT = decltype(slot_c)::value, CB = decltype(cb_c)::value;
#pragma unroll
      for (int k = 0; k < NWI; ++k) dma_row(rb + CB * 256, voff[k], dst0 + (uint32_t)(SLOT * SLOTB + k * NW * 1024));
      if (CB == NCB - 1) rb += rstride;
    };
    // fill: the first PF stages
    if (PF >= 1 && nst > 0) issue(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    if (PF >= 2 && nst > 1) issue(std::integral_constant<int, 1 % NS>{}, std::integral_constant<int, 1 % NCB>{});
    // every global load of the prologue (the A operand) and the stages just issued have landed; said with the builtin so that hipcc
    // KNOWS its vmcnt scoreboard is empty and puts no wait of its own into the loop (it cannot count the asm LDS-DMAs: a vmcnt(0) it
    // placed in front of an MFMA that reads `a` drained the ring every stage)
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0), expcnt / lgkmcnt untouched
    for (int st0 = 0; st0 < nst; st0 += UNR) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int st = st0 + u;
        if (st < nst) {
          if (st + PF < nst) {
            if (u == 0) issue(std::integral_constant<int, (0 + PF) % NS>{}, std::integral_constant<int, (0 + PF) % NCB>{});
            if (u == 1) issue(std::integral_constant<int, (1 + PF) % NS>{}, std::integral_constant<int, (1 + PF) % NCB>{});
            if (u == 2) issue(std::integral_constant<int, (2 + PF) % NS>{}, std::integral_constant<int, (2 + PF) % NCB>{});
            if (u == 3) issue(std::integral_constant<int, (3 + PF) % NS>{}, std::integral_constant<int, (3 + PF) % NCB>{});
            if (u == 4) issue(std::integral_constant<int, (4 + PF) % NS>{}, std::integral_constant<int, (4 + PF) % NCB>{});
            if (u == 5) issue(std::integral_constant<int, (5 + PF) % NS>{}, std::integral_constant<int, (5 + PF) % NCB>{});
            if (u == 6) issue(std::integral_constant<int, (6 + PF) % NS>{}, std::integral_constant<int, (6 + PF) % NCB>{});
            if (u == 7) issue(std::integral_constant<int, (7 + PF) % NS>{}, std::integral_constant<int, (7 + PF) % NCB>{});
            vm_wait<PF * NWI>();
          } else if (PF >= 2 && st + 1 < nst) {
            vm_wait<(PF - 1) * NWI>();
          } else {
            vm_wait<0>();
          }
          raw_barrier();
          const int cb = u % NCB;
          const int jr = js + (NCB == 1 ? st : st >> 1);
          if (jr >= gs0 && jr <= gs1) {
            // B fragments of two blocks at a time: sixteen reads in flight before the first MFMA waits
#pragma unroll
            for (int k0 = 0; k0 < NBLK; k0 += 2) {
              if (NBLK > 1 && k0 >= nblk_g) break;
              constexpr int KP = NBLK > 1 ? 2 : 1;
              u32x4 bq[KP][8];
#pragma unroll
              for (int kk = 0; kk < KP; ++kk) {
                const int k = k0 + kk < NBLK ? k0 + kk : NBLK - 1;
                if (k0 + kk >= NBLK || (NBLK > 1 && k >= nblk_g)) continue;
#pragma unroll
                for (int s = 0; s < 8; ++s)
                  bq[kk][s] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>((lds_u8*)(size_t)(bo[k][s]) + (u % NS) * SLOTB);
              }
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int kk = 0; kk < KP; ++kk) {
                const int k = k0 + kk < NBLK ? k0 + kk : NBLK - 1;
                if (k0 + kk >= NBLK || (NBLK > 1 && k >= nblk_g)) continue;
                if (cb == 0) acc[k] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 8; ++s) acc[k] = mfma16r(a[cb * 8 + s], bq[kk][s], acc[k], T{});
              }
            }
            if (cb == NCB - 1) emit_row(acc, jr * ROWB);
          }
        }
      }
    }
  };
  if constexpr (NIW <= 3) {
    switch (n_w) {
      case 0: run(std::integral_constant<int, 0>{}); break;
      case 1: run(std::integral_constant<int, 1>{}); break;
      case 2: if constexpr (NIW >= 2) run(std::integral_constant<int, 2>{}); break;
      default: if constexpr (NIW >= 3) run(std::integral_constant<int, 3>{}); break;
    }
  } else {
    // wide boxes: two instantiations; a wave issues up to one DMA more than its share (the surplus re-reads the box's last pixels: L1 hits)
    if (n_w * 2 <= NIW) run(std::integral_constant<int, (NIW + 1) / 2>{});
    else run(std::integral_constant<int, NIW>{});
  }
  if (gvirt && c_lo <= nrows && nrows < c_lo + per) {            // the virtual row below the image (the band that owns position nrows)
    float4_t z[NBLK];
#pragma unroll
    for (int k = 0; k < NBLK; ++k) z[k] = float4_t{0.f, 0.f, 0.f, 0.f};
    emit_row(z, (H - ba0) * ROWB);
  }

  // ---- write the group's output image: wave-private, no barrier; 32-bit element offsets from a wave-uniform base ----
  const T* ot = reinterpret_cast<const T*>(ot_all + g * G::OTB);
  if (p.bands > 1) {
    // this band's rows of every pixel: image rows i in [1, N1] whose emit position clamp(i + wy, 0, nrows) lies in [c_lo, c_lo + per)
    if (c_lo > nrows) return;
    const uint32_t plane = (uint32_t)(H * W);
    T* ob = p.out_nhwc ? out + (((size_t)b * H + gpy) * W + gpx) * p.out_pitch : out + ((size_t)b * p.out_pitch * H + gpy) * W + gpx;
    for (int m = 0; m < 16; ++m) {
      const int dy = m >> 2, dx = m & 3;
      if (gpy + dy >= H || gpx + dx >= W) continue;
      const int wy = __builtin_amdgcn_readfirstlane(s_y0[g * 16 + m]) - R - ba0;
      const int i_lo = c_lo <= 0 ? 1 : max(1, c_lo - wy);
      const int i_hi = c_lo + per > nrows ? N1 : min(N1, c_lo + per - 1 - wy);
      const int kk0 = (i_lo - 1) * N1, cnt = (i_hi - i_lo + 1) * N1;
      const T* om = ot + m * (PPB / 2) + N1 + kk0;
      if (p.out_nhwc) {
        T* o = ob + (uint32_t)((dy * W + dx) * p.out_pitch + kk0);
        for (int e = lane; e < cnt; e += 64) o[e] = om[e];
      } else {
        T* o = ob + (uint32_t)kk0 * plane + (uint32_t)(dy * W + dx);
        for (int e = lane; e < cnt; e += 64) o[(uint32_t)e * plane] = om[e];
      }
    }
    return;
  }
  if (p.out_nhwc) {
    T* ob = out + (((size_t)b * H + gpy) * W + gpx) * p.out_pitch;
    for (int e = lane; e < 16 * K; e += 64) {
      const int m = e / K, kk = e - m * K;
      const int dy = m >> 2, dx = m & 3;
      if (gpy + dy < H && gpx + dx < W) ob[(uint32_t)((dy * W + dx) * p.out_pitch + kk)] = ot[m * (PPB / 2) + N1 + kk];
    }
  } else {
    T* ob = out + ((size_t)b * p.out_pitch * H + gpy) * W + gpx;
    const uint32_t plane = (uint32_t)(H * W);
    for (int e = lane; e < 16 * K; e += 64) {
      const int kk = e >> 4, m = e & 15;
      const int dy = m >> 2, dx = m & 3;
      if (gpy + dy < H && gpx + dx < W) ob[(uint32_t)kk * plane + (uint32_t)(dy * W + dx)] = ot[m * (PPB / 2) + N1 + kk];
    }
  }
}

template <typename T, int R, int NCB, int NW>
int launch_rows(LCTileParams p, hipStream_t stream) {
  using G = RowsGeom<R, NW>;
  constexpr size_t smem = G::SMEM;
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = ensure_dyn_smem(reinterpret_cast<const void*>(local_corr_rows_kernel<T, R, NCB, NW>), (int)smem, attr_done, "roma_local_corr")) return rc;
  p.tiles_x = (p.W + kTW - 1) / kTW;
  p.tiles_y = (p.H + 2 * NW - 1) / (2 * NW);
  // bands (see the kernel): launches that leave most of the chip idle split every tile's box rows over several workgroups
  const int tiles = p.B * p.tiles_x * p.tiles_y;
  const int room = R >= 4 ? 256 : 512, most = R >= 4 ? 8 : 4;
  p.bands = tiles > 0 && tiles < room ? std::min(std::max(room / tiles, 1), most) : 1;
  if (const char* e = getenv("ROMA_LC_BANDS")) { const int v = atoi(e); if (v >= 1 && v <= 16) p.bands = v; }
  p.inv_bands = 1.0f / (float)p.bands;
  p.inv_tiles_per_img = 1.0f / (float)(p.tiles_x * p.tiles_y * p.bands);
  p.inv_tiles_x = 1.0f / (float)p.tiles_x;
  const int grid = tiles * p.bands;
  ROMA_REQUIRE(grid < (1 << 21), ROMA_E_SHAPE, "roma_local_corr: %d tiles in one launch (row-streaming kernel: < 2^21)", grid);
  hipLaunchKernelGGL((local_corr_rows_kernel<T, R, NCB, NW>), dim3(grid), dim3(64 * NW), smem, stream, p);
  ROMA_CHECK_LAUNCH();
}

template <typename T, int R>
int launch_rows_c(const LCTileParams& p, hipStream_t stream) {
  if (p.C == 256) return launch_rows<T, R, 1, 4>(p, stream);
  if (p.C == 512) return launch_rows<T, R, 2, 4>(p, stream);
  set_error("local_corr_rows: C = %d (supported: 256, 512)", p.C);
  return ROMA_E_UNSUPPORTED;
}

}  // namespace

// r <= 3: any map; r = 7 (the scale-16 level; the only r >= 4 instantiated): maps small enough that a tile's box may be the whole map
bool local_corr_rows_supports(int C, int r, int H, int W) {
  if (!(C == 256 || C == 512)) return false;
  if (r <= 3) return true;
  return r == 7 && W <= RowsGeom<7, 4>::BWMAX - 1 && H <= RowsGeom<7, 4>::MAXROWS;
}

int local_corr_rows(const LCTileParams& p, int r, int dtype, hipStream_t stream) {
  if (dtype == ROMA_F16) {
    switch (r) {
      case 1: return launch_rows_c<half_t, 1>(p, stream);
      case 2: return launch_rows_c<half_t, 2>(p, stream);
      case 3: return launch_rows_c<half_t, 3>(p, stream);
      case 7: return launch_rows_c<half_t, 7>(p, stream);
    }
  } else if (dtype == ROMA_BF16) {
    switch (r) {
      case 1: return launch_rows_c<bf16_t, 1>(p, stream);
      case 2: return launch_rows_c<bf16_t, 2>(p, stream);
      case 3: return launch_rows_c<bf16_t, 3>(p, stream);
      case 7: return launch_rows_c<bf16_t, 7>(p, stream);
    }
  }
  set_error("local_corr_rows: unsupported r=%d dtype=%d", r, dtype);
  return ROMA_E_UNSUPPORTED;
}

}  // namespace roma
