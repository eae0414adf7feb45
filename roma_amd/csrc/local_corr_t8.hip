// local_correlation, fp16 / bf16 channels-last, r <= 3: ONE 8x8-pixel tile per 256-thread workgroup.
// Reference: romatch/utils/local_correlation.py:4-48 (called at matcher.py:121-125); formulation as in local_corr.hip.
//
// What round 2's measurements say about this kernel family:
//   * the 8x4-tile kernel of local_corr.hip is bound by per-tile FIXED costs (flow -> box round trips, plan, ~16 barriers), hidden
//     only by 4-5 co-resident workgroups; a 64-pixel tile halves those costs per pixel and stages 3.9 + 1 instead of 5.2 + 1 rows
//     per pixel;
//   * the persistent loader / consumer ring (local_corr_ring.hip, removed in round 3) removed the DMA wait but has ONE workgroup per CU, so nothing
//     overlaps its bookkeeping: slower.  Its consumer half is what pays: wave g = 4x4 group g, all B fragments read up front,
//     back-to-back MFMAs, and the 4-tap blend in registers (DPP shift for the right-hand tap, the next accumulator block for the
//     lower one) instead of a zero-initialised (2r+2)^2 image per pixel.
// This kernel is that consumer half inside the one-tile-per-workgroup skeleton (3-4 workgroups per CU for latency hiding).
// Tiles whose targets are not compact take per-pixel patches on the VALU (same staging), like the other kernels.
#include "common.h"
#include "lc_device.h"
#include "lc_variants.h"

namespace roma {
namespace {

using namespace lc;

template <int R> struct T8Geom {
  static constexpr int N2 = 2 * R + 2;
  static constexpr int MAXR = R <= 2 ? 320 : 384;              // staged box rows
  static constexpr int GHMAX = N2 + 8;                         // union rows a 4x4 group may span (one 16-column run)
  static constexpr int ROWS = kTP + MAXR + 16 + 16;            // f0 rows, box, slack of a 16-row B read, 16 zero rows
};

template <typename T, int R>
__global__ __launch_bounds__(256, R <= 2 ? 4 : 3) void local_corr_t8_kernel(LCTileParams p) {
  using G = T8Geom<R>;
  constexpr int N1 = 2 * R + 1, N2 = G::N2, Q = N2 * N2, K = N1 * N1;
  constexpr int MAXR = G::MAXR, GHMAX = G::GHMAX, NB = GHMAX;
  constexpr int NL = ((kTP + MAXR) * 4 + 255) / 256;            // DMA rounds per chunk (64 rows each)
  constexpr int E16 = 8, CC = 32;
  constexpr int ZROW = kTP + MAXR + 16;                        // first all-zero row
  // slow path: SB pixels per pass get a private patch of QP rows each
  // (patches are packed back to back, Q rows each: the swizzled b128 reads are conflict-free for any row base)
  constexpr int NIT = (Q + 15) / 16, QP = Q;
  constexpr int SB = (MAXR / QP) >= 8 ? 8 : 4;                  // 8 pixels per pass at r <= 2, 4 at r = 3 (as the 8x4-tile kernel)
  constexpr int UB = (SB * NIT + 15) / 16;
  constexpr int SNL = ((kTP + SB * QP) * 4 + 255) / 256;
  static_assert(SB * QP <= MAXR, "patches must fit the staged area");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* rows = reinterpret_cast<u32x4*>(smem);
  float* s_D = reinterpret_cast<float*>(rows + G::ROWS * 4);   // [kTP][Q+1]
  int* s_x0 = reinterpret_cast<int*>(s_D + kTP * (Q + 1));     // group-major row index
  int* s_y0 = s_x0 + kTP;
  float* s_ax = reinterpret_cast<float*>(s_y0 + kTP);
  float* s_ay = s_ax + kTP;
  int* s_gbox = reinterpret_cast<int*>(s_ay + kTP);            // [kNG][8]: x0,y0,x1,y1,ok,inner

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, W = p.W;
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const int wid = xcd_remap(blockIdx.x, gridDim.x);
  const int b = wid / tiles_per_img;
  const int t = wid - b * tiles_per_img;
  const int ty0 = (t / p.tiles_x) * kTH, tx0 = (t % p.tiles_x) * kTW;
  const T* f0 = static_cast<const T*>(p.f0) + (size_t)b * H * W * p.f0_pitch;
  const T* f1 = static_cast<const T*>(p.f1) + (size_t)((b + p.f1_shift) % p.B) * H * W * p.f1_pitch;
  T* out = static_cast<T*>(p.out);
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  for (int i = tid; i < 64; i += 256) rows[ZROW * 4 + i] = u32x4{0, 0, 0, 0};
  // ---- targets and group boxes: wave 0, lanes in group-major order ----
  if (wave == 0) {
    const int g = lane >> 4, m = lane & 15;
    const int py = (g >> 1) * 4 + (m >> 2), px = (g & 1) * 4 + (m & 3);
    const int y = ty0 + py, x = tx0 + px;
    int x0 = 0, y0 = 0;
    float ax = 0.f, ay = 0.f;
    int lox = 0x3fffffff, loy = 0x3fffffff, hix = -0x3fffffff, hiy = -0x3fffffff;
    bool inner = false;
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
      const int a0 = max(x0 - R, 0), a1 = min(x0 + R + 1, W - 1);
      const int b0 = max(y0 - R, 0), b1 = min(y0 + R + 1, H - 1);
      if (a0 <= a1 && b0 <= b1) { lox = a0; hix = a1; loy = b0; hiy = b1; }
      inner = (x0 - R >= 0) && (y0 - R >= 0) && (x0 + R + 1 <= W - 1) && (y0 + R + 1 <= H - 1);
    }
    s_x0[lane] = x0; s_y0[lane] = y0; s_ax[lane] = ax; s_ay[lane] = ay;
    // all 16 pixels of the group inner?  (ballot over the wave, then the group's 16 bits)
    const unsigned long long bal = __ballot(inner);
    const bool all_inner = ((bal >> (g * 16)) & 0xffffull) == 0xffffull;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      lox = min(lox, __shfl_xor(lox, o, 64)); loy = min(loy, __shfl_xor(loy, o, 64));
      hix = max(hix, __shfl_xor(hix, o, 64)); hiy = max(hiy, __shfl_xor(hiy, o, 64));
    }
    if (m == 0) {
      int* gb = s_gbox + g * 8;
      const bool empty = hix < lox || hiy < loy;
      gb[0] = lox; gb[1] = loy; gb[2] = hix; gb[3] = hiy;
      gb[4] = empty || ((hix - lox + 1 <= 16) && (hiy - loy + 1 <= GHMAX));
      gb[5] = all_inner;
    }
  }
  __syncthreads();
  int bx0, by0, bw, bh;
  bool ok = true;
  {
    int lx = 0x3fffffff, ly = 0x3fffffff, hx = -0x3fffffff, hy = -0x3fffffff;
#pragma unroll
    for (int g = 0; g < kNG; ++g) {
      const int* gb = s_gbox + g * 8;
      if (gb[2] >= gb[0] && gb[3] >= gb[1]) { lx = min(lx, gb[0]); ly = min(ly, gb[1]); hx = max(hx, gb[2]); hy = max(hy, gb[3]); }
      ok = ok && gb[4];
    }
    const bool empty = hx < lx || hy < ly;
    bx0 = __builtin_amdgcn_readfirstlane(empty ? 0 : lx);
    by0 = __builtin_amdgcn_readfirstlane(empty ? 0 : ly);
    bw = __builtin_amdgcn_readfirstlane(empty ? 0 : hx - lx + 1);
    bh = __builtin_amdgcn_readfirstlane(empty ? 0 : hy - ly + 1);
    ok = ok && (bw * bh <= MAXR);
  }

  if (__builtin_amdgcn_readfirstlane((int)ok)) {
    // =========================== matrix-core path ===========================
    const int nrows = bw * bh;
    const int nl_used = __builtin_amdgcn_readfirstlane((kTP + nrows + 63) >> 6);
    // DMA source plan: per-lane BYTE offsets from the (batch item's) f0 / f1 base — 32 bits (the host checks one map is < 4 GB),
    // so 6 registers instead of 12, and the per-chunk channel advance is a scalar add on the base
    uint32_t soff[NL];
    {
      const int prow = tid >> 2;
      const int kk8 = (((tid & 3) ^ ((tid >> 4) & 1))) * E16;
      int fpy, fpx;
      fpix(prow, fpy, fpx);
      soff[0] = (uint32_t)(((size_t)min(ty0 + fpy, H - 1) * W + min(tx0 + fpx, W - 1)) * p.f0_pitch + kk8) * (uint32_t)sizeof(T);
      const float inv_bw = 1.0f / (float)max(bw, 1);
#pragma unroll
      for (int l = 1; l < NL; ++l) {
        const int rr = (l - 1) * 64 + prow;
        const int ry = min((int)(((float)rr + 0.5f) * inv_bw), max(bh - 1, 0));     // rr / bw, exact for these small integers
        const int y = by0 + ry, x = bx0 + rr - ry * bw;
        soff[l] = rr < nrows ? (uint32_t)(((size_t)y * W + x) * p.f1_pitch + kk8) * (uint32_t)sizeof(T) : 0u;   // slots past the box read a valid dummy
      }
    }
    const int g = wave;
    const int n16 = lane & 15, kg = lane >> 4;
    const int ks = (kg == 1) ? 2 : (kg == 2 ? 1 : kg);
    const int* gb = s_gbox + g * 8;
    const bool gempty = gb[2] < gb[0] || gb[3] < gb[1];
    const int gx0 = __builtin_amdgcn_readfirstlane(gb[0]), gy0 = __builtin_amdgcn_readfirstlane(gb[1]);
    const int gw = __builtin_amdgcn_readfirstlane(gempty ? 0 : gb[2] - gb[0] + 1);
    const int nblk = __builtin_amdgcn_readfirstlane(gempty ? 0 : gb[3] - gb[1] + 1);
    const int inner = __builtin_amdgcn_readfirstlane(gb[5]);
    const int aidx = swzr(g * 16 + n16, ks);
    const int row0 = kTP + (gy0 - by0) * bw + (gx0 - bx0) + n16;
    int bidx[NB];
    float4_t acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      bidx[j] = (j < nblk) ? swzr(row0 + j * bw, ks) : 0;        // unused blocks read row 0; never looked at
      acc[j] = float4_t{0.f, 0.f, 0.f, 0.f};
    }
    const uint32_t dbase = lds0 + (uint32_t)wave * 1024u;
    constexpr int GQ = 4, NQ = (NB + GQ - 1) / GQ;                  // B fragments are read GQ at a time, one group ahead of the MFMAs
    for (int c0 = 0; c0 < p.C; c0 += CC) {
      const T* b0 = f0 + c0;                                       // wave-uniform bases of this channel chunk
      const T* b1 = f1 + c0;
      dma16_so(b0, soff[0], dbase);
#pragma unroll
      for (int l = 1; l < NL; ++l)
        if (l < nl_used) dma16_so(b1, soff[l], dbase + (uint32_t)(l * 4096));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      raw_barrier();
      {
        const u32x4 a = rows[aidx];
        u32x4 bq[2][GQ];
#pragma unroll
        for (int t = 0; t < GQ; ++t) bq[0][t] = rows[bidx[t]];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          if (q + 1 < NQ) {
#pragma unroll
            for (int t = 0; t < GQ; ++t)
              if ((q + 1) * GQ + t < NB) bq[(q + 1) & 1][t] = rows[bidx[(q + 1) * GQ + t]];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < GQ; ++t)
            if (q * GQ + t < NB) acc[q * GQ + t] = mfma16r(a, bq[q & 1][t], acc[q * GQ + t], T{});
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      raw_barrier();
    }
    // ---- epilogue, wave-private ----
    if (inner) {
      // Every pixel of the group has its whole window inside the map and inside the group's union.  Lane (n16, kg) holds, for its
      // four pixels 4 kg + r4, union column n16 of every union row j.  The 4-tap blend happens in registers (DPP shift for the
      // right-hand tap, the next accumulator block for the lower one); what is left is a dynamic (row, column) selection per
      // pixel, done through a wave-private LDS column image WITHOUT predicates or per-element address arithmetic: the lane writes
      // its NB - 1 blended values of one pixel quad with immediate offsets, then lane (l16, kg) picks output kk = l16 + 16 it of
      // pixel 4 kg + r4 from column (window origin + kk % N1), row (origin + kk / N1).  (Round 2's first version stored each
      // value under a window predicate into a per-pixel image and walked it with an e / K loop: ~870 of the 1 309 VALU
      // instructions of a tile-wave, profiles/r02_local_corr_pmc.md.)  The staging rows are dead after the chunk loop's last
      // barrier, so the image lives there.
      constexpr int NBm1 = NB - 1;                                  // odd (11 / 13 / 15): conflict-free lane stride
      float* col = reinterpret_cast<float*>(rows) + g * (64 * NBm1);
      float axv[4], ayv[4];
      int sxv[4], syv[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int r = g * 16 + 4 * kg + r4;
        axv[r4] = s_ax[r];
        ayv[r4] = s_ay[r];
        sxv[r4] = s_x0[r] - R - gx0;                               // window origin inside the union: 0 <= sx, sx + N1 <= 16
        syv[r4] = s_y0[r] - R - gy0;                               //                                 0 <= sy, sy + N1 <= NB - 1
      }
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const float d = acc[j][r4];
          acc[j][r4] = d + axv[r4] * (right_neighbour(d) - d);
        }
      const size_t kstride = p.out_nhwc ? 1 : (size_t)H * W;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
#pragma unroll
        for (int j = 0; j < NBm1; ++j) col[lane * NBm1 + j] = acc[j][r4] + ayv[r4] * (acc[j + 1][r4] - acc[j][r4]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // same wave: LDS ops are ordered; this only pins the compiler
        const int mp = 4 * kg + r4;
        const int y = ty0 + (g >> 1) * 4 + (mp >> 2), x = tx0 + (g & 1) * 4 + (mp & 3);
        T* po = p.out_nhwc ? out + (((size_t)b * H + y) * W + x) * p.out_pitch : out + ((size_t)b * p.out_pitch * H + y) * W + x;
        const float* pc = col + ((kg * 16 + sxv[r4]) * NBm1 + syv[r4]);
#pragma unroll
        for (int it = 0; it < (K + 15) / 16; ++it) {
          const int kk = n16 + 16 * it;
          if (kk < K) {
            const int iy = kk / N1, ix = kk - iy * N1;
            po[(size_t)kk * kstride] = from_f32<T>(pc[ix * NBm1 + iy] * p.scale);
          }
        }
        asm volatile("" ::: "memory");                             // the next quad's writes stay behind these reads
      }
      return;
    }
    int wx[4], wy[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      wx[r4] = s_x0[g * 16 + 4 * kg + r4] - R;
      wy[r4] = s_y0[g * 16 + 4 * kg + r4] - R;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (j < nblk && n16 < gw) {
        const int ya = gy0 + j, xa = gx0 + n16;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int dy = ya - wy[r4], dx = xa - wx[r4];
          if (dy >= 0 && dy < N2 && dx >= 0 && dx < N2) s_D[(g * 16 + 4 * kg + r4) * (Q + 1) + dy * N2 + dx] = acc[j][r4] * p.scale;
        }
      }
    }
    for (int e = lane; e < 16 * K; e += 64) {
      int m, kk;
      if (p.out_nhwc) { m = e / K; kk = e - m * K; } else { kk = e >> 4; m = e & 15; }
      const int y = ty0 + (g >> 1) * 4 + (m >> 2), x = tx0 + (g & 1) * 4 + (m & 3);
      if (y >= H || x >= W) continue;
      const int iy = kk / N1, ix = kk - iy * N1;
      const int r = g * 16 + m;
      const float ax = s_ax[r], ay = s_ay[r];
      const int yy = s_y0[r] - R + iy, xx = s_x0[r] - R + ix;
      const float* d = s_D + r * (Q + 1) + iy * N2 + ix;
      const bool y0in = yy >= 0 && yy < H, y1in = yy + 1 >= 0 && yy + 1 < H;
      const bool x0in = xx >= 0 && xx < W, x1in = xx + 1 >= 0 && xx + 1 < W;
      const float d00 = (y0in && x0in) ? d[0] : 0.f, d01 = (y0in && x1in) ? d[1] : 0.f;
      const float d10 = (y1in && x0in) ? d[N2] : 0.f, d11 = (y1in && x1in) ? d[N2 + 1] : 0.f;
      const float top = d00 + ax * (d01 - d00);
      const float bot = d10 + ax * (d11 - d10);
      const size_t o = p.out_nhwc ? (((size_t)b * H + y) * W + x) * p.out_pitch + kk : (((size_t)b * p.out_pitch + kk) * H + y) * W + x;
      out[o] = from_f32<T>(top + ay * (bot - top));
    }
    return;
  }

  // =========================== incoherent tile: per-pixel patches on the VALU ===========================
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
  for (int i = tid; i < kTP * (Q + 1); i += 256) s_D[i] = 0.f;
  __syncthreads();
  for (int pass = 0; pass < kTP / SB; ++pass) {
    int pixw[UB], qw[UB], rowidx[UB];
    float sacc[UB];
#pragma unroll
    for (int w = 0; w < UB; ++w) {
      const int uu = g16 + 16 * w;
      const int sl = uu / NIT, it = uu - sl * NIT;
      int pix = (uu < SB * NIT) ? pass * SB + sl : kTP;
      const bool pact = pix < kTP;
      pix = pact ? pix : 0;
      int py, px;
      fpix(pix, py, px);
      const bool pvalid = pact && (ty0 + py < H) && (tx0 + px < W);
      const int q = it * 16 + idx;
      const int yy = s_y0[pix] - R + q / N2, xx = s_x0[pix] - R + q % N2;
      const bool qok = pvalid && (q < Q) && (yy >= 0) && (yy < H) && (xx >= 0) && (xx < W);
      pixw[w] = pvalid ? pix : -1;
      qw[w] = q;
      rowidx[w] = qok ? kTP + sl * QP + q : ZROW;
      sacc[w] = 0.f;
    }
    const T* ssrc[SNL];
    constexpr int used_rows = kTP + SB * QP;
#pragma unroll
    for (int l = 0; l < SNL; ++l) {
      const int slot = l * 256 + tid;
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
        if (l * 64 < used_rows) dma16(ssrc[l] + c0, lds0 + (uint32_t)(l * 256 + wave * 64) * 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int w = 0; w < UB; ++w) {
        const int prow = pixw[w] < 0 ? 0 : pixw[w];
        const int r0 = rowidx[w];
        float s = sacc[w];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) s = dot16<T>(rows[swzr(prow, kk)], rows[swzr(r0, kk)], s);
        sacc[w] = s;
      }
      __syncthreads();
    }
#pragma unroll
    for (int w = 0; w < UB; ++w)
      if (qw[w] < Q && pixw[w] >= 0) s_D[pixw[w] * (Q + 1) + qw[w]] = sacc[w] * p.scale;
  }
  __syncthreads();
  for (int e = tid; e < kTP * K; e += 256) {
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
    const size_t o = p.out_nhwc ? (((size_t)b * H + y) * W + x) * p.out_pitch + kk : (((size_t)b * p.out_pitch + kk) * H + y) * W + x;
    out[o] = from_f32<T>(top + ay * (bot - top));
  }
}

template <int R> constexpr size_t t8_smem_bytes() {
  using G = T8Geom<R>;
  constexpr int Q = G::N2 * G::N2;
  return (size_t)G::ROWS * 64 + (size_t)kTP * (Q + 1) * 4 + 4 * kTP * 4 + kNG * 8 * 4;
}

template <typename T, int R>
int launch_t8(LCTileParams p, hipStream_t stream) {
  constexpr size_t smem = t8_smem_bytes<R>();
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = ensure_dyn_smem(reinterpret_cast<const void*>(local_corr_t8_kernel<T, R>), (int)smem, attr_done, "roma_local_corr")) return rc;
  p.tiles_x = (p.W + kTW - 1) / kTW;
  p.tiles_y = (p.H + kTH - 1) / kTH;
  const int grid = p.B * p.tiles_x * p.tiles_y;
  hipLaunchKernelGGL((local_corr_t8_kernel<T, R>), dim3(grid), dim3(256), smem, stream, p);
  ROMA_CHECK_LAUNCH();
}

}  // namespace

// entry used by roma_local_corr (local_corr.hip) for 16-bit channels-last inputs with r <= 3 and C a multiple of 32
int local_corr_t8(const LCTileParams& p, int r, int dtype, hipStream_t stream) {
  if (dtype == ROMA_F16) {
    switch (r) {
      case 1: return launch_t8<half_t, 1>(p, stream);
      case 2: return launch_t8<half_t, 2>(p, stream);
      case 3: return launch_t8<half_t, 3>(p, stream);
    }
  } else if (dtype == ROMA_BF16) {
    switch (r) {
      case 1: return launch_t8<bf16_t, 1>(p, stream);
      case 2: return launch_t8<bf16_t, 2>(p, stream);
      case 3: return launch_t8<bf16_t, 3>(p, stream);
    }
  }
  set_error("local_corr_t8: unsupported r=%d dtype=%d", r, dtype);
  return ROMA_E_UNSUPPORTED;
}

}  // namespace roma
