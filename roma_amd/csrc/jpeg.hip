// Baseline JPEG decoding for RegressionMatcher.match() on file paths — reference: romatch/models/matcher.py:606-637, 667-676
// (`Image.open(path).convert("RGB")`, i.e. PIL -> libjpeg(-turbo) with its defaults: JDCT_ISLOW, fancy up-sampling).
// SURVEY §8(f) rank 3 asks for the pre-processing on the device.  The entropy-coded segment is a serial bit stream, so the split is
// the usual one: Huffman decoding on the host (roma_jpeg_entropy_decode: a HOST function, plain C++), everything after it on the GPU
// (roma_jpeg_reconstruct): de-quantisation + the 8x8 inverse DCT, chroma up-sampling, YCbCr -> RGB, one uint8 (H, W, 3) image in HBM
// that roma_resample_u8 / roma_normalize_u8 consume — the decoded photograph never crosses PCIe, only its quantised coefficients do.
// The arithmetic is libjpeg's, restated from its published description (the "islow" integer IDCT of Loeffler / Ligtenberg / Moschytz with
// 13-bit constants and a 2-bit first-pass scale; the h2v2 "triangle" up-sampler, 3/4 near + 1/4 far in each direction with its 8 / 7
// rounding biases and replicated edges; the 16-bit fixed-point colour tables), so the result is BIT-IDENTICAL to PIL's
// (tests/test_jpeg.py: the four bundled photographs, one of them 618 pixels wide, and synthetic 4:4:4 / grey / restart-interval streams).
// Supported: 8-bit baseline / extended sequential and PROGRESSIVE Huffman streams, 1 or 3 components, 4:4:4 / 4:2:2 / 4:2:0, restart
// intervals.  Anything else (CMYK, RGB-stored, 12-bit, arithmetic, lossless) is reported as ROMA_E_UNSUPPORTED and the caller decodes with PIL on the host.
#include <cstring>
#include <vector>
#include "common.h"

namespace roma {
namespace {

constexpr int kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
  // canonical code tables: for code length l (1..16): the first code value, the first symbol index, the number of codes
  int mincode[17], valptr[17], maxcode[18];
  uint8_t vals[256];
  uint16_t look[512];                                            // 9-bit fast path: (length << 8) | symbol, 0 = longer code
  bool present = false;
};

struct Comp { int id, h, v, tq, td, ta; };

struct Header {
  int width = 0, height = 0, ncomp = 0;
  Comp comp[3];
  uint16_t qt[4][64];                                            // natural order
  bool qt_present[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  int restart = 0;
  size_t scan = 0;                                               // offset of the entropy-coded data
  int hmax = 1, vmax = 1;
  bool adobe_rgb = false, jfif = false;
  bool progressive = false, sof = false;
  // the scan the last SOS announced: its components (indices into comp), spectral band, successive-approximation bits
  int sc_ns = 0, sc_ci[3] = {0, 0, 0}, Ss = 0, Se = 63, Ah = 0, Al = 0;
};

void build_huff(Huff& h, const uint8_t* counts, const uint8_t* vals, int nvals) {
  int code = 0, k = 0;
  for (int l = 1; l <= 16; ++l) {
    h.valptr[l] = k;
    h.mincode[l] = code;
    code += counts[l - 1];
    k += counts[l - 1];
    h.maxcode[l] = counts[l - 1] ? code - 1 : -1;
    code <<= 1;
  }
  h.maxcode[17] = 0x7fffffff;
  memcpy(h.vals, vals, (size_t)nvals);
  memset(h.look, 0, sizeof(h.look));
  code = 0;
  k = 0;
  for (int l = 1; l <= 9; ++l) {
    for (int i = 0; i < counts[l - 1]; ++i, ++k, ++code) {
      const int lo = code << (9 - l);
      for (int j = 0; j < (1 << (9 - l)); ++j) h.look[lo + j] = (uint16_t)((l << 8) | vals[k]);
    }
    code <<= 1;
  }
  h.present = true;
}

// parse the marker segments from byte `i` (2: right behind SOI) up to and including the next SOS: 0 (H.scan = the entropy-coded data, the
// scan's parameters in H), 1 = end of image (no further scan), or a negative roma error code
int parse(const uint8_t* d, size_t n, Header& H, size_t i = 2) {
  if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) { set_error("roma_jpeg: not a JPEG stream (no SOI)"); return ROMA_E_ARG; }
  bool& sof = H.sof;
  while (i + 2 <= n) {
    if (d[i] == 0xFF && d[i + 1] == 0xD9) return 1;               // EOI
    if (i + 4 > n) break;
    if (d[i] != 0xFF) { set_error("roma_jpeg: marker expected at byte %zu", i); return ROMA_E_ARG; }
    const int m = d[i + 1];
    if (m == 0xFF) { ++i; continue; }                            // fill byte
    const size_t L = ((size_t)d[i + 2] << 8) | d[i + 3];
    if (L < 2 || i + 2 + L > n) { set_error("roma_jpeg: truncated segment at byte %zu", i); return ROMA_E_ARG; }
    const uint8_t* s = d + i + 4;
    const size_t sl = L - 2;
    if (m == 0xC0 || m == 0xC1 || m == 0xC2) {                   // baseline / extended sequential / progressive, Huffman
      H.progressive = m == 0xC2;
      if (sl < 6 || s[0] != 8) { set_error("roma_jpeg: %d-bit samples", sl ? s[0] : 0); return ROMA_E_UNSUPPORTED; }
      H.height = (s[1] << 8) | s[2];
      H.width = (s[3] << 8) | s[4];
      H.ncomp = s[5];
      if ((H.ncomp != 1 && H.ncomp != 3) || sl < 6 + 3 * (size_t)H.ncomp || H.width == 0 || H.height == 0) {
        set_error("roma_jpeg: %d components, %d x %d", H.ncomp, H.width, H.height);
        return ROMA_E_UNSUPPORTED;
      }
      for (int c = 0; c < H.ncomp; ++c) {
        H.comp[c] = Comp{s[6 + 3 * c], s[7 + 3 * c] >> 4, s[7 + 3 * c] & 15, s[8 + 3 * c] & 3, 0, 0};
        H.hmax = std::max(H.hmax, H.comp[c].h);
        H.vmax = std::max(H.vmax, H.comp[c].v);
      }
      sof = true;
    } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      set_error("roma_jpeg: SOF%d (lossless / arithmetic / hierarchical) is not decoded here", m - 0xC0);
      return ROMA_E_UNSUPPORTED;
    } else if (m == 0xDB) {                                      // quantisation tables (zig-zag order in the file)
      size_t o = 0;
      while (o < sl) {
        const int pq = s[o] >> 4, tq = s[o] & 15;
        if (tq > 3 || o + 1 + (pq ? 128 : 64) > sl) { set_error("roma_jpeg: bad DQT"); return ROMA_E_ARG; }
        for (int k = 0; k < 64; ++k) H.qt[tq][kZigzag[k]] = pq ? (uint16_t)((s[o + 1 + 2 * k] << 8) | s[o + 2 + 2 * k]) : s[o + 1 + k];
        H.qt_present[tq] = true;
        o += 1 + (pq ? 128 : 64);
      }
    } else if (m == 0xC4) {                                      // Huffman tables
      size_t o = 0;
      while (o + 17 <= sl) {
        const int tc = s[o] >> 4, th = s[o] & 15;
        int nv = 0;
        for (int k = 0; k < 16; ++k) nv += s[o + 1 + k];
        if (tc > 1 || th > 3 || nv > 256 || o + 17 + nv > sl) { set_error("roma_jpeg: bad DHT"); return ROMA_E_ARG; }
        build_huff(tc ? H.ac[th] : H.dc[th], s + o + 1, s + o + 17, nv);
        o += 17 + nv;
      }
    } else if (m == 0xEE) {                                      // APP14 "Adobe": transform 0 with three components = stored as RGB, not YCbCr
      if (sl >= 12 && memcmp(s, "Adobe", 5) == 0 && s[11] == 0) H.adobe_rgb = true;
    } else if (m == 0xE0) {
      if (sl >= 5 && memcmp(s, "JFIF", 4) == 0) H.jfif = true;
    } else if (m == 0xDD) {
      if (sl >= 2) H.restart = (s[0] << 8) | s[1];
    } else if (m == 0xDA) {                                      // start of scan
      if (!sof) { set_error("roma_jpeg: SOS before SOF"); return ROMA_E_ARG; }
      const int ns = sl ? s[0] : 0;
      if (ns < 1 || ns > H.ncomp || sl < 1 + 2 * (size_t)ns + 3) { set_error("roma_jpeg: bad SOS"); return ROMA_E_ARG; }
      if (!H.progressive && ns != H.ncomp) {
        set_error("roma_jpeg: a scan with %d of %d components (non-interleaved sequential scans are not decoded here)", ns, H.ncomp);
        return ROMA_E_UNSUPPORTED;
      }
      H.sc_ns = ns;
      for (int c = 0; c < ns; ++c) {
        int ci = -1;
        for (int k = 0; k < H.ncomp; ++k)
          if (H.comp[k].id == s[1 + 2 * c]) ci = k;
        if (ci < 0 || (c > 0 && ci <= H.sc_ci[c - 1])) { set_error("roma_jpeg: scan component order"); return ROMA_E_UNSUPPORTED; }
        H.sc_ci[c] = ci;
        H.comp[ci].td = s[2 + 2 * c] >> 4;
        H.comp[ci].ta = s[2 + 2 * c] & 15;
      }
      H.Ss = s[1 + 2 * ns];
      H.Se = s[2 + 2 * ns];
      H.Ah = s[3 + 2 * ns] >> 4;
      H.Al = s[3 + 2 * ns] & 15;
      if (!H.progressive) { H.Ss = 0; H.Se = 63; H.Ah = H.Al = 0; }
      if (H.Ss > H.Se || H.Se > 63 || H.Al > 13 || (H.Ss == 0 && H.Se != 0 && H.progressive) || (H.Ss > 0 && ns != 1)) {
        set_error("roma_jpeg: bad progressive scan parameters");
        return ROMA_E_ARG;
      }
      for (int c = 0; c < ns; ++c) {
        const Comp& cp = H.comp[H.sc_ci[c]];
        const bool need_dc = H.Ss == 0 && H.Ah == 0, need_ac = H.Se > 0;
        if ((need_dc && !H.dc[cp.td].present) || (need_ac && !H.ac[cp.ta].present) || !H.qt_present[cp.tq]) {
          set_error("roma_jpeg: a table the scan refers to is missing");
          return ROMA_E_ARG;
        }
      }
      H.scan = i + 2 + L;
      // colour space: YCbCr is what the kernels convert.  libjpeg takes three components for RGB when an Adobe marker says
      // "transform 0", or — without a JFIF marker — when the component ids spell 'R', 'G', 'B'
      if (H.ncomp == 3 && (H.adobe_rgb || (!H.jfif && H.comp[0].id == 'R' && H.comp[1].id == 'G' && H.comp[2].id == 'B'))) {
        set_error("roma_jpeg: three components stored as RGB (no YCbCr transform) are not decoded here");
        return ROMA_E_UNSUPPORTED;
      }
      // sampling: grey, 4:4:4, 4:2:2 or 4:2:0
      if (H.ncomp == 3) {
        const bool c444 = H.comp[0].h == 1 && H.comp[0].v == 1, c420 = H.comp[0].h == 2 && H.comp[0].v == 2;
        const bool c422 = H.comp[0].h == 2 && H.comp[0].v == 1;
        if (!(c444 || c420 || c422) || H.comp[1].h != 1 || H.comp[1].v != 1 || H.comp[2].h != 1 || H.comp[2].v != 1) {
          set_error("roma_jpeg: chroma sampling %dx%d,%dx%d,%dx%d (4:4:4, 4:2:2 and 4:2:0 are decoded here)", H.comp[0].h, H.comp[0].v, H.comp[1].h,
                    H.comp[1].v, H.comp[2].h, H.comp[2].v);
          return ROMA_E_UNSUPPORTED;
        }
      } else {
        H.comp[0].h = H.comp[0].v = H.hmax = H.vmax = 1;          // a single component is never interleaved: one block per MCU
      }
      return 0;
    }
    i += 2 + L;
  }
  set_error("roma_jpeg: no start of scan");
  return ROMA_E_ARG;
}

struct Bits {
  const uint8_t* d;
  size_t n, pos;
  uint64_t acc = 0;
  int cnt = 0;
  bool hit_marker = false;
  void fill() {
    if (cnt > 32) return;
    // fast path: four stream bytes none of which is 0xFF (no stuffing, no marker) go in at once
    if (!hit_marker && pos + 4 <= n) {
      const uint32_t w = ((uint32_t)d[pos] << 24) | ((uint32_t)d[pos + 1] << 16) | ((uint32_t)d[pos + 2] << 8) | d[pos + 3];
      if (!((w & ~(w + 0x01010101u) & 0x80808080u))) {           // no byte equals 0xFF (0xFF + 1 carries out of its byte)
        acc |= (uint64_t)w << (32 - cnt);
        cnt += 32;
        pos += 4;
        return;
      }
    }
    while (cnt <= 48) {
      int b = 0;
      if (!hit_marker && pos < n) {
        b = d[pos];
        if (b == 0xFF) {
          if (pos + 1 < n && d[pos + 1] == 0) pos += 2;           // stuffed zero
          else { hit_marker = true; b = 0; }                      // a marker: feed zeros from here (like libjpeg's "insert zeros")
        } else {
          ++pos;
        }
      }
      acc |= (uint64_t)b << (56 - cnt);
      cnt += 8;
    }
  }
  int peek(int k) { return (int)(acc >> (64 - k)); }
  void skip(int k) { acc <<= k; cnt -= k; }
  int get(int k) {
    if (k == 0) return 0;
    const int v = peek(k);
    skip(k);
    return v;
  }
  void reset() { acc = 0; cnt = 0; hit_marker = false; }
};

inline int decode_sym(Bits& b, const Huff& h) {
  b.fill();
  const int look = h.look[b.peek(9)];
  if (look) { b.skip(look >> 8); return look & 255; }
  int code = b.peek(10), l = 10;
  while (l <= 16 && code > h.maxcode[l]) { ++l; code = b.peek(l); }
  if (l > 16) return -1;
  b.skip(l);
  return h.vals[h.valptr[l] + code - h.mincode[l]];
}

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

// ---- device side ----------------------------------------------------------------------------------------------------------------
// one thread per 8x8 block: de-quantise, inverse DCT (two passes of the 8-point butterfly, 13-bit constants), level shift, clamp
__device__ __forceinline__ void idct8(const int (&in)[8], int (&out)[8], int shift, bool first) {
  constexpr int F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299, F1_847 = 15137,
                F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
  int z2 = in[2], z3 = in[6];
  int z1 = (z2 + z3) * F0_541;
  const int t2 = z1 + z3 * (-F1_847), t3 = z1 + z2 * F0_765;
  z2 = in[0];
  z3 = in[4];
  const int t0 = (z2 + z3) << 13, t1 = (z2 - z3) << 13;
  const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
  int o0 = in[7], o1 = in[5], o2 = in[3], o3 = in[1];
  z1 = o0 + o3;
  z2 = o1 + o2;
  z3 = o0 + o2;
  int z4 = o1 + o3;
  const int z5 = (z3 + z4) * F1_175;
  o0 *= F0_298; o1 *= F2_053; o2 *= F3_072; o3 *= F1_501;
  z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
  z3 += z5;
  z4 += z5;
  o0 += z1 + z3; o1 += z2 + z4; o2 += z2 + z3; o3 += z1 + z4;
  const int r = 1 << (shift - 1);
  (void)first;
  out[0] = (t10 + o3 + r) >> shift; out[7] = (t10 - o3 + r) >> shift;
  out[1] = (t11 + o2 + r) >> shift; out[6] = (t11 - o2 + r) >> shift;
  out[2] = (t12 + o1 + r) >> shift; out[5] = (t12 - o1 + r) >> shift;
  out[3] = (t13 + o0 + r) >> shift; out[4] = (t13 - o0 + r) >> shift;
}

__global__ __launch_bounds__(64) void jpeg_idct_kernel(const int16_t* __restrict__ coef, const uint16_t* __restrict__ qt, uint8_t* __restrict__ plane,
                                                       int blocks_w, int nblocks, int pitch) {
  const int bi = blockIdx.x * 64 + threadIdx.x;
  if (bi >= nblocks) return;
  const int16_t* c = coef + (size_t)bi * 64;
  int ws[8][8];
#pragma unroll
  for (int col = 0; col < 8; ++col) {                            // pass 1: columns, results scaled by 2^2
    int in[8], out[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) in[r] = (int)c[r * 8 + col] * (int)qt[r * 8 + col];
    idct8(in, out, 13 - 2, true);
#pragma unroll
    for (int r = 0; r < 8; ++r) ws[r][col] = out[r];
  }
  const int by = bi / blocks_w, bx = bi - by * blocks_w;
  uint8_t* dst = plane + (size_t)by * 8 * pitch + bx * 8;
#pragma unroll
  for (int r = 0; r < 8; ++r) {                                  // pass 2: rows, remove the 2^2 and the 8 of the 2-D transform, + 128
    int out[8];
    idct8(ws[r], out, 13 + 2 + 3, false);
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      lo |= (uint32_t)min(max(out[k] + 128, 0), 255) << (8 * k);
      hi |= (uint32_t)min(max(out[4 + k] + 128, 0), 255) << (8 * k);
    }
    *reinterpret_cast<uint2*>(dst + (size_t)r * pitch) = uint2{lo, hi};
  }
}

// Y (full resolution) + Cb, Cr (same, or half in both directions: "fancy" triangle up-sampling) -> RGB, one thread per pixel
__global__ __launch_bounds__(256) void jpeg_rgb_kernel(const uint8_t* __restrict__ yp, const uint8_t* __restrict__ cbp, const uint8_t* __restrict__ crp,
                                                       uint8_t* __restrict__ rgb, int W, int H, int ypitch, int cpitch, int cw, int ch, int sub) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const int Y = yp[(size_t)y * ypitch + x];
  int cb, cr;
  if (sub == 0) {
    cb = cbp[(size_t)y * cpitch + x];
    cr = crp[(size_t)y * cpitch + x];
  } else if (sub < 0) {
    cb = cr = 128;
  } else if (cw <= 2) {                                          // libjpeg picks the triangle filter only for more than two chroma columns
    const int yc = sub == 2 ? y : y >> 1;
    cb = cbp[(size_t)yc * cpitch + (x >> 1)];
    cr = crp[(size_t)yc * cpitch + (x >> 1)];
  } else if (sub == 2) {
    // 4:2:2, horizontal triangle filter only: column 2 j = (3 c[j] + c[j - 1] + 1) >> 2, column 2 j + 1 = (3 c[j] + c[j + 1] + 2) >> 2,
    // the first and the last output column are the first / last sample themselves
    const int j = x >> 1;
    const uint8_t *r0 = cbp + (size_t)y * cpitch, *r1 = crp + (size_t)y * cpitch;
    if ((x & 1) == 0) {
      cb = j == 0 ? r0[0] : (3 * r0[j] + r0[j - 1] + 1) >> 2;
      cr = j == 0 ? r1[0] : (3 * r1[j] + r1[j - 1] + 1) >> 2;
    } else {
      cb = j == cw - 1 ? r0[j] : (3 * r0[j] + r0[j + 1] + 2) >> 2;
      cr = j == cw - 1 ? r1[j] : (3 * r1[j] + r1[j + 1] + 2) >> 2;
    }
  } else {
    // output row 2 r + v takes 3/4 of chroma row r and 1/4 of row r - 1 (v = 0) or r + 1 (v = 1), edges replicated; output column
    // 2 j takes 3/4 of column j and 1/4 of column j - 1 with bias 8, column 2 j + 1 takes 1/4 of column j + 1 with bias 7; at the first
    // / last column the missing neighbour is the column itself
    const int r = y >> 1, j = x >> 1;
    const int rf = (y & 1) ? min(r + 1, ch - 1) : max(r - 1, 0);
    const int jn = (x & 1) ? min(j + 1, cw - 1) : max(j - 1, 0);
    const int bias = (x & 1) ? 7 : 8;
    const uint8_t *n0 = cbp + (size_t)r * cpitch, *f0 = cbp + (size_t)rf * cpitch, *n1 = crp + (size_t)r * cpitch, *f1 = crp + (size_t)rf * cpitch;
    cb = (3 * (3 * n0[j] + f0[j]) + (3 * n0[jn] + f0[jn]) + bias) >> 4;
    cr = (3 * (3 * n1[j] + f1[j]) + (3 * n1[jn] + f1[jn]) + bias) >> 4;
  }
  int R, G, B;
  if (sub < 0) {
    R = G = B = Y;
  } else {
    // 16-bit fixed point: 1.40200, 1.77200, 0.71414, 0.34414 scaled by 65536 (+0.5), the ONE_HALF of the green sum on the Cb term
    const int xb = cb - 128, xr = cr - 128;
    R = Y + ((91881 * xr + 32768) >> 16);
    B = Y + ((116130 * xb + 32768) >> 16);
    G = Y + ((-22554 * xb + 32768 - 46802 * xr) >> 16);
  }
  uint8_t* o = rgb + ((size_t)y * W + x) * 3;
  o[0] = (uint8_t)min(max(R, 0), 255);
  o[1] = (uint8_t)min(max(G, 0), 255);
  o[2] = (uint8_t)min(max(B, 0), 255);
}

}  // namespace
}  // namespace roma

using namespace roma;

// info[0..7] = width, height, components, chroma subsampling (0: 4:4:4, 1: 4:2:0, 2: 4:2:2, -1: grey), luma blocks per row, luma block rows, chroma
// blocks per row, chroma block rows.  HOST function.
extern "C" int roma_jpeg_info(const void* data, long nbytes, int* info) {
  ROMA_REQUIRE(data && info && nbytes > 0, ROMA_E_ARG, "roma_jpeg_info: null pointer");
  Header H;
  if (int rc = parse(static_cast<const uint8_t*>(data), (size_t)nbytes, H)) return rc;
  const int mw = 8 * H.hmax, mh = 8 * H.vmax;
  const int mcux = (H.width + mw - 1) / mw, mcuy = (H.height + mh - 1) / mh;
  info[0] = H.width; info[1] = H.height; info[2] = H.ncomp;
  info[3] = H.ncomp == 1 ? -1 : (H.hmax == 2 ? (H.vmax == 2 ? 1 : 2) : 0);
  info[4] = mcux * H.hmax; info[5] = mcuy * H.vmax;
  info[6] = H.ncomp == 1 ? 0 : mcux; info[7] = H.ncomp == 1 ? 0 : mcuy;
  return 0;
}

// Huffman-decode the scan: coef = the quantised coefficients, int16, natural (row-major) order inside a block, the blocks of component c
// in raster order at coef + off_c * 64 (off_0 = 0, off_1 = luma blocks, off_2 = luma + chroma blocks); qt = 3 x 64 uint16 de-quantisation
// tables in natural order (the component's own table at row c).  HOST function: all pointers in host memory.
extern "C" int roma_jpeg_entropy_decode(const void* data, long nbytes, int16_t* coef, uint16_t* qt) {
  ROMA_REQUIRE(data && coef && qt && nbytes > 0, ROMA_E_ARG, "roma_jpeg_entropy_decode: null pointer");
  const uint8_t* d = static_cast<const uint8_t*>(data);
  Header H;
  if (int rc = parse(d, (size_t)nbytes, H)) return rc;
  const int mw = 8 * H.hmax, mh = 8 * H.vmax;
  const int mcux = (H.width + mw - 1) / mw, mcuy = (H.height + mh - 1) / mh;
  size_t off[3] = {0, 0, 0};
  int bw[3], bh[3];
  size_t total = 0;
  for (int c = 0; c < H.ncomp; ++c) {
    bw[c] = mcux * H.comp[c].h;
    bh[c] = mcuy * H.comp[c].v;
    off[c] = total;
    total += (size_t)bw[c] * bh[c];
    memcpy(qt + 64 * c, H.qt[H.comp[c].tq], 128);
  }
  memset(coef, 0, total * 64 * sizeof(int16_t));
  if (H.progressive) {
    // ---- progressive: every scan adds a spectral band and / or one more bit of precision to the coefficient array (ITU T.81 Annex G) ----
    for (int nscan = 0; nscan < 1000; ++nscan) {
      Bits b{d, (size_t)nbytes, H.scan};
      const int ns = H.sc_ns, Ss = H.Ss, Se = H.Se, Ah = H.Ah, Al = H.Al;
      const int p1 = 1 << Al, m1 = -(1 << Al);
      // an interleaved scan (DC only) walks MCUs; a single-component scan walks that component's own blocks, ceil(width_c / 8) per row
      int nbx = mcux, nby = mcuy;
      if (ns == 1) {
        const Comp& cp = H.comp[H.sc_ci[0]];
        nbx = ((H.width * cp.h + H.hmax - 1) / H.hmax + 7) / 8;
        nby = ((H.height * cp.v + H.vmax - 1) / H.vmax + 7) / 8;
      }
      int pred[3] = {0, 0, 0}, eobrun = 0, togo = H.restart;
      for (int my = 0; my < nby; ++my)
        for (int mx = 0; mx < nbx; ++mx) {
          if (H.restart && togo == 0) {
            size_t q = b.pos;
            while (q + 1 < (size_t)nbytes && !(d[q] == 0xFF && d[q + 1] >= 0xD0 && d[q + 1] <= 0xD7)) ++q;
            if (q + 1 >= (size_t)nbytes) { set_error("roma_jpeg: restart marker missing"); return ROMA_E_ARG; }
            b.pos = q + 2;
            b.reset();
            pred[0] = pred[1] = pred[2] = 0;
            eobrun = 0;
            togo = H.restart;
          }
          for (int sc = 0; sc < ns; ++sc) {
            const int c = H.sc_ci[sc];
            const int nh = ns == 1 ? 1 : H.comp[c].h, nv = ns == 1 ? 1 : H.comp[c].v;
            for (int v = 0; v < nv; ++v)
              for (int h = 0; h < nh; ++h) {
                int16_t* blk = coef + (off[c] + (size_t)(my * nv + v) * bw[c] + mx * nh + h) * 64;
                if (Ss == 0) {                                   // DC: first pass or one more bit
                  if (Ah == 0) {
                    const int sz = decode_sym(b, H.dc[H.comp[c].td]);
                    if (sz < 0 || sz > 15) { set_error("roma_jpeg: corrupt DC code"); return ROMA_E_ARG; }
                    b.fill();
                    if (sz) pred[c] += extend(b.get(sz), sz);
                    blk[0] = (int16_t)(pred[c] * p1);
                  } else {
                    b.fill();
                    if (b.get(1)) blk[0] |= (int16_t)p1;
                  }
                  continue;
                }
                const Huff& ha = H.ac[H.comp[c].ta];
                if (Ah == 0) {                                   // AC band, first pass
                  if (eobrun > 0) { --eobrun; continue; }
                  for (int k = Ss; k <= Se;) {
                    const int rs = decode_sym(b, ha);
                    if (rs < 0) { set_error("roma_jpeg: corrupt AC code"); return ROMA_E_ARG; }
                    const int r = rs >> 4, sz = rs & 15;
                    if (sz == 0) {
                      if (r < 15) {                              // end of band for 2^r + bits blocks, this one included
                        eobrun = (1 << r) - 1;
                        if (r) { b.fill(); eobrun += b.get(r); }
                        break;
                      }
                      k += 16;
                      continue;
                    }
                    k += r;
                    if (k > Se) { set_error("roma_jpeg: AC run past the band"); return ROMA_E_ARG; }
                    b.fill();
                    blk[kZigzag[k]] = (int16_t)(extend(b.get(sz), sz) * p1);
                    ++k;
                  }
                  continue;
                }
                // AC band, refinement: one correction bit for every coefficient that is already non-zero, new +-2^Al coefficients
                // placed after `r` still-zero positions (the decoder of Annex G.1.2.3)
                int k = Ss;
                if (eobrun == 0) {
                  for (; k <= Se; ++k) {
                    const int rs = decode_sym(b, ha);
                    if (rs < 0) { set_error("roma_jpeg: corrupt AC code"); return ROMA_E_ARG; }
                    int r = rs >> 4, val = rs & 15;
                    if (val) {
                      b.fill();
                      val = b.get(1) ? p1 : m1;
                    } else if (r != 15) {
                      eobrun = 1 << r;
                      if (r) { b.fill(); eobrun += b.get(r); }
                      break;
                    }
                    do {
                      int16_t* cf = blk + kZigzag[k];
                      if (*cf != 0) {
                        b.fill();
                        if (b.get(1) && (*cf & p1) == 0) *cf = (int16_t)(*cf + (*cf >= 0 ? p1 : m1));
                      } else if (--r < 0) {
                        break;
                      }
                      ++k;
                    } while (k <= Se);
                    if (val && k <= Se) blk[kZigzag[k]] = (int16_t)val;
                  }
                }
                if (eobrun > 0) {
                  for (; k <= Se; ++k) {
                    int16_t* cf = blk + kZigzag[k];
                    if (*cf != 0) {
                      b.fill();
                      if (b.get(1) && (*cf & p1) == 0) *cf = (int16_t)(*cf + (*cf >= 0 ? p1 : m1));
                    }
                  }
                  --eobrun;
                }
              }
          }
          if (H.restart) --togo;
        }
      // the next marker segment: further tables and scans, or the end of the image
      size_t q = b.pos;
      while (q + 1 < (size_t)nbytes && !(d[q] == 0xFF && d[q + 1] != 0 && !(d[q + 1] >= 0xD0 && d[q + 1] <= 0xD7) && d[q + 1] != 0xFF)) ++q;
      if (q + 1 >= (size_t)nbytes) break;                        // no EOI: what has been decoded stands (libjpeg warns and does the same)
      const int rc = parse(d, (size_t)nbytes, H, q);
      if (rc == 1) break;
      if (rc < 0) return rc;
    }
    for (int c = 0; c < H.ncomp; ++c) memcpy(qt + 64 * c, H.qt[H.comp[c].tq], 128);
    return 0;
  }
  Bits b{d, (size_t)nbytes, H.scan};
  int pred[3] = {0, 0, 0};
  int togo = H.restart;
  for (int my = 0; my < mcuy; ++my)
    for (int mx = 0; mx < mcux; ++mx) {
      if (H.restart && togo == 0) {                              // restart marker: byte-align, skip RSTn, reset the predictors
        size_t p = b.pos;
        while (p + 1 < (size_t)nbytes && !(d[p] == 0xFF && d[p + 1] >= 0xD0 && d[p + 1] <= 0xD7)) ++p;
        if (p + 1 >= (size_t)nbytes) { set_error("roma_jpeg: restart marker missing"); return ROMA_E_ARG; }
        b.pos = p + 2;
        b.reset();
        pred[0] = pred[1] = pred[2] = 0;
        togo = H.restart;
      }
      for (int c = 0; c < H.ncomp; ++c)
        for (int v = 0; v < H.comp[c].v; ++v)
          for (int h = 0; h < H.comp[c].h; ++h) {
            int16_t* blk = coef + (off[c] + (size_t)(my * H.comp[c].v + v) * bw[c] + mx * H.comp[c].h + h) * 64;
            const Huff& hd = H.dc[H.comp[c].td];
            const Huff& ha = H.ac[H.comp[c].ta];
            int s = decode_sym(b, hd);
            if (s < 0 || s > 15) { set_error("roma_jpeg: corrupt DC code"); return ROMA_E_ARG; }
            b.fill();
            if (s) pred[c] += extend(b.get(s), s);
            blk[0] = (int16_t)pred[c];
            for (int k = 1; k < 64;) {
              const int rs = decode_sym(b, ha);
              if (rs < 0) { set_error("roma_jpeg: corrupt AC code"); return ROMA_E_ARG; }
              const int r = rs >> 4, sz = rs & 15;
              if (sz == 0) {
                if (r != 15) break;                              // end of block
                k += 16;
                continue;
              }
              k += r;
              if (k > 63) { set_error("roma_jpeg: AC run past the block"); return ROMA_E_ARG; }
              b.fill();
              blk[kZigzag[k]] = (int16_t)extend(b.get(sz), sz);
              ++k;
            }
          }
      if (H.restart) --togo;
    }
  return 0;
}

// coef, qt: DEVICE copies of what roma_jpeg_entropy_decode produced; planes: device scratch of (luma blocks + 2 x chroma blocks) x 64
// bytes; rgb: uint8 (height, width, 3).  info as returned by roma_jpeg_info.
extern "C" int roma_jpeg_reconstruct(const int16_t* coef, const uint16_t* qt, void* planes, void* rgb, const int* info, void* stream) {
  ROMA_REQUIRE(coef && qt && planes && rgb && info, ROMA_E_ARG, "roma_jpeg_reconstruct: null pointer");
  const int W = info[0], Hh = info[1], nc = info[2], sub = info[3], ybw = info[4], ybh = info[5], cbw = info[6], cbh = info[7];
  ROMA_REQUIRE(W > 0 && Hh > 0 && (nc == 1 || nc == 3) && ybw > 0 && ybh > 0 && ybw * 8 >= W && ybh * 8 >= Hh, ROMA_E_SHAPE, "roma_jpeg_reconstruct: bad info");
  hipStream_t s = static_cast<hipStream_t>(stream);
  uint8_t* yp = static_cast<uint8_t*>(planes);
  const int ny = ybw * ybh, ncb = cbw * cbh;
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3((ny + 63) / 64), dim3(64), 0, s, coef, qt, yp, ybw, ny, ybw * 8);
  uint8_t *cbp = yp + (size_t)ny * 64, *crp = cbp + (size_t)ncb * 64;
  if (nc == 3) {
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((ncb + 63) / 64), dim3(64), 0, s, coef + (size_t)ny * 64, qt + 64, cbp, cbw, ncb, cbw * 8);
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((ncb + 63) / 64), dim3(64), 0, s, coef + (size_t)(ny + ncb) * 64, qt + 128, crp, cbw, ncb, cbw * 8);
  }
  const int cw = (W + 1) / 2, ch = (Hh + 1) / 2;                 // libjpeg's down-sampled dimensions: what the up-sampler walks
  hipLaunchKernelGGL(jpeg_rgb_kernel, dim3((W + 255) / 256, Hh), dim3(256), 0, s, yp, cbp, crp, static_cast<uint8_t*>(rgb), W, Hh, ybw * 8,
                     cbw * 8, cw, ch, sub);
  ROMA_CHECK_LAUNCH();
}
