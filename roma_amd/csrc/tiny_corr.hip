// TinyRoMa corr_volume + pos_embed fused on the matrix cores — reference: romatch/models/tiny.py:241-254
// (corr_volume = einsum('bci,bcj->bji')/sqrt(C)) and tiny.py:178-203 (pos_embed: soft-argmax of target coordinates).
// The (H1*W1) x (H0*W0) volume (92 MB / pair at 480x640) is never written: every wavefront owns 32 source pixels
// (MFMA columns), streams the target features in 32-row tiles through v_mfma_f32_32x32x2_f32 (exact fp32; TinyRoMa
// runs fp32) with the target index on the accumulator rows, so each lane holds 16 scores of ITS source pixel and the
// soft-argmax is an in-lane online softmax (flash-style) with one cross-half shuffle at the end.
//   exact : out_i = sum_j softmax_j(s_ij) * grid[j]                                        (tiny.py:201-202)
//   fast  : best_i = argmax_j s_ij; softmax over {s_ij : j on the stride-4 sub-grid} U {float(best_i)} (the reference
//           concatenates the integer arg-max INDEX as the extra logit, tiny.py:196); out = sum_c P_c grid_lr[c] +
//           P_last * grid[best_i]                                                          (tiny.py:187-198)
#include "common.h"

namespace roma {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float lin(int i, int n, float half_step) {   // torch.linspace(-1+hs, 1-hs, n)[i]
  const float start = -1.f + half_step, end = 1.f - half_step;
  if (n == 1) return start;
  const float step = (end - start) / (float)(n - 1);
  return (i < n / 2) ? start + step * i : end - step * (n - 1 - i);
}

template <int C, bool EXACT>
__global__ __launch_bounds__(256) void tiny_corr_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                        float* __restrict__ out, int H0, int W0, int H1, int W1, float scale) {
  constexpr int HK = C / 2;                       // k-steps; lane half h handles channels [h*HK, h*HK+HK)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int b = blockIdx.y;
  const int N0 = H0 * W0, N1 = H1 * W1;
  const int i = (blockIdx.x * 4 + wave) * 32 + col;          // source pixel of this lane
  const float* f0b = f0 + (size_t)b * N0 * C;
  const float* f1b = f1 + (size_t)b * N1 * C;
  float bq[HK];
  {
    const int ii = i < N0 ? i : N0 - 1;
#pragma unroll
    for (int k = 0; k < HK; k += 4) {
      const float4_t v = *reinterpret_cast<const float4_t*>(f0b + (size_t)ii * C + half * HK + k);
      bq[k] = v[0]; bq[k + 1] = v[1]; bq[k + 2] = v[2]; bq[k + 3] = v[3];
    }
  }
  float m = -INFINITY, den = 0.f, sx = 0.f, sy = 0.f;       // online softmax state
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int j0 = 0; j0 < N1; j0 += 32) {
    const int jr = min(j0 + col, N1 - 1);                    // A row fed by this lane
    float aq[HK];
#pragma unroll
    for (int k = 0; k < HK; k += 4) {
      const float4_t v = *reinterpret_cast<const float4_t*>(f1b + (size_t)jr * C + half * HK + k);
      aq[k] = v[0]; aq[k + 1] = v[1]; aq[k + 2] = v[2]; aq[k + 3] = v[3];
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int k = 0; k < HK; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[k], bq[k], acc, 0, 0, 0);
    // lane holds s[j = j0 + row(r)][i] for r = 0..15
    float tmax = -INFINITY;
    float s[16];
    bool use[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      s[r] = acc[r] * scale;
      const bool in = j < N1;
      if (!EXACT) {
        if (in && s[r] > best) { best = s[r]; bi = j; }      // ascending j within a lane: first maximum kept
        const int jy = j / W1, jx = j - jy * W1;
        use[r] = in && ((jy & 3) == 0) && ((jx & 3) == 0);
      } else {
        use[r] = in;
      }
      if (use[r]) tmax = fmaxf(tmax, s[r]);
    }
    if (tmax > m) {
      const float f = expf(m - tmax);                        // exp(-inf) = 0 on the first tile
      den *= f; sx *= f; sy *= f;
      m = tmax;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (use[r]) {
        const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int jy = j / W1, jx = j - jy * W1;
        const float e = expf(s[r] - m);
        const float gx = EXACT ? lin(jx, W1, 1.f / W1) : lin(jx >> 2, W1 >> 2, 4.f / W1);
        const float gy = EXACT ? lin(jy, H1, 1.f / H1) : lin(jy >> 2, H1 >> 2, 4.f / H1);
        den += e;
        sx = __builtin_fmaf(e, gx, sx);
        sy = __builtin_fmaf(e, gy, sy);
      }
    }
  }
  // combine the two lane halves (same source pixel, disjoint target rows)
  {
    const float om = __shfl_xor(m, 32, 64), od = __shfl_xor(den, 32, 64), ox = __shfl_xor(sx, 32, 64), oy = __shfl_xor(sy, 32, 64);
    const float mm = fmaxf(m, om);
    const float fa = (m == -INFINITY) ? 0.f : expf(m - mm), fb = (om == -INFINITY) ? 0.f : expf(om - mm);
    den = den * fa + od * fb; sx = sx * fa + ox * fb; sy = sy * fa + oy * fb; m = mm;
    const float ob = __shfl_xor(best, 32, 64);
    const int oi = __shfl_xor(bi, 32, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (half == 0 && i < N0) {
    float ox, oy;
    if (EXACT) {
      ox = sx / den; oy = sy / den;
    } else {
      const float extra = (float)bi;                         // the arg-max index itself is the extra logit
      const float mm = fmaxf(m, extra);
      const float fa = expf(m - mm), fe = expf(extra - mm);
      const float tot = den * fa + fe;
      const int by = bi / W1, bx = bi - by * W1;
      ox = (sx * fa + fe * lin(bx, W1, 1.f / W1)) / tot;
      oy = (sy * fa + fe * lin(by, H1, 1.f / H1)) / tot;
    }
    out[((size_t)b * 2 + 0) * N0 + i] = ox;
    out[((size_t)b * 2 + 1) * N0 + i] = oy;
  }
}

template <int C>
void launch_tiny(const float* f0, const float* f1, float* out, int B, int H0, int W0, int H1, int W1, int exact, hipStream_t s) {
  dim3 grid((H0 * W0 + 127) / 128, B);
  const float scale = 1.f / sqrtf((float)C);
  if (exact)
    hipLaunchKernelGGL((tiny_corr_kernel<C, true>), grid, dim3(256), 0, s, f0, f1, out, H0, W0, H1, W1, scale);
  else
    hipLaunchKernelGGL((tiny_corr_kernel<C, false>), grid, dim3(256), 0, s, f0, f1, out, H0, W0, H1, W1, scale);
}

}  // namespace
}  // namespace roma

using namespace roma;

extern "C" int roma_tiny_corr_posembed(const float* f0, const float* f1, float* out, int B, int C, int H0, int W0, int H1, int W1,
                                       int exact, void* stream) {
  ROMA_REQUIRE(f0 && f1 && out, ROMA_E_ARG, "roma_tiny_corr_posembed: null pointer");
  ROMA_REQUIRE(B > 0 && H0 > 0 && W0 > 0 && H1 > 0 && W1 > 0, ROMA_E_SHAPE, "roma_tiny_corr_posembed: bad shape");
  ROMA_REQUIRE(exact || (H1 % 4 == 0 && W1 % 4 == 0), ROMA_E_SHAPE, "roma_tiny_corr_posembed: fast path needs H1,W1 divisible by 4");
  ROMA_REQUIRE(aligned16(f0) && aligned16(f1), ROMA_E_ALIGN, "roma_tiny_corr_posembed: features must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (C) {
    case 16: launch_tiny<16>(f0, f1, out, B, H0, W0, H1, W1, exact, s); break;
    case 32: launch_tiny<32>(f0, f1, out, B, H0, W0, H1, W1, exact, s); break;
    case 64: launch_tiny<64>(f0, f1, out, B, H0, W0, H1, W1, exact, s); break;
    case 128: launch_tiny<128>(f0, f1, out, B, H0, W0, H1, W1, exact, s); break;
    default:
      set_error("roma_tiny_corr_posembed: C=%d not in {16,32,64,128}", C);
      return ROMA_E_UNSUPPORTED;
  }
  ROMA_CHECK_LAUNCH();
}
