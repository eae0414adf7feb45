/*
 * roma_hip.h — C ABI of libroma_hip.so: the MI355X (gfx950) kernels behind the RoMa dense-matching hot path
 * `RegressionMatcher.match()`.
 *
 * The reference (techshoww/RoMa) is pure Python/PyTorch and has NO FFI of its own; its seams are the Python call
 * signatures listed in SURVEY.md §8(b).  Every entry point below replaces the chain of stock ATen ops behind one of
 * those seams and cites it (file:line under the reference root).  A maintainer binds them with `ctypes`
 * (INTEGRATION.md shows the stubs); roma_amd/_lib.py is that binding for this repository.
 *
 * Conventions
 *   - plain C: pointers, ints, floats.  No torch / C++ types cross the boundary.
 *   - every pointer is DEVICE memory owned by the caller; the library allocates nothing and keeps no state
 *     except a thread-local error string.
 *   - every launch is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the default stream).
 *   - return 0 = ok; <0 = bad argument (ROMA_E_*), roma_last_error() has the text; >0 = a hipError_t.
 *   - dtype codes ROMA_F32/F16/BF16 describe feature/logit storage; accumulation is always fp32.
 *     flow / certainty maps are always fp32 (as in the reference, matcher.py:141,397-402).
 *   - feature layouts: ROMA_NCHW = (B,C,H,W) contiguous; ROMA_NHWC = (B,H,W,pitch) with the C channels of interest
 *     starting at the pointer and `pitch` >= C elements between consecutive pixels (so a channel slice of a wider
 *     channels-last buffer — e.g. the ConvRefiner concat buffer — can be read or written in place).
 *     For ROMA_NCHW `pitch` is the number of channels of the enclosing (B,pitch,H,W) buffer.
 */
#ifndef ROMA_HIP_H
#define ROMA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ROMA_ABI_VERSION 5

enum { ROMA_F32 = 0, ROMA_F16 = 1, ROMA_BF16 = 2 };
enum { ROMA_NCHW = 0, ROMA_NHWC = 1 };
/* roma_local_corr kernel selection for 16-bit channels-last inputs with r <= 3 (other inputs have one kernel): AUTO picks by launch size */
enum { ROMA_LC_AUTO = 0, ROMA_LC_TILE8X4 = 1, ROMA_LC_TILE8X8 = 2, ROMA_LC_ROWS8 = 3 };
enum { ROMA_E_ARG = -1, ROMA_E_DTYPE = -2, ROMA_E_SHAPE = -3, ROMA_E_ALIGN = -4, ROMA_E_UNSUPPORTED = -5 };

int roma_abi_version(void);
const char* roma_last_error(void);

/* local_correlation — romatch/utils/local_correlation.py:4-48, called from ConvRefiner.forward (matcher.py:121-125).
 *   out[b,k,y,x] = C^-1/2 * sum_c f0[b,c,y,x] * bilinear(f1[b,c], flow[b,:,y,x] + delta_k),  zeros padding,
 *   align_corners=False, k = iy*(2r+1)+ix, delta_k = ((ix-r)*2/W, (iy-r)*2/H).
 *   flow: (B,2,H,W) fp32 planar, (x,y) in [-1,1]; NULL = identity grid (local_correlation.py:16-27).
 *   f0,f1: `dtype`, `layout`, pitches f0_pitch/f1_pitch.  out: K=(2r+1)^2 channels, `dtype`, out_layout/out_pitch.
 *   r in {1..7}.  f1_batch_shift: f0's item b is correlated with f1's item (b + f1_batch_shift) % B — B/2 for
 *   forward_symmetric (matcher.py:516-528), whose second operand is the first with its batch halves swapped; 0 otherwise.
 *   variant: ROMA_LC_AUTO, or one of the kernels for 16-bit channels-last inputs with r <= 3 (8x4 / 8x8 tiles staged in 32-channel
 *   chunks; ROWS8 = the row-streaming kernel on 8x8 tiles, C = 256 or 512, else the 8x8 chunk kernel).  Same results
 *   to fp32 summation order; ignored where only one kernel applies. */
int roma_local_corr(const void* f0, const void* f1, const float* flow, void* out,
                    int B, int C, int H, int W, int r, int dtype,
                    int layout, int f0_pitch, int f1_pitch, int out_layout, int out_pitch, int f1_batch_shift, int variant,
                    void* stream);

/* F.grid_sample(y, flow^T, mode=bilinear, padding zeros, align_corners=False) — matcher.py:109 (ConvRefiner warp),
 * tiny.py:357,363.  src: (B,C,Hs,Ws); flow (B,2,H,W) fp32 planar; dst: (B,C,H,W).  Layout/pitch rules as above.
 * dst item b samples src item (b + src_batch_shift) % B (see roma_local_corr). */
int roma_warp_bilinear(const void* src, const float* flow, void* dst,
                       int B, int C, int Hs, int Ws, int H, int W, int dtype,
                       int layout, int src_pitch, int dst_layout, int dst_pitch, int src_batch_shift, void* stream);

/* displacement embedding — matcher.py:111-120: emb = Conv1x1(2->E)(gain * (flow - identity_grid)), gain = 40/32*scale_factor.
 *   weight (E,2) fp32, bias (E) fp32, flow (B,2,H,W) fp32, dst E channels of `dtype` in dst_layout/dst_pitch. */
int roma_disp_emb(const float* flow, const float* weight, const float* bias, void* dst,
                  int B, int E, int H, int W, float gain, int dtype, int dst_layout, int dst_pitch, void* stream);

/* F.interpolate(x, size=(Ho,Wo), mode="bilinear", align_corners=False) on fp32 planar maps — matcher.py:349-360,
 * 408-417, 657-659.  x: (N,Hi,Wi) planes, y: (N,Ho,Wo).  Optional fused flow update of Decoder.forward
 * (matcher.py:397-399): not here; see roma_flow_update. */
int roma_interp_bilinear(const float* x, float* y, int N, int Hi, int Wi, int Ho, int Wo, void* stream);

/* Decoder.forward update step — matcher.py:397-402:
 *   flow[b,0] += ins*delta[b,0]/(4*Wf); flow[b,1] += ins*delta[b,1]/(4*Hf); cert[b] += delta[b,2]
 *   delta: (B,3,H,W) fp32 planar (the refiner's out_conv result); cert may be NULL-initialised via cert_in==NULL (=0). */
int roma_flow_update(float* flow, float* cert, const float* cert_in, const float* delta,
                     int B, int H, int W, float sx, float sy, void* stream);

/* cls_to_flow_refine — romatch/utils/utils.py:301-323 (softmax over res^2 anchors, mode, 5-point refinement with
 * CLAMPED neighbour indices).  logits element (b, c, p) at logits[b*stride_b + c*stride_c + p*stride_p], p = y*W+x,
 * C = res*res classes (C a multiple of 64).  flow out: (B,2,H,W) fp32 planar (the caller's permute(0,3,1,2),
 * matcher.py:383-385).  If cert_out != NULL the extra logit c == C (transformer/__init__.py:45) is copied to
 * cert_out (B,1,H,W) fp32. */
int roma_cls_to_flow_refine(const void* logits, float* flow_out, float* cert_out,
                            int B, int C, int HW, long stride_b, long stride_c, long stride_p,
                            int dtype, void* stream);

/* CosKernel — matcher.py:154-163:  K[b,n,m] = exp((<x_n,y_m>/(|x_n||y_m| + eps) - 1)/T), fp32 arithmetic / fp32 out on the
 * exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).  x: (B,N,·) rows of `dtype` with x_pitch elements between rows (so the D feature
 * channels of a channels-last map can be read in place; 16-bit storage is widened exactly, i.e. the result equals the fp32
 * kernel on x.float(), matcher.py:254), y: (B,M,·) likewise; item b of x meets item (b + y_batch_shift) % B of y.
 * K: (B,N,M) fp32.  D a multiple of 16.  `diag_add` is added to K[b,i,i] (GP.forward's K_yy + sigma*I, matcher.py:259-261). */
int roma_cos_kernel(const void* x, const void* y, float* K, int B, int N, int M, int D, int dtype, int x_pitch, int y_pitch,
                    int y_batch_shift, float T, float eps, float diag_add, void* stream);

/* One diagonal-block step of the blocked Cholesky solve that replaces GP.forward's inv(K_yy + sigma I) @ f —
 * matcher.py:259-263.  For each of the B matrices: the nb x nb block at A (row-major, leading dimension lda, batch stride
 * strideA; only its lower triangle is read) is replaced by its Cholesky factor L (lower), and W (nb x nb, ldw, strideW)
 * receives L^-1.  A non-positive or NaN pivot p (0-based, within the block) is clamped and recorded: if info[b] == 0 it
 * becomes info_base + p + 1, so a zero-initialised info keeps the FIRST failing pivot of a whole blocked solve (the
 * reference's torch.linalg.inv raises in that case, matcher.py:261).  nb <= 64.  spd_solve (roma_amd/ops.py) calls it once, for the
 * first block; every later diagonal block is factored inside roma_chol_step, and the substitutions are roma_chol_subst_step. */
int roma_chol_diag_block(float* A, int lda, long strideA, float* W, int ldw, long strideW, int nb, int B, int* info,
                         int info_base, void* stream);

/* One fused forward step of the same blocked solve (matcher.py:259-263) on the augmented matrix A = [K | F] (B matrices, n rows,
 * ncols = n + m columns, row-major, lda, strideA): for the block rows [j, j+nb) whose diagonal block has already been factored
 * (W = the inverse of its Cholesky factor, from roma_chol_diag_block or from the previous call), with e = j + nb:
 *   R (nb x (ncols - e), ldr, strideR)  <-  W A[j:j+nb, e:]            (= [L[e:, j:j+nb]^T | Y[j:j+nb]]: needed by the back substitution)
 *   A[e:n, e:ncols]                     -=  R[:, :n-e]^T R             (only the 64 x 64 tiles on or right of the diagonal are updated)
 *   the next diagonal block A[e:e+nbn, e:e+nbn], nbn = min(64, n - e), is replaced by its Cholesky factor (lower triangle) and
 *   Wn (nbn x nbn, ldwn, strideWn) receives its inverse; info as in roma_chol_diag_block with info_base for THAT block.
 * Wn may be NULL (no factorisation; the last block).  One launch instead of the three per block of the GEMM formulation. */
int roma_chol_step(float* A, int lda, long strideA, int n, int ncols, int j, int nb, const float* W, int ldw, long strideW, float* R,
                   int ldr, long strideR, float* Wn, int ldwn, long strideWn, int* info, int info_base, int B, void* stream);

/* One fused step of a triangular substitution with the finished factor of the same solve.  R holds ALL panels written by
 * roma_chol_step: panel k of matrix b starts at R + b*strideRb + k*strideRs (nb rows, leading dimension ldr); its first
 * n - e_k columns, e_k = min((k+1) nb, n), are L[e_k:, block k]^T.  T is the right-hand side, consumed in place: block row i of
 * matrix b at T + b*strideTb + i*strideTs (leading dimension ldt), shifted by n - e_i columns when t_in_panel != 0 (the layout in
 * which roma_chol_step leaves Y: T = R, strideTs = strideRs, ldt = ldr).  W = the inverse factor of diagonal block s.
 *   dir < 0 (back substitution L^T X = T; call with s = S-1 down to 0):
 *       X[s nb : s nb + w_s, :] <- W^T T_s;   T_i -= L[block s, block i]^T X_s  for every i < s
 *   dir > 0 (forward substitution L Y = T for a NEW right-hand side — iterative refinement; call with s = 0 up to S-1):
 *       X[s nb : s nb + w_s, :] <- W T_s;     T_i -= L[block i, block s] Y_s    for every i > s
 * X: n x m, ldx, strideX.  nb = 64 (a single-block solve may use any nb <= 64).  One launch per block row instead of the two
 * GEMMs of the GEMM formulation. */
int roma_chol_subst_step(int dir, const float* W, int ldw, long strideW, const float* R, long strideRb, long strideRs, int ldr, float* T,
                         long strideTb, long strideTs, int ldt, int t_in_panel, float* X, int ldx, long strideX, int n, int m, int nb,
                         int s, int B, void* stream);

/* Multi-head attention forward, softmax(q k^T * scale) v, for the transformers of the path (DINOv2 blocks,
 * romatch/models/transformer/layers/attention.py:48-60; the decoder transformer, transformer/__init__.py:30-46): fp16 / bf16, head
 * dimension 64, no mask except "the first Nk tokens are keys / values" (the callers row-pad the sequence; padded tokens query only).
 * q, k, v, o are addressed as base + b*sb + token*sn + head*sh (element strides; each a multiple of 8, bases 16-byte aligned), so
 * the (B, N, 3, H, 64) output of the qkv projection is read in place and o can be the (B, N, H*64) input of the output projection.
 * Flash-style (no Nq x Nk matrix), fp32 softmax statistics, P rounded to the storage dtype before the second product. */
int roma_attention_fwd(const void* q, const void* k, const void* v, void* o, int B, int H, int Nq, int Nk, int head_dim, long q_sb,
                       long q_sn, long q_sh, long k_sb, long k_sn, long k_sh, long v_sb, long v_sn, long v_sh, long o_sb, long o_sn,
                       long o_sh, float scale, int dtype, void* stream);

/* match() post-processing — matcher.py:656-662, 684-718: certainty attenuation by the coarse scale-16 certainty,
 * sigmoid, zeroing where |flow|>1, clamp, symmetric concat.
 *   flow (2P,2,H,W), cert (2P,1,H,W) fp32 planar: first P = A->B, last P = B->A (forward_symmetric, matcher.py:516-528)
 *   cert16 (2P,1,H16,W16) or NULL (no attenuation).
 *   warp (P,H,2W,4) fp32, certainty (P,H,2W) fp32.   symmetric=0: flow (P,..), warp (P,H,W,4), certainty (P,H,W). */
int roma_match_finalize(const float* flow, const float* cert, const float* cert16, float* warp, float* certainty,
                        int P, int H, int W, int H16, int W16, int symmetric, void* stream);

/* kde — romatch/utils/kde.py:4-12: density[i] = sum_j exp(-|x_i - x_j|^2 / (2 std^2)), x: (N,4) fp32, ref points every
 * `down`-th row, density (N) fp32.  half_mode = 0: fp32 arithmetic (kde(half=False)).  half_mode = 1: x already holds
 * fp16-representable values (the caller's x.half()) and every term goes through the rounding points of the reference's fp16
 * evaluation (torch.cdist's matmul route in fp16, then fp16 **2, /, exp), summed in fp32 (the caller rounds the sum to fp16). */
int roma_kde_density(const float* x, float* density, int N, int down, float std, int half_mode, void* stream);

/* Exponential-race keys for sampling WITHOUT replacement — RegressionMatcher.sample (matcher.py:474-493), both of its
 * torch.multinomial(..., replacement=False) draws:  w_i = (thresh >= 0 && p_i > thresh) ? 1 : p_i  (the "threshold" sample
 * mode, matcher.py:474-477);  key_i = w_i / E_i with E_i = -ln(u_i) i.i.d. Exp(1); the k largest keys are a draw of k items
 * without replacement with probabilities proportional to w (what multinomial does internally).  u_i comes from a counter
 * hash of (seed, stage, c_i) — u = ((fmix32(fmix32(seed ^ stage * 0x9E3779B9) + c_i * 0x9E3779B1) >> 8) + 0.5) / 2^24, fmix32 = the
 * MurmurHash3 finaliser — so a CPU oracle reproduces the draw.  `stage` separates the draws of one sample() call (0 = the
 * certainty draw, 1 = the balanced draw): mixed non-linearly, so no (seed, stage) stream is a shifted copy of another.  c_i = counter[i] (int64, e.g. the item's index in the
 * population the first draw came from: the second draw then does not depend on the ORDER of the first) or i when counter
 * is NULL.  p, keys: (N) fp32; w_i <= 0 (or NaN) gives key 0. */
int roma_race_keys(const float* p, const long* counter, float* keys, long N, float thresh, unsigned seed, unsigned stage,
                   void* stream);

/* Nearest neighbours for RegressionMatcher.match_keypoints (matcher.py:576-591): idx[i] = arg min_j |q_i - r_j|^2 over 2-D points
 * (lowest j on exact ties).  The reference materialises cdist(x_A_to_B, x_B) and compares it with its row / column minima;
 * mutual nearest neighbours only need this arg-min in both directions.  q: (NQ,2), r: (NR,2) fp32; idx: (NQ) int32. */
int roma_nn_argmin(const float* q, const float* r, int* idx, int NQ, int NR, void* stream);

/* Pre-processing on the device — utils.py:165-261 (TupleResize = PIL bicubic, ToTensorScaled, TupleNormalize), bit-identical
 * to the host path.  One pass of PIL's 8-bit resampling (Pillow Resample.c): out = clip8((2^21 + sum_k in[lo+k] * coef[k]) >> 22).
 *   in: uint8 (H, W, C);  axis 1: out (H, out_size, C), axis 0: out (out_size, W, C);
 *   bounds: int32 (out_size, 2) = (first input index, tap count); coef: int32 (out_size, ksize) 22-bit fixed point
 *   (computed on the host like precompute_coeffs / normalize_coeffs_8bpc: roma_amd/preproc.py). */
int roma_resample_u8(const void* in, void* out, int H, int W, int C, int out_size, int axis, const int* bounds,
                     const int* coef, int ksize, void* stream);
/*   uint8 (H, W, 3) -> fp32 (3, H, W): ((v / 255) - mean[c]) / std[c];  mean3, std3 are HOST pointers to 3 floats. */
int roma_normalize_u8(const void* in, float* out, int H, int W, const float* mean3, const float* std3, void* stream);

/* VGG19-BN layer epilogue with BatchNorm folded into the convolution — encoders.py:68-78 (conv -> BN -> ReLU):
 *   x[b,c,:] = max(x[b,c,:] + bias[c], 0) in place on a planar (B,C,HW) map of `dtype`; bias (C) of `dtype`.  B*C <= 65535. */
int roma_bias_relu_nchw(void* x, const void* bias, int B, int C, int HW, int dtype, void* stream);

/* The same epilogue for the VGG19-BN layers that are followed by MaxPool2d(2, 2) (encoders.py:68-78; the feature is captured before
 * each pool): x is updated in place as above and pooled[b,c,y,x] = max of the 2x2 block of the result, (B,C,H/2,W/2) of `dtype`.
 * H and W even. */
int roma_bias_relu_pool2_nchw(void* x, const void* bias, void* pooled, int B, int C, int H, int W, int dtype, void* stream);

/* The same two epilogues on channels-last maps (x: npix x C rows, C a multiple of 8 (16-bit) / 4 (fp32), in place; pooled:
 * (B, H/2, W/2, C)): the layout the VGG19 stack runs in (encoders.py:68-78). */
int roma_bias_relu_nhwc(void* x, const void* bias, long npix, int C, int dtype, void* stream);
int roma_bias_relu_pool2_nhwc(void* x, const void* bias, void* pooled, int B, int C, int H, int W, int dtype, void* stream);

/* ConvRefiner block front half — matcher.py:77-103 (create_block: depthwise 5x5 conv, BatchNorm(eval), ReLU),
 * fused, channels-last.  BN is folded by the caller: y = relu(dwconv(x, w) * scale + shift).
 *   x,y: (B,H,W,pitch) `dtype`; w: (25, C) fp32 tap-major; scale, shift: (C) fp32. */
int roma_dwconv5x5_bn_relu(const void* x, const float* w, const float* scale, const float* shift, void* y,
                           int B, int C, int H, int W, int dtype, int x_pitch, int y_pitch, void* stream);

/* One whole ConvRefiner block, fused, for widths C <= 160 in fp16 / bf16 — matcher.py:77-103 (create_block) as applied
 * 9x per refiner at matcher.py:139-140:  y = bias + relu(dwconv5x5(x, w25) * scale + shift) @ wt^T.
 *   x, y: (B,H,W,pitch) channels-last `dtype` (C channels used; x and y must not alias);
 *   kpad: 32 or 160, the zero-padded channel count of the weight buffers (>= C);
 *   w25: (25, kpad) `dtype` tap-major; scale, shift, bias: (kpad) fp32; wt: (kpad, kpad) `dtype`, wt[n][k] = weight of
 *   input channel k for output channel n (the Conv2d(D, D, 1) weight itself), all zero-padded. */
int roma_refiner_block(const void* x, const void* w25, const float* scale, const float* shift, const void* wt,
                       const float* bias, void* y, int B, int C, int H, int W, int kpad, int dtype, int x_pitch, int y_pitch,
                       void* stream);

/* The same block at D = 576 (the scale-4 refiner), fp16, as ONE kernel: a 128-pixel x 576-channel output tile per workgroup, the
 * depthwise result computed once per pixel and fed to the matrix cores through LDS, the 1x1 weights streamed panel by panel.
 * matcher.py:77-103, 139-140.  x, y: (B,H,W,pitch) channels-last fp16 (must not alias); w25p: the depthwise taps in fp16 (the
 * reference's autocast convolution weights) as laid out by roma_refiner_wide_taps; scale, shift, bias: (D) fp32; wp: the
 * Conv2d(D, D, 1) weight [out][in] re-tiled by roma_refiner_wide_pack.  Both packers are HOST functions (all pointers in host memory):
 *   roma_refiner_wide_pack: wt, wp: D*D 16-bit elements each;
 *   roma_refiner_wide_taps: w25 (25, D) tap-major 16-bit -> w25p, 60*D elements: [D/32 panels][5 tap rows][4 packets][2 half-packets]
 *     [6 pair sets][4 channels][2] — the kernel multiplies horizontally adjacent PIXEL PAIRS with tap pairs (v_dot2_f32_f16): of a tap
 *     row (w0..w4) the even output column takes (w0,w1) (w2,w3) (w4,0), the odd one (0,w0) (w1,w2) (w3,w4). */
int roma_refiner_wide_pack(const void* wt, void* wp, int D);
int roma_refiner_wide_taps(const void* w25, void* w25p, int D);
int roma_refiner_block_wide(const void* x, const void* w25p, const float* scale, const float* shift, const void* wp,
                            const float* bias, void* y, int B, int H, int W, int D, int x_pitch, int y_pitch, int dtype,
                            void* stream);

/* ConvRefiner block back half at mid widths — matcher.py:102 (Conv2d(D, D, 1)) for 32 < D <= 160, fp16 / bf16, on the
 * matrix cores:  y[m][n] = bias[n] + sum_k x[m][k] * wt[n][k];  x, y: (M, pitch) channels-last rows (C used, may alias
 * only if identical); wt: (kpad, kpad) `dtype`, the Conv2d weight itself ([out][in]) zero-padded; bias (kpad) fp32. */
int roma_pointwise_mfma(const void* x, const void* wt, const float* bias, void* y, long M, int C, int kpad, int dtype,
                        int x_pitch, int y_pitch, void* stream);

/* ConvRefiner head + Decoder update fused — matcher.py:141 (out_conv, D -> 3, fp32 on d.float()) and :397-402:
 *   d = bo + x[m,:] @ wo;  flow[b,0] += sx*d0;  flow[b,1] += sy*d1;  cert_out = (cert_in ? cert_in : 0) + d2
 *   x: (B*H*W, pitch) channels-last rows of `dtype` (C channels used), wo (C,3) fp32 row-major, bo (3) fp32,
 *   flow (B,2,H,W) fp32 updated in place, cert_in (B,1,H,W) or NULL, cert_out (B,1,H,W), delta_out (B,3,H,W) or NULL
 *   (the raw out_conv result, for callers that want the reference's (displacement, certainty) return values). */
int roma_refiner_head(const void* x, const float* wo, const float* bo, float* flow, const float* cert_in, float* cert_out,
                      float* delta_out, int B, int H, int W, int C, int pitch, int dtype, float sx, float sy, void* stream);

/* ConvRefiner block back half for NARROW activations — matcher.py:102 (Conv2d(D, D, 1) of create_block) at D <= 32:
 *   y[m][n] = bias[n] + sum_k x[m][k] * wt[k][n],  x,y: (M, pitch) channels-last rows of `dtype`, wt (C,C) fp32 row-major
 *   (in, out), bias (C) fp32.  C a multiple of 8 (fp16/bf16) or 4 (fp32), C <= 32. */
int roma_pointwise_small(const void* x, const float* wt, const float* bias, void* y, long M, int C, int dtype,
                         int x_pitch, int y_pitch, void* stream);

/* Residual add (+ LayerScale) fused with the following LayerNorm — the seam between two transformer half-blocks
 * (transformer/layers/block.py:87-107) and the fp32 token stream of the decoder transformer under autocast
 * (transformer/__init__.py:30-46: cat(fp32 GP posterior, fp16 features) promotes to fp32; LayerNorm is fp32; GEMMs take
 * 16-bit casts):
 *     if (y) x[r,:] = round_to_x_dtype(x[r,:] + ls[:] * y[r,:])      (ls == NULL: plain add), written back in place
 *     out[r,:] = gamma ? LayerNorm(x[r,:]; eps) * gamma + beta : x[r,:]     cast to out_dtype
 *   x (rows, x_stride) of x_dtype; y (rows, y_stride) of y_dtype or NULL; ls, gamma, beta (C) fp32 or NULL; out (rows,
 *   out_stride) of out_dtype.  Supported: fp32 stream with fp32/16-bit branch, 16-bit stream with a branch of the same
 *   dtype.  C a multiple of 8, <= 2048; strides multiples of 8. */
int roma_add_layernorm(void* x, int x_dtype, long x_stride, const void* y, int y_dtype, long y_stride, const float* ls,
                       const float* gamma, const float* beta, void* out, int out_dtype, long out_stride, long rows, int C,
                       float eps, void* stream);

/* TinyRoMa corr_volume + pos_embed fused — tiny.py:241-254, 178-203: for every source pixel the soft-argmax target
 * coordinate over the full correlation row, without materialising the (H1W1 x H0W0) volume.
 *   f0: (B,H0*W0,C), f1: (B,H1*W1,C) row-major fp32 (C a multiple of 16); out (B,2,H0,W0) fp32.
 *   exact != 0: full softmax expectation (tiny.py:201-202); exact == 0: reference fast path (tiny.py:187-198). */
int roma_tiny_corr_posembed(const float* f0, const float* f1, float* out, int B, int C,
                            int H0, int W0, int H1, int W1, int exact, void* stream);

/* JPEG decoding for match() on file paths — romatch/models/matcher.py:606-637, 667-676 (`Image.open(path).convert("RGB")`:
 * PIL -> libjpeg-turbo with its defaults, JDCT_ISLOW + fancy up-sampling).  The entropy-coded segment is decoded on the HOST
 * (roma_jpeg_info, roma_jpeg_entropy_decode: host functions, all pointers in host memory); de-quantisation + inverse DCT, chroma
 * up-sampling and YCbCr -> RGB run on the device (roma_jpeg_reconstruct) and leave a uint8 (height, width, 3) image for
 * roma_resample_u8 / roma_normalize_u8.  Bit-identical to PIL.  Supported: 8-bit sequential (one interleaved scan) and progressive
 * (spectral selection + successive approximation) Huffman streams, 1 or 3 components, 4:4:4 / 4:2:2 / 4:2:0, restart intervals; anything
 * else (CMYK, RGB-stored, 12-bit, arithmetic, lossless) returns ROMA_E_UNSUPPORTED (decode with PIL on the host then).
 *   info[8]: width, height, components, sub-sampling (0: 4:4:4, 1: 4:2:0, 2: 4:2:2, -1: grey), luma blocks per row, luma block rows, chroma
 *            blocks per row, chroma block rows;
 *   coef: int16, (luma + 2 x chroma blocks) x 64, natural order inside a block, blocks of a component in raster order, components
 *         one after the other; qt: 3 x 64 uint16, the component's de-quantisation table in natural order;
 *   planes: device scratch, (luma + 2 x chroma blocks) x 64 bytes; rgb: device, height x width x 3 bytes. */
int roma_jpeg_info(const void* data, long nbytes, int* info);
int roma_jpeg_entropy_decode(const void* data, long nbytes, int16_t* coef, uint16_t* qt);
int roma_jpeg_reconstruct(const int16_t* coef, const uint16_t* qt, void* planes, void* rgb, const int* info, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ROMA_HIP_H */
