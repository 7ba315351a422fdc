/*
 * pfd_hip.h -- C ABI of libpfd_hip.so: the MI355X (gfx950) kernels behind the
 * Prompt-Free-Diffusion denoising hot path (SeeCoder encode -> SD-v1.5 UNet DDIM
 * loop -> AutoKL decode).
 *
 * The reference (SHI-Labs/Prompt-Free-Diffusion) has no FFI: its "operator API" is a
 * Python class registry over stock torch ops.  This header is therefore the native
 * boundary we define underneath that registry; every entry point names the reference
 * op sequence (file:line under the reference tree) it replaces.  The Python host
 * (prompt-free-diffusion_amd/lib/hip/binding.py) binds these with ctypes; see
 * INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions
 *   - all pointers are DEVICE pointers; activations/weights are IEEE fp16 ("f16"),
 *     statistics, latents and schedule scalars are fp32; indices int32/int64 as stated
 *   - activations are token-major / NHWC: element (b, y, x, c) of a [B,H,W,C] image
 *     lives at ((b*H + y)*W + x)*ld + c, ld >= C (a row stride, in elements)
 *   - no allocation, no host sync inside; all work is enqueued on `stream`
 *     (a hipStream_t passed as void*); safe under hipGraph capture
 *   - return 0 on success, negative PFD_E* on error (nothing is launched on error)
 */
#ifndef PFD_HIP_H
#define PFD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFD_OK 0
#define PFD_EINVAL (-1)   /* null pointer / non-positive size / misaligned stride   */
#define PFD_ESHAPE (-2)   /* shape outside what the kernels are built for           */
#define PFD_ELAUNCH (-3)  /* hipGetLastError() != hipSuccess after the launch       */

#define PFD_ABI_VERSION 9

typedef void* pfd_stream_t; /* hipStream_t */

/* epilogue activations of pfd_gemm_f16 */
#define PFD_ACT_NONE 0
#define PFD_ACT_GELU 1  /* exact erf GELU: F.gelu attention.py:51, nn.GELU swin.py:84        */
#define PFD_ACT_RELU 2  /* seecoder.py:65,214                                               */
#define PFD_ACT_SILU 3  /* nn.SiLU openaimodel.py:2631, :218                                */
#define PFD_ACT_GEGLU 4 /* x * gelu(gate), attention.py:49-51; W packed in 64-row blocks     */
                        /* [32 rows of x | 32 rows of gate]; output has N/2 columns          */

int pfd_abi_version(void);
/* last hip error string observed by this library on the calling thread (never NULL) */
const char* pfd_last_error(void);

/* ------------------------------------------------------------------------------------
 * Dense contraction on MFMA (v_mfma_f32_32x32x16_f16), fp32 accumulate.
 *
 *   ksize == 0 : C[m, n] = epi( sum_k A[m*lda + k] * W[n*ldw + k] )            (linear)
 *   ksize  > 0 : implicit-GEMM convolution over an NHWC image A = [B,H,W,Cin] (row
 *                stride lda): m = (b, oy, ox), k = (ky, kx, ci),
 *                iy = oy*stride + ky - pad, ix likewise; out-of-range taps read 0;
 *                with ups=1 the image is first nearest-2x upsampled (gather iy>>1).
 *                W is [N][ksize*ksize*Cin] (tap-major, channel-minor), ldw its row stride.
 *   epi(v) = act(v + bias[n or m] + rowvec[(m / rows_per_rv)*ldrv + n]) + R[m*ldr + n]
 *
 * Replaces: nn.Linear / 1x1 nn.Conv2d everywhere on the path (attention.py:169-176,
 * 47-51, 60-67, 329-347; swin.py:88-90,171-173,322; seecoder.py:74-77,216-218,358;
 * openaimodel.py:2629-2633,217-223,240); F.conv2d 3x3 s1/s2 (openaimodel.py:105,150,
 * 203,229; autokl_modules.py:47-51,66-70,93-108; controlnet.py:165-181) together with
 * the ops the reference runs around them: bias add, `h + emb_out` (openaimodel.py:272),
 * `skip_connection(x) + h` (:274), `attn(...) + x` (attention.py:303-305),
 * F.interpolate(scale 2, nearest) (openaimodel.py:114; autokl_modules.py:54) and
 * GEGLU (attention.py:49-51).
 *
 * Requirements: K % 64 == 0 (linear) or Cin % 64 == 0 (conv); lda/ldw % 8 == 0;
 * A, W 16-byte aligned.  M, N arbitrary (tails are masked).  N % 160 == 0 or N % 128 == 0 is
 * served by the LDS-DMA wide-tile kernels (160- / 128-wide tiles), anything else by the
 * register-staged 128x128 kernel.
 * PFD_ACT_GEGLU weight packing: when N % 160 == 0 (wide-tile kernel) every PAIR of outputs
 * stores its 2 x-rows then its 2 gate-rows (x0 x1 g0 g1 | x2 x3 g2 g3 | ...: the four accumulator
 * columns one lane owns); otherwise (N % 128 == 0) every group of 32 outputs stores 32 x-rows then
 * 32 gate-rows.  bias is packed the same way.
 * ---------------------------------------------------------------------------------- */
typedef struct PfdGemmDesc {
  const void* A;
  const void* W;
  const void* bias;   /* f16 [N] (or [M] if bias_per_row), may be NULL */
  const void* rowvec; /* f16, may be NULL                               */
  const void* R;      /* f16 residual, may be NULL                      */
  void* C;            /* f16 [M, ldc]; N/2 columns when act==GEGLU       */
  int64_t lda, ldw, ldr, ldc, ldrv;
  int32_t M, N, K;
  int32_t rows_per_rv;  /* >= 1 */
  int32_t act;          /* PFD_ACT_*  */
  int32_t bias_per_row; /* 0/1 */
  /* implicit-conv geometry; ksize==0 -> plain GEMM and the rest is ignored */
  int32_t ksize, stride, pad, ups;
  int32_t B, H, Wd, Cin; /* input image (before the optional upsample) */
  int32_t Ho, Wo;        /* output image; M must equal B*Ho*Wo          */
  /* optional scratch for split-K (small M*N, huge K: the 8x8 / 16x16 UNet levels); the
   * library never allocates.  NULL / 0 = never split.  Must not be shared by concurrent streams.
   * A problem is split only when splits * M * N * 4 <= ws_bytes (the rule of every ABI-9 build).  What the library
   * keeps in it is private: since round 6 the K-range partial sums are stored rounded to f16 and summed in fp32, in
   * slab order, by the reduction launch (deterministic; the same freedom the reference's default
   * torch.backends.cuda.matmul.allow_fp16_reduced_precision_reduction = True gives its own split-K GEMMs); a
   * partial sum beyond +-65504 becomes +-inf exactly as an f16 result of that magnitude would. */
  void* ws;
  size_t ws_bytes;
  /* optional transposed tail (ABI 3): with Ct != NULL the output columns n >= n_split are not
   * written to C but, transposed, to Ct[(n - n_split) * ldct + m] (+ bias[n] only).  This is how the
   * self-attention q | k | v projection (attention.py:169-176) is ONE launch over the shared
   * activation: q | k land token-major in C, v lands as the V^T [inner, tokens] operand
   * pfd_attention_f16 consumes.  Wide-tile path only: plain GEMM (ksize == 0), N a multiple of the
   * tile width T (160 if N % 160 == 0, else 128 if N % 128 == 0), n_split % T == 0,
   * ldct % 8 == 0, Ct 16-byte aligned, act == NONE, rowvec == R == NULL,
   * bias_per_row == 0; anything else is PFD_ESHAPE (there is no slow path behind it). */
  void* Ct;
  int64_t ldct;
  int32_t n_split;
  /* weight layout (ABI 7): 0 = row-major [N][ldw]; 1 = K-tile-contiguous: with T = 160 if N % 160 == 0 else 128
   * (N % 128 == 0), element (n, k) lives at (((n / T) * (K / 64) + k / 64) * T + n % T) * 64 + k % 64 -- the
   * (T x 64) weight tile a block stages per K step is one contiguous T x 128-byte run of memory instead of T pieces a
   * whole weight row apart (cold weight streams ran at ~1 TB/s on the row-major layout).  Wide-tile kernels only;
   * with w_tiled != 0 a shape they do not take is PFD_ESHAPE.  ldw is ignored. */
  int32_t w_tiled;
  /* optional GroupNorm(+SiLU) prologue (ABI 6; `GroupNorm32 -> SiLU -> conv3x3` of ResBlock._forward,
   * openaimodel.py:200-226, 254-274): with gn_table != NULL the convolution reads
   *     act(x[b, y, x, c] * gn_table[b][0][c] + gn_table[b][1][c])   rounded to f16
   * instead of x, where x is the virtual channel concat [A (gn_c1 channels, row stride lda) | A2 (Cin - gn_c1
   * channels, row stride lda2)] (A2 may be NULL when gn_c1 == Cin) and the zero padding pads the NORMALISED image.
   * gn_table is the f32 [B, 2, Cin] array (scale plane, shift plane) pfd_groupnorm_table_f16 writes; gn_act is PFD_ACT_NONE or PFD_ACT_SILU.
   * Served by the 3x3 patch kernel only: ksize 3, stride 1, pad 1, no upsample, Wd in {16, 32, 64},
   * H % (256 / Wd) == 0, N % 160 == 0, gn_c1 % 64 == 0, act != GEGLU; anything else is PFD_ESHAPE (callers run
   * pfd_groupnorm_f16 and a plain convolution instead; there is no slow path behind this one). */
  const void* gn_table;
  const void* A2;
  int64_t lda2;
  int32_t gn_c1;
  int32_t gn_act;
  /* LayerNorm folded into the contraction (ABI 7; BasicTransformerBlock, attention.py:294-306: every LayerNorm feeds
   * a Linear with nothing in between -- norm1 -> to_q|to_k|to_v, norm2 -> to_q, norm3 -> GEGLU.proj).  With
   * ln_stats != NULL the launch computes, for A = the UN-normalised tokens x and W = the gamma-scaled weight W o gamma,
   *     LN(x) W^T + b  =  rstd_m * (x (W o gamma)^T)[m, n]  -  rstd_m * mean_m * s_n  +  b'_n
   * with s_n = sum_k (W o gamma)[n, k] (ln_colsum, f32 [N], of the f16-rounded products) and
   * b'_n = sum_k beta_k W[n, k] + b_n passed as `bias`; the affine map is applied to the fp32 accumulator in front
   * of the rest of the epilogue (bias, activation / GEGLU, transposed tail, split-K reduce alike).
   * ln_stats is the f32 [M][ln_parts][2] array of PARTIAL row sums (sum x, sum x^2) over the 160-column slices of A
   * (ln_parts = K / 160 <= 8; mean and rstd are formed from their totals with ln_eps): it is what the PRODUCER of x
   * writes when its own descriptor has ln_out != NULL (f32 [M][N / 160][2], N % 160 == 0, act != GEGLU, no Ct) --
   * the out-projection / proj_in launch that stores x emits the statistics of the rows it holds, so no LayerNorm
   * launch and no extra pass over x exist.  pfd_ln_rowstats_f16 writes the same array for tensors that no such
   * launch produced.  Wide-tile linear path only (N % 160 == 0 or N % 128 == 0, ksize == 0); anything else is
   * PFD_ESHAPE (callers run pfd_layernorm_f16 and a plain GEMM instead). */
  const void* ln_stats;
  const void* ln_colsum;
  int32_t ln_parts;
  float ln_eps;
  void* ln_out;
  /* Two-source contraction and zero rows (ABI 8; plain GEMM, ksize == 0, wide-tile kernels only: N % 160 == 0 or
   * N % 128 == 0, anything else is PFD_ESHAPE and callers run the two-launch forms).
   * k_split > 0: the operand is the virtual COLUMN concat [A (k_split columns, row stride lda) | A2 (K - k_split
   *   columns, row stride lda2)], k_split % 64 == 0 -- the 1x1 skip convolution of an output-half ResBlock over
   *   `torch.cat([h, skip], dim=1)` (openaimodel.py:274 after pfd.py:356) as ONE launch instead of a GEMM over h and a
   *   second GEMM over the skip tensor that re-reads and re-writes the result (gn_c1 / gn_table are not involved).
   * zero_rows > 0: output rows m < zero_rows have an all-zero operand row (they are never read; A / A2 point at the
   *   data of row zero_rows), so their result is epi(0) = act(bias + rowvec) + R, and tiles that lie entirely below
   *   zero_rows skip their K loop -- the cross-attention out-projection of a CFG batch whose unconditional context is
   *   all zero (app.py:236: K = V = 0, so `to_out(attn) + x` of those samples is `to_out.bias + x`, attention.py:
   *   178-201) in ONE launch with the conditional half, statistics (ln_out) included.  Not with ln_stats. */
  int32_t k_split;
  int32_t zero_rows;
  /* GroupNorm statistics of the OUTPUT from the launch that stores it (ABI 8; `GroupNorm32 -> SiLU -> conv` /
   * SpatialTransformer.norm read what a convolution or out-projection just wrote: openaimodel.py:200-226, 254-274,
   * attention.py:83-84, 352-371): with gn_out != NULL the store pass (or the split-K reduction) also writes, for every
   * slab of 64 consecutive output rows and every group of N / 32 output channels, (sum x, sum x^2) of the f16 values it
   * stored:  gn_out[(slab * (N / 160) + n / 160) * 16 + (n % 160) / (N / 32)]  (float2; slab = m / 64; 16 slots per
   * 160-column tile of which 160 / (N / 32) are used).  pfd_groupnorm_pstats_f16 normalises from these sums: the
   * statistics pass over the tensor (a third of the two-launch GroupNorm) disappears.  Fixed summation order
   * (deterministic); per-sample as long as a slab does not straddle two samples (rows per sample % 64 == 0).
   * Wide-tile kernels only: N % 160 == 0, N / 32 >= 8 and a divisor of 160 (N = 320 | 640 | 1280), M % 64 == 0,
   * act != GEGLU, no Ct / ln_out / bias_per_row; anything else is PFD_ESHAPE. */
  void* gn_out;
  /* GroupNorm(32 groups)(+SiLU) of the OUTPUT inside the split-K reduction (ABI 9).  At the 8^2 / 16^2 UNet levels every
   * 3x3 convolution splits its contraction (M <= 2048 rows cannot fill 256 CUs otherwise) and a reduction launch sums the
   * partial-sum slabs, applies the epilogue and stores the f16 result -- which a single-launch GroupNorm then reads once more to
   * normalise it (`h = in_layers(x) + emb_out; h = out_layers(h)`: GroupNorm32 -> SiLU -> conv, openaimodel.py:254-272;
   * eps 1e-5).  With gnf_y != NULL the reduction is done by blocks that own one (sample, group) slab of the output
   * (gnf_rows rows x N / 32 channels): they form the epilogue's f16 values, their statistics and
   *     gnf_y[m, n] = act((out[m, n] - mean) * rstd * gnf_gamma[n] + gnf_beta[n])      (act = NONE | SILU)
   * in one launch; the raw result is ALSO stored to C unless gnf_skip_raw != 0 (a tensor only its GroupNorm reads: the
   * first convolution of a ResBlock).  Same arithmetic, in the same order, as the plain reduction followed by
   * pfd_groupnorm_f16 on its output: the same bits wherever pfd_groupnorm_f16 takes its single-launch form (N >= 1024);
   * at N = 640 (20 channels per group: the 32x32 UNet level, served since round 6) the raw result is still bitwise the
   * two-call form's and the normalised one agrees with it up to the summation order of the statistics (last-bit
   * roundings in < 0.1 % of the elements).  Served only where the library splits K (it decides; M small) and
   * N % 160 == 0, (N / 32) % 4 == 0, 20 <= N / 32 <= 256, gnf_rows > 0, M % gnf_rows == 0, gnf_rows * (N / 128) <= 8192, at least four
   * samples ((M / gnf_rows) * 32 >= 128 slabs), ws != NULL, gnf_ldy % 4 == 0, gnf_y 8-byte aligned, gnf_act in {NONE, SILU},
   * rowvec == NULL or one row vector per sample (rows_per_rv % gnf_rows == 0 or rows_per_rv >= M), act != GEGLU, no Ct /
   * ln_stats / ln_out / gn_out / bias_per_row / res_rows; anything else -- including a problem the library would not split
   * -- is PFD_ESHAPE with NOTHING launched: callers then run the two-call form (there is no slow path behind this one).  A
   * forced variant (pfd_gemm_f16_ex) the wide-tile dispatcher does not know is PFD_EINVAL, as for every other launch. */
  const void* gnf_gamma; /* f16 [N] */
  const void* gnf_beta;  /* f16 [N] */
  void* gnf_y;           /* f16 [M, gnf_ldy] */
  int64_t gnf_ldy;
  float gnf_eps;
  int32_t gnf_act;
  int32_t gnf_rows;      /* rows per sample (Ho * Wo of a convolution) */
  int32_t gnf_skip_raw;
  /* Residual stored ONCE for a doubled batch (ABI 9).  The classifier-free-guidance batch is [x | x] with one timestep
   * (ddim.py:145-149 `torch.cat([x] * 2)`), so everything in front of the first cross-attention is identical for the two
   * halves and is computed once (lib/model_zoo/attention.py, cfg_pair); the launches that re-join the halves -- the
   * cross-attention out-projection `attn2(...) + x` and `proj_out(...) + x_in` of the first SpatialTransformer
   * (attention.py:303-305, 370) -- then add a residual that exists as ONE copy: with 0 < res_rows < M output row m adds
   * R[m - res_rows] for m >= res_rows (M / 2 <= res_rows: at most one wrap).  0 (or M) = one residual row per output row.
   * Wide-tile kernels only (PFD_ESHAPE otherwise); not together with gnf_y. */
  int32_t res_rows;
} PfdGemmDesc;
int pfd_gemm_f16(const PfdGemmDesc* d, pfd_stream_t stream);
/* Same, with the kernel variant forced (tests and tuning only); 0 = the library's heuristic.
 *   tile in {22, 21, 12, 11}: register-staged kernel, (64*TM) x (64*TN) x 64 tile, tile = 10*TM+TN
 *   tile = 1000 + 100*v + s : LDS-DMA wide-tile kernel (needs N % 160 == 0 or N % 128 == 0), split-K factor s in 0..8
 *          (0 = heuristic), e.g. 1000 + 4400 + 2 = 5402.  v (0 = heuristic):
 *            44 / 24 / 22     256 / 128 / 64 rows x (160 | 128) columns, two operand stages (4 x 2 / 2 x 2 wave layouts)
 *            82 / 41          the 128- / 64-row tiles on eight waves (160-wide only)
 *            25 / 83, 23 / 43 the same tiles on 3- / 4-stage operand rings with counted waits (160-wide only)
 *            84               256 x 320 tile, GEGLU projections only
 *            48 / 47          256 rows with 4 dedicated LDS-DMA loader waves, two / three operand stages
 *            99 / 98 / 96     the 3x3 patch kernel: 8-wave form / loader waves with two weight stages / three (the default)
 *          A variant that does not serve the launch (shape, epilogue, ABI 8 / 9 fields) is PFD_ESHAPE. */
int pfd_gemm_f16_ex(const PfdGemmDesc* d, int32_t tile, pfd_stream_t stream);

/* GEGLU weight packing the serving kernel expects for a projection of N = 2*dim_out output rows: the
 * returned g is the interleave granularity -- packed rows come in groups [x(g rows) | gate(g rows)]
 * (g = 2: the four accumulator columns one lane of the wide-tile kernel owns; g = 32: one 32-column MFMA
 * tile pair of the 128x128 kernel).  pfd_gemm_f16 with act = PFD_ACT_GEGLU returns PFD_ESHAPE rather than
 * fall through to a kernel with a different packing.  (GEGLU, attention.py:44-51.) */
int32_t pfd_gemm_geglu_group(int32_t N);

/* ------------------------------------------------------------------------------------
 * Fused scaled-dot-product attention, online softmax in fp32 (never materialises the
 * score matrix).  O[b,i,h,:] = softmax_j(scale * <Q[b,i,h,:], K[b,j,h,:]>) . V[b,j,h,:]
 *   Q  element (b,i,h,d) at Q [b*q_bs + i*ldq + h*D + d]
 *   K  element (b,j,h,d) at K [b*k_bs + j*ldk + h*D + d]
 *   Vt element (b,j,h,d) at Vt[(h*D + d)*ldvt + b*vt_bs + j]     (V transposed: the
 *      producer GEMM writes V^T = Wv . X^T directly, so no transpose pass exists)
 *   O  element (b,i,h,d) at O [b*o_bs + i*ldo + h*D + d]
 * Replaces CrossAttention.forward attention.py:178-201 (einsum, *scale, softmax,
 * einsum and the two rearranges), xformers memory_efficient_attention :264, and
 * nn.MultiheadAttention's core in seecoder.py:133,186; with D = 512, H = 1 also the VAE
 * mid-block AttnBlock.forward autokl_modules.py:186-197 (bmm, *C^-1/2, softmax, bmm).
 * D in {40, 80, 96, 160}; ldq/ldk/ldo % 8 == 0, ldvt/vt_bs % 8 == 0; Nq, Nk arbitrary.
 * D = 512: H must be 1 and Nk a multiple of 32 (PFD_ESHAPE otherwise); Nq arbitrary.
 * Every V^T row must be readable up to the next multiple of 8 keys (the values there are masked).
 * ---------------------------------------------------------------------------------- */
typedef struct PfdAttnDesc {
  const void* Q;
  const void* K;
  const void* Vt;
  void* O;
  int64_t ldq, ldk, ldvt, ldo;
  int64_t q_bs, k_bs, vt_bs, o_bs;
  int32_t B, H, Nq, Nk, D;
  float scale;
} PfdAttnDesc;
int pfd_attention_f16(const PfdAttnDesc* d, pfd_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Swin (shifted-)window attention core for one image: pad to multiples of `ws`, cyclic
 * roll by -shift, window partition, q*scale, QK^T + relative-position bias + shift mask
 * (-100, not -inf), softmax, .V, window reverse, roll back, crop -- all as index math.
 *   qkv   f16 [B*H*W, 3C]: LN(x) . Wqkv^T + b   (columns: q | k | v, head-major inside)
 *   qkv_bias f16 [3C]: value of a zero-padded token's q|k|v (padding happens after
 *            norm1 and before the qkv Linear, swin.py:266-273,186)
 *   rpb   f16 [(2ws-1)^2, nH] relative_position_bias_table (swin.py:155-156,192-195)
 *   out   f16 [B*H*W, C]  (input of WindowAttention.proj)
 * Replaces swin.py:266-302 + 186-207 + the mask construction 421-440.
 * head_dim must be 32, ws 12 (Swin-L, configs/model/swin.yaml:18-29).
 * ---------------------------------------------------------------------------------- */
typedef struct PfdSwinAttnDesc {
  const void* qkv;
  const void* qkv_bias;
  const void* rpb;
  void* out;
  int32_t B, H, W, C, nH, ws, shift;
  float scale;
} PfdSwinAttnDesc;
int pfd_swin_window_attention_f16(const PfdSwinAttnDesc* d, pfd_stream_t stream);

/* ------------------------------------------------------------------------------------
 * GroupNorm (+ optional SiLU) over NHWC, statistics in fp32.  The input may be the
 * channel concatenation of two tensors (skip connections, pfd.py:356 / :519) which is
 * never materialised: x2 may be NULL (C2 = 0).
 *   y[b,p,c] = act( (x[b,p,c] - mean[b,g]) * rstd[b,g] * gamma[c] + beta[c] )
 * ws: fp32 workspace of at least pfd_groupnorm_ws_bytes(B, C1+C2, HW) bytes.
 * Replaces GroupNorm32 + SiLU (openaimodel.py:200-202,224-226,2732-2734; eps 1e-5),
 * Normalize (attention.py:83-84, autokl_modules.py:38-39; eps 1e-6) + nonlinearity
 * (autokl_modules.py:33-35), nn.GroupNorm(32, .) (seecoder.py:359,383).
 * C1, C2 % 8 == 0; (C1+C2) % G == 0.
 * ---------------------------------------------------------------------------------- */
size_t pfd_groupnorm_ws_bytes(int32_t B, int32_t C, int32_t HW);
/* GroupNorm(+SiLU) from the statistics its producers emitted (PfdGemmDesc.gn_out) -- ONE launch, no pass over the input
 * for statistics.  x1 / x2 as in pfd_groupnorm_f16 (virtual channel concat); st1 / st2 are the gn_out arrays of the
 * launches that wrote x1 / x2 (float2 [B * HW / 64][C_src / 160][16]; st2 NULL iff C2 == 0).  A group of this norm
 * ((C1 + C2) / G channels) must be a whole number of producer groups (C_src / 32 channels) of ONE source and the shape must
 * be one pfd_groupnorm_f16 would serve with its two-launch form: pfd_groupnorm_takes_pstats says so (1 / 0); otherwise
 * PFD_ESHAPE and callers run pfd_groupnorm_f16.  Same normalisation arithmetic as pfd_groupnorm_f16; the statistics are
 * sums in a different (fixed) order, i.e. equal up to fp32 rounding. */
int32_t pfd_groupnorm_takes_pstats(int32_t B, int32_t C1, int32_t C2, int32_t HW, int32_t G);
int pfd_groupnorm_pstats_f16(const void* x1, int32_t C1, int64_t ldx1, const void* st1, const void* x2, int32_t C2,
                             int64_t ldx2, const void* st2, const void* gamma, const void* beta, void* y, int64_t ldy,
                             int32_t B, int32_t HW, int32_t G, float eps, int32_t act, pfd_stream_t stream);

/* GroupNorm statistics only: writes table[b][0][c] = rstd * gamma[c], table[b][1][c] = beta[c] - mean * rstd * gamma[c]
 * (f32 [B, 2, C1+C2], 16-byte aligned) for PfdGemmDesc.gn_table -- same arguments and workspace as pfd_groupnorm_f16 minus y / act. */
int pfd_groupnorm_table_f16(const void* x1, int32_t C1, int64_t ldx1, const void* x2, int32_t C2, int64_t ldx2,
                            const void* gamma, const void* beta, void* table, int32_t B, int32_t HW, int32_t G,
                            float eps, void* ws, size_t ws_bytes, pfd_stream_t stream);
int pfd_groupnorm_f16(const void* x1, int32_t C1, int64_t ldx1, const void* x2, int32_t C2,
                      int64_t ldx2, const void* gamma, const void* beta, void* y, int64_t ldy,
                      int32_t B, int32_t HW, int32_t G, float eps, int32_t act /*NONE|SILU*/,
                      void* ws, size_t ws_bytes, pfd_stream_t stream);

/* LayerNorm over the last dim of a [M, C] token matrix, fp32 statistics.
 * gather4 != 0: PatchMerging gather -- row r=(b,oy,ox) of the normalised matrix is the
 * concat [x(2oy,2ox) | x(2oy+1,2ox) | x(2oy,2ox+1) | x(2oy+1,2ox+1)] of a [B,H,W,C/4]
 * image (zero beyond H/W when odd) (swin.py:334-348).
 * Replaces nn.LayerNorm (attention.py:294-296, swin.py:241,247,323,600, seecoder.py:72,
 * 79,113,163,219).  C % 8 == 0, C <= 8192. */
int pfd_layernorm_f16(const void* x, int64_t ldx, const void* gamma, const void* beta, void* y,
                      int64_t ldy, int32_t M, int32_t C, float eps, int32_t gather4, int32_t B,
                      int32_t H, int32_t W, pfd_stream_t stream);

/* Partial row statistics of a [M, C] f16 token matrix in the layout PfdGemmDesc.ln_stats takes:
 * out[m][p] = (sum, sum of squares) of x[m, 160 p .. 160 p + 159], f32 [M][C / 160][2].  C % 160 == 0, C <= 1280. */
int pfd_ln_rowstats_f16(const void* x, int64_t ldx, int32_t M, int32_t C, void* out, pfd_stream_t stream);

/* Row softmax with pre-scale: y[r,:] = softmax(scale * x[r,:]) (fp32 math), [R, N] f16.
 * Used by the VAE mid-block single-head attention (autokl_modules.py:186-197). */
int pfd_softmax_rows_f16(const void* x, int64_t ldx, void* y, int64_t ldy, int32_t R, int32_t N,
                         float scale, pfd_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Boundary / elementwise kernels (HBM-bound)
 * ---------------------------------------------------------------------------------- */
/* NCHW (fp32 if src_f32 else f16) -> NHWC f16, y = x*mul + add.  `rep` > 1 writes the
 * batch `rep` times back to back (CFG batch doubling torch.cat([x]*2), ddim.py:145). */
int pfd_nchw_to_nhwc_f16(const void* x, int32_t src_f32, void* y, int32_t B, int32_t C, int32_t H,
                         int32_t W, float mul, float add, int32_t rep, pfd_stream_t stream);
/* NHWC f16 -> NCHW (fp32 if dst_f32 else f16), y = clamp(x*mul + add, lo, hi).
 * With mul=.5, add=.5, lo=0, hi=1 this is AutoencoderKL.decode's tail (autokl.py:47,53). */
int pfd_nhwc_to_nchw(const void* x, void* y, int32_t dst_f32, int32_t B, int32_t C, int32_t H,
                     int32_t W, float mul, float add, float lo, float hi, pfd_stream_t stream);
/* im2col for convolutions whose Cin is not a multiple of 64 (UNet stem 4ch, ControlNet
 * hint encoder, VAE conv_in, Swin patch-embed 4x4/4): col[m, (ky*k+kx)*Cin + ci], zero
 * padded to Kpad columns.  NHWC f16 in. */
int pfd_im2col_f16(const void* x, int64_t ldx, void* col, int32_t B, int32_t H, int32_t W,
                   int32_t Cin, int32_t ksize, int32_t stride, int32_t pad, int32_t Ho, int32_t Wo,
                   int32_t Kpad, pfd_stream_t stream);
/* sinusoidal timestep embedding, fp32 math, cos half first (diffusion_utils.py:131-151),
 * cast to f16 (pfd.py:486).  t: int64 [B]; out f16 [B, dim]. */
int pfd_timestep_embedding_f16(const int64_t* t, void* out, int32_t B, int32_t dim,
                               float max_period, pfd_stream_t stream);
/* Classifier-free-guidance combine + DDIM update, fp32 (ddim.py:150-151,159-171):
 *   e = e_u + s*(e_c - e_u);  pred_x0 = (x - sqrt(1-a_t) e)/sqrt(a_t)
 *   x_prev = sqrt(a_prev) pred_x0 + sqrt(1 - a_prev - sigma^2) e + sigma*noise
 * eps: NHWC f16 [nb*B, h, w, C] (uncond batch first; nb=1 -> e = s*eps, ddim.py:142-143)
 * x, x_prev, pred_x0: NCHW fp32 [B,C,h,w]; noise may be NULL (eta == 0).
 * coef: device fp32 [5] = {a_t, a_prev, sigma_t, sqrt(1-a_t), guidance scale}.
 * xin_next (may be NULL): NHWC f16 [rep*B,h,w,C] = x_prev duplicated `rep` times: rep = nb is the
 * next step's CFG-doubled UNet input (ddim.py:145 fused); rep = 1 when the UNet shares the layers in
 * front of its first cross-attention between the two halves of the pair. */
int pfd_cfg_ddim_step(const void* eps, int32_t nb, const float* x, const float* noise,
                      const float* coef, float* x_prev, float* pred_x0, void* xin_next, int32_t rep,
                      int32_t B, int32_t C, int32_t h, int32_t w, pfd_stream_t stream);
/* y = a + b (f16, fp32 add), n elements; b may be NULL (copy). */
int pfd_add_f16(const void* a, const void* b, void* y, int64_t n, pfd_stream_t stream);
/* y = alpha*a + beta*b (f16 storage, fp32 math), n elements; b may be NULL (y = alpha*a).  The
 * ratio-weighted sum of the per-context SpatialTransformer outputs of multi-context sampling
 * (pfd.py:375-380 `h = h + module(x, emb, c) * r`). */
int pfd_axpby_f16(const void* a, float alpha, const void* b, float beta, void* y, int64_t n,
                  pfd_stream_t stream);
/* y[r, c] = x[r, c] + v[c] for a [R, C] f16 matrix (level/positional embeddings,
 * seecoder.py:402,515). */
int pfd_add_rowvec_f16(const void* x, int64_t ldx, const void* v, void* y, int64_t ldy, int32_t R,
                       int32_t C, pfd_stream_t stream);

/* The same with the partial row sums of y in the layout PfdGemmDesc.ln_stats takes (f32 [R][C / 160][2]; C % 160 == 0,
 * C <= 1280), in the summation order of pfd_ln_rowstats_f16 / the statistics-emitting GEMM epilogue: the `x + to_out.bias`
 * rows of the zero-context cross-attention shortcut feed the next folded LayerNorm without another pass. */
int pfd_add_rowvec_lnstats_f16(const void* x, int64_t ldx, const void* v, void* y, int64_t ldy, int32_t R, int32_t C,
                               void* stats, pfd_stream_t stream);

/* NHWC f16 -> packed uint8 image(s) [B, H, W, C]: v = clamp(x*mul + add, 0, 1), then uint8(v * 255)
 * truncated -- bit for bit what the reference's output stage does to the decoded image:
 * AutoencoderKL.decode's (x+1)/2 + clamp (autokl.py:47,53) followed by torchvision's ToPILImage, i.e.
 * `pic.mul(255).byte()` (app.py:273-275).  f16_image != 0: the image tensor is fp16 (fp16 model, app.py
 * default): v and the product v*255 are each rounded to f16 first, as on that tensor; 0: fp32 image.
 * The request front-end hands the bytes to the client without a float image ever leaving the device. */
int pfd_image_u8_f16(const void* x, void* y, int64_t n, float mul, float add, int32_t f16_image,
                     pfd_stream_t stream);

/* y = act(x) elementwise (act in NONE|GELU|RELU|SILU), n f16 elements.  The SiLU in front
 * of every ResBlock emb_layers Linear (openaimodel.py:217-218) applied once to the shared
 * time embedding, and nonlinearity() on its own (autokl_modules.py:33-35). */
int pfd_act_f16(const void* x, void* y, int64_t n, int32_t act, pfd_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Measurement hooks (no reference counterpart: the reference has no profiling).
 * pfd_prof_enable(1) resets the counters and makes the GEMM/conv, attention and GroupNorm
 * entry points bracket their kernel with a HIP event pair on the launch stream;
 * pfd_prof_read waits for outstanding events and returns, for one kernel bucket, the summed
 * kernel time [ms], launch count and the ALGORITHMIC flops / HBM bytes of those launches
 * (2MNK; operands and result once).  Not for use under hipGraph capture.
 * ---------------------------------------------------------------------------------- */
int pfd_prof_enable(int32_t on);
int pfd_prof_read(int32_t bucket, double* ms, int64_t* launches, double* flops, double* bytes);
const char* pfd_prof_bucket_name(int32_t bucket);
int pfd_prof_num_buckets(void);

#ifdef __cplusplus
}
#endif
#endif /* PFD_HIP_H */
