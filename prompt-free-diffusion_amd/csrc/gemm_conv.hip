// pfd_gemm_f16: linear / 1x1 / 3x3 implicit-GEMM convolution on v_mfma_f32_32x32x16_f16.
//
// Tile: BM x BN x 64, BM = 64*TM, BN = 64*TN; 256 threads = 4 waves as 2(M) x 2(N); each wave
// owns TM x TN MFMA tiles of 32x32 (fp32 accumulators live in the unified VGPR/AGPR file).
// Operands are staged global -> registers -> LDS (two LDS buffers, one barrier per K step;
// the next tile's global loads are in flight during the current tile's MFMAs).  Register
// staging rather than LDS-DMA because the convolution's A operand is a gather with zero
// fill at the image border.  LDS rows are 64 halfs + 8 pad = 144 B = 9 x 16 B: an odd
// number of 16-B slots makes every ds_read_b128 lane group hit 16 distinct slots.
// Epilogue: accumulators (+bias, +row vector, activation / GEGLU) go to an fp32 LDS image,
// then every thread adds the residual and writes whole 16-B rows segments.
//
// HBM bytes per tile step: (BM + BN) * 64 * 2; MFMA flops: 2 * BM * BN * 64.
#include "pfd_common.h"

namespace {

__device__ __attribute__((aligned(16))) half_t g_zero_halves[8];   // zero-initialised: source of absent epilogue operands
__device__ __forceinline__ const half_t* pfd_zero_halves() { return g_zero_halves; }

constexpr int BK = 64;
constexpr int LDS_LD = BK + 8;  // halfs per LDS row

struct GemmParams {
  const half_t* A;
  const half_t* W;
  const half_t* bias;
  const half_t* rowvec;
  const half_t* R;
  half_t* C;
  long lda, ldw, ldr, ldc, ldrv;
  int M, N, K;
  int rows_per_rv, act, bias_per_row;
  int ksize, stride, pad, ups;
  int B, H, Wd, Cin, Ho, Wo;
  int tiles_m, tiles_n;
};

template <int TM, int TN>
struct Cfg {
  static constexpr int BM = 64 * TM;
  static constexpr int BN = 64 * TN;
  static constexpr int A_CH = BM * 8 / 256;  // 16-B chunks per thread for the A tile
  static constexpr int B_CH = BN * 8 / 256;
  static constexpr int STAGE_HALFS = (BM + BN) * LDS_LD;
  static constexpr int MAIN_BYTES = 2 * STAGE_HALFS * 2;
  static constexpr int EPI_LD = BN + 4;  // floats per staged output row
  static constexpr int EPI_BYTES = BM * EPI_LD * 4;
  static constexpr int SMEM = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
};

template <int TM, int TN, bool CONV>
__global__ __launch_bounds__(256) void gemm_conv_kernel(const GemmParams p) {
  using C_ = Cfg<TM, TN>;
  constexpr int BM = C_::BM, BN = C_::BN;
  __shared__ __attribute__((aligned(16))) char smem[C_::SMEM];
  half_t* lds = reinterpret_cast<half_t*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;

  const int nblk = p.tiles_m * p.tiles_n;
  const int t = xcd_remap(blockIdx.x, nblk);
  const int tile_m = t / p.tiles_n;
  const int tile_n = t - tile_m * p.tiles_n;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  // ---- per-thread staging geometry (rows are fixed over the K loop) ----
  const int cc = tid & 7;    // 16-B chunk inside the 128-B K slab
  const int row0 = tid >> 3; // 0..31; chunk i covers row0 + 32*i
  bool a_ok[C_::A_CH];
  long a_off[C_::A_CH];                                   // plain GEMM: element offset of the row
  int a_b[C_::A_CH], a_y[C_::A_CH], a_x[C_::A_CH];        // conv: output coordinates
#pragma unroll
  for (int i = 0; i < C_::A_CH; ++i) {
    const int mr = m0 + row0 + 32 * i;
    a_ok[i] = mr < p.M;
    const int m = min(mr, p.M - 1);   // rows past M read row M - 1 and are masked to zero
    if (CONV) {
      const int hw = p.Ho * p.Wo;
      const int b = m / hw;
      const int rem = m - b * hw;
      const int oy = rem / p.Wo;
      a_b[i] = b;
      a_y[i] = oy * p.stride - p.pad;
      a_x[i] = (rem - oy * p.Wo) * p.stride - p.pad;
      a_off[i] = 0;
    } else {
      a_off[i] = (long)m * p.lda + cc * 8;
      a_b[i] = a_y[i] = a_x[i] = 0;
    }
  }
  bool b_ok[C_::B_CH];
  long b_off[C_::B_CH];
#pragma unroll
  for (int i = 0; i < C_::B_CH; ++i) {
    const int n = n0 + row0 + 32 * i;
    b_ok[i] = n < p.N;
    b_off[i] = (long)min(n, p.N - 1) * p.ldw + cc * 8;
  }

  uint4 ra[C_::A_CH], rb[C_::B_CH];
  const int Hin = p.ups ? 2 * p.H : p.H;
  const int Win = p.ups ? 2 * p.Wd : p.Wd;

  // Tile loads are UNCONDITIONAL: the address is clamped into the operand and the 16 bytes are AND-ed with an all-ones /
  // all-zero mask.  `ok ? load : 0` is compiled as branch + load + s_waitcnt vmcnt(0) -- one load in flight per thread
  // instead of A_CH + B_CH (tools/isa_audit.py) -- and a select after the load may be folded back into that branch.
  auto masked = [](uint4 v, bool ok) {
    const unsigned m = ok ? 0xFFFFFFFFu : 0u;
    return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
  };
  auto load_tile = [&](int kt) {
    const int k0 = kt * BK;
    if (CONV) {
      const int tap = k0 / p.Cin;
      const int ci0 = k0 - tap * p.Cin;
      const int ky = tap / p.ksize;
      const int kx = tap - ky * p.ksize;
#pragma unroll
      for (int i = 0; i < C_::A_CH; ++i) {
        int iy = a_y[i] + ky, ix = a_x[i] + kx;
        const bool ok = a_ok[i] && iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
        iy = min(max(iy, 0), Hin - 1);
        ix = min(max(ix, 0), Win - 1);
        if (p.ups) {
          iy >>= 1;
          ix >>= 1;
        }
        const long off = (((long)a_b[i] * p.H + iy) * p.Wd + ix) * p.lda + ci0 + cc * 8;
        ra[i] = masked(*reinterpret_cast<const uint4*>(p.A + off), ok);
      }
    } else {
#pragma unroll
      for (int i = 0; i < C_::A_CH; ++i) ra[i] = masked(*reinterpret_cast<const uint4*>(p.A + a_off[i] + k0), a_ok[i]);
    }
#pragma unroll
    for (int i = 0; i < C_::B_CH; ++i) rb[i] = masked(*reinterpret_cast<const uint4*>(p.W + b_off[i] + k0), b_ok[i]);
  };
  auto store_tile = [&](int buf) {
    half_t* As = lds + buf * C_::STAGE_HALFS;
    half_t* Bs = As + BM * LDS_LD;
#pragma unroll
    for (int i = 0; i < C_::A_CH; ++i)
      *reinterpret_cast<uint4*>(As + (row0 + 32 * i) * LDS_LD + cc * 8) = ra[i];
#pragma unroll
    for (int i = 0; i < C_::B_CH; ++i)
      *reinterpret_cast<uint4*>(Bs + (row0 + 32 * i) * LDS_LD + cc * 8) = rb[i];
  };

  float16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const half_t* As = lds + buf * C_::STAGE_HALFS + (wm * 32 * TM + l31) * LDS_LD + hi * 8;
    const half_t* Bs = lds + buf * C_::STAGE_HALFS + BM * LDS_LD + (wn * 32 * TN + l31) * LDS_LD + hi * 8;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      half8_t af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *reinterpret_cast<const half8_t*>(As + i * 32 * LDS_LD + s * 16);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bf[j] = *reinterpret_cast<const half8_t*>(Bs + j * 32 * LDS_LD + s * 16);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue phase 1: registers -> fp32 LDS image (bias, row vector, activation) ----
  float* Cs = reinterpret_cast<float*>(smem);
  const bool geglu = p.act == PFD_ACT_GEGLU;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nl = wn * 32 * TN + j * 32 + l31;
      const int n = n0 + nl;
      // all 1 + 16 + 16 scalar operand loads of this accumulator tile first, unconditional (clamped index; the zero page
      // with stride 0 for an absent operand): written as `if (p.rowvec) v += rowvec[...]` per element they were 32
      // serial L2 round trips per tile (tools/isa_audit.py)
      const int nc = min(n, p.N - 1);
      const half_t* zp = pfd_zero_halves();
      const half_t* bcol_p = (p.bias && !p.bias_per_row) ? p.bias + nc : zp;
      const half_t* brow_p = (p.bias && p.bias_per_row) ? p.bias : zp;
      const int brow_s = (p.bias && p.bias_per_row) ? 1 : 0;
      const half_t* rv_p = p.rowvec ? p.rowvec + nc : zp;
      const long rv_s = p.rowvec ? p.ldrv : 0;
      const half_t bcol_h = *bcol_p;
      half_t brow_h[16], rv_h[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = min(m0 + wm * 32 * TM + i * 32 + mfma32_row(r, hi), p.M - 1);
        brow_h[r] = brow_p[(long)m * brow_s];
        rv_h[r] = rv_p[(long)(m / p.rows_per_rv) * rv_s];
      }
      const float bcol = (float)bcol_h;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[i][j][r] + bcol + (float)brow_h[r] + (float)rv_h[r];   // rows / columns past M, N are never stored
        if (p.act == PFD_ACT_GELU) v = pfd_gelu(v);
        else if (p.act == PFD_ACT_RELU) v = fmaxf(v, 0.f);
        else if (p.act == PFD_ACT_SILU) v = pfd_silu(v);
        acc[i][j][r] = v;
      }
    }
  }
  if (geglu) {
    if constexpr (TN == 2) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ml = wm * 32 * TM + i * 32 + mfma32_row(r, hi);
          Cs[ml * C_::EPI_LD + wn * 32 + l31] = acc[i][0][r] * pfd_gelu(acc[i][1][r]);
        }
    }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ml = wm * 32 * TM + i * 32 + mfma32_row(r, hi);
          Cs[ml * C_::EPI_LD + wn * 32 * TN + j * 32 + l31] = acc[i][j][r];
        }
  }
  __syncthreads();

  // ---- epilogue phase 2: + residual, 16-B row segments to HBM ----
  const int bn_out = geglu ? BN / 2 : BN;
  const int n_out0 = geglu ? n0 / 2 : n0;
  const int N_out = geglu ? p.N / 2 : p.N;
  const int cpr = bn_out / 8;  // chunks per row
  const bool vec_ok = ((p.ldc & 7) == 0) && (p.R == nullptr || (p.ldr & 7) == 0) &&
                      ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) &&
                      (p.R == nullptr || (reinterpret_cast<uintptr_t>(p.R) & 15) == 0);
  for (int c = tid; c < BM * cpr; c += 256) {
    const int ml = c / cpr;
    const int nc = (c - ml * cpr) * 8;
    const int m = m0 + ml;
    const int n = n_out0 + nc;
    if (m >= p.M || n >= N_out) continue;
    const float4_t v0 = *reinterpret_cast<const float4_t*>(Cs + ml * C_::EPI_LD + nc);
    const float4_t v1 = *reinterpret_cast<const float4_t*>(Cs + ml * C_::EPI_LD + nc + 4);
    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    if (vec_ok && n + 8 <= N_out) {
      if (p.R) {
        Pack16 r;
        r.u = *reinterpret_cast<const uint4*>(p.R + (long)m * p.ldr + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)r.e[e];
      }
      Pack16 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o.e[e] = (half_t)v[e];
      *reinterpret_cast<uint4*>(p.C + (long)m * p.ldc + n) = o.u;
    } else {
      for (int e = 0; e < 8 && n + e < N_out; ++e) {
        float x = v[e];
        if (p.R) x += (float)p.R[(long)m * p.ldr + n + e];
        p.C[(long)m * p.ldc + n + e] = (half_t)x;
      }
    }
  }
}

template <int TM, int TN>
int launch(const GemmParams& p0, hipStream_t s) {
  GemmParams p = p0;
  using C_ = Cfg<TM, TN>;
  p.tiles_m = (p.M + C_::BM - 1) / C_::BM;
  p.tiles_n = (p.N + C_::BN - 1) / C_::BN;
  const int grid = p.tiles_m * p.tiles_n;
  const bool prof = pfd_prof_on();
  if (prof) {
    // algorithmic work of this launch: 2MNK flops; bytes = A read once + W read once + C written
    // (+ residual read); for the implicit conv, A is the input image read once (not 9x)
    const double a_bytes = p.ksize > 0 ? 2.0 * p.B * p.H * p.Wd * p.Cin : 2.0 * p.M * p.K;
    const double bytes = a_bytes + 2.0 * p.N * p.K + 2.0 * p.M * p.N * (p.R ? 2 : 1);
    pfd_prof_begin((TM - 1) * 2 + (TN - 1) + (p.ksize > 0 ? 4 : 0), 2.0 * p.M * p.N * p.K, bytes, s);
  }
  if (p.ksize > 0)
    hipLaunchKernelGGL((gemm_conv_kernel<TM, TN, true>), dim3(grid), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((gemm_conv_kernel<TM, TN, false>), dim3(grid), dim3(256), 0, s, p);
  if (prof) pfd_prof_end(s);
  return pfd_check_launch("pfd_gemm_f16");
}

// ------------------------------------------------------------------------------------------------
// conv3x3_narrow_kernel (round 6): 3x3 / stride 1 / pad 1 convolution with N <= 16 output channels -- the UNet head
// (GroupNorm -> SiLU -> conv 320 -> 4, openaimodel.py:2732-2737 of the reference), the VAE's conv_out (128 -> 3).  The 64 x 64
// tile of gemm_conv_kernel spent 60 of its 64 columns on padding (32768 x 4 x 2880: 52 us, 14 TFLOP/s).  Here:
//   * a block = 64 consecutive output pixels of one image row (16 per wave); per 64-channel block of the input it stages the
//     3 x 66-pixel patch once (global -> registers -> LDS, next block's loads in flight under the MFMAs, one barrier per
//     channel block; pixels outside the image are zeros) and all nine taps read shifted pixels of it;
//   * v_mfma_f32_16x16x32_f16 with the WEIGHTS as the A operand (row n = lane & 15, rows >= N alias row 0 and are never
//     stored) and the activations as B (column = pixel): the accumulator holds, per lane, output channels 4 (lane >> 4) + r
//     of ONE pixel -- the lanes of row group 0 store 8 contiguous bytes per pixel for N = 4;
//   * weights come straight from global memory (N x K x 2 bytes <= 92 KB for the whole launch: L1 / L2 resident), the 18
//     fragments of a channel block requested before its barrier;
//   * 144-byte pixel pitch in LDS (9 x 16 bytes: every 16-lane group of a ds_read_b128 hits 16 different bank quads).
// Algorithmic work: 2 M N K flops; bytes = input image once + weights + output.
constexpr int NRW_PX = 64, NRW_PW = NRW_PX + 2, NRW_PATCH = 3 * NRW_PW, NRW_PITCH = BK + 8;
constexpr int NRW_CHUNKS = NRW_PATCH * (BK / 8), NRW_PT = (NRW_CHUNKS + 255) / 256;

__global__ __launch_bounds__(256) void conv3x3_narrow_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) half_t lds[2][NRW_PATCH * NRW_PITCH];
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kg = lane >> 4;
  const int segs = (p.Wd + NRW_PX - 1) / NRW_PX;
  const int seg = blockIdx.x % segs, y = (blockIdx.x / segs) % p.H, b = blockIdx.x / (segs * p.H);
  const int x0 = seg * NRW_PX;
  const int ncb = p.Cin / BK;

  // staging: chunk c = tid + 256 j: patch pixel c >> 3 (row (c >> 3) / 66, column (c >> 3) % 66), 16-byte channel chunk c & 7
  const half_t* src[NRW_PT];
  int dst[NRW_PT];
  bool live[NRW_PT], inside[NRW_PT];
#pragma unroll
  for (int j = 0; j < NRW_PT; ++j) {
    const int c = tid + 256 * j;
    live[j] = c < NRW_CHUNKS;
    const int pp = min(c, NRW_CHUNKS - 1) >> 3, cc = c & 7;
    const int pr = pp / NRW_PW, px = pp - pr * NRW_PW;
    const int yy = y + pr - 1, xx = x0 + px - 1;
    inside[j] = yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd;
    const int yc = min(max(yy, 0), p.H - 1), xc = min(max(xx, 0), p.Wd - 1);   // unconditional loads from a valid pixel, masked below
    src[j] = p.A + (((long)b * p.H + yc) * p.Wd + xc) * p.lda + cc * 8;
    dst[j] = pp * NRW_PITCH + cc * 8;
  }
  u32x4 reg[NRW_PT];
  auto load_patch = [&](int cb) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NRW_PT; ++j) reg[j] = *reinterpret_cast<const u32x4*>(src[j] + cb * BK);
  };
  auto store_patch = [&](int stage) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NRW_PT; ++j) {
      const u32x4 v = inside[j] ? reg[j] : (u32x4){0u, 0u, 0u, 0u};
      if (live[j]) *reinterpret_cast<u32x4*>(&lds[stage][dst[j]]) = v;
    }
  };
  // weight fragment of (tap, k half): row (l15 < N ? l15 : 0), 8 halfs at tap * Cin + cb * 64 + ks * 32 + kg * 8
  const half_t* wrow = p.W + (long)(l15 < p.N ? l15 : 0) * p.ldw + kg * 8;
  const int xfrag = (wave * 16 + l15) * NRW_PITCH + kg * 8;   // + (ky * 66 + kx) * PITCH + ks * 32

  float4_t acc = {0.f, 0.f, 0.f, 0.f};
  load_patch(0);
  store_patch(0);
  for (int cb = 0; cb < ncb; ++cb) {
    if (cb + 1 < ncb) load_patch(cb + 1);
    half8_t wf[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) wf[t][ks] = *reinterpret_cast<const half8_t*>(wrow + (long)t * p.Cin + cb * BK + ks * 32);
    __syncthreads();
    const half_t* L = lds[cb & 1];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ky = t / 3, kx = t - ky * 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const half8_t xf = *reinterpret_cast<const half8_t*>(L + xfrag + (ky * NRW_PW + kx) * NRW_PITCH + ks * 32);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t][ks], xf, acc, 0, 0, 0);
      }
    }
    if (cb + 1 < ncb) store_patch((cb + 1) & 1);
  }
  // epilogue: lane (pixel l15, row group kg) holds channels 4 kg + r
  const int x = x0 + wave * 16 + l15, n0 = 4 * kg;
  if (x < p.Wd && n0 < p.N) {
    const long m = ((long)b * p.H + y) * p.Wd + x;
    // bias: four unconditional loads issued together (an absent bias reads the zero page; `p.bias ? load : 0` is a branch +
    // load + vmcnt(0) per element -- tools/isa_audit.py)
    const half_t* bsrc = p.bias ? p.bias : pfd_zero_halves();
    half_t bv[4], o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bsrc[p.bias ? min(n0 + r, p.N - 1) : r];
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (half_t)(acc[r] + (float)bv[r]);
    half_t* cp = p.C + m * p.ldc + n0;
    if (n0 + 4 <= p.N && (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 7) == 0) {
      Pack8 q;
#pragma unroll
      for (int r = 0; r < 4; ++r) q.e[r] = o[r];
      *reinterpret_cast<uint2*>(cp) = q.u;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n0 + r < p.N) cp[r] = o[r];
    }
  }
}

// the narrow-output 3x3 convolution: shapes conv3x3_narrow_kernel serves
static bool narrow_conv_takes(const GemmParams& p) {
  return p.ksize == 3 && p.stride == 1 && p.pad == 1 && !p.ups && p.Ho == p.H && p.Wo == p.Wd && p.N <= 16 && p.Cin % BK == 0 &&
         !p.rowvec && !p.R && p.act == PFD_ACT_NONE && !p.bias_per_row;
}

static int launch_narrow_conv(const GemmParams& p, hipStream_t s) {
  const bool prof = pfd_prof_on();
  if (prof)
    pfd_prof_begin(20, 2.0 * p.M * p.N * p.K, 2.0 * p.B * p.H * p.Wd * p.Cin + 2.0 * p.N * p.K + 2.0 * p.M * p.N, s);
  const int segs = (p.Wd + NRW_PX - 1) / NRW_PX;
  hipLaunchKernelGGL(conv3x3_narrow_kernel, dim3(segs * p.H * p.B), dim3(256), 0, s, p);
  if (prof) pfd_prof_end(s);
  return pfd_check_launch("pfd_gemm_f16(conv3x3 narrow)");
}

}  // namespace

int pfd_gemm160_try(const PfdGemmDesc* d, int variant, int splits, hipStream_t s);  // gemm_glds.hip

extern "C" int32_t pfd_gemm_geglu_group(int32_t N) { return N % 160 == 0 ? 2 : 32; }

extern "C" int pfd_gemm_f16_ex(const PfdGemmDesc* d, int32_t tile, pfd_stream_t stream) {
  if (!d || !d->A || !d->W || !d->C) return PFD_EINVAL;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return PFD_EINVAL;
  if ((d->lda & 7) || (d->ldw & 7)) return PFD_EINVAL;
  if ((reinterpret_cast<uintptr_t>(d->A) & 15) || (reinterpret_cast<uintptr_t>(d->W) & 15)) return PFD_EINVAL;
  if (d->K % BK) return PFD_ESHAPE;
  if (d->rowvec && d->rows_per_rv < 1) return PFD_EINVAL;
  if (d->act < PFD_ACT_NONE || d->act > PFD_ACT_GEGLU) return PFD_EINVAL;
  if (d->ksize > 0) {
    if (d->Cin <= 0 || d->K != d->ksize * d->ksize * d->Cin) return PFD_EINVAL;
    if ((long)d->B * d->Ho * d->Wo != d->M) return PFD_EINVAL;
    if (d->stride < 1) return PFD_EINVAL;
  }
  if (d->Ct && (d->act != PFD_ACT_NONE || d->rowvec || d->R || d->bias_per_row || (tile != 0 && tile < 1000)))
    return PFD_EINVAL;
  if (tile == 0 || tile >= 1000) {  // wide-tile LDS-DMA path (N % 160 == 0)
    const int enc = tile >= 1000 ? tile - 1000 : 0;
    const int rc = pfd_gemm160_try(d, enc / 100, enc % 100, (hipStream_t)stream);
    if (rc <= 0) return rc;
    if (tile >= 1000 || d->Ct) return PFD_ESHAPE;  // forced (or transposed tail) but not applicable
  }
  if (d->gn_table) {
    pfd_set_error("pfd_gemm_f16: the GroupNorm prologue is served by the 3x3 patch kernel only (see PfdGemmDesc.gn_table)");
    return PFD_ESHAPE;
  }
  if (d->gn_out) {
    pfd_set_error("pfd_gemm_f16: GroupNorm statistics of the output are emitted by the wide-tile kernels only (see PfdGemmDesc.gn_out)");
    return PFD_ESHAPE;
  }
  if (d->res_rows > 0 && d->res_rows != d->M) {
    pfd_set_error("pfd_gemm_f16: a residual stored once for a doubled batch is read by the wide-tile kernels only (see PfdGemmDesc.res_rows)");
    return PFD_ESHAPE;
  }
  if (d->gnf_y) {
    pfd_set_error("pfd_gemm_f16: the fused GroupNorm lives in the split-K reduction of the wide-tile kernels only (see PfdGemmDesc.gnf_y)");
    return PFD_ESHAPE;
  }
  if (d->k_split > 0 || d->zero_rows > 0) {
    pfd_set_error("pfd_gemm_f16: k_split / zero_rows are served by the wide-tile linear kernels only (see PfdGemmDesc.k_split)");
    return PFD_ESHAPE;
  }
  if (d->w_tiled) {
    pfd_set_error("pfd_gemm_f16: K-tile-contiguous weights are read by the wide-tile kernels only (see PfdGemmDesc.w_tiled)");
    return PFD_ESHAPE;
  }
  if (d->ln_stats || d->ln_out) {
    pfd_set_error("pfd_gemm_f16: the LayerNorm fold is served by the wide-tile linear kernels only (see PfdGemmDesc.ln_stats)");
    return PFD_ESHAPE;
  }
  GemmParams p;
  p.A = (const half_t*)d->A;
  p.W = (const half_t*)d->W;
  p.bias = (const half_t*)d->bias;
  p.rowvec = (const half_t*)d->rowvec;
  p.R = (const half_t*)d->R;
  p.C = (half_t*)d->C;
  p.lda = d->lda; p.ldw = d->ldw; p.ldr = d->ldr; p.ldc = d->ldc; p.ldrv = d->ldrv;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.rows_per_rv = d->rows_per_rv > 0 ? d->rows_per_rv : 1;
  p.act = d->act; p.bias_per_row = d->bias_per_row;
  p.ksize = d->ksize; p.stride = d->stride; p.pad = d->pad; p.ups = d->ups;
  p.B = d->B; p.H = d->H; p.Wd = d->Wd; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo;
  p.tiles_m = p.tiles_n = 0;
  if (p.ksize > 0) {
    if (p.Cin % BK) return PFD_ESHAPE;
    if (p.K != p.ksize * p.ksize * p.Cin) return PFD_EINVAL;
    if ((long)p.B * p.Ho * p.Wo != p.M) return PFD_EINVAL;
    if (p.stride < 1) return PFD_EINVAL;
  }
  if (p.act < PFD_ACT_NONE || p.act > PFD_ACT_GEGLU) return PFD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (tile == 0 && narrow_conv_takes(p)) return launch_narrow_conv(p, s);
  if (p.act == PFD_ACT_GEGLU) {
    // N % 160 == 0 is packed in pairs for the wide-tile kernel (pfd_gemm_geglu_group): if that kernel declined
    // the call (unaligned C / ldc, forced tile), running the 32-block-interleave kernel here would pair the
    // wrong x / gate rows silently
    if (p.N % 128 || pfd_gemm_geglu_group(p.N) != 32) return PFD_ESHAPE;
    return launch<2, 2>(p, s);
  }
  switch (tile) {  // explicit tile (tests / tuning); 0 = heuristic below
    case 0: break;
    case 22: return launch<2, 2>(p, s);
    case 21: return launch<2, 1>(p, s);
    case 12: return launch<1, 2>(p, s);
    case 11: return launch<1, 1>(p, s);
    default: return PFD_EINVAL;
  }
  // Tile choice: fill the 256 CUs first, then prefer the larger (higher-intensity) tile.
  auto blocks = [&](int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
  auto waste = [&](int bn) { return (double)(((p.N + bn - 1) / bn) * bn) / p.N; };
  if (blocks(128, 128) >= 512 && waste(128) <= 1.13) return launch<2, 2>(p, s);
  if (blocks(128, 64) >= 384 && p.M >= 128) return launch<2, 1>(p, s);
  if (p.N > 64 && blocks(64, 128) >= 384 && waste(128) <= 1.13) return launch<1, 2>(p, s);
  return launch<1, 1>(p, s);
}

extern "C" int pfd_gemm_f16(const PfdGemmDesc* d, pfd_stream_t stream) { return pfd_gemm_f16_ex(d, 0, stream); }
