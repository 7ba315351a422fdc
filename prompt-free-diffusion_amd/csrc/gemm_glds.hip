// pfd_gemm_f16, wide-tile path: linear / 1x1 / 3x3 implicit-GEMM convolution on v_mfma_f32_16x16x32_f16
// for N % 160 == 0 (every SD-v1.5 UNet / ControlNet width is a multiple of 320; 160-wide tiles, NT = 5) or
// N % 128 == 0 (VAE / Swin / SeeCoder widths; 128-wide tiles, NT = 4).
//
// Block tile BM x BN x 64 with BM = WAVES_M * WMB * 16, BN = 32 * NT; waves laid out WAVES_M (M) x 2 (N);
// each wave owns a (WMB*16) x (16*NT) sub-tile = WMB x NT MFMA tiles of 16x16 (fp32 accumulators:
// WMB*NT*4 registers).  Variants: <4,4> 256xBN (8 waves), <2,4> 128xBN, <2,2> 64xBN (4 waves);
// small-MN / huge-K problems add split-K over gridDim.z with fp32 slabs and a reduce+epilogue kernel.
//
// Staging is LDS-DMA (global_load_lds_dwordx4): one wave instruction moves 8 rows x 128 B straight
// from global memory into LDS, two LDS stages, the next K tile in flight during the MFMAs of the
// current one, one vmcnt(0)+barrier per K step.  The LDS image is lane-linear, so the bank-
// conflict swizzle is applied to the per-lane SOURCE address: 16-byte chunk c of row r is stored
// at chunk position c ^ ((r >> 1) & 7); the fragment ds_read_b128 applies the same XOR.  With
// 128-byte rows two rows share one 256-byte bank row, so (row parity, chunk ^ (row>>1)) gives every
// lane group of a ds_read_b128 sixteen distinct 16-byte slots (conflict free).
//
// Convolution: the A operand is a per-lane gather -- the source address of lane (row, chunk) is
// the input pixel under the current filter tap; taps outside the image (and rows past M) read a
// 256-byte zero page instead, so zero padding, stride 2 and the fused nearest-2x upsample
// (openaimodel.py:114) are pure address arithmetic.  K is walked tap-major / channel-minor with
// counters (no integer division in the loop).
//
// Epilogue: straight from the accumulators (operands are swapped in the MFMA so a lane owns 4 consecutive
// output columns of one row): + bias + per-sample row vector (time embedding) -> activation (or GEGLU:
// the packed weight stores x0 x1 g0 g1 | x2 x3 g2 g3 ..., a lane's four columns) -> + residual -> fp16.
//
// Algorithmic bytes / flops per launch: see pfd_prof_begin below (operands + result once; 2MNK).
#include <type_traits>

#include <stdlib.h>

#include "pfd_common.h"

int pfd_ln_rowstats_launch(const half_t* x, long ldx, int M, int C, float* out, hipStream_t s, bool prof);   // norm.hip

namespace {

constexpr int BK = 64;
constexpr int BN = 160;             // patch kernel / default tile width (gemm160_kernel derives its own from NT)
constexpr int ROWB = BK * 2;         // bytes per LDS row (128)

__device__ __attribute__((aligned(256))) half_t g_zero_page[128];  // zero-initialised: OOB source of the gather

struct G160Params {
  const half_t* A;
  const half_t* W;
  const half_t* bias;
  const half_t* rowvec;
  const half_t* R;
  half_t* C;
  float* ws;  // split-K slabs [splits][M][N]: f16 partial sums (round 6; fp32 through round 5), summed in fp32 in slab order by the reduction kernels.
              // The host still sizes / requires 4 bytes per element (PfdGemmDesc.ws_bytes: the contract of ABI 9 is unchanged)
  half_t* Ct;  // transposed tail: columns >= n_split -> Ct[(n - n_split) * ldct + m]
  long ldct;
  int n_split;
  long lda, ldw, ldr, ldc, ldrv;
  int M, N, K;
  int rows_per_rv, act;
  int ksize, stride, pad, ups;
  int B, H, Wd, Cin, Ho, Wo;
  int tiles_m, tiles_n, splits, kt_per_split;
  int nmajor;  // XCD-contiguous tile order: 0 = all N tiles of an M tile together, 1 = all M tiles of an N tile
  int krot;    // 1 = every M tile starts its K loop at a different K tile (see k_rotation below)
  // 3x3 patch kernels on images wider than a tile row (round 3: 48- and 96-wide latents): an output tile is TH x pt_w
  // pixels (pt_w = 16 | 32, TH = 256 / pt_w) instead of whole image rows, so the 256 rows of a tile are not consecutive
  // in M: row r of the tile is output row m0 + (r >> pt_sh) * Wd + (r & (pt_w - 1)), m0 = the tile's first pixel.
  // pt_w == 0: rows are consecutive (every other kernel, and patch tiles of whole image rows).
  int pt_w, pt_sh;
  // weight layout: row-major [N][ldw] (w_tu == 0) or K-tile-contiguous [N / w_tu][K / 64][w_tu][64] (w_tu = 160 | 128):
  // a block's weight tile of one K step is then ONE contiguous (w_tu x 128)-byte run of HBM instead of w_tu separate
  // 128-byte pieces a whole weight row (2 K bytes) apart -- the cold weight streams of the 8^2 / 16^2 levels ran at
  // ~1 TB/s on the row-major layout whatever the tile shape (profiles/r03_tile_variants_replay.log)
  int w_tu;
  long w_kstep;   // halfs from one K tile to the next: 64 (row-major) or w_tu * 64
  // GroupNorm(+SiLU) folded into the patch convolution's input staging (PfdGemmDesc.gn_table)
  // LayerNorm folded into the contraction (PfdGemmDesc.ln_stats / ln_out)
  const float2* ln_in;     // [M][ln_P]: partial (sum x, sum x^2) of the rows of A; nullptr = no fold
  const float* ln_cs;      // [N]: column sums of the gamma-scaled weight
  float2* ln_out;          // [M][tiles_n]: partial row sums of the OUTPUT of this launch; nullptr = none
  int ln_P;
  float ln_eps;
  const float* gn_table;   // [B][2][Cin]: scale plane, shift plane
  const half_t* A2;        // channels >= gn_c1 of the virtual concat (GroupNorm prologue) / columns >= k_split (linear)
  long lda2;
  int gn_c1, gn_act;
  float2* gn_out;          // GroupNorm statistics of the OUTPUT, per 64-row slab and group (PfdGemmDesc.gn_out); nullptr = none
  int k_split;             // linear kernels: K tiles at k >= k_split come from A2 (== K when there is no second source)
  int zero_rows;           // linear kernels: operand rows below this are all zero and are never read (PfdGemmDesc.zero_rows)
  int r_wrap;              // residual rows (PfdGemmDesc.res_rows): output row m adds R[m >= r_wrap ? m - r_wrap : m]; INT_MAX = no wrap
};

// residual row of output row m (clamped to the problem): the residual of a CFG pair [x | x] is stored once (res_rows)
__device__ __forceinline__ long res_row(const G160Params& p, int m) {
  const int mc = min(m, p.M - 1);
  return mc >= p.r_wrap ? mc - p.r_wrap : mc;
}

__device__ __forceinline__ void glds16(const void* src, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// address of 16-byte chunk c8 (in halfs) of weight row n at K tile 0, in either layout
__device__ __forceinline__ const half_t* w_row_ptr(const G160Params& p, int n, int c8) {
  if (p.w_tu == 0) return p.W + (long)n * p.ldw + c8;
  const int tn = n / p.w_tu;
  return p.W + ((long)tn * (p.K / BK) * p.w_tu + (n - tn * p.w_tu)) * BK + c8;
}

// Kernel prologues (round 5).  The ISA of these kernels ran 460-980 instructions before the FIRST operand load was issued;
// two mechanical causes are removed (linears: 462-639 -> 361-515 instructions, 3-5 -> 2-3 argument waits; -0.5 % per batch on
// MI355X, profiles/r05_e2e_ab_candidates.log, same results bit for bit):
//   * the kernel arguments arrived in 3-5 dependent s_load -> s_waitcnt batches because hipcc sinks each argument load into the
//     branch that first uses it: PFD_ARG_BATCH names the scalars of the setup in one statement at entry (one batch);
//   * w_row_ptr divided by the weight layout's tile width once per weight piece, although that width is the kernel's own tile
//     width (or half of it, for the 320-wide GEGLU tile): w_row_ptr_tile forms the same address from the tile index.
#if !defined(PFD_CPU_EMU)
#define PFD_ARG_BATCH_LIN(p)                                                                                              \
  asm volatile("" ::"s"((p).tiles_m), "s"((p).tiles_n), "s"((p).nmajor), "s"((p).kt_per_split), "s"((p).K), "s"((p).M),      \
               "s"((p).zero_rows), "s"((p).k_split), "s"((p).krot), "s"((p).w_tu), "s"((p).A), "s"((p).A2), "s"((p).W),      \
               "s"((p).lda), "s"((p).lda2), "s"((p).ldw), "s"((p).w_kstep))
#define PFD_ARG_BATCH_CONV(p)                                                                                             \
  asm volatile("" ::"s"((p).tiles_m), "s"((p).tiles_n), "s"((p).nmajor), "s"((p).kt_per_split), "s"((p).K), "s"((p).M),      \
               "s"((p).krot), "s"((p).w_tu), "s"((p).A), "s"((p).W), "s"((p).lda), "s"((p).ldw), "s"((p).w_kstep),           \
               "s"((p).Cin), "s"((p).H), "s"((p).Wd), "s"((p).Ho), "s"((p).Wo), "s"((p).stride), "s"((p).pad), "s"((p).ups), \
               "s"((p).ksize))
#else
#define PFD_ARG_BATCH_LIN(p) ((void)0)
#define PFD_ARG_BATCH_CONV(p) ((void)0)
#endif
// address of 16-byte chunk c8 (in halfs) of row r of weight tile tile_n (BN rows) at K tile 0, in either layout, without a division
template <int BN_>
__device__ __forceinline__ const half_t* w_row_ptr_tile(const G160Params& p, int tile_n, int r, int c8) {
  if (p.w_tu == 0) return p.W + (long)(tile_n * BN_ + r) * p.ldw + c8;
  const int sub = (BN_ > 160 && r >= p.w_tu) ? 1 : 0;               // host: w_tu == BN_, or BN_ == 2 w_tu (the 320-wide tile)
  const int tn = tile_n * (BN_ > 160 ? 2 : 1) + sub;
  return p.W + ((long)tn * (p.K / BK) * p.w_tu + (r - sub * p.w_tu)) * BK + c8;
}
#define PFD_W_ROW_PTR(BN_, p, tile_n, n0, r, c8) w_row_ptr_tile<BN_>(p, tile_n, r, c8)

// Epilogue operands read LATE.  hipcc loads every kernel-argument field a kernel uses with one batch of s_load at the entry
// and keeps it in SGPRs until its last use: the ~25 scalars only the epilogue needs (bias / row-vector / residual / output
// pointers and strides, statistics pointers, M, N ...) then sit on top of the main loop's own scalars for the whole launch,
// and the 12-wave patch kernel -- whose loader setup is at the 102-SGPR limit -- spills.  The epilogue of that kernel
// reads its copy of the argument block through an opaque pointer instead: the loads are issued where the copy is made.
__device__ __forceinline__ G160Params reload_params() {
  typedef __attribute__((address_space(4))) const unsigned* KArgPtr;
  KArgPtr k = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();   // the G160Params block is the kernel's only argument (offset 0)
  asm volatile("" : "+s"(k)::"memory");                          // loads below cannot be hoisted above this point
  static_assert(sizeof(G160Params) % 4 == 0, "argument block read as dwords");
  unsigned w[sizeof(G160Params) / 4];
#pragma unroll
  for (unsigned i = 0; i < sizeof(G160Params) / 4; ++i) w[i] = k[i];
  G160Params q;
  __builtin_memcpy(&q, w, sizeof(q));
  return q;
}

// Global-memory accesses of the epilogues, with the address space spelled out at the access: a pointer that came out of
// reload_params() is generic to the compiler (flat_load / flat_store, which also tick lgkmcnt and order against the LDS reads of
// the store pass); through these they are global_load / global_store whatever the pointer's provenance.
template <int BYTES> struct GRaw;
template <> struct GRaw<2>  { typedef unsigned short type; };
template <> struct GRaw<8>  { typedef unsigned int type __attribute__((ext_vector_type(2))); };
template <> struct GRaw<16> { typedef unsigned int type __attribute__((ext_vector_type(4))); };
template <class T>
__device__ __forceinline__ T gld(const void* ptr) {
  typedef typename GRaw<sizeof(T)>::type V;
  const V v = *reinterpret_cast<const __attribute__((address_space(1))) V*>((unsigned long)ptr);
  T out;
  __builtin_memcpy(&out, &v, sizeof(T));
  return out;
}
template <class T>
__device__ __forceinline__ void gst(void* ptr, T val) {
  typedef typename GRaw<sizeof(T)>::type V;
  V v;
  __builtin_memcpy(&v, &val, sizeof(T));
  *reinterpret_cast<__attribute__((address_space(1))) V*>((unsigned long)ptr) = v;
}

// output row (index into M) of row `row` of the tile that starts at m0 (see G160Params.pt_w)
// PT = false (every kernel but the 3x3 patch kernels): rows are consecutive, the mapping folds away
template <bool PT>
__device__ __forceinline__ int tile_row_m(const G160Params& p, int m0, int row) {
  if constexpr (PT) return p.pt_w ? m0 + (row >> p.pt_sh) * p.Wd + (row & (p.pt_w - 1)) : m0 + row;
  else return m0 + row;
}

// K rotation.  All blocks of a launch start together and take the same time per K step, so the tiles_m blocks that
// share one weight tile (same N tile, same split) ask for the SAME K tile of W at the same moment: one of them misses
// to HBM, the others wait on that miss, and at any time only (ring depth) x (N tiles) x (splits) distinct weight
// tiles are in flight chip-wide -- 1-4 MB against the >= 12 MB that 6 TB/s x ~2 us of loaded HBM latency need.  The
// layers' weights are cold every time (1.7 GB of other layers pass through the 256 MB MALL between two uses), so the
// 8^2 / 16^2 levels streamed W at ~1 TB/s (512 x 1280 x 11520: 29.5 MB in 34 us).  Starting M tile m at K tile
// (m * stride) mod n makes the sharers lead on DIFFERENT parts of W (each part is fetched from HBM once, by its
// leader, and hit in L2 by the others later): the number of distinct tiles in flight grows by min(tiles_m, n).
// The fp32 summation order of a row then depends on its M tile (deterministic; PFD_KROT=0 restores the plain order).
__device__ __forceinline__ int k_rotation(int krot, int tile_m, int tiles_m, int nsteps) {
  if (!krot || nsteps < 2) return 0;
  const int stride = max(1, nsteps / max(1, tiles_m));
  return (tile_m * stride) % nsteps;
}

// 16-byte LDS store the compiler cannot see: next to pending LDS-DMA pieces hipcc orders every ds_write it knows
// about behind `s_waitcnt vmcnt(0)` (the DMA is a pending LDS write on the VM counter), which would drain the
// loader's whole prefetch queue once per store.  The caller waits lgkmcnt(0) before the barrier that publishes it.
__device__ __forceinline__ void lds_store16_opaque(void* lds_dst, uint4 v) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_dst;
  const u32x4 d = {v.x, v.y, v.z, v.w};
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(d) : "memory");
}

// Epilogue shared by the wide-tile kernels.  The MFMAs are issued with the operands SWAPPED (W fragment as
// the A operand, activation fragment as B), so a 16x16 accumulator tile holds, per lane, FOUR CONSECUTIVE
// OUTPUT COLUMNS of ONE output row: row m = tile row (lane & 15), columns n = 4*(lane >> 4) + r.  Bias / row
// vector / residual are 8-byte loads, the result is an 8-byte store, and nothing goes through LDS (the
// row-major staging the natural layout needed cost 2 x 164 KB of LDS traffic per 256x160 tile -- as much as
// two K steps, on problems that have 5 to 20 of them).  GEGLU packs its weight rows so that a lane's four
// columns are two (x, gate) pairs.
// (Split-K keeps a separate reduce launch.  An in-kernel "last block to arrive reduces the slabs" variant
//  was built and measured: with agent-scope fences the L2 write-back/invalidate per block took the UNet
//  loop from 753 to 1046 ms; with sc1 / sc0+sc1 coherent slab accesses instead of fences, 685 -> 868 ms --
//  the uncached slab round trip sits on the critical path of every tile's last block, while the reduce
//  kernel streams the same data with 2048 blocks in ~11 us.)
// Row stride (bytes) of the fp16 staging image of an output tile of `cols` columns: + 16 bytes makes the
// 8-byte (4-byte for GEGLU) accumulator writes of 16 consecutive rows hit 16 distinct bank groups and keeps
// every row 16-byte aligned for the ds_read_b128 of the store pass (cols = 160: 336 B = 84 dwords, row r
// starts at bank 20 r mod 64; 128: 272 B -> 4 r; 80: 176 B -> 44 r mod 64; 64: 144 B -> 36 r mod 64).
constexpr int stage_row_bytes(int cols) { return cols * 2 + 16; }

// pass 1 (the waves that hold accumulators): returns true when the tile was staged in LDS and needs pass 2
// lnstat: per-row {rstd, -rstd * mean} of this block's rows in LDS (LayerNorm folded into the GEMM), or nullptr
template <int WMB, int NT, bool PT = false>
__device__ __forceinline__ bool epilogue_stage(float4_t (&acc)[WMB][NT], const G160Params& p, int lane, int m0,
                                               int n0, int wm, int wn, int split, char* smem,
                                               const float2* lnstat = nullptr) {
  constexpr int BN = 32 * NT;
  const int l15 = lane & 15, g = lane >> 4;
  const int lr0 = wm * WMB * 16 + l15;       // + i*16: this lane's row inside the tile, row-tile i
  auto mrow = [&](int lr) __attribute__((always_inline)) { return tile_row_m<PT>(p, m0, lr); };   // -> output row (index into M)
  const int nw = n0 + wn * (16 * NT) + 4 * g;       // + j*16: first of this lane's 4 columns in column-tile j
  constexpr bool GEGLU_ONLY = NT == 10;      // the 320-wide tile is dispatched for GEGLU projections only
  if (lnstat && p.splits == 1) {
    // LN(x) W^T = rstd_m (x (W o gamma)^T)[m, n] - rstd_m mean_m s_n (+ b'_n, the `bias` of this launch): the affine map
    // goes onto the accumulators, everything behind it (bias, activation / GEGLU, transposed tail) is unchanged
    // The row statistics first, with their own wait, far in front of their first use.  (Round 4: with the LDS reads issued
    // right in front of the packed-f32 multiplies that consume them -- `ds_read2_b64 ...; s_waitcnt vmcnt(9) lgkmcnt(0);
    // v_pk_mul_f32` -- lanes 48-63 of the FIRST product after the wait occasionally saw the previous contents of the
    // destination registers: one accumulator register of one 16-row tile per ~10^7 elements differed between two launches
    // of the same problem.  Which launches did so changed with the placement of the kernel in the code object, i.e. with
    // unrelated edits; tools/determinism_gemm.py repeats single launches and compares bits.  Read early + explicit wait:
    // 40 / 40 identical in every configuration, profiles/r04_ln_fold_determinism.log.)
    float2 abv[WMB];
#pragma unroll
    for (int i = 0; i < WMB; ++i) abv[i] = lnstat[wm * WMB * 16 + i * 16 + l15];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    float4_t cs[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) cs[j] = *reinterpret_cast<const float4_t*>(p.ln_cs + nw + j * 16);
#pragma unroll
    for (int i = 0; i < WMB; ++i) {
      const float2 ab = abv[i];
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaf(acc[i][j][r], ab.x, ab.y * cs[j][r]);
    }
  }

  // split-K: this block's partial sums go to its slab [split][M][N] rounded to f16 (see G160Params.ws) -- through the SAME staging
  // image and store pass as a finished tile (whole 16-byte chunks of contiguous row segments; the 8-byte stores straight from the
  // accumulators are the 2.3 TB/s form described below), only without bias / row vector / activation, which the reduction applies
  const bool slab = !GEGLU_ONLY && p.splits > 1;

  if (!GEGLU_ONLY && !slab && p.Ct && n0 >= p.n_split) {  // transposed tail (tile-uniform): Ct[(n - n_split) * ldct + m] (+ bias)
    float bv[NT][4];
    const half_t* bp = p.bias ? p.bias + nw : g_zero_page;   // unconditional 8-byte loads (see pass 1 below)
    const int bstep = p.bias ? 16 : 0;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      Pack8 b;
      b.u = gld<uint2>(bp + j * bstep);
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[j][r] = (float)b.e[r];
    }
#pragma unroll
    for (int i = 0; i < WMB; ++i) {
      const int m = mrow(lr0 + i * 16);
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)   // 16 lanes = 16 consecutive m: 32-byte segments per output row
          gst<half_t>(p.Ct + (long)(nw + j * 16 + r - p.n_split) * p.ldct + m, (half_t)(acc[i][j][r] + bv[j][r]));
    }
    return false;
  }

  // ---- pass 1: bias + per-sample row vector -> activation (or GEGLU) in registers, fp16 into the LDS image ----
  // (The accumulator layout gives a lane 4 consecutive columns of ONE row, 16 rows per instruction: stored to
  //  HBM directly that is 16 scattered 32-byte segments per wave instruction -- 8-byte stores at 2.3 TB/s,
  //  the GEGLU's 4-byte ones at 1.4 TB/s, measured; on the UNet's short-K linears that store pass was 40-60 %
  //  of the launch (profiles/r02_ring_and_ablation.log).  Through LDS the stores are whole 16-byte chunks of
  //  contiguous row segments and the residual is read the same way.)
  const bool geglu = GEGLU_ONLY || p.act == PFD_ACT_GEGLU;
  const int lrow = wm * WMB * 16 + l15;
  // Every global load of this pass is UNCONDITIONAL and issued before the first use: `if (ptr) v = load` is compiled
  // as branch + load + s_waitcnt vmcnt(0), so the NT bias loads and WMB x NT row-vector loads went out one L2 round
  // trip at a time (up to 25 per tile, on tiles whose whole K loop is 5 steps).  An absent bias reads the zero page
  // with stride 0; a row past M reads row M - 1 (its result is never stored).
  const bool has_bias = p.bias && !slab;
  const half_t* bp = has_bias ? p.bias + nw : g_zero_page;
  const int bstep = has_bias ? 16 : 0;
  Pack8 bq[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bq[j].u = gld<uint2>(bp + j * bstep);
  // (the bias stays packed f16 and is widened where it is added: 10 registers instead of 20 through the staging pass of
  //  the loader-wave kernels, which sit at their 168-register limit)
  if (geglu) {
    // packed weight rows come in groups of four: x(2c), x(2c+1), gate(2c), gate(2c+1) -- exactly the four
    // columns a lane owns, so out(2c..2c+1) = x * gelu(gate) needs no exchange
    constexpr int RS = stage_row_bytes(BN / 2);
#pragma unroll
    for (int i = 0; i < WMB; ++i) {
      char* sp = smem + (lrow + i * 16) * RS + (wn * (8 * NT) + 2 * g) * 2;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        half2_t o;
        o[0] = (half_t)((acc[i][j][0] + (float)bq[j].e[0]) * pfd_gelu(acc[i][j][2] + (float)bq[j].e[2]));
        o[1] = (half_t)((acc[i][j][1] + (float)bq[j].e[1]) * pfd_gelu(acc[i][j][3] + (float)bq[j].e[3]));
        *reinterpret_cast<half2_t*>(sp + j * 16) = o;
      }
    }
  } else if constexpr (!GEGLU_ONLY) {
    constexpr int RS = stage_row_bytes(BN);
    char* sp0 = smem + lrow * RS + (wn * (16 * NT) + 4 * g) * 2;
    // one copy of the staging loop per activation (p.act is launch-uniform): `if (p.act == ...) else if ...` per
    // element is a chain of scalar compares and branches per element, 80 elements per lane
    auto put_row = [&](auto act, auto itag, const Pack8(&rv)[NT]) __attribute__((always_inline)) {
      constexpr int ACT = decltype(act)::value;
      constexpr int i = decltype(itag)::value;
      char* sp = sp0 + i * 16 * RS;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        Pack8 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[i][j][r] + (float)bq[j].e[r] + (float)rv[j].e[r];
          if constexpr (ACT == PFD_ACT_GELU) v = pfd_gelu(v);
          else if constexpr (ACT == PFD_ACT_RELU) v = fmaxf(v, 0.f);
          else if constexpr (ACT == PFD_ACT_SILU) v = pfd_silu(v);
          o.e[r] = (half_t)v;
        }
        *reinterpret_cast<uint2*>(sp + j * 32) = o.u;
      }
    };
    auto load_rv = [&](Pack8(&rv)[NT], int m) __attribute__((always_inline)) {   // row vector of output row m (clamped)
      const half_t* rvp = p.rowvec + (long)(min(m, p.M - 1) / p.rows_per_rv) * p.ldrv + nw;
#pragma unroll
      for (int j = 0; j < NT; ++j) rv[j].u = gld<uint2>(rvp + j * 16);
    };
    auto for_rows = [&](auto&& f) __attribute__((always_inline)) {
      f(std::integral_constant<int, 0>{});
      if constexpr (WMB > 1) f(std::integral_constant<int, 1>{});
      if constexpr (WMB > 2) f(std::integral_constant<int, 2>{});
      if constexpr (WMB > 3) f(std::integral_constant<int, 3>{});
      static_assert(WMB <= 4, "row tiles per wave");
    };
    auto run = [&](auto act) __attribute__((always_inline)) {
      Pack8 cur[NT];
      const int first = mrow(wm * WMB * 16), last = mrow(wm * WMB * 16 + WMB * 16 - 1);   // this wave's first / last output row
      if (!p.rowvec || slab) {
#pragma unroll
        for (int j = 0; j < NT; ++j) cur[j].u = make_uint2(0, 0);
        for_rows([&](auto it) __attribute__((always_inline)) { put_row(act, it, cur); });
      } else if (PT || min(first, p.M - 1) / p.rows_per_rv == min(last, p.M - 1) / p.rows_per_rv) {
        // (a patch tile never leaves its sample: the per-lane form below is not even compiled for those kernels)
        // every row of this wave belongs to one sample (the convolutions' per-sample time embedding): NT loads
        load_rv(cur, first);
        for_rows([&](auto it) __attribute__((always_inline)) { put_row(act, it, cur); });
      } else if constexpr (!PT) {
        // rows of several samples: per-lane vectors, the next row tile's loads in flight under the current one
        Pack8 nxt[NT];
        load_rv(cur, mrow(lr0));
        for_rows([&](auto it) __attribute__((always_inline)) {
          constexpr int i = decltype(it)::value;
          if constexpr (i + 1 < WMB) load_rv(nxt, mrow(lr0 + (i + 1) * 16));
          put_row(act, it, cur);
          if constexpr (i + 1 < WMB) {
#pragma unroll
            for (int j = 0; j < NT; ++j) cur[j].u = nxt[j].u;
          }
        });
      }
    };
    switch (slab ? PFD_ACT_NONE : p.act) {
      case PFD_ACT_GELU: run(std::integral_constant<int, PFD_ACT_GELU>{}); break;
      case PFD_ACT_RELU: run(std::integral_constant<int, PFD_ACT_RELU>{}); break;
      case PFD_ACT_SILU: run(std::integral_constant<int, PFD_ACT_SILU>{}); break;
      default: run(std::integral_constant<int, PFD_ACT_NONE>{}); break;
    }
  }
  return true;
}

// ---- GroupNorm statistics from the producer (PfdGemmDesc.gn_out, round 4) ------------------------------------------
// Every GroupNorm input of the UNet is written by one of these store passes (or by the split-K reduce below), so the
// statistics pass of the two-launch GroupNorm (a full extra read of the tensor: 14.8 ms of a 532 ms batch) is replaced by
// partial sums the producer forms from the f16 values it stores: per slab of 64 consecutive rows of a tile and per group
// of cpg = N / 32 output channels, (sum x, sum x^2) -> gn_out[(slab * (N / 160) + n0 / 160) * 16 + local group].  The slab
// height is 64 whatever the tile height, so the layout (and the consumer, pfd_groupnorm_pstats_f16) does not depend on
// the tile variant the heuristic picked.  Fixed summation order: deterministic.
// Thread mapping: a wave covers 3 rows x 20 chunks of 16 bytes (lanes 60-63 idle), the block sweeps 3 W rows at a time;
// a lane keeps 8 column sums + 8 sums of squares per slab.  Reduction: the three lanes of a wave that own the same chunk
// (2 bpermutes per value), then across waves and over a group's columns through the slab's own, now dead, image region.
constexpr int GN_SLAB = 64;

// cs / cq: this lane's partial column sums of its chunk of ITS slab (wave w works on slab w / WPS, WPS = waves per slab,
// so a lane carries 16 sums whatever the tile height).  red_base(s): scratch of slab s (>= (WPS + 1) * 1280 bytes).
template <int NSLAB, int NTHREADS, class RedBase>
__device__ __forceinline__ void gn_slab_reduce(float (&cs)[8], float (&cq)[8], int tid, int cpg, int slab0, int nslab_total,
                                               int tiles_n, int tile_n, float2* __restrict__ out, RedBase red_base) {
  constexpr int W = NTHREADS / 64, WPS = W / NSLAB;
  static_assert(W % NSLAB == 0, "whole waves per slab");
  const int lane = tid & 63, w = tid >> 6;
  const int cc = lane % 20;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    cs[e] += __shfl(cs[e], (lane + 20) & 63, 64) + __shfl(cs[e], (lane + 40) & 63, 64);
    cq[e] += __shfl(cq[e], (lane + 20) & 63, 64) + __shfl(cq[e], (lane + 40) & 63, 64);
  }
  __syncthreads();   // every row of the staged image has been read: its slabs become scratch
  if (lane < 20) {
    float4_t* dst = reinterpret_cast<float4_t*>(red_base(w / WPS) + ((w % WPS) * 20 + cc) * 16);
    dst[0] = (float4_t){cs[0], cs[1], cs[2], cs[3]};
    dst[1] = (float4_t){cs[4], cs[5], cs[6], cs[7]};
    dst[2] = (float4_t){cq[0], cq[1], cq[2], cq[3]};
    dst[3] = (float4_t){cq[4], cq[5], cq[6], cq[7]};
  }
  __syncthreads();
  for (int idx = tid; idx < NSLAB * 320; idx += NTHREADS) {   // (slab, chunk, element): sum over the slab's waves, first wave first
    const int s = idx / 320, j = idx - s * 320;
    const float* r = red_base(s) + j;
    float a = 0.f;
#pragma unroll
    for (int ww = 0; ww < WPS; ++ww) a += r[ww * 320];
    red_base(s)[WPS * 320 + j] = a;
  }
  __syncthreads();
  const int ngl = 160 / cpg;
  for (int idx = tid; idx < NSLAB * ngl; idx += NTHREADS) {   // (slab, local group): its cpg columns in ascending order
    const int s = idx / ngl, gl = idx - s * ngl;
    const float* fin = red_base(s) + WPS * 320;
    float a = 0.f, q = 0.f;
    for (int c = gl * cpg; c < (gl + 1) * cpg; ++c) {
      a += fin[(c >> 3) * 16 + (c & 7)];
      q += fin[(c >> 3) * 16 + 8 + (c & 7)];
    }
    if (slab0 + s < nslab_total)   // (a tile past M holds slabs that do not exist)
      gst<float2>(out + ((long)(slab0 + s) * tiles_n + tile_n) * 16 + gl, make_float2(a, q));
  }
}

template <int BM, int NTHREADS, bool PT>
__device__ __forceinline__ void epilogue_store_gn(const G160Params& p, int m0, int n0, int slab0, char* smem, int tid) {
  constexpr int RS = stage_row_bytes(160);
  constexpr int NSLAB = BM / GN_SLAB, W = NTHREADS / 64, WPS = W / NSLAB, SWEEP = 3 * WPS;
  constexpr int ITERS = (GN_SLAB + SWEEP - 1) / SWEEP;
  // residual chunks in flight together (the 12-wave kernels sit at their 168-register limit: two)
  // (round 5: 4 / 8 chunks in flight in the 12-wave kernels compile without scratch and measure the same, profiles/r05_epilogue_depth_ab.log)
  constexpr int UB = NTHREADS >= 768 ? 2 : (ITERS < 4 ? ITERS : 4);
  static_assert(BM % GN_SLAB == 0 && W % NSLAB == 0 && (WPS + 1) * 1280 <= GN_SLAB * RS, "a slab's image region holds its scratch");
  const int lane = tid & 63, w = tid >> 6;
  const bool active = lane < 60;
  const int rsub = lane / 20, cc = lane % 20;          // (idle lanes: rsub 3, masked below)
  const int s = w / WPS;                               // this wave's slab
  const int row0 = s * GN_SLAB + (w % WPS) * 3 + rsub;
  const bool has_r = p.R != nullptr;
  float cs[8], cq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) cs[e] = cq[e] = 0.f;
  for (int it0 = 0; it0 < ITERS; it0 += UB) {
    Pack16 r[UB];
    // the residual chunks of this group first, unconditional (clamped row): see store_pass below
    if (has_r) {
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int row = min(row0 + (it0 + u) * SWEEP, (s + 1) * GN_SLAB - 1);
        r[u].u = gld<uint4>(p.R + res_row(p, tile_row_m<PT>(p, m0, row)) * p.ldr + n0 + cc * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int row_u = row0 + (it0 + u) * SWEEP;
      const int row = min(row_u, (s + 1) * GN_SLAB - 1);
      const int m = tile_row_m<PT>(p, m0, row);
      const bool ok = active && it0 + u < ITERS && row_u < (s + 1) * GN_SLAB && m < p.M;
      Pack16 v;
      v.u = *reinterpret_cast<const uint4*>(smem + row * RS + cc * 16);
      if (has_r) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v.e[e] = (half_t)((float)v.e[e] + (float)r[u].e[e]);
      }
      if (ok) gst<uint4>(p.C + (long)m * p.ldc + n0 + cc * 8, v.u);
      const float mk = ok ? 1.f : 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = (float)v.e[e] * mk;
        cs[e] += f;
        cq[e] = fmaf(f, f, cq[e]);
      }
    }
  }
  gn_slab_reduce<NSLAB, NTHREADS>(cs, cq, tid, p.N / 32, slab0, p.M / GN_SLAB, p.N / 160, n0 / 160, p.gn_out,
                                  [&](int sl) { return reinterpret_cast<float*>(smem + sl * GN_SLAB * RS); });
}

// pass 2 (every thread of the block, after a barrier): + residual, 16-byte chunks of contiguous row segments
// LNOUT: compile the statistics-emitting store pass (linear kernels only: convolutions never feed a LayerNorm, and the
// loader-wave kernels have no registers to spare for it)
template <int BM, int NT, int NTHREADS, bool LNOUT = false, bool PT = false>
__device__ __forceinline__ void epilogue_store(const G160Params& p, int m0, int n0, char* smem, int tid, int slab0 = 0) {
  constexpr int BN = 32 * NT;
  constexpr bool GEGLU_ONLY = NT == 10;
  const bool slab = !GEGLU_ONLY && p.splits > 1;   // split-K: the staged partial sums go to this block's slab, nothing else happens here
  if (!GEGLU_ONLY && !slab && p.Ct && n0 >= p.n_split) return;   // written directly by pass 1 (tile-uniform)
  const bool geglu = GEGLU_ONLY || (!slab && p.act == PFD_ACT_GEGLU);
  half_t* const Cb = slab ? reinterpret_cast<half_t*>(p.ws) + (long)blockIdx.z * p.M * p.N : p.C;
  const long ldc = slab ? (long)p.N : p.ldc;
  if constexpr (NT == 5) {
    if (p.gn_out && !slab) {   // the store pass that also forms the GroupNorm statistics of what it stores (host: act != GEGLU, no Ct)
      epilogue_store_gn<BM, NTHREADS, PT>(p, m0, n0, slab0, smem, tid);
      return;
    }
  }
  if constexpr (LNOUT && !GEGLU_ONLY && (BN / 8) % 4 == 0) {
    if (p.ln_out && !slab) {
      // This launch's output feeds a LayerNorm that is folded into ITS consumer GEMM: emit the partial row sums
      // (sum, sum of squares of the f16 values stored, i.e. exactly what a LayerNorm kernel would read) of the BN
      // columns this tile holds.  Four lanes per row, lane k takes chunks k, k + 4, ...: every store instruction still
      // writes 64 contiguous bytes per row; two shuffles finish a row.
      constexpr int RS = stage_row_bytes(BN);
      constexpr int CPL = BN / 8 / 4;             // chunks per lane: 5 (4)
      constexpr int ROWS_IT = NTHREADS / 4;
      const int k = tid & 3;
      const bool has_r = p.R != nullptr;
      const int tile_n = n0 / BN, tiles_n = p.N / BN;
      for (int row0 = 0; row0 < BM; row0 += ROWS_IT) {
        const int row = row0 + (tid >> 2);
        const int rowc = min(row, BM - 1);
        const int m = m0 + row;
        const bool ok = row < BM && m < p.M;
        Pack16 r[CPL];
        if (has_r) {
#pragma unroll
          for (int j = 0; j < CPL; ++j)
            r[j].u = gld<uint4>(p.R + res_row(p, m0 + rowc) * p.ldr + n0 + (k + 4 * j) * 8);
        }
        float sum = 0.f, sq = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          Pack16 v;
          v.u = *reinterpret_cast<const uint4*>(smem + rowc * RS + (k + 4 * j) * 16);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (has_r) v.e[e] = (half_t)((float)v.e[e] + (float)r[j].e[e]);
            const float f = (float)v.e[e];
            sum += f;
            sq = fmaf(f, f, sq);
          }
          if (ok) gst<uint4>(p.C + (long)m * p.ldc + n0 + (k + 4 * j) * 8, v.u);
        }
        sum += __shfl_xor(sum, 1, 64);
        sq += __shfl_xor(sq, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        sq += __shfl_xor(sq, 2, 64);
        if (ok && k == 0) gst<float2>(p.ln_out + (long)m * tiles_n + tile_n, make_float2(sum, sq));
      }
      return;
    }
  }
  auto store_pass = [&](auto cols_tag) {
    constexpr int COLS = decltype(cols_tag)::value;
    constexpr int RS = stage_row_bytes(COLS);
    constexpr int CPR = COLS / 8;   // 16-byte chunks per row
    const int nc0 = geglu ? n0 / 2 : n0;
    constexpr int TOTAL = BM * CPR;
    constexpr int ITERS = (TOTAL + NTHREADS - 1) / NTHREADS;
    constexpr int U = ITERS < 5 ? ITERS : 5;   // chunks per thread whose residual loads are in flight together
    const bool has_r = p.R != nullptr && !slab;
    for (int it0 = 0; it0 < ITERS; it0 += U) {
      Pack16 r[U];
      // the residual chunks of this group first, unconditional (clamped chunk / row): one load inside `if (p.R)` per
      // loop iteration is one L2 round trip per 16 bytes per thread
      if (has_r) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int c = min(tid + (it0 + u) * NTHREADS, TOTAL - 1);
          const int row = c / CPR, cc = c - row * CPR;
          r[u].u = gld<uint4>(p.R + res_row(p, tile_row_m<PT>(p, m0, row)) * p.ldr + nc0 + cc * 8);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = tid + (it0 + u) * NTHREADS;
        const int row = c / CPR, cc = c - row * CPR;
        const int m = tile_row_m<PT>(p, m0, row);
        if (it0 + u >= ITERS || c >= TOTAL || m >= p.M) continue;
        Pack16 v;
        v.u = *reinterpret_cast<const uint4*>(smem + row * RS + cc * 16);
        if (has_r) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v.e[e] = (half_t)((float)v.e[e] + (float)r[u].e[e]);
        }
        gst<uint4>(Cb + (long)m * ldc + nc0 + cc * 8, v.u);
      }
    }
  };
  if (geglu) store_pass(std::integral_constant<int, BN / 2>{});
  else if constexpr (!GEGLU_ONLY) store_pass(std::integral_constant<int, BN>{});
}

template <int WMB, int NT, int NTHREADS, bool LNOUT = false, bool PT = false>
__device__ __forceinline__ void epilogue160(float4_t (&acc)[WMB][NT], const G160Params& p, int lane, int m0,
                                            int n0, int wm, int wn, int split, char* smem, int tid,
                                            const float2* lnstat = nullptr, int slab0 = 0) {
  epilogue_stage<WMB, NT, PT>(acc, p, lane, m0, n0, wm, wn, split, smem, lnstat);
  __syncthreads();
  epilogue_store<(NTHREADS / 128) * WMB * 16, NT, NTHREADS, LNOUT, PT>(p, m0, n0, smem, tid, slab0);
}

template <int WAVES_M, int WMB, bool CONV, int NBUF, int NT>
__global__ __launch_bounds__(WAVES_M * 128) void gemm160_kernel(const G160Params p) {
  constexpr int BN = 32 * NT;   // 160 (every UNet / ControlNet width) or 128 (VAE, Swin, SeeCoder widths)
  constexpr int NW = WAVES_M * 2;
  constexpr int BM = WAVES_M * WMB * 16;
  constexpr int A_INSTR = BM / 8;                  // wave-instructions per A tile
  constexpr int B_INSTR = BN / 8;                  // 20 / 16
  constexpr int A_PER_WAVE = A_INSTR / NW;         // 4, 4, 2
  constexpr int B_PER_WAVE = (B_INSTR + NW - 1) / NW;
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int MAIN_BYTES = NBUF * STAGE;
  constexpr int DEPTH = NBUF - 1;  // K tiles in flight ahead of the one being consumed
  // counted vmcnt (NBUF > 2) needs the same DMA count in every wave: when the B pieces do not split evenly, the
  // surplus slots re-issue the piece NW below (same source, same destination)
  constexpr bool B_DUP = NBUF > 2 && B_INSTR % NW != 0;
  static_assert(MAIN_BYTES <= 160 * 1024, "operand ring exceeds the 160 KiB LDS");
  // NT = 10 (256 x 320 tile) serves the GEGLU projections only: its staged image is BN / 2 columns wide
  static_assert(BM * stage_row_bytes(NT == 10 ? BN / 2 : BN) <= MAIN_BYTES, "the epilogue's staging image reuses the operand ring");
  constexpr int SMEM = MAIN_BYTES;
  static_assert(A_INSTR % NW == 0, "A tile must split evenly over the waves");
  static_assert(WAVES_M * 128 >= BM, "one thread per row forms the LayerNorm statistics");
  // + BM x {rstd, -rstd mean} behind the operand ring (LayerNorm folded into the GEMM): ONE __shared__ object, a
  // second one would make hipcc drain the LDS-DMA queue in front of every fragment read (guide 5, trap (a))
  __shared__ __attribute__((aligned(1024))) char smem[SMEM + (CONV ? 0 : BM * 8)];
  float2* const lnstat = reinterpret_cast<float2*>(smem + SMEM);
  if constexpr (CONV) PFD_ARG_BATCH_CONV(p);
  else PFD_ARG_BATCH_LIN(p);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, g = lane >> 4;

  const int nblk = p.tiles_m * p.tiles_n;
  const int t = xcd_remap(blockIdx.x, nblk);
  // every XCD has its own L2 and works on one contiguous chunk of the tile order.  M-major order keeps an
  // activation panel XCD-local but makes every XCD stream ALL of W; when W is the bigger operand (the
  // 8^2 / 16^2 levels: 29.5 MB of weights against 1.3-5 MB of activations) the order is flipped so each
  // XCD streams 1/8 of W instead -- 3.4-6x less L2-miss traffic on exactly the weight-bound launches
  const int tile_m = p.nmajor ? t % p.tiles_m : t / p.tiles_n;
  const int tile_n = p.nmajor ? t / p.tiles_m : t - tile_m * p.tiles_n;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;
  const int split = blockIdx.z;
  const int kt_begin = split * p.kt_per_split;
  const int nk_total = p.K / BK;
  const int kt_end = min(nk_total, kt_begin + p.kt_per_split);

  // ---- staging state: this lane moves chunk position (lane & 7) of row (q*8 + lane>>3) ----
  const int srow = lane >> 3;
  const int cpos = lane & 7;
  const half_t* a_ptr[A_PER_WAVE];   // linear mode: pointer to (row, swizzled source chunk) at k = 0
  const half_t* a2_ptr[A_PER_WAVE];  // ... of the second source (columns >= k_split), see PfdGemmDesc.k_split
  int a_oy[A_PER_WAVE], a_ox[A_PER_WAVE];
  long a_img[A_PER_WAVE];
  bool a_ok[A_PER_WAVE];
  int a_chunk[A_PER_WAVE];
#pragma unroll
  for (int j = 0; j < A_PER_WAVE; ++j) {
    const int r = (wave + NW * j) * 8 + srow;
    const int c = cpos ^ ((r >> 1) & 7);
    a_chunk[j] = c * 8;
    const int m = m0 + r;
    a_ok[j] = m < p.M;
    if (CONV) {
      const int hw = p.Ho * p.Wo;
      const int b = m / hw;
      const int rem = m - b * hw;
      const int oy = rem / p.Wo;
      a_oy[j] = oy * p.stride - p.pad;
      a_ox[j] = (rem - oy * p.Wo) * p.stride - p.pad;
      a_img[j] = (long)b * p.H * p.Wd * p.lda;
      a_ptr[j] = a2_ptr[j] = p.A;
    } else {
      const int mz = m - p.zero_rows;           // A / A2 start at the first row that holds data
      a_ok[j] = a_ok[j] && mz >= 0;
      a_ptr[j] = a_ok[j] ? p.A + (long)mz * p.lda + c * 8 : g_zero_page;
      a2_ptr[j] = a_ok[j] ? p.A2 + (long)mz * p.lda2 + c * 8 : g_zero_page;
      a_oy[j] = a_ox[j] = 0;
      a_img[j] = 0;
    }
  }
  const half_t* b_ptr[B_PER_WAVE];
#pragma unroll
  for (int j = 0; j < B_PER_WAVE; ++j) {
    int q = wave + NW * j;
    if (q >= B_INSTR) q = B_DUP ? q - NW : 0;
    const int r = q * 8 + srow;
    const int c = cpos ^ ((r >> 1) & 7);
    b_ptr[j] = PFD_W_ROW_PTR(BN, p, tile_n, n0, r, c * 8);
  }
  const int Hin = p.ups ? 2 * p.H : p.H;
  const int Win = p.ups ? 2 * p.Wd : p.Wd;

  // K tiles are walked in rotated order (k_rotation): step i works on tile kt_begin + (i + rot) % nsteps
  // (a tile whose rows all lie below zero_rows has nothing to contract: epi(0))
  const int nsteps = (!CONV && m0 + BM <= p.zero_rows) ? 0 : kt_end - kt_begin;
  int kt_issue = kt_begin + k_rotation(p.krot, tile_m, p.tiles_m, nsteps);   // next tile to be issued (wave-uniform)
  // conv K walk: tap (ky, kx) outer, channel block inner
  int tap_ky = 0, tap_kx = 0, ci0 = 0;
  const half_t* a_tap[A_PER_WAVE];  // conv: source pointer for the current tap at ci = 0 (or zero page)
  auto set_tap = [&]() {
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j) {
      int iy = a_oy[j] + tap_ky, ix = a_ox[j] + tap_kx;
      const bool ok = a_ok[j] && iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
      if (p.ups) {
        iy >>= 1;
        ix >>= 1;
      }
      a_tap[j] = ok ? p.A + a_img[j] + ((long)iy * p.Wd + ix) * p.lda + a_chunk[j] : nullptr;
    }
  };
  auto seek = [&](int kt) {   // (tap, channel block) of K tile kt: once per block and once per wrap-around
    const int k0 = kt * BK;
    const int tap = k0 / p.Cin;
    ci0 = k0 - tap * p.Cin;
    tap_ky = tap / p.ksize;
    tap_kx = tap - tap_ky * p.ksize;
    set_tap();
  };
  if (CONV) seek(kt_issue);

  auto issue = [&](int stage) {
    char* As = smem + stage * STAGE;
    char* Bs = As + BM * ROWB;
    const int k0 = kt_issue * BK;
    const long kw = kt_issue * p.w_kstep;
    if (CONV) {
#pragma unroll
      for (int j = 0; j < A_PER_WAVE; ++j) {
        const half_t* src = a_tap[j] ? a_tap[j] + ci0 : g_zero_page;
        glds16(src, As + (wave + NW * j) * 1024);
      }
    } else {
      const bool first = k0 < p.k_split;   // wave-uniform: which source this K tile comes from
#pragma unroll
      for (int j = 0; j < A_PER_WAVE; ++j) {
        const half_t* src = first ? a_ptr[j] + k0 : a2_ptr[j] + (k0 - p.k_split);
        glds16(a_ok[j] ? src : a_ptr[j], As + (wave + NW * j) * 1024);
      }
    }
#pragma unroll
    for (int j = 0; j < B_PER_WAVE; ++j) {
      const int q = wave + NW * j;
      if (q < B_INSTR) glds16(b_ptr[j] + kw, Bs + q * 1024);
      else if (B_DUP) glds16(b_ptr[j] + kw, Bs + (q - NW) * 1024);
    }
    if (++kt_issue == kt_end) {  // wrap-around of the rotated walk (wave-uniform)
      kt_issue = kt_begin;
      if (CONV) seek(kt_begin);
    } else if (CONV) {
      ci0 += BK;
      if (ci0 >= p.Cin) {  // next tap (wave-uniform)
        ci0 = 0;
        if (++tap_kx == p.ksize) {
          tap_kx = 0;
          ++tap_ky;
        }
        set_tap();
      }
    }
  };

  float4_t acc[WMB][NT];
#pragma unroll
  for (int i = 0; i < WMB; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes): row (.. + l15), chunk (ks*4 + g) ^ ((l15 >> 1) & 7)
  const int sw = (l15 >> 1) & 7;
  const int off_k0 = ((0 + g) ^ sw) * 16 + l15 * ROWB;
  const int off_k1 = ((4 + g) ^ sw) * 16 + l15 * ROWB;
  const int a_row0 = wm * WMB * 16 * ROWB;
  const int b_row0 = BM * ROWB + wn * (16 * NT) * ROWB;

  // (Also A/B-tested and dropped: the same 256x160 tile as 4 waves of 128x80 -- 28 % fewer LDS fragment
  //  bytes per MFMA but one wave per SIMD, so nothing covers the ds_read latency: conv 32^2 106 -> 163 us,
  //  profiles/r01_gemm_replay_variants.log.)
  // Operand ring of NBUF stages, DEPTH = NBUF-1 K tiles in flight.  NBUF = 2 is the throughput form
  // (one tile ahead; a deeper ring A/B-tested slower on the big warm problems: conv 64^2 893 -> 803 TF,
  // profiles/r01_selftest_ring_ab.log).  Problems that fill the chip only once (<= 256 blocks) and stream
  // COLD weights are bound by the DMA round trip per K tile instead (~1.9 us per tile on the 8^2 convs),
  // so they run NBUF = 4-5 with a counted vmcnt wait: only the oldest tile has to have landed.
  if (nsteps > 0) {
    constexpr int PER_STEP = A_PER_WAVE + B_PER_WAVE;
    constexpr int KEEP = PER_STEP * (DEPTH - 1);  // DMA instructions that may stay outstanding
    static_assert(KEEP < 64, "vmcnt is 6 bits");
    constexpr int WAIT_KEEP = (KEEP & 15) | ((KEEP >> 4) << 14) | (7 << 4) | (15 << 8);
#pragma unroll
    for (int s = 0; s < DEPTH; ++s)
      if (s < nsteps) issue(s);
    if constexpr (!CONV) {
      if (p.ln_in && tid < BM) {   // row statistics of this block's rows from the producer's partial sums
        const int m = min(m0 + tid, p.M - 1);
        const float2* src = p.ln_in + (long)m * p.ln_P;
        float sum = 0.f, sq = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {   // all loads in flight, unconditional (ln_P <= 8)
          const float2 v = src[min(j, p.ln_P - 1)];
          const float w = j < p.ln_P ? 1.f : 0.f;
          sum = fmaf(v.x, w, sum);
          sq = fmaf(v.y, w, sq);
        }
        const float inv_k = 1.0f / (float)p.K;
        const float mean = sum * inv_k;
        const float var = fmaxf(fmaf(-mean, mean, sq * inv_k), 0.f);
        const float rstd = rsqrtf(var + p.ln_eps);
        lnstat[tid] = make_float2(rstd, -rstd * mean);
      }
    }
    int buf = 0;
    for (int st = 0; st < nsteps; ++st) {
      // tile kt visible to all waves; everyone is done with the buffer of tile kt-1
      if constexpr (NBUF > 2) {
        // counted wait: only the OLDEST tile has to have landed.  The barrier must be the raw instruction:
        // __syncthreads() carries a fence that hipcc lowers to s_waitcnt vmcnt(0) -- every LDS-DMA piece is a pending
        // LDS write on the VM counter -- which drains the whole ring once per K step (that is how the round-1 ring
        // "measured slower": it never had more than one tile in flight)
        if (st + DEPTH - 1 < nsteps) __builtin_amdgcn_s_waitcnt(WAIT_KEEP);
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      } else {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
      }
      if (st + DEPTH < nsteps) {
        int nb = buf + DEPTH;
        if (nb >= NBUF) nb -= NBUF;
        issue(nb);
      }
      const char* base = smem + buf * STAGE;
      constexpr int NTH = NT > 5 ? 5 : NT;   // B fragments held at a time (the 320-wide tile walks its columns in halves)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int off = ks ? off_k1 : off_k0;
        half8_t af[WMB];
#pragma unroll
        for (int i = 0; i < WMB; ++i)
          af[i] = *reinterpret_cast<const half8_t*>(base + a_row0 + i * 16 * ROWB + off);
#pragma unroll
        for (int jh = 0; jh < NT; jh += NTH) {
          half8_t bf[NTH];
#pragma unroll
          for (int j = 0; j < NTH; ++j)
            bf[j] = *reinterpret_cast<const half8_t*>(base + b_row0 + (jh + j) * 16 * ROWB + off);
#pragma unroll
          for (int i = 0; i < WMB; ++i)
#pragma unroll
            for (int j = 0; j < NTH; ++j)
              acc[i][jh + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][jh + j], 0, 0, 0);
        }
      }
      if (++buf == NBUF) buf = 0;
    }
    __syncthreads();  // the epilogue reuses the ring as staging space
  }

  epilogue160<WMB, NT, WAVES_M * 128, !CONV>(acc, p, lane, m0, n0, wm, wn, split, smem, tid,
                                             (!CONV && p.ln_in) ? lnstat : nullptr, tile_m * (BM / GN_SLAB));
}

// ------------------------------------------------------------------------------------------------
// Wave-specialised form of the 256 x BN tile kernel: 8 consumer waves (the 4 x 2 MFMA layout above) that
// never touch VMEM in the K loop + 4 loader waves (one per SIMD) that do nothing but issue the LDS-DMA pieces.
// Why: on the kernel above, DMA-only (33 us) and MFMA-only (29 us) times of the GEGLU GEMM simply ADD to the
// 59 us of the full K loop (profiles/r02_ring_and_ablation.log) -- a wave that is stuck in VMEM issue (a 1 KiB
// LDS-DMA piece costs 100-185 issue cycles inside a busy phase) cannot issue its MFMAs, and since every wave
// passes the same barrier, all eight are stuck at the same time, whether the pieces sit behind the barrier or
// are interleaved with the MFMAs (both measured).  With the pieces on their own waves the consumers' issue
// stream is ds_read + MFMA only.  52 (48) pieces per K tile = 13 (12) per loader.  One raw s_barrier per K
// step for all 12 waves: loaders arrive after vmcnt(0) (their pieces of tile kt landed), consumers after the
// MFMAs of tile kt - 1; then loaders refill the buffer the consumers just left.
// ------------------------------------------------------------------------------------------------
// (Ping-pong consumer groups -- the two halves of the consumer waves one barrier interval apart, four intervals per K
//  step -- measured no better than lock-step consumers in round 3, profiles/r03_krot_pp_replay.log, and were removed in round 5.)
// NST: operand stages.  2 = one K tile ahead (loaders wait vmcnt(0) per K step); 3 = two K tiles ahead with a counted
// vmcnt (round 4; 156 KiB of LDS at the 160-wide tile), lock-step consumers only.
template <bool CONV, int NT, int NST = 2>
__global__ __launch_bounds__(768) void gemm160ws_kernel(const G160Params p) {
  constexpr int WMB = 4, NCW = 8, NLW = 4;
  constexpr int BN = 32 * NT, BM = 256;
  constexpr int A_INSTR = BM / 8, B_INSTR = BN / 8;
  constexpr int A_PL = A_INSTR / NLW, B_PL = B_INSTR / NLW;      // pieces per loader wave: 8 + 5 | 4
  static_assert(A_INSTR % NLW == 0 && B_INSTR % NLW == 0, "pieces must split evenly over the loader waves");
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int SMEM = NST * STAGE;
  static_assert(NST == 2 || NST == 3, "operand stages");
  static_assert(SMEM <= 160 * 1024, "LDS");
  static_assert(BM * stage_row_bytes(BN) <= SMEM, "the epilogue's staging image reuses the operand ring");
  __shared__ __attribute__((aligned(1024))) char smem[SMEM];
  if constexpr (CONV) PFD_ARG_BATCH_CONV(p);
  else PFD_ARG_BATCH_LIN(p);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = p.tiles_m * p.tiles_n;
  const int t = xcd_remap(blockIdx.x, nblk);
  const int tile_m = p.nmajor ? t % p.tiles_m : t / p.tiles_n;
  const int tile_n = p.nmajor ? t / p.tiles_m : t - tile_m * p.tiles_n;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;
  const int split = blockIdx.z;
  const int kt_begin = split * p.kt_per_split;
  const int kt_end = min(p.K / BK, kt_begin + p.kt_per_split);
  const int nsteps = kt_end - kt_begin;

  auto block_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  if (wave >= NCW) {
    // ================================ loader waves ================================
    const int lw = wave - NCW;
    const int srow = lane >> 3, cpos = lane & 7;
    const half_t* a_ptr[A_PL];
    int a_oy[A_PL], a_ox[A_PL];
    long a_img[A_PL];
    bool a_ok[A_PL];
    int a_chunk[A_PL];
#pragma unroll
    for (int j = 0; j < A_PL; ++j) {
      const int r = (lw + NLW * j) * 8 + srow;
      const int c = cpos ^ ((r >> 1) & 7);
      a_chunk[j] = c * 8;
      const int m = m0 + r;
      a_ok[j] = m < p.M;
      if (CONV) {
        const int hw = p.Ho * p.Wo;
        const int b = m / hw;
        const int rem = m - b * hw;
        const int oy = rem / p.Wo;
        a_oy[j] = oy * p.stride - p.pad;
        a_ox[j] = (rem - oy * p.Wo) * p.stride - p.pad;
        a_img[j] = (long)b * p.H * p.Wd * p.lda;
        a_ptr[j] = p.A;
      } else {
        a_ptr[j] = a_ok[j] ? p.A + (long)m * p.lda + c * 8 : g_zero_page;
        a_oy[j] = a_ox[j] = 0;
        a_img[j] = 0;
      }
    }
    const half_t* b_ptr[B_PL];
#pragma unroll
    for (int j = 0; j < B_PL; ++j) {
      const int r = (lw + NLW * j) * 8 + srow;
      const int c = cpos ^ ((r >> 1) & 7);
      b_ptr[j] = PFD_W_ROW_PTR(BN, p, tile_n, n0, r, c * 8);
    }
    const int Hin = p.ups ? 2 * p.H : p.H;
    const int Win = p.ups ? 2 * p.Wd : p.Wd;
    int kt_issue = kt_begin + k_rotation(p.krot, tile_m, p.tiles_m, nsteps);   // rotated K walk, see k_rotation
    int tap_ky = 0, tap_kx = 0, ci0 = 0;
    const half_t* a_tap[A_PL];
    auto set_tap = [&]() {
#pragma unroll
      for (int j = 0; j < A_PL; ++j) {
        int iy = a_oy[j] + tap_ky, ix = a_ox[j] + tap_kx;
        const bool ok = a_ok[j] && iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
        if (p.ups) {
          iy >>= 1;
          ix >>= 1;
        }
        a_tap[j] = ok ? p.A + a_img[j] + ((long)iy * p.Wd + ix) * p.lda + a_chunk[j] : nullptr;
      }
    };
    auto seek = [&](int kt) {
      const int k0 = kt * BK;
      const int tap = k0 / p.Cin;
      ci0 = k0 - tap * p.Cin;
      tap_ky = tap / p.ksize;
      tap_kx = tap - tap_ky * p.ksize;
      set_tap();
    };
    if (CONV) seek(kt_issue);
    auto issue = [&](int stage) {
      char* As = smem + stage * STAGE;
      char* Bs = As + BM * ROWB;
      const int k0 = kt_issue * BK;
      const long kw = kt_issue * p.w_kstep;
#pragma unroll
      for (int j = 0; j < A_PL; ++j) {
        const half_t* src;
        if (CONV) src = a_tap[j] ? a_tap[j] + ci0 : g_zero_page;
        else src = a_ok[j] ? a_ptr[j] + k0 : a_ptr[j];
        glds16(src, As + (lw + NLW * j) * 1024);
      }
#pragma unroll
      for (int j = 0; j < B_PL; ++j) glds16(b_ptr[j] + kw, Bs + (lw + NLW * j) * 1024);
      if (++kt_issue == kt_end) {
        kt_issue = kt_begin;
        if (CONV) seek(kt_begin);
      } else if (CONV) {
        ci0 += BK;
        if (ci0 >= p.Cin) {
          ci0 = 0;
          if (++tap_kx == p.ksize) {
            tap_kx = 0;
            ++tap_ky;
          }
          set_tap();
        }
      }
    };
    if constexpr (NST == 3) {
      constexpr int PER_STEP = A_PL + B_PL;   // 13 (12): the pieces of tile s + 1 may stay in flight behind barrier (A) of s
      static_assert(PER_STEP < 16, "one s_waitcnt immediate");
      if (nsteps > 0) issue(0);
      if (nsteps > 1) issue(1);
      int nst = 2;
      for (int s = 0; s < nsteps; ++s) {
        if (s + 1 < nsteps) __builtin_amdgcn_s_waitcnt(0x0F70 | PER_STEP);
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        block_barrier();                      // (A) tile s complete; every consumer is done reading tile s - 1
        if (s + 2 < nsteps) issue(nst);       // into the stage tile s - 1 just left
        if (++nst == 3) nst = 0;
      }
    } else {
    if (nsteps > 0) issue(0);
    for (int s = 0; s < nsteps; ++s) {
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's pieces of K tile s are in LDS
      block_barrier();                      // (A) tile s complete; every consumer is done reading tile s - 1
      if (s + 1 < nsteps) issue((s + 1) & 1);
    }
    }
    block_barrier();                        // (B) consumers finished reading the last tile: LDS is free
    block_barrier();                        // (C) the staging image is written
  } else {
    // ================================ consumer waves ================================
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;
    float4_t acc[WMB][NT];
#pragma unroll
    for (int i = 0; i < WMB; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    const int sw = (l15 >> 1) & 7;
    const int off_k0 = ((0 + g) ^ sw) * 16 + l15 * ROWB;
    const int off_k1 = ((4 + g) ^ sw) * 16 + l15 * ROWB;
    const int a_row0 = wm * WMB * 16 * ROWB;
    const int b_row0 = BM * ROWB + wn * (16 * NT) * ROWB;
    {
      int cst = 0;
      for (int s = 0; s < nsteps; ++s) {
        block_barrier();                      // (A)
        const char* base = smem + cst * STAGE;
        if (++cst == NST) cst = 0;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int off = ks ? off_k1 : off_k0;
          half8_t af[WMB], bf[NT];
#pragma unroll
          for (int i = 0; i < WMB; ++i)
            af[i] = *reinterpret_cast<const half8_t*>(base + a_row0 + i * 16 * ROWB + off);
#pragma unroll
          for (int j = 0; j < NT; ++j)
            bf[j] = *reinterpret_cast<const half8_t*>(base + b_row0 + j * 16 * ROWB + off);
#pragma unroll
          for (int i = 0; i < WMB; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
      }
      block_barrier();                        // (B)
    }
    epilogue_stage<WMB, NT>(acc, p, lane, m0, n0, wm, wn, split, smem);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    block_barrier();                          // (C)
  }
  epilogue_store<BM, NT, 768>(p, m0, n0, smem, tid, tile_m * (BM / GN_SLAB));
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution, "patch" form.  The LDS-DMA path sustains only ~50-70 GB/s per CU
// (profiles/r01_gemm_ablation_loads_vs_mfma.log), so the implicit-GEMM kernel above -- which re-stages
// the A tile for every one of the nine taps, 53 KB per K step -- is DMA-bound at ~36 % MFMA
// utilisation.  Here the block stages the input PATCH of its 256 output pixels (TH = 256/W full image
// rows plus a one-pixel halo, <= 400 pixels x 64 channels = 50 KB) ONCE per 64-channel block and all
// nine taps read their A fragments from it at shifted LDS rows; only the 20 KB weight tile moves per
// K step (25.5 KB per step on average, 2.1x less DMA traffic).  Padding = zero-page source rows.
// Same 4 x 2 wave layout, fragment swizzle, epilogue and split-K (over channel blocks) as above.
// Requires W in {16, 32, 64}, H % (256/W) == 0.
// ------------------------------------------------------------------------------------------------
constexpr int PATCH_ROWS = 400;

__global__ __launch_bounds__(512) void conv3x3_patch_kernel(const G160Params p) {
  constexpr int NW = 8, WMB = 4;
  constexpr int PATCH_BYTES = PATCH_ROWS * ROWB;  // 51200
  constexpr int WT_BYTES = BN * ROWB;             // 20480
  constexpr int OFF_W = 2 * PATCH_BYTES;
  constexpr int MAIN_BYTES = OFF_W + 2 * WT_BYTES;  // 143360
  constexpr int SMEM = MAIN_BYTES;
  constexpr int P_INSTR = PATCH_ROWS / 8;   // 50 DMA instructions per patch
  constexpr int P_SLOTS = (P_INSTR + NW - 1) / NW;  // 7 per wave
  __shared__ __attribute__((aligned(1024))) char smem[SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, g = lane >> 4;
  const int nblk = p.tiles_m * p.tiles_n;
  const int t = xcd_remap(blockIdx.x, nblk);
  // every XCD has its own L2 and works on one contiguous chunk of the tile order.  M-major order keeps an
  // activation panel XCD-local but makes every XCD stream ALL of W; when W is the bigger operand (the
  // 8^2 / 16^2 levels: 29.5 MB of weights against 1.3-5 MB of activations) the order is flipped so each
  // XCD streams 1/8 of W instead -- 3.4-6x less L2-miss traffic on exactly the weight-bound launches
  const int tile_m = p.nmajor ? t % p.tiles_m : t / p.tiles_n;
  const int tile_n = p.nmajor ? t / p.tiles_m : t - tile_m * p.tiles_n;
  const int n0 = tile_n * BN;
  const int split = blockIdx.z;
  const int ncb = p.Cin / BK;
  const int cb_begin = split * p.kt_per_split;
  const int cb_end = min(ncb, cb_begin + p.kt_per_split);

  // output tile: TH x TW pixels (TW = the image width when a tile is whole image rows, else p.pt_w)
  const int W = p.Wd, H = p.H;
  const int TW = p.pt_w ? p.pt_w : W;
  const int TH = 256 / TW, PW = TW + 2;
  const int hw = H * W;
  const int tpi = hw / 256, ntx = W / TW;
  const int b = tile_m / tpi;
  const int ti = tile_m - b * tpi;
  const int y0 = (ti / ntx) * TH, x0 = (ti % ntx) * TW;
  const int m0 = b * hw + y0 * W + x0;          // first output pixel of the tile (see tile_row_m)
  const int prow_count = (TH + 2) * PW;
  const half_t* img = p.A + (long)b * hw * p.lda;

  // patch staging: instruction q = wave + 8*j covers LDS rows q*8 .. q*8+7 (one input pixel each)
  const int srow = lane >> 3, cpos = lane & 7;
  const half_t* pp[P_SLOTS];
#pragma unroll
  for (int j = 0; j < P_SLOTS; ++j) {
    const int r = (wave + NW * j) * 8 + srow;
    const int py = r / PW, px = r - py * PW;
    const int y = y0 - 1 + py, x = x0 - 1 + px;
    const int c = (cpos - (r & ~1)) & 7;   // rotation swizzle of the patch rows, see the fragment read below
    const bool ok = r < prow_count && y >= 0 && y < H && x >= 0 && x < W;
    pp[j] = ok ? img + ((long)y * W + x) * p.lda + c * 8 : nullptr;
  }
  // weight tile staging (as in gemm160_kernel)
  const half_t* wp[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int q = wave + NW * j;
    const int r = q * 8 + srow;
    const int c = cpos ^ ((r >> 1) & 7);
    wp[j] = w_row_ptr(p, n0 + (q < 20 ? r : 0), c * 8);
  }
  auto issue_patch_slot = [&](int buf, int cb, int j) {
    const int q = wave + NW * j;
    if (q < P_INSTR) glds16(pp[j] ? pp[j] + cb * BK : g_zero_page, smem + buf * PATCH_BYTES + q * 1024);
  };
  auto issue_w = [&](int stage, int tap, int cb) {
    const long k0 = (long)(tap * (p.Cin / BK) + cb) * p.w_kstep;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int q = wave + NW * j;
      if (q < 20) glds16(wp[j] + k0, smem + OFF_W + stage * WT_BYTES + q * 1024);
    }
  };

  // A fragment rows: output pixel (wm*64 + i*16 + l15) of the tile -> patch row of its (0,0) tap
  int rbase[WMB];
#pragma unroll
  for (int i = 0; i < WMB; ++i) {
    const int ql = wm * 64 + i * 16;
    const int ty = ql / TW, tx = ql - ty * TW;
    rbase[i] = ty * PW + tx + l15;
  }
  const int sw = (l15 >> 1) & 7;
  const int boff0 = wn * 80 * ROWB + l15 * ROWB + (((0 + g) ^ sw) << 4);
  const int boff1 = wn * 80 * ROWB + l15 * ROWB + (((4 + g) ^ sw) << 4);

  float4_t acc[WMB][5];
#pragma unroll
  for (int i = 0; i < WMB; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

  if (cb_begin < cb_end) {
#pragma unroll
    for (int j = 0; j < P_SLOTS; ++j) issue_patch_slot(0, cb_begin, j);
    issue_w(0, 0, cb_begin);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    int stage = 0;
    for (int cb = cb_begin; cb < cb_end; ++cb) {
      const int pbuf = (cb - cb_begin) & 1;
      const char* patch = smem + pbuf * PATCH_BYTES;
      int toff = 0;  // ky*PW + kx
#pragma nounroll
      for (int ky = 0; ky < 3; ++ky) {
#pragma nounroll
        for (int kx = 0; kx < 3; ++kx) {
          const int tap = ky * 3 + kx;
          // next weight tile, and one slot of the next channel block's patch
          if (tap < 8) issue_w(stage ^ 1, tap + 1, cb);
          else if (cb + 1 < cb_end) issue_w(stage ^ 1, 0, cb + 1);
          if (cb + 1 < cb_end && tap < P_SLOTS) issue_patch_slot(pbuf ^ 1, cb + 1, tap);
          const char* wt = smem + OFF_W + stage * WT_BYTES;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            half8_t af[WMB], bf[5];
#pragma unroll
            for (int i = 0; i < WMB; ++i) {
              // The nine taps read the SAME stored patch at nine different row offsets, so the 16 rows of
              // a fragment start anywhere.  ds_read_b128 is served in four fixed 16-lane groups that mix the
              // k-chunks of lane rows {0-3, 12-15} of one g with rows {4-11} of the next g; the XOR swizzle
              // c ^ (row >> 1) of the GEMM kernel is conflict-free only for 16-aligned starts (PMC: 23 % of the
              // LDS cycles of this kernel were bank conflicts), no XOR-by-row function is for every start, but
              // the ROTATION pos = (chunk + 2 * (row >> 1)) mod 8 is: within a parity class the four rows of
              // one sub-set always land on the four even positions + chunk and the other four on the odd ones.
              const int r = rbase[i] + toff;
              af[i] = *reinterpret_cast<const half8_t*>(patch + r * ROWB + ((((ks * 4 + g) + (r & ~1)) & 7) << 4));
            }
            const int bo = ks ? boff1 : boff0;
#pragma unroll
            for (int j = 0; j < 5; ++j) bf[j] = *reinterpret_cast<const half8_t*>(wt + bo + j * 16 * ROWB);
#pragma unroll
            for (int i = 0; i < WMB; ++i)
#pragma unroll
              for (int j = 0; j < 5; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
          }
          __syncthreads();  // carries vmcnt(0)
          stage ^= 1;
          toff += 1;
        }
        toff += PW - 3;
      }
    }
  }
  epilogue160<WMB, 5, 512, false, true>(acc, p, lane, m0, n0, wm, wn, split, smem, tid, nullptr, tile_m * (256 / GN_SLAB));
}

// ------------------------------------------------------------------------------------------------
// Wave-specialised patch kernel: the 8 consumer waves of conv3x3_patch_kernel + 4 loader waves (one per SIMD)
// that issue every LDS-DMA piece -- per tap 20 weight pieces (5 per loader) and, for the next channel block's
// patch, slots [6 tap, 6 tap + 6) of its 50 pieces.  Same LDS layout, swizzles, barrier count (one per tap, all
// 12 waves) and epilogue; the consumers' instruction stream is ds_read + MFMA only.
// ------------------------------------------------------------------------------------------------
// GN: 0 = plain input, 1 = GroupNorm affine map in the staging path, 2 = affine map + SiLU
// NWS: weight stages.  2 = one tap ahead, loaders wait vmcnt(0) per tap (a tap's 20 KB weight tile has one tap of MFMA
//      work, ~0.55 us, to arrive -- less than the loaded LDS-DMA round trip, so every tap ends in a wait);
//      3 = two taps ahead with counted vmcnt (round 4): the whole 160 KiB of LDS (2 patches + 3 weight tiles).
template <int GN, int NWS = 2>
__global__ __launch_bounds__(768) void conv3x3_patch_ws_kernel(const G160Params p) {
  constexpr int NCW = 8, NLW = 4, WMB = 4;
  constexpr int PATCH_BYTES = PATCH_ROWS * ROWB;  // 51200
  constexpr int WT_BYTES = BN * ROWB;             // 20480
  constexpr int OFF_W = 2 * PATCH_BYTES;
  constexpr int SMEM = OFF_W + NWS * WT_BYTES;    // 143360 | 163840
  static_assert(NWS == 2 || (NWS == 3 && GN == 0), "the 3-stage weight ring serves the plain form");
  static_assert(SMEM <= 160 * 1024, "LDS");
  constexpr int P_INSTR = PATCH_ROWS / 8;          // 50 DMA pieces per patch
  __shared__ __attribute__((aligned(1024))) char smem[SMEM];
  PFD_ARG_BATCH_CONV(p);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = p.tiles_m * p.tiles_n;
  const int t = xcd_remap(blockIdx.x, nblk);
  const int tile_m = p.nmajor ? t % p.tiles_m : t / p.tiles_n;
  const int tile_n = p.nmajor ? t / p.tiles_m : t - tile_m * p.tiles_n;
  const int n0 = tile_n * BN;
  const int split = blockIdx.z;
  const int ncb = p.Cin / BK;
  const int cb_begin = split * p.kt_per_split;
  const int cb_end = min(ncb, cb_begin + p.kt_per_split);
  // output tile: TH x TW pixels (TW = the image width when a tile is whole image rows, else p.pt_w)
  const int W = p.Wd, H = p.H;
  const int hw = H * W;
  // (Round 5 also ran 8 x 8 images through this kernel -- tiles of four whole samples, 4 x 100 patch rows -- instead of the
  //  64 x 160 ring kernels: 43 vs 35 us per 512 x 1280 x 11520 convolution with 7 splits, +1.1 % per batch; with 10 / 16
  //  splits still +0.5 %: 112-160 blocks of a kernel with a ~10 us fixed cost do not beat 256 blocks of latency-chained
  //  ring tiles.  Removed; profiles/r05_patch8_ab.log.)
  const int TW = p.pt_w ? p.pt_w : W;
  const int TH = 256 / TW, PW = TW + 2;
  const int tpi = hw / 256, ntx = W / TW;
  const int b = tile_m / tpi;
  const int ti = tile_m - b * tpi;
  const int y0 = (ti / ntx) * TH, x0 = (ti % ntx) * TW;
  const int m0 = b * hw + y0 * W + x0;          // first output pixel of the tile (see tile_row_m)
  const int ncbs = max(0, cb_end - cb_begin);
  const int nsteps = ncbs * 9;
  // (the rotated K walk of the linear kernels -- k_rotation -- measured 2-3 % slower here and is not compiled in: the
  //  loader setup of this kernel is at its scalar-register limit)
  auto cbv = [&](int i) -> int { return cb_begin + i; };

  auto block_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  if (wave >= NCW) {
    // ================================ loader waves ================================
    const int lw = wave - NCW;
    const int srow = lane >> 3, cpos = lane & 7;
    const int prow_count = (TH + 2) * PW;
    const half_t* img = p.A + (long)b * hw * p.lda;
    // patch piece q covers LDS rows 8 q .. 8 q + 7; this loader owns q = 6 tap + lw (all loaders) and
    // q = 6 tap + 4 + lw (loaders 0-1), tap = 0..8, q < 50
    auto piece_off = [&](int q) -> int {   // element offset of this lane's source chunk from `img`, -1 = zero padding
      const int r = q * 8 + srow;
      const int py = r / PW, px = r - py * PW;
      const int y = y0 - 1 + py, x = x0 - 1 + px;
      const int c = (cpos - (r & ~1)) & 7;   // rotation swizzle of the patch rows
      const bool ok = q < P_INSTR && r < prow_count && y >= 0 && y < H && x >= 0 && x < W;
      return ok ? (int)((y * W + x) * p.lda + c * 8) : -1;
    };
    int off_a[9], off_b[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      off_a[tp] = piece_off(6 * tp + lw);
      off_b[tp] = piece_off(6 * tp + 4 + lw);
    }
    auto issue_patch = [&](int buf, int cb, int q, int off) {
      glds16(off >= 0 ? (const void*)(img + off + cb * BK) : (const void*)g_zero_page,
             smem + buf * PATCH_BYTES + q * 1024);
    };
    const half_t* wp[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int r = (lw + NLW * j) * 8 + srow;
      const int c = cpos ^ ((r >> 1) & 7);
      wp[j] = PFD_W_ROW_PTR(BN, p, tile_n, n0, r, c * 8);
    }
    auto issue_w = [&](int stage, int tap, int cb) {
      const long k0 = (long)(tap * (p.Cin / BK) + cb) * p.w_kstep;
#pragma unroll
      for (int j = 0; j < 5; ++j) glds16(wp[j] + k0, smem + OFF_W + stage * WT_BYTES + (lw + NLW * j) * 1024);
    };
    if constexpr (GN != 0) {
      // GroupNorm-apply (+ SiLU) in the staging path: the patch pieces go global -> VGPR -> affine map / activation in
      // fp32 -> fp16 -> ds_write (same lane-linear image the DMA writes), the weights stay on LDS-DMA.  A lane's
      // chunk index c = (cpos - (srow & ~1)) & 7 does not depend on the piece (8 q is a multiple of 8), so it maps
      // the SAME eight channels of every channel block: their {scale, shift} pairs are loaded once per block.
      // Pixels are requested one tap ahead of their transform, so a load has a whole tap (~1 us) in flight.
      const int c8 = ((cpos - (srow & ~1)) & 7) * 8;
      const float* tab = p.gn_table + (long)b * 2 * p.Cin + c8;
      const half_t* img2 = p.A2 + (long)b * hw * p.lda2;
      auto piece_pix = [&](int q) -> int {   // pixel index of this lane's patch row inside the sample, -1 = zero padding
        const int r = q * 8 + srow;
        const int py = r / PW, px = r - py * PW;
        const int y = y0 - 1 + py, x = x0 - 1 + px;
        const bool ok = q < P_INSTR && r < prow_count && y >= 0 && y < H && x >= 0 && x < W;
        return ok ? y * W + x : -1;
      };
      int pix_a[9], pix_b[9];
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        pix_a[tp] = piece_pix(6 * tp + lw);
        pix_b[tp] = piece_pix(6 * tp + 4 + lw);
      }
      float4_t sc[2], sh[2];                  // this lane's eight {scale}, {shift}: pairs of channels sit in
      auto load_tab = [&](int cb) {           // adjacent registers, so the affine map is four v_pk_fma_f32
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          sc[j] = *reinterpret_cast<const float4_t*>(tab + cb * BK + 4 * j);
          sh[j] = *reinterpret_cast<const float4_t*>(tab + p.Cin + cb * BK + 4 * j);
        }
      };
      auto xform = [&](uint4 raw, bool ok) -> uint4 {
        Pack16 in, out;
        in.u = raw;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = (float)in.e[e] * sc[e >> 2][e & 3] + sh[e >> 2][e & 3];
          if constexpr (GN == 2) v = pfd_silu(v);
          out.e[e] = (half_t)v;
        }
        return ok ? out.u : make_uint4(0, 0, 0, 0);   // zero padding stays zero (it pads the NORMALISED image)
      };
      auto put = [&](int buf, int q, uint4 v) {
        lds_store16_opaque(smem + buf * PATCH_BYTES + q * 1024 + lane * 16, v);
      };
      auto fetch = [&](int cb, int pix) -> uint4 {   // the virtual channel concat [A | A2] is resolved per channel block
        const int ch = cb * BK;
        const int px = pix < 0 ? 0 : pix;             // padding rows load pixel 0 (always mapped) and are zeroed by xform:
        const half_t* src = ch < p.gn_c1 ? img + (long)px * p.lda + (ch + c8)    // no branch around a load
                                         : img2 + (long)px * p.lda2 + (ch - p.gn_c1 + c8);
        return *reinterpret_cast<const uint4*>(src);
      };
      uint4 ra = make_uint4(0, 0, 0, 0), rb = ra;
      if (nsteps > 0) {
        issue_w(0, 0, cbv(0));
        load_tab(cbv(0));
        {                                    // the whole first patch: every load first, then the transforms
          uint4 fa[9], fb[9];
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) {
            fa[tp] = fetch(cbv(0), pix_a[tp]);
            fb[tp] = lw < 2 ? fetch(cbv(0), pix_b[tp]) : make_uint4(0, 0, 0, 0);
          }
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) {
            if (6 * tp + lw < P_INSTR) put(0, 6 * tp + lw, xform(fa[tp], pix_a[tp] >= 0));
            if (lw < 2 && 6 * tp + 4 + lw < P_INSTR) put(0, 6 * tp + 4 + lw, xform(fb[tp], pix_b[tp] >= 0));
          }
        }
        if (1 < ncbs) {                      // request tap 0's pieces of the second patch + its table
          load_tab(cbv(1));
          ra = fetch(cbv(1), pix_a[0]);
          if (lw < 2) rb = fetch(cbv(1), pix_b[0]);
        }
      }
      int stage = 0;
      for (int ci = 0; ci < ncbs; ++ci) {
        const int cb = cbv(ci);
        const int pbuf = ci & 1;
        const bool more = ci + 1 < ncbs;
        const int cb1 = more ? cbv(ci + 1) : cb, cb2 = ci + 2 < ncbs ? cbv(ci + 2) : cb;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0): W pieces landed, pixels arrived, ds_writes done
          block_barrier();                      // (A)
          if (tap < 8) issue_w(stage ^ 1, tap + 1, cb);
          else if (more) issue_w(stage ^ 1, 0, cb1);
          if (more) {
            // request the next pieces BEFORE transforming the ones that arrived: their latency runs under the VALU work
            uint4 na = ra, nb = rb;
            if (tap < 8) {                        // the next tap's pieces (same patch)
              na = fetch(cb1, pix_a[tap + 1]);
              if (lw < 2) nb = fetch(cb1, pix_b[tap + 1]);
            } else if (ci + 2 < ncbs) {           // ... or tap 0 of the patch after
              na = fetch(cb2, pix_a[0]);
              if (lw < 2) nb = fetch(cb2, pix_b[0]);
            }
            const uint4 va = xform(ra, pix_a[tap] >= 0);
            uint4 vb = make_uint4(0, 0, 0, 0);
            if (lw < 2) vb = xform(rb, pix_b[tap] >= 0);
            if (tap == 8 && ci + 2 < ncbs) load_tab(cb2);   // after the last transform with the current table
            if (6 * tap + lw < P_INSTR) put(pbuf ^ 1, 6 * tap + lw, va);
            if (lw < 2 && 6 * tap + 4 + lw < P_INSTR) put(pbuf ^ 1, 6 * tap + 4 + lw, vb);
            ra = na;
            rb = nb;
          }
          stage ^= 1;
        }
      }
    } else if constexpr (NWS == 3) {
      // Two taps of weights in flight.  Patch pieces of the NEXT channel block: slots s = 2 tap + {0, 1} for tap <= 6,
      // piece q = 4 s + lw (56 slots for 50 pieces; a slot past the patch re-issues piece q - 8, which this loader
      // issued one tap earlier -- same source, same destination -- so every loader issues the SAME number of pieces
      // per tap and the counted waits below are wave-independent constants).
      // Wait before barrier (A) of step t: W(t) must have landed.  Issued after W(t): P(t-2), W(t+1), P(t-1), so
      // vmcnt may stay at np(t-2) + 5 + np(t-1) (np = 2 on taps 0..6 of a block that has a successor, else 0);
      // at tap 0 that is 5: everything older than W(t+1), i.e. the whole patch of this block, has landed too.
      int off_p[14], q_p[14];
#pragma unroll
      for (int sl = 0; sl < 14; ++sl) {
        int q = 4 * sl + lw;
        if (q >= P_INSTR) q -= 8;
        q_p[sl] = q;
        off_p[sl] = piece_off(q);
      }
      if (nsteps > 0) {
#pragma unroll
        for (int sl = 0; sl < 14; ++sl)
          if (4 * sl + lw < P_INSTR) issue_patch(0, cbv(0), q_p[sl], off_p[sl]);
        issue_w(0, 0, cbv(0));
        issue_w(1, 1, cbv(0));
      }
      for (int ci = 0; ci < ncbs; ++ci) {
        const int cb = cbv(ci);
        const int pbuf = ci & 1;
        const bool more = ci + 1 < ncbs;
        const int cb1 = more ? cbv(ci + 1) : cb;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          if (more) {
            if (tap == 0) __builtin_amdgcn_s_waitcnt(0x0F70 | 5);
            else if (tap == 1 || tap == 8) __builtin_amdgcn_s_waitcnt(0x0F70 | 7);
            else __builtin_amdgcn_s_waitcnt(0x0F70 | 9);
          } else {
            if (tap <= 7) __builtin_amdgcn_s_waitcnt(0x0F70 | 5);
            else __builtin_amdgcn_s_waitcnt(0x0F70);
          }
          block_barrier();                      // (A) W(t) complete; every consumer is done with step t - 1
          if (tap + 2 < 9) issue_w((tap + 2) % 3, tap + 2, cb);
          else if (more) issue_w((tap + 2) % 3, tap + 2 - 9, cb1);
          if (more && tap <= 6) {
            issue_patch(pbuf ^ 1, cb1, q_p[2 * tap], off_p[2 * tap]);
            issue_patch(pbuf ^ 1, cb1, q_p[2 * tap + 1], off_p[2 * tap + 1]);
          }
        }
      }
    } else {
      if (nsteps > 0) {
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {     // the whole first patch
          if (6 * tp + lw < P_INSTR) issue_patch(0, cbv(0), 6 * tp + lw, off_a[tp]);
          if (lw < 2 && 6 * tp + 4 + lw < P_INSTR) issue_patch(0, cbv(0), 6 * tp + 4 + lw, off_b[tp]);
        }
        issue_w(0, 0, cbv(0));
      }
      int stage = 0;
      for (int ci = 0; ci < ncbs; ++ci) {
        const int cb = cbv(ci);
        const int pbuf = ci & 1;
        const bool more = ci + 1 < ncbs;
        const int cb1 = more ? cbv(ci + 1) : cb;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's pieces for this tap (and patch) are in LDS
          block_barrier();                      // (A)
          if (tap < 8) issue_w(stage ^ 1, tap + 1, cb);
          else if (more) issue_w(stage ^ 1, 0, cb1);
          if (more) {
            if (6 * tap + lw < P_INSTR) issue_patch(pbuf ^ 1, cb1, 6 * tap + lw, off_a[tap]);
            if (lw < 2 && 6 * tap + 4 + lw < P_INSTR) issue_patch(pbuf ^ 1, cb1, 6 * tap + 4 + lw, off_b[tap]);
          }
          stage ^= 1;
        }
      }
    }
    block_barrier();                          // (B) consumers finished the last tap: LDS is free
    block_barrier();                          // (C) the staging image is written
  } else {
    // ================================ consumer waves ================================
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;
    int rbase[WMB];
#pragma unroll
    for (int i = 0; i < WMB; ++i) {
      const int ql = wm * 64 + i * 16;
      const int ty = ql / TW, tx = ql - ty * TW;
      rbase[i] = ty * PW + tx + l15;
    }
    const int sw = (l15 >> 1) & 7;
    const int boff0 = wn * 80 * ROWB + l15 * ROWB + (((0 + g) ^ sw) << 4);
    const int boff1 = wn * 80 * ROWB + l15 * ROWB + (((4 + g) ^ sw) << 4);
    float4_t acc[WMB][5];
#pragma unroll
    for (int i = 0; i < WMB; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    auto a_off = [&](int i, int toff, int ks) -> int {   // byte offset of A fragment i in the patch (rotation swizzle)
      const int r = rbase[i] + toff;
      return r * ROWB + ((((ks * 4 + g) + (r & ~1)) & 7) << 4);
    };
    {
      int stage = 0;
      for (int ci = 0; ci < ncbs; ++ci) {
        const char* patch = smem + (ci & 1) * PATCH_BYTES;
        int toff = 0;
#pragma nounroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma nounroll
          for (int kx = 0; kx < 3; ++kx) {
            block_barrier();                    // (A)
            const char* wt = smem + OFF_W + stage * WT_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              half8_t af[WMB], bf[5];
#pragma unroll
              for (int i = 0; i < WMB; ++i) af[i] = *reinterpret_cast<const half8_t*>(patch + a_off(i, toff, ks));
              const int bo = ks ? boff1 : boff0;
#pragma unroll
              for (int j = 0; j < 5; ++j) bf[j] = *reinterpret_cast<const half8_t*>(wt + bo + j * 16 * ROWB);
#pragma unroll
              for (int i = 0; i < WMB; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
            }
            if constexpr (NWS == 2) stage ^= 1;
            else if (++stage == NWS) stage = 0;
            toff += 1;
          }
          toff += PW - 3;
        }
      }
      block_barrier();                          // (B)
    }
    {
      const G160Params pe = reload_params();
      epilogue_stage<WMB, 5, true>(acc, pe, lane, m0, n0, wm, wn, split, smem);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    block_barrier();                          // (C)
  }
  const G160Params pe = reload_params();
  epilogue_store<256, 5, 768, false, true>(pe, m0, n0, smem, tid, tile_m * (256 / GN_SLAB));
}

// (Round 5: a form of this kernel that hands tiles over through per-wave progress words in LDS instead of one block-wide
//  barrier per tap -- forced variant 95, bit-identical, validated on hardware -- measured +0.2 % per batch and was removed,
//  profiles/r05_e2e_ab_candidates.log; the protocol and its failed two-counter draft are in the git history.)

// sum the split-K slabs and apply the epilogue (bias, row vector, activation, residual)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const G160Params p) {
  const int nv = p.N / 8;
  const long nvec = (long)p.M * nv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
    const int m = (int)(i / nv);
    const int n = (int)(i - (long)m * nv) * 8;
    // every load is unconditional (an absent operand reads the zero page) and the slabs go out four splits at a time:
    // `if (p.bias) load` / a load per loop iteration are one round trip each (see epilogue_stage)
    Pack16 bb, rv, rr;
    bb.u = *reinterpret_cast<const uint4*>(p.bias ? p.bias + n : g_zero_page);
    rv.u = *reinterpret_cast<const uint4*>(p.rowvec ? p.rowvec + (long)(m / p.rows_per_rv) * p.ldrv + n : g_zero_page);
    rr.u = *reinterpret_cast<const uint4*>(p.R ? p.R + res_row(p, m) * p.ldr + n : g_zero_page);
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int s0 = 0; s0 < p.splits; s0 += 4) {
      Pack16 a[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        a[u].u = *reinterpret_cast<const uint4*>(reinterpret_cast<const half_t*>(p.ws) +
                                                 ((long)min(s0 + u, p.splits - 1) * p.M + m) * p.N + n);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float w = s0 + u < p.splits ? 1.f : 0.f;   // the clamped duplicates add nothing
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)a[u].e[e] * w;
      }
    }
    if (p.ln_in) {   // LayerNorm folded into this GEMM: the affine map of epilogue_stage, on the reduced accumulator
      const float2* src = p.ln_in + (long)m * p.ln_P;
      float sum = 0.f, sq = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float2 t = src[min(j, p.ln_P - 1)];
        const float w = j < p.ln_P ? 1.f : 0.f;
        sum = fmaf(t.x, w, sum);
        sq = fmaf(t.y, w, sq);
      }
      const float inv_k = 1.0f / (float)p.K;
      const float mean = sum * inv_k;
      const float rstd = rsqrtf(fmaxf(fmaf(-mean, mean, sq * inv_k), 0.f) + p.ln_eps);
      const float4_t c0 = *reinterpret_cast<const float4_t*>(p.ln_cs + n);
      const float4_t c1 = *reinterpret_cast<const float4_t*>(p.ln_cs + n + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = fmaf(v[e], rstd, -rstd * mean * c0[e]);
        v[4 + e] = fmaf(v[4 + e], rstd, -rstd * mean * c1[e]);
      }
    }
    Pack16 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = v[e] + (float)bb.e[e];
      x += (float)rv.e[e];
      if (p.act == PFD_ACT_GELU) x = pfd_gelu(x);
      else if (p.act == PFD_ACT_RELU) x = fmaxf(x, 0.f);
      else if (p.act == PFD_ACT_SILU) x = pfd_silu(x);
      o.e[e] = (half_t)(x + (float)rr.e[e]);
    }
    *reinterpret_cast<uint4*>(p.C + (long)m * p.ldc + n) = o.u;
  }
}

// The same reduction for a launch whose output feeds a GroupNorm (PfdGemmDesc.gn_out): one block per 64-row slab x 160-column
// tile, the store-pass mapping of epilogue_store_gn (a wave = 3 rows x 20 chunks), statistics through gn_slab_reduce.  The
// 64-row slab is walked in six sweeps of 12 rows; the loads of THREE sweeps (row vector / residual / 4-8 slab slices) are
// requested before the first add (round 5: the one-sweep-at-a-time form was six dependent round trips, 14.7 us per launch for
// 2-10 MB at the 8^2 / 16^2 levels; same slab order per element, same order of the rows in the statistics = the same bits;
// adopted with the GroupNorm apply's grouped partial loads at -0.3 % per batch, profiles/r05_e2e_ab_candidates.log).
// Up to 8 splits x 3 sweeps x 32 bytes = 96 + 24 VGPRs of loads.
// (round 5: 16 waves per slab -- the whole 64 rows requested in one round trip -- measured +0.3 % per batch against these 4,
//  profiles/r05_small_kernel_threads_ab.log: unlike the single-launch GroupNorm kernels this grid is already 256-512 blocks)
constexpr int RGN_T = 256;
__global__ __launch_bounds__(RGN_T) void splitk_reduce_gn_kernel(const G160Params p) {
  constexpr int RW = RGN_T / 64, ROWS_SW = 3 * RW;      // rows of one sweep of the block
  __shared__ __attribute__((aligned(16))) float red[(RW + 1) * 320];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int tiles_n = p.N / 160;
  const int slab = blockIdx.x / tiles_n, tile_n = blockIdx.x - slab * tiles_n;
  const int n = tile_n * 160 + (lane % 20) * 8;
  const bool active = lane < 60;
  const int rsub = lane / 20;
  float cs[8], cq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) cs[e] = cq[e] = 0.f;
  Pack16 bb;
  bb.u = *reinterpret_cast<const uint4*>(p.bias ? p.bias + n : g_zero_page);
  constexpr int SWEEPS = (GN_SLAB + ROWS_SW - 1) / ROWS_SW, U = SWEEPS < 3 ? SWEEPS : 3;
  static_assert(SWEEPS % U == 0, "sweeps in groups of U");
  for (int it0 = 0; it0 < SWEEPS; it0 += U) {
    int m[U];
    bool ok[U];
    Pack16 rv[U], rr[U];
    float v[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row_u = w * 3 + rsub + (it0 + u) * ROWS_SW;
      m[u] = min(slab * GN_SLAB + min(row_u, GN_SLAB - 1), p.M - 1);
      ok[u] = active && row_u < GN_SLAB && slab * GN_SLAB + row_u < p.M;
      rv[u].u = *reinterpret_cast<const uint4*>(p.rowvec ? p.rowvec + (long)(m[u] / p.rows_per_rv) * p.ldrv + n : g_zero_page);
      rr[u].u = *reinterpret_cast<const uint4*>(p.R ? p.R + res_row(p, m[u]) * p.ldr + n : g_zero_page);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[u][e] = 0.f;
    }
    for (int s0 = 0; s0 < p.splits; s0 += 4) {   // same order of the slabs as splitk_reduce_kernel
      Pack16 a[U][4];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          a[u][k].u = *reinterpret_cast<const uint4*>(reinterpret_cast<const half_t*>(p.ws) +
                                                      ((long)min(s0 + k, p.splits - 1) * p.M + m[u]) * p.N + n);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float wgt = s0 + k < p.splits ? 1.f : 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[u][e] += (float)a[u][k].e[e] * wgt;
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {   // the rows in the order of the plain kernel's sweeps
      Pack16 o;
      const float mk = ok[u] ? 1.f : 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = v[u][e] + (float)bb.e[e];
        x += (float)rv[u].e[e];
        if (p.act == PFD_ACT_GELU) x = pfd_gelu(x);
        else if (p.act == PFD_ACT_RELU) x = fmaxf(x, 0.f);
        else if (p.act == PFD_ACT_SILU) x = pfd_silu(x);
        o.e[e] = (half_t)(x + (float)rr[u].e[e]);
        const float f = (float)o.e[e] * mk;
        cs[e] += f;
        cq[e] = fmaf(f, f, cq[e]);
      }
      if (ok[u]) *reinterpret_cast<uint4*>(p.C + (long)m[u] * p.ldc + n) = o.u;
    }
  }
  gn_slab_reduce<1, RGN_T>(cs, cq, tid, p.N / 32, slab, p.M / GN_SLAB, tiles_n, tile_n, p.gn_out, [&](int) { return red; });
}


// ------------------------------------------------------------------------------------------------
// Round 5 (ABI 9, PfdGemmDesc.gnf_*): split-K reduction + GroupNorm(32)(+SiLU) of the result in ONE launch.  At the 8^2 /
// 16^2 levels of the UNet every 3x3 convolution splits K, a reduction launch stores the f16 result and the single-launch
// GroupNorm (gn_small_kernel, norm.hip) reads it again: 20 such pairs per UNet pass, ~8 us + a launch boundary each.  Here
// a block owns one (sample, group) slab of the OUTPUT -- gnf_rows rows x N / 32 channels, the unit gn_small_kernel owns --
// and forms it from the fp32 slabs: epilogue as in splitk_reduce_kernel (same slab order, same operation order: the same
// f16 values), statistics and normalisation as in gn_small_kernel (same thread -> chunk mapping, same order of the adds,
// same expressions: the same bits).  The raw result is stored too unless skip_raw.  Loads are unconditional and issued a
// group of two chunks x four slabs at a time (8 x 16 bytes in flight per thread, 1024 threads), indices walk incrementally (no
// division per chunk), gamma / beta of the group sit in LDS before the statistics barrier.
struct GnFuse {
  const half_t* gamma;
  const half_t* beta;
  half_t* y;
  long ldy;
  float eps;
  int act, rows, skip_raw;
};
// 1024 threads per (sample, group) slab: the job is one round trip of 40-660 KB per block, i.e. bound by how many loads a
// block has in flight; with 256 threads it took as long as the plain reduction (2048 blocks) + gn_small_kernel together
// (16.8 us, profiles/r05_fused_gnorm_ab.log)
#ifndef PFD_GN_THREADS
#define PFD_GN_THREADS 1024   // (-DPFD_GN_THREADS=256: A/B builds; must match norm.hip for the bit identity)
#endif
constexpr int GNF_T = PFD_GN_THREADS;
constexpr int GNF_MAX = 8192 / GNF_T;      // chunks (4 halfs) per thread: rows * (N / 128) <= GNF_T * GNF_MAX
constexpr int GNF_U = 2;        // chunks whose slab slices are requested together: 2 x 4 slabs x 16 bytes = 32 VGPRs of loads (128-VGPR budget at 16 waves per CU)

__global__ __launch_bounds__(GNF_T) void splitk_reduce_gnorm_kernel(const G160Params p, const GnFuse f) {
  __shared__ float red[2 * GNF_T / 64];
  __shared__ float gam_s[256], bet_s[256];
  const int cpg = p.N / 32, cpr = cpg / 4;
  // XCD-aware order: the 32 group slabs of a sample are 160-byte column strips of the same rows, i.e. neighbours share
  // cache lines -- all of them on one XCD (one L2), as many samples per XCD as the grid has (round 5: the plain (g, b) grid
  // spread the groups of a sample over all eight L2s and every strip edge was fetched twice from the memory side)
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int g = lin & 31, b = lin >> 5, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int HW = f.rows, total = HW * cpr;
  const int n_g = g * cpg;
  const long m_b = (long)b * HW;
  const int dr = GNF_T / cpr, dc = GNF_T - dr * cpr;
  // one row vector per sample (host: rows_per_rv % rows == 0 or rows_per_rv >= M)
  const half_t* rvb = p.rowvec ? p.rowvec + (m_b / p.rows_per_rv) * p.ldrv + n_g : g_zero_page;
  const int rv_step = p.rowvec ? 4 : 0, b_step = p.bias ? 4 : 0;
  const half_t* bb0 = p.bias ? p.bias + n_g : g_zero_page;
  const long r_ld = p.R ? p.ldr : 0;
  const int r_step = p.R ? 4 : 0;
  const half_t* rb = p.R ? p.R + m_b * p.ldr + n_g : g_zero_page;
  uint2 v[GNF_MAX];
  int r = tid / cpr, ch = tid - r * cpr;
#pragma unroll
  for (int k0 = 0; k0 < GNF_MAX; k0 += GNF_U) {
    if (GNF_T * k0 >= total) {          // block-uniform: nothing of this group (or any later one) exists
#pragma unroll
      for (int u = 0; u < GNF_U; ++u) v[k0 + u] = make_uint2(0, 0);
      continue;
    }
    int ru[GNF_U], cu[GNF_U];
    bool on[GNF_U];
#pragma unroll
    for (int u = 0; u < GNF_U; ++u) {
      on[u] = tid + GNF_T * (k0 + u) < total;
      ru[u] = on[u] ? r : 0;          // slots past the slab read the slab's first chunk and are dropped below
      cu[u] = on[u] ? ch : 0;
      r += dr;
      ch += dc;
      if (ch >= cpr) {
        ch -= cpr;
        ++r;
      }
    }
    Pack8 bq[GNF_U], rq[GNF_U], xq[GNF_U];
#pragma unroll
    for (int u = 0; u < GNF_U; ++u) {
      bq[u].u = *reinterpret_cast<const uint2*>(bb0 + cu[u] * b_step);
      rq[u].u = *reinterpret_cast<const uint2*>(rvb + cu[u] * rv_step);
      xq[u].u = *reinterpret_cast<const uint2*>(rb + (long)ru[u] * r_ld + cu[u] * r_step);
    }
    float acc[GNF_U][4];
#pragma unroll
    for (int u = 0; u < GNF_U; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[u][e] = 0.f;
    for (int s0 = 0; s0 < p.splits; s0 += 4) {   // the slab order of splitk_reduce_kernel
      Pack8 a[GNF_U][4];
#pragma unroll
      for (int u = 0; u < GNF_U; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          a[u][q].u = *reinterpret_cast<const uint2*>(reinterpret_cast<const half_t*>(p.ws) +
                                                      ((long)min(s0 + q, p.splits - 1) * p.M + m_b + ru[u]) * p.N + n_g + cu[u] * 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float w = s0 + q < p.splits ? 1.f : 0.f;   // the clamped duplicates add nothing
#pragma unroll
        for (int u = 0; u < GNF_U; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[u][e] += (float)a[u][q].e[e] * w;
      }
    }
#pragma unroll
    for (int u = 0; u < GNF_U; ++u) {
      Pack8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = acc[u][e] + (float)bq[u].e[e];
        x += (float)rq[u].e[e];
        if (p.act == PFD_ACT_GELU) x = pfd_gelu(x);
        else if (p.act == PFD_ACT_RELU) x = fmaxf(x, 0.f);
        else if (p.act == PFD_ACT_SILU) x = pfd_silu(x);
        o.e[e] = (half_t)(x + (float)xq[u].e[e]);
      }
      if (on[u] && !f.skip_raw) *reinterpret_cast<uint2*>(p.C + (m_b + ru[u]) * p.ldc + n_g + cu[u] * 4) = o.u;
      v[k0 + u] = on[u] ? o.u : make_uint2(0, 0);
    }
  }
  if (tid < cpg) {
    gam_s[tid] = (float)f.gamma[n_g + tid];
    bet_s[tid] = (float)f.beta[n_g + tid];
  }
  float sm = 0.f, sq = 0.f;
#pragma unroll
  for (int k = 0; k < GNF_MAX; ++k) {   // gn_small_kernel's order: chunk k of this thread, element e
    Pack8 q;
    q.u = v[k];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = (float)q.e[e];
      sm += x;
      sq += x * x;
    }
  }
  sm = wave_sum(sm);
  sq = wave_sum(sq);
  constexpr int NWV = GNF_T / 64;
  if (lane == 0) {
    red[wave] = sm;
    red[NWV + wave] = sq;
  }
  __syncthreads();
  const float count = (float)HW * (float)cpg;
  float ts = red[0], tq = red[NWV];
#pragma unroll
  for (int w = 1; w < NWV; ++w) {   // gn_small_kernel's order
    ts += red[w];
    tq += red[NWV + w];
  }
  const float mean = ts / count;
  const float rstd = rsqrtf(fmaxf(tq / count - mean * mean, 0.f) + f.eps);
  half_t* yb = f.y + m_b * f.ldy + n_g;
  r = tid / cpr;
  ch = tid - r * cpr;
  asm volatile("" : "+v"(r), "+v"(ch));   // a second walk, not the index registers of the first one kept alive
#pragma unroll
  for (int k = 0; k < GNF_MAX; ++k) {
    if (tid + GNF_T * k < total) {
      Pack8 q, o;
      q.u = v[k];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float w = rstd * gam_s[ch * 4 + e];
        float t = (float)q.e[e] * w + (bet_s[ch * 4 + e] - mean * w);
        if (f.act == PFD_ACT_SILU) t = pfd_silu(t);
        o.e[e] = (half_t)t;
      }
      *reinterpret_cast<uint2*>(yb + (long)r * f.ldy + ch * 4) = o.u;
    }
    r += dr;
    ch += dc;
    if (ch >= cpr) {
      ch -= cpr;
      ++r;
    }
  }
}

// the fused GroupNorm request of the launch being dispatched on this thread (set by pfd_gemm160_try, consumed by
// launch_splitk_reduce): kept out of G160Params so that the kernel-argument block of every other kernel -- and with it
// their hardware-validated instruction streams -- stays what it was
thread_local GnFuse t_gnf = {nullptr, nullptr, nullptr, 0, 0.f, 0, 0, 0};

// launches the reduction of a split-K launch (plain, or the statistics-emitting form) and the LayerNorm fallback statistics
inline void launch_splitk_reduce(const G160Params& p, hipStream_t s) {
  // Round 6: the reduction is timed as its own bucket (it is HBM-bound glue, not part of the GEMM's MFMA time): the caller's
  // event pair is closed here and a new one covers the reduction launch(es); the caller's pfd_prof_end then closes this one.
  // Algorithmic bytes: the fp32 slabs once + the f16 result (+ residual, + the normalised copy of the fused GroupNorm form).
  if (pfd_prof_on()) {
    pfd_prof_end(s);
    const double mn = (double)p.M * p.N;
    pfd_prof_begin(21, 0.0, 2.0 * p.splits * mn + 2.0 * mn * (1 + (p.R ? 1 : 0) + (t_gnf.y ? 1 : 0)), s);
  }
  if (t_gnf.y) {   // (host: no gn_out / ln_out with it)
    const GnFuse f = t_gnf;
    hipLaunchKernelGGL(splitk_reduce_gnorm_kernel, dim3(32 * (p.M / f.rows)), dim3(GNF_T), 0, s, p, f);
    return;
  }
  if (p.gn_out) {
    hipLaunchKernelGGL(splitk_reduce_gn_kernel, dim3((p.M / GN_SLAB) * (p.N / 160)), dim3(RGN_T), 0, s, p);
  } else {
    const long nvec = (long)p.M * (p.N / 8);
    int g = (int)((nvec + 255) / 256);
    if (g > 2048) g = 2048;   // (1024 / 8192: no difference end to end, round 5)
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(g), dim3(256), 0, s, p);
  }
  // a split-K output that feeds a folded LayerNorm: its row statistics by the stand-alone kernel
  if (p.ln_out) pfd_ln_rowstats_launch(p.C, p.ldc, p.M, p.N, reinterpret_cast<float*>(p.ln_out), s, false);   // inside this launch's event pair
}

// History of the switches that used to live here (removed in round 5; their measurements are in profiles/ and DESIGN 3.5):
//  * rotated K walk (k_rotation; PFD_KROT): gains on single linears of the cold replay, nothing end to end (559.0 vs 559.2 /
//    560.8 ms per batch), and it makes a row's fp32 summation order depend on where in the batch the row sits -- the kernels
//    keep the parameter at 0;
//  * ping-pong consumer groups of the loader-wave kernels (template flag PP; PFD_PP, forced variants 49 / 97): no better than
//    lock-step consumers (profiles/r03_krot_pp_replay.log) -- never instantiated any more;
//  * PFD_PATCH_RING / PFD_WS_RING = 0 (two weight stages instead of three, -4.6 ms per batch for three), PFD_R3TILES = 0 (the
//    round-2 tile rules), PFD_RING = 0 (no deep operand rings): A/B switches of decisions that have stood for two rounds.

// 1 when streaming W once per XCD would cost more L2-miss traffic than streaming the activations once per XCD
inline int pick_nmajor(const G160Params& p) {
  const double a_bytes = p.ksize > 0 ? 2.0 * p.B * p.H * p.Wd * p.Cin : 2.0 * p.M * p.K;
  const double w_bytes = 2.0 * p.N * p.K;
  return (p.tiles_n >= 2 && p.tiles_m >= 2 && w_bytes > a_bytes) ? 1 : 0;
}

template <int WAVES_M, int WMB, int NBUF = 2, int NT = 5>
int launch160(G160Params& p, int bucket, hipStream_t s) {
  constexpr int BM = WAVES_M * WMB * 16;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = p.N / (32 * NT);
  p.nmajor = pick_nmajor(p);
  p.krot = 0;
  const int nk = p.K / BK;
  p.kt_per_split = (nk + p.splits - 1) / p.splits;
  p.splits = (nk + p.kt_per_split - 1) / p.kt_per_split;
  dim3 grid(p.tiles_m * p.tiles_n, 1, p.splits);
  const bool prof = pfd_prof_on();
  if (prof) {
    const double a_bytes = p.ksize > 0 ? 2.0 * p.B * p.H * p.Wd * p.Cin : 2.0 * p.M * p.K;
    const double n_out = p.act == PFD_ACT_GEGLU ? p.N / 2 : p.N;
    pfd_prof_begin(bucket, 2.0 * p.M * p.N * p.K, a_bytes + 2.0 * p.N * p.K + 2.0 * p.M * n_out * (p.R ? 2 : 1), s);
  }
  if (p.ksize > 0)
    hipLaunchKernelGGL((gemm160_kernel<WAVES_M, WMB, true, NBUF, NT>), grid, dim3(WAVES_M * 128), 0, s, p);
  else
    hipLaunchKernelGGL((gemm160_kernel<WAVES_M, WMB, false, NBUF, NT>), grid, dim3(WAVES_M * 128), 0, s, p);
  if (p.splits > 1) launch_splitk_reduce(p, s);
  if (prof) pfd_prof_end(s);
  return pfd_check_launch("pfd_gemm_f16(wide)");
}

template <int NT>
// mode: 0 = two stages, 2 = 3-stage ring
int launch160ws(G160Params& p, int bucket, hipStream_t s, int pp) {
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = p.N / (32 * NT);
  p.nmajor = pick_nmajor(p);
  p.krot = 0;
  const int nk = p.K / BK;
  p.kt_per_split = (nk + p.splits - 1) / p.splits;
  p.splits = (nk + p.kt_per_split - 1) / p.kt_per_split;
  dim3 grid(p.tiles_m * p.tiles_n, 1, p.splits);
  const bool prof = pfd_prof_on();
  if (prof) {
    const double a_bytes = p.ksize > 0 ? 2.0 * p.B * p.H * p.Wd * p.Cin : 2.0 * p.M * p.K;
    const double n_out = p.act == PFD_ACT_GEGLU ? p.N / 2 : p.N;
    pfd_prof_begin(bucket, 2.0 * p.M * p.N * p.K, a_bytes + 2.0 * p.N * p.K + 2.0 * p.M * n_out * (p.R ? 2 : 1), s);
  }
  if (p.ksize > 0) {
    if (pp == 2) hipLaunchKernelGGL((gemm160ws_kernel<true, NT, 3>), grid, dim3(768), 0, s, p);
    else hipLaunchKernelGGL((gemm160ws_kernel<true, NT>), grid, dim3(768), 0, s, p);
  } else {
    if (pp == 2) hipLaunchKernelGGL((gemm160ws_kernel<false, NT, 3>), grid, dim3(768), 0, s, p);
    else hipLaunchKernelGGL((gemm160ws_kernel<false, NT>), grid, dim3(768), 0, s, p);
  }
  if (p.splits > 1) launch_splitk_reduce(p, s);
  if (prof) pfd_prof_end(s);
  return pfd_check_launch("pfd_gemm_f16(wave-specialised)");
}

// ws: 0 = 8-wave kernel, 1 = + 4 loader waves (two weight stages), 3 = loader waves + 3-stage weight ring (the default)
int launch_patch(G160Params& p, hipStream_t s, int ws) {
  p.tiles_m = p.M / 256;
  p.tiles_n = p.N / BN;
  p.nmajor = pick_nmajor(p);
  p.krot = 0;
  const int ncb = p.Cin / BK;
  p.kt_per_split = (ncb + p.splits - 1) / p.splits;   // channel blocks per split
  p.splits = (ncb + p.kt_per_split - 1) / p.kt_per_split;
  dim3 grid(p.tiles_m * p.tiles_n, 1, p.splits);
  const bool prof = pfd_prof_on();
  if (prof)
    pfd_prof_begin(19, 2.0 * p.M * p.N * p.K,
                   2.0 * p.B * p.H * p.Wd * p.Cin + 2.0 * p.N * p.K + 2.0 * p.M * p.N * (p.R ? 2 : 1), s);
  if (p.gn_table && p.gn_act == PFD_ACT_SILU) hipLaunchKernelGGL((conv3x3_patch_ws_kernel<2>), grid, dim3(768), 0, s, p);
  else if (p.gn_table) hipLaunchKernelGGL((conv3x3_patch_ws_kernel<1>), grid, dim3(768), 0, s, p);
  else if (ws == 3) hipLaunchKernelGGL((conv3x3_patch_ws_kernel<0, 3>), grid, dim3(768), 0, s, p);
  else if (ws) hipLaunchKernelGGL((conv3x3_patch_ws_kernel<0>), grid, dim3(768), 0, s, p);
  else hipLaunchKernelGGL(conv3x3_patch_kernel, grid, dim3(512), 0, s, p);
  if (p.splits > 1) launch_splitk_reduce(p, s);
  if (prof) pfd_prof_end(s);
  return pfd_check_launch("pfd_gemm_f16(conv3x3 patch)");
}

}  // namespace

// Called by pfd_gemm_f16_ex (gemm_conv.hip).  Returns 1 if the problem is not for this path.
// variant: 0 = heuristic, 44 / 24 / 22 force <WAVES_M,WMB>; splits: 0 = heuristic.
// N % 160 == 0 runs 160-wide tiles (every UNet / ControlNet width); otherwise N % 128 == 0 runs the same
// kernel with 128-wide tiles (wave tile x 64: the VAE's 128/256/512 channels, Swin / SeeCoder widths).
int pfd_gemm160_try(const PfdGemmDesc* d, int variant, int splits, hipStream_t s) {
  const int bn = d->N % 160 == 0 ? 160 : 128;
  if (d->N % bn || d->bias_per_row || d->K % BK) return 1;
  if (bn == 128 && d->act == PFD_ACT_GEGLU) return 1;   // GEGLU packing is defined per serving kernel (pfd_hip.h)
  if (d->ksize > 0 && (d->Cin % BK)) return 1;
  if (d->act == PFD_ACT_GEGLU && (d->rowvec || d->R)) return 1;
  if ((d->ldc & 7) || (reinterpret_cast<uintptr_t>(d->C) & 15)) return 1;
  if (d->bias && (reinterpret_cast<uintptr_t>(d->bias) & 15)) return 1;
  if (d->R && ((d->ldr & 7) || (reinterpret_cast<uintptr_t>(d->R) & 15))) return 1;
  if (d->rowvec && ((d->ldrv & 7) || (reinterpret_cast<uintptr_t>(d->rowvec) & 15))) return 1;
  if (d->Ct) {  // transposed tail: validated by the caller to be a plain, epilogue-free GEMM
    if (d->ksize > 0 || d->n_split <= 0 || d->n_split >= d->N || d->n_split % bn || (d->ldct & 7) ||
        (reinterpret_cast<uintptr_t>(d->Ct) & 15))
      return 1;
    splits = 1;
  }
  G160Params p;
  p.Ct = (half_t*)d->Ct; p.ldct = d->ldct; p.n_split = d->n_split;
  p.A = (const half_t*)d->A; p.W = (const half_t*)d->W; p.bias = (const half_t*)d->bias;
  p.rowvec = (const half_t*)d->rowvec; p.R = (const half_t*)d->R; p.C = (half_t*)d->C;
  p.ws = (float*)d->ws;
  p.lda = d->lda; p.ldw = d->ldw; p.ldr = d->ldr; p.ldc = d->ldc; p.ldrv = d->ldrv;
  p.w_tu = d->w_tiled ? bn : 0;
  p.w_kstep = d->w_tiled ? (long)bn * BK : BK;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.rows_per_rv = d->rows_per_rv > 0 ? d->rows_per_rv : 1;
  p.act = d->act;
  p.ksize = d->ksize; p.stride = d->stride; p.pad = d->pad; p.ups = d->ups;
  p.B = d->B; p.H = d->H; p.Wd = d->Wd; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo;
  p.tiles_m = p.tiles_n = 0;
  p.kt_per_split = 0;
  p.pt_w = p.pt_sh = 0;
  p.gn_table = (const float*)d->gn_table; p.A2 = (const half_t*)d->A2; p.lda2 = d->lda2;
  p.gn_c1 = d->gn_c1; p.gn_act = d->gn_act;
  p.ln_in = (const float2*)d->ln_stats; p.ln_cs = (const float*)d->ln_colsum; p.ln_P = d->ln_parts; p.ln_eps = d->ln_eps;
  p.ln_out = (float2*)d->ln_out;
  // GroupNorm statistics of the output (ABI 8): 160-wide tiles, whole 64-row slabs, groups that do not straddle a tile
  p.gn_out = (float2*)d->gn_out;
  if (p.gn_out) {
    if (bn != 160 || (d->N % 32) || d->N / 32 < 8 || (160 % (d->N / 32)) || (d->M % GN_SLAB) || d->act == PFD_ACT_GEGLU || d->Ct ||
        d->ln_out || d->bias_per_row || (reinterpret_cast<uintptr_t>(p.gn_out) & 7))
      return 1;
    if (d->ksize > 0 && ((long)d->Ho * d->Wo) % GN_SLAB) return 1;   // a slab must not straddle two samples
  }
  // GroupNorm fused into the split-K reduction (ABI 9): validated here, armed once the split count is known (gnf_arm below)
  t_gnf.y = nullptr;
  const bool gnf = d->gnf_y != nullptr;
  if (gnf) {
    const int cpg = d->N / 32;
    if (bn != 160 || (d->N % 32) || (cpg % 4) || cpg < 20 || cpg > 256 || d->gnf_rows <= 0 || (d->M % d->gnf_rows) ||
        (long)d->gnf_rows * (cpg / 4) > (long)GNF_T * GNF_MAX || (long)(d->M / d->gnf_rows) * 32 < 128 || !d->gnf_gamma || !d->gnf_beta ||
        (d->gnf_ldy & 3) || (reinterpret_cast<uintptr_t>(d->gnf_y) & 7) || d->act == PFD_ACT_GEGLU || d->Ct || d->ln_stats ||
        d->ln_out || d->gn_out || d->bias_per_row || !d->ws ||
        (d->gnf_act != PFD_ACT_NONE && d->gnf_act != PFD_ACT_SILU))
      return 1;
    const int rpr = d->rows_per_rv > 0 ? d->rows_per_rv : 1;
    if (d->rowvec && (rpr % d->gnf_rows) && rpr < d->M) return 1;   // one row vector per sample
  }
  auto gnf_arm = [&](int nsplits) -> bool {   // false: the problem is not split -> not served (nothing launched yet)
    if (!gnf) return true;
    if (nsplits <= 1) return false;
    t_gnf = GnFuse{(const half_t*)d->gnf_gamma, (const half_t*)d->gnf_beta, (half_t*)d->gnf_y, (long)d->gnf_ldy, d->gnf_eps,
                   d->gnf_act, d->gnf_rows, d->gnf_skip_raw};
    return true;
  };
  struct GnfDisarm { ~GnfDisarm() { t_gnf.y = nullptr; } } gnf_disarm;
  // two-source contraction / zero rows (ABI 8): the 8-wave / 4-wave linear kernels only
  // residual stored once for a doubled batch (ABI 9): one wrap at most, whole rows; not with the fused GroupNorm reduction
  p.r_wrap = 0x7fffffff;
  if (d->res_rows > 0 && d->res_rows != d->M) {
    if (!d->R || d->res_rows >= d->M || 2L * d->res_rows < d->M || d->gnf_y) return 1;
    p.r_wrap = d->res_rows;
  }
  p.k_split = d->K; p.zero_rows = 0;
  if (d->k_split > 0 || d->zero_rows > 0) {
    if (d->ksize > 0 || d->gn_table || d->bias_per_row || d->Ct) return 1;
    if (d->zero_rows < 0 || d->zero_rows >= d->M || (d->zero_rows > 0 && d->ln_stats)) return 1;
    if (variant == 47 || variant == 48 || variant == 84) return 1;
    if (d->k_split > 0) {
      if (d->k_split >= d->K || (d->k_split % BK) || !d->A2 || (d->lda2 & 7) || (reinterpret_cast<uintptr_t>(d->A2) & 15)) return 1;
      p.k_split = d->k_split;
    }
    p.zero_rows = d->zero_rows;
  }
  if (p.ksize == 0 && p.k_split == d->K) { p.A2 = p.A; p.lda2 = p.lda; }
  if (p.ln_in) {   // LayerNorm fold: plain linear, statistics over K = ln_parts slices of 160 columns
    if (d->ksize > 0 || !p.ln_cs || p.ln_P < 1 || p.ln_P > 8 || p.ln_P * 160 != d->K) return 1;
    if ((reinterpret_cast<uintptr_t>(p.ln_in) & 7) || (reinterpret_cast<uintptr_t>(p.ln_cs) & 15)) return 1;
    if (variant == 47 || variant == 48) return 1;   // the loader-wave kernels serve convolutions
  }
  if (p.ln_out) {  // statistics of the output rows for the consumer's fold
    if (bn != 160 || d->ksize > 0 || d->act == PFD_ACT_GEGLU || d->Ct || (reinterpret_cast<uintptr_t>(p.ln_out) & 7)) return 1;
    if (variant == 47 || variant == 48) return 1;
  }
  if (p.gn_table) {   // GroupNorm prologue: patch kernel or nothing (validated here, PFD_ESHAPE by the caller otherwise)
    if (bn != 160 || (variant != 0 && variant != 98)) return 1;
    variant = 98;   // loader waves with two weight stages (the form the prologue instances are built on)
    if (p.gn_c1 <= 0 || p.gn_c1 > p.Cin || (p.gn_c1 % BK) || (p.gn_c1 < p.Cin && !p.A2)) return 1;
    if ((p.lda2 & 7) || (reinterpret_cast<uintptr_t>(p.A2) & 15) || (reinterpret_cast<uintptr_t>(p.gn_table) & 15))
      return 1;
    if (p.gn_act != PFD_ACT_NONE && p.gn_act != PFD_ACT_SILU) return 1;
    if (p.gn_c1 == p.Cin) { p.A2 = p.A; p.lda2 = p.lda; }
  }
  const int tn = p.N / bn;
  auto tiles = [&](int bm) { return (long)((p.M + bm - 1) / bm) * tn; };
  // 3x3 / s1 / p1 convolution on a 16-, 32- or 64-wide image: the patch kernel (variant 0 or 99)
  // patch tile: whole image rows on 16- / 32- / 64-wide images, TH x 32 or TH x 16 pixel tiles on wider ones whose width
  // they divide (96-, 48-wide latents of the 768^2 / 512 x 768 configurations; round 3)
  int pt_w = 0;
  if (p.ksize == 3 && p.Wd != 16 && p.Wd != 32 && p.Wd != 64) {
    if (p.Wd % 32 == 0 && p.H % 8 == 0) pt_w = 32;
    else if (p.Wd % 16 == 0 && p.H % 16 == 0) pt_w = 16;
  }
  const bool patch_w = p.Wd == 16 || p.Wd == 32 || p.Wd == 64 || pt_w != 0;
  if (bn == 160 && (variant == 0 || variant == 99 || variant == 98 || variant == 96) && p.ksize == 3 && p.stride == 1 && p.pad == 1 && !p.ups &&
      patch_w && p.Ho == p.H && p.Wo == p.Wd && (pt_w != 0 || p.H % (256 / p.Wd) == 0) &&
      p.M % 256 == 0 && ((long)p.H * p.Wd) % 256 == 0 && p.act != PFD_ACT_GEGLU) {
    p.pt_w = pt_w;
    p.pt_sh = pt_w == 32 ? 5 : 4;
    // (The 8-wave 128-row ring beats the patch kernel on its smallest problems -- 16384 x 320 x 2880: 50 -> 43 us,
    //  profiles/r03_tile_variants_replay.log -- but the GroupNorm-prologue form lives in the patch kernel only and the two
    //  must stay bit-identical, for 1 ms per batch: not taken.)
    const int ncb = p.Cin / BK;
    if (splits == 0) {
      splits = 1;
      const long tl = tiles(256);
      // (round 5, end to end: no split below 200 tiles -> below 100: +2.3 % per batch; aiming at 512 blocks instead of 256: +3.7 %)
      if (d->ws && tl < 200) {
        splits = (int)((256 + tl - 1) / tl);
        if (splits > 8) splits = 8;
        while (splits > 1 && ncb / splits < 2) --splits;
        while (splits > 1 && (size_t)splits * p.M * p.N * 4 > d->ws_bytes) --splits;
      }
    }
    if (splits > 1 && (!d->ws || (size_t)splits * p.M * p.N * 4 > d->ws_bytes)) splits = 1;
    p.splits = splits;
    {   // (launch_patch turns the request into channel blocks per split: the count it will really launch)
      const int kps = (ncb + splits - 1) / splits;
      if (!gnf_arm((ncb + kps - 1) / kps)) return 1;
    }
    // default: the wave-specialised form (4 loader waves; +4 ... 13 % on every patch-eligible conv of the UNet, most on the
    // long-K ones, profiles/r02_patch_ws_ab.log) with the 3-stage weight ring (two taps of weights in flight, counted vmcnt;
    // round 4, -4.6 ms per batch); 99 forces the 8-wave form, 98 the loader-wave form with two weight stages (the
    // GroupNorm-prologue instances are built on it)
    const int ws = variant == 99 ? 0 : variant == 98 ? 1 : 3;   // 96 forces what is the default anyway
    return launch_patch(p, s, ws) < 0 ? PFD_ELAUNCH : 0;
  }
  if (variant == 99 || variant == 98 || variant == 96 || p.gn_table) return 1;
  const bool auto_variant = variant == 0;
  const int nk_all = p.K / BK;
  if (auto_variant) {
    // measured on MI355X: every GEMM/conv of a UNet pass replayed with COLD weights (the 1.7 GB of other
    // layers evict every W between two uses) under each forced variant (profiles/r01_gemm_replay_variants.log),
    // then A/B-ed end to end in one job: 256x160 when its tiles fill the chip; otherwise 128x160 (two
    // co-resident 4-wave blocks, + split-K when there are < 256 of them) -- except problems with a short K
    // loop and too few 128-row tiles to fill the chip twice, which take 64x160 (8192x640x640: 24 -> 20 us).
    // (In the cold replay 128x160 also beat 256x160 on the big problems -- GEGLU 99 -> 93 us -- but inside
    //  the real loop it lost: 664 vs 661 ms per batch.)
    const long t128 = tiles(128);
    const bool few = nk_all >= 16 ? t128 < 256 : t128 < 384;
    variant = tiles(256) >= 200 ? 44 : (nk_all <= 24 && few) ? 22 : 24;
    // round 2 (coalesced epilogue, cold replay of the sampler's launch list, profiles/r02_gemm_replay_variants.log):
    // short-K linears on >= 8192 rows are epilogue / HBM bound: two co-resident 128-row blocks overlap one
    // block's store pass with the other's K loop (GEGLU 32768x2560x320: 87 -> 79 us, qkv 8192x1920x640: 42 -> 35);
    // mid-K problems with <= 256 tiles of 128 rows take 64-row tiles (8192x640x2560: 54 -> 46 us)
        // (nk_all <= 40 since round 5, end to end -0.25 %; <= 20 came from the cold replay)
    if (p.ksize == 0 && variant == 44 && p.M >= 8192 && nk_all <= 40) variant = 24;
    // 128-row tiles that fill the chip at most once run the 3-stage ring (one 110 KB block per CU is no loss
    // there): 8192x640x2560 46 -> 38 us, 8192x640x1280 25 -> 24 (profiles/r02_ring_replay.log)
    if (bn == 160 && p.ksize == 0 && variant == 24 && t128 <= 256 && nk_all >= 16) variant = 25;
    if (p.ksize == 0 && variant == 24 && nk_all <= 40 && t128 <= 256 && tiles(64) >= 384) variant = 22;
    // implicit-GEMM convolutions (stride 2, fused upsample, widths the patch kernel does not take) run long K loops
    // of 53 KB stages: with the DMA pieces on four dedicated loader waves they gain 5-17 % (32768 x 640 x 5760
    // upsample conv: 225 -> 192 us = 1260 TF); the short-K linears do not (profiles/r02_wave_specialised_ab.log)
    if (variant == 44 && p.ksize > 0) variant = 47;
    // round 3 (cold replay under every forced variant, profiles/r03_tile_variants_replay.log): the 128-row tile on EIGHT
    // waves (4 x 2 wave layout, wave tile 32 x 80; variants 82 / 83) instead of four beats the 4-wave form wherever
    // that was chosen (qkv 8192 x 1920 x 640: 32.5 -> 27.3 us, 32768 x 960 x 320: 34.3 -> 31.8, ff-out 8192 x 640 x 2560
    // 39.4 -> 36.1, upsample conv 2048 x 1280 x 11520: 94 -> 73); GEGLU projections with >= 10 K tiles take the 256 x 320
    // tile (84: 8192 x 5120 x 640 69 -> 60 us, 2048 x 10240 x 1280 63 -> 51); long-K problems on <= 2048 rows and the
    // stride-2 convolutions take the 64-row tile on eight waves with the 4-stage ring and NO split-K (43: 2048 x 1280 x
    // 5120 55 -> 48 us, 2048 x 640 x 5760 / s2 52 -> 37, 8192 x 320 x 2880 / s2 38.5 -> 30.7).
    if (bn == 160) {
      // (nk_all >= 5 since round 5 -- the 64^2 GEGLU projection, K = 320, too: -0.3 % end to end; >= 10 came from the cold replay)
            if (p.act == PFD_ACT_GEGLU && p.ksize == 0 && p.N % 320 == 0 && p.M >= 2048 && nk_all >= 5) variant = 84;
      else if (p.ksize == 0 && (variant == 24 || variant == 25) && p.M <= 2048 && nk_all >= 32) variant = 43;   // (>= 64 until round 5; >= 32: -0.5 % end to end)
      else if (p.ksize > 0 && p.stride == 2 && p.M <= 8192 && (variant == 24 || variant == 25 || variant == 22)) variant = 43;
      else if (variant == 24) variant = 82;
      else if (variant == 25) variant = 83;
    }
  }
  const int bm = (variant == 44 || variant == 48 || variant == 47 || variant == 84) ? 256
                 : (variant == 24 || variant == 25 || variant == 82 || variant == 83) ? 128 : 64;
  if (splits == 0) {
    splits = 1;
    const long tl = tiles(bm);
    const int nk = nk_all;
    if (p.act != PFD_ACT_GEGLU && d->ws && (variant == 44 || variant == 48 || variant == 47) && tl < 200 && nk >= 48 &&
        (size_t)2 * p.M * p.N * 4 <= d->ws_bytes) {
      splits = 2;  // 128 tiles of 256x160: two K halves fill the chip (758 vs 579 TF at 640->640 @32^2)
    } else if (p.act != PFD_ACT_GEGLU && d->ws && (variant == 24 || variant == 25 || variant == 82 || variant == 83) && tl < 256) {
      splits = (int)((512 + tl - 1) / tl);
      if (splits > 8) splits = 8;
      while (splits > 1 && nk / splits < 16) --splits;  // the slab round trip must stay small vs the K loop
      while (splits > 1 && (size_t)splits * p.M * p.N * 4 > d->ws_bytes) --splits;
    } else if (p.act != PFD_ACT_GEGLU && d->ws && (variant == 22 || variant == 23 || variant == 41 || variant == 43) && tl <= 128 && nk >= 16) {
      splits = (int)(256 / tl);   // M <= 1024 rows (8^2 level, cond-half projections): 25 -> 21 us
      if (splits > 4) splits = 4;
      while (splits > 1 && (size_t)splits * p.M * p.N * 4 > d->ws_bytes) --splits;
    }
  }
  if (splits > 1 && (!d->ws || (size_t)splits * p.M * p.N * 4 > d->ws_bytes || p.act == PFD_ACT_GEGLU)) splits = 1;
  p.splits = splits;
  {   // (the launchers turn the request into K tiles per split: the count they will really launch)
    const int kps = (nk_all + splits - 1) / splits;
    if (!gnf_arm((nk_all + kps - 1) / kps)) return 1;
  }
  if (auto_variant && bn == 160) {
    // Problems whose blocks fill the chip once (the 16^2 / 8^2 levels: <= 256 tiles, or split-K slices of them) are
    // bound by the DMA round trip per K tile, not by MFMA or LDS capacity: they take the deep operand rings (3 K tiles
    // in flight on 64-row tiles, 2 on 128-row tiles; counted vmcnt + raw barrier).  Cold replay of the sampler's launch
    // list: 2048x1280x1280 23 -> 18 us, 4096x640x640 15 -> 12.5, 8^2 convs 512x1280x11520 38 -> 34, 512x1280x23040
    // 60 -> 51, GEGLU 512x10240x1280 30 -> 24 (profiles/r02_ring_replay.log).
    const int nk_split = nk_all / splits;
    if (variant == 22 && tiles(64) * splits <= 256 && nk_split >= 8) variant = 23;
    if (variant == 24 && tiles(128) < 256 && nk_split >= 6) variant = 25;
    if (variant == 82 && (p.ksize == 0 || (p.stride == 1 && !p.ups)) && tiles(128) <= 256 && nk_split >= 6) variant = 83;
  }
  const int conv = p.ksize > 0 ? 1 : 0;
  // round 5: the 64-row tiles run on EIGHT waves (4 x 1 wave layout, wave tile 16 x 160: variants 41 / 43) wherever the rules
  // above picked the four-wave forms 22 / 23 -- twice the waves issuing LDS-DMA pieces per CU on launches that are chains of
  // round trips.  Same tile, same split counts, same K order.  End to end -0.2 % (22 -> 41) and -0.3 % (23 -> 43), alternating on
  // one box (profiles/r05_e2e_ab_candidates.log); round 3 had adopted the eight-wave forms only where the COLD REPLAY showed a gain.
  if (auto_variant && bn == 160) variant = variant == 22 ? 41 : variant == 23 ? 43 : variant;
  if (variant == 48 || variant == 47) {   // 8 MFMA waves + 4 loader waves (48: two operand stages, 47: 3-stage ring)
    const int mode = variant == 47 ? 2 : 0;
    if (bn == 128) return launch160ws<4>(p, 12 + 4 * conv, s, mode) < 0 ? PFD_ELAUNCH : 0;
    return launch160ws<5>(p, 12 + 4 * conv, s, mode) < 0 ? PFD_ELAUNCH : 0;
  }
  if (bn == 128) {
    switch (variant) {
      case 44: return launch160<4, 4, 2, 4>(p, 12 + 4 * conv, s) < 0 ? PFD_ELAUNCH : 0;
      case 24: return launch160<2, 4, 2, 4>(p, 13 + 4 * conv, s) < 0 ? PFD_ELAUNCH : 0;
      case 22: return launch160<2, 2, 2, 4>(p, 14 + 4 * conv, s) < 0 ? PFD_ELAUNCH : 0;
      default: return variant == 99 ? 1 : PFD_EINVAL;
    }
  }
  if (variant == 84) {   // 256 x 320 tile (wave tile 64 x 160): GEGLU projections whose tiles fill the chip
    if (p.act != PFD_ACT_GEGLU || conv || p.N % 320 || p.splits != 1) return PFD_EINVAL;
    return launch160<4, 4, 2, 10>(p, 12, s) < 0 ? PFD_ELAUNCH : 0;
  }
  switch (variant) {
    case 44: return launch160<4, 4>(p, 12 + 4 * conv, s) < 0 ? PFD_ELAUNCH : 0;
    case 24: return launch160<2, 4>(p, 13 + 4 * conv, s) < 0 ? PFD_ELAUNCH : 0;
    case 22: return launch160<2, 2>(p, 14 + 4 * conv, s) < 0 ? PFD_ELAUNCH : 0;
    // deep operand rings (counted vmcnt): K tiles in flight ahead of the MFMAs = 3 (64-row tile) / 2 (128-row tile)
    case 23: return launch160<2, 2, 4>(p, 14 + 4 * conv, s) < 0 ? PFD_ELAUNCH : 0;
    case 25: return launch160<2, 4, 3>(p, 13 + 4 * conv, s) < 0 ? PFD_ELAUNCH : 0;
    // (round 5: the same tiles on a 5-stage ring -- 4 K tiles in flight -- measured 0.0 % end to end and were removed,
    //  profiles/r05_e2e_ab_candidates.log)
    // round 3 experiments: the same tiles on 8 waves (4 x 2 wave layout, wave tile 16 x 80 / 32 x 80): twice the waves
    // issuing LDS-DMA pieces per CU and two waves per SIMD on the problems whose one 4-wave block per CU is bound by the
    // piece issue rate (64-row tiles: 41 two stages, 43 four-stage ring; 128-row tiles: 82 two stages, 83 three)
    case 41: return launch160<4, 1, 2>(p, 14 + 4 * conv, s) < 0 ? PFD_ELAUNCH : 0;
    case 43: return launch160<4, 1, 4>(p, 14 + 4 * conv, s) < 0 ? PFD_ELAUNCH : 0;
    case 82: return launch160<4, 2, 2>(p, 13 + 4 * conv, s) < 0 ? PFD_ELAUNCH : 0;
    case 83: return launch160<4, 2, 3>(p, 13 + 4 * conv, s) < 0 ? PFD_ELAUNCH : 0;
    default: return PFD_EINVAL;
  }
}
