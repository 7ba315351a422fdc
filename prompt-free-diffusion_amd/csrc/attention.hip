// pfd_attention_f16: fused QK^T -> online softmax -> PV on v_mfma_f32_32x32x16_f16.
//
// The kernel works on the TRANSPOSED problem so that the softmax reduction axis is
// (almost) lane-local and P never leaves registers:
//     S^T[kv, q] = sum_d K[kv, d] Q[q, d]        A = K tile (LDS), B = Q (registers)
//     O^T[d,  q] = sum_kv V^T[d, kv] P^T[kv, q]   A = V^T tile (LDS), B = P^T (registers)
// A 32x32 S^T accumulator has col = q = lane&31 and rows kv = (r&3) + 8(r>>2) + 4(lane>>5):
// one q column is split over lanes l and l^32, so the row max needs exactly one cross-lane
// exchange.  Registers 8b..8b+7 of that accumulator are, for 16-kv block b, the 8 values an MFMA
// B operand wants -- in the k order pi(hi, j) = (j&3) + 8(j>>2) + 4hi.  MFMA only needs A and B
// to agree on which k a (lane-group, j) slot means, so the V^T A operand is read in the same pi
// order: two 8-byte LDS reads per fragment, no permutation of P.
// V arrives already transposed from its producer GEMM (see pfd_hip.h), K row-major.
//
// Block = 4 waves x 32 query rows; KV tile = 64 keys, two LDS stages: the global loads of tile
// t+1 are issued (to registers) before the MFMAs of tile t and written to the other stage after
// them -- one barrier per tile, HBM/L2 latency hidden under compute.
//
// The path is VALU-bound at head dim 40 (160 MFMA flops per score vs ~4 VALU ops), so the softmax
// is kept to max / fma / v_exp / cvt per score:
//   * scores stay raw; p = exp2(s*c - m*c) with c = scale*log2(e) is one fma + one v_exp_f32;
//   * the row sum costs nothing when the V^T tile has padding rows (D = 40, 80): row D of the LDS
//     tile is set to 1.0, so O^T[D, q] accumulates sum_kv P -- with exactly the fp16-rounded P that
//     multiplies V, and it is rescaled with the rest of the accumulator;
//   * the accumulator is rescaled only when some row of the wave actually raised its running max
//     (exact: alpha == 1 otherwise).
// Per tile per wave: 2*DQK/16 + 4*DV/32 MFMAs (DQK = D rounded to 16, DV = D rounded to 32).
#include <stdlib.h>

#include "pfd_common.h"
#include "attention_params.h"

namespace {

// Occupancy: D = 40 needs 136 VGPRs when left alone (3 waves per SIMD); asked for 4 waves per SIMD the
// allocation fits 128 without scratch and self-attention at N = 4096 gains 4.6 % (452 -> 473 TF), the 148-key
// cross-attention 16 % (profiles/r02_attention_ablation.log).  A 128-key tile (half the barriers, 62 KB LDS,
// 184 VGPRs) measured 2 % slower.  Larger head dims keep their natural allocation.
// (The round-2 kernel this describes -- attention_kernel<D, 64>, PFD_ATTN=0 -- and the intermediate stages PFD_ATTN=1..5 were
//  removed in round 5; git history and profiles/r03_attention_modes.log hold their measurements.)
// ------------------------------------------------------------------------------------------------
// The attention kernel (round 3 on; same math, same LDS K image, same transposed formulation as the round-2 kernel):
//  * the ragged last KV tile is PEELED: the loop over full tiles has no key masking, no clamped row indices and no
//    64-bit address arithmetic (per-thread source pointers advance by a constant per tile).  The ISA of the old loop
//    spent ~60 of its ~200 VALU instructions per tile on that, and the softmax path is VALU bound at d = 40.
//  * NWAVES = 8: 256 queries per block share one staged K / V^T tile (the global loads, ds_writes and the barrier of
//    a tile are paid once per 256 queries instead of per 128; the "no loads / staging" ablation of the old kernel
//    was -29 %).  Same 4 waves per SIMD (two 512-thread blocks per CU).
//  * PV16 (d = 40): O^T = V^T . P^T on v_mfma_f32_16x16x32_f16, so the head dim pads 40 -> 48 instead of 64 (12
//    MFMAs x 16 cycles per tile instead of 8 x 32).  The 32x32 S^T accumulator holds query l & 31 in lane l; a
//    16x16x32 B operand wants the SAME 16 queries in all four 16-lane rows.  One v_permlane16_swap per packed
//    register pair (16-key blocks bb = 0 / 1 of a 32-key half) does it: afterwards the first register holds queries
//    0-15 in every row (row kg = 2 hi + bb), the second queries 16-31.  MFMA only needs A and B to agree on which key a
//    (row, slot) pair means, so the V^T A operand is read in that order: keys 32 u + 16 bb + 4 hi + {0..3} and + 8.
//    The V^T image is 64 halfs per row without padding; 8-byte slot s of row d is stored at s ^ sigma(d),
//    sigma = d[1] | d[2] << 1 | d[3] << 3, which makes both ds_read_b64 of a fragment conflict free (rows of equal
//    parity share a bank half; {sigma} and {sigma ^ 4} partition its 16 slots).
// ------------------------------------------------------------------------------------------------
//  * FOLD (d = 40): the running maximum is subtracted BY THE QK^T MFMA.  D pads 40 -> 48 in the contraction, so slot 40
//    is free: K's LDS image carries 1.0 there, the (pre-scaled, Q' = scale log2(e) Q) query fragment carries -m, and the
//    accumulator comes out as s' - m -- the per-score fma in front of v_exp_f32 (31 of the ~105 VALU instructions of a
//    tile) disappears.  m is kept as an f16 value (it lives in an f16 operand slot); every use of it -- the fold, the
//    accumulator rescale -- sees the same rounded number, so the rounding cancels in O / l exactly like any other
//    common factor.  The maximum is only RAISED when a row's tile maximum exceeds the folded one by more than 6 (P <= 64
//    in f16; the guide's T13 deferred rescale), or on the first tile; that path subtracts the increment explicitly.
// The LDS image (K / V^T tiles incl. padding) is cleared with 16-byte stores before the first tile (round 5: the rolled loop of
// ds_write_b16 it replaces was 169 trips per thread at d = 160, ~2 us of an 18 us launch; -0.95 % per batch, same bits --
// profiles/r05_e2e_ab_candidates.log.  s_setprio(1) around the two MFMA clusters of a tile, the guide's T5, measured +0.2 %
// here and is not compiled in.)
template <int D, int NWAVES, bool PV16, bool FOLD = false>
__global__ __launch_bounds__(NWAVES * 64, (D <= 40 ? 4 : 1)) void attention2_kernel(const AttnParams p) {
  static_assert(!PV16 || D == 40, "the 16x16x32 PV path is laid out for d = 40 (48 padded rows, row 40 = ones)");
  static_assert(!FOLD || D == 40, "the folded maximum uses contraction slot 40 of the d = 40 build (DQK = 48)");
  constexpr int KV_TILE = 64, NU = 2;
  constexpr int NTHR = NWAVES * 64;
  constexpr int QB = NWAVES * 32;
  constexpr int DQK = (D + 15) / 16 * 16;
  constexpr int DV = PV16 ? 48 : (D + 31) / 32 * 32;
  constexpr int VT_LD = PV16 ? KV_TILE : KV_TILE + 4;
  constexpr int K_LD = DQK + 8;
  constexpr int NS = DQK / 16;
  constexpr int ND = DV / 32;        // 32-row O^T tiles (PV on 32x32x16)
  constexpr int NDT = DV / 16;       // 16-row O^T tiles (PV on 16x16x32)
  constexpr bool SUM_MFMA = DV > D;
  constexpr int KCH = D / 8;
  constexpr int K_CHUNKS = KV_TILE * KCH;
  constexpr int V_CHUNKS = D * (KV_TILE / 8);
  constexpr int K_PT = (K_CHUNKS + NTHR - 1) / NTHR;
  constexpr int V_PT = (V_CHUNKS + NTHR - 1) / NTHR;
  constexpr int K_TILE_HALFS = KV_TILE * K_LD;
  constexpr int V_TILE_HALFS = DV * VT_LD;
  __shared__ __attribute__((aligned(16))) half_t lds[2 * K_TILE_HALFS + 2 * V_TILE_HALFS];
  half_t* const Ks = lds;
  half_t* const Vts = lds + 2 * K_TILE_HALFS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nqb = (p.Nq + QB - 1) / QB;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = lin / nqb;
  const int qb = lin - bh * nqb;
  const int b = bh / p.H, h = bh - b * p.H;
  const int q_row = qb * QB + wave * 32 + l31;

  static_assert((2 * K_TILE_HALFS + 2 * V_TILE_HALFS) % 8 == 0, "16-byte clears");
  for (int i = tid; i < (2 * K_TILE_HALFS + 2 * V_TILE_HALFS) / 8; i += NTHR)
    reinterpret_cast<uint4*>(lds)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  if (SUM_MFMA) {   // row D of V^T = 1: O^T[D, q] accumulates sum_kv P (every slot of the row, so the swizzle is moot)
    for (int i = tid; i < 2 * KV_TILE; i += NTHR)
      Vts[(i / KV_TILE) * V_TILE_HALFS + D * VT_LD + (i % KV_TILE)] = (half_t)1.f;
  }
  if (FOLD) {       // column D of K = 1 in both stages (the staging writes columns 0 .. D - 1 only)
    for (int i = tid; i < 2 * KV_TILE; i += NTHR)
      Ks[(i / KV_TILE) * K_TILE_HALFS + (i % KV_TILE) * K_LD + D] = (half_t)1.f;
  }

  half8_t qf[NS];
  {
    const half_t* qp = p.Q + (long)b * p.q_bs + (long)min(q_row, p.Nq - 1) * p.ldq + h * D;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int d = s * 16 + hi * 8;
      Pack16 t;
      t.u = *reinterpret_cast<const uint4*>(qp + min(d, D - 8));   // unconditional; padding slots zeroed below
      if (d >= D) t.u = make_uint4(0, 0, 0, 0);
      qf[s] = t.h;
      if (FOLD) {
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[s][e] = (half_t)((float)qf[s][e] * p.scale_log2);
      }
    }
  }
  bool first_tile = true;   // FOLD: the first tile always sets the folded maximum

  float16_t o32[PV16 ? 1 : ND];          // PV on 32x32x16: O^T tile i, col = q = l31
  float4_t o16[PV16 ? NDT : 1][2];       // PV on 16x16x32: [d tile][q tile], col = q % 16 = lane & 15, rows 4 (lane >> 4) + r
#pragma unroll
  for (int i = 0; i < (PV16 ? 1 : ND); ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o32[i][r] = 0.f;
#pragma unroll
  for (int i = 0; i < (PV16 ? NDT : 1); ++i)
#pragma unroll
    for (int q = 0; q < 2; ++q) o16[i][q] = (float4_t){0.f, 0.f, 0.f, 0.f};
  float m_run = FOLD ? 0.f : -INFINITY, l_run = 0.f;

  const half_t* kbase = p.K + (long)b * p.k_bs + h * D;
  const half_t* vbase = p.Vt + (long)h * D * p.ldvt + (long)b * p.vt_bs;
  const float c = p.scale_log2;

  // ---- staging: chunk = 16 bytes; thread t moves K chunks t + NTHR j and V^T chunks (t + NTHR / 2) % NTHR + NTHR j ----
  // (inactive slots load a clamped, valid chunk and do not store it: no branch around a load.  Row / chunk indices are
  //  recomputed where the ragged tile needs them: kept live they cost the d = 40 build its 128-register budget.)
  auto k_chunk = [&](int j, int& row, int& cc) __attribute__((always_inline)) {
    const int chc = min(tid + NTHR * j, K_CHUNKS - 1);
    row = chc / KCH;
    cc = chc - row * KCH;
  };
  auto v_chunk = [&](int j, int& d, int& cc) __attribute__((always_inline)) {
    const int chc = min(((tid + NTHR / 2) & (NTHR - 1)) + NTHR * j, V_CHUNKS - 1);
    d = chc / (KV_TILE / 8);
    cc = chc % (KV_TILE / 8);
  };
  int k_lds[K_PT], v_lds0[V_PT];
  bool k_on[K_PT], v_on[V_PT];
  const half_t* kptr[K_PT];
  const half_t* vptr[V_PT];
#pragma unroll
  for (int j = 0; j < K_PT; ++j) {
    int row, cc;
    k_chunk(j, row, cc);
    k_on[j] = tid + NTHR * j < K_CHUNKS;
    k_lds[j] = row * K_LD + cc * 8;
    kptr[j] = kbase + (long)row * p.ldk + cc * 8;
  }
#pragma unroll
  for (int j = 0; j < V_PT; ++j) {
    int d, cc;
    v_chunk(j, d, cc);
    v_on[j] = ((tid + NTHR / 2) & (NTHR - 1)) + NTHR * j < V_CHUNKS;
    if (PV16) {
      const int dl_ = d & 15;
      const int sg_ = ((dl_ >> 1) & 3) | (((dl_ >> 3) & 1) << 3);
      v_lds0[j] = d * VT_LD + 4 * ((2 * cc) ^ sg_);   // second 8-byte half: slot (2 cc + 1) ^ sigma = this address ^ 4 halfs
    } else {
      v_lds0[j] = d * VT_LD + cc * 8;                 // second half: + 4 halfs
    }
    vptr[j] = vbase + (long)d * p.ldvt + cc * 8;
  }
  const int v_last = max(0, ((p.Nk + 7) & ~7) - 8);
  // (first-class vector values, not the uint4 struct: a struct copy global -> private -> LDS is folded by the compiler's
  //  memcpy forwarding into ONE copy at the store point, i.e. the prefetch load lands right in front of its ds_write)
  u32x4 kregA[K_PT], vregA[V_PT];

  auto load_full = [&](u32x4* kreg, u32x4* vreg) __attribute__((always_inline)) {   // the tile the pointers stand on
#pragma unroll
    for (int j = 0; j < K_PT; ++j) {
      kreg[j] = *reinterpret_cast<const u32x4*>(kptr[j]);
      kptr[j] += (long)KV_TILE * p.ldk;
    }
#pragma unroll
    for (int j = 0; j < V_PT; ++j) {
      vreg[j] = *reinterpret_cast<const u32x4*>(vptr[j]);
      vptr[j] += KV_TILE;
    }
  };
  auto load_ragged = [&](u32x4* kreg, u32x4* vreg, int kv0) __attribute__((always_inline)) {   // clamped to valid rows / chunks
#pragma unroll
    for (int j = 0; j < K_PT; ++j) {
      int row, cc;
      k_chunk(j, row, cc);
      kreg[j] = *reinterpret_cast<const u32x4*>(kbase + (long)min(kv0 + row, p.Nk - 1) * p.ldk + cc * 8);
    }
#pragma unroll
    for (int j = 0; j < V_PT; ++j) {
      int d, cc;
      v_chunk(j, d, cc);
      vreg[j] = *reinterpret_cast<const u32x4*>(vbase + (long)d * p.ldvt + min(kv0 + cc * 8, v_last));
    }
  };
  // RAG is a constant at every call site (the lambdas are not generic on purpose: a generic lambda is inlined too late
  // for the staging arrays to be promoted to registers)
  auto store_tile = [&](const u32x4* kreg, const u32x4* vreg, int stage, const bool RAG, int kv0) __attribute__((always_inline)) {
    half_t* Kd = Ks + stage * K_TILE_HALFS;
    half_t* Vd = Vts + stage * V_TILE_HALFS;
#pragma unroll
    for (int j = 0; j < K_PT; ++j)
      if (k_on[j]) *reinterpret_cast<u32x4*>(Kd + k_lds[j]) = kreg[j];
#pragma unroll
    for (int j = 0; j < V_PT; ++j) {
      unsigned w0 = vreg[j][0], w1 = vreg[j][1], w2 = vreg[j][2], w3 = vreg[j][3];
      if (RAG) {   // keys past Nk: P is 0 there, but 0 x NaN garbage must not reach the accumulator
        int d, cc;
        v_chunk(j, d, cc);
        const int valid = p.Nk - (kv0 + cc * 8);
        w0 = valid >= 2 ? w0 : (valid == 1 ? (w0 & 0xFFFFu) : 0u);
        w1 = valid >= 4 ? w1 : (valid == 3 ? (w1 & 0xFFFFu) : 0u);
        w2 = valid >= 6 ? w2 : (valid == 5 ? (w2 & 0xFFFFu) : 0u);
        w3 = valid >= 8 ? w3 : (valid == 7 ? (w3 & 0xFFFFu) : 0u);
      }
      if (v_on[j]) {
        *reinterpret_cast<uint2*>(Vd + v_lds0[j]) = make_uint2(w0, w1);
        *reinterpret_cast<uint2*>(Vd + (PV16 ? v_lds0[j] ^ 4 : v_lds0[j] + 4)) = make_uint2(w2, w3);
      }
    }
  };

  // V^T fragment addressing of the 16x16x32 path (halfs, relative to the tile): row dl, slots (4 bb + hi) and + 2
  const int dl = lane & 15, kg = lane >> 4;
  const int sg = ((dl >> 1) & 3) | (((dl >> 3) & 1) << 3);
  const int x0 = (4 * (kg & 1) + (kg >> 1)) ^ (sg & 7), x1 = (4 * (kg & 1) + (kg >> 1) + 2) ^ (sg & 7);
  const int u_flip = sg >> 3;
  const int vrow16 = dl * VT_LD;

  auto compute_tile = [&](int stage, const bool RAG, int kv0) __attribute__((always_inline)) {
    const half_t* Kt = Ks + stage * K_TILE_HALFS;
    const half_t* Vt = Vts + stage * V_TILE_HALFS;
    float16_t st[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[u][r] = 0.f;
      const half_t* kp = Kt + (u * 32 + l31) * K_LD + hi * 8;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const half8_t kf = *reinterpret_cast<const half8_t*>(kp + s * 16);
        st[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], st[u], 0, 0, 0);
      }
    }
    if (RAG) {
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kv0 + u * 32 + mfma32_row(r, hi) >= p.Nk) st[u][r] = -INFINITY;
    }
    float mx = st[0][0];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[u][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    auto rescale_o = [&](float alpha) __attribute__((always_inline)) {
      if constexpr (PV16) {
        const float a0 = __shfl(alpha, dl, 64), a1 = __shfl(alpha, dl + 16, 64);   // accumulator columns: q = dl + 16 qt
#pragma unroll
        for (int i = 0; i < NDT; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            o16[i][0][r] *= a0;
            o16[i][1][r] *= a1;
          }
      } else {
#pragma unroll
        for (int i = 0; i < ND; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o32[i][r] *= alpha;
      }
    };
    float rs = 0.f;
    half8_t pf[NU][2];
    if constexpr (FOLD) {
      // st = s' - m_run already (m_run: the f16-representable maximum folded into the query fragment; 0 before the first tile)
      if (first_tile || __any(mx > 6.0f)) {
        const bool up = first_tile || mx > 0.f;
        const float m_new = up ? (float)(half_t)(m_run + mx) : m_run;
        const float delta = m_new - m_run;                       // exact: both are f16 values
        rescale_o(__builtin_amdgcn_exp2f(-delta));               // (accumulators are 0 on the first tile)
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) st[u][r] -= delta;
        m_run = m_new;
        if (hi) qf[NS - 1][0] = (half_t)(-m_new);                // contraction slot D = 40: k block 2, upper lane half, element 0
        first_tile = false;
      }
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[u][bb][j] = (half_t)__builtin_amdgcn_exp2f(st[u][bb * 8 + j]);
    } else {
      const float m_new = fmaxf(m_run, mx);
      if (__any(m_new > m_run)) {
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);   // 1 for rows whose max did not move; 0 at the start
        l_run *= alpha;
        rescale_o(alpha);
        m_run = m_new;
      }
      const float mc = m_run * c;
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float e = __builtin_amdgcn_exp2f(fmaf(st[u][bb * 8 + j], c, -mc));
            if (!SUM_MFMA) rs += e;
            pf[u][bb][j] = (half_t)e;
          }
    }
    if (!SUM_MFMA) {
      rs += __shfl_xor(rs, 32, 64);
      l_run += rs;
    }
    if constexpr (PV16) {
      union H8 {
        half8_t h;
        unsigned w[4];
      };
      half8_t pb[NU][2];   // [32-key half][q tile]
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        H8 a, bq, r0, r1;
        a.h = pf[u][0];
        bq.h = pf[u][1];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const auto sw = __builtin_amdgcn_permlane16_swap(a.w[k], bq.w[k], false, false);
          r0.w[k] = sw[0];   // rows: (q 0-15, hi 0, bb 0) (q 0-15, hi 0, bb 1) (q 0-15, hi 1, bb 0) (q 0-15, hi 1, bb 1)
          r1.w[k] = sw[1];   // the same for q 16-31
        }
        pb[u][0] = r0.h;
        pb[u][1] = r1.h;
      }
  #pragma unroll
      for (int i = 0; i < NDT; ++i)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const half_t* vp = Vt + i * 16 * VT_LD + vrow16 + 32 * (u ^ u_flip);
          const half4_t lo4 = *reinterpret_cast<const half4_t*>(vp + 4 * x0);
          const half4_t hi4 = *reinterpret_cast<const half4_t*>(vp + 4 * x1);
          half8_t vf;
          vf[0] = lo4[0]; vf[1] = lo4[1]; vf[2] = lo4[2]; vf[3] = lo4[3];
          vf[4] = hi4[0]; vf[5] = hi4[1]; vf[6] = hi4[2]; vf[7] = hi4[3];
          o16[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pb[u][0], o16[i][0], 0, 0, 0);
          o16[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pb[u][1], o16[i][1], 0, 0, 0);
        }
      } else {
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const half_t* vp = Vt + (i * 32 + l31) * VT_LD + 4 * hi;
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) {
            const half4_t lo4 = *reinterpret_cast<const half4_t*>(vp + u * 32 + bb * 16);
            const half4_t hi4 = *reinterpret_cast<const half4_t*>(vp + u * 32 + bb * 16 + 8);
            half8_t vf;
            vf[0] = lo4[0]; vf[1] = lo4[1]; vf[2] = lo4[2]; vf[3] = lo4[3];
            vf[4] = hi4[0]; vf[5] = hi4[1]; vf[6] = hi4[2]; vf[7] = hi4[3];
            o32[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[u][bb], o32[i], 0, 0, 0);
          }
      }
    }
  };

  const int nfull = p.Nk / KV_TILE;
  const bool rag = (p.Nk % KV_TILE) != 0;
  // first tile
  if (nfull > 0) {
    load_full(kregA, vregA);
    store_tile(kregA, vregA, 0, false, 0);
  } else {
    load_ragged(kregA, vregA, 0);
    store_tile(kregA, vregA, 0, true, 0);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) on every path into the loop (Q fragments included)
  __syncthreads();
  int t = 0;
  for (; t + 1 < nfull; ++t) {          // full tile followed by a full tile: the hot loop
    load_full(kregA, vregA);
    compute_tile(t & 1, false, 0);
    store_tile(kregA, vregA, (t & 1) ^ 1, false, 0);
    __syncthreads();
  }
  if (nfull > 0) {                      // last full tile; its successor is the ragged tile or nothing
    if (rag) load_ragged(kregA, vregA, (t + 1) * KV_TILE);
    compute_tile(t & 1, false, 0);
    if (rag) store_tile(kregA, vregA, (t & 1) ^ 1, true, (t + 1) * KV_TILE);
    __syncthreads();
    ++t;
  }
  if (rag) compute_tile(t & 1, true, t * KV_TILE);

  // ---- epilogue: O[q, d] = O^T[d, q] / l ----
  if constexpr (PV16) {
    // row 40 of O^T (d tile 2, local row 8 = lane row 2, register 0) = sum_kv P of query (lane & 15) + 16 qt
    const float l0 = __shfl(o16[2][0][0], 32 + dl, 64), l1 = __shfl(o16[2][1][0], 32 + dl, 64);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int q = qb * QB + wave * 32 + qt * 16 + dl;
      const float inv = 1.0f / (qt ? l1 : l0);
      if (q < p.Nq) {
        half_t* op = p.O + (long)b * p.o_bs + (long)q * p.ldo + h * D;
#pragma unroll
        for (int i = 0; i < NDT; ++i) {
          const int d0 = i * 16 + 4 * kg;
          if (d0 < D) {
            half4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (half_t)(o16[i][qt][e] * inv);
            *reinterpret_cast<half4_t*>(op + d0) = o;
          }
        }
      }
    }
  } else {
    if (SUM_MFMA) {
      constexpr int lr = D % 32;
      constexpr int src_hi = (lr >> 2) & 1;
      constexpr int reg = (lr & 3) + 4 * (lr >> 3);
      const float v = o32[D / 32][reg];
      l_run = __shfl(v, src_hi * 32 + l31, 64);
    }
    if (q_row < p.Nq) {
      const float inv = 1.0f / l_run;
      half_t* op = p.O + (long)b * p.o_bs + (long)q_row * p.ldo + h * D;
#pragma unroll
      for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int d0 = i * 32 + 8 * rq + 4 * hi;
          if (d0 < D) {
            half4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (half_t)(o32[i][rq * 4 + e] * inv);
            *reinterpret_cast<half4_t*>(op + d0) = o;
          }
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// d = 512, one head: the VAE mid-block attention (autokl_modules.py:186-197).  Same transposed formulation
// (S^T = K Q^T, O^T = V^T P^T on v_mfma_f32_32x32x16_f16, online softmax per query = per lane column), but the head does
// not fit one wave's registers as a whole: a wave keeps the Q fragments of its 32 queries for the full contraction
// (32 k-steps x 4 VGPRs = 128) and the O^T accumulators of ONE 128-column slice of V (4 tiles x 16 = 64); the four
// slices are four blocks (blockIdx -> (batch, query tile, slice)), each of which recomputes QK^T -- 4 x 34 + 34 GFLOP per
// 512^2 image instead of 68, for no [N, N] score matrix in HBM at any resolution.  KV tile = 32 keys: the K image
// (32 x 520 halfs) and the V^T slice image (128 x 36 halfs) are double buffered in 85 KB of LDS; one block per CU,
// four waves on four SIMDs, up to 512 unified registers per wave (accumulators in AGPRs).
// Requires Nk % 32 == 0 (the VAE's token counts are multiples of 64).
// ------------------------------------------------------------------------------------------------
template <int NSL>   // V column slices per query tile: 4 (128 columns, 64 accumulator registers) or 2 (256 columns, 128)
__global__ __launch_bounds__(256, 1) void attention512_kernel(const AttnParams p) {
  constexpr int D = 512, DVC = D / NSL, KV_TILE = 32, NTHR = 256, QB = 128;
  constexpr int K_LD = D + 8;           // 1040-B rows: an odd multiple of 16 B, conflict-free ds_read_b128 over 32 rows
  constexpr int VT_LD = KV_TILE + 4;    // 72-B rows: 18 dwords, conflict-free ds_read_b64 over 32 rows
  constexpr int NS = D / 16;            // 32 k-steps of the QK^T contraction
  constexpr int ND = DVC / 32;          // 4 O^T tiles
  constexpr int K_PT = KV_TILE * (D / 8) / NTHR;        // 8 16-byte chunks of K per thread and tile
  constexpr int V_PT = DVC * (KV_TILE / 8) / NTHR;      // 2 of V^T
  constexpr int K_TILE_HALFS = KV_TILE * K_LD;
  constexpr int V_TILE_HALFS = DVC * VT_LD;
  __shared__ __attribute__((aligned(16))) half_t lds[2 * K_TILE_HALFS + 2 * V_TILE_HALFS];
  half_t* const Ks = lds;
  half_t* const Vts = lds + 2 * K_TILE_HALFS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nqb = (p.Nq + QB - 1) / QB;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  constexpr int SL_SH = NSL == 4 ? 2 : 1;
  const int slice = lin & (NSL - 1);             // the slices of a query tile are neighbours: they read the same K stream
  const int qb = (lin >> SL_SH) % nqb;
  const int b = (lin >> SL_SH) / nqb;
  const int q_row = qb * QB + wave * 32 + l31;

  half8_t qf[NS];
  {
    const half_t* qp = p.Q + (long)b * p.q_bs + (long)min(q_row, p.Nq - 1) * p.ldq + hi * 8;
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = *reinterpret_cast<const half8_t*>(qp + s * 16);
  }
  float16_t o32[ND];
#pragma unroll
  for (int i = 0; i < ND; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o32[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float c = p.scale_log2;

  // staging: thread t moves K chunks t + 256 j (row = chunk / 64, 16-byte column = chunk % 64) and V^T chunks t + 256 j
  // (row = chunk / 4 of the slice, 8-key column = chunk % 4); the source pointers advance by a constant per tile
  const half_t* kptr[K_PT];
  const half_t* vptr[V_PT];
  int k_lds[K_PT], v_lds[V_PT];
#pragma unroll
  for (int j = 0; j < K_PT; ++j) {
    const int ch = tid + NTHR * j, row = ch >> 6, cc = ch & 63;
    k_lds[j] = row * K_LD + cc * 8;
    kptr[j] = p.K + (long)b * p.k_bs + (long)row * p.ldk + cc * 8;
  }
#pragma unroll
  for (int j = 0; j < V_PT; ++j) {
    const int ch = tid + NTHR * j, d = ch >> 2, cc = ch & 3;
    v_lds[j] = d * VT_LD + cc * 8;
    vptr[j] = p.Vt + (long)(slice * DVC + d) * p.ldvt + (long)b * p.vt_bs + cc * 8;
  }
  u32x4 kreg[K_PT], vreg[V_PT];
  auto load_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < K_PT; ++j) {
      kreg[j] = *reinterpret_cast<const u32x4*>(kptr[j]);
      kptr[j] += (long)KV_TILE * p.ldk;
    }
#pragma unroll
    for (int j = 0; j < V_PT; ++j) {
      vreg[j] = *reinterpret_cast<const u32x4*>(vptr[j]);
      vptr[j] += KV_TILE;
    }
  };
  auto store_tile = [&](int stage) __attribute__((always_inline)) {
    half_t* Kd = Ks + stage * K_TILE_HALFS;
    half_t* Vd = Vts + stage * V_TILE_HALFS;
#pragma unroll
    for (int j = 0; j < K_PT; ++j) *reinterpret_cast<u32x4*>(Kd + k_lds[j]) = kreg[j];
#pragma unroll
    for (int j = 0; j < V_PT; ++j) {   // 72-byte rows are 8-byte aligned only
      *reinterpret_cast<uint2*>(Vd + v_lds[j]) = make_uint2(vreg[j][0], vreg[j][1]);
      *reinterpret_cast<uint2*>(Vd + v_lds[j] + 4) = make_uint2(vreg[j][2], vreg[j][3]);
    }
  };
  auto compute_tile = [&](int stage) __attribute__((always_inline)) {
    const half_t* Kt = Ks + stage * K_TILE_HALFS;
    const half_t* Vt = Vts + stage * V_TILE_HALFS;
    float16_t st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
    const half_t* kp = Kt + l31 * K_LD + hi * 8;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const half8_t kf = *reinterpret_cast<const half8_t*>(kp + s * 16);
      st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], st, 0, 0, 0);
    }
    float mx = st[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    if (__any(m_new > m_run)) {
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);   // 0 on the first tile, 1 where the maximum stayed
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o32[i][r] *= alpha;
      m_run = m_new;
    }
    const float mc = m_run * c;
    float rs = 0.f;
    half8_t pf[2];
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float e = __builtin_amdgcn_exp2f(fmaf(st[bb * 8 + j], c, -mc));
        rs += e;
        pf[bb][j] = (half_t)e;
      }
    rs += __shfl_xor(rs, 32, 64);
    l_run += rs;
    // P^T slot j of lane half hi in 16-key block bb is key 16 bb + 8 (j / 4) + 4 hi + j % 4: read V^T in that order
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const half_t* vp = Vt + (i * 32 + l31) * VT_LD + 4 * hi;
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        const half4_t lo4 = *reinterpret_cast<const half4_t*>(vp + bb * 16);
        const half4_t hi4 = *reinterpret_cast<const half4_t*>(vp + bb * 16 + 8);
        half8_t vf;
        vf[0] = lo4[0]; vf[1] = lo4[1]; vf[2] = lo4[2]; vf[3] = lo4[3];
        vf[4] = hi4[0]; vf[5] = hi4[1]; vf[6] = hi4[2]; vf[7] = hi4[3];
        o32[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[bb], o32[i], 0, 0, 0);
      }
    }
  };

  const int ntiles = p.Nk / KV_TILE;
  load_tile();
  store_tile(0);
  __syncthreads();
  for (int t = 0; t + 1 < ntiles; ++t) {
    load_tile();
    __builtin_amdgcn_sched_barrier(0);   // the next tile's loads stay in front of this tile's MFMAs (left alone, the
    compute_tile(t & 1);                 // scheduler sinks them behind the QK^T chain to shorten their live ranges)
    store_tile((t & 1) ^ 1);
    __syncthreads();
  }
  compute_tile((ntiles - 1) & 1);

  if (q_row < p.Nq) {
    const float inv = 1.0f / l_run;
    half_t* op = p.O + (long)b * p.o_bs + (long)q_row * p.ldo + slice * DVC;
#pragma unroll
    for (int i = 0; i < ND; ++i)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        half4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (half_t)(o32[i][rq * 4 + e] * inv);
        *reinterpret_cast<half4_t*>(op + i * 32 + 8 * rq + 4 * hi) = o;
      }
  }
}

static int launch512(const AttnParams& p, hipStream_t s) {
  if (p.H != 1 || p.Nk % 32 != 0) return PFD_ESHAPE;
  const bool prof = pfd_prof_on();
  if (prof)
    pfd_prof_begin(8, 4.0 * p.B * (double)p.Nq * p.Nk * 512, 2.0 * p.B * 512 * (2.0 * p.Nq + 2.0 * p.Nk), s);
  // PFD_ATTN512_SLICES: 2 (default since round 4) or 4 -- two 256-column slices recompute QK^T twice instead of four times
  // and keep 128 accumulator registers per wave in AGPRs (256 + 129 registers, no scratch): 0.363 vs 0.547 ms at the C2
  // shape, 1.39 vs 1.61 ms at 96^2 (profiles/r03_k18_vae_attention512.log).  One 512-column slice (256 accumulator
  // registers) crashes hipcc 7.2's register allocator, so it is not instantiated.  Read per launch (once per decoded
  // batch), so the GPU suite exercises both instantiations in one process.
  const char* nsl_env = getenv("PFD_ATTN512_SLICES");
  const int nsl = nsl_env && atoi(nsl_env) == 4 ? 4 : 2;
  dim3 grid(((p.Nq + 127) / 128) * nsl * p.B);
  if (nsl == 2) hipLaunchKernelGGL(attention512_kernel<2>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(attention512_kernel<4>, grid, dim3(256), 0, s, p);
  if (prof) pfd_prof_end(s);
  return pfd_check_launch("pfd_attention_f16");
}

// PFD_ATTN_FORCE8=1 takes the 8-wave form for every d = 40 problem (test hook: selftest --attn, tools/cpu_emu).
static bool attn_force8() {
  static const bool f = getenv("PFD_ATTN_FORCE8") && atoi(getenv("PFD_ATTN_FORCE8")) != 0;
  return f;
}

template <int D>
int launch(const AttnParams& p, hipStream_t s) {
  const bool prof = pfd_prof_on();
  if (prof)
    pfd_prof_begin(8, 4.0 * p.B * p.H * (double)p.Nq * p.Nk * D,
                   2.0 * p.B * p.H * D * (2.0 * p.Nq + 2.0 * p.Nk), s);
  // 8-wave blocks pay when a block has many queries to amortise the staging over and the grid still fills the chip twice
  if constexpr (D == 40) {
    // round 6: the software-pipelined 64-queries-per-wave kernel (attention3.hip) for self-attention-sized problems (decided on
    // hardware against the kernel below: 260 vs 284 us at B 8 / 64^2, profiles/r06_bench_attn_*.log; the A/B switch is gone)
    if (pfd_attention3_takes(p)) {
      pfd_attention3_launch(p, s);
      if (prof) pfd_prof_end(s);
      return pfd_check_launch("pfd_attention_f16");
    }
  }
  const bool big = p.Nq >= 1024 && (long)p.B * p.H * ((p.Nq + 255) / 256) >= 512;
  const bool w8 = D == 40 && (big || attn_force8());
  // (round 5: 64-query blocks of two waves for the grids that fill less than the chip -- d = 160 at 16^2 / 8^2 -- measured
  //  the same or slightly slower, profiles/r05_e2e_ab_candidates.log)
  const int qb = w8 ? 256 : 128;
  dim3 grid(((p.Nq + qb - 1) / qb) * p.H * p.B);
  if constexpr (D == 40) {
    // 8 waves: PV on 16x16x32 + the maximum folded into the QK^T MFMA (the 4-wave PV16 build spills 24 bytes at 128 VGPRs)
    if (w8) hipLaunchKernelGGL((attention2_kernel<D, 8, true, true>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((attention2_kernel<D, 4, false>), grid, dim3(256), 0, s, p);
  } else {
    hipLaunchKernelGGL((attention2_kernel<D, 4, false>), grid, dim3(256), 0, s, p);
  }
  if (prof) pfd_prof_end(s);
  return pfd_check_launch("pfd_attention_f16");
}

}  // namespace

extern "C" int pfd_attention_f16(const PfdAttnDesc* d, pfd_stream_t stream) {
  if (!d || !d->Q || !d->K || !d->Vt || !d->O) return PFD_EINVAL;
  if (d->B <= 0 || d->H <= 0 || d->Nq <= 0 || d->Nk <= 0) return PFD_EINVAL;
  if ((d->ldq & 7) || (d->ldk & 7) || (d->ldo & 7) || (d->ldvt & 7) || (d->vt_bs & 7) || (d->q_bs & 7) ||
      (d->k_bs & 7) || (d->o_bs & 7))
    return PFD_EINVAL;
  if ((reinterpret_cast<uintptr_t>(d->Q) & 15) || (reinterpret_cast<uintptr_t>(d->K) & 15) ||
      (reinterpret_cast<uintptr_t>(d->Vt) & 15) || (reinterpret_cast<uintptr_t>(d->O) & 15))
    return PFD_EINVAL;
  AttnParams p;
  p.Q = (const half_t*)d->Q; p.K = (const half_t*)d->K; p.Vt = (const half_t*)d->Vt; p.O = (half_t*)d->O;
  p.ldq = d->ldq; p.ldk = d->ldk; p.ldvt = d->ldvt; p.ldo = d->ldo;
  p.q_bs = d->q_bs; p.k_bs = d->k_bs; p.vt_bs = d->vt_bs; p.o_bs = d->o_bs;
  p.B = d->B; p.H = d->H; p.Nq = d->Nq; p.Nk = d->Nk; p.D = d->D;
  p.scale_log2 = d->scale * 1.4426950408889634f;
  hipStream_t s = (hipStream_t)stream;
  switch (d->D) {
    case 40: return launch<40>(p, s);
    case 80: return launch<80>(p, s);
    case 96: return launch<96>(p, s);
    case 160: return launch<160>(p, s);
    case 512: return launch512(p, s);
    default: return PFD_ESHAPE;
  }
}
