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
#include "pfd_common.h"

namespace {

#ifndef ATT_ABL
#define ATT_ABL 0   // measurement-only ablation bits: 1 no exp, 2 no PV MFMAs, 4 no K/V loads + staging, 8 no QK MFMAs
#endif

struct AttnParams {
  const half_t* Q;
  const half_t* K;
  const half_t* Vt;
  half_t* O;
  long ldq, ldk, ldvt, ldo;
  long q_bs, k_bs, vt_bs, o_bs;
  int B, H, Nq, Nk, D;
  float scale_log2;
};

// Occupancy: D = 40 needs 136 VGPRs when left alone (3 waves per SIMD); asked for 4 waves per SIMD the
// allocation fits 128 without scratch and self-attention at N = 4096 gains 4.6 % (452 -> 473 TF), the 148-key
// cross-attention 16 % (profiles/r02_attention_ablation.log).  A 128-key tile (half the barriers, 62 KB LDS,
// 184 VGPRs) measured 2 % slower.  Larger head dims keep their natural allocation.
template <int D, int KV_TILE>
__global__ __launch_bounds__(256, (D <= 40 ? 4 : 1)) void attention_kernel(const AttnParams p) {
  constexpr int NU = KV_TILE / 32;    // 32-key halves per tile
  constexpr int VT_LD = KV_TILE + 4;  // halfs; 136-B rows: conflict-free ds_read_b64 over 32 rows
  constexpr int DQK = (D + 15) / 16 * 16;
  constexpr int DV = (D + 31) / 32 * 32;
  constexpr int K_LD = DQK + 8;  // halfs; (DQK+8)*2 B is an odd multiple of 16 B for D in {40,80,96,160}
  constexpr int NS = DQK / 16;
  constexpr int ND = DV / 32;
  constexpr bool SUM_MFMA = DV > D;       // a spare V^T row carries the softmax denominator
  constexpr int KCH = D / 8;              // 16-B chunks per K row
  constexpr int K_CHUNKS = KV_TILE * KCH; // per tile
  constexpr int V_CHUNKS = D * (KV_TILE / 8);
  constexpr int K_PT = (K_CHUNKS + 255) / 256;
  constexpr int V_PT = (V_CHUNKS + 255) / 256;
  constexpr int K_TILE_HALFS = KV_TILE * K_LD;
  constexpr int V_TILE_HALFS = DV * VT_LD;
  __shared__ __attribute__((aligned(16))) half_t Ks[2 * K_TILE_HALFS];
  __shared__ __attribute__((aligned(16))) half_t Vts[2 * V_TILE_HALFS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  // 1-D grid, XCD-aware: block id -> (sample*head, q block) such that all q blocks of one head run
  // on ONE XCD (block b lands on XCD b % 8) and share that L2's copy of the head's K/V.  Without
  // it every XCD fetches its own copy: 357 MB instead of 63 MB per launch at N = 4096 (PMC
  // FETCH_SIZE, profiles/r01_pmc_traffic_per_shape.md).  Placement is speed only.
  const int nqb = (p.Nq + 127) / 128;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = lin / nqb;
  const int qb = lin - bh * nqb;
  const int b = bh / p.H, h = bh - b * p.H;
  const int q_row = qb * 128 + wave * 32 + l31;

  // zero the LDS padding once (columns D..DQK of K, rows D..DV of V^T); row D of V^T = 1 (row sums)
  for (int i = tid; i < 2 * K_TILE_HALFS; i += 256) Ks[i] = (half_t)0.f;
  for (int i = tid; i < 2 * V_TILE_HALFS; i += 256) Vts[i] = (half_t)0.f;
  __syncthreads();
  if (SUM_MFMA) {
    for (int i = tid; i < 2 * KV_TILE; i += 256)
      Vts[(i / KV_TILE) * V_TILE_HALFS + D * VT_LD + (i % KV_TILE)] = (half_t)1.f;
  }

  // Q fragments: B operand, col = q = lane&31, k = d
  half8_t qf[NS];
  {
    const half_t* qp = p.Q + (long)b * p.q_bs + (long)q_row * p.ldq + h * D;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int d = s * 16 + hi * 8;
      Pack16 t;
      t.u = make_uint4(0, 0, 0, 0);
      if (q_row < p.Nq && d < D) t.u = *reinterpret_cast<const uint4*>(qp + d);
      qf[s] = t.h;
    }
  }

  float16_t o_acc[ND];
#pragma unroll
  for (int i = 0; i < ND; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;  // m_run: running max of the RAW scores

  const half_t* kbase = p.K + (long)b * p.k_bs + h * D;
  const half_t* vbase = p.Vt + (long)h * D * p.ldvt + (long)b * p.vt_bs;
  const float c = p.scale_log2;

  // one staging register set: tile t+1 is loaded (global -> registers) while tile t is computed and is
  // written to the other LDS stage afterwards.  Rows / key chunks past Nk are CLAMPED to valid
  // addresses rather than predicated per row (their scores are masked to -inf, so P = 0).
  uint4 kregA[K_PT], vregA[V_PT];
  const int v_last = max(0, ((p.Nk + 7) & ~7) - 8);
  auto load_tile = [&](uint4* kreg, uint4* vreg, int kv0) {
#pragma unroll
    for (int j = 0; j < K_PT; ++j) {
      const int ch = tid + 256 * j;
      const int row = ch / KCH, cc = ch - row * KCH;
      kreg[j] = make_uint4(0, 0, 0, 0);
      if (ch < K_CHUNKS)
        kreg[j] = *reinterpret_cast<const uint4*>(kbase + (long)min(kv0 + row, p.Nk - 1) * p.ldk + cc * 8);
    }
#pragma unroll
    for (int j = 0; j < V_PT; ++j) {
      const int ch = tid + 256 * j;
      const int d = ch / (KV_TILE / 8), cc = ch % (KV_TILE / 8);
      vreg[j] = make_uint4(0, 0, 0, 0);
      if (ch < V_CHUNKS)   // key chunks past Nk are clamped (their P is 0); the ragged tile zeroes them below
        vreg[j] = *reinterpret_cast<const uint4*>(vbase + (long)d * p.ldvt + min(kv0 + cc * 8, v_last));
    }
  };
  // (the ragged-tile masking lives HERE, after the MFMAs of the current tile: touching the freshly loaded
  //  registers inside load_tile made the compiler wait for the global loads before the MFMAs they are
  //  supposed to overlap with)
  auto store_tile = [&](const uint4* kreg, uint4* vreg, int stage, int kv0) {
    half_t* Kd = Ks + stage * K_TILE_HALFS;
    half_t* Vd = Vts + stage * V_TILE_HALFS;
    if (kv0 + KV_TILE > p.Nk) {  // ragged last tile only (wave-uniform)
#pragma unroll
      for (int j = 0; j < V_PT; ++j) {
        const int valid = p.Nk - (kv0 + ((tid + 256 * j) % (KV_TILE / 8)) * 8);  // valid halfs in this 8-chunk
        unsigned w[4] = {vreg[j].x, vreg[j].y, vreg[j].z, vreg[j].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int v2 = valid - 2 * q;
          w[q] = v2 >= 2 ? w[q] : (v2 == 1 ? (w[q] & 0xFFFFu) : 0u);
        }
        vreg[j] = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
#pragma unroll
    for (int j = 0; j < K_PT; ++j) {
      const int ch = tid + 256 * j;
      const int row = ch / KCH, cc = ch - row * KCH;
      if (ch < K_CHUNKS) *reinterpret_cast<uint4*>(Kd + row * K_LD + cc * 8) = kreg[j];
    }
#pragma unroll
    for (int j = 0; j < V_PT; ++j) {
      const int ch = tid + 256 * j;
      const int d = ch / (KV_TILE / 8), cc = ch % (KV_TILE / 8);
      if (ch < V_CHUNKS) {
        uint2* dst = reinterpret_cast<uint2*>(Vd + d * VT_LD + cc * 8);
        dst[0] = make_uint2(vreg[j].x, vreg[j].y);
        dst[1] = make_uint2(vreg[j].z, vreg[j].w);
      }
    }
  };

  const int ntiles = (p.Nk + KV_TILE - 1) / KV_TILE;
  auto compute_tile = [&](int t, int stage) {
    const int kv0 = t * KV_TILE;
    const half_t* Kt = Ks + stage * K_TILE_HALFS;
    const half_t* Vt = Vts + stage * V_TILE_HALFS;

    // ---- S^T = K . Q^T for the two 32-key halves of the tile ----
    float16_t st[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[u][r] = 0.f;
      const half_t* kp = Kt + (u * 32 + l31) * K_LD + hi * 8;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const half8_t kf = *reinterpret_cast<const half8_t*>(kp + s * 16);
        if (!(ATT_ABL & 8)) st[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], st[u], 0, 0, 0);
        else st[u][s] += (float)kf[0] * (float)qf[s][0];
      }
    }
    // ---- online softmax over this tile's 64 keys ----
    if (kv0 + KV_TILE > p.Nk) {  // ragged last tile: keys past Nk never win the max nor add weight
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
    const float m_new = fmaxf(m_run, mx);  // finite: every tile holds at least one valid key
    if (__any(m_new > m_run)) {            // wave-uniform: rescale only when some row's max moved
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[i][r] *= alpha;
      m_run = m_new;
    }
    const float mc = m_run * c;
    float rs = 0.f;
    half8_t pf[NU][2];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float e = (ATT_ABL & 1) ? fmaf(st[u][bb * 8 + j], c, -mc) : __builtin_amdgcn_exp2f(fmaf(st[u][bb * 8 + j], c, -mc));
          if (!SUM_MFMA) rs += e;
          pf[u][bb][j] = (half_t)e;
        }
    if (!SUM_MFMA) {
      rs += __shfl_xor(rs, 32, 64);
      l_run += rs;
    }

    // ---- O^T += V^T . P^T ----
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
          if (!(ATT_ABL & 2)) o_acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[u][bb], o_acc[i], 0, 0, 0);
          else o_acc[i][0] += (float)pf[u][bb][0] + (float)vf[0];
        }
    }
  };

  load_tile(kregA, vregA, 0);
  store_tile(kregA, vregA, 0, 0);
  // vmcnt(0) on EVERY path into the loop (the waits above sit inside exec-masked blocks): otherwise the
  // compiler must assume the Q fragments may still be in flight at the first MFMA of each tile and, vmcnt
  // being an in-order counter, drains the just-issued K/V prefetch there as well
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int stage = t & 1;
    if (t + 1 < ntiles && !(ATT_ABL & 4)) load_tile(kregA, vregA, (t + 1) * KV_TILE);  // in flight during this tile's MFMAs
    compute_tile(t, stage);
    if (t + 1 < ntiles && !(ATT_ABL & 4)) store_tile(kregA, vregA, stage ^ 1, (t + 1) * KV_TILE);
    __syncthreads();
  }

  // ---- epilogue: O[q, d] = O^T[d, q] / l ----
  if (SUM_MFMA) {
    // row D of O^T = sum_kv P: tile D/32, local row D%32 -> register r with mfma32_row(r, hi') = D%32
    constexpr int lr = D % 32;
    constexpr int src_hi = (lr >> 2) & 1;
    constexpr int reg = (lr & 3) + 4 * (lr >> 3);
    const float v = o_acc[D / 32][reg];
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
          for (int e = 0; e < 4; ++e) o[e] = (half_t)(o_acc[i][rq * 4 + e] * inv);
          *reinterpret_cast<half4_t*>(op + d0) = o;
        }
      }
  }
}

template <int D>
int launch(const AttnParams& p, hipStream_t s) {
  dim3 grid(((p.Nq + 127) / 128) * p.H * p.B);
  const bool prof = pfd_prof_on();
  if (prof)
    pfd_prof_begin(8, 4.0 * p.B * p.H * (double)p.Nq * p.Nk * D,
                   2.0 * p.B * p.H * D * (2.0 * p.Nq + 2.0 * p.Nk), s);
  hipLaunchKernelGGL((attention_kernel<D, 64>), grid, dim3(256), 0, s, p);
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
    default: return PFD_ESHAPE;
  }
}
