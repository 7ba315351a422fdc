// HBM-bound normalisation kernels: GroupNorm(+SiLU) over NHWC (optionally over the virtual
// channel-concat of two tensors), LayerNorm (optionally over the PatchMerging gather) and a
// scaled row softmax.  All statistics in fp32; every global access is a 16-byte vector.
//
// GroupNorm algorithmic HBM bytes per element: 2 (stats read) + 2 (apply read) + 2 (write).
#include <stdlib.h>

#include "pfd_common.h"

namespace {

constexpr int GN_MAX_CHUNKS = 256;
constexpr int GN_MAX_G = 64;
#ifndef GN_ROWS
#define GN_ROWS 8   // rows a statistics thread keeps in flight
#endif

struct GnSrc {
  const half_t* x1;
  const half_t* x2;
  long ld1, ld2;
  int C1, C2;
};

__device__ __forceinline__ uint4 gn_load(const GnSrc& s, long row, int v) {
  const int c = v * 8;
  if (c < s.C1) return *reinterpret_cast<const uint4*>(s.x1 + row * s.ld1 + c);
  return *reinterpret_cast<const uint4*>(s.x2 + row * s.ld2 + (c - s.C1));
}

// Thread (v, rt): owns 8 channels v*8.. (two vec slots when C > 2048) and rows rt, rt+RT, ...
// of its chunk, so a wave's loads sweep contiguous memory when ld == C.
__global__ __launch_bounds__(256) void gn_stats_kernel(GnSrc s, int HW, int G, int rows_per_chunk,
                                                       float* __restrict__ partial) {
  __shared__ float red[256 * 16];
  __shared__ float chan[4096 * 2];
  const int C = s.C1 + s.C2;
  const int nvec = C / 8;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int tid = threadIdx.x;
  const int VT = nvec < 256 ? nvec : 256;
  const int RT = 256 / VT;
  const int v0 = tid % VT, rt = tid / VT;
  const int r_beg = chunk * rows_per_chunk;
  const int r_end = min(HW, r_beg + rows_per_chunk);
  const int nslot = (nvec + 255) / 256;  // 1 or 2
  for (int slot = 0; slot < nslot; ++slot) {
    const int v = v0 + slot * 256;
    float sm[8], sq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sm[e] = sq[e] = 0.f;
    if (rt < RT && v < nvec) {
      // GN_ROWS rows in flight per thread: the kernel is a pure stream and one 16-byte load per thread
      // (~4 MB in flight chip-wide) cannot cover the HBM latency.  Every load is UNCONDITIONAL (row clamped into
      // the chunk, the sample's source resolved once per thread): a load inside `if (row < end)` is compiled as
      // branch + load + s_waitcnt vmcnt(0), i.e. one load in flight however many are written down.
      const int c = v * 8;
      const half_t* src = c < s.C1 ? s.x1 + c : s.x2 + (c - s.C1);
      const long ld = c < s.C1 ? s.ld1 : s.ld2;
      src += (long)b * HW * ld;
      for (int r = r_beg + rt; r < r_end; r += GN_ROWS * RT) {
        Pack16 p[GN_ROWS];
#pragma unroll
        for (int u = 0; u < GN_ROWS; ++u)
          p[u].u = *reinterpret_cast<const uint4*>(src + (long)min(r + u * RT, r_end - 1) * ld);
#pragma unroll
        for (int u = 0; u < GN_ROWS; ++u) {
          const float m = r + u * RT < r_end ? 1.f : 0.f;   // clamped duplicates add nothing
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x = (float)p[u].e[e] * m;
            sm[e] += x;
            sq[e] += x * x;
          }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[tid * 16 + e] = sm[e];
      red[tid * 16 + 8 + e] = sq[e];
    }
    __syncthreads();
    if (rt == 0 && v < nvec) {
      for (int e = 0; e < 8; ++e) {
        float a = 0.f, q = 0.f;
        for (int k = 0; k < RT; ++k) {
          a += red[(k * VT + v0) * 16 + e];
          q += red[(k * VT + v0) * 16 + 8 + e];
        }
        chan[(v * 8 + e) * 2] = a;
        chan[(v * 8 + e) * 2 + 1] = q;
      }
    }
    __syncthreads();
  }
  const int cpg = C / G;
  if (tid < G) {
    float a = 0.f, q = 0.f;
    for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
      a += chan[c * 2];
      q += chan[c * 2 + 1];
    }
    float* out = partial + (((long)b * gridDim.x + chunk) * G + tid) * 2;
    out[0] = a;
    out[1] = q;
  }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(GnSrc s, const half_t* __restrict__ gamma,
                                                       const half_t* __restrict__ beta,
                                                       const float* __restrict__ partial, half_t* __restrict__ y,
                                                       long ldy, int HW, int G, int rows_per_chunk, int act,
                                                       int nchunks, float count, float eps) {
  __shared__ float sc[4096], sh[4096];
  __shared__ float ra[256], rq[256], gmean[GN_MAX_G], grstd[GN_MAX_G];
  const int C = s.C1 + s.C2;
  const int nvec = C / 8;
  const int cpg = C / G;
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  // finalize in the prologue (was a separate 5 us launch): every block reduces the nchunks x G
  // partial sums of ITS sample (<= 64 KB, L2 resident) -- thread (part, g) sums chunks part, part+P, ...
  {
    const int P = 256 / G;
    const int g = tid % G, part = tid / G;
    float a = 0.f, q = 0.f;
    if (part < P) {
      for (int k = part; k < nchunks; k += P) {
        const float2 v = *reinterpret_cast<const float2*>(partial + (((long)b * nchunks + k) * G + g) * 2);
        a += v.x;
        q += v.y;
      }
    }
    ra[tid] = a;
    rq[tid] = q;
    __syncthreads();
    if (tid < G) {
      for (int k = 1; k < P; ++k) {
        a += ra[k * G + tid];
        q += rq[k * G + tid];
      }
      const float mean = a / count;
      gmean[tid] = mean;
      grstd[tid] = rsqrtf(fmaxf(q / count - mean * mean, 0.f) + eps);
    }
    __syncthreads();
  }
  for (int c = tid; c < C; c += 256) {
    const int g = c / cpg;
    const float mean = gmean[g];
    const float rstd = grstd[g];
    const float w = rstd * (float)gamma[c];
    sc[c] = w;
    sh[c] = (float)beta[c] - mean * w;
  }
  __syncthreads();
  const int VT = nvec < 256 ? nvec : 256;
  const int RT = 256 / VT;
  const int v0 = tid % VT, rt = tid / VT;
  if (rt >= RT) return;
  const int r_beg = chunk * rows_per_chunk;
  const int r_end = min(HW, r_beg + rows_per_chunk);
  for (int v = v0; v < nvec; v += 256) {
    float w[8], o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      w[e] = sc[v * 8 + e];
      o[e] = sh[v * 8 + e];
    }
    for (int r = r_beg + rt; r < r_end; r += 4 * RT) {
      Pack16 p[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r + u * RT < r_end) p[u].u = gn_load(s, (long)b * HW + r + u * RT, v);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (r + u * RT >= r_end) break;
        const long row = (long)b * HW + r + u * RT;
        Pack16 q;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t = (float)p[u].e[e] * w[e] + o[e];
          if (act == PFD_ACT_SILU) t = pfd_silu(t);
          q.e[e] = (half_t)t;
        }
        *reinterpret_cast<uint4*>(y + row * ldy + v * 8) = q.u;
      }
    }
  }
}

// ---- GroupNorm apply with the statistics of the PRODUCERS (pfd_groupnorm_pstats_f16, round 4) ----
// The tensors were written by launches that emitted, per 64-row slab and per group of C_src / 32 channels, the sums of the
// f16 values they stored (PfdGemmDesc.gn_out: st[(slab * (C_src / 160) + tile) * 16 + local group], slab = row / 64 over
// the whole batch).  A group of THIS GroupNorm (cpg = (C1 + C2) / G channels of the virtual concat) is a whole number of
// producer groups of one source (checked by the host), so its statistics are sums of producer partials: no pass over the
// tensor.  Same apply loop as gn_apply_kernel.
struct GnPStats {
  const float2* st1;
  const float2* st2;
  int tn1, tn2;      // C_src / 160
  int cpp1, cpp2;    // channels per producer group (C_src / 32)
};

// PAR (the default wherever npg <= 2, round 5): the fold of the partials issues the loads of eight slabs before the first
// add instead of one dependent load per slab (64^2: 8 L2 round trips per block in front of the first row), and the apply
// loop requests eight rows per round trip instead of four (a block's 64 rows at 64^2 x 320: two trips instead of three);
// same values added in the same order -- `selftest --r5` compares the two forms bit for bit (281 checks green on MI355X).
template <bool PAR>
__global__ __launch_bounds__(256) void gn_apply_pstats_kernel(GnSrc s, GnPStats ps, const half_t* __restrict__ gamma,
                                                              const half_t* __restrict__ beta, half_t* __restrict__ y,
                                                              long ldy, int HW, int G, int rows_per_chunk, int act,
                                                              float count, float eps) {
  __shared__ float sc[4096], sh[4096];
  __shared__ float ra[256], rq[256], gmean[GN_MAX_G], grstd[GN_MAX_G];
  const int C = s.C1 + s.C2;
  const int nvec = C / 8;
  const int cpg = C / G;
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  {
    const int nslab = HW / 64;
    const int P = 256 / G;
    const int g = tid % G, part = tid / G;
    float a = 0.f, q = 0.f;
    if (part < P) {
      const int c0 = g * cpg;
      const bool first = c0 < s.C1;
      const float2* st = first ? ps.st1 : ps.st2;
      const int tn = first ? ps.tn1 : ps.tn2, cpp = first ? ps.cpp1 : ps.cpp2;
      const int cl = first ? c0 : c0 - s.C1;
      const int npg = cpg / cpp;                       // producer groups per group of this norm
      if constexpr (PAR) {                             // host: npg <= 2 for both sources
        const int c1 = cl + (npg > 1 ? cpp : 0);
        const int col0 = (cl / 160) * 16 + (cl % 160) / cpp, col1 = (c1 / 160) * 16 + (c1 % 160) / cpp;
        const float w1 = npg > 1 ? 1.f : 0.f;
        for (int k0 = part; k0 < nslab; k0 += 8 * P) {
          float2 v0[8], v1[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int k = min(k0 + u * P, nslab - 1);
            const float2* row = st + ((long)(b * nslab + k) * tn) * 16;
            v0[u] = row[col0];
            v1[u] = row[col1];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {                // slab outer, producer group inner: the order of the plain loop
            const float w = k0 + u * P < nslab ? 1.f : 0.f;
            a += v0[u].x * w;
            q += v0[u].y * w;
            a += v1[u].x * (w * w1);
            q += v1[u].y * (w * w1);
          }
        }
      } else {
        for (int k = part; k < nslab; k += P) {
          const float2* row = st + ((long)(b * nslab + k) * tn) * 16;
          for (int j = 0; j < npg; ++j) {
            const int c = cl + j * cpp;
            const float2 v = row[(c / 160) * 16 + (c % 160) / cpp];
            a += v.x;
            q += v.y;
          }
        }
      }
    }
    ra[tid] = a;
    rq[tid] = q;
    __syncthreads();
    if (tid < G) {
      for (int k = 1; k < P; ++k) {
        a += ra[k * G + tid];
        q += rq[k * G + tid];
      }
      const float mean = a / count;
      gmean[tid] = mean;
      grstd[tid] = rsqrtf(fmaxf(q / count - mean * mean, 0.f) + eps);
    }
    __syncthreads();
  }
  for (int c = tid; c < C; c += 256) {
    const int g = c / cpg;
    const float w = grstd[g] * (float)gamma[c];
    sc[c] = w;
    sh[c] = (float)beta[c] - gmean[g] * w;
  }
  __syncthreads();
  const int VT = nvec < 256 ? nvec : 256;
  const int RT = 256 / VT;
  const int v0 = tid % VT, rt = tid / VT;
  if (rt >= RT) return;
  const int r_beg = chunk * rows_per_chunk;
  const int r_end = min(HW, r_beg + rows_per_chunk);
  for (int v = v0; v < nvec; v += 256) {
    float w[8], o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      w[e] = sc[v * 8 + e];
      o[e] = sh[v * 8 + e];
    }
    constexpr int U = PAR ? 8 : 4;   // rows requested per round trip
    for (int r = r_beg + rt; r < r_end; r += U * RT) {
      Pack16 p[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (r + u * RT < r_end) p[u].u = gn_load(s, (long)b * HW + r + u * RT, v);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (r + u * RT >= r_end) break;
        const long row = (long)b * HW + r + u * RT;
        Pack16 qv;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t = (float)p[u].e[e] * w[e] + o[e];
          if (act == PFD_ACT_SILU) t = pfd_silu(t);
          qv.e[e] = (half_t)t;
        }
        *reinterpret_cast<uint4*>(y + row * ldy + v * 8) = qv.u;
      }
    }
  }
}

// GroupNorm folded into the consumer (pfd_groupnorm_table_f16): instead of writing the normalised tensor, write the
// per-(sample, channel) affine map  y = x * scale + shift  as two planes [B][2][C]  (scale = rstd * gamma, shift = beta - mean * scale; the
// same fp32 expressions, reduced in the same order, as gn_apply_kernel) -- the 3x3 patch convolution applies it
// (+ SiLU) while it stages its input patch (gemm_glds.hip, conv3x3_patch_ws_kernel<true>).
__global__ __launch_bounds__(256) void gn_table_kernel(const half_t* __restrict__ gamma, const half_t* __restrict__ beta,
                                                       const float* __restrict__ partial, float* __restrict__ table,
                                                       int C, int G, int nchunks, float count, float eps) {
  __shared__ float ra[256], rq[256], gmean[GN_MAX_G], grstd[GN_MAX_G];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int cpg = C / G;
  const int P = 256 / G;
  const int g = tid % G, part = tid / G;
  float a = 0.f, q = 0.f;
  if (part < P) {
    for (int k = part; k < nchunks; k += P) {
      const float2 v = *reinterpret_cast<const float2*>(partial + (((long)b * nchunks + k) * G + g) * 2);
      a += v.x;
      q += v.y;
    }
  }
  ra[tid] = a;
  rq[tid] = q;
  __syncthreads();
  if (tid < G) {
    for (int k = 1; k < P; ++k) {
      a += ra[k * G + tid];
      q += rq[k * G + tid];
    }
    const float mean = a / count;
    gmean[tid] = mean;
    grstd[tid] = rsqrtf(fmaxf(q / count - mean * mean, 0.f) + eps);
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    const int gg = c / cpg;
    const float w = grstd[gg] * (float)gamma[c];
    table[(long)b * 2 * C + c] = w;                                  // scale plane
    table[(long)b * 2 * C + C + c] = (float)beta[c] - gmean[gg] * w;    // shift plane
  }
}

// ---- GroupNorm, small-slab form: one block per (sample, group) keeps the group's HW x C/G slab in
// registers (<= 32 chunks of 4 halves per thread), so statistics + normalise + activation are ONE launch
// and the input is read once (4 B/element).  Serves the 8^2 / 16^2 / 32^2 UNet levels, where the
// two-launch form is bound by launch latency (1280 @ 8^2: 10 us for 1.3 MB).  Needs (C/G) % 4 == 0.
// 1024 threads per block since round 5 (four times fewer before): the launch is one round trip of the slab, so its time is set
// by the loads a block has in flight.  The fused split-K reduction + GroupNorm of gemm_glds.hip (splitk_reduce_gnorm_kernel)
// uses the SAME thread -> chunk mapping and the same order of the adds: the two give the same bits.
#ifndef PFD_GN_THREADS
#define PFD_GN_THREADS 1024   // (-DPFD_GN_THREADS=256: A/B builds)
#endif
constexpr int GNS_T = PFD_GN_THREADS;
constexpr int GNS_MAX = 8192 / GNS_T;

__global__ __launch_bounds__(GNS_T) void gn_small_kernel(GnSrc s, const half_t* __restrict__ gamma,
                                                       const half_t* __restrict__ beta, half_t* __restrict__ y,
                                                       long ldy, int HW, int G, int act, float eps) {
  __shared__ float red[2 * GNS_T / 64];
  const int C = s.C1 + s.C2;
  const int cpg = C / G, cpr = cpg / 4;
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int total = HW * cpr;
  uint2 v[GNS_MAX];
  float sm = 0.f, sq = 0.f;
#pragma unroll
  for (int k = 0; k < GNS_MAX; ++k) {
    const int idx = tid + GNS_T * k;
    v[k] = make_uint2(0, 0);
    if (idx < total) {
      const int r = idx / cpr, c = g * cpg + (idx - r * cpr) * 4;
      const long row = (long)b * HW + r;
      v[k] = c < s.C1 ? *reinterpret_cast<const uint2*>(s.x1 + row * s.ld1 + c)
                      : *reinterpret_cast<const uint2*>(s.x2 + row * s.ld2 + (c - s.C1));
    }
  }
#pragma unroll
  for (int k = 0; k < GNS_MAX; ++k) {   // padding slots hold zeros: they add nothing to either sum
    Pack8 p;
    p.u = v[k];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = (float)p.e[e];
      sm += x;
      sq += x * x;
    }
  }
  sm = wave_sum(sm);
  sq = wave_sum(sq);
  constexpr int NWV = GNS_T / 64;
  if (lane == 0) {
    red[wave] = sm;
    red[NWV + wave] = sq;
  }
  __syncthreads();
  const float count = (float)HW * (float)cpg;
  float ts = red[0], tq = red[NWV];
#pragma unroll
  for (int w = 1; w < NWV; ++w) {
    ts += red[w];
    tq += red[NWV + w];
  }
  const float mean = ts / count;
  const float rstd = rsqrtf(fmaxf(tq / count - mean * mean, 0.f) + eps);
#pragma unroll
  for (int k = 0; k < GNS_MAX; ++k) {
    const int idx = tid + GNS_T * k;
    if (idx < total) {
      const int r = idx / cpr, c = g * cpg + (idx - r * cpr) * 4;
      Pack8 p, ga, be, o;
      p.u = v[k];
      ga.u = *reinterpret_cast<const uint2*>(gamma + c);
      be.u = *reinterpret_cast<const uint2*>(beta + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float w = rstd * (float)ga.e[e];
        float t = (float)p.e[e] * w + ((float)be.e[e] - mean * w);
        if (act == PFD_ACT_SILU) t = pfd_silu(t);
        o.e[e] = (half_t)t;
      }
      *reinterpret_cast<uint2*>(y + ((long)b * HW + r) * ldy + c) = o.u;
    }
  }
}

// ---- LayerNorm: one wave per row, the row lives in registers (<= 8 vecs of 8 per lane) ----
constexpr int LN_MAXV = 8;

template <bool GATHER, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x, long ldx,
                                                        const half_t* __restrict__ gamma,
                                                        const half_t* __restrict__ beta, half_t* __restrict__ y,
                                                        long ldy, int M, int C, float eps, int B, int H, int W) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nvec = C / 8;
  int gb = 0, gy = 0, gx = 0, Cq = 1;
  if constexpr (GATHER) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    gb = row / (Ho * Wo);
    const int rem = row - gb * Ho * Wo;
    gy = rem / Wo;
    gx = rem - gy * Wo;
    Cq = C / 4;
  }
  // all loads of the row first, unconditional (clamped address; lanes past the row / the image are masked when the
  // values are used): a load inside an `if` is compiled as branch + load + s_waitcnt vmcnt(0), one load in flight
  Pack16 p[NV];
  bool ok[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int v = lane + 64 * k;
    const int vc = min(v, nvec - 1);
    const half_t* src = x + (long)row * ldx + vc * 8;
    ok[k] = v < nvec;
    if constexpr (GATHER) {
      const int c = vc * 8;
      const int part = c / Cq;
      const int iy = 2 * gy + (part & 1), ix = 2 * gx + (part >> 1);
      ok[k] = ok[k] && iy < H && ix < W;
      src = x + (((long)gb * H + min(iy, H - 1)) * W + min(ix, W - 1)) * ldx + (c - part * Cq);
    }
    p[k].u = *reinterpret_cast<const uint4*>(src);
  }
  Pack16 g[NV], bt[NV];   // requested before the reductions (unconditional, clamped), used after them
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int vc = min(lane + 64 * k, nvec - 1);
    g[k].u = *reinterpret_cast<const uint4*>(gamma + vc * 8);
    bt[k].u = *reinterpret_cast<const uint4*>(beta + vc * 8);
  }
  float vals[NV][8];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      vals[k][e] = ok[k] ? (float)p[k].e[e] : 0.f;   // out-of-image taps of the PatchMerging gather read zero
      sum += vals[k][e];
    }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (lane + 64 * k < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = vals[k][e] - mean;
        sq += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int v = lane + 64 * k;
    if (v < nvec) {
      Pack16 o;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        o.e[e] = (half_t)((vals[k][e] - mean) * rstd * (float)g[k].e[e] + (float)bt[k].e[e]);
      *reinterpret_cast<uint4*>(y + (long)row * ldy + v * 8) = o.u;
    }
  }
}

// LayerNorm of many short rows (UNet transformer blocks: C = 320 / 640 / 1280, M = 2 K .. 32 K tokens): each
// wave normalises ROWS consecutive rows and issues all their loads up front -- with one row per wave the
// kernel had a single 16-byte load per lane in flight and ran at a third of the HBM rate.
template <int NV, int ROWS>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const half_t* __restrict__ x, long ldx,
                                                             const half_t* __restrict__ gamma,
                                                             const half_t* __restrict__ beta, half_t* __restrict__ y,
                                                             long ldy, int M, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
  if (row0 >= M) return;
  const int nvec = C / 8;
  // every load below is unconditional (index clamped, result zeroed by a select): a load inside an `if` is compiled
  // as branch + load + s_waitcnt vmcnt(0) and the ROWS x NV loads would go out one at a time
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  Pack16 p[ROWS][NV];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int v = lane + 64 * k;
      const uint4 t = *reinterpret_cast<const uint4*>(x + (long)min(row0 + r, M - 1) * ldx + min(v, nvec - 1) * 8);
      p[r][k].u = (v < nvec && row0 + r < M) ? t : zero4;
    }
  Pack16 g[NV], bt[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int v = min(lane + 64 * k, nvec - 1);   // lanes past the row hold a copy they never store
    g[k].u = *reinterpret_cast<const uint4*>(gamma + v * 8);
    bt[k].u = *reinterpret_cast<const uint4*>(beta + v * 8);
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    if (row0 + r >= M) break;
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += (float)p[r][k].e[e];   // padding lanes hold zeros
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (lane + 64 * k < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = (float)p[r][k].e[e] - mean;
          sq += d * d;
        }
      }
    const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int v = lane + 64 * k;
      if (v < nvec) {
        Pack16 o;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          o.e[e] = (half_t)(((float)p[r][k].e[e] - mean) * rstd * (float)g[k].e[e] + (float)bt[k].e[e]);
        *reinterpret_cast<uint4*>(y + (long)(row0 + r) * ldy + v * 8) = o.u;
      }
    }
  }
}

// ---- scaled row softmax: one block per row ----
constexpr int SM_MAXV = 8;  // N <= 256*8*8 = 16384

__global__ __launch_bounds__(256) void softmax_rows_kernel(const half_t* __restrict__ x, long ldx,
                                                           half_t* __restrict__ y, long ldy, int N, float scale) {
  __shared__ float red[8];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nvec = N / 8;
  float vals[SM_MAXV][8];
  float mx = -INFINITY;
  Pack16 p[SM_MAXV];
#pragma unroll
  for (int k = 0; k < SM_MAXV; ++k)   // all loads first, unconditional (clamped): see layernorm_rows_kernel
    p[k].u = *reinterpret_cast<const uint4*>(x + (long)row * ldx + min(tid + 256 * k, nvec - 1) * 8);
#pragma unroll
  for (int k = 0; k < SM_MAXV; ++k) {
    if (tid + 256 * k < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        vals[k][e] = (float)p[k].e[e] * scale;
        mx = fmaxf(mx, vals[k][e]);
      }
    }
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < SM_MAXV; ++k) {
    if (tid + 256 * k < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        vals[k][e] = __expf(vals[k][e] - mx);
        sum += vals[k][e];
      }
    }
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
#pragma unroll
  for (int k = 0; k < SM_MAXV; ++k) {
    const int v = tid + 256 * k;
    if (v < nvec) {
      Pack16 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o.e[e] = (half_t)(vals[k][e] * inv);
      *reinterpret_cast<uint4*>(y + (long)row * ldy + v * 8) = o.u;
    }
  }
}

// rows longer than the register-resident form (VAE mid attention of 1152..1536-wide outputs: N up to
// 36 864 keys): three streaming passes over the row (max, sum, write); the row (<= 72 KB) stays in L2.
__global__ __launch_bounds__(256) void softmax_rows_long_kernel(const half_t* __restrict__ x, long ldx,
                                                                half_t* __restrict__ y, long ldy, int N, float scale) {
  __shared__ float red[8];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nvec = N / 8;
  const half_t* xr = x + (long)row * ldx;
  float mx = -INFINITY;
  for (int v = tid; v < nvec; v += 256) {
    Pack16 p;
    p.u = *reinterpret_cast<const uint4*>(xr + v * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)p.e[e] * scale);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int v = tid; v < nvec; v += 256) {
    Pack16 p;
    p.u = *reinterpret_cast<const uint4*>(xr + v * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += __expf((float)p.e[e] * scale - mx);
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
  for (int v = tid; v < nvec; v += 256) {   // in-place safe: each thread rewrites only what it just read
    Pack16 p, o;
    p.u = *reinterpret_cast<const uint4*>(xr + v * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) o.e[e] = (half_t)(__expf((float)p.e[e] * scale - mx) * inv);
    *reinterpret_cast<uint4*>(y + (long)row * ldy + v * 8) = o.u;
  }
}

}  // namespace

// chunking of the two-launch form: ~512 blocks over the chip, at least 4 row sweeps per block (every apply block
// re-reduces its sample's nchunks x G partials, so nchunks stays moderate)
static void gn_chunks(int B, int C, int HW, int* nchunks_out, int* rpc_out) {
  const int nvec = C / 8;
  const int RT = nvec < 256 ? 256 / nvec : 1;
  constexpr int target_blocks = 512;   // (1024 / 2048 measured slower, profiles/r04_end_to_end_ab.log; 256: +0.2 %, round 5)
  int nchunks = (target_blocks + B - 1) / B;
  const int max_by_rows = (HW + RT * 4 - 1) / (RT * 4);
  if (nchunks > max_by_rows) nchunks = max_by_rows;
  if (nchunks > GN_MAX_CHUNKS) nchunks = GN_MAX_CHUNKS;
  if (nchunks < 1) nchunks = 1;
  const int rpc = (HW + nchunks - 1) / nchunks;
  *nchunks_out = (HW + rpc - 1) / rpc;
  *rpc_out = rpc;
}

// 1 when pfd_groupnorm_f16 would take the single-launch small-slab form (which needs no separate statistics)
static bool gn_is_small(int B, int C, int HW, int G) {
  const int cpg = C / G;
  return cpg % 4 == 0 && cpg >= 32 && (long)HW * (cpg / 4) <= GNS_T * GNS_MAX && (long)B * G >= 128;
}

extern "C" size_t pfd_groupnorm_ws_bytes(int32_t B, int32_t C, int32_t HW) {
  (void)C;
  (void)HW;
  return (size_t)B * (GN_MAX_CHUNKS + 1) * GN_MAX_G * 2 * sizeof(float);
}

extern "C" int pfd_groupnorm_f16(const void* x1, int32_t C1, int64_t ldx1, const void* x2, int32_t C2,
                                 int64_t ldx2, const void* gamma, const void* beta, void* y, int64_t ldy,
                                 int32_t B, int32_t HW, int32_t G, float eps, int32_t act, void* ws,
                                 size_t ws_bytes, pfd_stream_t stream) {
  if (!x1 || !gamma || !beta || !y || !ws) return PFD_EINVAL;
  if (C2 > 0 && !x2) return PFD_EINVAL;
  if (C2 < 0 || C1 <= 0 || B <= 0 || HW <= 0 || G <= 0 || G > GN_MAX_G) return PFD_EINVAL;
  const int C = C1 + C2;
  if ((C1 & 7) || (C2 & 7) || (C % G) || C > 4096) return PFD_ESHAPE;
  if ((ldx1 & 7) || (ldx2 & 7) || (ldy & 7)) return PFD_EINVAL;
  if (act != PFD_ACT_NONE && act != PFD_ACT_SILU) return PFD_EINVAL;
  if (ws_bytes < pfd_groupnorm_ws_bytes(B, C, HW)) return PFD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  GnSrc src{(const half_t*)x1, (const half_t*)x2, ldx1, ldx2, C1, C2};
  const bool prof = pfd_prof_on();
  if (gn_is_small(B, C, HW, G)) {  // narrower groups: 40-byte row segments, the two-launch form wins (640 @ 32^2: 15 vs 17 us)
    if (prof) pfd_prof_begin(10, 8.0 * B * HW * C, 4.0 * B * HW * C, s);  // 2B read + 2B write
    // (round 5: a form with incremental indices and gamma / beta in LDS -- 5419 -> 3172 instructions -- measured -0.04 %
    //  per batch, profiles/r05_e2e_ab_candidates.log: not kept; the fused split-K reduction + GroupNorm of gemm_glds.hip
    //  takes most of these launches instead)
    hipLaunchKernelGGL(gn_small_kernel, dim3(G, B), dim3(GNS_T), 0, s, src, (const half_t*)gamma, (const half_t*)beta,
                       (half_t*)y, (long)ldy, HW, G, act, eps);
    if (prof) pfd_prof_end(s);
    return pfd_check_launch("pfd_groupnorm_f16(small)");
  }
  int nchunks, rpc;
  gn_chunks(B, C, HW, &nchunks, &rpc);
  float* partial = (float*)ws;
  if (prof) pfd_prof_begin(10, 8.0 * B * HW * C, 6.0 * B * HW * C, s);  // 2B stats read + 2B read + 2B write
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunks, B), dim3(256), 0, s, src, HW, G, rpc, partial);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(nchunks, B), dim3(256), 0, s, src, (const half_t*)gamma,
                     (const half_t*)beta, partial, (half_t*)y, (long)ldy, HW, G, rpc, act, nchunks,
                     (float)HW * (float)(C / G), eps);
  if (prof) pfd_prof_end(s);
  return pfd_check_launch("pfd_groupnorm_f16");
}

extern "C" int32_t pfd_groupnorm_takes_pstats(int32_t B, int32_t C1, int32_t C2, int32_t HW, int32_t G) {
  if (B <= 0 || C1 <= 0 || C2 < 0 || HW <= 0 || G <= 0 || G > GN_MAX_G) return 0;
  const int C = C1 + C2;
  if ((C % G) || C > 4096 || (HW % 64) || (C1 % 160) || (C2 % 160) || (C1 % 32) || (C2 % 32)) return 0;
  if (gn_is_small(B, C, HW, G)) return 0;
  const int cpg = C / G, cpp1 = C1 / 32, cpp2 = C2 ? C2 / 32 : cpp1;
  if (cpp1 < 8 || (160 % cpp1) || (cpg % cpp1) || (C1 % cpg)) return 0;       // groups of this norm = whole producer groups
  if (C2 && (cpp2 < 8 || (160 % cpp2) || (cpg % cpp2))) return 0;
  return 1;
}

extern "C" int pfd_groupnorm_pstats_f16(const void* x1, int32_t C1, int64_t ldx1, const void* st1, const void* x2,
                                        int32_t C2, int64_t ldx2, const void* st2, const void* gamma, const void* beta,
                                        void* y, int64_t ldy, int32_t B, int32_t HW, int32_t G, float eps, int32_t act,
                                        pfd_stream_t stream) {
  if (!x1 || !st1 || !gamma || !beta || !y) return PFD_EINVAL;
  if (C2 > 0 && (!x2 || !st2)) return PFD_EINVAL;
  if (act != PFD_ACT_NONE && act != PFD_ACT_SILU) return PFD_EINVAL;
  if ((ldx1 & 7) || (ldx2 & 7) || (ldy & 7) || (reinterpret_cast<uintptr_t>(st1) & 7) || (reinterpret_cast<uintptr_t>(st2) & 7))
    return PFD_EINVAL;
  if (!pfd_groupnorm_takes_pstats(B, C1, C2, HW, G)) return PFD_ESHAPE;
  hipStream_t s = (hipStream_t)stream;
  const int C = C1 + C2;
  GnSrc src{(const half_t*)x1, (const half_t*)x2, ldx1, ldx2, C1, C2};
  GnPStats ps{(const float2*)st1, (const float2*)st2, C1 / 160, C2 ? C2 / 160 : 0, C1 / 32, C2 ? C2 / 32 : C1 / 32};
  int nchunks, rpc;
  gn_chunks(B, C, HW, &nchunks, &rpc);
  PfdProfScope prof_scope(10, 8.0 * B * HW * C, 4.0 * B * HW * C, s);   // 2 B read + 2 B write
  const int cpg = C / G;
  // grouped partial loads where a group of this norm is at most two producer groups per source (every UNet shape but the
  // 3-source-group concats); adopted in round 5 at -0.3 % per batch (profiles/r05_e2e_ab_candidates.log), same bits
  const bool par = cpg / ps.cpp1 <= 2 && cpg / ps.cpp2 <= 2;
  if (par)
    hipLaunchKernelGGL(gn_apply_pstats_kernel<true>, dim3(nchunks, B), dim3(256), 0, s, src, ps, (const half_t*)gamma,
                       (const half_t*)beta, (half_t*)y, (long)ldy, HW, G, rpc, act, (float)HW * (float)(C / G), eps);
  else
    hipLaunchKernelGGL(gn_apply_pstats_kernel<false>, dim3(nchunks, B), dim3(256), 0, s, src, ps, (const half_t*)gamma,
                       (const half_t*)beta, (half_t*)y, (long)ldy, HW, G, rpc, act, (float)HW * (float)(C / G), eps);
  return pfd_check_launch("pfd_groupnorm_pstats_f16");
}

extern "C" int pfd_groupnorm_table_f16(const void* x1, int32_t C1, int64_t ldx1, const void* x2, int32_t C2,
                                       int64_t ldx2, const void* gamma, const void* beta, void* table, int32_t B,
                                       int32_t HW, int32_t G, float eps, void* ws, size_t ws_bytes,
                                       pfd_stream_t stream) {
  if (!x1 || !gamma || !beta || !table || !ws) return PFD_EINVAL;
  if (C2 > 0 && !x2) return PFD_EINVAL;
  if (C2 < 0 || C1 <= 0 || B <= 0 || HW <= 0 || G <= 0 || G > GN_MAX_G) return PFD_EINVAL;
  const int C = C1 + C2;
  if ((C1 & 7) || (C2 & 7) || (C % G) || C > 4096) return PFD_ESHAPE;
  if ((ldx1 & 7) || (ldx2 & 7) || (reinterpret_cast<uintptr_t>(table) & 15)) return PFD_EINVAL;
  if (ws_bytes < pfd_groupnorm_ws_bytes(B, C, HW)) return PFD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  GnSrc src{(const half_t*)x1, (const half_t*)x2, ldx1, ldx2, C1, C2};
  int nchunks, rpc;
  gn_chunks(B, C, HW, &nchunks, &rpc);
  const bool prof = pfd_prof_on();
  if (prof) pfd_prof_begin(10, 4.0 * B * HW * C, 2.0 * B * HW * C, s);   // one read of the input
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunks, B), dim3(256), 0, s, src, HW, G, rpc, (float*)ws);
  hipLaunchKernelGGL(gn_table_kernel, dim3(B), dim3(256), 0, s, (const half_t*)gamma, (const half_t*)beta,
                     (const float*)ws, (float*)table, C, G, nchunks, (float)HW * (float)(C / G), eps);
  if (prof) pfd_prof_end(s);
  return pfd_check_launch("pfd_groupnorm_table_f16");
}

extern "C" int pfd_layernorm_f16(const void* x, int64_t ldx, const void* gamma, const void* beta, void* y,
                                 int64_t ldy, int32_t M, int32_t C, float eps, int32_t gather4, int32_t B,
                                 int32_t H, int32_t W, pfd_stream_t stream) {
  if (!x || !gamma || !beta || !y || M <= 0 || C <= 0) return PFD_EINVAL;
  if ((C & 7) || C > 64 * 8 * LN_MAXV) return PFD_ESHAPE;
  if ((ldx & 7) || (ldy & 7)) return PFD_EINVAL;
  if (gather4) {
    if ((C % 32) || B <= 0 || H <= 0 || W <= 0) return PFD_EINVAL;
    if ((long)B * ((H + 1) / 2) * ((W + 1) / 2) != M) return PFD_EINVAL;
  }
  PfdProfScope prof_scope(11, 0.0, 4.0 * M * C, (hipStream_t)stream);   // read + write once
  if (!gather4 && C <= 1536 && M >= 8192) {  // below that one row per wave gives more blocks than CUs (2048 x 1280: 7.3 vs 11.5 us)
    constexpr int ROWS = 4;
    const dim3 grid((M + 4 * ROWS - 1) / (4 * ROWS));
    const int nv = (C / 8 + 63) / 64;
#define PFD_LN_ROWS(NV)                                                                                         \
  hipLaunchKernelGGL((layernorm_rows_kernel<NV, ROWS>), grid, dim3(256), 0, (hipStream_t)stream, (const half_t*)x, \
                     (long)ldx, (const half_t*)gamma, (const half_t*)beta, (half_t*)y, (long)ldy, M, C, eps)
    if (nv == 1) PFD_LN_ROWS(1);
    else if (nv == 2) PFD_LN_ROWS(2);
    else PFD_LN_ROWS(3);
#undef PFD_LN_ROWS
    return pfd_check_launch("pfd_layernorm_f16(rows)");
  }
  // NV = 16-byte vectors per lane (64 lanes per row), rounded up to an instantiated count
  const int nv = (C / 8 + 63) / 64;
#define PFD_LN(G, NV)                                                                                              \
  hipLaunchKernelGGL((layernorm_kernel<G, NV>), dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream,              \
                     (const half_t*)x, (long)ldx, (const half_t*)gamma, (const half_t*)beta, (half_t*)y, (long)ldy, \
                     M, C, eps, B, H, W)
#define PFD_LN_NV(G)                                                                                               \
  do {                                                                                                             \
    if (nv <= 1) PFD_LN(G, 1);                                                                                     \
    else if (nv <= 2) PFD_LN(G, 2);                                                                                \
    else if (nv <= 3) PFD_LN(G, 3);                                                                                \
    else if (nv <= 4) PFD_LN(G, 4);                                                                                \
    else if (nv <= 6) PFD_LN(G, 6);                                                                                \
    else PFD_LN(G, 8);                                                                                             \
  } while (0)
  if (gather4) PFD_LN_NV(true);
  else PFD_LN_NV(false);
#undef PFD_LN_NV
#undef PFD_LN
  return pfd_check_launch("pfd_layernorm_f16");
}

// Partial row sums for the LayerNorm folded into the consumer GEMM (PfdGemmDesc.ln_stats).  Fallback producer only
// (tensors that were not written by a wide-tile GEMM epilogue: the `x + bias` rows of the zero-context shortcut, split-K
// outputs).  SAME summation order as the statistics-emitting store pass of the GEMM epilogue (gemm_glds.hip,
// epilogue_store): four lanes per (row, 160-column slice), lane k adds chunks k, k + 4, ..., k + 16 element by element,
// then xor-1 and xor-2 exchanges -- so a row's statistics do not depend on which of the two wrote them (the
// zero-context shortcut stays bit-identical to the full computation).
__global__ __launch_bounds__(256) void ln_rowstats_kernel(const half_t* __restrict__ x, long ldx, int M, int P,
                                                          float2* __restrict__ out) {
  const long grp = ((long)blockIdx.x * 256 + threadIdx.x) >> 2;
  const int k = threadIdx.x & 3;
  const long ngrp = (long)M * P;
  const long g = min(grp, ngrp - 1);
  const int m = (int)(g / P), p = (int)(g - (long)m * P);
  const half_t* src = x + (long)m * ldx + p * 160;
  Pack16 v[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) v[j].u = *reinterpret_cast<const uint4*>(src + (k + 4 * j) * 8);   // all loads first
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float f = (float)v[j].e[e];
      s += f;
      q = fmaf(f, f, q);
    }
  s += __shfl_xor(s, 1, 64);
  q += __shfl_xor(q, 1, 64);
  s += __shfl_xor(s, 2, 64);
  q += __shfl_xor(q, 2, 64);
  if (k == 0 && grp < ngrp) out[grp] = make_float2(s, q);
}

int pfd_ln_rowstats_launch(const half_t* x, long ldx, int M, int C, float* out, hipStream_t s, bool prof) {
  const int P = C / 160;
  const long ngrp = (long)M * P;
  const bool on = prof && pfd_prof_on();   // (event pairs do not nest: the GEMM launchers call this inside their own)
  if (on) pfd_prof_begin(11, 0.0, 2.0 * M * C, s);
  hipLaunchKernelGGL(ln_rowstats_kernel, dim3((unsigned)((ngrp + 63) / 64)), dim3(256), 0, s, x, ldx, M, P,
                     reinterpret_cast<float2*>(out));
  if (on) pfd_prof_end(s);
  return pfd_check_launch("pfd_ln_rowstats_f16");
}

extern "C" int pfd_ln_rowstats_f16(const void* x, int64_t ldx, int32_t M, int32_t C, void* out, pfd_stream_t stream) {
  if (!x || !out || M <= 0 || C <= 0) return PFD_EINVAL;
  if ((C % 160) || C > 1280) return PFD_ESHAPE;
  if ((ldx & 7) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 7)) return PFD_EINVAL;
  return pfd_ln_rowstats_launch((const half_t*)x, (long)ldx, M, C, (float*)out, (hipStream_t)stream, true);
}

extern "C" int pfd_softmax_rows_f16(const void* x, int64_t ldx, void* y, int64_t ldy, int32_t R, int32_t N,
                                    float scale, pfd_stream_t stream) {
  if (!x || !y || R <= 0 || N <= 0) return PFD_EINVAL;
  if (N & 7) return PFD_ESHAPE;
  if ((ldx & 7) || (ldy & 7)) return PFD_EINVAL;
  PfdProfScope prof_scope(15, 0.0, 4.0 * R * N, (hipStream_t)stream);
  if (N > 256 * 8 * SM_MAXV)
    hipLaunchKernelGGL(softmax_rows_long_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, (const half_t*)x,
                       (long)ldx, (half_t*)y, (long)ldy, N, scale);
  else
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, (const half_t*)x,
                       (long)ldx, (half_t*)y, (long)ldy, N, scale);
  return pfd_check_launch("pfd_softmax_rows_f16");
}
