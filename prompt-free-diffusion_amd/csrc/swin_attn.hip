// pfd_swin_window_attention_f16: Swin-L (shifted) window attention core, one workgroup per
// (window, head), v_mfma_f32_16x16x32_f16 (head_dim 32 == one MFMA K step).
//
// Everything the reference does with tensor copies is index math here: zero-pad to a multiple
// of 12 (padded tokens carry q|k|v = qkv bias), cyclic roll by -shift, window partition, the
// 9-region shift mask (-100), the [529, nH] relative-position-bias gather, window reverse,
// roll back and crop.  Same transposed formulation as attention.hip:
//   S^T[kv, q] = K Q^T (9x9 tiles of 16x16), softmax over kv = over registers + 2 shuffles,
//   O^T[d, q]  = V^T P^T with the k-slot order pi(g, j) = 16(j>>2) + 4g + (j&3).
#include "pfd_common.h"

namespace {

constexpr int WS = 12;
constexpr int NT = WS * WS;      // 144 tokens per window
constexpr int HD = 32;           // head dim
constexpr int QK_LD = HD + 8;    // halfs
constexpr int VT_LD = 160 + 4;   // halfs; keys padded to 160 = 5 MFMA K steps

struct SwinParams {
  const half_t* qkv;
  const half_t* qkv_bias;
  const half_t* rpb;
  half_t* out;
  int B, H, W, C, nH, shift;
  int Hp, Wp, nWx, nWy;
  float scale;
};

// window-local token n -> row of the [B*H*W, .] token matrix, or -1 for a padded position;
// also the shift-mask region id of that token.
__device__ __forceinline__ int token_row(const SwinParams& p, int b, int wy, int wx, int n, int& region) {
  const int iy = n / WS, ix = n - iy * WS;
  const int py = wy * WS + iy, px = wx * WS + ix;  // position in the rolled, padded frame
  const int rh = py < p.Hp - WS ? 0 : (py < p.Hp - p.shift ? 1 : 2);
  const int rw = px < p.Wp - WS ? 0 : (px < p.Wp - p.shift ? 1 : 2);
  region = rh * 3 + rw;
  int oy = py + p.shift, ox = px + p.shift;  // roll(-shift): rolled[i] = x[(i + shift) mod n]
  if (oy >= p.Hp) oy -= p.Hp;
  if (ox >= p.Wp) ox -= p.Wp;
  if (oy >= p.H || ox >= p.W) return -1;
  return (b * p.H + oy) * p.W + ox;
}

__global__ __launch_bounds__(192) void swin_attn_kernel(const SwinParams p) {
  __shared__ __attribute__((aligned(16))) half_t Qs[NT * QK_LD];
  __shared__ __attribute__((aligned(16))) half_t Ks[NT * QK_LD];
  __shared__ __attribute__((aligned(16))) half_t Vts[HD * VT_LD];
  __shared__ int rows[NT];
  __shared__ int regs[NT];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int win = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  const int wy = win / p.nWx, wx = win - wy * p.nWx;
  const int C3 = 3 * p.C;

  for (int i = tid; i < HD * VT_LD / 2; i += 192) reinterpret_cast<uint32_t*>(Vts)[i] = 0u;
  for (int n = tid; n < NT; n += 192) {
    int reg;
    rows[n] = token_row(p, b, wy, wx, n, reg);
    regs[n] = reg;
  }
  __syncthreads();

  // Q, K: [144][32] row-major
  for (int it = tid; it < NT * 4; it += 192) {
    const int n = it >> 2, cc = it & 3;
    const int r = rows[n];
    const half_t* src = r >= 0 ? p.qkv + (long)r * C3 : p.qkv_bias;
    const int col = head * HD + cc * 8;
    *reinterpret_cast<uint4*>(Qs + n * QK_LD + cc * 8) = *reinterpret_cast<const uint4*>(src + col);
    *reinterpret_cast<uint4*>(Ks + n * QK_LD + cc * 8) = *reinterpret_cast<const uint4*>(src + p.C + col);
  }
  // V^T: [32][160], written as key pairs
  for (int it = tid; it < (NT / 2) * 4; it += 192) {
    const int pp = it >> 2, cc = it & 3;
    const int r0 = rows[2 * pp], r1 = rows[2 * pp + 1];
    const int col = 2 * p.C + head * HD + cc * 8;
    Pack16 a, c;
    a.u = *reinterpret_cast<const uint4*>((r0 >= 0 ? p.qkv + (long)r0 * C3 : p.qkv_bias) + col);
    c.u = *reinterpret_cast<const uint4*>((r1 >= 0 ? p.qkv + (long)r1 * C3 : p.qkv_bias) + col);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      half2_t v;
      v[0] = a.e[e];
      v[1] = c.e[e];
      *reinterpret_cast<half2_t*>(Vts + (cc * 8 + e) * VT_LD + 2 * pp) = v;
    }
  }
  __syncthreads();

  for (int qi = 0; qi < 3; ++qi) {
    const int qt = wave * 3 + qi;
    const int qn = qt * 16 + l15;  // this lane's query token
    const half8_t qf = *reinterpret_cast<const half8_t*>(Qs + qn * QK_LD + g * 8);
    float4_t s[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const half8_t kf = *reinterpret_cast<const half8_t*>(Ks + (t * 16 + l15) * QK_LD + g * 8);
      float4_t z = {0.f, 0.f, 0.f, 0.f};
      s[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf, z, 0, 0, 0);
    }
    // scale, relative position bias, shift mask
    const int qy = qn / WS, qx = qn - qy * WS;
    const int qreg = regs[qn];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kn = t * 16 + 4 * g + r;
        const int ky = kn / WS, kx = kn - ky * WS;
        const int idx = (qy - ky + WS - 1) * (2 * WS - 1) + (qx - kx + WS - 1);
        float v = s[t][r] * p.scale + (float)p.rpb[idx * p.nH + head];
        if (p.shift > 0 && regs[kn] != qreg) v += -100.0f;
        s[t][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(s[t][r] - mx);
        s[t][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    // O^T = V^T P^T, 5 K steps of 32 keys (the 10th key tile is zero padding)
    float4_t o[2];
    o[0] = (float4_t){0.f, 0.f, 0.f, 0.f};
    o[1] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      half8_t pf;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        pf[j] = (half_t)(s[2 * c][j] * inv);
        pf[4 + j] = (2 * c + 1 < 9) ? (half_t)(s[(2 * c + 1) % 9][j] * inv) : (half_t)0.f;
      }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const half_t* vp = Vts + (dt * 16 + l15) * VT_LD + c * 32 + 4 * g;
        const half4_t lo4 = *reinterpret_cast<const half4_t*>(vp);
        const half4_t hi4 = *reinterpret_cast<const half4_t*>(vp + 16);
        half8_t vf;
        vf[0] = lo4[0]; vf[1] = lo4[1]; vf[2] = lo4[2]; vf[3] = lo4[3];
        vf[4] = hi4[0]; vf[5] = hi4[1]; vf[6] = hi4[2]; vf[7] = hi4[3];
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[dt], 0, 0, 0);
      }
    }
    const int orow = rows[qn];
    if (orow >= 0) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        half4_t ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = (half_t)o[dt][e];
        *reinterpret_cast<half4_t*>(p.out + (long)orow * p.C + head * HD + dt * 16 + 4 * g) = ov;
      }
    }
  }
}

}  // namespace

extern "C" int pfd_swin_window_attention_f16(const PfdSwinAttnDesc* d, pfd_stream_t stream) {
  if (!d || !d->qkv || !d->qkv_bias || !d->rpb || !d->out) return PFD_EINVAL;
  if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->nH <= 0) return PFD_EINVAL;
  if (d->ws != WS || d->C != d->nH * HD) return PFD_ESHAPE;
  if (d->shift < 0 || d->shift >= WS) return PFD_EINVAL;
  SwinParams p;
  p.qkv = (const half_t*)d->qkv; p.qkv_bias = (const half_t*)d->qkv_bias; p.rpb = (const half_t*)d->rpb;
  p.out = (half_t*)d->out;
  p.B = d->B; p.H = d->H; p.W = d->W; p.C = d->C; p.nH = d->nH; p.shift = d->shift;
  p.Hp = (d->H + WS - 1) / WS * WS;
  p.Wp = (d->W + WS - 1) / WS * WS;
  p.nWy = p.Hp / WS; p.nWx = p.Wp / WS;
  p.scale = d->scale;
  dim3 grid(p.nWx * p.nWy, p.nH, p.B);
  PfdProfScope prof_scope(9, 4.0 * p.B * p.nH * (double)(p.nWx * p.nWy) * 144 * 144 * 32, 0.0, (hipStream_t)stream);
  hipLaunchKernelGGL(swin_attn_kernel, grid, dim3(192), 0, (hipStream_t)stream, p);
  return pfd_check_launch("pfd_swin_window_attention_f16");
}
