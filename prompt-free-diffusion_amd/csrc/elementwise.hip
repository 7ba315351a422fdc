// Boundary and elementwise kernels: layout conversion at the NCHW <-> NHWC boundary, im2col for
// the few narrow-channel convolutions, timestep embedding, the fused CFG + DDIM update, adds.
// All HBM-bound; small tensors (latents are 4 channels), so simplicity over peak bandwidth
// except pfd_nhwc_to_nchw / pfd_add_f16 which see full-size images.
#include "pfd_common.h"

namespace {

__global__ void nchw_to_nhwc_kernel(const void* __restrict__ x, int src_f32, half_t* __restrict__ y, int B,
                                    int C, int H, int W, float mul, float add, int rep) {
  const long n = (long)B * C * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    // i indexes the NHWC destination so writes are coalesced
    const int c = (int)(i % C);
    long t = i / C;
    const int xw = (int)(t % W);
    t /= W;
    const int yh = (int)(t % H);
    const int b = (int)(t / H);
    const long si = (((long)b * C + c) * H + yh) * W + xw;
    const float v = (src_f32 ? ((const float*)x)[si] : (float)((const half_t*)x)[si]) * mul + add;
    for (int r = 0; r < rep; ++r) y[(long)r * n + i] = (half_t)v;
  }
}

// One block transposes a [32 pixels] x [C] slab through LDS so both sides stay coalesced.
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const half_t* __restrict__ x, void* __restrict__ y,
                                                           int dst_f32, int B, int C, int HW, float mul,
                                                           float add, float lo, float hi) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int p = p0 + r, c = c0 + tx;
    float v = 0.f;
    if (p < HW && c < C) v = (float)x[((long)b * HW + p) * C + c];
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r, p = p0 + tx;
    if (p < HW && c < C) {
      const float v = fminf(fmaxf(tile[tx][r] * mul + add, lo), hi);
      const long di = ((long)b * C + c) * HW + p;
      if (dst_f32) ((float*)y)[di] = v;
      else ((half_t*)y)[di] = (half_t)v;
    }
  }
}

__global__ void im2col_kernel(const half_t* __restrict__ x, long ldx, half_t* __restrict__ col, int B, int H,
                              int W, int Cin, int ks, int stride, int pad, int Ho, int Wo, int Kpad) {
  const long n = (long)B * Ho * Wo * Kpad;
  const int Kreal = ks * ks * Cin;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % Kpad);
    const long m = i / Kpad;
    half_t v = (half_t)0.f;
    if (k < Kreal) {
      const int tap = k / Cin, ci = k - tap * Cin;
      const int ky = tap / ks, kx = tap - ky * ks;
      const int ox = (int)(m % Wo);
      const long t = m / Wo;
      const int oy = (int)(t % Ho);
      const int b = (int)(t / Ho);
      const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((long)b * H + iy) * W + ix) * ldx + ci];
    }
    col[i] = v;
  }
}

__global__ void timestep_embedding_kernel(const int64_t* __restrict__ t, half_t* __restrict__ out, int B, int dim,
                                          float max_period) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * dim) return;
  const int b = i / dim, j = i - b * dim;
  float v = 0.f;
  if (j < 2 * half) {
    const int f = j < half ? j : j - half;
    // freqs = exp(-ln(max_period) * f / half) in fp32, args = t.float() * freqs
    const float freq = expf(-logf(max_period) * (float)f / (float)half);
    const float arg = (float)t[b] * freq;
    v = j < half ? cosf(arg) : sinf(arg);
  }
  out[i] = (half_t)v;
}

__global__ void cfg_ddim_kernel(const half_t* __restrict__ eps, int nb, const float* __restrict__ x,
                                const float* __restrict__ noise, const float* __restrict__ coef,
                                float* __restrict__ x_prev, float* __restrict__ pred_x0,
                                half_t* __restrict__ xin_next, int rep, int B, int C, int h, int w) {
  const long n = (long)B * C * h * w;
  const float a_t = coef[0], a_prev = coef[1], sigma = coef[2], s1mat = coef[3], scale = coef[4];
  const float isq_at = 1.0f / sqrtf(a_t);
  const float sq_aprev = sqrtf(a_prev);
  const float dir = sqrtf(fmaxf(1.0f - a_prev - sigma * sigma, 0.f));
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    // i indexes NCHW
    const int xw = (int)(i % w);
    long t = i / w;
    const int yh = (int)(t % h);
    t /= h;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    const long ei = (((long)b * h + yh) * w + xw) * C + c;  // NHWC
    float e;
    if (nb == 2) {
      const float eu = (float)eps[ei];
      const float ec = (float)eps[n + ei];
      e = eu + scale * (ec - eu);
    } else {
      e = (float)eps[ei] * scale;
    }
    const float xv = x[i];
    const float p0 = (xv - s1mat * e) * isq_at;
    float xp = sq_aprev * p0 + dir * e;
    if (noise) xp += sigma * noise[i];
    x_prev[i] = xp;
    pred_x0[i] = p0;
    if (xin_next) {
      const half_t hv = (half_t)xp;
      for (int r = 0; r < rep; ++r) xin_next[(long)r * n + ei] = hv;
    }
  }
}

__global__ void add_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b, half_t* __restrict__ y,
                           long n) {
  const long nv = n / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    Pack16 p, q, o;
    p.u = reinterpret_cast<const uint4*>(a)[i];
    if (b) {
      q.u = reinterpret_cast<const uint4*>(b)[i];
#pragma unroll
      for (int e = 0; e < 8; ++e) o.e[e] = (half_t)((float)p.e[e] + (float)q.e[e]);
    } else {
      o = p;
    }
    reinterpret_cast<uint4*>(y)[i] = o.u;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const long i = nv * 8 + threadIdx.x;
    y[i] = b ? (half_t)((float)a[i] + (float)b[i]) : a[i];
  }
}

__global__ void axpby_kernel(const half_t* __restrict__ a, float alpha, const half_t* __restrict__ b, float beta,
                             half_t* __restrict__ y, long n) {
  const long nv = n / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    Pack16 p, q, o;
    p.u = reinterpret_cast<const uint4*>(a)[i];
    q.u = make_uint4(0, 0, 0, 0);
    if (b) q.u = reinterpret_cast<const uint4*>(b)[i];
#pragma unroll
    for (int e = 0; e < 8; ++e) o.e[e] = (half_t)fmaf(alpha, (float)p.e[e], beta * (float)q.e[e]);
    reinterpret_cast<uint4*>(y)[i] = o.u;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const long i = nv * 8 + threadIdx.x;
    y[i] = (half_t)fmaf(alpha, (float)a[i], b ? beta * (float)b[i] : 0.f);
  }
}

__global__ void add_rowvec_kernel(const half_t* __restrict__ x, long ldx, const half_t* __restrict__ v,
                                  half_t* __restrict__ y, long ldy, int R, int C) {
  const int nvec = C / 8;
  const long n = (long)R * nvec;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / nvec), c = (int)(i - (long)r * nvec) * 8;
    Pack16 p, q, o;
    p.u = *reinterpret_cast<const uint4*>(x + (long)r * ldx + c);
    q.u = *reinterpret_cast<const uint4*>(v + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) o.e[e] = (half_t)((float)p.e[e] + (float)q.e[e]);
    *reinterpret_cast<uint4*>(y + (long)r * ldy + c) = o.u;
  }
}

__global__ void act_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, long n, int act) {
  const long nv = n / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    Pack16 p, o;
    p.u = reinterpret_cast<const uint4*>(x)[i];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = (float)p.e[e];
      o.e[e] = (half_t)(act == PFD_ACT_SILU ? pfd_silu(v) : act == PFD_ACT_GELU ? pfd_gelu(v)
                        : act == PFD_ACT_RELU ? fmaxf(v, 0.f) : v);
    }
    reinterpret_cast<uint4*>(y)[i] = o.u;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const long i = nv * 8 + threadIdx.x;
    const float v = (float)x[i];
    y[i] = (half_t)(act == PFD_ACT_SILU ? pfd_silu(v) : act == PFD_ACT_GELU ? pfd_gelu(v)
                    : act == PFD_ACT_RELU ? fmaxf(v, 0.f) : v);
  }
}

// fp16 rounding that the compiler cannot elide: hipcc evaluates _Float16 expressions with excess (fp32) precision
// and drops the intermediate narrowing, which changes 4 % of the bytes below (trunc(v * 255) vs trunc(f16(v * 255)))
__device__ __forceinline__ float round_f16(float x) {
  const unsigned short bits = __builtin_bit_cast(unsigned short, (half_t)x);
  unsigned short b2;
  asm volatile("v_mov_b32 %0, %1" : "=v"(b2) : "v"(bits));
  return (float)__builtin_bit_cast(half_t, b2);
}

// 8 pixels-channels per thread: one 16-byte load, one 8-byte store
__global__ void image_u8_kernel(const half_t* __restrict__ x, uint8_t* __restrict__ y, long n, float mul, float add,
                                int f16_image) {
  const long nv = n / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    Pack16 p;
    p.u = reinterpret_cast<const uint4*>(x)[i];
    union { uint2 u; uint8_t b[8]; } o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = fminf(fmaxf((float)p.e[e] * mul + add, 0.f), 1.f);
      if (f16_image) v = round_f16(v);                                 // the image tensor the reference holds (model dtype)
      const float m = v * 255.f;                                       // pic.mul(255) in that dtype ...
      o.b[e] = (uint8_t)(f16_image ? round_f16(m) : m);                // ... then .byte(): truncation
    }
    reinterpret_cast<uint2*>(y)[i] = o.u;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const long i = nv * 8 + threadIdx.x;
    float v = fminf(fmaxf((float)x[i] * mul + add, 0.f), 1.f);
    if (f16_image) v = round_f16(v);
    const float m = v * 255.f;
    y[i] = (uint8_t)(f16_image ? round_f16(m) : m);
  }
}

inline int grid_for(long n, int block) {
  long g = (n + block - 1) / block;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int pfd_nchw_to_nhwc_f16(const void* x, int32_t src_f32, void* y, int32_t B, int32_t C, int32_t H,
                                    int32_t W, float mul, float add, int32_t rep, pfd_stream_t stream) {
  if (!x || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0 || rep < 1) return PFD_EINVAL;
  const long n = (long)B * C * H * W;
  PfdProfScope prof_scope(15, 0.0, 0.0, (hipStream_t)stream);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, src_f32,
                     (half_t*)y, B, C, H, W, mul, add, rep);
  return pfd_check_launch("pfd_nchw_to_nhwc_f16");
}

extern "C" int pfd_nhwc_to_nchw(const void* x, void* y, int32_t dst_f32, int32_t B, int32_t C, int32_t H,
                                int32_t W, float mul, float add, float lo, float hi, pfd_stream_t stream) {
  if (!x || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0) return PFD_EINVAL;
  const int HW = H * W;
  dim3 grid((HW + 63) / 64, (C + 63) / 64, B);
  PfdProfScope prof_scope(15, 0.0, 0.0, (hipStream_t)stream);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const half_t*)x, y, dst_f32, B,
                     C, HW, mul, add, lo, hi);
  return pfd_check_launch("pfd_nhwc_to_nchw");
}

extern "C" int pfd_im2col_f16(const void* x, int64_t ldx, void* col, int32_t B, int32_t H, int32_t W, int32_t Cin,
                              int32_t ksize, int32_t stride, int32_t pad, int32_t Ho, int32_t Wo, int32_t Kpad,
                              pfd_stream_t stream) {
  if (!x || !col || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || ksize <= 0 || stride <= 0 || Ho <= 0 || Wo <= 0)
    return PFD_EINVAL;
  if (Kpad < ksize * ksize * Cin) return PFD_EINVAL;
  const long n = (long)B * Ho * Wo * Kpad;
  PfdProfScope prof_scope(15, 0.0, 0.0, (hipStream_t)stream);
  hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x,
                     (long)ldx, (half_t*)col, B, H, W, Cin, ksize, stride, pad, Ho, Wo, Kpad);
  return pfd_check_launch("pfd_im2col_f16");
}

extern "C" int pfd_timestep_embedding_f16(const int64_t* t, void* out, int32_t B, int32_t dim, float max_period,
                                          pfd_stream_t stream) {
  if (!t || !out || B <= 0 || dim <= 1) return PFD_EINVAL;
  const int n = B * dim;
  PfdProfScope prof_scope(15, 0.0, 0.0, (hipStream_t)stream);
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t,
                     (half_t*)out, B, dim, max_period);
  return pfd_check_launch("pfd_timestep_embedding_f16");
}

extern "C" int pfd_cfg_ddim_step(const void* eps, int32_t nb, const float* x, const float* noise, const float* coef,
                                 float* x_prev, float* pred_x0, void* xin_next, int32_t rep, int32_t B, int32_t C,
                                 int32_t h, int32_t w, pfd_stream_t stream) {
  if (!eps || !x || !coef || !x_prev || !pred_x0) return PFD_EINVAL;
  if (nb < 1 || nb > 2 || B <= 0 || C <= 0 || h <= 0 || w <= 0 || (xin_next && (rep < 1 || rep > 2))) return PFD_EINVAL;
  const long n = (long)B * C * h * w;
  PfdProfScope prof_scope(15, 0.0, 0.0, (hipStream_t)stream);
  hipLaunchKernelGGL(cfg_ddim_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)eps, nb, x, noise, coef, x_prev, pred_x0, (half_t*)xin_next, rep, B, C, h, w);
  return pfd_check_launch("pfd_cfg_ddim_step");
}

extern "C" int pfd_add_f16(const void* a, const void* b, void* y, int64_t n, pfd_stream_t stream) {
  if (!a || !y || n <= 0) return PFD_EINVAL;
  if ((reinterpret_cast<uintptr_t>(a) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) ||
      (b && (reinterpret_cast<uintptr_t>(b) & 15)))
    return PFD_EINVAL;
  PfdProfScope prof_scope(15, 0.0, 0.0, (hipStream_t)stream);
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 8 + 1, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)a, (const half_t*)b, (half_t*)y, (long)n);
  return pfd_check_launch("pfd_add_f16");
}

extern "C" int pfd_axpby_f16(const void* a, float alpha, const void* b, float beta, void* y, int64_t n,
                             pfd_stream_t stream) {
  if (!a || !y || n <= 0) return PFD_EINVAL;
  if ((reinterpret_cast<uintptr_t>(a) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) ||
      (b && (reinterpret_cast<uintptr_t>(b) & 15)))
    return PFD_EINVAL;
  PfdProfScope prof_scope(15, 0.0, 0.0, (hipStream_t)stream);
  hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n / 8 + 1, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)a, alpha, (const half_t*)b, beta, (half_t*)y, (long)n);
  return pfd_check_launch("pfd_axpby_f16");
}

// y = x + v with the partial row sums of y (PfdGemmDesc.ln_stats layout) in the same launch: four lanes per
// (row, 160-column slice), lane k takes chunks k, k + 4, ..., k + 16 -- the summation order of ln_rowstats_kernel
// (norm.hip) and of the statistics-emitting GEMM epilogue, so the `x + bias` rows of the zero-context shortcut carry
// bit-for-bit the statistics the full out-projection would have written.
__global__ __launch_bounds__(256) void add_rowvec_lnstats_kernel(const half_t* __restrict__ x, long ldx,
                                                                 const half_t* __restrict__ v, half_t* __restrict__ y,
                                                                 long ldy, int R, int P, float2* __restrict__ st) {
  const long grp = ((long)blockIdx.x * 256 + threadIdx.x) >> 2;
  const int k = threadIdx.x & 3;
  const long ngrp = (long)R * P;
  const long g = min(grp, ngrp - 1);
  const int m = (int)(g / P), p = (int)(g - (long)m * P);
  const half_t* src = x + (long)m * ldx + p * 160;
  const half_t* vv = v + p * 160;
  Pack16 a[5], b[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {   // all loads first
    a[j].u = *reinterpret_cast<const uint4*>(src + (k + 4 * j) * 8);
    b[j].u = *reinterpret_cast<const uint4*>(vv + (k + 4 * j) * 8);
  }
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    Pack16 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      o.e[e] = (half_t)((float)a[j].e[e] + (float)b[j].e[e]);
      const float f = (float)o.e[e];
      s += f;
      q = fmaf(f, f, q);
    }
    if (grp < ngrp) *reinterpret_cast<uint4*>(y + (long)m * ldy + p * 160 + (k + 4 * j) * 8) = o.u;
  }
  s += __shfl_xor(s, 1, 64);
  q += __shfl_xor(q, 1, 64);
  s += __shfl_xor(s, 2, 64);
  q += __shfl_xor(q, 2, 64);
  if (k == 0 && grp < ngrp) st[grp] = make_float2(s, q);
}

extern "C" int pfd_add_rowvec_lnstats_f16(const void* x, int64_t ldx, const void* v, void* y, int64_t ldy, int32_t R,
                                          int32_t C, void* stats, pfd_stream_t stream) {
  if (!x || !v || !y || !stats || R <= 0 || C <= 0) return PFD_EINVAL;
  if ((C % 160) || C > 1280) return PFD_ESHAPE;
  if ((ldx & 7) || (ldy & 7) || (reinterpret_cast<uintptr_t>(stats) & 7)) return PFD_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(v) & 15) || (reinterpret_cast<uintptr_t>(y) & 15))
    return PFD_EINVAL;
  const int P = C / 160;
  const long ngrp = (long)R * P;
  PfdProfScope prof_scope(15, 0.0, 0.0, (hipStream_t)stream);
  hipLaunchKernelGGL(add_rowvec_lnstats_kernel, dim3((unsigned)((ngrp + 63) / 64)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)x, (long)ldx, (const half_t*)v, (half_t*)y, (long)ldy, R, P,
                     reinterpret_cast<float2*>(stats));
  return pfd_check_launch("pfd_add_rowvec_lnstats_f16");
}

extern "C" int pfd_add_rowvec_f16(const void* x, int64_t ldx, const void* v, void* y, int64_t ldy, int32_t R,
                                  int32_t C, pfd_stream_t stream) {
  if (!x || !v || !y || R <= 0 || C <= 0) return PFD_EINVAL;
  if ((C & 7) || (ldx & 7) || (ldy & 7)) return PFD_EINVAL;
  const long n = (long)R * (C / 8);
  PfdProfScope prof_scope(15, 0.0, 0.0, (hipStream_t)stream);
  hipLaunchKernelGGL(add_rowvec_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)x, (long)ldx, (const half_t*)v, (half_t*)y, (long)ldy, R, C);
  return pfd_check_launch("pfd_add_rowvec_f16");
}

extern "C" int pfd_image_u8_f16(const void* x, void* y, int64_t n, float mul, float add, int32_t f16_image,
                                pfd_stream_t stream) {
  if (!x || !y || n <= 0) return PFD_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 7)) return PFD_EINVAL;
  PfdProfScope prof_scope(15, 0.0, 0.0, (hipStream_t)stream);
  hipLaunchKernelGGL(image_u8_kernel, dim3(grid_for(n / 8 + 1, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)x, (uint8_t*)y, (long)n, mul, add, f16_image);
  return pfd_check_launch("pfd_image_u8_f16");
}

extern "C" int pfd_act_f16(const void* x, void* y, int64_t n, int32_t act, pfd_stream_t stream) {
  if (!x || !y || n <= 0) return PFD_EINVAL;
  if (act < PFD_ACT_NONE || act > PFD_ACT_SILU) return PFD_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15)) return PFD_EINVAL;
  PfdProfScope prof_scope(15, 0.0, 0.0, (hipStream_t)stream);
  hipLaunchKernelGGL(act_kernel, dim3(grid_for(n / 8 + 1, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)x, (half_t*)y, (long)n, act);
  return pfd_check_launch("pfd_act_f16");
}
