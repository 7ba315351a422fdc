// Shared device-side helpers for the gfx950 kernels of libpfd_hip.so.
// Wavefront = 64 lanes, MFMA f16 shapes 32x32x16 / 16x16x32, fp32 accumulate.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pfd_hip.h"

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));

#define PFD_WAVE 64

// MFMA C/D layouts (guide §3):
//   32x32: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5), r in [0,16)
//   16x16: col = lane & 15, row = 4*(lane >> 4) + r,                   r in [0,4)
// A operand: lane holds 8 consecutive k of row (lane & 31 | lane & 15); B likewise for a column.
// The k owned by (lane-group, j) only has to agree between A and B.
__device__ __forceinline__ int mfma32_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// exact-erf GELU (torch F.gelu default) evaluated with the Abramowitz-Stegun 7.1.26 erfc form:
// erfc(z) = t*P4(t)*exp(-z^2), t = 1/(1 + p z); |abs error| < 5e-7 in gelu(x), far inside one fp16 ulp of
// the stored result, and ~13 VALU ops instead of ocml erff's branchy ~40 -- the GEGLU epilogue of the
// 64^2 feed-forward evaluates 42 M of these per launch. The negative side uses erfc directly, so there
// is no 1 - erf cancellation in the tail.
__device__ __forceinline__ float pfd_gelu(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(t, 1.061405429f, -1.453152027f);
  poly = fmaf(t, poly, 1.421413741f);
  poly = fmaf(t, poly, -0.284496736f);
  poly = fmaf(t, poly, 0.254829592f);
  const float half_erfc = 0.5f * t * poly * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
  return x * (x >= 0.0f ? 1.0f - half_erfc : half_erfc);
}
__device__ __forceinline__ float pfd_silu(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

union Pack16 {
  uint4 u;
  half8_t h;
  half_t e[8];
};

union Pack8 {
  uint2 u;
  half_t e[4];
};

// XCD-aware, bijective remap of a linear block id: the dispatcher places block b on XCD
// b % 8; give every XCD one contiguous chunk of the tile space so neighbouring tiles
// (which share an operand panel) meet in the same L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + idx;
}

// optional per-launch event timing (capi.hip); buckets: 0-7 gemm <TM,TN,CONV>, 8 attention,
// 9 swin attention, 10 groupnorm, 11 layernorm
bool pfd_prof_on();
void pfd_prof_begin(int bucket, double flops, double bytes, hipStream_t stream);
void pfd_prof_end(hipStream_t stream);
// event pair around every launch of the enclosing scope (no-op unless pfd_prof_enable(1)); the destructor runs after the
// `return pfd_check_launch(...)` expression, i.e. after the launches
struct PfdProfScope {
  hipStream_t s;
  bool on;
  PfdProfScope(int bucket, double flops, double bytes, hipStream_t st) : s(st), on(pfd_prof_on()) {
    if (on) pfd_prof_begin(bucket, flops, bytes, s);
  }
  ~PfdProfScope() {
    if (on) pfd_prof_end(s);
  }
  PfdProfScope(const PfdProfScope&) = delete;
  PfdProfScope& operator=(const PfdProfScope&) = delete;
};

// host-side error plumbing (defined in capi.cpp)
int pfd_check_launch(const char* what);
void pfd_set_error(const char* msg);
