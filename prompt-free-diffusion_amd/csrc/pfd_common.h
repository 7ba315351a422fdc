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

__device__ __forceinline__ float pfd_gelu(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float pfd_silu(float x) { return x / (1.0f + __expf(-x)); }

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

// host-side error plumbing (defined in capi.cpp)
int pfd_check_launch(const char* what);
void pfd_set_error(const char* msg);
