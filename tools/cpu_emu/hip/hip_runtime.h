// CPU emulation shim for the gfx950 kernels of csrc/ (tools/cpu_emu/README in emu_gemm.cpp): this header stands in for
// <hip/hip_runtime.h> when a kernel file is compiled for the HOST by tools/cpu_emu/build.py.  Test infrastructure only --
// nothing in the product includes or links it.
//
// Execution model: one OS thread per GPU thread of ONE block at a time (blocks run one after another), __shared__ objects
// are function-local statics, s_barrier / __syncthreads is a pthread barrier over the block, and every wave-level
// operation (MFMA, shuffles, ballot, readfirstlane) is a rendezvous of the wave's 64 threads on a per-wave barrier.
// Waits (s_waitcnt), priorities, sleeps and scheduling barriers are no-ops: a copy "lands" when it is issued, so the model
// checks indexing, slot arithmetic and barrier / hand-over protocols (threads really run concurrently), not wait counts.
#pragma once
#define PFD_CPU_EMU 1
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define address_space(x)   /* __attribute__((address_space(n))) -> __attribute__(()) */

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct float2 { float x, y; };
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
typedef void* hipStream_t;
typedef int hipError_t;

namespace emu {
struct Wave {
  pthread_barrier_t bar;
  uint64_t slot[64];
  _Float16 a[64][8], b[64][8];
};
struct Block {
  pthread_barrier_t bar;
  std::vector<Wave> waves;
};
inline thread_local dim3 t_idx, b_idx;
inline dim3 b_dim, g_dim;
inline Block* cur = nullptr;
inline const void* kernarg = nullptr;
inline bool dry_run = false;                // record the launches, do not execute them
inline std::vector<std::string> launched;   // the kernel expression of every launch, in order (hipLaunchKernelGGL's first argument)
inline int lane() { return t_idx.x & 63; }
inline Wave& wave() { return cur->waves[t_idx.x >> 6]; }
inline void wave_sync() { pthread_barrier_wait(&wave().bar); }
inline void block_sync() { pthread_barrier_wait(&cur->bar); }
// one team of block.x OS threads per launch walks the blocks of the grid one after another (a block barrier between two
// blocks keeps them sequential: the __shared__ statics are the CU's LDS, reused by the next block as on the device)
inline void launch(const std::function<void()>& body, dim3 grid, dim3 block, const void* arg0) {
  if (dry_run) return;
  const int nthr = (int)block.x, nw = (nthr + 63) / 64;
  if (nthr % 64) { fprintf(stderr, "emu: block size %d is not a multiple of 64\n", nthr); abort(); }
  b_dim = block;
  g_dim = grid;
  kernarg = arg0;
  Block blk;
  blk.waves = std::vector<Wave>(nw);
  pthread_barrier_init(&blk.bar, nullptr, nthr);
  for (auto& w : blk.waves) pthread_barrier_init(&w.bar, nullptr, 64);
  cur = &blk;
  std::vector<std::thread> th;
  th.reserve(nthr);
  for (int t = 0; t < nthr; ++t)
    th.emplace_back([&, t]() {
      t_idx = dim3(t, 0, 0);
      for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
          for (unsigned bx = 0; bx < grid.x; ++bx) {
            b_idx = dim3(bx, by, bz);
            body();
            pthread_barrier_wait(&blk.bar);
          }
    });
  for (auto& x : th) x.join();
  for (auto& w : blk.waves) pthread_barrier_destroy(&w.bar);
  pthread_barrier_destroy(&blk.bar);
  cur = nullptr;
}
template <class T>
inline T exchange(T v, int src) {   // every lane of the wave calls this; returns lane src's value
  static_assert(sizeof(T) <= 8, "exchange");
  Wave& w = wave();
  uint64_t raw = 0;
  memcpy(&raw, &v, sizeof(T));
  w.slot[lane()] = raw;
  wave_sync();
  T out;
  memcpy(&out, &w.slot[src & 63], sizeof(T));
  wave_sync();
  return out;
}
}  // namespace emu

#define threadIdx emu::t_idx
#define blockIdx emu::b_idx
#define blockDim emu::b_dim
#define gridDim emu::g_dim

#define EMU_FIRST(a, ...) a
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                                      \
  do {                                                                                                    \
    auto emu_arg0 = EMU_FIRST(__VA_ARGS__);                                                               \
    emu::launched.push_back(#kernel);                                                                     \
    emu::launch([&]() { kernel(__VA_ARGS__); }, grid, block, &emu_arg0);                                  \
  } while (0)

template <class T> static inline T min(T a, T b) { return a < b ? a : b; }
template <class T> static inline T max(T a, T b) { return a > b ? a : b; }
static inline long min(long a, int b) { return a < b ? a : (long)b; }
static inline long max(long a, int b) { return a > b ? a : (long)b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
#define __expf(x) expf(x)   /* glibc declares (but does not export) __expf */

template <class T> static inline T __shfl(T v, int src, int = 64) { return emu::exchange(v, src); }
template <class T> static inline T __shfl_xor(T v, int mask, int = 64) { return emu::exchange(v, emu::lane() ^ mask); }

#define __syncthreads() emu::block_sync()
#define __builtin_amdgcn_s_barrier() emu::block_sync()
// a wave runs in lockstep: "my copies have landed" (s_waitcnt) is a statement about all 64 lanes.  The lanes are independent
// threads here, so every wait is a rendezvous of the wave -- what a hand-over protocol that lets ONE lane publish the wave's
// progress (variant 95) relies on
#define __builtin_amdgcn_s_waitcnt(x) emu::wave_sync()
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) std::this_thread::yield()
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_kernarg_segment_ptr() (emu::kernarg)
// wave-uniform by construction at every use in csrc/ (tid >> 6, a block-uniform flag): lane 0's value
#define __builtin_amdgcn_readfirstlane(x) emu::exchange((x), 0)

static inline uint64_t emu_ballot(bool p) {
  emu::Wave& w = emu::wave();
  w.slot[emu::lane()] = p ? 1 : 0;
  emu::wave_sync();
  uint64_t m = 0;
  for (int l = 0; l < 64; ++l) m |= (uint64_t)(w.slot[l] & 1) << l;
  emu::wave_sync();
  return m;
}
#define __builtin_amdgcn_ballot_w64(p) emu_ballot(p)
static inline int __any(int p) { return emu_ballot(p != 0) != 0; }
static inline int __all(int p) { return emu_ballot(p != 0) == ~0ull; }

// LDS-DMA: lane l's 16 bytes go to (wave-uniform destination) + 16 l
static inline void emu_global_load_lds(const void* src, void* dst, int bytes, int, int) {
  memcpy((char*)dst + emu::lane() * bytes, src, bytes);
}
#define __builtin_amdgcn_global_load_lds(src, dst, bytes, a, b) emu_global_load_lds((const void*)(src), (void*)(dst), bytes, a, b)

// v_mfma_f32_16x16x32_f16: D[i][j] = C[i][j] + sum_k A[i][k] B[k][j]; lane l holds 8 consecutive k (group l >> 4) of row
// (A) / column (B) l & 15, and D / C as col = l & 15, rows 4 (l >> 4) + r  (pfd_common.h, guide section 3)
typedef _Float16 emu_h8 __attribute__((ext_vector_type(8)));
typedef float emu_f4 __attribute__((ext_vector_type(4)));
static inline emu_f4 emu_mfma_16x16x32_f16(emu_h8 a, emu_h8 b, emu_f4 c) {
  emu::Wave& w = emu::wave();
  const int l = emu::lane();
  for (int e = 0; e < 8; ++e) {
    w.a[l][e] = a[e];
    w.b[l][e] = b[e];
  }
  emu::wave_sync();
  emu_f4 d;
  const int j = l & 15;
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * (l >> 4) + r;
    float s = c[r];
    for (int g = 0; g < 4; ++g)
      for (int e = 0; e < 8; ++e) s += (float)w.a[i + 16 * g][e] * (float)w.b[j + 16 * g][e];
    d[r] = s;
  }
  emu::wave_sync();
  return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) emu_mfma_16x16x32_f16(a, b, c)

// v_mfma_f32_32x32x16_f16: lane l holds 8 consecutive k (group l >> 5) of row (A) / column (B) l & 31; D / C as
// col = l & 31, row = (r & 3) + 8 (r >> 2) + 4 (l >> 5), r in [0, 16)  (pfd_common.h mfma32_row, guide section 3)
typedef float emu_f16v __attribute__((ext_vector_type(16)));
static inline emu_f16v emu_mfma_32x32x16_f16(emu_h8 a, emu_h8 b, emu_f16v c) {
  emu::Wave& w = emu::wave();
  const int l = emu::lane();
  for (int e = 0; e < 8; ++e) {
    w.a[l][e] = a[e];
    w.b[l][e] = b[e];
  }
  emu::wave_sync();
  emu_f16v d;
  const int j = l & 31, hi = l >> 5;
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float s = c[r];
    for (int g = 0; g < 2; ++g)
      for (int e = 0; e < 8; ++e) s += (float)w.a[i + 32 * g][e] * (float)w.b[j + 32 * g][e];
    d[r] = s;
  }
  emu::wave_sync();
  return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emu_mfma_32x32x16_f16(a, b, c)

// v_permlane16_swap: the odd 16-lane rows of the first operand trade places with the even rows of the second
// ({a', b'}: a' = a.row0 b.row0 a.row2 b.row2, b' = a.row1 b.row1 a.row3 b.row3)
struct emu_u2 {
  unsigned v[2];
  unsigned operator[](int i) const { return v[i]; }
};
static inline emu_u2 emu_permlane16_swap(unsigned a, unsigned b) {
  emu::Wave& w = emu::wave();
  const int l = emu::lane(), row = l >> 4, pos = l & 15;
  w.slot[l] = ((uint64_t)b << 32) | a;
  emu::wave_sync();
  auto A = [&](int lane) { return (unsigned)(w.slot[lane] & 0xffffffffu); };
  auto B = [&](int lane) { return (unsigned)(w.slot[lane] >> 32); };
  emu_u2 r;
  r.v[0] = (row & 1) ? B((row - 1) * 16 + pos) : A(l);
  r.v[1] = (row & 1) ? B(l) : A((row + 1) * 16 + pos);
  emu::wave_sync();
  return r;
}
#define __builtin_amdgcn_permlane16_swap(a, b, fi, bc) emu_permlane16_swap(a, b)
// v_permlane32_swap: the upper 32 lanes of the first operand trade places with the lower 32 lanes of the second
// ({a', b'}: a' = a.lo b.lo, b' = a.hi b.hi)
static inline emu_u2 emu_permlane32_swap(unsigned a, unsigned b) {
  emu::Wave& w = emu::wave();
  const int l = emu::lane();
  w.slot[l] = ((uint64_t)b << 32) | a;
  emu::wave_sync();
  auto A = [&](int lane) { return (unsigned)(w.slot[lane] & 0xffffffffu); };
  auto B = [&](int lane) { return (unsigned)(w.slot[lane] >> 32); };
  emu_u2 r;
  r.v[0] = l < 32 ? A(l) : B(l - 32);
  r.v[1] = l < 32 ? A(l + 32) : B(l);
  emu::wave_sync();
  return r;
}
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) emu_permlane32_swap(a, b)
