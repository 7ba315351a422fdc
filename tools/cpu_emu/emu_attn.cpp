// CPU emulation of csrc/attention.hip (see hip/hip_runtime.h): the fused attention kernels (8-wave d = 40 form with the PV
// product on 16x16x32 and the maximum folded into the QK^T MFMA; 4-wave form for every head dim) against a double-precision
// softmax(scale Q K^T) V.
//   usage: emu_attn [--quick] [--w4]     (--w4: the 4-wave d = 40 form; prints one line per case with a checksum of the output bits)
#include <stdio.h>

#include <random>
#include <string>

#include "hip/hip_runtime.h"
#include "pfd_common.h"
bool pfd_prof_on() { return false; }
void pfd_prof_begin(int, double, double, hipStream_t) {}
void pfd_prof_end(hipStream_t) {}
int pfd_check_launch(const char*) { return 0; }
void pfd_set_error(const char*) {}

#include "attention_emu.inc"
#include "attention3_kernel.h"   // round 6: the software-pipelined d = 40 kernel (no inline asm: included as it is)

typedef _Float16 h16;
static std::mt19937 rng(11);
static std::vector<h16> rand_h(size_t n, float scale) {
  std::uniform_real_distribution<float> d(-1.f, 1.f);
  std::vector<h16> v(n);
  for (auto& x : v) x = (h16)(d(rng) * scale);
  return v;
}
static int g_fail = 0, g_total = 0;

static void run_case(int B, int H, int Nq, int Nk, int D, int spike = 0) {
  const int C = H * D;
  const long nkp = (Nk + 7) / 8 * 8;
  auto Q = rand_h((size_t)B * Nq * C, 1.5f), K = rand_h((size_t)B * Nk * C, 1.5f), V = rand_h((size_t)B * Nk * C, 1.f);
  if (spike > 0 && spike < Nk)   // key `spike` = 4 x query (spike % Nq): the row maximum jumps in the middle of the stream
    for (int b = 0; b < B; ++b)
      for (int c = 0; c < C; ++c) K[((size_t)b * Nk + spike) * C + c] = (h16)(4.0f * (float)Q[((size_t)b * Nq + spike % Nq) * C + c]);
  std::vector<h16> Vt((size_t)C * B * nkp, (h16)0.f), O((size_t)B * Nq * C, (h16)-9.f);
  for (int b = 0; b < B; ++b)
    for (int j = 0; j < Nk; ++j)
      for (int c = 0; c < C; ++c) Vt[(size_t)c * B * nkp + (size_t)b * nkp + j] = V[((size_t)b * Nk + j) * C + c];
  PfdAttnDesc d;
  memset(&d, 0, sizeof(d));
  d.Q = Q.data(); d.K = K.data(); d.Vt = Vt.data(); d.O = O.data();
  d.ldq = C; d.ldk = C; d.ldo = C; d.ldvt = (long)B * nkp;
  d.q_bs = (long)Nq * C; d.k_bs = (long)Nk * C; d.o_bs = (long)Nq * C; d.vt_bs = nkp;
  d.B = B; d.H = H; d.Nq = Nq; d.Nk = Nk; d.D = D;
  d.scale = 1.0f / sqrtf((float)D);
  const int rc = pfd_attention_f16(&d, nullptr);
  double me = 0, mr = 0;
  std::vector<double> s(Nk);
  for (int b = 0; b < B && rc == 0; ++b)
    for (int h = 0; h < H; ++h)
      for (int i = 0; i < Nq; ++i) {
        double mx = -1e300;
        for (int j = 0; j < Nk; ++j) {
          double a = 0;
          for (int e = 0; e < D; ++e) a += (double)Q[((size_t)b * Nq + i) * C + h * D + e] * (double)K[((size_t)b * Nk + j) * C + h * D + e];
          s[j] = a * d.scale;
          mx = std::max(mx, s[j]);
        }
        double den = 0;
        for (int j = 0; j < Nk; ++j) { s[j] = exp(s[j] - mx); den += s[j]; }
        for (int e = 0; e < D; ++e) {
          double o = 0;
          for (int j = 0; j < Nk; ++j) o += s[j] * (double)V[((size_t)b * Nk + j) * C + h * D + e];
          o /= den;
          mr = std::max(mr, fabs(o));
          me = std::max(me, fabs(o - (double)O[((size_t)b * Nq + i) * C + h * D + e]));
        }
      }
  uint64_t sum = 1469598103934665603ull;   // FNV-1a over the output bits
  for (auto& x : O) { unsigned short u; memcpy(&u, &x, 2); sum = (sum ^ u) * 1099511628211ull; }
  const bool ok = rc == 0 && me <= 4e-3 * std::max(1.0, mr);
  ++g_total;
  g_fail += !ok;
  printf("%s attention B%d H%d Nq%d Nk%d D%d  rc %d  max err %.2e (max |ref| %.2f)  bits %016llx\n", ok ? "ok  " : "FAIL", B, H, Nq, Nk, D, rc, me,
         mr, (unsigned long long)sum);
  fflush(stdout);
}

int main(int argc, char** argv) {
  bool quick = false, force8 = true;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--a3")) {       // attention3_kernel: whole 64-key tiles; one block / ragged queries / three heads
      setenv("PFD_ATTN3_FORCE", "1", 1);
      run_case(1, 1, 256, 192, 40);
      run_case(1, 2, 300, 128, 40);
      run_case(2, 1, 256, 320, 40);
      run_case(1, 1, 256, 320, 40, 200);   // deferred rescale taken in tile 3 (sub-block A or B by the query's position)
      run_case(1, 1, 256, 256, 40, 100);   // ... in tile 1, query 100 = sub-block B of wave 1
      run_case(1, 1, 256, 704, 40, 500);   // 11 tiles: both rings wrap twice (odd count: single-tile interval + five pairs)
      run_case(1, 1, 256, 768, 40);        // 12 tiles: pairs only
      printf("%d cases, %d failed\n", g_total, g_fail);
      return g_fail;
    }
    quick = quick || !strcmp(argv[i], "--quick");
    if (!strcmp(argv[i], "--w4")) force8 = false;   // the 4-wave d = 40 form (what small launches take)
  }
  if (force8) setenv("PFD_ATTN_FORCE8", "1", 1);      // the 8-wave d = 40 form at these (small) sizes
  else unsetenv("PFD_ATTN_FORCE8");
  run_case(1, 2, 300, 200, 40);           // ragged queries and keys, 4 KV tiles (3 full + 1 peeled)
  run_case(1, 1, 256, 148, 40);           // the cross-attention length (148 context tokens)
  if (!quick) {
    run_case(2, 1, 64, 64, 160);          // 8^2 level
    run_case(1, 1, 100, 77, 80);
    run_case(1, 1, 130, 96, 96);          // SeeCoder
  }
  printf("%d cases, %d failed\n", g_total, g_fail);
  return g_fail;
}
