// CPU emulation of csrc/norm.hip (test infrastructure; see hip/hip_runtime.h in this directory for the execution model): the
// single-launch small-slab GroupNorm and the GroupNorm apply from producer statistics (eight slabs' partials / eight rows in
// flight) against a double-precision GroupNorm.
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

#include "norm_emu.inc"

typedef _Float16 h16;
static std::mt19937 rng(5);
static std::vector<h16> rand_h(size_t n, float scale, float off = 0.f) {
  std::uniform_real_distribution<float> d(-1.f, 1.f);
  std::vector<h16> v(n);
  for (auto& x : v) x = (h16)(d(rng) * scale + off);
  return v;
}
static int g_fail = 0, g_total = 0;

static void check(const char* name, const std::vector<h16>& got, const std::vector<double>& ref, const std::vector<h16>* same, const char* same_what) {
  double me = 0, mr = 0;
  for (size_t i = 0; i < got.size(); ++i) { me = std::max(me, fabs((double)got[i] - ref[i])); mr = std::max(mr, fabs(ref[i])); }
  bool ok = me <= 6e-3 * std::max(1.0, mr);
  std::string extra;
  if (same) {
    size_t nd = 0;
    for (size_t i = 0; i < got.size(); ++i) nd += memcmp(&got[i], &(*same)[i], sizeof(h16)) != 0;
    if (nd) { ok = false; extra = std::string(" | ") + same_what + ": " + std::to_string(nd) + " elements differ"; }
    else extra = std::string(" | == ") + same_what + " bitwise";
  }
  ++g_total;
  g_fail += !ok;
  printf("%s %-64s max err %.2e (max |ref| %.2f)%s\n", ok ? "ok  " : "FAIL", name, me, mr, extra.c_str());
  fflush(stdout);
}

static std::vector<double> gn_ref(const std::vector<h16>& x1, const std::vector<h16>& x2, const std::vector<h16>& gm, const std::vector<h16>& bt,
                                  int B, int HW, int C1, int C2, int G, float eps, int act) {
  const int C = C1 + C2, cpg = C / G;
  std::vector<double> ref((size_t)B * HW * C);
  auto at = [&](int b, int r, int c) { return c < C1 ? (double)x1[((size_t)b * HW + r) * C1 + c] : (double)x2[((size_t)b * HW + r) * C2 + c - C1]; };
  for (int b = 0; b < B; ++b)
    for (int g = 0; g < G; ++g) {
      double a = 0, q = 0;
      for (int r = 0; r < HW; ++r)
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) { const double v = at(b, r, c); a += v; q += v * v; }
      const double n = (double)HW * cpg, mean = a / n, rstd = 1.0 / sqrt(std::max(q / n - mean * mean, 0.0) + eps);
      for (int r = 0; r < HW; ++r)
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
          double v = (at(b, r, c) - mean) * rstd * (double)gm[c] + (double)bt[c];
          if (act == PFD_ACT_SILU) v = v / (1.0 + exp(-v));
          ref[((size_t)b * HW + r) * C + c] = v;
        }
    }
  return ref;
}

// pfd_groupnorm_f16 (small-slab single launch where the shape qualifies) against double precision
static void small_case(int B, int HW, int C1, int C2, int act) {
  const int C = C1 + C2, G = 32;
  auto x1 = rand_h((size_t)B * HW * C1, 2.f, 0.7f), x2 = rand_h((size_t)B * HW * std::max(C2, 8), 1.f), gm = rand_h(C, 1.f), bt = rand_h(C, 0.5f);
  std::vector<h16> y((size_t)B * HW * C, (h16)-7.f);
  const size_t wsb = pfd_groupnorm_ws_bytes(B, C, HW);
  std::vector<char> ws(wsb);
  int rc = pfd_groupnorm_f16(x1.data(), C1, C1, C2 ? x2.data() : nullptr, C2, C2, gm.data(), bt.data(), y.data(), C, B, HW, G, 1e-5f, act, ws.data(), wsb, nullptr);
  char name[160];
  snprintf(name, sizeof(name), "groupnorm B%d HW%d C%d+%d act%d (rc %d)", B, HW, C1, C2, act, rc);
  check(name, y, gn_ref(x1, x2, gm, bt, B, HW, C1, C2, G, 1e-5f, act), nullptr, "");
}

// pfd_groupnorm_pstats_f16 (statistics in the producers' layout, computed on the host here; grouped partial loads where a
// group is at most two producer groups per source) against double precision
static void pstats_case(int B, int HW, int C1, int C2, int act) {
  const int C = C1 + C2, G = 32;
  auto x1 = rand_h((size_t)B * HW * C1, 1.5f, 0.3f), x2 = rand_h((size_t)B * HW * std::max(C2, 8), 1.f), gm = rand_h(C, 1.f), bt = rand_h(C, 0.5f);
  auto mk = [&](const std::vector<h16>& x, int Cs) {
    const int cpp = Cs / 32, tn = Cs / 160;
    std::vector<float> st((size_t)(B * HW / 64) * tn * 32, 0.f);
    for (int sl = 0; sl < B * HW / 64; ++sl)
      for (int c = 0; c < Cs; ++c) {
        double a = 0, q = 0;
        for (int r = 0; r < 64; ++r) { const double v = (double)x[(size_t)(sl * 64 + r) * Cs + c]; a += v; q += v * v; }
        const size_t o = (((size_t)sl * tn + c / 160) * 16 + (c % 160) / cpp) * 2;
        st[o] += (float)a; st[o + 1] += (float)q;
      }
    return st;
  };
  auto s1 = mk(x1, C1);
  std::vector<float> s2 = C2 ? mk(x2, C2) : std::vector<float>(2, 0.f);
  std::vector<h16> y((size_t)B * HW * C, (h16)-7.f);
  char name[160];
  if (!pfd_groupnorm_takes_pstats(B, C1, C2, HW, G)) { ++g_total; ++g_fail; printf("FAIL pstats B%d HW%d C%d+%d: shape refused\n", B, HW, C1, C2); return; }
  int rc = pfd_groupnorm_pstats_f16(x1.data(), C1, C1, s1.data(), C2 ? x2.data() : nullptr, C2, C2, C2 ? s2.data() : nullptr, gm.data(), bt.data(), y.data(), C, B, HW, G, 1e-5f, act, nullptr);
  snprintf(name, sizeof(name), "groupnorm pstats B%d HW%d C%d+%d act%d (rc %d)", B, HW, C1, C2, act, rc);
  check(name, y, gn_ref(x1, x2, gm, bt, B, HW, C1, C2, G, 1e-5f, act), nullptr, "");
}

int main(int argc, char** argv) {
  const bool quick = argc > 1 && !strcmp(argv[1], "--quick");   // the CPU suite's subset
  small_case(8, 64, 1280, 0, PFD_ACT_SILU);      // 8^2: 640 chunks per group slab, 3 slots per thread
  small_case(4, 64, 1280, 1280, PFD_ACT_NONE);   // skip concat, cpg 80
  small_case(4, 100, 1408, 0, PFD_ACT_SILU);     // cpg 44, cpr 11 (carries in the index walk), ragged HW
  if (!quick) {
    small_case(4, 256, 1280, 0, PFD_ACT_SILU);   // 16^2 (B x G = 128): 2560 chunks
    small_case(4, 64, 1024, 0, PFD_ACT_SILU);    // cpg 32, cpr 8: no carries
    small_case(4, 64, 1280, 640, PFD_ACT_SILU);  // cpg 60 straddles the two sources: the plain form both times
  }
  pstats_case(1, 4096, 320, 0, PFD_ACT_SILU);    // 64^2: 64 slabs, 8 per thread
  pstats_case(1, 1024, 320, 320, PFD_ACT_SILU);  // skip concat: two producer groups per group
  if (!quick) {
    pstats_case(1, 4608, 320, 0, PFD_ACT_NONE);  // 72 slabs: a second trip of the chunked fold
    pstats_case(2, 256, 640, 0, PFD_ACT_SILU);   // 4 slabs, six clamped slots
  }
  printf("%d cases, %d failed\n", g_total, g_fail);
  return g_fail;
}
