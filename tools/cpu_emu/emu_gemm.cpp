// CPU emulation of the wide-tile GEMM / convolution kernels of csrc/gemm_glds.hip (test infrastructure; see hip/hip_runtime.h
// in this directory for the execution model).  tools/cpu_emu/build.py writes gemm_glds_emu.inc -- the kernel file with its
// gfx950 inline-asm statements replaced by their C meaning -- and compiles this file for the host.  The same host dispatcher
// (pfd_gemm160_try: tile choice, split-K, forced variants) and the same device code (index arithmetic, LDS images and
// swizzles, ring slots, barriers, epilogues, the split-K reductions) then run on CPU threads, and the result is
// compared with a double-precision reference.  What it cannot see: s_waitcnt counts (a copy lands when it is issued),
// register allocation, timing.
//   usage: emu_gemm [case ...]      no argument = the built-in list; exit code = number of failed cases
#include <stdio.h>

#include <random>
#include <string>

#include "hip/hip_runtime.h"


// ---- what the kernel file expects from the rest of the library ----
#include "pfd_common.h"
bool pfd_prof_on() { return false; }
void pfd_prof_begin(int, double, double, hipStream_t) {}
void pfd_prof_end(hipStream_t) {}
static std::string g_err;
int pfd_check_launch(const char*) { return 0; }
void pfd_set_error(const char* m) { g_err = m; }
int pfd_ln_rowstats_launch(const half_t*, long, int, int, float*, hipStream_t, bool) { return PFD_ESHAPE; }

#include "gemm_glds_emu.inc"

// ---- cases ----
typedef _Float16 h16;
static std::mt19937 rng(99);
static std::vector<h16> rand_h(size_t n, float scale) {
  std::uniform_real_distribution<float> d(-1.f, 1.f);
  std::vector<h16> v(n);
  for (auto& x : v) x = (h16)(d(rng) * scale);
  return v;
}

struct Case {
  const char* what;
  int M, N, K, variant, splits;
  bool res = false, rowvec = false;
  int act = 0, k_split = 0, zero_rows = 0;
  int ksize = 0, stride = 1, pad = 0, ups = 0, B = 0, H = 0, W = 0, Cin = 0;
  int base_variant = -1;   // >= 0: additionally demand the same bits as this variant
  bool w_tiled = false;    // the weight is handed over K-tile-contiguous (PfdGemmDesc.w_tiled)
  bool gn_stats = false;   // the launch emits GroupNorm statistics (PfdGemmDesc.gn_out): checked against the sums of what it stored
  int gnf = 0;             // 1 / 2: GroupNorm(+SiLU) fused into the split-K reduction (PfdGemmDesc.gnf_y), raw tensor skipped / kept
  int res_rows = 0;        // > 0: the residual holds that many rows, read with one wrap (PfdGemmDesc.res_rows)
};

static int run_variant(const Case& c, int variant, const std::vector<h16>& A, const std::vector<h16>& A2, const std::vector<h16>& Wt,
                       const std::vector<h16>& bias, const std::vector<h16>& rv, const std::vector<h16>& R, int M, int K, int Ho, int Wo,
                       std::vector<h16>& C, std::vector<float>& ws, std::vector<float>* gn = nullptr, std::vector<h16>* gnf_y = nullptr,
                       const std::vector<h16>* gnf_gb = nullptr, int gnf_skip_raw = 0) {
  const bool conv = c.ksize > 0;
  PfdGemmDesc d;
  memset(&d, 0, sizeof(d));
  d.M = M; d.N = c.N; d.K = K;
  std::vector<h16> Wtiled;
  if (c.w_tiled) {   // (n, k) -> (((n / 160) * (K / 64) + k / 64) * 160 + n % 160) * 64 + k % 64
    Wtiled.resize(Wt.size());
    for (int n = 0; n < c.N; ++n)
      for (int k = 0; k < K; ++k) Wtiled[(((size_t)(n / 160) * (K / 64) + k / 64) * 160 + n % 160) * 64 + k % 64] = Wt[(size_t)n * K + k];
  }
  d.A = A.data(); d.W = c.w_tiled ? Wtiled.data() : Wt.data(); d.bias = bias.data(); d.C = C.data();
  d.w_tiled = c.w_tiled ? 1 : 0;
  d.R = c.res ? R.data() : nullptr;
  d.rowvec = c.rowvec ? rv.data() : nullptr;
  d.lda = conv ? c.Cin : (c.k_split ? c.k_split : K);
  d.ldw = K; d.ldc = c.N; d.ldr = c.N; d.ldrv = c.N;
  d.rows_per_rv = conv ? Ho * Wo : 64;
  d.act = c.act;
  d.ksize = c.ksize; d.stride = c.stride; d.pad = c.pad; d.ups = c.ups;
  d.B = c.B; d.H = c.H; d.Wd = c.W; d.Cin = c.Cin; d.Ho = Ho; d.Wo = Wo;
  if (c.k_split) { d.k_split = c.k_split; d.A2 = A2.data(); d.lda2 = K - c.k_split; }
  d.zero_rows = c.zero_rows;
  d.res_rows = c.res_rows;
  d.ws = ws.data(); d.ws_bytes = ws.size() * sizeof(float);
  if (gn) d.gn_out = gn->data();
  if (gnf_y) {
    d.gnf_gamma = gnf_gb->data(); d.gnf_beta = gnf_gb->data() + c.N; d.gnf_y = gnf_y->data(); d.gnf_ldy = c.N; d.gnf_eps = 1e-5f;
    d.gnf_act = PFD_ACT_SILU; d.gnf_rows = conv ? Ho * Wo : M; d.gnf_skip_raw = gnf_skip_raw;
  }
  return pfd_gemm160_try(&d, variant, c.splits, nullptr);
}

static int run_case(const Case& c) {
  const bool conv = c.ksize > 0;
  int M = c.M, K = c.K, Ho = 0, Wo = 0;
  if (conv) {
    const int Hin = c.ups ? 2 * c.H : c.H, Win = c.ups ? 2 * c.W : c.W;
    Ho = (Hin + 2 * c.pad - c.ksize) / c.stride + 1;
    Wo = (Win + 2 * c.pad - c.ksize) / c.stride + 1;
    M = c.B * Ho * Wo;
    K = c.ksize * c.ksize * c.Cin;
  }
  const int N = c.N;
  const long a_rows = conv ? (long)c.B * c.H * c.W : M, lda_full = conv ? c.Cin : K;
  auto Afull = rand_h((size_t)a_rows * lda_full, 1.f);
  auto Wt = rand_h((size_t)N * K, 1.7f / sqrtf((float)K)), bias = rand_h(N, 0.5f), R = rand_h((size_t)M * N, 1.f);
  const int rows_per_rv = conv ? Ho * Wo : 64;
  auto rv = rand_h((size_t)((M + rows_per_rv - 1) / rows_per_rv) * N, 0.5f);
  for (long r = 0; r < c.zero_rows; ++r)
    for (long k = 0; k < lda_full; ++k) Afull[r * lda_full + k] = (h16)0;
  // device operands: rows below zero_rows are not stored; columns >= k_split live in a second buffer
  const int K1 = c.k_split ? c.k_split : (int)lda_full, K2 = c.k_split ? K - c.k_split : 0;
  const long Mz = a_rows - c.zero_rows;
  std::vector<h16> A1((size_t)Mz * K1 + 64), A2((size_t)Mz * std::max(K2, 1) + 64);
  for (long m = 0; m < Mz; ++m) {
    for (int k = 0; k < K1; ++k) A1[(size_t)m * K1 + k] = Afull[(size_t)(m + c.zero_rows) * lda_full + k];
    for (int k = 0; k < K2; ++k) A2[(size_t)m * K2 + k] = Afull[(size_t)(m + c.zero_rows) * lda_full + K1 + k];
  }
  std::vector<h16> C((size_t)M * N, (h16)-77.f);
  std::vector<float> ws((size_t)8 * M * N + 64);
  g_err.clear();
  std::vector<float> gn1((size_t)(M / 64 + 1) * (N / 160) * 32, -1.f), gn2 = gn1;
  const int rc = run_variant(c, c.variant, A1, A2, Wt, bias, rv, R, M, K, Ho, Wo, C, ws, c.gn_stats ? &gn1 : nullptr);
  if (rc != 0) { printf("FAIL %-70s rc=%d %s\n", c.what, rc, g_err.c_str()); return 1; }
  // double-precision reference
  double max_err = 0, max_ref = 0;
  for (int m = 0; m < M; ++m) {
    int b = 0, oy = 0, ox = 0;
    if (conv) { b = m / (Ho * Wo); oy = (m % (Ho * Wo)) / Wo; ox = m % Wo; }
    for (int n = 0; n < N; ++n) {
      double s = 0;
      if (!conv) {
        for (int k = 0; k < K; ++k) s += (double)Afull[(size_t)m * K + k] * (double)Wt[(size_t)n * K + k];
      } else {
        const int Hin = c.ups ? 2 * c.H : c.H, Win = c.ups ? 2 * c.W : c.W;
        for (int ky = 0; ky < c.ksize; ++ky)
          for (int kx = 0; kx < c.ksize; ++kx) {
            int iy = oy * c.stride + ky - c.pad, ix = ox * c.stride + kx - c.pad;
            if (iy < 0 || iy >= Hin || ix < 0 || ix >= Win) continue;
            if (c.ups) { iy /= 2; ix /= 2; }
            const h16* ap = &Afull[(((size_t)b * c.H + iy) * c.W + ix) * c.Cin];
            const h16* wp = &Wt[(size_t)n * K + (ky * c.ksize + kx) * c.Cin];
            for (int ci = 0; ci < c.Cin; ++ci) s += (double)ap[ci] * (double)wp[ci];
          }
      }
      s += (double)bias[n];
      if (c.rowvec) s += (double)rv[(size_t)(m / rows_per_rv) * N + n];
      if (c.act == PFD_ACT_SILU) s = s / (1.0 + exp(-s));
      else if (c.act == PFD_ACT_RELU) s = s > 0 ? s : 0;
      if (c.res) s += (double)R[(size_t)(c.res_rows > 0 && m >= c.res_rows ? m - c.res_rows : m) * N + n];
      max_ref = std::max(max_ref, fabs(s));
      max_err = std::max(max_err, fabs(s - (double)C[(size_t)m * N + n]));
    }
  }
  const bool ok = max_err <= 4e-3 * std::max(1.0, max_ref);
  int fails = ok ? 0 : 1;
  std::string extra;
  if (ok && c.base_variant >= 0) {
    std::vector<h16> C2((size_t)M * N, (h16)-55.f);
    const int rc2 = run_variant(c, c.base_variant, A1, A2, Wt, bias, rv, R, M, K, Ho, Wo, C2, ws);
    size_t nd = 0;
    for (size_t i = 0; i < C.size(); ++i) nd += memcmp(&C[i], &C2[i], sizeof(h16)) != 0;
    if (rc2 != 0 || nd) { fails = 1; extra = " | vs variant " + std::to_string(c.base_variant) + ": rc " + std::to_string(rc2) + ", " + std::to_string(nd) + " elements differ"; }
    else extra = " | == variant " + std::to_string(c.base_variant) + " bitwise";
  }
  if (ok && c.gn_stats) {
    // the statistics are the sums of the f16 values the launch stored, per 64-row slab and group of N / 32 channels
    const int cpg = N / 32, tn = N / 160, ngl = 160 / cpg;
    double worst = 0;
    for (int sl = 0; sl < M / 64; ++sl)
      for (int t = 0; t < tn; ++t)
        for (int gl = 0; gl < ngl; ++gl) {
          double a = 0, q = 0;
          for (int r = 0; r < 64; ++r)
            for (int cc = 0; cc < cpg; ++cc) { const double v = (double)C[(size_t)(sl * 64 + r) * N + t * 160 + gl * cpg + cc]; a += v; q += v * v; }
          const size_t o = (((size_t)sl * tn + t) * 16 + gl) * 2;
          worst = std::max(worst, std::max(fabs(a - gn1[o]) / (1 + fabs(a)), fabs(q - gn1[o + 1]) / (1 + fabs(q))));
        }
    if (worst > 1e-3) fails = 1;
    extra += std::string(" | statistics err ") + std::to_string(worst);
  }
  if (ok && c.gnf) {
    // the same launch with the fused GroupNorm request: raw result (when kept) bit for bit the plain reduction's, the normalised
    // tensor against a double-precision GroupNorm(32) + SiLU of that raw result
    auto gb = rand_h((size_t)2 * N, 1.f);
    std::vector<h16> C2((size_t)M * N, (h16)-55.f), Y((size_t)M * N, (h16)-33.f);
    const int rc2 = run_variant(c, c.variant, A1, A2, Wt, bias, rv, R, M, K, Ho, Wo, C2, ws, nullptr, &Y, &gb, c.gnf == 1 ? 1 : 0);
    size_t nd = 0, untouched = 0;
    for (size_t i = 0; i < C.size(); ++i) { nd += memcmp(&C[i], &C2[i], sizeof(h16)) != 0; untouched += (float)C2[i] == -55.f; }
    const int HW = conv ? Ho * Wo : M, Bn = M / HW, cpg = N / 32;
    double werr = 0;
    for (int b = 0; b < Bn; ++b)
      for (int g = 0; g < 32; ++g) {
        double a = 0, q = 0;
        for (int p = 0; p < HW; ++p)
          for (int ch = g * cpg; ch < (g + 1) * cpg; ++ch) { const double v = (double)C[((size_t)b * HW + p) * N + ch]; a += v; q += v * v; }
        const double n = (double)HW * cpg, mean = a / n, rstd = 1.0 / sqrt(q / n - mean * mean + 1e-5);
        for (int p = 0; p < HW; ++p)
          for (int ch = g * cpg; ch < (g + 1) * cpg; ++ch) {
            double v = ((double)C[((size_t)b * HW + p) * N + ch] - mean) * rstd * (double)gb[ch] + (double)gb[N + ch];
            v = v / (1.0 + exp(-v));
            werr = std::max(werr, fabs(v - (double)Y[((size_t)b * HW + p) * N + ch]) / std::max(1.0, fabs(v)));
          }
      }
    const bool raw_ok = c.gnf == 2 ? nd == 0 : untouched == C2.size();
    if (rc2 != 0 || !raw_ok || werr > 4e-3) fails = 1;
    extra += " | fused GroupNorm: rc " + std::to_string(rc2) + (c.gnf == 2 ? ", raw differs in " + std::to_string(nd) : ", raw elements written " + std::to_string(C2.size() - untouched)) +
             ", normalised err " + std::to_string(werr);
    if (!fails) {   // and a request the library must decline without launching: the same problem unsplit
      Case u = c; u.splits = 1;
      std::vector<h16> C3((size_t)M * N, (h16)-55.f), Y3((size_t)M * N, (h16)-33.f);
      const int rc3 = run_variant(u, c.variant, A1, A2, Wt, bias, rv, R, M, K, Ho, Wo, C3, ws, nullptr, &Y3, &gb, 0);
      size_t touched = 0;
      for (size_t i = 0; i < C3.size(); ++i) touched += ((float)C3[i] != -55.f) + ((float)Y3[i] != -33.f);
      if (rc3 != 1 || touched) { fails = 1; extra += " | UNSPLIT REQUEST NOT DECLINED (rc " + std::to_string(rc3) + ", " + std::to_string(touched) + " written)"; }
      else extra += " | unsplit request declined, nothing written";
    }
  }
  printf("%s %-70s max err %.2e (max |ref| %.2f)%s\n", fails ? "FAIL" : "ok  ", c.what, max_err, max_ref, extra.c_str());
  fflush(stdout);
  return fails;
}

int main(int argc, char** argv) {
  std::vector<Case> cases;
  auto lin = [&](const char* w, int M, int N, int K, int v, int sp, bool res, int base) {
    Case c{w, M, N, K, v, sp}; c.res = res; c.base_variant = base; cases.push_back(c); return &cases.back();
  };
  auto conv = [&](const char* w, int N, int v, int sp, int ks, int st, int pad, int ups, int B, int H, int W, int Cin, bool res, int base) {
    Case c{w, 0, N, 0, v, sp}; c.res = res; c.ksize = ks; c.stride = st; c.pad = pad; c.ups = ups; c.B = B; c.H = H; c.W = W; c.Cin = Cin;
    c.base_variant = base; cases.push_back(c); return &cases.back();
  };
  // the hardware-validated LDS-ring kernels (sanity of the emulation itself)
  lin("variant 23 (64x160, 4-stage LDS ring) 200x160x512", 200, 160, 512, 23, 1, true, -1);
  lin("variant 83 (128x160, 8 waves, 3-stage LDS ring) 200x160x320", 200, 160, 320, 83, 1, false, -1);
  // residual stored once for a doubled batch (PfdGemmDesc.res_rows, round 5): store pass and plain split-K reduction, + zero rows
  { auto c = lin("residual read with one wrap (res_rows 128), 8-wave 128-row tile", 256, 160, 256, 82, 1, true, -1); c->res_rows = 128; }
  { auto c = lin("residual read with one wrap + zero rows (the CFG re-join), 64-row ring", 256, 320, 512, 23, 1, true, -1); c->res_rows = 128; c->zero_rows = 128; }
  { auto c = lin("residual read with one wrap, split-K 2 (plain reduction)", 256, 160, 1024, 23, 2, true, -1); c->res_rows = 128; }
  // split-K reduction that also emits the GroupNorm statistics (three row sweeps in flight)
  { auto c = lin("split-K 4 + GroupNorm statistics (N 320: cpg 10), residual", 128, 320, 1024, 23, 4, true, -1); c->gn_stats = true; }
  { auto c = conv("split-K 2 conv 8x8x128 -> 1280 + statistics (cpg 40), SiLU-free", 1280, 83, 2, 3, 1, 1, 0, 2, 8, 8, 128, true, -1); c->gn_stats = true; }
  // split-K reduction that also NORMALISES (PfdGemmDesc.gnf_y, round 5): 8 samples so that B * 32 >= 128 blocks as the host demands
  { auto c = conv("fused GroupNorm: split-K 2 conv 4x4x128 -> 1280 (cpg 40), row vector, raw skipped", 1280, 83, 2, 3, 1, 1, 0, 8, 4, 4, 128, false, -1); c->rowvec = true; c->gnf = 1; }
  { auto c = conv("fused GroupNorm: split-K 3 conv 4x4x192 -> 1280, residual, raw kept", 1280, 23, 3, 3, 1, 1, 0, 4, 4, 4, 192, true, -1); c->gnf = 2; }
  { auto c = conv("fused GroupNorm: split-K 2 patch conv 16x16x128 -> 1280, raw skipped", 1280, 98, 2, 3, 1, 1, 0, 4, 16, 16, 128, false, -1); c->rowvec = true; c->gnf = 1; }
  // the barrier forms of the patch kernel (sanity of the emulation on the hardware-validated kernels)
  conv("variant 98 patch conv 16x16 (loader waves, barrier per tap), 2 channel blocks", 160, 98, 1, 3, 1, 1, 0, 1, 16, 16, 128, true, -1);
  conv("variant 96 patch conv 16x16 (3-stage weight ring), 2 channel blocks", 160, 96, 1, 3, 1, 1, 0, 1, 16, 16, 128, true, 98);
  { auto c = conv("variant 96 patch conv 16x16, K-tile-contiguous weights, two column tiles", 320, 96, 1, 3, 1, 1, 0, 1, 16, 16, 128, true, -1); c->w_tiled = true; }
  { auto c = conv("variant 83 implicit-GEMM conv, K-tile-contiguous weights, two column tiles", 320, 83, 1, 3, 1, 1, 0, 1, 8, 8, 128, false, -1); c->w_tiled = true; }
  conv("variant 99 patch conv 16x16 (8-wave form), 2 channel blocks", 160, 99, 1, 3, 1, 1, 0, 1, 16, 16, 128, true, 98);
  int fails = 0, n = 0;
  for (const auto& c : cases) {
    if (argc > 1) {
      bool hit = false;
      for (int i = 1; i < argc; ++i) hit = hit || strstr(c.what, argv[i]);
      if (!hit) continue;
    }
    fails += run_case(c);
    ++n;
  }
  printf("%d cases, %d failed\n", n, fails);
  return fails;
}
