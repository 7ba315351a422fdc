// ABI version + error plumbing shared by every entry point of libpfd_hip.so.
#include <string.h>

#include "pfd_common.h"

static thread_local char g_err[256] = "";

void pfd_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

int pfd_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return PFD_OK;
  char buf[256];
  snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
  pfd_set_error(buf);
  return PFD_ELAUNCH;
}

extern "C" int pfd_abi_version(void) { return PFD_ABI_VERSION; }
extern "C" const char* pfd_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------
// Optional in-process kernel timing (bench.py's roofline leg): when enabled, instrumented launch
// sites bracket their kernel with a HIP event pair recorded on the launch stream and tally the
// algorithmic flops/bytes of the launch.  Events live in a ring; a slot is harvested (elapsed time
// added to its bucket) when the ring wraps or when the counters are read.  Off by default: the
// cost is one predictable branch per launch.
// ------------------------------------------------------------------------------------------------
#include <mutex>
#include <vector>

namespace {
constexpr int kProfBuckets = 22;
constexpr int kProfRing = 16384;
struct ProfSlot {
  hipEvent_t a = nullptr, b = nullptr;
  int bucket = -1;
};
struct ProfState {
  bool on = false;
  std::vector<ProfSlot> ring;
  size_t head = 0;  // next slot to use
  double ms[kProfBuckets] = {0};
  double flops[kProfBuckets] = {0};
  double bytes[kProfBuckets] = {0};
  long launches[kProfBuckets] = {0};
  int open_slot = -1;
};
ProfState g_prof;
std::mutex g_prof_mu;
const char* kBucketNames[kProfBuckets] = {
    "gemm_conv_kernel<1,1,false>", "gemm_conv_kernel<1,2,false>", "gemm_conv_kernel<2,1,false>",
    "gemm_conv_kernel<2,2,false>", "gemm_conv_kernel<1,1,true>",  "gemm_conv_kernel<1,2,true>",
    "gemm_conv_kernel<2,1,true>",  "gemm_conv_kernel<2,2,true>",  "attention_kernel",
    "swin_attn_kernel",            "groupnorm(stats+finalize+apply)", "layernorm_kernel",
    "gemm160_kernel<4,4>(256x160)", "gemm160_kernel<2,4>(128x160)", "gemm160_kernel<2,2>(64x160)",
    "elementwise / glue",          "gemm160_kernel<4,4,conv>(256x160)", "gemm160_kernel<2,4,conv>(128x160)",
    "gemm160_kernel<2,2,conv>(64x160)", "conv3x3_patch_kernel(256x160)", "conv3x3_narrow_kernel(N<=16)",
    "splitk_reduce(+epilogue / GroupNorm)"};

void harvest(ProfSlot& s) {
  if (s.bucket < 0) return;
  float t = 0.f;
  if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&t, s.a, s.b) == hipSuccess)
    g_prof.ms[s.bucket] += t;
  s.bucket = -1;
}
}  // namespace

bool pfd_prof_on() { return g_prof.on; }

void pfd_prof_begin(int bucket, double flops, double bytes, hipStream_t stream) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof.on || bucket < 0 || bucket >= kProfBuckets) return;
  if (g_prof.ring.empty()) g_prof.ring.resize(kProfRing);
  ProfSlot& s = g_prof.ring[g_prof.head];
  harvest(s);
  if (!s.a) {
    (void)hipEventCreate(&s.a);
    (void)hipEventCreate(&s.b);
  }
  s.bucket = bucket;
  g_prof.flops[bucket] += flops;
  g_prof.bytes[bucket] += bytes;
  g_prof.launches[bucket] += 1;
  g_prof.open_slot = (int)g_prof.head;
  g_prof.head = (g_prof.head + 1) % kProfRing;
  (void)hipEventRecord(s.a, stream);
}

void pfd_prof_end(hipStream_t stream) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof.on || g_prof.open_slot < 0) return;
  (void)hipEventRecord(g_prof.ring[g_prof.open_slot].b, stream);
  g_prof.open_slot = -1;
}

extern "C" int pfd_prof_enable(int32_t on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& s : g_prof.ring) harvest(s);
  for (int i = 0; i < kProfBuckets; ++i) {
    g_prof.ms[i] = g_prof.flops[i] = g_prof.bytes[i] = 0;
    g_prof.launches[i] = 0;
  }
  g_prof.on = on != 0;
  g_prof.open_slot = -1;
  return PFD_OK;
}

extern "C" int pfd_prof_read(int32_t bucket, double* ms, int64_t* launches, double* flops, double* bytes) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (bucket < 0 || bucket >= kProfBuckets) return PFD_EINVAL;
  for (auto& s : g_prof.ring) harvest(s);
  if (ms) *ms = g_prof.ms[bucket];
  if (launches) *launches = g_prof.launches[bucket];
  if (flops) *flops = g_prof.flops[bucket];
  if (bytes) *bytes = g_prof.bytes[bucket];
  return PFD_OK;
}

extern "C" const char* pfd_prof_bucket_name(int32_t bucket) {
  return (bucket >= 0 && bucket < kProfBuckets) ? kBucketNames[bucket] : "";
}
extern "C" int pfd_prof_num_buckets(void) { return kProfBuckets; }
