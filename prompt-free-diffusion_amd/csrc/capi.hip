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
