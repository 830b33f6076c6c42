// hdu_core.cpp -- error state and identification of the C-ABI library.
#include "hdu_host.h"

#include <cstdio>
#include <cstring>

static thread_local char g_err[512] = "";

int hdu_set_error(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

int hdu_check_launch(const char* what) {
#ifdef HDU_EMU
  (void)what;
  return 0;
#else
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return HDU_ERR_LAUNCH;
  }
  return 0;
#endif
}

extern "C" const char* hdu_last_error(void) { return g_err; }
extern "C" const char* hdu_backend(void) {
#ifdef HDU_EMU
  return "emu-x86";
#else
  return "hip-gfx950";
#endif
}
extern "C" int hdu_abi_version(void) { return HDU_ABI_VERSION; }
extern "C" size_t hdu_sizeof_conv_desc(void) { return sizeof(hdu_conv_desc); }
