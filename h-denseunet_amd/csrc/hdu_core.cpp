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

// ---------------------------------------------------------------- launch profiler (include/hdu.h: hdu_profile_*)
#include <cxxabi.h>
#include <dlfcn.h>

#include <cstdlib>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

std::atomic<int> g_hdu_prof_on{0};

namespace {
struct ProfRec {
  const void* addr;
#ifndef HDU_EMU
  hipEvent_t e0, e1;
#endif
};
std::vector<ProfRec> g_prof;
size_t g_prof_cap = 0;
std::mutex g_prof_mutex;
#ifndef HDU_EMU
std::vector<hipEvent_t> g_prof_pool;      // events are created once and reused by later profiling windows
#endif

// kernel address -> "conv_igemm_dma_kernel<unsigned short, 128, 96, 4, 1, true, false, 0>": the symbol the address belongs
// to (the kernels are template instantiations with default visibility: they are in the dynamic symbol table), demangled,
// without the return type and the parameter list -- spelled as rocprofv3 prints the kernel
std::string kernel_of(const void* addr) {
  Dl_info info;
  if (!addr || !dladdr(addr, &info) || !info.dli_sname) return "unknown_kernel";
  int status = 0;
  char* dm = abi::__cxa_demangle(info.dli_sname, nullptr, nullptr, &status);
  std::string s = (status == 0 && dm) ? dm : info.dli_sname;
  free(dm);
  if (s.compare(0, 5, "void ") == 0) s = s.substr(5);
  const std::string stub = "__device_stub__";
  size_t q = s.find(stub);
  if (q != std::string::npos) s.erase(q, stub.size());
  int depth = 0;
  for (size_t i = 0; i < s.size(); ++i) {            // cut at the '(' of the parameter list (outside template brackets)
    if (s[i] == '<') ++depth;
    else if (s[i] == '>') --depth;
    else if (s[i] == '(' && depth == 0) { s.resize(i); break; }
  }
  return s;
}
}  // namespace

#ifndef HDU_EMU
int hdu_prof_next(const void* kernel_addr, hipStream_t stream, hipEvent_t* e0, hipEvent_t* e1) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return 0;                                  // a captured launch takes the plain path
  }
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  if (g_prof.size() >= g_prof_cap) return 0;
  const size_t i = g_prof.size();
  while (g_prof_pool.size() < 2 * (i + 1)) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return 0;
    g_prof_pool.push_back(e);
  }
  ProfRec r;
  r.addr = kernel_addr;
  r.e0 = g_prof_pool[2 * i];
  r.e1 = g_prof_pool[2 * i + 1];
  g_prof.push_back(r);
  *e0 = r.e0;
  *e1 = r.e1;
  return 1;
}
#else
int hdu_prof_note(const void* kernel_addr) {
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  if (g_prof.size() >= g_prof_cap) return 0;
  ProfRec r;
  r.addr = kernel_addr;
  g_prof.push_back(r);
  return 1;
}
#endif

extern "C" int hdu_profile_begin(int max_records) {
  if (max_records <= 0) return hdu_set_error(HDU_ERR_ARG, "profile_begin: max_records must be positive");
  if (g_hdu_prof_on) return hdu_set_error(HDU_ERR_ARG, "profile_begin: a profiling window is already armed (end it first)");
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  g_prof.clear();
  g_prof.reserve((size_t)max_records);
  g_prof_cap = (size_t)max_records;
  g_hdu_prof_on = 1;
  return 0;
}

extern "C" int hdu_profile_count(void) {
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  return (int)g_prof.size();
}

extern "C" int hdu_profile_end(void) {
  g_hdu_prof_on = 0;
  std::lock_guard<std::mutex> lock(g_prof_mutex);
#ifndef HDU_EMU
  // records may come from several threads / streams: EVERY record's stop event is waited for, not only the last one's
  for (const ProfRec& r : g_prof)
    if (hipEventSynchronize(r.e1) != hipSuccess) {
      (void)hipGetLastError();
      return hdu_set_error(HDU_ERR_LAUNCH, "profile_end: waiting for a recorded kernel failed");
    }
#endif
  return (int)g_prof.size();
}

extern "C" int hdu_profile_get(int i, char* name_buf, size_t buflen, float* ms) {
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  if (i < 0 || (size_t)i >= g_prof.size() || !name_buf || buflen < 2 || !ms)
    return hdu_set_error(HDU_ERR_ARG, "profile_get: bad index / buffer");
  const std::string k = kernel_of(g_prof[(size_t)i].addr);
  snprintf(name_buf, buflen, "%s", k.c_str());
  *ms = 0.f;
#ifndef HDU_EMU
  if (hipEventElapsedTime(ms, g_prof[(size_t)i].e0, g_prof[(size_t)i].e1) != hipSuccess) {
    (void)hipGetLastError();
    return hdu_set_error(HDU_ERR_LAUNCH, "profile_get: hipEventElapsedTime failed");
  }
#endif
  return 0;
}
