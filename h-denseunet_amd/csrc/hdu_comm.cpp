// hdu_comm.cpp -- thin RCCL wrappers behind the C-ABI (include/hdu.h, "collectives"): gradient all-reduce and the
// depth-neighbour exchange of a sharded volume over xGMI, on the caller's stream, with no torch in the signature.
//
// RCCL is bound at run time (dlopen of the librccl a process already holds -- PyTorch-ROCm ships one -- or of the ROCm
// installation's), so that libhdu.so itself has no link-time dependency on it: single-GPU users and the CPU test tier
// never touch it.  Only the stable NCCL-API entry points are used (rccl.h:187-933: ncclGetUniqueId, ncclCommInitRank,
// ncclAllReduce, ncclSend / ncclRecv inside ncclGroupStart / ncclGroupEnd, ncclCommDestroy); their prototypes are
// restated locally because the header is not needed for anything else.
#include "hdu_host.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#ifndef HDU_EMU
#include <dlfcn.h>

typedef int ncclResult_t;                       // ncclSuccess == 0
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;   // NCCL_UNIQUE_ID_BYTES (rccl.h:40-43)
enum { HDU_NCCL_SUM = 0, HDU_NCCL_INT8 = 0, HDU_NCCL_FLOAT32 = 7 };   // rccl.h:448-466

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static void rccl_bind(RcclApi& api) {
  // 1. the symbols a process already holds (PyTorch-ROCm maps its own, possibly renamed, librccl with global visibility):
  //    RTLD_DEFAULT finds them whatever the file is called, and no second copy of the library enters the process
  bool from_process = dlsym(RTLD_DEFAULT, "ncclCommInitRank") != nullptr;
  void* h = RTLD_DEFAULT;
  if (!from_process) {
    // 2. an explicit library (HDU_RCCL_LIB), then the usual names: already-mapped copies first (RTLD_NOLOAD), then a load
    const char* env = getenv("HDU_RCCL_LIB");
    const char* names[] = {env ? env : "librccl.so", "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    h = nullptr;
    for (const char* n : names) {
      h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
      if (h) break;
    }
    for (size_t i = 0; !h && i < sizeof(names) / sizeof(names[0]); ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
  }
#define HDU_SYM(field, name) *(void**)(&api.field) = dlsym(h, name)
  HDU_SYM(GetUniqueId, "ncclGetUniqueId");
  HDU_SYM(CommInitRank, "ncclCommInitRank");
  HDU_SYM(CommDestroy, "ncclCommDestroy");
  HDU_SYM(AllReduce, "ncclAllReduce");
  HDU_SYM(Send, "ncclSend");
  HDU_SYM(Recv, "ncclRecv");
  HDU_SYM(GroupStart, "ncclGroupStart");
  HDU_SYM(GroupEnd, "ncclGroupEnd");
  HDU_SYM(GetErrorString, "ncclGetErrorString");
#undef HDU_SYM
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.Send || !api.Recv || !api.GroupStart ||
      !api.GroupEnd) {
    if (!from_process) dlclose(h);
    return;
  }
  api.handle = from_process ? (void*)&api : h;        // non-null = bound
}

static RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;                          // (two threads binding at the same time)
  std::call_once(once, [] { rccl_bind(api); });
  return api.handle ? &api : nullptr;
}

struct hdu_comm {
  ncclComm_t comm;
  int rank, world;
};

static int rccl_fail(RcclApi* a, ncclResult_t r, const char* what) {
  char msg[256];
  snprintf(msg, sizeof(msg), "%s: RCCL error %d (%s)", what, (int)r, a->GetErrorString ? a->GetErrorString(r) : "?");
  return hdu_set_error(HDU_ERR_LAUNCH, msg);
}

extern "C" int hdu_comm_unique_id(void* id128) {
  RcclApi* a = rccl_api();
  if (!a) return hdu_set_error(HDU_ERR_ARG, "comm: librccl could not be loaded");
  if (!id128) return hdu_set_error(HDU_ERR_ARG, "comm_unique_id: null buffer");
  ncclUniqueId id;
  if (ncclResult_t r = a->GetUniqueId(&id)) return rccl_fail(a, r, "comm_unique_id");
  memcpy(id128, &id, sizeof(id));
  return 0;
}

extern "C" int hdu_comm_init(hdu_comm** out, int rank, int world, const void* id128) {
  RcclApi* a = rccl_api();
  if (!a) return hdu_set_error(HDU_ERR_ARG, "comm: librccl could not be loaded");
  if (!out || !id128 || world <= 0 || rank < 0 || rank >= world) return hdu_set_error(HDU_ERR_ARG, "comm_init: bad args");
  // the communicator binds to the CALLING thread's current HIP device (one process per GPU: the caller has selected it,
  // e.g. torch.cuda.set_device(LOCAL_RANK)); a process that has no usable device is refused here rather than inside RCCL
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) {
    (void)hipGetLastError();
    return hdu_set_error(HDU_ERR_ARG, "comm_init: no current HIP device (select the rank's GPU before creating the communicator)");
  }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  hdu_comm* c = new hdu_comm{nullptr, rank, world};
  if (ncclResult_t r = a->CommInitRank(&c->comm, world, id, rank)) {
    delete c;
    return rccl_fail(a, r, "comm_init");
  }
  *out = c;
  return 0;
}

extern "C" int hdu_comm_destroy(hdu_comm* c) {
  RcclApi* a = rccl_api();
  if (!c) return 0;
  if (a && c->comm) a->CommDestroy(c->comm);
  delete c;
  return 0;
}

extern "C" int hdu_comm_allreduce_f32(hdu_comm* c, float* buf, int64_t n, void* stream) {
  RcclApi* a = rccl_api();
  if (!a || !c || !buf || n < 0) return hdu_set_error(HDU_ERR_ARG, "comm_allreduce: bad args");
  if (n == 0) return 0;
  if (ncclResult_t r = a->AllReduce(buf, buf, (size_t)n, HDU_NCCL_FLOAT32, HDU_NCCL_SUM, c->comm, (hipStream_t)stream))
    return rccl_fail(a, r, "comm_allreduce");
  return 0;
}

extern "C" int hdu_comm_sendrecv(hdu_comm* c, int lo_rank, const void* send_lo, void* recv_lo, int hi_rank, const void* send_hi,
                                 void* recv_hi, size_t bytes, void* stream) {
  RcclApi* a = rccl_api();
  if (!a || !c) return hdu_set_error(HDU_ERR_ARG, "comm_sendrecv: bad args");
  if (bytes == 0 || (lo_rank < 0 && hi_rank < 0)) return 0;
  if ((lo_rank >= 0 && (!send_lo || !recv_lo || lo_rank >= c->world)) || (hi_rank >= 0 && (!send_hi || !recv_hi || hi_rank >= c->world)))
    return hdu_set_error(HDU_ERR_ARG, "comm_sendrecv: a neighbour needs both a send and a receive buffer");
  hipStream_t s = (hipStream_t)stream;
  ncclResult_t r = a->GroupStart();
  if (!r && lo_rank >= 0) r = a->Send(send_lo, bytes, HDU_NCCL_INT8, lo_rank, c->comm, s);
  if (!r && lo_rank >= 0) r = a->Recv(recv_lo, bytes, HDU_NCCL_INT8, lo_rank, c->comm, s);
  if (!r && hi_rank >= 0) r = a->Send(send_hi, bytes, HDU_NCCL_INT8, hi_rank, c->comm, s);
  if (!r && hi_rank >= 0) r = a->Recv(recv_hi, bytes, HDU_NCCL_INT8, hi_rank, c->comm, s);
  const ncclResult_t e = a->GroupEnd();
  if (r || e) return rccl_fail(a, r ? r : e, "comm_sendrecv");
  return 0;
}

#else   // ---- x86 emulator build of the kernel sources: there is no RCCL; the CPU tests use gloo through torch.distributed
struct hdu_comm { int unused; };
extern "C" int hdu_comm_unique_id(void*) { return hdu_set_error(HDU_ERR_ARG, "comm: not available in the emulator build"); }
extern "C" int hdu_comm_init(hdu_comm**, int, int, const void*) { return hdu_set_error(HDU_ERR_ARG, "comm: not available in the emulator build"); }
extern "C" int hdu_comm_destroy(hdu_comm*) { return 0; }
extern "C" int hdu_comm_allreduce_f32(hdu_comm*, float*, int64_t, void*) { return hdu_set_error(HDU_ERR_ARG, "comm: not available in the emulator build"); }
extern "C" int hdu_comm_sendrecv(hdu_comm*, int, const void*, void*, int, const void*, void*, size_t, void*) {
  return hdu_set_error(HDU_ERR_ARG, "comm: not available in the emulator build");
}
#endif
