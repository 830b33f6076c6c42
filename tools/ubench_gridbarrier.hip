// ubench_gridbarrier.hip -- what does an in-kernel grid barrier cost on MI355X next to the kernel boundary it would replace?
//
// DESIGN.md section 9.1 argues that the "one persistent launch per dense layer / block" design trades every
// kernel boundary (measured 4.1-4.3 us begin-to-end for a launch with next to no work inside the captured step) for a grid barrier
// that has to do the same cache maintenance across 8 non-coherent XCD L2s -- without a measurement of that barrier.  This is the
// measurement: 256 co-resident workgroups (one per CU) of 256 or 512 threads run ITERS phases; per phase every workgroup dirties
// 0 / 64 KB / 1 MB of its own slab with plain 16-byte stores, then meets the others at
//   (flat) one monotonic device counter: lane 0 agent-scope release fence + drained vmcnt -> relaxed atomic add -> relaxed sc1 poll
//          with s_sleep -> agent-scope acquire fence -> __syncthreads()
//   (xcd)  the XCD-hierarchical form of MI355X_MICROARCH.md "barrier-xcd": per-XCD arrival counter; the XCD's last arriver does
//          the ONE release fence of its L2, arrives at the top counter, waits for the 8 leaders, bumps its XCD's generation word;
//          every workgroup polls its XCD's generation and does one acquire fence
// and after the barrier READS what its neighbour (blockIdx + 1: another XCD) wrote in the phase before and checks every word
// (L1-warm: the same addresses were read the phase before), so a barrier that is fast because it is wrong is reported as wrong.
// Reported: us per phase with the barrier minus us per phase of the same kernel without it (the dirtying alone), next to the cost of
// a dependent kernel boundary measured the same way (a chain of ITERS launches of the dirtying body, one phase each, in a hipGraph).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_gridbarrier tools/ubench_gridbarrier.hip && tools/ubench_gridbarrier
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#define CHECK(x)                                                                                  \
  do {                                                                                            \
    hipError_t e_ = (x);                                                                          \
    if (e_ != hipSuccess) {                                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                   \
      exit(1);                                                                                    \
    }                                                                                             \
  } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ unsigned ld_relaxed(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned add_relaxed(unsigned* p, unsigned v) {
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct Sync {
  unsigned* flat;      // [1]
  unsigned* xcd_cnt;   // [8 * 32] one counter per XCD, 128 B apart
  unsigned* xcd_gen;   // [8 * 32]
  unsigned* top;       // [1]
  unsigned* timeout;   // [1] set when a spin gave up
};

enum { MODE_NONE = 0, MODE_FLAT = 1, MODE_XCD = 2 };

// one phase of work: the workgroup dirties `bytes` of its own slab (value = phase tag), 16 B per lane per store
__device__ __forceinline__ void dirty_phase(u32x4* slab, int bytes, unsigned tag) {
  const int n16 = bytes >> 4;
  for (int i = threadIdx.x; i < n16; i += blockDim.x) {
    u32x4 v = {tag, tag ^ (unsigned)i, tag, tag};
    slab[i] = v;
  }
}

__device__ __forceinline__ bool spin_until(unsigned* p, unsigned target, unsigned* timeout) {
  unsigned spins = 0;
  while ((int)(ld_relaxed(p) - target) < 0) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1u << 22)) {
      *timeout = 1;
      return false;
    }
  }
  return true;
}

template <int MODE>
__global__ void __launch_bounds__(512) phases_kernel(Sync s, u32x4* buf, size_t slab16, int bytes, int iters, unsigned phase0,
                                                      unsigned* errors) {
  const unsigned nwg = gridDim.x, wg = blockIdx.x;
  u32x4* mine = buf + (size_t)wg * slab16;
  const u32x4* theirs = buf + (size_t)((wg + 1) % nwg) * slab16;
  const unsigned xcd = wg & 7u, per_xcd = (nwg + 7u - xcd) / 8u;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned tag = phase0 + (unsigned)it + 1u;
    dirty_phase(mine, bytes, tag);
    if (MODE == MODE_NONE) continue;
    __syncthreads();     // every wave's stores are issued; lane 0's release below covers the workgroup
    if (threadIdx.x == 0) {
      if (MODE == MODE_FLAT) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        add_relaxed(s.flat, 1u);
        spin_until(s.flat, (unsigned)(it + 1) * nwg, s.timeout);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this workgroup's stores have reached the XCD's L2
        const unsigned old = add_relaxed(s.xcd_cnt + xcd * 32, 1u);
        if (old + 1u == (unsigned)(it + 1) * per_xcd) {       // the XCD's last arriver: ONE write-back of the XCD's L2
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          add_relaxed(s.top, 1u);
          spin_until(s.top, (unsigned)(it + 1) * 8u, s.timeout);
          __hip_atomic_store(s.xcd_gen + xcd * 32, (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          spin_until(s.xcd_gen + xcd * 32, (unsigned)(it + 1), s.timeout);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    }
    __syncthreads();
    // consumer side, L1-warm: read the neighbour's slab of THIS phase (first / last 4 KB of what it dirtied) and check every word
    const int n16 = bytes >> 4;
    const int chk = n16 < 256 ? n16 : 256;
    for (int i = threadIdx.x; i < chk; i += blockDim.x) {
      const int j = (i & 1) ? n16 - 1 - (i >> 1) : (i >> 1);
      const u32x4 v = theirs[j];
      bad += (v.x != tag) | (v.y != (tag ^ (unsigned)j)) | (v.z != tag) | (v.w != tag);
    }
  }
  if (bad) atomicAdd(errors, bad);
}

static double time_launches(hipStream_t st, int reps, const std::function<void()>& f) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  f();
  CHECK(hipStreamSynchronize(st));
  CHECK(hipEventRecord(e0, st));
  for (int r = 0; r < reps; ++r) f();
  CHECK(hipEventRecord(e1, st));
  CHECK(hipStreamSynchronize(st));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  return (double)ms * 1e3 / reps;      // us per call of f
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  const int reps = 5;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int nwg = prop.multiProcessorCount;
  printf("# %s, %d CUs; %d workgroups (one per CU), %d phases per launch, %d launches per figure\n", prop.gcnArchName, nwg, nwg, iters, reps);
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  const size_t slab_bytes = 1 << 20;
  u32x4* buf;
  CHECK(hipMalloc(&buf, slab_bytes * nwg));
  CHECK(hipMemset(buf, 0, slab_bytes * nwg));
  unsigned* words;
  CHECK(hipMalloc(&words, 4096 * 4));
  unsigned* errors = words + 2048;
  Sync s{words, words + 64, words + 64 + 256, words + 32, words + 2049};
  unsigned phase0 = 0;

  printf("# columns: threads  dirtied_per_wg  us/phase(no barrier)  us/phase(flat)  barrier_flat_us  us/phase(xcd)  barrier_xcd_us  "
         "us/launch(chain of 1-phase kernels, hipGraph)  boundary_us  errors  timeouts\n");
  for (int threads : {256, 512}) {
    for (int bytes : {0, 64 << 10, 1 << 20}) {
      auto launch = [&](int mode, int n_it) {
        CHECK(hipMemsetAsync(words, 0, 2048 * 4, st));      // barrier state re-initialised every launch (Guideline 16)
        if (mode == MODE_NONE)
          hipLaunchKernelGGL(phases_kernel<MODE_NONE>, dim3(nwg), dim3(threads), 0, st, s, buf, slab_bytes / 16, bytes, n_it, phase0, errors);
        else if (mode == MODE_FLAT)
          hipLaunchKernelGGL(phases_kernel<MODE_FLAT>, dim3(nwg), dim3(threads), 0, st, s, buf, slab_bytes / 16, bytes, n_it, phase0, errors);
        else
          hipLaunchKernelGGL(phases_kernel<MODE_XCD>, dim3(nwg), dim3(threads), 0, st, s, buf, slab_bytes / 16, bytes, n_it, phase0, errors);
        phase0 += (unsigned)n_it;
      };
      CHECK(hipMemset(errors, 0, 8));
      const double t_none = time_launches(st, reps, [&] { launch(MODE_NONE, iters); }) / iters;
      const double t_flat = time_launches(st, reps, [&] { launch(MODE_FLAT, iters); }) / iters;
      const double t_xcd = time_launches(st, reps, [&] { launch(MODE_XCD, iters); }) / iters;
      // the boundary: the same dirtying body as `iters` dependent one-phase launches, captured in a graph (what the step does today)
      hipGraph_t g;
      hipGraphExec_t ge;
      CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL(phases_kernel<MODE_NONE>, dim3(nwg), dim3(threads), 0, st, s, buf, slab_bytes / 16, bytes, 1, phase0 + i, errors);
      CHECK(hipStreamEndCapture(st, &g));
      CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      const double t_chain = time_launches(st, reps, [&] { CHECK(hipGraphLaunch(ge, st)); }) / iters;
      CHECK(hipGraphExecDestroy(ge));
      CHECK(hipGraphDestroy(g));
      unsigned h[2];
      CHECK(hipMemcpy(h, errors, 8, hipMemcpyDeviceToHost));
      printf("%4d  %8d  %8.2f  %8.2f  %8.2f  %8.2f  %8.2f  %8.2f  %8.2f  %u  %u\n", threads, bytes, t_none, t_flat, t_flat - t_none, t_xcd,
             t_xcd - t_none, t_chain, t_chain - t_none, h[0], h[1]);
      fflush(stdout);
    }
  }
  return 0;
}
