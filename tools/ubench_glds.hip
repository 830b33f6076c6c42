// tools/ubench_glds.hip -- what can one CU pull through `global_load_lds_dwordx4`?  (the large-grid
// implicit GEMM sits at 11-16 B/clk/CU of operand DMA; this measures the ceiling of that path by access shape.)
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_glds tools/ubench_glds.hip && tools/ubench_glds
//
// Every wave issues 1 KiB LDS-DMA instructions (64 lanes x 16 B).  The 1 KiB is cut into segments of `seg` bytes
// (128 = one 64-channel bf16 pixel row of the GEMM's K step, 1024 = fully contiguous); consecutive segments are `stride`
// bytes apart.  mode 0: DEPTH instructions stay in flight per wave (counted vmcnt); mode 1: bursts of DEPTH then vmcnt(0)
// + barrier (the 2-stage GEMM's shape).  Sources: a 2 MiB window shared by all workgroups (L2 hits after the first
// touch), or a private stream per workgroup (HBM).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int DEPTH, int MODE>
__global__ __launch_bounds__(256) void glds_kernel(const char* src, size_t region_bytes, int nregions, int seg, size_t stride,
                                                   int iters, unsigned* sink) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int segs = 1024 / seg;
  const size_t in_seg = (size_t)(lane * 16) % seg, sidx = (size_t)(lane * 16) / seg;
  const char* base = src + (size_t)(blockIdx.x % nregions) * region_bytes;
  const size_t mask = region_bytes - 1;
  char* my = lds + w * DEPTH * 1024;
  for (int it = 0; it < iters; ++it) {
    const size_t row = ((size_t)(it + blockIdx.x * 17) * 4 + w) * segs + sidx;      // workgroups start at different phases
    const char* g = base + ((row * stride + in_seg) & mask);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(my + (it % DEPTH) * 1024), 16, 0, 0);
    if (MODE == 0) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
    } else if ((it % DEPTH) == DEPTH - 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = *(unsigned*)lds;
}

// the same stream through `buffer_load_dwordx4 ... offen lds` (SRD in SGPRs + one 32-bit offset VGPR per lane)
template <int DEPTH>
__global__ __launch_bounds__(256) void bufload_kernel(const char* src, size_t region_bytes, int nregions, int seg, size_t stride,
                                                      int iters, unsigned* sink) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int segs = 1024 / seg;
  const unsigned in_seg = (unsigned)(lane * 16) % seg, sidx = (unsigned)(lane * 16) / seg;
  const char* base = src + (size_t)(blockIdx.x % nregions) * region_bytes;
  const unsigned mask = (unsigned)region_bytes - 1u;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (unsigned)region_bytes, 0x00020000);
  char* my = lds + w * DEPTH * 1024;
  for (int it = 0; it < iters; ++it) {
    const unsigned row = ((unsigned)(it + blockIdx.x * 17) * 4 + w) * segs + sidx;
    const unsigned off = (row * (unsigned)stride + in_seg) & mask;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(my + (it % DEPTH) * 1024), 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = *(unsigned*)lds;
}

// does an out-of-range lane of a buffer LDS-DMA write ZEROS into its 16 bytes of LDS?  (free zero padding for the GEMM)
__global__ __launch_bounds__(64) void oob_check_kernel(const char* src, unsigned nbytes, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned l[256];
  for (int i = threadIdx.x; i < 256; i += 64) l[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  unsigned off = threadIdx.x * 16;
  if (threadIdx.x & 1) off = 0xfffffff0u;                 // odd lanes: far out of range
  if (threadIdx.x == 62) off = nbytes - 8;                // straddles the end
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)l, 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = l[i];
}

// the same traffic through registers (global_load_dwordx4), for comparison
template <int DEPTH>
__global__ __launch_bounds__(256) void gload_kernel(const char* src, size_t region_bytes, int nregions, int seg, size_t stride,
                                                    int iters, unsigned* sink) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int segs = 1024 / seg;
  const size_t in_seg = (size_t)(lane * 16) % seg, sidx = (size_t)(lane * 16) / seg;
  const char* base = src + (size_t)(blockIdx.x % nregions) * region_bytes;
  const size_t mask = region_bytes - 1;
  uint4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; it += DEPTH) {
    uint4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const size_t row = ((size_t)(it + d + blockIdx.x * 17) * 4 + w) * segs + sidx;
      v[d] = *(const uint4*)(base + ((row * stride + in_seg) & mask));
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { acc.x ^= v[d].x; acc.y ^= v[d].y; acc.z ^= v[d].z; acc.w ^= v[d].w; }
  }
  if (sink && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[blockIdx.x] = 1;
}

struct Case { const char* src_name; size_t region; int nregions; int seg; size_t stride; };

template <int DEPTH, int MODE>
static double run(const char* buf, const Case& c, int wgs_per_cu, int iters, unsigned* sink) {
  const int grid = 256 * wgs_per_cu;
  // dynamic LDS sized so that exactly `wgs_per_cu` workgroups fit a CU (160 KiB), at least the ring
  size_t lds = (160 * 1024 / wgs_per_cu) & ~(size_t)1023;
  if (lds < (size_t)4 * DEPTH * 1024) return 0.0;        // the ring does not fit at this occupancy
  if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)glds_kernel<DEPTH, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  double best = 1e30;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((glds_kernel<DEPTH, MODE>), dim3(grid), dim3(256), lds, 0, buf, c.region, c.nregions == 0 ? grid : c.nregions,
                       c.seg, c.stride, iters, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (rep && ms < best) best = ms;
  }
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return (double)grid * iters * 4096.0 / (best * 1e-3);      // bytes / s
}

template <int DEPTH>
static double run_buf(const char* buf, const Case& c, int wgs_per_cu, int iters, unsigned* sink) {
  const int grid = 256 * wgs_per_cu;
  size_t lds = (160 * 1024 / wgs_per_cu) & ~(size_t)1023;
  if (lds < (size_t)4 * DEPTH * 1024 || c.region >= ((size_t)1 << 32)) return 0.0;
  if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)bufload_kernel<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  double best = 1e30;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((bufload_kernel<DEPTH>), dim3(grid), dim3(256), lds, 0, buf, c.region, c.nregions == 0 ? grid : c.nregions,
                       c.seg, c.stride, iters, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (rep && ms < best) best = ms;
  }
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return (double)grid * iters * 4096.0 / (best * 1e-3);
}

template <int DEPTH>
static double run_reg(const char* buf, const Case& c, int wgs_per_cu, int iters, unsigned* sink) {
  const int grid = 256 * wgs_per_cu;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  double best = 1e30;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((gload_kernel<DEPTH>), dim3(grid), dim3(256), 0, 0, buf, c.region, c.nregions == 0 ? grid : c.nregions, c.seg,
                       c.stride, iters, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (rep && ms < best) best = ms;
  }
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return (double)grid * iters * 4096.0 / (best * 1e-3);
}

int main() {
  const size_t total = (size_t)4 << 30;
  char* buf; unsigned* sink;
  CK(hipMalloc(&buf, total)); CK(hipMemset(buf, 1, total)); CK(hipMalloc(&sink, 4096 * 4));
  const int iters = 1024;                               // x 4 KiB per workgroup = 4 MiB streamed per workgroup
  const size_t priv = (size_t)iters * 4096;             // private stream: no reuse (HBM)
  std::vector<Case> cases = {
      {"L2  shared 2MiB", (size_t)2 << 20, 1, 1024, 1024}, {"L2  shared 2MiB", (size_t)2 << 20, 1, 256, 256},
      {"L2  shared 2MiB", (size_t)2 << 20, 1, 128, 128},   {"L2  shared 2MiB", (size_t)2 << 20, 1, 128, 192},
      {"L2  shared 2MiB", (size_t)2 << 20, 1, 128, 4416},  {"L2  shared 2MiB", (size_t)2 << 20, 1, 64, 64},
      {"MALL 64 x 1MiB ", (size_t)1 << 20, 64, 1024, 1024}, {"MALL 64 x 1MiB ", (size_t)1 << 20, 64, 128, 128},
      {"HBM private    ", priv, 0, 1024, 1024},            {"HBM private    ", priv, 0, 128, 128},
  };
  {   // out-of-range behaviour of the buffer form
    unsigned* o; CK(hipMalloc(&o, 1024));
    hipLaunchKernelGGL(oob_check_kernel, dim3(1), dim3(64), 0, 0, buf, 4096u, o);
    unsigned h[256]; CK(hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost));
    int zero_ok = 1, data_ok = 1;
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        const unsigned v = h[l * 4 + j];
        if ((l & 1) && v != 0u) zero_ok = 0;
        if (!(l & 1) && l != 62 && v != 0x01010101u) data_ok = 0;
      }
    printf("buffer_load ... lds: out-of-range lanes write zeros to LDS: %s; in-range lanes load data: %s; straddling lane 62: %08x %08x %08x %08x\n",
           zero_ok ? "YES" : "NO", data_ok ? "yes" : "NO", h[62 * 4], h[62 * 4 + 1], h[62 * 4 + 2], h[62 * 4 + 3]);
  }
  const double clk = 2.4e9;
  printf("%-16s seg stride wgs/cu | B/clk/CU: stream d2  d4  d8  d16 | burst d4  d8  d16 | regs d4  d8 | buffer-lds d4 d8\n", "source");
  for (const Case& c : cases)
    for (int wpc = 1; wpc <= 4; ++wpc) {
      if ((size_t)256 * wpc * priv > total && c.nregions == 0) continue;
      double r[11];
      r[0] = run<2, 0>(buf, c, wpc, iters, sink);  r[1] = run<4, 0>(buf, c, wpc, iters, sink);
      r[2] = run<8, 0>(buf, c, wpc, iters, sink);  r[3] = run<16, 0>(buf, c, wpc, iters, sink);
      r[4] = run<4, 1>(buf, c, wpc, iters, sink);  r[5] = run<8, 1>(buf, c, wpc, iters, sink);
      r[6] = run<16, 1>(buf, c, wpc, iters, sink);
      r[7] = run_reg<4>(buf, c, wpc, iters, sink); r[8] = run_reg<8>(buf, c, wpc, iters, sink);
      r[9] = run_buf<4>(buf, c, wpc, iters, sink); r[10] = run_buf<8>(buf, c, wpc, iters, sink);
      printf("%-16s %4d %5zu   %d    |          %6.1f %6.1f %6.1f %6.1f | %6.1f %6.1f %6.1f | %6.1f %6.1f | buf %6.1f %6.1f  (%.2f TB/s best)\n", c.src_name,
             c.seg, c.stride, wpc, r[0] / clk / 256, r[1] / clk / 256, r[2] / clk / 256, r[3] / clk / 256, r[4] / clk / 256,
             r[5] / clk / 256, r[6] / clk / 256, r[7] / clk / 256, r[8] / clk / 256, r[9] / clk / 256, r[10] / clk / 256,
             [&] { double m = 0; for (double v : r) m = v > m ? v : m; return m; }() / 1e12);
      fflush(stdout);
    }
  return 0;
}
