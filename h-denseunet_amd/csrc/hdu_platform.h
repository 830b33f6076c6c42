// hdu_platform.h -- build-target glue.
//
// The product build is gfx950 only (hipcc --offload-arch=gfx950).  The same
// sources also compile for x86 with -DHDU_EMU against tests/hipemu (test
// infrastructure: lets the CPU-only test tier execute kernel logic).  There is
// no other platform and no runtime dispatch between the two.
#pragma once

#include <atomic>

#ifdef HDU_EMU
#include "hipemu.h"
extern std::atomic<int> g_hdu_prof_on;
int hdu_prof_note(const void* kernel_addr);
#define HDU_LAUNCH(kern, grid, block, smem, stream, ...)                      \
  do {                                                                        \
    if (g_hdu_prof_on) (void)hdu_prof_note((const void*)(kern));              \
    hipemu::launch((grid), (block), (smem), [=]() { kern(__VA_ARGS__); });    \
  } while (0)
#define HDU_DYN_SMEM(name) char* name = hipemu::g_cur->dyn_smem
#define HDU_LANE() (hipemu::g_cur->lane)
#define HDU_LAUNCH_OK() 0
#else
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
// Launch profiler (include/hdu.h: hdu_profile_*).  While armed, every HDU_LAUNCH goes through hipExtLaunchKernelGGL with
// a start / stop event pair attached to THAT dispatch (the events take the dispatch's own begin / end timestamps -- what
// rocprofv3 --kernel-trace reports -- not the time between two marker packets, which adds a box-dependent 2-9 us to
// every launch); the record keeps the kernel's address (hdu_profile_get resolves it to the instantiated name: dladdr +
// demangling).  Not armed (always, outside bench.py's one instrumented step): one predictable branch.
extern std::atomic<int> g_hdu_prof_on;
int hdu_prof_next(const void* kernel_addr, hipStream_t stream, hipEvent_t* e0, hipEvent_t* e1);
// hipGetLastError() is sticky per thread: clear whatever an unrelated earlier runtime call left behind so that
// hdu_check_launch() reports THIS launch only
#define HDU_LAUNCH(kern, grid, block, smem, stream, ...)                                                         \
  do {                                                                                                           \
    (void)hipGetLastError();                                                                                     \
    hipEvent_t pe0_, pe1_;                                                                                       \
    if (g_hdu_prof_on && hdu_prof_next((const void*)(kern), (stream), &pe0_, &pe1_))                                       \
      hipExtLaunchKernelGGL(kern, (grid), (block), (smem), (stream), pe0_, pe1_, 0, __VA_ARGS__);                \
    else                                                                                                         \
      hipLaunchKernelGGL(kern, (grid), (block), (smem), (stream), __VA_ARGS__);                                  \
  } while (0)
#define HDU_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define HDU_LANE() ((int)(threadIdx.x & 63))
#define HDU_LAUNCH_OK() ((int)hipGetLastError())
#endif

#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

#ifndef HDU_EMU
typedef __bf16 hdu_bf16x8 __attribute__((ext_vector_type(8)));
#endif


// NOTE: never __builtin_bit_cast a vector-element lvalue (v.y) directly -- clang reads element 0 of the
// vector's storage.  These by-value helpers force an rvalue.
__host__ __device__ __forceinline__ float hdu_u2f(unsigned u) { return __builtin_bit_cast(float, u); }
__host__ __device__ __forceinline__ unsigned hdu_f2u(float f) { return __builtin_bit_cast(unsigned, f); }

// ---- bf16 storage helpers (raw 16-bit patterns; round-to-nearest-even) ----
typedef unsigned short bf16_t;

__host__ __device__ __forceinline__ float bf16_to_f32(bf16_t h) {
  return __builtin_bit_cast(float, (unsigned)h << 16);
}
// two floats -> packed bf16x2 (round-to-nearest-even).  gfx950: one v_cvt_pk_bf16_f32.
__device__ __forceinline__ unsigned hdu_pack_bf16x2(float lo, float hi);

__host__ __device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  unsigned u = hdu_f2u(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

#ifdef HDU_EMU
__device__ __forceinline__ unsigned hdu_pack_bf16x2(float lo, float hi) {
  return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
}
__device__ __forceinline__ bf16_t hdu_f32_to_bf16_dev(float f) { return f32_to_bf16(f); }
#else
typedef __bf16 hdu_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hdu_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned hdu_pack_bf16x2(float lo, float hi) {
  const hdu_bf16x2 r = __builtin_convertvector(hdu_f32x2{lo, hi}, hdu_bf16x2);
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ bf16_t hdu_f32_to_bf16_dev(float f) {
  const __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}
#endif

// ---- 16-byte chunk <-> float lanes, per storage type ----
template <typename T> struct Chunk;  // CH elements per 16 bytes

template <> struct Chunk<float> {
  static constexpr int CH = 4;
  __device__ __forceinline__ static void unpack(u32x4 v, float* f) {
    f[0] = hdu_u2f(v.x); f[1] = hdu_u2f(v.y);
    f[2] = hdu_u2f(v.z); f[3] = hdu_u2f(v.w);
  }
  __device__ __forceinline__ static u32x4 pack(const float* f) {
    return u32x4{hdu_f2u(f[0]), hdu_f2u(f[1]),
                 hdu_f2u(f[2]), hdu_f2u(f[3])};
  }
  __device__ __forceinline__ static float load1(const float* p) { return *p; }
  __device__ __forceinline__ static void store1(float* p, float v) { *p = v; }
  // 4 consecutive elements, 16-byte aligned destination
  __device__ __forceinline__ static void store4(float* p, const float* v) {
    *(u32x4*)p = u32x4{hdu_f2u(v[0]), hdu_f2u(v[1]), hdu_f2u(v[2]), hdu_f2u(v[3])};
  }
  __device__ __forceinline__ static void load4(const float* p, float* v) {
    const u32x4 q = *(const u32x4*)p;
    v[0] = hdu_u2f(q.x); v[1] = hdu_u2f(q.y); v[2] = hdu_u2f(q.z); v[3] = hdu_u2f(q.w);
  }
  __device__ __forceinline__ static float rounded(float v) { return v; }     // the value as stored
};

template <> struct Chunk<bf16_t> {
  static constexpr int CH = 8;
  __device__ __forceinline__ static void unpack(u32x4 v, float* f) {
    f[0] = hdu_u2f(v.x << 16); f[1] = hdu_u2f(v.x & 0xffff0000u);
    f[2] = hdu_u2f(v.y << 16); f[3] = hdu_u2f(v.y & 0xffff0000u);
    f[4] = hdu_u2f(v.z << 16); f[5] = hdu_u2f(v.z & 0xffff0000u);
    f[6] = hdu_u2f(v.w << 16); f[7] = hdu_u2f(v.w & 0xffff0000u);
  }
  __device__ __forceinline__ static u32x4 pack(const float* f) {
    return u32x4{hdu_pack_bf16x2(f[0], f[1]), hdu_pack_bf16x2(f[2], f[3]), hdu_pack_bf16x2(f[4], f[5]),
                 hdu_pack_bf16x2(f[6], f[7])};
  }
  __device__ __forceinline__ static float load1(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ __forceinline__ static void store1(bf16_t* p, float v) { *p = hdu_f32_to_bf16_dev(v); }
  // 4 consecutive elements, 8-byte aligned destination
  __device__ __forceinline__ static void store4(bf16_t* p, const float* v) {
    *(u32x2*)p = u32x2{hdu_pack_bf16x2(v[0], v[1]), hdu_pack_bf16x2(v[2], v[3])};
  }
  __device__ __forceinline__ static void load4(const bf16_t* p, float* v) {
    const u32x2 q = *(const u32x2*)p;
    v[0] = hdu_u2f(q.x << 16); v[1] = hdu_u2f(q.x & 0xffff0000u); v[2] = hdu_u2f(q.y << 16); v[3] = hdu_u2f(q.y & 0xffff0000u);
  }
  __device__ __forceinline__ static float rounded(float v) { return bf16_to_f32(hdu_f32_to_bf16_dev(v)); }   // the value as stored
};

// ---- MFMA wrappers: one "k-group" = 16 bytes of k per lane for A and B ----
// bf16: one v_mfma_f32_16x16x32_bf16 (K=32); f32: four v_mfma_f32_16x16x4_f32
// (K=16, lane's 4 consecutive k are fed as 4 successive k-slices; A and B use
// the same permutation of k so the contraction is unchanged).
template <typename T> struct Mma;

template <> struct Mma<bf16_t> {
  template <bool SPLIT = false> __device__ __forceinline__ static f32x4 kgroup(u32x4 a, u32x4 b, f32x4 c) {
#ifdef HDU_EMU
    return hipemu_mfma_16x16x32_bf16(__builtin_bit_cast(hipemu_u16x8, a), __builtin_bit_cast(hipemu_u16x8, b), c);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(hdu_bf16x8, a),
                                                   __builtin_bit_cast(hdu_bf16x8, b), c, 0, 0, 0);
#endif
  }
};

// float32 operands as bf16 hi + lo (x = hi + lo + r, |r| <= 2^-18 |x|: both parts round to nearest even)
__device__ __forceinline__ void hdu_split_bf16(u32x4 v, u32x2& hi, u32x2& lo) {
  const float f0 = hdu_u2f(v.x), f1 = hdu_u2f(v.y), f2 = hdu_u2f(v.z), f3 = hdu_u2f(v.w);
  hi = u32x2{hdu_pack_bf16x2(f0, f1), hdu_pack_bf16x2(f2, f3)};
  lo = u32x2{hdu_pack_bf16x2(f0 - hdu_u2f(hi.x << 16), f1 - hdu_u2f(hi.x & 0xffff0000u)),
             hdu_pack_bf16x2(f2 - hdu_u2f(hi.y << 16), f3 - hdu_u2f(hi.y & 0xffff0000u))};
}
// v_mfma_f32_16x16x16_bf16: a lane holds 4 consecutive k of its row / column, lane group g the k block 4g..4g+3 -- the
// k order of the float32 k-group above
__device__ __forceinline__ f32x4 hdu_mfma_16x16x16_bf16(u32x2 a, u32x2 b, f32x4 c) {
#ifdef HDU_EMU
  return hipemu_mfma_16x16x16_bf16(__builtin_bit_cast(hipemu_u16x4, a), __builtin_bit_cast(hipemu_u16x4, b), c);
#else
  typedef short hdu_s16x4 __attribute__((ext_vector_type(4)));
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(hdu_s16x4, a), __builtin_bit_cast(hdu_s16x4, b), c, 0, 0, 0);
#endif
}

// SPLIT (chosen per launch from ConvK::f32_split): false = exact float32 (four v_mfma_f32_16x16x4_f32, 4 x 32 cycles);
// true = "bf16 x 3": a.b ~ ah.bh + ah.bl + al.bh on three v_mfma_f32_16x16x16_bf16 with the float32 accumulator -- the
// dropped terms (al.bl and the two split remainders) are <= 3 * 2^-18 of |a.b| per product
template <> struct Mma<float> {
  template <bool SPLIT = false> __device__ __forceinline__ static f32x4 kgroup(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (SPLIT) {
      u32x2 ah, al, bh, bl;
      hdu_split_bf16(a, ah, al);
      hdu_split_bf16(b, bh, bl);
      c = hdu_mfma_16x16x16_bf16(al, bh, c);
      c = hdu_mfma_16x16x16_bf16(ah, bl, c);
      return hdu_mfma_16x16x16_bf16(ah, bh, c);
    }
#ifdef HDU_EMU
    c = hipemu_mfma_16x16x4_f32(hdu_u2f(a.x), hdu_u2f(b.x), c);
    c = hipemu_mfma_16x16x4_f32(hdu_u2f(a.y), hdu_u2f(b.y), c);
    c = hipemu_mfma_16x16x4_f32(hdu_u2f(a.z), hdu_u2f(b.z), c);
    c = hipemu_mfma_16x16x4_f32(hdu_u2f(a.w), hdu_u2f(b.w), c);
#else
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(hdu_u2f(a.x), hdu_u2f(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(hdu_u2f(a.y), hdu_u2f(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(hdu_u2f(a.z), hdu_u2f(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(hdu_u2f(a.w), hdu_u2f(b.w), c, 0, 0, 0);
#endif
    return c;
  }
};

// sum over the 16 lanes of a DPP row (lanes 16r..16r+15), result in every lane: quad_perm xor 1, xor 2, row_half_mirror,
// row_mirror -- four VALU adds with DPP operands instead of four ds_bpermute round trips.  (After the two quad steps the
// values are quad-uniform, so the mirrors pair the same quads / halves as xor 4 / xor 8 would: bit-identical sums.)
__device__ __forceinline__ float hdu_row16_sum(float v) {
#ifdef HDU_EMU
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
#else
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
#endif
  return v;
}

// LDS transpose read: lane i of each 16-lane group passes the address of row (i>>2), columns (i&3)*4.. of a 4x16
// row-major 16-bit tile and receives column i (4 elements, rows 0..3) -- ds_read_b64_tr_b16.
__device__ __forceinline__ u32x2 hdu_lds_tr16_b64(const void* p) {
#ifdef HDU_EMU
  return __builtin_bit_cast(u32x2, hipemu_ds_read_tr16_b64(p));
#else
  typedef short hdu_v4i16 __attribute__((ext_vector_type(4)));
  const hdu_v4i16 r =
      __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) hdu_v4i16*)(p));
  return __builtin_bit_cast(u32x2, r);
#endif
}

// async global -> LDS copy of 16 bytes per lane (global_load_lds_dwordx4): the LDS destination is the wave-uniform
// `lds_wave_base` + lane*16 (lane-linear, 1 KiB per wave instruction); the global source is per lane.
__device__ __forceinline__ void hdu_glds16(const void* gsrc, char* lds_wave_base) {
#ifdef HDU_EMU
  __builtin_memcpy(lds_wave_base + HDU_LANE() * 16, gsrc, 16);
#else
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#endif
}

// The same async copy through a buffer resource: `buffer_load_dwordx4 v_off, s[rsrc], 0 offen lds` -- the base lives in four
// SGPRs, a lane supplies ONE 32-bit byte offset, and a lane whose offset (+16) lies beyond `nbytes` writes ZEROS to its
// LDS slot (measured on gfx950, tools/ubench_glds.hip: "out-of-range lanes write zeros to LDS: YES"), so padding and tile
// tails need no zero page and no 64-bit pointer select.  Issue rate (same tool): 42 B/clk/CU from one 4-wave workgroup
// against 30 for global_load_lds (two address VGPRs), 53 against 46 with two workgroups per CU.
#ifdef HDU_EMU
struct hdu_bufsrd { const char* base; unsigned nbytes; };
__device__ __forceinline__ hdu_bufsrd hdu_make_srd(const void* p, unsigned nbytes) { return hdu_bufsrd{(const char*)p, nbytes}; }
__device__ __forceinline__ void hdu_bufload_lds16(const hdu_bufsrd& r, unsigned byte_off, char* lds_wave_base) {
  char* dst = lds_wave_base + HDU_LANE() * 16;
  if ((unsigned long long)byte_off + 16ull <= (unsigned long long)r.nbytes) __builtin_memcpy(dst, r.base + byte_off, 16);
  else __builtin_memset(dst, 0, 16);
}
#else
typedef __amdgpu_buffer_rsrc_t hdu_bufsrd;
__device__ __forceinline__ hdu_bufsrd hdu_make_srd(const void* p, unsigned nbytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)nbytes, 0x00020000);
}
__device__ __forceinline__ void hdu_bufload_lds16(const hdu_bufsrd& r, unsigned byte_off, char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, byte_off, 0, 0, 0);
}
#endif
#define HDU_OOB 0xffffffffu

// raw workgroup barrier / counted vector-memory wait (lets async LDS-DMA tiles stay in flight across a barrier;
// __syncthreads() would drain them with vmcnt(0)).  LDS traffic is still fenced with lgkmcnt(0).
#ifdef HDU_EMU
#define HDU_WAIT_VMCNT(n) do { } while (0)
#define HDU_RAW_BARRIER() __syncthreads()
#else
#define HDU_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define HDU_RAW_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
#endif

// A 16-byte global load that hipcc does NOT count (cdna_hip_programming.md section 5.7, form (ii)): the compiler's own waitcnt
// bookkeeping waits for a register load with vmcnt(k) where k only counts what IT issued after the load -- in a loop that also
// keeps LDS-DMA tiles and stores in flight that comes out as vmcnt(0) at the end of every tile (ISA of round 5's
// conv_pw_bstat_kernel: the ring's prefetch depth was drained once per tile).  HDU_ASYNC_BUFLOAD16 issues the load invisibly;
// hdu_wait_vmcnt_regs4<N> is the counted wait that names the destination registers ("+v": no consumer is scheduled above it).
// The CALLER counts: N = vector-memory LOADS (register or LDS-DMA) issued after the ones it wants -- loads retire in order among
// themselves, so "at most N outstanding" then implies the wanted ones have landed whatever the stores in between do.
// Addressing: a raw buffer resource in four SGPRs + ONE 32-bit byte offset per lane (a lane out of range reads zeros / stores
// nothing: HDU_OOB).  Destination: ACCUMULATOR registers ("=a": gfx950 vector-memory loads may target AGPRs) -- with the MFMA
// accumulators in VGPRs (-amdgpu-mfma-vgpr-form) the 256 AGPRs of a wave are free, so chunks in flight cost no architectural
// VGPR.  (With "=v" destinations and three sets in flight hipcc ran out of VGPRs and parked the just-issued, NOT YET LANDED
// destination registers in AGPRs with v_accvgpr_write -- the silent-garbage case of cdna_hip_programming.md section 5.7 item 1;
// tools/disasm_kernel.py + the scan in DESIGN.md section 3.9 found it.)  The consumer reads them after the counted wait.
#ifdef HDU_EMU
struct hdu_rawsrd { char* base; unsigned nbytes; };
__device__ __forceinline__ hdu_rawsrd hdu_make_rawsrd(const void* p, unsigned nbytes) { return hdu_rawsrd{(char*)p, nbytes}; }
#define HDU_ASYNC_BUFLOAD16(dst, srd, off)                                                                     \
  do {                                                                                                         \
    if ((unsigned long long)(off) + 16ull <= (unsigned long long)(srd).nbytes) (dst) = *(const u32x4*)((srd).base + (off)); \
    else (dst) = u32x4{0u, 0u, 0u, 0u};                                                                        \
  } while (0)
__device__ __forceinline__ void hdu_bufstore16(const hdu_rawsrd& r, unsigned off, u32x4 v) {
  if ((unsigned long long)off + 16ull <= (unsigned long long)r.nbytes) *(u32x4*)(r.base + off) = v;
}
template <int N> __device__ __forceinline__ void hdu_wait_vmcnt_regs4(u32x4&, u32x4&, u32x4&, u32x4&) {}
#else
// the same raw buffer twice: as the compiler's resource type (stores through the builtin) and as its four descriptor words,
// wave-uniform (readfirstlane), the "s" operand of the asm load below
struct hdu_rawsrd { __amdgpu_buffer_rsrc_t r; u32x4 w; };
__device__ __forceinline__ hdu_rawsrd hdu_make_rawsrd(const void* p, unsigned nbytes) {
  const unsigned long long a = (unsigned long long)p;
  hdu_rawsrd s;
  s.r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)nbytes, 0x00020000);
  s.w.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  s.w.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);      // stride 0: raw buffer
  s.w.z = __builtin_amdgcn_readfirstlane(nbytes);
  s.w.w = __builtin_amdgcn_readfirstlane(0x00020000u);
  return s;
}
#define HDU_ASYNC_BUFLOAD16(dst, srd, off) \
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=a"(dst) : "v"(off), "s"((srd).w) : "memory")
__device__ __forceinline__ void hdu_bufstore16(const hdu_rawsrd& r, unsigned off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r.r, (int)off, 0, 0);
}
template <int N> __device__ __forceinline__ void hdu_wait_vmcnt_regs4(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
  static_assert(N >= 0 && N < 64, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%4)" : "+a"(a), "+a"(b), "+a"(c), "+a"(d) : "i"(N) : "memory");
}
#endif

// LDS hand-off INSIDE one wave (lane A writes, lane B of the same wave reads): a wave's DS instructions execute in order, so
// the data is there once the writes have been issued and counted down; no workgroup barrier needed
#ifdef HDU_EMU
#define HDU_WAVE_LDS_SYNC() hipemu::wave_barrier()
#else
#define HDU_WAVE_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif

// keeps the instruction scheduler from sinking a group of hoisted loads back towards their uses
#ifdef HDU_EMU
#define HDU_SCHED_BARRIER() do { } while (0)
#else
#define HDU_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif

// "last workgroup finishes the job" hand-off without cache flushes.  An agent-scope release fence (buffer_wbl2) writes
// back EVERY dirty line of the XCD's L2 -- measured +17 us per launch here, the L2 is full of the previous kernel's
// output.  Instead the partial results themselves are agent-scope relaxed atomic stores / loads (sc1: written through
// to / read from the device coherence point, no fence needed for coherence of the location); ordering comes from
// s_waitcnt vmcnt(0) (the stores have been acknowledged) + the workgroup barrier before ONE lane takes the ticket.
#ifdef HDU_EMU
__device__ __forceinline__ unsigned hdu_ticket(unsigned* counter) { return atomicAdd(counter, 1u); }
__device__ __forceinline__ void hdu_store_agent(float* p, float v) { *p = v; }
__device__ __forceinline__ float hdu_load_agent(const float* p) { return *p; }
__device__ __forceinline__ void hdu_store_agent_u32(unsigned* p, unsigned v) { *p = v; }
#define HDU_WAIT_STORES() do { } while (0)
#else
__device__ __forceinline__ unsigned hdu_ticket(unsigned* counter) {
  return __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void hdu_store_agent(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float hdu_load_agent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void hdu_store_agent_u32(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#define HDU_WAIT_STORES() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

// split-K hand-off (cdna_hip_programming.md Guideline 16, recipe R1): the partial tile is stored WRITE-THROUGH (sc1: the
// bytes leave the XCD's L2, no release fence -- a release would write back every dirty line the previous kernel left
// there), every storing wave drains its stores, the workgroup meets at a barrier and ONE lane takes a ticket with a
// relaxed agent-scope atomic; the tile's last arriver issues ONE agent-scope acquire (drops its CU's L1) and reads the
// other partial tiles with plain loads.  Placement-independent: no assumption on which XCD runs which split.
#ifdef HDU_EMU
__device__ __forceinline__ void hdu_store_wt16(float* p, f32x4 v) { *(f32x4*)p = v; }
__device__ __forceinline__ void hdu_acquire_agent() {}
#else
__device__ __forceinline__ void hdu_store_wt16(float* p, f32x4 v) {
  // 16-byte global store with sc1 (write-through); the s_nop keeps the data registers alive until the store has read them
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void hdu_acquire_agent() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
#endif

// counter-based hash RNG for dropout masks (stateless; fwd and bwd regenerate the same mask)
__host__ __device__ __forceinline__ unsigned hdu_hash32(unsigned long long idx, unsigned seed) {
  unsigned h = (unsigned)idx * 0x9E3779B1u;
  h ^= (unsigned)(idx >> 32) * 0xC2B2AE3Du + (seed + 1u) * 0x85EBCA77u;
  h ^= h >> 16; h *= 0x85EBCA6Bu;
  h ^= h >> 13; h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
