// hipemu.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny cooperative-fiber emulator of the HIP execution model (grid / block /
// 64-lane wavefronts, __syncthreads, wave shuffles, LDS, the two gfx950 MFMA
// builtins our kernels use).  It lets the *same kernel sources* under
// h-denseunet_amd/csrc be compiled for x86 (-DHDU_EMU) so that kernel index
// math, tiling and barrier placement can be checked against the oracle in the
// `-m "not gpu"` suite, in a container that has no GPU.
//
// It is NOT a product backend: nothing under h-denseunet_amd/ loads the
// emulator library unless a test explicitly asks for it, and bench.py /
// __graft_entry__.smoke() never do.
//
// Scheduling is run-until-blocked, round-robin over the fibers of one block:
// a missing barrier shows up as a wrong result rather than being hidden by
// lock-step execution.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <functional>

namespace hipemu {

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_ { unsigned x, y, z; };

struct ThreadCtx {
  uint3_ tid;
  uint3_ bid;
  uint3_ bdim;
  uint3_ gdim;
  char* dyn_smem;
  int lane;   // 0..63
  int wave;   // wave index in block
};

extern thread_local ThreadCtx* g_cur;

void launch(dim3 grid, dim3 block, size_t dyn_smem_bytes, const std::function<void()>& body);

// --- synchronisation primitives (called from inside kernels) ---
void sync_threads();
void wave_barrier();
// exchange: every lane of the wave deposits `nbytes` at its slot, after the
// call `all` points at 64 consecutive slots of `slot_stride` bytes.
// Must be followed by wave_release() once the lane is done reading.
const char* wave_exchange(const void* mine, size_t nbytes, size_t* slot_stride);
void wave_release();

template <typename T>
inline T shfl_idx(T v, int src_lane) {
  size_t stride;
  const char* all = wave_exchange(&v, sizeof(T), &stride);
  T r;
  std::memcpy(&r, all + (size_t)(src_lane & 63) * stride, sizeof(T));
  wave_release();
  return r;
}

}  // namespace hipemu

// ---------------------------------------------------------------- HIP surface
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __restrict__ __restrict

using dim3 = hipemu::dim3;
typedef void* hipStream_t;

#define threadIdx (hipemu::g_cur->tid)
#define blockIdx (hipemu::g_cur->bid)
#define blockDim (hipemu::g_cur->bdim)
#define gridDim (hipemu::g_cur->gdim)

static inline void __syncthreads() { hipemu::sync_threads(); }

template <typename T> static inline T __shfl_xor(T v, int mask) {
  return hipemu::shfl_idx(v, hipemu::g_cur->lane ^ mask);
}
template <typename T> static inline T __shfl_down(T v, int d) {
  int s = hipemu::g_cur->lane + d;
  return hipemu::shfl_idx(v, s > 63 ? hipemu::g_cur->lane : s);
}
template <typename T> static inline T __shfl(T v, int lane) { return hipemu::shfl_idx(v, lane); }

static inline float atomicAdd(float* p, float v) {
  unsigned* q = (unsigned*)p;
  unsigned old = __atomic_load_n(q, __ATOMIC_RELAXED), nv;
  float of;
  do {
    std::memcpy(&of, &old, 4);
    const float nf = of + v;
    std::memcpy(&nv, &nf, 4);
  } while (!__atomic_compare_exchange_n(q, &old, nv, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  return of;
}
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

// ---- MFMA builtins (fragment maps: cdna_hip_programming.md section 3) ----
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short hipemu_u16x8 __attribute__((ext_vector_type(8)));

static inline float hipemu_bf16_to_f32(unsigned short h) {
  unsigned u = (unsigned)h << 16; float f; std::memcpy(&f, &u, 4); return f;
}

// D = A(16x32) * B(32x16) + C ; A: lane l holds A[l&15][(l>>4)*8+j]; B: lane l holds B[(l>>4)*8+j][l&15];
// C/D: lane l reg r -> row (l>>4)*4+r, col l&15.
static inline hipemu_f32x4 hipemu_mfma_16x16x32_bf16(hipemu_u16x8 a, hipemu_u16x8 b, hipemu_f32x4 c) {
  struct Slot { unsigned short a[8], b[8]; } mine;
  for (int j = 0; j < 8; ++j) { mine.a[j] = a[j]; mine.b[j] = b[j]; }
  size_t stride;
  const char* all = hipemu::wave_exchange(&mine, sizeof(mine), &stride);
  const int lane = hipemu::g_cur->lane;
  const int col = lane & 15;
  hipemu_f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    const int row = (lane >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 32; ++k) {
      const Slot* sa = (const Slot*)(all + (size_t)(row + 16 * (k >> 3)) * stride);
      const Slot* sb = (const Slot*)(all + (size_t)(col + 16 * (k >> 3)) * stride);
      acc += hipemu_bf16_to_f32(sa->a[k & 7]) * hipemu_bf16_to_f32(sb->b[k & 7]);
    }
    d[r] = acc;
  }
  hipemu::wave_release();
  return d;
}
// D = A(32x16) * B(16x32) + C (v_mfma_f32_32x32x16_bf16); A: lane l holds A[l&31][(l>>5)*8+j]; B: lane l holds B[(l>>5)*8+j][l&31];
// C/D: lane l reg r -> row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31.
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));
static inline hipemu_f32x16 hipemu_mfma_32x32x16_bf16(hipemu_u16x8 a, hipemu_u16x8 b, hipemu_f32x16 c) {
  struct Slot { unsigned short a[8], b[8]; } mine;
  for (int j = 0; j < 8; ++j) { mine.a[j] = a[j]; mine.b[j] = b[j]; }
  size_t stride;
  const char* all = hipemu::wave_exchange(&mine, sizeof(mine), &stride);
  const int lane = hipemu::g_cur->lane;
  const int col = lane & 31;
  hipemu_f32x16 d = c;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    float acc = c[r];
    for (int k = 0; k < 16; ++k) {
      const Slot* sa = (const Slot*)(all + (size_t)(row + 32 * (k >> 3)) * stride);
      const Slot* sb = (const Slot*)(all + (size_t)(col + 32 * (k >> 3)) * stride);
      acc += hipemu_bf16_to_f32(sa->a[k & 7]) * hipemu_bf16_to_f32(sb->b[k & 7]);
    }
    d[r] = acc;
  }
  hipemu::wave_release();
  return d;
}
// D = A(16x4) * B(4x16) + C ; A: lane l holds A[l&15][l>>4]; B: lane l holds B[l>>4][l&15].
static inline hipemu_f32x4 hipemu_mfma_16x16x4_f32(float a, float b, hipemu_f32x4 c) {
  struct Slot { float a, b; } mine{a, b};
  size_t stride;
  const char* all = hipemu::wave_exchange(&mine, sizeof(mine), &stride);
  const int lane = hipemu::g_cur->lane;
  const int col = lane & 15;
  hipemu_f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    const int row = (lane >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) {
      const Slot* sa = (const Slot*)(all + (size_t)(row + 16 * k) * stride);
      const Slot* sb = (const Slot*)(all + (size_t)(col + 16 * k) * stride);
      acc = __builtin_fmaf(sa->a, sb->b, acc);
    }
    d[r] = acc;
  }
  hipemu::wave_release();
  return d;
}

// D = A(16x16) * B(16x16) + C, bf16 operands (v_mfma_f32_16x16x16_bf16): lane l holds A[l&15][(l>>4)*4+j] / B[(l>>4)*4+j][l&15].
typedef unsigned short hipemu_u16x4 __attribute__((ext_vector_type(4)));
static inline hipemu_f32x4 hipemu_mfma_16x16x16_bf16(hipemu_u16x4 a, hipemu_u16x4 b, hipemu_f32x4 c) {
  struct Slot { unsigned short a[4], b[4]; } mine;
  for (int j = 0; j < 4; ++j) { mine.a[j] = a[j]; mine.b[j] = b[j]; }
  size_t stride;
  const char* all = hipemu::wave_exchange(&mine, sizeof(mine), &stride);
  const int lane = hipemu::g_cur->lane;
  const int col = lane & 15;
  hipemu_f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    const int row = (lane >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 16; ++k) {
      const Slot* sa = (const Slot*)(all + (size_t)(row + 16 * (k >> 2)) * stride);
      const Slot* sb = (const Slot*)(all + (size_t)(col + 16 * (k >> 2)) * stride);
      acc += hipemu_bf16_to_f32(sa->a[k & 3]) * hipemu_bf16_to_f32(sb->b[k & 3]);
    }
    d[r] = acc;
  }
  hipemu::wave_release();
  return d;
}

// ds_read_b64_tr_b16 (gfx950 LDS transpose read), lane map measured on MI355X with tools/probe_tr.hip:
// within each 16-lane group the lanes' 8-byte pieces form a 4x16 row-major bf16 tile (lane i supplies row i>>2,
// columns (i&3)*4..+3); lane i receives column i, rows 0..3.
static inline hipemu_u16x4 hipemu_ds_read_tr16_b64(const void* lds_addr) {
  unsigned short mine[4];
  std::memcpy(mine, lds_addr, 8);
  size_t stride;
  const char* all = hipemu::wave_exchange(mine, 8, &stride);
  const int lane = hipemu::g_cur->lane, i = lane & 15, g = lane >> 4;
  hipemu_u16x4 r;
  for (int j = 0; j < 4; ++j) {
    const unsigned short* src = (const unsigned short*)(all + (size_t)(16 * g + 4 * j + (i >> 2)) * stride);
    r[j] = src[i & 3];
  }
  hipemu::wave_release();
  return r;
}
