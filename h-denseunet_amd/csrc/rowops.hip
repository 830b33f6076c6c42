// rowops.hip -- the HBM-bound kernels of the hot path: per-channel reductions (BN statistics, BN
// backward sums, bias gradients), the BN(+Scale) fold / backward coefficient kernels, element-wise
// apply kernels, pooling, nearest-upsample gradient, weighted cross-entropy, Nesterov SGD and the
// 2.5D <-> 3D plumbing.  All activations are channels-last rows [M][ld]; every thread moves 16-byte
// chunks (8 bf16 / 4 f32) so a wavefront touches contiguous 1 KiB runs whenever C allows it.
#include "hdu_host.h"

// ====================================================================== per-channel reductions
enum { RED_STATS = 0, RED_BNBWD = 1, RED_COLSUM = 2 };
enum { ROW_UNROLL = 4 };   // rows of loads each thread of a row kernel keeps in flight

// optional per-channel epilogue of the finalize kernel (saves the separate bn_fold / bn_bwd_coef launches)
struct FinK {
  int kind;                 // 0 none, 1 BN fold (after RED_STATS), 2 BN backward coefficients (after RED_BNBWD)
  const float* shift_f32;   // RED_STATS: float shift array used by a conv epilogue (else the shift is row 0 of x)
  const float* gamma; const float* beta; const float* sgamma; const float* sbeta;
  float eps, momentum, invM;
  int batch_stats;
  float* a; float* b; float* rstd; float* mov_mean; float* mov_var;          // kind 1 outputs
  const float* rstd_in;                                                        // kind 2 input
  float* k1; float* k2; float* k3; float* dgamma; float* dbeta; float* dsgamma; float* dsbeta;  // kind 2 outputs
  // kind 3 (sums of a fused conv-epilogue BN backward): parameter gradients as kind 2, and the deferred part of du
  // accumulated per stored channel: corr3 += k3, corr4 += k3*mean - k2
  const float* mean_in; float* corr3; float* corr4;
  int skip_lo, skip_hi;     // kind 3: channels [skip_lo, skip_hi) leave corr3 / corr4 alone (hdu_bn_bwd_finalize_correct consumes them in the same launch)
};

// Sum of (a1, a2) over the 32 "partial lanes" of a channel in the 8-channel x 32-lane finalize geometry (thread = pl * 8 + cl):
// three xor-shuffles inside each wave (lanes 8 apart share a channel), then ONE barrier for the four waves -- the result is
// valid in the threads with pl == 0.  (Round 2 used a 5-step LDS tree with 6 barriers: ~0.5 us of a 4.8 us launch.)
__device__ __forceinline__ void fin_reduce32(double& a1, double& a2, double (*red)[4][8]) {
  const int cl = threadIdx.x & 7, wave = threadIdx.x >> 6;
#pragma unroll
  for (int mask = 8; mask < 64; mask <<= 1) {
    a1 += __shfl_xor(a1, mask);
    a2 += __shfl_xor(a2, mask);
  }
  if ((threadIdx.x & 63) < 8) { red[0][wave][cl] = a1; red[1][wave][cl] = a2; }
  __syncthreads();
  if (threadIdx.x < 8) {
    a1 = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
    a2 = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
  }
}

// per-channel parameters of a finalize launch, requested at kernel ENTRY by the thread that will use them (pl == 0) so that
// they travel together with the partial sums instead of behind them (one memory round trip less per launch)
struct FinPre { float g, be, sg, sb, mm, mv, rs, mu; };
__device__ __forceinline__ FinPre fin_prefetch(const FinK& fin, int c, bool use) {
  FinPre q;
  q.g = (use && fin.gamma) ? fin.gamma[c] : 1.f;
  q.be = (use && fin.beta) ? fin.beta[c] : 0.f;
  q.sg = (use && fin.sgamma) ? fin.sgamma[c] : 1.f;
  q.sb = (use && fin.sbeta) ? fin.sbeta[c] : 0.f;
  q.mm = (use && fin.mov_mean) ? fin.mov_mean[c] : 0.f;
  q.mv = (use && fin.mov_var) ? fin.mov_var[c] : 0.f;
  q.rs = (use && fin.rstd_in) ? fin.rstd_in[c] : 0.f;
  q.mu = (use && fin.mean_in) ? fin.mean_in[c] : 0.f;
  return q;
}

// bn_fold_channel with prefetched parameters
__device__ __forceinline__ void bn_fold_channel_pre(int c, float mu, float v, const FinK& fin, const FinPre& q) {
  const float r = 1.0f / sqrtf(v + fin.eps);
  const float inv = q.g * r;
  fin.a[c] = q.sg * inv;
  fin.b[c] = q.sg * (q.be - mu * inv) + q.sb;
  if (fin.rstd) fin.rstd[c] = r;
  if (fin.mov_mean) fin.mov_mean[c] = q.mm - (q.mm - mu) * (1.f - fin.momentum);
  if (fin.mov_var) fin.mov_var[c] = q.mv - (q.mv - v) * (1.f - fin.momentum);
}

// tf.nn.batch_normalization: x*(g*r) + (beta - mu*g*r); Scale on top: sg*y + sb
__device__ __forceinline__ void bn_fold_coef(int c, float mu, float v, const float* gamma, const float* beta, float eps,
                                             const float* sgamma, const float* sbeta, float* a, float* b, float* r_out) {
  const float r = 1.0f / sqrtf(v + eps);
  const float g = gamma ? gamma[c] : 1.f;
  const float be = beta ? beta[c] : 0.f;
  const float sg = sgamma ? sgamma[c] : 1.f;
  const float sb = sbeta ? sbeta[c] : 0.f;
  const float inv = g * r;
  *a = sg * inv;
  *b = sg * (be - mu * inv) + sb;
  *r_out = r;
}

__device__ __forceinline__ void bn_fold_channel(int c, float mu, float v, const float* gamma, const float* beta,
                                                float eps, const float* sgamma, const float* sbeta, float* a, float* b,
                                                float* rstd, float* mov_mean, float* mov_var, float momentum) {
  float ac, bc, r;
  bn_fold_coef(c, mu, v, gamma, beta, eps, sgamma, sbeta, &ac, &bc, &r);
  a[c] = ac;
  b[c] = bc;
  if (rstd) rstd[c] = r;
  if (mov_mean) mov_mean[c] -= (mov_mean[c] - mu) * (1.f - momentum);
  if (mov_var) mov_var[c] -= (mov_var[c] - v) * (1.f - momentum);
}

__device__ __forceinline__ void bn_coef_channel(int c, float invM, int batch_stats, float S1, float S2,
                                                const float* gamma, const float* beta, const float* sgamma,
                                                const float* rstd, float* k1, float* k2, float* k3, float* dgamma,
                                                float* dbeta, float* dsgamma, float* dsbeta) {
  const float g = gamma ? gamma[c] : 1.f;
  const float be = beta ? beta[c] : 0.f;
  const float sg = sgamma ? sgamma[c] : 1.f;
  const float r = rstd[c];
  const float kk = sg * g * r;
  k1[c] = kk;
  k2[c] = batch_stats ? kk * S1 * invM : 0.f;
  k3[c] = batch_stats ? kk * r * S2 * invM : 0.f;
  // y = g*xhat + beta ; z = sg*y + sb :  d sg = sum g_s*y = g*S2 + beta*S1 ; d sb = S1 ; d g = sg*S2 ; d beta = sg*S1
  if (dgamma) dgamma[c] = sg * S2;
  if (dbeta) dbeta[c] = sg * S1;
  if (dsgamma) dsgamma[c] = g * S2 + be * S1;
  if (dsbeta) dsbeta[c] = S1;
}

// one channel's totals -> outputs (+ the optional BN fold / BN-backward-coefficient epilogue)
template <typename T, int MODE>
__device__ __forceinline__ void finalize_channel(int c, double a1, double a2, long long M, const void* x, float* o1,
                                                 float* o2, const FinK& fin, const FinPre& pre) {
  if (MODE == 0) {   // RED_STATS
    const double shift = fin.shift_f32 ? (double)fin.shift_f32[c] : (double)Chunk<T>::load1((const T*)x + c);
    const double m1 = a1 / (double)M;
    double var = a2 / (double)M - m1 * m1;
    if (var < 0.0) var = 0.0;
    o1[c] = (float)(shift + m1);
    o2[c] = (float)var;
    if (fin.kind == 1) bn_fold_channel_pre(c, (float)(shift + m1), (float)var, fin, pre);
  } else {
    if (o1) o1[c] = (float)a1;
    if (o2) o2[c] = (float)a2;
    if (MODE == 1 && (fin.kind == 2 || fin.kind == 3)) {   // RED_BNBWD: coefficients + parameter gradients (prefetched parameters)
      const float S1 = (float)a1, S2 = (float)a2;
      const float kk = pre.sg * pre.g * pre.rs;
      const float k2 = fin.batch_stats ? kk * S1 * fin.invM : 0.f;
      const float k3 = fin.batch_stats ? kk * pre.rs * S2 * fin.invM : 0.f;
      // y = g*xhat + beta ; z = sg*y + sb :  d sg = sum g_s*y = g*S2 + beta*S1 ; d sb = S1 ; d g = sg*S2 ; d beta = sg*S1
      if (fin.dgamma) fin.dgamma[c] = pre.sg * S2;
      if (fin.dbeta) fin.dbeta[c] = pre.sg * S1;
      if (fin.dsgamma) fin.dsgamma[c] = pre.g * S2 + pre.be * S1;
      if (fin.dsbeta) fin.dsbeta[c] = S1;
      if (fin.kind == 2) {
        fin.k1[c] = kk; fin.k2[c] = k2; fin.k3[c] = k3;
      } else if (fin.batch_stats && !(c >= fin.skip_lo && c < fin.skip_hi)) {      // kind 3: the deferred part of du, accumulated per stored channel
        fin.corr3[c] += k3;
        fin.corr4[c] += k3 * pre.mu - k2;
      }
    }
  }
}

struct RedK {
  const void* x;
  const void* dz;
  long long ldx, lddz, M, rows_per_block;
  int C;
  int relu;
  const float* a;
  const float* b;
  const float* mean;
  const float* rstd;
  float* partial;  // [gridDim.x][2][C]; slots > 0: a ZEROED [slots][2][C] table, row (blockIdx.x % slots), float atomics
  int slots;
};

template <typename T, int MODE, int COLS>
__global__ __launch_bounds__(256) void reduce_rows_kernel(RedK p) {
  constexpr int CH = Chunk<T>::CH;
  constexpr int ROWS = 256 / COLS;
  __shared__ float red[2][ROWS][COLS * CH];
  const int tid = threadIdx.x;
  const int cc = tid % COLS;
  const int rl = tid / COLS;
  const int c0 = (blockIdx.y * COLS + cc) * CH;
  const bool active = c0 < p.C;
  const T* __restrict__ xp = (const T*)p.x;
  const T* __restrict__ dzp = (const T*)p.dz;

  float s1[CH], s2[CH], k0[CH], k1[CH], k2[CH], k3[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) { s1[j] = 0.f; s2[j] = 0.f; k0[j] = 0.f; k1[j] = 0.f; k2[j] = 0.f; k3[j] = 0.f; }
  if (active) {
    if (MODE == RED_STATS) {
      // shifted sums: the shift (row 0 of the tensor) removes the cancellation of E[x^2]-E[x]^2
      Chunk<T>::unpack(*(const u32x4*)(xp + c0), k0);
    } else if (MODE == RED_BNBWD) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        k0[j] = p.a[c0 + j]; k1[j] = p.b[c0 + j]; k2[j] = p.mean[c0 + j]; k3[j] = p.rstd[c0 + j];
      }
    }
    const long long r_begin = (long long)blockIdx.x * p.rows_per_block;
    long long r_end = r_begin + p.rows_per_block;
    if (r_end > p.M) r_end = p.M;
    // one row's contribution; the loop below keeps ROW_UNROLL rows of loads in flight per thread (a one-row loop
    // serialises on the full memory latency twice per row: measured as the limiter of every row kernel)
    auto body = [&](const u32x4& xv, const u32x4& gv) {
      float f[CH];
      Chunk<T>::unpack(xv, f);
      if (MODE == RED_STATS) {
#pragma unroll
        for (int j = 0; j < CH; ++j) { const float d = f[j] - k0[j]; s1[j] += d; s2[j] += d * d; }
      } else if (MODE == RED_BNBWD) {
        float g[CH];
        Chunk<T>::unpack(gv, g);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const float s = k0[j] * f[j] + k1[j];
          const float gg = (!p.relu || s > 0.f) ? g[j] : 0.f;
          s1[j] += gg;
          s2[j] += gg * ((f[j] - k2[j]) * k3[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < CH; ++j) s1[j] += f[j];
      }
    };
    long long r = r_begin + rl;
    for (; r + (ROW_UNROLL - 1) * ROWS < r_end; r += ROW_UNROLL * ROWS) {
      u32x4 xv[ROW_UNROLL], gv[ROW_UNROLL];
#pragma unroll
      for (int u = 0; u < ROW_UNROLL; ++u) xv[u] = *(const u32x4*)(xp + (r + u * ROWS) * p.ldx + c0);
      if (MODE == RED_BNBWD) {
#pragma unroll
        for (int u = 0; u < ROW_UNROLL; ++u) gv[u] = *(const u32x4*)(dzp + (r + u * ROWS) * p.lddz + c0);
      }
      HDU_SCHED_BARRIER();
#pragma unroll
      for (int u = 0; u < ROW_UNROLL; ++u) body(xv[u], gv[u]);
    }
    for (; r < r_end; r += ROWS) {
      u32x4 gv = u32x4{0u, 0u, 0u, 0u};
      if (MODE == RED_BNBWD) gv = *(const u32x4*)(dzp + r * p.lddz + c0);
      body(*(const u32x4*)(xp + r * p.ldx + c0), gv);
    }
  }
#pragma unroll
  for (int j = 0; j < CH; ++j) { red[0][rl][cc * CH + j] = s1[j]; red[1][rl][cc * CH + j] = s2[j]; }
  __syncthreads();
  for (int q = tid; q < 2 * COLS * CH; q += 256) {
    const int s = q / (COLS * CH);
    const int col = q % (COLS * CH);
    const int c = blockIdx.y * COLS * CH + col;
    if (c < p.C) {
      float t = 0.f;
#pragma unroll 4
      for (int r = 0; r < ROWS; ++r) t += red[s][r][col];
      if (p.slots > 0) atomicAdd(p.partial + ((long long)(blockIdx.x % (unsigned)p.slots) * 2 + s) * p.C + c, t);
      else p.partial[((long long)blockIdx.x * 2 + s) * p.C + c] = t;
    }
  }
}

// sums the per-block partials (double accumulation) and post-processes per mode.
// 256 threads = 8 channels x 32 partial lanes: lane p strides over the row blocks, then an LDS tree over the lanes.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void reduce_finalize_kernel(const float* __restrict__ partial, int nblk, int C,
                                                              long long M, const void* x, float* __restrict__ o1,
                                                              float* __restrict__ o2, FinK fin) {
  __shared__ double red[2][4][8];
  const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  const FinPre pre = fin_prefetch(fin, c < C ? c : 0, pl == 0 && c < C && fin.kind != 0);
  double a1 = 0.0, a2 = 0.0;
  if (c < C) {
    int b = pl;
    for (; b + 7 * 32 < nblk; b += 8 * 32) {      // 16 loads in flight per lane (a one-partial loop pays the latency each time)
      float v1[8], v2[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v1[u] = partial[((long long)(b + u * 32) * 2 + 0) * C + c];
        v2[u] = MODE != RED_COLSUM ? partial[((long long)(b + u * 32) * 2 + 1) * C + c] : 0.f;
      }
      HDU_SCHED_BARRIER();
#pragma unroll
      for (int u = 0; u < 8; ++u) { a1 += (double)v1[u]; a2 += (double)v2[u]; }
    }
    for (; b < nblk; b += 32) {
      a1 += (double)partial[((long long)b * 2 + 0) * C + c];
      if (MODE != RED_COLSUM) a2 += (double)partial[((long long)b * 2 + 1) * C + c];
    }
  }
  fin_reduce32(a1, a2, red);
  if (pl == 0 && c < C) finalize_channel<T, MODE>(c, a1, a2, M, x, o1, o2, fin, pre);
}

static int red_cols_for(int nchunks) {
  int cols = 4;
  while (cols < nchunks && cols < 32) cols <<= 1;
  return cols;
}
static void red_geometry(int dtype, long long M, int C, int* cols, unsigned* gx, unsigned* gy, long long* rpb) {
  const int ch = dtype == HDU_BF16 ? 8 : 4;
  const int nchunks = (C + ch - 1) / ch;
  *cols = red_cols_for(nchunks);
  *gy = (unsigned)((nchunks + *cols - 1) / *cols);
  const int rows = 256 / *cols;
  long long want = (g_tuning[HDU_TUNE_RED_WGS] > 0 ? g_tuning[HDU_TUNE_RED_WGS] : 512) / *gy;
  if (want < 1) want = 1;
  long long maxb = (M + (long long)rows * 4 - 1) / ((long long)rows * 4);  // >= 4 rows per thread
  if (maxb < 1) maxb = 1;
  if (want > maxb) want = maxb;
  *rpb = (M + want - 1) / want;
  if (*rpb < 1) *rpb = 1;
  *gx = (unsigned)((M + *rpb - 1) / *rpb);
  if (*gx < 1) *gx = 1;
}

extern "C" size_t hdu_reduce_ws_bytes(int64_t M, int C) {
  // geometry upper bound over both dtypes
  size_t best = 0;
  for (int dt = 0; dt < 2; ++dt) {
    int cols; unsigned gx, gy; long long rpb;
    red_geometry(dt, M, C, &cols, &gx, &gy, &rpb);
    size_t b = (size_t)gx * 2 * (size_t)C * sizeof(float);
    if (b > best) best = b;
  }
  return best + 256;
}

template <typename T, int MODE>
static int run_reduce(RedK k, int cols, unsigned gx, unsigned gy, hipStream_t s) {
  switch (cols) {
    case 4: HDU_LAUNCH((reduce_rows_kernel<T, MODE, 4>), dim3(gx, gy), dim3(256), 0, s, k); break;
    case 8: HDU_LAUNCH((reduce_rows_kernel<T, MODE, 8>), dim3(gx, gy), dim3(256), 0, s, k); break;
    case 16: HDU_LAUNCH((reduce_rows_kernel<T, MODE, 16>), dim3(gx, gy), dim3(256), 0, s, k); break;
    default: HDU_LAUNCH((reduce_rows_kernel<T, MODE, 32>), dim3(gx, gy), dim3(256), 0, s, k); break;
  }
  return 0;
}

template <int MODE>
static int reduce_entry(int dtype, RedK k, float* o1, float* o2, void* ws, size_t ws_bytes, hipStream_t s,
                        const char* what, FinK fin = FinK{}) {
  if (dtype != HDU_BF16 && dtype != HDU_F32) return hdu_set_error(HDU_ERR_ARG, "reduce: bad dtype");
  const int ch = dtype == HDU_BF16 ? 8 : 4;
  if (k.C <= 0 || k.C % ch || k.ldx % ch || (k.dz && k.lddz % ch))
    return hdu_set_error(HDU_ERR_ARG, "reduce: C and pixel strides must be multiples of the 16-byte chunk");
  if (k.M <= 0) return hdu_set_error(HDU_ERR_ARG, "reduce: M must be positive");
  int cols; unsigned gx, gy; long long rpb;
  red_geometry(dtype, k.M, k.C, &cols, &gx, &gy, &rpb);
  if (!ws || ws_bytes < (size_t)gx * 2 * (size_t)k.C * sizeof(float))
    return hdu_set_error(HDU_ERR_WORKSPACE, "reduce: workspace too small (see hdu_reduce_ws_bytes)");
  k.rows_per_block = rpb;
  k.partial = (float*)ws;
  const unsigned fb = (unsigned)((k.C + 7) / 8);
  if (dtype == HDU_BF16) {
    run_reduce<bf16_t, MODE>(k, cols, gx, gy, s);
    HDU_LAUNCH((reduce_finalize_kernel<bf16_t, MODE>), dim3(fb), dim3(256), 0, s, (const float*)ws, (int)gx, k.C,
               k.M, k.x, o1, o2, fin);
  } else {
    run_reduce<float, MODE>(k, cols, gx, gy, s);
    HDU_LAUNCH((reduce_finalize_kernel<float, MODE>), dim3(fb), dim3(256), 0, s, (const float*)ws, (int)gx, k.C, k.M,
               k.x, o1, o2, fin);
  }
  return hdu_check_launch(what);
}

extern "C" int hdu_bn_stats(int dtype, const void* x, int64_t ldx, int64_t M, int C, float* mean, float* var,
                            void* ws, size_t ws_bytes, void* stream) {
  if (!x || !mean || !var) return hdu_set_error(HDU_ERR_ARG, "bn_stats: null pointer");
  RedK k{};
  k.x = x; k.ldx = ldx; k.M = M; k.C = C;
  return reduce_entry<RED_STATS>(dtype, k, mean, var, ws, ws_bytes, (hipStream_t)stream, "bn_stats");
}

extern "C" int hdu_bn_stats_fold(int dtype, const void* x, int64_t ldx, int64_t M, int C, float* mean, float* var,
                                 const float* gamma, const float* beta, float eps, const float* sgamma,
                                 const float* sbeta, float* a, float* b, float* rstd, float* mov_mean, float* mov_var,
                                 float momentum, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !mean || !var || !a || !b) return hdu_set_error(HDU_ERR_ARG, "bn_stats_fold: null pointer");
  RedK k{};
  k.x = x; k.ldx = ldx; k.M = M; k.C = C;
  FinK f{};
  f.kind = 1; f.gamma = gamma; f.beta = beta; f.sgamma = sgamma; f.sbeta = sbeta; f.eps = eps; f.momentum = momentum;
  f.a = a; f.b = b; f.rstd = rstd; f.mov_mean = mov_mean; f.mov_var = mov_var;
  return reduce_entry<RED_STATS>(dtype, k, mean, var, ws, ws_bytes, (hipStream_t)stream, "bn_stats_fold", f);
}

extern "C" int hdu_bn_stats_finalize(const float* partial, int slots, int64_t M, int C, const float* shift, float* mean,
                                     float* var, const float* gamma, const float* beta, float eps, const float* sgamma,
                                     const float* sbeta, float* a, float* b, float* rstd, float* mov_mean, float* mov_var,
                                     float momentum, void* stream) {
  if (!partial || slots <= 0 || M <= 0 || C <= 0 || !shift || !mean || !var || ((a == nullptr) != (b == nullptr)))
    return hdu_set_error(HDU_ERR_ARG, "bn_stats_finalize: bad args");
  FinK f{};
  f.shift_f32 = shift;
  if (a) {
    f.kind = 1; f.gamma = gamma; f.beta = beta; f.sgamma = sgamma; f.sbeta = sbeta; f.eps = eps; f.momentum = momentum;
    f.a = a; f.b = b; f.rstd = rstd; f.mov_mean = mov_mean; f.mov_var = mov_var;
  }
  HDU_LAUNCH((reduce_finalize_kernel<float, RED_STATS>), dim3((unsigned)((C + 7) / 8)), dim3(256), 0, (hipStream_t)stream,
             partial, slots, C, (long long)M, (const void*)nullptr, mean, var, f);
  return hdu_check_launch("bn_stats_finalize");
}

// finalize of a slab SEGMENT's epilogue statistics + fold of the NEXT BatchNormalization(+Scale), whose input is the whole
// slab [0, C_all) ending with that segment (dense blocks: layer l's 3x3 conv writes channels [c, c+g); layer l+1's first BN
// reads [0, c+g)).  One thread column per slab channel: segment channels sum their slot rows first (and publish mean / var),
// the others read the stored moments; every channel then folds -- no cross-channel dependency, one launch instead of two.
__global__ __launch_bounds__(256) void finalize_fold_next_kernel(const float* __restrict__ partial, int slots, int Cseg,
                                                                 int seg_c0, int C_all, long long M,
                                                                 const float* __restrict__ shift_all,
                                                                 float* __restrict__ mean_all, float* __restrict__ var_all,
                                                                 FinK fin) {
  __shared__ double red[2][4][8];
  const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  const bool in_seg = c >= seg_c0 && c < seg_c0 + Cseg;
  const bool wg_has_seg = (int)blockIdx.x * 8 + 7 >= seg_c0 && (int)blockIdx.x * 8 < seg_c0 + Cseg;   // workgroup-uniform
  const bool mine = pl == 0 && c < C_all;
  // everything this thread will need is requested up front: parameters, the stored moments, its slot rows
  const FinPre pre = fin_prefetch(fin, mine ? c : 0, mine);
  const float mean_old = mine ? (in_seg ? shift_all[c] : mean_all[c]) : 0.f;
  const float var_old = mine ? var_all[c] : 0.f;
  double a1 = 0.0, a2 = 0.0;
  if (wg_has_seg) {
    if (in_seg) {
      const int cs = c - seg_c0;
      for (int b = pl; b < slots; b += 32) {
        a1 += (double)partial[((long long)b * 2 + 0) * Cseg + cs];
        a2 += (double)partial[((long long)b * 2 + 1) * Cseg + cs];
      }
    }
    fin_reduce32(a1, a2, red);
  }
  if (!mine) return;
  float mu, v;
  if (in_seg) {
    const double m1 = a1 / (double)M;
    double var = a2 / (double)M - m1 * m1;
    if (var < 0.0) var = 0.0;
    mu = (float)((double)mean_old + m1);             // the epilogue's shift (the mean of an earlier pass)
    v = (float)var;
    mean_all[c] = mu;
    var_all[c] = v;
  } else {
    mu = mean_old;
    v = var_old;
  }
  bn_fold_channel_pre(c, mu, v, fin, pre);
}

extern "C" int hdu_bn_stats_finalize_fold_next(const float* partial, int slots, int64_t M, int Cseg, int seg_c0, int C_all,
                                               const float* shift_all, float* mean_all, float* var_all, const float* gamma,
                                               const float* beta,
                                               float eps, const float* sgamma, const float* sbeta, float* a, float* b,
                                               float* rstd, float* mov_mean, float* mov_var, float momentum, void* stream) {
  if (!partial || slots <= 0 || M <= 0 || Cseg <= 0 || seg_c0 < 0 || C_all < seg_c0 + Cseg || !shift_all || !mean_all ||
      !var_all || !a || !b)
    return hdu_set_error(HDU_ERR_ARG, "bn_stats_finalize_fold_next: bad args");
  FinK f{};
  f.kind = 1; f.gamma = gamma; f.beta = beta; f.sgamma = sgamma; f.sbeta = sbeta; f.eps = eps; f.momentum = momentum;
  f.a = a; f.b = b; f.rstd = rstd; f.mov_mean = mov_mean; f.mov_var = mov_var;
  HDU_LAUNCH(finalize_fold_next_kernel, dim3((unsigned)((C_all + 7) / 8)), dim3(256), 0, (hipStream_t)stream, partial, slots,
             Cseg, seg_c0, C_all, (long long)M, shift_all, mean_all, var_all, f);
  return hdu_check_launch("bn_stats_finalize_fold_next");
}

extern "C" int hdu_bn_bwd_finalize(const float* partial, int slots, int64_t M, int C, int batch_stats, const float* gamma,
                                   const float* beta, const float* sgamma, const float* mean, const float* rstd,
                                   float* dgamma, float* dbeta, float* dsgamma, float* dsbeta, float* corr3, float* corr4,
                                   void* stream) {
  if (!partial || slots <= 0 || M <= 0 || C <= 0 || !rstd || (batch_stats && (!mean || !corr3 || !corr4)))
    return hdu_set_error(HDU_ERR_ARG, "bn_bwd_finalize: bad args");
  FinK f{};
  f.kind = 3; f.gamma = gamma; f.beta = beta; f.sgamma = sgamma; f.rstd_in = rstd; f.invM = 1.0f / (float)M;
  f.batch_stats = batch_stats; f.dgamma = dgamma; f.dbeta = dbeta; f.dsgamma = dsgamma; f.dsbeta = dsbeta;
  f.mean_in = mean; f.corr3 = corr3; f.corr4 = corr4;
  HDU_LAUNCH((reduce_finalize_kernel<float, RED_BNBWD>), dim3((unsigned)((C + 7) / 8)), dim3(256), 0, (hipStream_t)stream,
             partial, slots, C, (long long)M, (const void*)nullptr, (float*)nullptr, (float*)nullptr, f);
  return hdu_check_launch("bn_bwd_finalize");
}

extern "C" int hdu_bn_bwd_reduce_coef(int dtype, const void* dz, int64_t lddz, const void* x, int64_t ldx, int64_t M,
                                      int C, const float* a, const float* b, int relu, const float* mean,
                                      const float* rstd, int batch_stats, const float* gamma, const float* beta,
                                      const float* sgamma, float* s1, float* s2, float* k1, float* k2, float* k3,
                                      float* dgamma, float* dbeta, float* dsgamma, float* dsbeta, void* ws,
                                      size_t ws_bytes, void* stream) {
  if (!dz || !x || !a || !b || !mean || !rstd || !s1 || !s2 || !k1 || !k2 || !k3)
    return hdu_set_error(HDU_ERR_ARG, "bn_bwd_reduce_coef: null pointer");
  RedK k{};
  k.x = x; k.ldx = ldx; k.dz = dz; k.lddz = lddz; k.M = M; k.C = C;
  k.a = a; k.b = b; k.mean = mean; k.rstd = rstd; k.relu = relu;
  FinK f{};
  f.kind = 2; f.gamma = gamma; f.beta = beta; f.sgamma = sgamma; f.rstd_in = rstd; f.invM = 1.0f / (float)M;
  f.batch_stats = batch_stats; f.k1 = k1; f.k2 = k2; f.k3 = k3;
  f.dgamma = dgamma; f.dbeta = dbeta; f.dsgamma = dsgamma; f.dsbeta = dsbeta;
  return reduce_entry<RED_BNBWD>(dtype, k, s1, s2, ws, ws_bytes, (hipStream_t)stream, "bn_bwd_reduce_coef", f);
}

extern "C" int hdu_bn_bwd_reduce(int dtype, const void* dz, int64_t lddz, const void* x, int64_t ldx, int64_t M,
                                 int C, const float* a, const float* b, int relu, const float* mean,
                                 const float* rstd, float* s1, float* s2, void* ws, size_t ws_bytes, void* stream) {
  if (!dz || !x || !a || !b || !mean || !rstd || !s1 || !s2) return hdu_set_error(HDU_ERR_ARG, "bn_bwd_reduce: null pointer");
  RedK k{};
  k.x = x; k.ldx = ldx; k.dz = dz; k.lddz = lddz; k.M = M; k.C = C;
  k.a = a; k.b = b; k.mean = mean; k.rstd = rstd; k.relu = relu;
  return reduce_entry<RED_BNBWD>(dtype, k, s1, s2, ws, ws_bytes, (hipStream_t)stream, "bn_bwd_reduce");
}

extern "C" int hdu_colsum(int dtype, const void* x, int64_t ldx, int64_t M, int C, float* out, void* ws,
                          size_t ws_bytes, void* stream) {
  if (!x || !out) return hdu_set_error(HDU_ERR_ARG, "colsum: null pointer");
  RedK k{};
  k.x = x; k.ldx = ldx; k.M = M; k.C = C;
  return reduce_entry<RED_COLSUM>(dtype, k, out, nullptr, ws, ws_bytes, (hipStream_t)stream, "colsum");
}

// ====================================================================== BN fold / backward coefficients
__global__ __launch_bounds__(256) void bn_fold_kernel(int C, const float* mean, const float* var, const float* gamma,
                                                      const float* beta, float eps, const float* sgamma,
                                                      const float* sbeta, float* a, float* b, float* rstd,
                                                      float* mov_mean, float* mov_var, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  bn_fold_channel(c, mean[c], var[c], gamma, beta, eps, sgamma, sbeta, a, b, rstd, mov_mean, mov_var, momentum);
}

extern "C" int hdu_bn_fold(int C, const float* mean, const float* var, const float* gamma, const float* beta,
                           float eps, const float* sgamma, const float* sbeta, float* a, float* b, float* rstd,
                           float* mov_mean, float* mov_var, float momentum, void* stream) {
  if (C <= 0 || !mean || !var || !a || !b) return hdu_set_error(HDU_ERR_ARG, "bn_fold: bad args");
  HDU_LAUNCH(bn_fold_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, C, mean, var,
             gamma, beta, eps, sgamma, sbeta, a, b, rstd, mov_mean, mov_var, momentum);
  return hdu_check_launch("bn_fold");
}

// all inference-mode folds of a pass in one launch: block -> (entry, 256-channel block) by binary search over begins[]
__global__ __launch_bounds__(256) void bn_fold_batched_kernel(const hdu_fold_entry* __restrict__ table,
                                                              const unsigned* __restrict__ begins, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (begins[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const hdu_fold_entry e = table[lo];
  const int c = (int)(blockIdx.x - begins[lo]) * 256 + (int)threadIdx.x;
  if (c >= e.C) return;
  bn_fold_channel(c, e.mean[c], e.var[c], e.gamma, e.beta, e.eps, e.sgamma, e.sbeta, e.a, e.b, e.rstd, nullptr, nullptr,
                  0.f);
}

extern "C" int hdu_bn_fold_batched(const hdu_fold_entry* table, const uint32_t* begins, int n, uint32_t total_blocks,
                                   void* stream) {
  if (!table || !begins || n <= 0 || total_blocks == 0) return hdu_set_error(HDU_ERR_ARG, "bn_fold_batched: bad args");
  HDU_LAUNCH(bn_fold_batched_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, table, begins, n);
  return hdu_check_launch("bn_fold_batched");
}

// parameter gradients of MANY inference-mode BN(+Scale) layers from the slot sums their data-gradient epilogues left: one
// launch at the end of the backward pass instead of one ~4.6 us finalize per layer inside it (nothing in the backward chain
// reads them: frozen statistics have no k2 / k3 terms).  Geometry of reduce_finalize_kernel: 8 channels x 32 slot lanes.
__global__ __launch_bounds__(256) void bn_bwd_finalize_batched_kernel(const hdu_bnbwd_entry* __restrict__ table,
                                                                      const unsigned* __restrict__ begins, int n) {
  __shared__ double red[2][4][8];
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (begins[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const hdu_bnbwd_entry e = table[lo];
  const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int c = (int)(blockIdx.x - begins[lo]) * 8 + cl;
  const bool mine = pl == 0 && c < e.C;
  const float g = (mine && e.gamma) ? e.gamma[c] : 1.f, be = (mine && e.beta) ? e.beta[c] : 0.f;
  const float sg = (mine && e.sgamma) ? e.sgamma[c] : 1.f;
  double a1 = 0.0, a2 = 0.0;
  if (c < e.C) {
    for (int b = pl; b < e.slots; b += 32) {
      a1 += (double)e.partial[((long long)b * 2 + 0) * e.C + c];
      a2 += (double)e.partial[((long long)b * 2 + 1) * e.C + c];
    }
  }
  fin_reduce32(a1, a2, red);
  if (!mine) return;
  const float S1 = (float)a1, S2 = (float)a2;
  if (e.dgamma) e.dgamma[c] = sg * S2;
  if (e.dbeta) e.dbeta[c] = sg * S1;
  if (e.dsgamma) e.dsgamma[c] = g * S2 + be * S1;
  if (e.dsbeta) e.dsbeta[c] = S1;
}

extern "C" int hdu_bn_bwd_finalize_batched(const hdu_bnbwd_entry* table, const uint32_t* begins, int n, uint32_t total_blocks,
                                           void* stream) {
  if (!table || !begins || n <= 0 || total_blocks == 0) return hdu_set_error(HDU_ERR_ARG, "bn_bwd_finalize_batched: bad args");
  HDU_LAUNCH(bn_bwd_finalize_batched_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, table, begins, n);
  return hdu_check_launch("bn_bwd_finalize_batched");
}

__global__ __launch_bounds__(256) void bn_bwd_coef_kernel(int C, float invM, int batch_stats, const float* s1,
                                                          const float* s2, const float* gamma, const float* beta,
                                                          const float* sgamma, const float* rstd, float* k1,
                                                          float* k2, float* k3, float* dgamma, float* dbeta,
                                                          float* dsgamma, float* dsbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  bn_coef_channel(c, invM, batch_stats, s1 ? s1[c] : 0.f, s2 ? s2[c] : 0.f, gamma, beta, sgamma, rstd, k1, k2, k3,
                  dgamma, dbeta, dsgamma, dsbeta);
}

extern "C" int hdu_bn_bwd_coef(int C, int64_t M, int batch_stats, const float* s1, const float* s2,
                               const float* gamma, const float* beta, const float* sgamma, const float* rstd,
                               float* k1, float* k2, float* k3, float* dgamma, float* dbeta, float* dsgamma,
                               float* dsbeta, void* stream) {
  if (C <= 0 || M <= 0 || !rstd || !k1 || !k2 || !k3) return hdu_set_error(HDU_ERR_ARG, "bn_bwd_coef: bad args");
  if (batch_stats && (!s1 || !s2)) return hdu_set_error(HDU_ERR_ARG, "bn_bwd_coef: batch_stats needs s1/s2");
  HDU_LAUNCH(bn_bwd_coef_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, C,
             1.0f / (float)M, batch_stats, s1, s2, gamma, beta, sgamma, rstd, k1, k2, k3, dgamma, dbeta, dsgamma,
             dsbeta);
  return hdu_check_launch("bn_bwd_coef");
}

// sync-BN of a depth-sharded volume (shard.py).  Every rank writes (n_i, mean_i, var_i) into ITS slot of a zeroed
// [world][1 + 2C] table; a sum all-reduce of the table is then an all-gather (x + 0 is exact), and every rank combines the
// slots in rank order with the pairwise-moments formula
//     mean = sum n_i mean_i / N,    var = sum n_i (var_i + (mean_i - mean)^2) / N
// -- no E[x^2] - E[x]^2: round 2 all-reduced (n mean, n (var + mean^2)) and subtracted mean^2 afterwards, which loses
// mean^2 / var relative precision in float32 (a channel with |mean| = 30 sigma: 1e-4 of its variance, amplified by every
// BatchNormalization behind it -- found by tests/test_depth_shard_gloo.py[3dpart] in round 3).
__global__ __launch_bounds__(256) void stats_pack_kernel(int C, const float* mean, const float* var, float n_local, int rank,
                                                         int world, float* buf) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int SL = 1 + 2 * C;
  if (c >= C) return;
  for (int r = 0; r < world; ++r) {
    const bool mine = r == rank;
    if (c == 0) buf[r * SL] = mine ? n_local : 0.f;
    buf[r * SL + 1 + c] = mine ? mean[c] : 0.f;
    buf[r * SL + 1 + C + c] = mine ? var[c] : 0.f;
  }
}
__global__ __launch_bounds__(256) void stats_unpack_kernel(int C, const float* buf, int world, float* mean, float* var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int SL = 1 + 2 * C;
  if (c >= C) return;
  float N = 0.f, m = 0.f;
  for (int r = 0; r < world; ++r) { N += buf[r * SL]; m += buf[r * SL] * buf[r * SL + 1 + c]; }
  m /= N;
  float v = 0.f;
  for (int r = 0; r < world; ++r) {
    const float d = buf[r * SL + 1 + c] - m;
    v += buf[r * SL] * (buf[r * SL + 1 + C + c] + d * d);
  }
  mean[c] = m;
  var[c] = v / N;
}
extern "C" size_t hdu_stats_sync_floats(int C, int world) { return (size_t)world * (1 + 2 * (size_t)C); }
extern "C" int hdu_stats_pack(int C, const float* mean, const float* var, int64_t n_local, int rank, int world, float* buf,
                              void* stream) {
  if (C <= 0 || !mean || !var || !buf || n_local <= 0 || world <= 0 || rank < 0 || rank >= world)
    return hdu_set_error(HDU_ERR_ARG, "stats_pack: bad args");
  HDU_LAUNCH(stats_pack_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, C, mean, var,
             (float)n_local, rank, world, buf);
  return hdu_check_launch("stats_pack");
}
extern "C" int hdu_stats_unpack(int C, const float* buf, int world, float* mean, float* var, void* stream) {
  if (C <= 0 || !mean || !var || !buf || world <= 0) return hdu_set_error(HDU_ERR_ARG, "stats_unpack: bad args");
  HDU_LAUNCH(stats_unpack_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, C, buf, world, mean,
             var);
  return hdu_check_launch("stats_unpack");
}

// ====================================================================== element-wise row kernels
struct RowK {
  const void* x;
  const void* dz;
  void* out;
  long long ldx, lddz, ldo, M, rows_per_block;
  int C;
  int relu, accumulate;
  const float* a;
  const float* b;
  const float* mean;
  const float* k1;
  const float* k2;
  const float* k3;
  float drop_scale;
  unsigned drop_thresh, drop_seed;
  const unsigned* drop_seed_dev;
  // bn_bwd_apply_kernel<.., SUMS = true> (hdu_bn_bwd_fused): the coefficients come from the reduction's slot sums
  const float* sums;          // [slots][2][C]
  int slots, batch_stats;
  float invM;
  const float* gamma; const float* beta; const float* sgamma; const float* rstd;
  float* dgamma; float* dbeta; float* dsgamma; float* dsbeta;
};

// Column totals of a [slots][2][C] slot table for the thread's own CH channels.  The ROWS row lanes of a column chunk split
// the slot rows (one memory round trip for the whole table part of the workgroup), meet in LDS, and every lane then adds the
// per-lane partials in a fixed order.  ALL 256 threads call it; lanes without a column pass active = false.
// Two halves so that a caller can issue its first data rows between them: slot_loads() only requests, slot_reduce() meets at a
// RAW barrier (s_waitcnt lgkmcnt(0) + s_barrier: __syncthreads() carries a workgroup release fence, which on gfx9 is
// s_waitcnt vmcnt(0) and would drain the caller's data loads too -- loads and stores share vmcnt).
template <int CH, int COLS>
__device__ __forceinline__ void slot_loads(const float* __restrict__ tbl, int slots, int C, int c0, bool active, int rl,
                                           float (&t1)[CH], float (&t2)[CH]) {
  constexpr int ROWS = 256 / COLS;
  constexpr int IT = 4;          // slots <= 32 <= IT * ROWS (ROWS >= 8): every slot row of this lane is requested at once --
  // a runtime-count loop issues one row, waits, adds, and only then requests the next: 2-4 serial round trips per launch
  f32x4 v1[IT][CH / 4], v2[IT][CH / 4];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int sl = rl + it * ROWS;
    const long long row = (active && sl < slots) ? sl : 0;      // unconditional, clamped (see igemm_epilogue)
#pragma unroll
    for (int j = 0; j < CH; j += 4) {
      v1[it][j / 4] = *(const f32x4*)(tbl + (row * 2 + 0) * C + c0 + j);
      v2[it][j / 4] = *(const f32x4*)(tbl + (row * 2 + 1) * C + c0 + j);
    }
  }
#pragma unroll
  for (int j = 0; j < CH; ++j) { t1[j] = 0.f; t2[j] = 0.f; }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const bool ok = active && rl + it * ROWS < slots;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      t1[j] += ok ? v1[it][j / 4][j % 4] : 0.f;
      t2[j] += ok ? v2[it][j / 4][j % 4] : 0.f;
    }
  }
}
template <int CH, int COLS>
__device__ __forceinline__ void slot_reduce(int slots, int cc, int rl, const float (&t1)[CH], const float (&t2)[CH],
                                            float (&s1)[CH], float (&s2)[CH]) {
  constexpr int ROWS = 256 / COLS;
  __shared__ float red[ROWS][2][COLS * CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) { red[rl][0][cc * CH + j] = t1[j]; red[rl][1][cc * CH + j] = t2[j]; }
  HDU_RAW_BARRIER();
  const int nr = slots < ROWS ? slots : ROWS;
#pragma unroll
  for (int j = 0; j < CH; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  for (int r = 0; r < nr; ++r) {
#pragma unroll
    for (int j = 0; j < CH; ++j) { s1[j] += red[r][0][cc * CH + j]; s2[j] += red[r][1][cc * CH + j]; }
  }
}

// Row kernels: thread (cc, rl) owns 16-byte channel chunk cc of its column group for the whole launch (coefficients
// live in registers) and strides over the rows of its row block.
template <typename T, int COLS>
__global__ __launch_bounds__(256) void affine_act_kernel(RowK p) {
  constexpr int CH = Chunk<T>::CH;
  constexpr int ROWS = 256 / COLS;
  const int cc = threadIdx.x % COLS, rl = threadIdx.x / COLS;
  const int c0 = (blockIdx.y * COLS + cc) * CH;
  if (c0 >= p.C) return;
  const T* __restrict__ xp = (const T*)p.x;
  T* __restrict__ op = (T*)p.out;
  float a[CH], b[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) { a[j] = p.a ? p.a[c0 + j] : 1.f; b[j] = p.a ? p.b[c0 + j] : 0.f; }
  const long long r_begin = (long long)blockIdx.x * p.rows_per_block;
  long long r_end = r_begin + p.rows_per_block;
  if (r_end > p.M) r_end = p.M;
  auto body = [&](long long m, const u32x4& xv) {
    float f[CH];
    Chunk<T>::unpack(xv, f);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      float s = a[j] * f[j] + b[j];
      if (p.relu) s = s > 0.f ? s : 0.f;
      f[j] = s;
    }
    *(u32x4*)(op + m * p.ldo + c0) = Chunk<T>::pack(f);
  };
  long long m = r_begin + rl;
  for (; m + (ROW_UNROLL - 1) * ROWS < r_end; m += ROW_UNROLL * ROWS) {
    u32x4 xv[ROW_UNROLL];
#pragma unroll
    for (int u = 0; u < ROW_UNROLL; ++u) xv[u] = *(const u32x4*)(xp + (m + u * ROWS) * p.ldx + c0);
    HDU_SCHED_BARRIER();
#pragma unroll
    for (int u = 0; u < ROW_UNROLL; ++u) body(m + u * ROWS, xv[u]);
  }
  for (; m < r_end; m += ROWS) body(m, *(const u32x4*)(xp + m * p.ldx + c0));
}

// PRE (round 6; SUMS launches whose threads own at most ROW_UNROLL rows -- the M <= 8192 dense layers, 60 of the 2D net's 78): the
// thread's rows are requested BEFORE the slot-table round trip, so the launch is ONE memory round trip instead of two or three (with
// 2 rows per thread the main loop below never ran and its tail loop took the rows one round trip at a time).  Its own instantiation:
// the 24-36 registers the rows occupy during the prologue cost the LARGE layers an occupancy step (measured in round 3: 2D +0.3 ms when
// every launch did it) -- a launch of <= 512 workgroups of this size has no occupancy to lose.
template <typename T, int COLS, bool SUMS = false, bool PRE = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(RowK p) {
  constexpr int CH = Chunk<T>::CH;
  constexpr int ROWS = 256 / COLS;
  const int cc = threadIdx.x % COLS, rl = threadIdx.x / COLS;
  const int c0 = (blockIdx.y * COLS + cc) * CH;
  const bool active = c0 < p.C;
  if (!SUMS && !active) return;
  const T* __restrict__ xp = (const T*)p.x;
  const T* __restrict__ dzp = (const T*)p.dz;
  T* __restrict__ op = (T*)p.out;
  const unsigned dseed = p.drop_seed + (p.drop_seed_dev ? *p.drop_seed_dev : 0u);
  const long long r_begin = (long long)blockIdx.x * p.rows_per_block;
  long long r_end = r_begin + p.rows_per_block;
  if (r_end > p.M) r_end = p.M;
  const u32x4 z4 = u32x4{0u, 0u, 0u, 0u};
  long long m = r_begin + rl;
  u32x4 pxv[PRE ? ROW_UNROLL : 1], pgv[PRE ? ROW_UNROLL : 1], pov[PRE ? ROW_UNROLL : 1];
  if constexpr (PRE) {
#pragma unroll
    for (int u = 0; u < ROW_UNROLL; ++u) {
      const long long mm = m + u * ROWS;
      const bool ok = active && mm < r_end;
      pxv[u] = ok ? *(const u32x4*)(xp + mm * p.ldx + c0) : z4;
      pgv[u] = ok ? *(const u32x4*)(dzp + mm * p.lddz + c0) : z4;
      pov[u] = (ok && p.accumulate) ? *(const u32x4*)(op + mm * p.ldo + c0) : z4;
    }
  }
  // dx = k1*g - k2 - k3*(x-mean) = k1*g - k3*x + k4,  k4 = k3*mean - k2
  float a[CH], b[CH], k1[CH], k3[CH], k4[CH];
  if constexpr (SUMS) {
    // hdu_bn_bwd_fused: no finalize launch -- this workgroup sums the reduction's slot rows for its own channels and derives
    // the coefficients in registers (bn_coef_channel's formulas); its first row block also writes the parameter gradients
    const int cq = active ? c0 : 0;
    float g[CH], be[CH], sg[CH], rs[CH], mu[CH];
#pragma unroll
    for (int j = 0; j < CH; j += 4) {                // parameters requested before the slot rows: one round trip together
      const f32x4 va = *(const f32x4*)(p.a + cq + j), vb = *(const f32x4*)(p.b + cq + j);
      const f32x4 vr = *(const f32x4*)(p.rstd + cq + j), vm = *(const f32x4*)(p.mean + cq + j);
      const f32x4 vg = p.gamma ? *(const f32x4*)(p.gamma + cq + j) : f32x4{1.f, 1.f, 1.f, 1.f};
      const f32x4 ve = p.beta ? *(const f32x4*)(p.beta + cq + j) : f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 vs = p.sgamma ? *(const f32x4*)(p.sgamma + cq + j) : f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a[j + r] = va[r]; b[j + r] = vb[r]; rs[j + r] = vr[r]; mu[j + r] = vm[r];
        g[j + r] = vg[r]; be[j + r] = ve[r]; sg[j + r] = vs[r];
      }
    }
    float T1[CH], T2[CH], S1[CH], S2[CH];
    slot_loads<CH, COLS>(p.sums, p.slots, p.C, cq, active, rl, T1, T2);
    slot_reduce<CH, COLS>(p.slots, cc, rl, T1, T2, S1, S2);
    if (!active) return;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const float kk = sg[j] * g[j] * rs[j];
      const float k2 = p.batch_stats ? kk * S1[j] * p.invM : 0.f;
      k1[j] = kk;
      k3[j] = p.batch_stats ? kk * rs[j] * S2[j] * p.invM : 0.f;
      k4[j] = k3[j] * mu[j] - k2;
    }
    if (blockIdx.x == 0 && rl == 0) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int c = c0 + j;
        if (p.dgamma) p.dgamma[c] = sg[j] * S2[j];
        if (p.dbeta) p.dbeta[c] = sg[j] * S1[j];
        if (p.dsgamma) p.dsgamma[c] = g[j] * S2[j] + be[j] * S1[j];
        if (p.dsbeta) p.dsbeta[c] = S1[j];
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int c = c0 + j;
      a[j] = p.a[c]; b[j] = p.b[c]; k1[j] = p.k1[c]; k3[j] = p.k3[c];
      k4[j] = p.k3[c] * p.mean[c] - p.k2[c];
    }
  }
  auto body = [&](long long m, const u32x4& xv, const u32x4& gv, const u32x4& ov) {
    float f[CH], g[CH], o[CH];
    Chunk<T>::unpack(xv, f);
    Chunk<T>::unpack(gv, g);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const float s = a[j] * f[j] + b[j];
      const float gg = (!p.relu || s > 0.f) ? g[j] : 0.f;
      float d = k1[j] * gg - k3[j] * f[j] + k4[j];
      if (p.drop_scale != 0.f) {
        const unsigned h = hdu_hash32((unsigned long long)m * (unsigned)p.C + (unsigned)(c0 + j), dseed);
        d = h < p.drop_thresh ? d * p.drop_scale : 0.f;
      }
      o[j] = d;
    }
    if (p.accumulate) {
      float old[CH];
      Chunk<T>::unpack(ov, old);
#pragma unroll
      for (int j = 0; j < CH; ++j) o[j] += old[j];
    }
    *(u32x4*)(op + m * p.ldo + c0) = Chunk<T>::pack(o);
  };
  if constexpr (PRE) {                        // (the host launches this form only when ROW_UNROLL rows per thread cover the row block)
#pragma unroll
    for (int u = 0; u < ROW_UNROLL; ++u)
      if (m + u * ROWS < r_end) body(m + u * ROWS, pxv[u], pgv[u], pov[u]);
    return;
  }
  for (; m + (ROW_UNROLL - 1) * ROWS < r_end; m += ROW_UNROLL * ROWS) {
    u32x4 xv[ROW_UNROLL], gv[ROW_UNROLL], ov[ROW_UNROLL];
#pragma unroll
    for (int u = 0; u < ROW_UNROLL; ++u) xv[u] = *(const u32x4*)(xp + (m + u * ROWS) * p.ldx + c0);
#pragma unroll
    for (int u = 0; u < ROW_UNROLL; ++u) gv[u] = *(const u32x4*)(dzp + (m + u * ROWS) * p.lddz + c0);
#pragma unroll
    for (int u = 0; u < ROW_UNROLL; ++u) ov[u] = p.accumulate ? *(const u32x4*)(op + (m + u * ROWS) * p.ldo + c0) : z4;
    HDU_SCHED_BARRIER();
#pragma unroll
    for (int u = 0; u < ROW_UNROLL; ++u) body(m + u * ROWS, xv[u], gv[u], ov[u]);
  }
  for (; m < r_end; m += ROWS)
    body(m, *(const u32x4*)(xp + m * p.ldx + c0), *(const u32x4*)(dzp + m * p.lddz + c0),
         p.accumulate ? *(const u32x4*)(op + m * p.ldo + c0) : z4);
}

// du += -corr3*u + corr4 (hdu_bn_bwd_correct): the deferred part of the BN backward of every consumer of these channels
template <typename T, int COLS>
__global__ __launch_bounds__(256) void bn_bwd_correct_kernel(RowK p) {
  constexpr int CH = Chunk<T>::CH;
  constexpr int ROWS = 256 / COLS;
  const int cc = threadIdx.x % COLS, rl = threadIdx.x / COLS;
  const int c0 = (blockIdx.y * COLS + cc) * CH;
  if (c0 >= p.C) return;
  const T* __restrict__ xp = (const T*)p.x;
  T* __restrict__ op = (T*)p.out;
  float k3[CH], k4[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) { k3[j] = p.k3[c0 + j]; k4[j] = p.k2[c0 + j]; }
  const long long r_begin = (long long)blockIdx.x * p.rows_per_block;
  long long r_end = r_begin + p.rows_per_block;
  if (r_end > p.M) r_end = p.M;
  auto body = [&](long long m, const u32x4& xv, const u32x4& ov) {
    float f[CH], o[CH];
    Chunk<T>::unpack(xv, f);
    Chunk<T>::unpack(ov, o);
#pragma unroll
    for (int j = 0; j < CH; ++j) o[j] += k4[j] - k3[j] * f[j];
    *(u32x4*)(op + m * p.ldo + c0) = Chunk<T>::pack(o);
  };
  long long m = r_begin + rl;
  for (; m + (ROW_UNROLL - 1) * ROWS < r_end; m += ROW_UNROLL * ROWS) {
    u32x4 xv[ROW_UNROLL], ov[ROW_UNROLL];
#pragma unroll
    for (int u = 0; u < ROW_UNROLL; ++u) xv[u] = *(const u32x4*)(xp + (m + u * ROWS) * p.ldx + c0);
#pragma unroll
    for (int u = 0; u < ROW_UNROLL; ++u) ov[u] = *(const u32x4*)(op + (m + u * ROWS) * p.ldo + c0);
    HDU_SCHED_BARRIER();
#pragma unroll
    for (int u = 0; u < ROW_UNROLL; ++u) body(m + u * ROWS, xv[u], ov[u]);
  }
  for (; m < r_end; m += ROWS) body(m, *(const u32x4*)(xp + m * p.ldx + c0), *(const u32x4*)(op + m * p.ldo + c0));
}

// hdu_bn_bwd_finalize_correct (round 4): ONE launch for two things that follow each other in the backward chain of a dense
// block -- the finalize of the fused BN backward of layer i (slot sums -> parameter gradients, corr3 / corr4 += for all its
// channels) and the correction  du += -corr3 * u + corr4  of the 48 channels layer i-1 produced, right before layer i-1 reads
// its gradient.  Workgroups [0, nfin): the ordinary finalize, 8 channels each, except that they leave the accumulators of the
// corrected channels [cs0, cs0 + Cc) alone.  Workgroups nfin ..: row blocks of the correction; each derives the FINAL
// coefficients of its own channel chunk itself: the stored accumulators (all earlier consumers) + this BN's share from the
// slot sums (32 slot rows x 16 values per chunk: the row lanes take one slot each, an LDS column sum in slot order).
struct FinCorK {
  const float* partial; int slots; int C; long long M;      // the BN's slot sums [slots][2][C]
  FinK fin;
  int nfin;
  int cs0, Cc;                                              // corrected channels, relative to the BN's channel 0
  const void* u; void* du; long long ldu, lddu, rows_per_block;
};

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_finalize_correct_kernel(FinCorK p) {
  constexpr int CH = Chunk<T>::CH;
  __shared__ double red[2][4][8];
  __shared__ float slotv[32][8][2 * CH];                    // [slot][chunk column][S1 x CH | S2 x CH]
  __shared__ float tot[8][2 * CH];
  if ((int)blockIdx.x < p.nfin) {                           // ---- finalize part (reduce_finalize_kernel<float, RED_BNBWD>)
    if (blockIdx.y != 0) return;                            // (the grid's second dimension belongs to the correction part)
    const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cl;
    const FinPre pre = fin_prefetch(p.fin, c < p.C ? c : 0, pl == 0 && c < p.C);
    double a1 = 0.0, a2 = 0.0;
    if (c < p.C)
      for (int b = pl; b < p.slots; b += 32) {
        a1 += (double)p.partial[((long long)b * 2 + 0) * p.C + c];
        a2 += (double)p.partial[((long long)b * 2 + 1) * p.C + c];
      }
    fin_reduce32(a1, a2, red);
    if (pl == 0 && c < p.C) finalize_channel<float, RED_BNBWD>(c, a1, a2, p.M, nullptr, nullptr, nullptr, p.fin, pre);
    return;
  }
  // ---- correction part: 8 chunk columns x 32 row lanes; blockIdx.y = group of 8 chunk columns
  const int cc = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c0 = (blockIdx.y * 8 + cc) * CH;                // relative to cs0
  const bool active = c0 < p.Cc;
  const int cb = p.cs0 + (active ? c0 : 0);                 // BN channel of this thread's chunk
  {
    float v[2 * CH];
#pragma unroll
    for (int j = 0; j < 2 * CH; ++j) v[j] = 0.f;
    if (active && rl < p.slots) {
#pragma unroll
      for (int j = 0; j < CH; j += 4) {
        const f32x4 q1 = *(const f32x4*)(p.partial + ((long long)rl * 2 + 0) * p.C + cb + j);
        const f32x4 q2 = *(const f32x4*)(p.partial + ((long long)rl * 2 + 1) * p.C + cb + j);
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[j + r] = q1[r]; v[CH + j + r] = q2[r]; }
      }
    }
#pragma unroll
    for (int j = 0; j < 2 * CH; ++j) slotv[rl][cc][j] = v[j];
  }
  // parameters of the chunk, requested before the barrier
  float g[CH], sg[CH], rs[CH], mu[CH], o3[CH], o4[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    g[j] = p.fin.gamma ? p.fin.gamma[cb + j] : 1.f;
    sg[j] = p.fin.sgamma ? p.fin.sgamma[cb + j] : 1.f;
    rs[j] = p.fin.rstd_in[cb + j];
    mu[j] = p.fin.mean_in[cb + j];
    o3[j] = p.fin.corr3[cb + j];
    o4[j] = p.fin.corr4[cb + j];
  }
  __syncthreads();
  if (threadIdx.x < 8 * 2 * CH) {                           // thread (column cq, value vq): sum over the slots in slot order
    const int cq = threadIdx.x / (2 * CH), vq = threadIdx.x % (2 * CH);
    float t = 0.f;
    for (int b = 0; b < p.slots && b < 32; ++b) t += slotv[b][cq][vq];
    tot[cq][vq] = t;
  }
  __syncthreads();
  if (!active) return;
  float k3[CH], k4[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const float S1 = tot[cc][j], S2 = tot[cc][CH + j];
    const float kk = sg[j] * g[j] * rs[j];
    const float k2 = kk * S1 * p.fin.invM;
    const float kk3 = kk * rs[j] * S2 * p.fin.invM;
    k3[j] = o3[j] + kk3;
    k4[j] = o4[j] + (kk3 * mu[j] - k2);
  }
  // (Round 6: requesting the first ROW_UNROLL rows before the slot sums meet, as bn_bwd_apply_kernel's PRE form does, was measured
  // here too -- same-box A/B, profiles/r06_experiment_rows_first.txt: 2D +0.18 ms against the gain of the other two kernels; not kept.)
  const T* __restrict__ xp = (const T*)p.u;
  T* __restrict__ op = (T*)p.du;
  const long long r_begin = (long long)(blockIdx.x - p.nfin) * p.rows_per_block;
  long long r_end = r_begin + p.rows_per_block;
  if (r_end > p.M) r_end = p.M;
  auto body = [&](long long m, const u32x4& xv, const u32x4& ov) {
    float f[CH], o[CH];
    Chunk<T>::unpack(xv, f);
    Chunk<T>::unpack(ov, o);
#pragma unroll
    for (int j = 0; j < CH; ++j) o[j] += k4[j] - k3[j] * f[j];
    *(u32x4*)(op + m * p.lddu + c0) = Chunk<T>::pack(o);
  };
  long long m = r_begin + rl;
  for (; m + (ROW_UNROLL - 1) * 32 < r_end; m += ROW_UNROLL * 32) {
    u32x4 xv[ROW_UNROLL], ov[ROW_UNROLL];
#pragma unroll
    for (int q = 0; q < ROW_UNROLL; ++q) xv[q] = *(const u32x4*)(xp + (m + q * 32) * p.ldu + c0);
#pragma unroll
    for (int q = 0; q < ROW_UNROLL; ++q) ov[q] = *(const u32x4*)(op + (m + q * 32) * p.lddu + c0);
    HDU_SCHED_BARRIER();
#pragma unroll
    for (int q = 0; q < ROW_UNROLL; ++q) body(m + q * 32, xv[q], ov[q]);
  }
  for (; m < r_end; m += 32) body(m, *(const u32x4*)(xp + m * p.ldu + c0), *(const u32x4*)(op + m * p.lddu + c0));
}

// geometry shared by the row kernels: column groups of COLS chunks, row blocks sized to ~512 workgroups
static void row_geometry(int dtype, long long M, int C, int* cols, unsigned* gx, unsigned* gy, long long* rpb) {
  const int ch = dtype == HDU_BF16 ? 8 : 4;
  const int nchunks = C / ch;
  int c = 4;
  while (c < nchunks && c < 32) c <<= 1;
  *cols = c;
  *gy = (unsigned)((nchunks + c - 1) / c);
  const int rows = 256 / c;
  // ~2 workgroups per CU (swept in round 3 after the statistics / coefficient prologues moved into these kernels:
  // 256 / 384 / 512 / 768 / 1024 / 2048 / 4096 -> 2D step 19.92 / 19.60 / 19.18 / 19.30 / 19.62 / 19.69 / 20.01 ms; rounds 1-2: 2048)
  long long want = (g_tuning[HDU_TUNE_ROW_WGS] > 0 ? g_tuning[HDU_TUNE_ROW_WGS] : 512) / *gy;
  if (want < 1) want = 1;
  long long maxb = (M + (long long)rows * 4 - 1) / ((long long)rows * 4);
  if (maxb < 1) maxb = 1;
  if (want > maxb) want = maxb;
  *rpb = (M + want - 1) / want;
  *gx = (unsigned)((M + *rpb - 1) / *rpb);
}

#define HDU_ROW_LAUNCH(kern, T, cols, gx, gy, stream, k)                                                  \
  switch (cols) {                                                                                         \
    case 4: HDU_LAUNCH((kern<T, 4>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, k); break;         \
    case 8: HDU_LAUNCH((kern<T, 8>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, k); break;         \
    case 16: HDU_LAUNCH((kern<T, 16>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, k); break;       \
    default: HDU_LAUNCH((kern<T, 32>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, k); break;       \
  }

struct MatK {
  const void* x;
  const void* skip;
  void* out;
  const float* a;
  const float* b;
  long long ldx, ldskip, ldo, Mo, rows_per_block;
  int N, D, H, W, C;
  int ud, uh, uw, relu;
  // materialize_kernel<.., STATS = true> (hdu_materialize_stats): the BatchNormalization is folded HERE from the epilogue sums
  // of the conv that wrote the segment [seg_c0, seg_c0 + Cseg) of its input channels (the other channels: stored moments)
  const float* st_partial;    // [st_slots][2][Cseg]
  int st_slots, Cseg, seg_c0;
  long long st_M;
  const float* st_shift;      // [C]
  float* st_mean; float* st_var;   // [C]
  const float* gamma; const float* beta; const float* sgamma; const float* sbeta;
  float eps, momentum;
  float* fa; float* fb; float* frstd; float* mov_mean; float* mov_var;
};

// PRE: as bn_bwd_apply_kernel's -- STATS launches without up-sampling / skip whose threads own at most ROW_UNROLL rows
template <typename T, int COLS, bool STATS = false, bool PRE = false>
__global__ __launch_bounds__(256) void materialize_kernel(MatK p) {
  constexpr int CH = Chunk<T>::CH;
  constexpr int ROWS = 256 / COLS;
  const int cc = threadIdx.x % COLS, rl = threadIdx.x / COLS;
  const int c0 = (blockIdx.y * COLS + cc) * CH;
  const bool active = c0 < p.C;
  if (!STATS && !active) return;
  const T* __restrict__ xp = (const T*)p.x;
  const T* __restrict__ sp = (const T*)p.skip;
  T* __restrict__ op = (T*)p.out;
  const int We = p.W << p.uw, He = p.H << p.uh, De = p.D << p.ud;
  const bool ups = (p.ud | p.uh | p.uw) != 0;
  const long long r_begin = (long long)blockIdx.x * p.rows_per_block;
  long long r_end = r_begin + p.rows_per_block;
  if (r_end > p.Mo) r_end = p.Mo;
  long long m = r_begin + rl;
  int w = 0, h = 0, d = 0, n = 0;
  if (ups && m < r_end) {
    w = (int)(m % We);
    long long t = m / We;
    h = (int)(t % He);
    t /= He;
    d = (int)(t % De);
    n = (int)(t / De);
  }
  auto src_of = [&]() -> long long {
    return ups ? ((((long long)n * p.D + (d >> p.ud)) * p.H + (h >> p.uh)) * p.W + (w >> p.uw)) : m;
  };
  auto advance = [&]() {
    m += ROWS;
    if (ups) {
      w += ROWS;
      while (w >= We) {
        w -= We;
        if (++h == He) {
          h = 0;
          if (++d == De) {
            d = 0;
            ++n;
          }
        }
      }
    }
  };
  const u32x4 z4 = u32x4{0u, 0u, 0u, 0u};
  u32x4 pxv[PRE ? ROW_UNROLL : 1];
  if constexpr (PRE) {
#pragma unroll
    for (int u = 0; u < ROW_UNROLL; ++u) {
      const long long mm = m + u * ROWS;
      pxv[u] = (active && mm < r_end) ? *(const u32x4*)(xp + mm * p.ldx + c0) : z4;
    }
  }
  float a[CH], b[CH];
  if constexpr (STATS) {
    // (Round 3: requesting the thread's first data rows BEFORE the workgroup meets in EVERY launch -- round 6 does it in the PRE
    // instantiation, for the small launches only.  The round-3 note:)
    // (Requesting the thread's first data rows BEFORE the workgroup meets, so that the two round trips overlap, was measured
    // slower in three forms -- branch-guarded, unconditional, and unconditional behind the slot rows with a raw barrier and a
    // scheduling barrier: the 16-32 extra live registers cost the large layers an occupancy step; 2D 20.46 -> 20.74 ms.)
    // No finalize launch between the producing conv and this pass: every workgroup turns the slot sums of its own segment
    // channels into (mean, var) -- or reads the stored moments of the other channels --, folds the BN(+Scale) in registers,
    // and the first row block publishes a / b / rstd / the moments / the moving averages for the backward pass and the
    // later layers.  Segment bounds are multiples of the 16-byte chunk: a thread's channels are all inside or all outside.
    const int cq = active ? c0 : 0;
    const bool in_seg = active && c0 >= p.seg_c0 && c0 < p.seg_c0 + p.Cseg;
    float g[CH], be[CH], sg[CH], sb[CH], mu[CH], vv[CH], mm[CH], mv[CH];
    const bool pub = blockIdx.x == 0 && rl == 0;
#pragma unroll
    for (int j = 0; j < CH; j += 4) {                  // everything requested up front: one memory round trip with the slot rows
      const f32x4 one = f32x4{1.f, 1.f, 1.f, 1.f}, zero = f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 vg = p.gamma ? *(const f32x4*)(p.gamma + cq + j) : one, ve = p.beta ? *(const f32x4*)(p.beta + cq + j) : zero;
      const f32x4 vs = p.sgamma ? *(const f32x4*)(p.sgamma + cq + j) : one, vb = p.sbeta ? *(const f32x4*)(p.sbeta + cq + j) : zero;
      const f32x4 vm = *(const f32x4*)((in_seg ? p.st_shift : (const float*)p.st_mean) + cq + j);
      const f32x4 vr = *(const f32x4*)(p.st_var + cq + j);
      const f32x4 v1 = (pub && p.mov_mean) ? *(const f32x4*)(p.mov_mean + cq + j) : zero;
      const f32x4 v2 = (pub && p.mov_var) ? *(const f32x4*)(p.mov_var + cq + j) : zero;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        g[j + r] = vg[r]; be[j + r] = ve[r]; sg[j + r] = vs[r]; sb[j + r] = vb[r]; mu[j + r] = vm[r]; vv[j + r] = vr[r];
        mm[j + r] = v1[r]; mv[j + r] = v2[r];
      }
    }
    float T1[CH], T2[CH], S1[CH], S2[CH];
    slot_loads<CH, COLS>(p.st_partial, p.st_slots, p.Cseg, in_seg ? c0 - p.seg_c0 : 0, in_seg, rl, T1, T2);
    slot_reduce<CH, COLS>(p.st_slots, cc, rl, T1, T2, S1, S2);
    if (!active) return;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      if (in_seg) {                                    // sums of (y - shift), (y - shift)^2: shift = the previous pass's mean
        const double m1 = (double)S1[j] / (double)p.st_M;
        double var = (double)S2[j] / (double)p.st_M - m1 * m1;
        if (var < 0.0) var = 0.0;
        mu[j] = (float)((double)mu[j] + m1);
        vv[j] = (float)var;
      }
      const float r = 1.0f / sqrtf(vv[j] + p.eps);
      const float inv = g[j] * r;
      a[j] = sg[j] * inv;
      b[j] = sg[j] * (be[j] - mu[j] * inv) + sb[j];
      if (pub) {
        const int c = c0 + j;
        p.fa[c] = a[j]; p.fb[c] = b[j];
        if (p.frstd) p.frstd[c] = r;
        if (in_seg) { p.st_mean[c] = mu[j]; p.st_var[c] = vv[j]; }
        if (p.mov_mean) p.mov_mean[c] = mm[j] - (mm[j] - mu[j]) * (1.f - p.momentum);
        if (p.mov_var) p.mov_var[c] = mv[j] - (mv[j] - vv[j]) * (1.f - p.momentum);
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < CH; ++j) { a[j] = p.a ? p.a[c0 + j] : 1.f; b[j] = p.a ? p.b[c0 + j] : 0.f; }
  }
  auto body = [&](long long mo, const u32x4& xv, const u32x4& sv) {
    float f[CH];
    Chunk<T>::unpack(xv, f);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      float s = a[j] * f[j] + b[j];
      if (p.relu) s = s > 0.f ? s : 0.f;
      f[j] = s;
    }
    if (sp) {
      float g[CH];
      Chunk<T>::unpack(sv, g);
#pragma unroll
      for (int j = 0; j < CH; ++j) f[j] += g[j];
    }
    *(u32x4*)(op + mo * p.ldo + c0) = Chunk<T>::pack(f);
  };
  if constexpr (PRE) {
#pragma unroll
    for (int u = 0; u < ROW_UNROLL; ++u)
      if (m + u * ROWS < r_end) body(m + u * ROWS, pxv[u], z4);
    return;
  }
  while (m + (ROW_UNROLL - 1) * ROWS < r_end) {
    u32x4 xv[ROW_UNROLL], sv[ROW_UNROLL];
    long long mo[ROW_UNROLL];
#pragma unroll
    for (int u = 0; u < ROW_UNROLL; ++u) {
      mo[u] = m;
      xv[u] = *(const u32x4*)(xp + src_of() * p.ldx + c0);
      advance();
    }
#pragma unroll
    for (int u = 0; u < ROW_UNROLL; ++u) sv[u] = sp ? *(const u32x4*)(sp + mo[u] * p.ldskip + c0) : z4;
    HDU_SCHED_BARRIER();
#pragma unroll
    for (int u = 0; u < ROW_UNROLL; ++u) body(mo[u], xv[u], sv[u]);
  }
  while (m < r_end) {
    const long long mo = m;
    const u32x4 xv = *(const u32x4*)(xp + src_of() * p.ldx + c0);
    const u32x4 sv = sp ? *(const u32x4*)(sp + mo * p.ldskip + c0) : z4;
    body(mo, xv, sv);
    advance();
  }
}

static int rowk_check(int dtype, const RowK& k, const char* what) {
  if (dtype != HDU_BF16 && dtype != HDU_F32) return hdu_set_error(HDU_ERR_ARG, what);
  const int ch = dtype == HDU_BF16 ? 8 : 4;
  if (k.C <= 0 || k.C % ch || k.ldx % ch || k.ldo % ch || (k.dz && k.lddz % ch) || k.M < 0)
    return hdu_set_error(HDU_ERR_ARG, what);
  return 0;
}

extern "C" int hdu_affine_act(int dtype, const void* x, int64_t ldx, int64_t M, int C, const float* a,
                              const float* b, int relu, void* z, int64_t ldz, void* stream) {
  RowK k{};
  k.x = x; k.ldx = ldx; k.out = z; k.ldo = ldz; k.M = M; k.C = C; k.a = a; k.b = b; k.relu = relu;
  if (!x || !z || ((a == nullptr) != (b == nullptr))) return hdu_set_error(HDU_ERR_ARG, "affine_act: bad pointers");
  if (int e = rowk_check(dtype, k, "affine_act: C / strides must be multiples of the 16-byte chunk")) return e;
  if (M == 0) return 0;
  int cols; unsigned gx, gy;
  row_geometry(dtype, M, C, &cols, &gx, &gy, &k.rows_per_block);
  if (dtype == HDU_BF16) { HDU_ROW_LAUNCH(affine_act_kernel, bf16_t, cols, gx, gy, stream, k); }
  else { HDU_ROW_LAUNCH(affine_act_kernel, float, cols, gx, gy, stream, k); }
  return hdu_check_launch("affine_act");
}

extern "C" int hdu_materialize(int dtype, const void* x, int64_t ldx, int N, int D, int H, int W, int C, const float* a,
                               const float* b, int relu, int ud, int uh, int uw, const void* skip, int64_t ldskip,
                               void* out, int64_t ldout, void* stream) {
  if (dtype != HDU_BF16 && dtype != HDU_F32) return hdu_set_error(HDU_ERR_ARG, "materialize: bad dtype");
  const int ch = dtype == HDU_BF16 ? 8 : 4;
  if (!x || !out || ((a == nullptr) != (b == nullptr)) || ((ud | uh | uw) & ~1))
    return hdu_set_error(HDU_ERR_ARG, "materialize: bad pointers / upsample shifts");
  if (C <= 0 || C % ch || ldx % ch || ldout % ch || (skip && ldskip % ch) || N <= 0 || D <= 0 || H <= 0 || W <= 0)
    return hdu_set_error(HDU_ERR_ARG, "materialize: C / strides must be multiples of the 16-byte chunk");
  MatK k{};
  k.x = x; k.skip = skip; k.out = out; k.a = a; k.b = b; k.ldx = ldx; k.ldskip = ldskip; k.ldo = ldout;
  k.N = N; k.D = D; k.H = H; k.W = W; k.C = C; k.ud = ud; k.uh = uh; k.uw = uw; k.relu = relu;
  k.Mo = (long long)N * (D << ud) * (H << uh) * (W << uw);
  int cols; unsigned gx, gy;
  row_geometry(dtype, k.Mo, C, &cols, &gx, &gy, &k.rows_per_block);
  if (dtype == HDU_BF16) { HDU_ROW_LAUNCH(materialize_kernel, bf16_t, cols, gx, gy, stream, k); }
  else { HDU_ROW_LAUNCH(materialize_kernel, float, cols, gx, gy, stream, k); }
  return hdu_check_launch("materialize");
}

extern "C" int hdu_materialize_stats(int dtype, const void* x, int64_t ldx, int N, int D, int H, int W, int C,
                                     const hdu_stats_fold_desc* f, int relu, int ud, int uh, int uw, const void* skip,
                                     int64_t ldskip, void* out, int64_t ldout, void* stream) {
  if (dtype != HDU_BF16 && dtype != HDU_F32) return hdu_set_error(HDU_ERR_ARG, "materialize_stats: bad dtype");
  const int ch = dtype == HDU_BF16 ? 8 : 4;
  if (!x || !out || !f || ((ud | uh | uw) & ~1)) return hdu_set_error(HDU_ERR_ARG, "materialize_stats: bad pointers / upsample shifts");
  if (C <= 0 || C % ch || ldx % ch || ldout % ch || (skip && ldskip % ch) || N <= 0 || D <= 0 || H <= 0 || W <= 0)
    return hdu_set_error(HDU_ERR_ARG, "materialize_stats: C / strides must be multiples of the 16-byte chunk");
  if (!f->partial || f->slots <= 0 || f->slots > 32 || f->M <= 0 || f->Cseg <= 0 || f->Cseg % ch || f->seg_c0 < 0 ||
      f->seg_c0 % ch || f->seg_c0 + f->Cseg > C || !f->shift || !f->mean || !f->var || !f->a || !f->b)
    return hdu_set_error(HDU_ERR_ARG, "materialize_stats: bad statistics block (segment bounds are multiples of the 16-byte chunk)");
  if (((uintptr_t)f->partial | (uintptr_t)f->shift | (uintptr_t)f->mean | (uintptr_t)f->var | (uintptr_t)f->gamma |
       (uintptr_t)f->beta | (uintptr_t)f->sgamma | (uintptr_t)f->sbeta | (uintptr_t)f->mov_mean | (uintptr_t)f->mov_var) & 15)
    return hdu_set_error(HDU_ERR_ARG, "materialize_stats: per-channel vectors must be 16-byte aligned");
  MatK k{};
  k.x = x; k.skip = skip; k.out = out; k.ldx = ldx; k.ldskip = ldskip; k.ldo = ldout;
  k.N = N; k.D = D; k.H = H; k.W = W; k.C = C; k.ud = ud; k.uh = uh; k.uw = uw; k.relu = relu;
  k.Mo = (long long)N * (D << ud) * (H << uh) * (W << uw);
  k.st_partial = f->partial; k.st_slots = f->slots; k.Cseg = f->Cseg; k.seg_c0 = f->seg_c0; k.st_M = f->M;
  k.st_shift = f->shift; k.st_mean = f->mean; k.st_var = f->var;
  k.gamma = f->gamma; k.beta = f->beta; k.sgamma = f->sgamma; k.sbeta = f->sbeta; k.eps = f->eps; k.momentum = f->momentum;
  k.fa = f->a; k.fb = f->b; k.frstd = f->rstd; k.mov_mean = f->mov_mean; k.mov_var = f->mov_var;
  int cols; unsigned gx, gy;
  row_geometry(dtype, k.Mo, C, &cols, &gx, &gy, &k.rows_per_block);
  const bool pre = k.rows_per_block <= (long long)ROW_UNROLL * (256 / cols) && !(ud | uh | uw) && skip == nullptr &&
                   !(g_tuning[HDU_TUNE_DEBUG] & 2048);
#define HDU_MAT_STATS(T, PRE)                                                                                                   \
  switch (cols) {                                                                                                               \
    case 4: HDU_LAUNCH((materialize_kernel<T, 4, true, PRE>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, k); break;       \
    case 8: HDU_LAUNCH((materialize_kernel<T, 8, true, PRE>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, k); break;       \
    case 16: HDU_LAUNCH((materialize_kernel<T, 16, true, PRE>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, k); break;     \
    default: HDU_LAUNCH((materialize_kernel<T, 32, true, PRE>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, k); break;     \
  }
  if (dtype == HDU_BF16) { if (pre) { HDU_MAT_STATS(bf16_t, true) } else { HDU_MAT_STATS(bf16_t, false) } }
  else { if (pre) { HDU_MAT_STATS(float, true) } else { HDU_MAT_STATS(float, false) } }
#undef HDU_MAT_STATS
  return hdu_check_launch("materialize_stats");
}

extern "C" int hdu_bn_bwd_apply(int dtype, const void* dz, int64_t lddz, const void* x, int64_t ldx, int64_t M,
                                int C, const float* a, const float* b, int relu, const float* mean, const float* k1,
                                const float* k2, const float* k3, void* dx, int64_t lddx, int accumulate,
                                float drop_keep, uint32_t drop_seed, const uint32_t* drop_seed_dev, void* stream) {
  RowK k{};
  k.x = x; k.ldx = ldx; k.dz = dz; k.lddz = lddz; k.out = dx; k.ldo = lddx; k.M = M; k.C = C;
  k.a = a; k.b = b; k.relu = relu; k.mean = mean; k.k1 = k1; k.k2 = k2; k.k3 = k3; k.accumulate = accumulate;
  if (drop_keep > 0.f && drop_keep < 1.f) {
    k.drop_scale = 1.f / drop_keep;
    k.drop_thresh = (unsigned)((double)drop_keep * 4294967296.0);
  }
  k.drop_seed = drop_seed;
  k.drop_seed_dev = drop_seed_dev;
  if (!x || !dz || !dx || !a || !b || !mean || !k1 || !k2 || !k3) return hdu_set_error(HDU_ERR_ARG, "bn_bwd_apply: null pointer");
  if (int e = rowk_check(dtype, k, "bn_bwd_apply: C / strides must be multiples of the 16-byte chunk")) return e;
  if (M == 0) return 0;
  int cols; unsigned gx, gy;
  row_geometry(dtype, M, C, &cols, &gx, &gy, &k.rows_per_block);
  if (dtype == HDU_BF16) { HDU_ROW_LAUNCH(bn_bwd_apply_kernel, bf16_t, cols, gx, gy, stream, k); }
  else { HDU_ROW_LAUNCH(bn_bwd_apply_kernel, float, cols, gx, gy, stream, k); }
  return hdu_check_launch("bn_bwd_apply");
}

static int bn_bwd_fused_impl(bool reduce, int dtype, const void* dz, int64_t lddz, const void* x, int64_t ldx, int64_t M, int C,
                             const float* a, const float* b, int relu, const float* mean, const float* rstd,
                             int batch_stats, const float* gamma, const float* beta, const float* sgamma, float* sums,
                             int slots, float* dgamma, float* dbeta, float* dsgamma, float* dsbeta, void* dx,
                             int64_t lddx, int accumulate, float drop_keep, uint32_t drop_seed,
                             const uint32_t* drop_seed_dev, void* stream) {
  if (!dz || !x || !a || !b || !mean || !rstd || !sums || !dx) return hdu_set_error(HDU_ERR_ARG, "bn_bwd_fused: null pointer");
  if (slots <= 0 || slots > 32) return hdu_set_error(HDU_ERR_ARG, "bn_bwd_fused: 1 <= slots <= 32");
  if (((uintptr_t)sums | (uintptr_t)a | (uintptr_t)b | (uintptr_t)mean | (uintptr_t)rstd | (uintptr_t)gamma | (uintptr_t)beta |
       (uintptr_t)sgamma) & 15)
    return hdu_set_error(HDU_ERR_ARG, "bn_bwd_fused: per-channel vectors must be 16-byte aligned");
  RowK k{};
  k.x = x; k.ldx = ldx; k.dz = dz; k.lddz = lddz; k.out = dx; k.ldo = lddx; k.M = M; k.C = C;
  k.a = a; k.b = b; k.relu = relu; k.mean = mean; k.accumulate = accumulate;
  if (drop_keep > 0.f && drop_keep < 1.f) {
    k.drop_scale = 1.f / drop_keep;
    k.drop_thresh = (unsigned)((double)drop_keep * 4294967296.0);
  }
  k.drop_seed = drop_seed;
  k.drop_seed_dev = drop_seed_dev;
  k.sums = sums; k.slots = slots; k.batch_stats = batch_stats; k.invM = 1.0f / (float)M;
  k.gamma = gamma; k.beta = beta; k.sgamma = sgamma; k.rstd = rstd;
  k.dgamma = dgamma; k.dbeta = dbeta; k.dsgamma = dsgamma; k.dsbeta = dsbeta;
  if (int e = rowk_check(dtype, k, "bn_bwd_fused: C / strides must be multiples of the 16-byte chunk")) return e;
  if (M == 0) return 0;
  // launch 1: column sums of (g, g * xhat) into the slot rows (hdu_bn_bwd_apply_sums: the caller's data-gradient epilogue did it)
  if (reduce) {
    RedK r{};
    r.x = x; r.ldx = ldx; r.dz = dz; r.lddz = lddz; r.M = M; r.C = C;
    r.a = a; r.b = b; r.mean = mean; r.rstd = rstd; r.relu = relu;
    r.partial = sums; r.slots = slots;
    int rcols; unsigned rgx, rgy;
    red_geometry(dtype, M, C, &rcols, &rgx, &rgy, &r.rows_per_block);
    if (dtype == HDU_BF16) run_reduce<bf16_t, RED_BNBWD>(r, rcols, rgx, rgy, (hipStream_t)stream);
    else run_reduce<float, RED_BNBWD>(r, rcols, rgx, rgy, (hipStream_t)stream);
  }
  // launch 2: coefficients from the sums + dx
  int cols; unsigned gx, gy;
  row_geometry(dtype, M, C, &cols, &gx, &gy, &k.rows_per_block);
  // few rows per thread: the form that requests them before the slot rows (bn_bwd_apply_kernel PRE); HDU_TUNE_DEBUG bit 11: off (A/B)
  const bool pre = k.rows_per_block <= (long long)ROW_UNROLL * (256 / cols) && !(g_tuning[HDU_TUNE_DEBUG] & 2048);
#define HDU_APPLY_SUMS(T, PRE)                                                                                                    \
  switch (cols) {                                                                                                                 \
    case 4: HDU_LAUNCH((bn_bwd_apply_kernel<T, 4, true, PRE>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, k); break;        \
    case 8: HDU_LAUNCH((bn_bwd_apply_kernel<T, 8, true, PRE>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, k); break;        \
    case 16: HDU_LAUNCH((bn_bwd_apply_kernel<T, 16, true, PRE>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, k); break;      \
    default: HDU_LAUNCH((bn_bwd_apply_kernel<T, 32, true, PRE>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, k); break;      \
  }
  if (dtype == HDU_BF16) { if (pre) { HDU_APPLY_SUMS(bf16_t, true) } else { HDU_APPLY_SUMS(bf16_t, false) } }
  else { if (pre) { HDU_APPLY_SUMS(float, true) } else { HDU_APPLY_SUMS(float, false) } }
#undef HDU_APPLY_SUMS
  return hdu_check_launch("bn_bwd_fused");
}

extern "C" int hdu_bn_bwd_fused(int dtype, const void* dz, int64_t lddz, const void* x, int64_t ldx, int64_t M, int C,
                                const float* a, const float* b, int relu, const float* mean, const float* rstd,
                                int batch_stats, const float* gamma, const float* beta, const float* sgamma, float* sums,
                                int slots, float* dgamma, float* dbeta, float* dsgamma, float* dsbeta, void* dx,
                                int64_t lddx, int accumulate, float drop_keep, uint32_t drop_seed,
                                const uint32_t* drop_seed_dev, void* stream) {
  return bn_bwd_fused_impl(true, dtype, dz, lddz, x, ldx, M, C, a, b, relu, mean, rstd, batch_stats, gamma, beta, sgamma, sums, slots,
                           dgamma, dbeta, dsgamma, dsbeta, dx, lddx, accumulate, drop_keep, drop_seed, drop_seed_dev, stream);
}

// Round 6: the apply half alone -- `sums` was filled by the epilogue of the data-gradient launch that produced dz
// (hdu_conv_desc.bnb_relu bit 2: raw dz stored, S1 / S2 into bnb_partial = sums, bnb_slots = slots): ONE launch per BN backward.
extern "C" int hdu_bn_bwd_apply_sums(int dtype, const void* dz, int64_t lddz, const void* x, int64_t ldx, int64_t M, int C,
                                     const float* a, const float* b, int relu, const float* mean, const float* rstd,
                                     int batch_stats, const float* gamma, const float* beta, const float* sgamma, float* sums,
                                     int slots, float* dgamma, float* dbeta, float* dsgamma, float* dsbeta, void* dx,
                                     int64_t lddx, int accumulate, float drop_keep, uint32_t drop_seed,
                                     const uint32_t* drop_seed_dev, void* stream) {
  return bn_bwd_fused_impl(false, dtype, dz, lddz, x, ldx, M, C, a, b, relu, mean, rstd, batch_stats, gamma, beta, sgamma, sums, slots,
                           dgamma, dbeta, dsgamma, dsbeta, dx, lddx, accumulate, drop_keep, drop_seed, drop_seed_dev, stream);
}

extern "C" int hdu_bn_bwd_correct(int dtype, const void* u, int64_t ldu, int64_t M, int C, const float* corr3,
                                  const float* corr4, void* du, int64_t lddu, void* stream) {
  RowK k{};
  k.x = u; k.ldx = ldu; k.out = du; k.ldo = lddu; k.M = M; k.C = C;
  k.k3 = corr3; k.k2 = corr4;          // the kernel reads corr3 through k3 and corr4 through k2
  if (!u || !du || !corr3 || !corr4) return hdu_set_error(HDU_ERR_ARG, "bn_bwd_correct: null pointer");
  if (int e = rowk_check(dtype, k, "bn_bwd_correct: C / strides must be multiples of the 16-byte chunk")) return e;
  if (M == 0) return 0;
  int cols; unsigned gx, gy;
  row_geometry(dtype, M, C, &cols, &gx, &gy, &k.rows_per_block);
  if (dtype == HDU_BF16) { HDU_ROW_LAUNCH(bn_bwd_correct_kernel, bf16_t, cols, gx, gy, stream, k); }
  else { HDU_ROW_LAUNCH(bn_bwd_correct_kernel, float, cols, gx, gy, stream, k); }
  return hdu_check_launch("bn_bwd_correct");
}

extern "C" int hdu_bn_bwd_finalize_correct(int dtype, const float* partial, int slots, int64_t M, int C, const float* gamma,
                                           const float* beta, const float* sgamma, const float* mean, const float* rstd,
                                           float* dgamma, float* dbeta, float* dsgamma, float* dsbeta, float* corr3, float* corr4,
                                           int cs0, int Cc, const void* u, int64_t ldu, void* du, int64_t lddu, void* stream) {
  const int ch = dtype == HDU_BF16 ? 8 : 4;
  if ((dtype != HDU_BF16 && dtype != HDU_F32) || !partial || slots <= 0 || slots > 32 || M <= 0 || C <= 0 || !rstd || !mean || !corr3 ||
      !corr4 || !u || !du || cs0 < 0 || Cc <= 0 || cs0 + Cc > C || cs0 % ch || Cc % ch || C % 4 || ldu % ch || lddu % ch ||
      ((uintptr_t)u | (uintptr_t)du | (uintptr_t)partial) % 16)
    return hdu_set_error(HDU_ERR_ARG, "bn_bwd_finalize_correct: bad args (corrected channels inside the BN's, chunk-aligned; <= 32 slots)");
  FinCorK k{};
  k.partial = partial; k.slots = slots; k.C = C; k.M = M;
  k.fin.kind = 3; k.fin.gamma = gamma; k.fin.beta = beta; k.fin.sgamma = sgamma; k.fin.rstd_in = rstd; k.fin.invM = 1.0f / (float)M;
  k.fin.batch_stats = 1; k.fin.dgamma = dgamma; k.fin.dbeta = dbeta; k.fin.dsgamma = dsgamma; k.fin.dsbeta = dsbeta;
  k.fin.mean_in = mean; k.fin.corr3 = corr3; k.fin.corr4 = corr4; k.fin.skip_lo = cs0; k.fin.skip_hi = cs0 + Cc;
  k.nfin = (C + 7) / 8;
  k.cs0 = cs0; k.Cc = Cc; k.u = u; k.du = du; k.ldu = ldu; k.lddu = lddu;
  const unsigned gy = (unsigned)((Cc / ch + 7) / 8);
  long long want = (g_tuning[HDU_TUNE_ROW_WGS] > 0 ? g_tuning[HDU_TUNE_ROW_WGS] : 512) / gy;
  long long maxb = (M + 127) / 128;                        // >= 4 rows per row lane
  if (want > maxb) want = maxb;
  if (want < 1) want = 1;
  k.rows_per_block = (M + want - 1) / want;
  const unsigned gxc = (unsigned)((M + k.rows_per_block - 1) / k.rows_per_block);
  if (dtype == HDU_BF16) HDU_LAUNCH((bn_bwd_finalize_correct_kernel<bf16_t>), dim3((unsigned)k.nfin + gxc, gy), dim3(256), 0, (hipStream_t)stream, k);
  else HDU_LAUNCH((bn_bwd_finalize_correct_kernel<float>), dim3((unsigned)k.nfin + gxc, gy), dim3(256), 0, (hipStream_t)stream, k);
  return hdu_check_launch("bn_bwd_finalize_correct");
}

// ====================================================================== pooling / resampling
struct PoolK {
  const void* x;
  const void* dy;
  void* out;
  unsigned char* idx;
  long long ldx, lddy, ldo;
  int N, D, H, W, C;      // input dims
  int Do, Ho, Wo;
  int accumulate;
  int ud, uh, uw;
  int pad_d;               // max pool: zero padding on the depth axis (0 when neighbour planes are supplied as halo)
};

// forward: also records, per output element, which window tap won (255 = the zero padding): the backward pass
// then needs one byte + the output gradient per covering window instead of re-reading the whole window.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(PoolK p) {
  constexpr int CH = Chunk<T>::CH;
  const int ncc = p.C / CH;
  const long long total = (long long)p.N * p.Do * p.Ho * p.Wo * ncc;
  const T* __restrict__ xp = (const T*)p.x;
  T* __restrict__ op = (T*)p.out;
  const int kdn = p.D == 1 ? 1 : 3, pdd = p.D == 1 ? 0 : p.pad_d, sdd = p.D == 1 ? 1 : 2;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total;
       q += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(q % ncc) * CH;
    long long t = q / ncc;
    const long long opix = t;
    const int ow = (int)(t % p.Wo); t /= p.Wo;
    const int oh = (int)(t % p.Ho); t /= p.Ho;
    const int od = (int)(t % p.Do);
    const int n = (int)(t / p.Do);
    float best[CH];
    int bi[CH];
    bool any_pad = false;
#pragma unroll
    for (int j = 0; j < CH; ++j) { best[j] = -3.0e38f; bi[j] = 255; }
    for (int kd = 0; kd < kdn; ++kd) {
      const int id = od * sdd - pdd + kd;
      for (int kh = 0; kh < 3; ++kh) {
        const int ih = oh * 2 - 1 + kh;
        for (int kw = 0; kw < 3; ++kw) {
          const int iw = ow * 2 - 1 + kw;
          if ((unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) {
            float f[CH];
            Chunk<T>::unpack(*(const u32x4*)(xp + ((((long long)n * p.D + id) * p.H + ih) * p.W + iw) * p.ldx + c0), f);
            const int tap = (kd * 3 + kh) * 3 + kw;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
              if (f[j] > best[j]) { best[j] = f[j]; bi[j] = tap; }
            }
          } else {
            any_pad = true;
          }
        }
      }
    }
    if (any_pad) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        if (!(best[j] > 0.f)) { best[j] = 0.f; bi[j] = 255; }   // the explicit zero padding competes (and wins ties)
      }
    }
    *(u32x4*)(op + opix * p.ldo + c0) = Chunk<T>::pack(best);
    if (p.idx) {
      unsigned char* ip = p.idx + opix * p.C + c0;
#pragma unroll
      for (int j = 0; j < CH; ++j) ip[j] = (unsigned char)bi[j];
    }
  }
}

// backward from the recorded argmax: each input pixel visits the <=2 (x2 x2) windows covering it.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(PoolK p) {
  constexpr int CH = Chunk<T>::CH;
  const int ncc = p.C / CH;
  const long long total = (long long)p.N * p.D * p.H * p.W * ncc;
  const T* __restrict__ dyp = (const T*)p.dy;
  T* __restrict__ op = (T*)p.out;
  const int pdd = p.D == 1 ? 0 : p.pad_d, sdd = p.D == 1 ? 1 : 2;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total;
       q += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(q % ncc) * CH;
    long long t = q / ncc;
    const long long ipix = t;
    const int iw = (int)(t % p.W); t /= p.W;
    const int ih = (int)(t % p.H); t /= p.H;
    const int id = (int)(t % p.D);
    const int n = (int)(t / p.D);
    float acc[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) acc[j] = 0.f;
    const int od_lo = p.D == 1 ? 0 : (id + pdd - 2 > 0 ? (id + pdd - 1) / 2 : 0), od_hi = p.D == 1 ? 0 : (id + pdd) / 2;
    const int oh_lo = ih / 2, oh_hi = (ih + 1) / 2;
    const int ow_lo = iw / 2, ow_hi = (iw + 1) / 2;
    for (int od = od_lo; od <= od_hi && od < p.Do; ++od)
      for (int oh = oh_lo; oh <= oh_hi && oh < p.Ho; ++oh)
        for (int ow = ow_lo; ow <= ow_hi && ow < p.Wo; ++ow) {
          const int tap = ((id - (od * sdd - pdd)) * 3 + (ih - (oh * 2 - 1))) * 3 + (iw - (ow * 2 - 1));
          const long long opix = (((long long)n * p.Do + od) * p.Ho + oh) * p.Wo + ow;
          const unsigned char* ip = p.idx + opix * p.C + c0;
          float g[CH];
          Chunk<T>::unpack(*(const u32x4*)(dyp + opix * p.lddy + c0), g);
#pragma unroll
          for (int j = 0; j < CH; ++j) acc[j] += (ip[j] == tap) ? g[j] : 0.f;
        }
    T* dst = op + ipix * p.ldo + c0;
    if (p.accumulate) {
      float old[CH];
      Chunk<T>::unpack(*(const u32x4*)dst, old);
#pragma unroll
      for (int j = 0; j < CH; ++j) acc[j] += old[j];
    }
    *(u32x4*)dst = Chunk<T>::pack(acc);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(PoolK p) {
  constexpr int CH = Chunk<T>::CH;
  const int ncc = p.C / CH;
  const long long total = (long long)p.N * p.D * p.Ho * p.Wo * ncc;
  const T* __restrict__ xp = (const T*)p.x;
  T* __restrict__ op = (T*)p.out;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total;
       q += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(q % ncc) * CH;
    long long t = q / ncc;
    const long long opix = t;
    const int ow = (int)(t % p.Wo); t /= p.Wo;
    const int oh = (int)(t % p.Ho); t /= p.Ho;   // t = n*D + d
    const long long base = (t * p.H + oh * 2) * p.W + ow * 2;
    float s[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) s[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float f[CH];
      Chunk<T>::unpack(*(const u32x4*)(xp + (base + (i >> 1) * p.W + (i & 1)) * p.ldx + c0), f);
#pragma unroll
      for (int j = 0; j < CH; ++j) s[j] += f[j];
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) s[j] *= 0.25f;
    *(u32x4*)(op + opix * p.ldo + c0) = Chunk<T>::pack(s);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(PoolK p) {
  constexpr int CH = Chunk<T>::CH;
  const int ncc = p.C / CH;
  const long long total = (long long)p.N * p.D * p.H * p.W * ncc;
  const T* __restrict__ dyp = (const T*)p.dy;
  T* __restrict__ op = (T*)p.out;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total;
       q += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(q % ncc) * CH;
    long long t = q / ncc;
    const long long ipix = t;
    const int iw = (int)(t % p.W); t /= p.W;
    const int ih = (int)(t % p.H); t /= p.H;
    float g[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) g[j] = 0.f;
    if ((ih >> 1) < p.Ho && (iw >> 1) < p.Wo) {
      Chunk<T>::unpack(*(const u32x4*)(dyp + ((t * p.Ho + (ih >> 1)) * p.Wo + (iw >> 1)) * p.lddy + c0), g);
#pragma unroll
      for (int j = 0; j < CH; ++j) g[j] *= 0.25f;
    }
    T* dst = op + ipix * p.ldo + c0;
    if (p.accumulate) {
      float old[CH];
      Chunk<T>::unpack(*(const u32x4*)dst, old);
#pragma unroll
      for (int j = 0; j < CH; ++j) g[j] += old[j];
    }
    *(u32x4*)dst = Chunk<T>::pack(g);
  }
}

// dz[n,d,h,w] = sum over children of the up-sampled gradient
template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(PoolK p) {
  constexpr int CH = Chunk<T>::CH;
  const int ncc = p.C / CH;
  const long long total = (long long)p.N * p.D * p.H * p.W * ncc;
  const T* __restrict__ gp = (const T*)p.dy;
  T* __restrict__ op = (T*)p.out;
  const int He = p.H << p.uh, We = p.W << p.uw, De = p.D << p.ud;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total;
       q += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(q % ncc) * CH;
    long long t = q / ncc;
    const long long ipix = t;
    const int w = (int)(t % p.W); t /= p.W;
    const int h = (int)(t % p.H); t /= p.H;
    const int d = (int)(t % p.D);
    const int n = (int)(t / p.D);
    float s[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) s[j] = 0.f;
    for (int a = 0; a <= p.ud; ++a)
      for (int b = 0; b <= p.uh; ++b)
        for (int c = 0; c <= p.uw; ++c) {
          const long long e = (((long long)n * De + ((d << p.ud) + a)) * He + ((h << p.uh) + b)) * We + ((w << p.uw) + c);
          float f[CH];
          Chunk<T>::unpack(*(const u32x4*)(gp + e * p.lddy + c0), f);
#pragma unroll
          for (int j = 0; j < CH; ++j) s[j] += f[j];
        }
    T* dst = op + ipix * p.ldo + c0;
    if (p.accumulate) {
      float old[CH];
      Chunk<T>::unpack(*(const u32x4*)dst, old);
#pragma unroll
      for (int j = 0; j < CH; ++j) s[j] += old[j];
    }
    *(u32x4*)dst = Chunk<T>::pack(s);
  }
}

static int poolk_check(int dtype, const PoolK& k, const char* what) {
  if (dtype != HDU_BF16 && dtype != HDU_F32) return hdu_set_error(HDU_ERR_ARG, what);
  const int ch = dtype == HDU_BF16 ? 8 : 4;
  if (k.C <= 0 || k.C % ch || k.ldx % ch || k.ldo % ch || k.lddy % ch || k.N <= 0 || k.D <= 0 || k.H <= 0 || k.W <= 0)
    return hdu_set_error(HDU_ERR_ARG, what);
  return 0;
}

#define HDU_POOL_LAUNCH(kern, items)                                                                   \
  do {                                                                                                 \
    const int ch_ = dtype == HDU_BF16 ? 8 : 4;                                                         \
    const unsigned g_ = hdu_grid_1d((long long)(items) * (C / ch_), 256, 8192);                        \
    if (dtype == HDU_BF16) HDU_LAUNCH((kern<bf16_t>), dim3(g_), dim3(256), 0, (hipStream_t)stream, k); \
    else HDU_LAUNCH((kern<float>), dim3(g_), dim3(256), 0, (hipStream_t)stream, k);                    \
  } while (0)

extern "C" int hdu_maxpool3s2_fwd(int dtype, const void* x, int64_t ldx, int N, int D, int H, int W, int C, void* y,
                                  int64_t ldy, uint8_t* argmax, int pad_d, void* stream) {
  PoolK k{};
  k.idx = argmax;
  k.pad_d = pad_d;
  k.x = x; k.ldx = ldx; k.out = y; k.ldo = ldy; k.N = N; k.D = D; k.H = H; k.W = W; k.C = C;
  k.Do = D == 1 ? 1 : (D + 2 * pad_d - 3) / 2 + 1; k.Ho = (H - 1) / 2 + 1; k.Wo = (W - 1) / 2 + 1;
  if (!x || !y || (pad_d & ~1)) return hdu_set_error(HDU_ERR_ARG, "maxpool_fwd: null pointer / bad pad_d");
  if (int e = poolk_check(dtype, k, "maxpool_fwd: bad dims / strides")) return e;
  HDU_POOL_LAUNCH(maxpool_fwd_kernel, (long long)N * k.Do * k.Ho * k.Wo);
  return hdu_check_launch("maxpool_fwd");
}

extern "C" int hdu_maxpool3s2_bwd(int dtype, const uint8_t* argmax, const void* dy, int64_t lddy, int N, int D,
                                  int H, int W, int C, void* dx, int64_t lddx, int accumulate, int pad_d,
                                  void* stream) {
  PoolK k{};
  k.pad_d = pad_d;
  const void* x = argmax;
  const int64_t ldx = 16;
  k.idx = const_cast<uint8_t*>(argmax);
  k.ldx = ldx; k.dy = dy; k.lddy = lddy; k.out = dx; k.ldo = lddx; k.N = N; k.D = D; k.H = H; k.W = W; k.C = C;
  k.Do = D == 1 ? 1 : (D + 2 * pad_d - 3) / 2 + 1; k.Ho = (H - 1) / 2 + 1; k.Wo = (W - 1) / 2 + 1; k.accumulate = accumulate;
  if (!x || !dy || !dx) return hdu_set_error(HDU_ERR_ARG, "maxpool_bwd: null pointer (argmax / dy / dx)");
  if (int e = poolk_check(dtype, k, "maxpool_bwd: bad dims / strides")) return e;
  HDU_POOL_LAUNCH(maxpool_bwd_kernel, (long long)N * D * H * W);
  return hdu_check_launch("maxpool_bwd");
}

extern "C" int hdu_avgpool2_fwd(int dtype, const void* x, int64_t ldx, int N, int D, int H, int W, int C, void* y,
                                int64_t ldy, void* stream) {
  PoolK k{};
  k.x = x; k.ldx = ldx; k.out = y; k.ldo = ldy; k.N = N; k.D = D; k.H = H; k.W = W; k.C = C;
  k.Do = D; k.Ho = H / 2; k.Wo = W / 2;
  if (!x || !y) return hdu_set_error(HDU_ERR_ARG, "avgpool_fwd: null pointer");
  if (int e = poolk_check(dtype, k, "avgpool_fwd: bad dims / strides")) return e;
  HDU_POOL_LAUNCH(avgpool_fwd_kernel, (long long)N * D * k.Ho * k.Wo);
  return hdu_check_launch("avgpool_fwd");
}

extern "C" int hdu_avgpool2_bwd(int dtype, const void* dy, int64_t lddy, int N, int D, int H, int W, int C, void* dx,
                                int64_t lddx, int accumulate, void* stream) {
  PoolK k{};
  k.dy = dy; k.lddy = lddy; k.out = dx; k.ldo = lddx; k.N = N; k.D = D; k.H = H; k.W = W; k.C = C;
  k.Do = D; k.Ho = H / 2; k.Wo = W / 2; k.accumulate = accumulate;
  if (!dy || !dx) return hdu_set_error(HDU_ERR_ARG, "avgpool_bwd: null pointer");
  if (int e = poolk_check(dtype, k, "avgpool_bwd: bad dims / strides")) return e;
  HDU_POOL_LAUNCH(avgpool_bwd_kernel, (long long)N * D * H * W);
  return hdu_check_launch("avgpool_bwd");
}

extern "C" int hdu_upsample_bwd(int dtype, const void* dxe, int64_t lddxe, int N, int D, int H, int W, int C, int ud,
                                int uh, int uw, void* dz, int64_t lddz, int accumulate, void* stream) {
  PoolK k{};
  k.dy = dxe; k.lddy = lddxe; k.out = dz; k.ldo = lddz; k.N = N; k.D = D; k.H = H; k.W = W; k.C = C;
  k.ud = ud; k.uh = uh; k.uw = uw; k.accumulate = accumulate;
  if (!dxe || !dz || ((ud | uh | uw) & ~1)) return hdu_set_error(HDU_ERR_ARG, "upsample_bwd: bad args");
  if (int e = poolk_check(dtype, k, "upsample_bwd: bad dims / strides")) return e;
  HDU_POOL_LAUNCH(upsample_bwd_kernel, (long long)N * D * H * W);
  return hdu_check_launch("upsample_bwd");
}

// ====================================================================== weighted cross-entropy (loss.py:5-46)
template <typename T>
__global__ __launch_bounds__(256) void wce_kernel(const T* __restrict__ logits, long long ldl,
                                                  const uint8_t* __restrict__ labels, long long M, float w0, float w1,
                                                  float w2, float grad_scale, T* __restrict__ dlogits, long long lddl,
                                                  int Cpad, float* __restrict__ partial) {
  __shared__ float red[4][256];
  float lsum = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long long)gridDim.x * blockDim.x) {
    const float z0 = Chunk<T>::load1(logits + i * ldl), z1 = Chunk<T>::load1(logits + i * ldl + 1),
                z2 = Chunk<T>::load1(logits + i * ldl + 2);
    const int lab = labels[i];
    const float mx = fmaxf(z0, fmaxf(z1, z2));
    const float e0 = expf(z0 - mx), e1 = expf(z1 - mx), e2 = expf(z2 - mx);
    const float inv = 1.f / (e0 + e1 + e2);
    const float p0 = e0 * inv, p1 = e1 * inv, p2 = e2 * inv;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    if (lab <= 2) {
      const float pc = lab == 0 ? p0 : (lab == 1 ? p1 : p2);
      const float w = lab == 0 ? w0 : (lab == 1 ? w1 : w2);
      const float pcl = fminf(fmaxf(pc, 1e-10f), 1.0f);
      lsum += -w * logf(pcl);
      n0 += lab == 0; n1 += lab == 1; n2 += lab == 2;
      if (pc >= 1e-10f && pc <= 1.0f) {  // tf.clip_by_value passes the gradient only inside the range
        const float gs = grad_scale * w;
        d0 = gs * (p0 - (lab == 0 ? 1.f : 0.f));
        d1 = gs * (p1 - (lab == 1 ? 1.f : 0.f));
        d2 = gs * (p2 - (lab == 2 ? 1.f : 0.f));
      }
    }
    if (dlogits) {
      T* o = dlogits + i * lddl;
      Chunk<T>::store1(o, d0); Chunk<T>::store1(o + 1, d1); Chunk<T>::store1(o + 2, d2);
      for (int j = 3; j < Cpad; ++j) Chunk<T>::store1(o + j, 0.f);
    }
  }
  red[0][threadIdx.x] = lsum; red[1][threadIdx.x] = n0; red[2][threadIdx.x] = n1; red[3][threadIdx.x] = n2;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
#pragma unroll
      for (int q = 0; q < 4; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) partial[(long long)blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}

// 256 threads = 4 quantities (loss, 3 class counts) x 64 partial lanes, then an LDS tree (a single thread per quantity
// walking 2048 partials cost 170 us per step)
__global__ __launch_bounds__(256) void wce_finalize_kernel(const float* __restrict__ partial, int nblk, float* loss_sum,
                                                            float* class_count) {
  __shared__ double red[64][4];
  const int q = threadIdx.x & 3, pl = threadIdx.x >> 2;
  double a = 0.0;
  for (int b = pl; b < nblk; b += 64) a += (double)partial[(long long)b * 4 + q];
  red[pl][q] = a;
  __syncthreads();
  for (int s = 32; s > 0; s >>= 1) {
    if (pl < s) red[pl][q] += red[pl + s][q];
    __syncthreads();
  }
  if (pl == 0) {
    if (q == 0) *loss_sum += (float)red[0][0];
    else if (class_count) class_count[q - 1] += (float)red[0][q];
  }
}

extern "C" int hdu_wce_loss(int dtype, const void* logits, int64_t ldl, const uint8_t* labels, int64_t M, float w0,
                            float w1, float w2, float grad_scale, void* dlogits, int64_t lddl, int C_pad,
                            float* loss_sum, float* class_count, void* ws, size_t ws_bytes, void* stream) {
  if (!logits || !labels || !loss_sum || M <= 0 || ldl < 3) return hdu_set_error(HDU_ERR_ARG, "wce_loss: bad args");
  if (dlogits && (lddl < C_pad || C_pad < 3)) return hdu_set_error(HDU_ERR_ARG, "wce_loss: bad dlogits stride");
  const unsigned g = hdu_grid_1d(M, 256, 2048);
  if (!ws || ws_bytes < (size_t)g * 4 * sizeof(float)) return hdu_set_error(HDU_ERR_WORKSPACE, "wce_loss: workspace too small (32 KiB)");
  if (dtype == HDU_BF16)
    HDU_LAUNCH((wce_kernel<bf16_t>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logits, (long long)ldl,
               labels, (long long)M, w0, w1, w2, grad_scale, (bf16_t*)dlogits, (long long)lddl, C_pad, (float*)ws);
  else if (dtype == HDU_F32)
    HDU_LAUNCH((wce_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)logits, (long long)ldl,
               labels, (long long)M, w0, w1, w2, grad_scale, (float*)dlogits, (long long)lddl, C_pad, (float*)ws);
  else
    return hdu_set_error(HDU_ERR_ARG, "wce_loss: bad dtype");
  HDU_LAUNCH(wce_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ws, (int)g, loss_sum,
             class_count);
  return hdu_check_launch("wce_loss");
}

// ====================================================================== Nesterov SGD (K.optimizers.py:168-185)
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, float* __restrict__ v,
                                                  const float* __restrict__ g, long long n, float lr, float mom,
                                                  float gs) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gr = g[i] * gs;
    const float vn = mom * v[i] - lr * gr;
    v[i] = vn;
    p[i] = p[i] + mom * vn - lr * gr;
  }
}

extern "C" int hdu_sgd_nesterov(float* p, float* v, const float* g, int64_t n, float lr, float momentum,
                                float grad_scale, void* stream) {
  if (!p || !v || !g || n < 0) return hdu_set_error(HDU_ERR_ARG, "sgd: bad args");
  if (n == 0) return 0;
  HDU_LAUNCH(sgd_kernel, dim3(hdu_grid_1d(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, p, v, g, (long long)n, lr,
             momentum, grad_scale);
  return hdu_check_launch("sgd");
}

// ====================================================================== 2.5D <-> 3D plumbing and boundary casts
template <typename T>
__global__ __launch_bounds__(256) void slab25d_kernel(const float* __restrict__ vol, int D, int H, int W,
                                                      T* __restrict__ out, int Cpad) {
  const long long HW = (long long)H * W, total = (long long)D * HW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i / HW);
    const long long hw = i % HW;
    const int km = k > 0 ? k - 1 : 0, kp = k < D - 1 ? k + 1 : D - 1;
    T* o = out + i * Cpad;
    Chunk<T>::store1(o, vol[km * HW + hw]);
    Chunk<T>::store1(o + 1, vol[k * HW + hw]);
    Chunk<T>::store1(o + 2, vol[kp * HW + hw]);
    for (int j = 3; j < Cpad; ++j) Chunk<T>::store1(o + j, 0.f);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void make_input3d_kernel(const float* __restrict__ vol, const T* __restrict__ lg,
                                                           long long ldl, float scale, long long M, T* __restrict__ out,
                                                           int Cpad) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long long)gridDim.x * blockDim.x) {
    T* o = out + i * Cpad;
    Chunk<T>::store1(o, vol[i]);
    for (int j = 0; j < 3; ++j) Chunk<T>::store1(o + 1 + j, scale * Chunk<T>::load1(lg + i * ldl + j));
    for (int j = 4; j < Cpad; ++j) Chunk<T>::store1(o + j, 0.f);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void make_input3d_bwd_kernel(const T* __restrict__ din, int Cpad, float scale,
                                                               long long M, T* __restrict__ dlg, long long lddl,
                                                               int Cpad_l, int accumulate) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long long)gridDim.x * blockDim.x) {
    T* o = dlg + i * lddl;
    for (int j = 0; j < 3; ++j) {
      float v = scale * Chunk<T>::load1(din + i * Cpad + 1 + j);
      if (accumulate) v += Chunk<T>::load1(o + j);
      Chunk<T>::store1(o + j, v);
    }
    if (!accumulate)
      for (int j = 3; j < Cpad_l; ++j) Chunk<T>::store1(o + j, 0.f);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cast_pad_kernel(const float* __restrict__ src, long long M, int C,
                                                       T* __restrict__ dst, long long ldd, int Cpad) {
  const long long total = M * Cpad;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    const long long m = i / Cpad;
    Chunk<T>::store1(dst + m * ldd + c, c < C ? src[m * C + c] : 0.f);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cast_out_kernel(const T* __restrict__ src, long long lds, long long M, int C,
                                                       float* __restrict__ dst) {
  const long long total = M * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long m = i / C;
    dst[i] = Chunk<T>::load1(src + m * lds + c);
  }
}

// score[m][j] += softmax(logits[m][0..2])[j], j < num: the per-window step of the z-sliding-window inference
// (lib/funcs.py:31-34: K.softmax + K.eval + `score[...] += result`), one thread per voxel, logits read once
template <typename T>
__global__ __launch_bounds__(256) void softmax_accumulate_kernel(const T* __restrict__ logits, long long ldl, long long M, int num,
                                                                 float* __restrict__ score) {
  for (long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long long)gridDim.x * blockDim.x) {
    const T* r = logits + m * ldl;
    const float z0 = Chunk<T>::load1(r), z1 = Chunk<T>::load1(r + 1), z2 = Chunk<T>::load1(r + 2);
    const float mx = fmaxf(z0, fmaxf(z1, z2));
    const float e0 = expf(z0 - mx), e1 = expf(z1 - mx), e2 = expf(z2 - mx);
    const float inv = 1.0f / (e0 + e1 + e2);
    float* o = score + m * num;
    o[0] += e0 * inv;
    if (num > 1) o[1] += e1 * inv;
    if (num > 2) o[2] += e2 * inv;
  }
}

#define HDU_T_LAUNCH(T, kern, n, ...) \
  HDU_LAUNCH((kern<T>), dim3(hdu_grid_1d((n), 256, 4096)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__)
#define HDU_CHECK_DTYPE(what) \
  if (dtype != HDU_BF16 && dtype != HDU_F32) return hdu_set_error(HDU_ERR_ARG, what ": bad dtype")

extern "C" int hdu_slab25d(int dtype, const float* vol, int D, int H, int W, void* out, int Cpad, void* stream) {
  if (!vol || !out || D <= 0 || H <= 0 || W <= 0 || Cpad < 3) return hdu_set_error(HDU_ERR_ARG, "slab25d: bad args");
  if (dtype == HDU_BF16) { HDU_T_LAUNCH(bf16_t, slab25d_kernel, (long long)D * H * W, vol, D, H, W, (bf16_t*)out, Cpad); }
  else { HDU_T_LAUNCH(float, slab25d_kernel, (long long)D * H * W, vol, D, H, W, (float*)out, Cpad); }
  return hdu_check_launch("slab25d");
}

extern "C" int hdu_make_input3d(int dtype, const float* vol, const void* logits2d, int64_t ldl, float scale, int D,
                                int H, int W, void* out, int Cpad, void* stream) {
  if (!vol || !logits2d || !out || Cpad < 4 || ldl < 3) return hdu_set_error(HDU_ERR_ARG, "make_input3d: bad args");
  const long long M = (long long)D * H * W;
  if (dtype == HDU_BF16) { HDU_T_LAUNCH(bf16_t, make_input3d_kernel, M, vol, (const bf16_t*)logits2d, (long long)ldl, scale, M, (bf16_t*)out, Cpad); }
  else { HDU_T_LAUNCH(float, make_input3d_kernel, M, vol, (const float*)logits2d, (long long)ldl, scale, M, (float*)out, Cpad); }
  return hdu_check_launch("make_input3d");
}

extern "C" int hdu_make_input3d_bwd(int dtype, const void* dinput3d, int Cpad, float scale, int64_t M, void* dlogits2d,
                                    int64_t lddl, int Cpad_l, int accumulate, void* stream) {
  if (!dinput3d || !dlogits2d || Cpad < 4 || lddl < 3) return hdu_set_error(HDU_ERR_ARG, "make_input3d_bwd: bad args");
  if (dtype == HDU_BF16) { HDU_T_LAUNCH(bf16_t, make_input3d_bwd_kernel, M, (const bf16_t*)dinput3d, Cpad, scale, (long long)M, (bf16_t*)dlogits2d, (long long)lddl, Cpad_l, accumulate); }
  else { HDU_T_LAUNCH(float, make_input3d_bwd_kernel, M, (const float*)dinput3d, Cpad, scale, (long long)M, (float*)dlogits2d, (long long)lddl, Cpad_l, accumulate); }
  return hdu_check_launch("make_input3d_bwd");
}

extern "C" int hdu_cast_pad(int dtype, const float* src, int64_t M, int C, void* dst, int64_t lddst, int Cpad,
                            void* stream) {
  if (!src || !dst || C <= 0 || Cpad < C || lddst < Cpad) return hdu_set_error(HDU_ERR_ARG, "cast_pad: bad args");
  if (dtype == HDU_BF16) { HDU_T_LAUNCH(bf16_t, cast_pad_kernel, M * Cpad, src, (long long)M, C, (bf16_t*)dst, (long long)lddst, Cpad); }
  else { HDU_T_LAUNCH(float, cast_pad_kernel, M * Cpad, src, (long long)M, C, (float*)dst, (long long)lddst, Cpad); }
  return hdu_check_launch("cast_pad");
}

extern "C" int hdu_cast_out(int dtype, const void* src, int64_t ldsrc, int64_t M, int C, float* dst, void* stream) {
  if (!src || !dst || C <= 0 || ldsrc < C) return hdu_set_error(HDU_ERR_ARG, "cast_out: bad args");
  if (dtype == HDU_BF16) { HDU_T_LAUNCH(bf16_t, cast_out_kernel, M * C, (const bf16_t*)src, (long long)ldsrc, (long long)M, C, dst); }
  else { HDU_T_LAUNCH(float, cast_out_kernel, M * C, (const float*)src, (long long)ldsrc, (long long)M, C, dst); }
  return hdu_check_launch("cast_out");
}

extern "C" int hdu_softmax_accumulate(int dtype, const void* logits, int64_t ldl, int64_t M, int num, float* score, void* stream) {
  if (!logits || !score || ldl < 3 || num < 1 || num > 3 || M < 0) return hdu_set_error(HDU_ERR_ARG, "softmax_accumulate: bad args (3 classes, num in 1..3)");
  if (dtype != HDU_BF16 && dtype != HDU_F32) return hdu_set_error(HDU_ERR_ARG, "softmax_accumulate: bad dtype");
  if (M == 0) return 0;
  if (dtype == HDU_BF16) { HDU_T_LAUNCH(bf16_t, softmax_accumulate_kernel, M, (const bf16_t*)logits, (long long)ldl, (long long)M, num, score); }
  else { HDU_T_LAUNCH(float, softmax_accumulate_kernel, M, (const float*)logits, (long long)ldl, (long long)M, num, score); }
  return hdu_check_launch("softmax_accumulate");
}

// ------------------------------------------------------------------ per-step re-initialisation (include/hdu.h)
__global__ __launch_bounds__(256) void zero_regions_kernel(const hdu_zero_entry* __restrict__ table, int n, unsigned* counter,
                                                          unsigned counter_inc) {
  if (counter != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *counter += counter_inc;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].block_begin <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const hdu_zero_entry e = table[lo];
  const unsigned long long off = (unsigned long long)(blockIdx.x - e.block_begin) * HDU_ZERO_BLOCK_BYTES;
  if (off >= e.bytes) return;
  unsigned long long len = e.bytes - off;
  if (len > HDU_ZERO_BLOCK_BYTES) len = HDU_ZERO_BLOCK_BYTES;
  char* base = (char*)e.ptr + off;
  const unsigned long long n16 = len >> 4;
  const unsigned long long tail0 = n16 << 4;
  if (e.src != nullptr) {                              // a COPY region (the statistics shifts <- last step's means)
    const char* sb = (const char*)e.src + off;
    for (unsigned long long i = threadIdx.x; i < n16; i += 256) *(u32x4*)(base + (i << 4)) = *(const u32x4*)(sb + (i << 4));
    for (unsigned long long i = tail0 + 4ull * threadIdx.x; i + 4 <= len; i += 1024) *(unsigned*)(base + i) = *(const unsigned*)(sb + i);
    return;
  }
  const u32x4 z = u32x4{0u, 0u, 0u, 0u};
  for (unsigned long long i = threadIdx.x; i < n16; i += 256) *(u32x4*)(base + (i << 4)) = z;
  for (unsigned long long i = tail0 + 4ull * threadIdx.x; i + 4 <= len; i += 1024) *(unsigned*)(base + i) = 0u;
}

__global__ __launch_bounds__(256) void zero_one_kernel(char* __restrict__ p, unsigned long long bytes) {
  const u32x4 z = u32x4{0u, 0u, 0u, 0u};
  const unsigned long long n16 = bytes >> 4;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * 256)
    *(u32x4*)(p + (i << 4)) = z;
  if (blockIdx.x == 0)
    for (unsigned long long i = (n16 << 4) + 4ull * threadIdx.x; i + 4 <= bytes; i += 1024) *(unsigned*)(p + i) = 0u;
}

extern "C" int hdu_zero_regions(const hdu_zero_entry* dev_table, int n, uint32_t total_blocks, uint32_t* counter,
                                uint32_t counter_inc, void* stream) {
  if (!dev_table || n <= 0 || total_blocks == 0) return hdu_set_error(HDU_ERR_ARG, "zero_regions: bad args");
  HDU_LAUNCH(zero_regions_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, dev_table, n, counter, counter_inc);
  return hdu_check_launch("zero_regions");
}

extern "C" int hdu_zero(void* ptr, uint64_t bytes, void* stream) {
  if (!ptr || ((uintptr_t)ptr & 15) || (bytes & 3)) return hdu_set_error(HDU_ERR_ARG, "zero: ptr must be 16-byte aligned, bytes a multiple of 4");
  if (bytes == 0) return 0;
  HDU_LAUNCH(zero_one_kernel, dim3(hdu_grid_1d((long long)(bytes >> 4), 256 * 8, 2048)), dim3(256), 0, (hipStream_t)stream,
             (char*)ptr, (unsigned long long)bytes);
  return hdu_check_launch("zero");
}

// ------------------------------------------------------------------ float32 -> bf16 hi / lo planes (include/hdu.h: hdu_split3_*)
// The filter gradient of a float32 network in the split-bf16 modes: dW = sum_m dy[m] (x) x[m] with both operands contracted
// over PIXELS.  Splitting an operand on the fly costs VALU work per consuming wave (conv_wgrad_kernel<float>: 80 TF); splitting it
// ONCE into bf16 planes turns the layer into three bf16 products with the same contraction index -- and three products over M
// pixels are ONE product over 3 M pixels: the operand is written as the image triple (hi, lo, hi), the gradient as (hi, hi, lo),
// and the bf16 filter-gradient kernels (halo-tile / DMA families, ~1 PF) run unchanged on N' = 3 N images.
// Table-driven like zero_regions: one launch splits every operand of the backward pass.  Thread (cc, rl) owns an 8-channel
// column of its column group (the affine of a BN prologue stays in registers) and 8 rows of the block's row range.
template <int CHL>      // channels per lane: 8 (two 16-byte loads, 16-byte stores) or 4 (one 16-byte load, 8-byte stores)
__global__ __launch_bounds__(256) void split3_batched_kernel(const hdu_split3_entry* __restrict__ table, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].block_begin <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const hdu_split3_entry e = table[lo];
  const unsigned blk = blockIdx.x - e.block_begin;
  const unsigned cg = blk % e.col_groups;
  const unsigned long long rb = blk / e.col_groups;
  const unsigned cols = e.cols, rstep = 256u / cols;
  const unsigned cc = threadIdx.x % cols, rl = threadIdx.x / cols;
  const unsigned c0 = (cg * cols + cc) * CHL;
  if (c0 >= e.C) return;
  float a[CHL], b[CHL];
#pragma unroll
  for (int j = 0; j < CHL; ++j) { a[j] = e.a ? e.a[c0 + j] : 1.f; b[j] = e.a ? e.b[c0 + j] : 0.f; }
  const float* __restrict__ sp = e.src;
  bf16_t* __restrict__ dp = (bf16_t*)e.dst;
  const unsigned long long plane = e.rows * (unsigned long long)e.C;
  const unsigned long long p1 = plane, p2 = 2ull * plane;
  const bool grad = e.pattern != 0;      // operand: (hi, lo, hi); gradient: (hi, hi, lo)
  for (unsigned it = 0; it < e.iters; ++it) {
    const unsigned long long r_begin = (rb * e.iters + it) * (8ull * rstep);
    if (r_begin >= e.rows) return;
    u32x4 v0[8], v1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned long long m = r_begin + rl + (unsigned long long)u * rstep;
      if (m < e.rows) {
        const float* q = sp + m * e.ld_src + c0;
        v0[u] = *(const u32x4*)q;
        if constexpr (CHL == 8) v1[u] = *(const u32x4*)(q + 4);
      }
    }
    HDU_SCHED_BARRIER();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned long long m = r_begin + rl + (unsigned long long)u * rstep;
      if (m >= e.rows) continue;
      float f[8] = {hdu_u2f(v0[u].x), hdu_u2f(v0[u].y), hdu_u2f(v0[u].z), hdu_u2f(v0[u].w), 0.f, 0.f, 0.f, 0.f};
      if constexpr (CHL == 8) { f[4] = hdu_u2f(v1[u].x); f[5] = hdu_u2f(v1[u].y); f[6] = hdu_u2f(v1[u].z); f[7] = hdu_u2f(v1[u].w); }
      if (e.a) {
#pragma unroll
        for (int j = 0; j < CHL; ++j) {
          float s = a[j] * f[j] + b[j];
          if (e.relu) s = s > 0.f ? s : 0.f;
          f[j] = s;
        }
      }
      unsigned hp[CHL / 2], lp[CHL / 2];
#pragma unroll
      for (int j = 0; j < CHL / 2; ++j) {
        const unsigned hh = hdu_pack_bf16x2(f[2 * j], f[2 * j + 1]);
        hp[j] = hh;
        lp[j] = hdu_pack_bf16x2(f[2 * j] - hdu_u2f(hh << 16), f[2 * j + 1] - hdu_u2f(hh & 0xffff0000u));
      }
      bf16_t* o = dp + m * e.C + c0;
      if constexpr (CHL == 8) {
        const u32x4 h = u32x4{hp[0], hp[1], hp[2], hp[3]}, l = u32x4{lp[0], lp[1], lp[2], lp[3]};
        *(u32x4*)o = h;
        *(u32x4*)(o + p1) = grad ? h : l;
        *(u32x4*)(o + p2) = grad ? l : h;
      } else {
        const u32x2 h = u32x2{hp[0], hp[1]}, l = u32x2{lp[0], lp[1]};
        *(u32x2*)o = h;
        *(u32x2*)(o + p1) = grad ? h : l;
        *(u32x2*)(o + p2) = grad ? l : h;
      }
    }
  }
}

extern "C" int hdu_split3_entry_fill(hdu_split3_entry* e, const float* src, int64_t ld_src, int64_t rows, int C, const float* a,
                                     const float* b, int relu, int pattern, void* dst, uint32_t block_begin, uint32_t* nblocks) {
  if (!e || !src || !dst || !nblocks || rows <= 0 || C <= 0 || C % 8 || ld_src < C || ld_src % 4 || ((a == nullptr) != (b == nullptr)))
    return hdu_set_error(HDU_ERR_ARG, "split3_entry_fill: C must be a multiple of 8 (one 16-byte bf16 chunk), ld_src >= C a multiple of 4, a / b both or neither");
  if (((uintptr_t)src | (uintptr_t)dst | (uintptr_t)a | (uintptr_t)b) & 15)
    return hdu_set_error(HDU_ERR_ARG, "split3_entry_fill: pointers must be 16-byte aligned");
  if (pattern != HDU_SPLIT3_OPERAND && pattern != HDU_SPLIT3_GRADIENT) return hdu_set_error(HDU_ERR_ARG, "split3_entry_fill: bad pattern");
  const int form = g_tuning[HDU_TUNE_SPLIT3_FORM];      // low nibble: channels per lane (0 = default 8; 4), bits 4..: row groups of 8 per thread (0 = 1)
  const unsigned chl = (form & 15) == 4 ? 4u : 8u, iters = (form >> 4) > 0 ? (unsigned)(form >> 4) : 1u;
  const unsigned cpr = (unsigned)C / chl;
  unsigned best = 8, best_pad = ~0u;
  for (unsigned c = 256u / chl; c >= 64u / chl; c >>= 1) {          // fewest idle lanes, ties to the wider group
    const unsigned pad = (cpr + c - 1) / c * c;
    if (pad < best_pad) { best = c; best_pad = pad; }
  }
  e->src = src; e->dst = dst; e->a = a; e->b = b;
  e->ld_src = (uint64_t)ld_src; e->rows = (uint64_t)rows;
  e->C = (uint32_t)C; e->relu = relu ? 1u : 0u; e->pattern = (uint32_t)pattern;
  e->cols = best; e->col_groups = (cpr + best - 1) / best;
  e->block_begin = block_begin; e->iters = iters; e->chl = chl;
  const uint64_t rpb = 8ull * (256u / best) * iters;
  const uint64_t nb = ((uint64_t)rows + rpb - 1) / rpb * e->col_groups;
  if (nb + block_begin >= (1ull << 31)) return hdu_set_error(HDU_ERR_ARG, "split3_entry_fill: too many blocks for one launch");
  *nblocks = (uint32_t)nb;
  return 0;
}

extern "C" int hdu_split3_batched(const hdu_split3_entry* dev_table, int n, uint32_t total_blocks, int chl, void* stream) {
  if (!dev_table || n <= 0 || total_blocks == 0 || (chl != 4 && chl != 8)) return hdu_set_error(HDU_ERR_ARG, "split3_batched: bad args (chl = the entries' chl: 8 or 4)");
  if (chl == 4) HDU_LAUNCH(split3_batched_kernel<4>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, dev_table, n);
  else HDU_LAUNCH(split3_batched_kernel<8>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, dev_table, n);
  return hdu_check_launch("split3_batched");
}
