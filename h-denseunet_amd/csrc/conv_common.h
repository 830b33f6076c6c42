// conv_common.h -- kernel-side parameter block shared by the conv kernels.
#pragma once
#include "hdu_platform.h"

struct ConvK {
  const void* x;
  const void* skip;
  const void* w;
  void* y;
  const float* pro_a;
  const float* pro_b;
  const float* bias;
  const float* epi_a;         // optional output affine (+ReLU) after bias / dropout (hdu_conv_desc.epi_*)
  const float* epi_b;
  int epi_relu;
  long long ldx, ldskip, ldy;
  long long M;                // N*Do*Ho*Wo
  int N, Di, Hi, Wi, Cin;     // stored input dims
  int De, He, We;             // effective (up-sampled) input dims
  int ud, uh, uw;
  int KD, KH, KW, sd, sh, sw, pd, ph, pw;
  int Do, Ho, Wo, Cout;
  int Ktot;                   // KD*KH*KW*Cin
  // output pixel index -> (n, od, oh, ow) without hardware division: q = umulhi(m, mul) >> shr for m < 2^31 (mul == 0:
  // divisor 1).  Filled on the host (fill_convk); the per-row state of a tile costs a few VALU ops instead of three
  // ~30-instruction division sequences per row (measured: 2-3 us of a 10-15 us short-K launch).
  unsigned div_wo_mul, div_wo_shr, div_ho_mul, div_ho_shr, div_do_mul, div_do_shr;
  unsigned spread_h, spread_d;   // sum_{q<KH} 2^(q*KW), sum_{q<KD} 2^(q*KH*KW) when KD*KH*KW <= 32 (FAST tap masks)
  unsigned div_cin_mul, div_cin_shr, div_kw_mul, div_kw_shr, div_kh_mul, div_kh_shr;   // k -> (tap, channel), tap -> (kd, kh, kw)
  unsigned x_bytes, w_bytes;   // extents of the input tensor (from x, FAST addressing only) and of the filter, for buffer loads
  unsigned sk_div_mul, sk_div_shr;     // split-K: division by the split count (set at launch)
  int pro_relu, accumulate;
  float drop_scale;           // 1/keep (0 => dropout disabled)
  unsigned drop_thresh;       // keep iff hash < thresh
  unsigned drop_seed;
  const unsigned* drop_seed_dev;
  int xcd_swizzle;
  int wg_gx, wg_gy, wg_gz;      // filter-gradient work grid (k-column tiles, filter-row tiles, pixel splits); 1-D launch
  int debug_flags;            // developer experiments only (HDU_TUNE_DEBUG): 1 = skip operand DMA, 2 = skip MFMA
  int vec_out;                // output rows are 16-byte addressable (DMA kernels' vector epilogue)
  int f32_split;              // float32 launches: 0 = exact f32 MFMA, 1 = bf16 hi/lo operand split, three bf16 MFMAs (Mma<float>)
  // optional per-channel statistics of the OUTPUT, accumulated by the epilogue (saves the separate reduction pass over
  // a tensor that is still in LDS): partial[slot][0][c] += sum(y - shift[c]), partial[slot][1][c] += sum((y - shift[c])^2)
  float* stats_partial;       // [stats_slots][2][Cout] float, zeroed by the caller; NULL = off
  const float* stats_shift;   // [Cout] shift against E[x^2]-E[x]^2 cancellation (any value near the mean)
  int stats_slots;
  // fused BN(+Scale)+ReLU backward of a data-gradient launch (hdu_conv_desc.bnb_*)
  const void* bnb_u; long long bnb_ldu;
  const float* bnb_a; const float* bnb_b; const float* bnb_mean; const float* bnb_rstd;
  float* bnb_partial; int bnb_slots; int bnb_relu;
  // split-K (ring kernel): gridDim.z workgroups share one output tile; partial accumulators meet in sk_ws
  float* sk_ws;               // [tile][split][wave][fragment][lane] f32x4, write-through stores
  unsigned* sk_cnt;           // [tile] arrival tickets, zero between launches
  long long M_layer;          // hdu_conv_desc.layer_rows (>= M): output pixels of the whole layer this launch is part of (host-side decisions only)
};

__device__ __forceinline__ unsigned hdu_fastdiv(unsigned n, unsigned mul, unsigned shr) {
  return mul == 0u ? n : (unsigned)(((unsigned long long)n * mul) >> 32) >> shr;
}

// first K-tile state of a lane: GEMM column k -> channel c inside tap (kd, kh, kw)
__device__ __forceinline__ void hdu_k_state(const ConvK& p, int k, int& c, int& kd, int& kh, int& kw, int& tap_i) {
  const unsigned tap = hdu_fastdiv((unsigned)k, p.div_cin_mul, p.div_cin_shr);
  tap_i = (int)tap;
  c = k - (int)tap * p.Cin;
  const unsigned t = hdu_fastdiv(tap, p.div_kw_mul, p.div_kw_shr);
  kw = (int)(tap - t * (unsigned)p.KW);
  const unsigned d = hdu_fastdiv(t, p.div_kh_mul, p.div_kh_shr);
  kh = (int)(t - d * (unsigned)p.KH);
  kd = (int)d;
}

// bits [lo, hi) of a 32-bit word, 0 <= lo < hi <= 32
__device__ __forceinline__ unsigned hdu_bit_range(int lo, int hi) {
  return (hi >= 32 ? 0xffffffffu : (1u << hi) - 1u) & ~((1u << lo) - 1u);
}

// taps q in [0, K) of one axis whose input coordinate i0 + q lies inside [0, E): a contiguous bit range (K <= 32)
__device__ __forceinline__ unsigned hdu_tap_range_mask(int i0, int K, int E) {
  const int lo = i0 < 0 ? -i0 : 0;
  const int hi = E - i0 < K ? E - i0 : K;
  return hi > lo ? (hi >= 32 ? 0xffffffffu : (1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
}

// per-row state of an implicit-GEMM tile row (output pixel m): the row's top-left-front input coordinate and, in the
// FAST form, its tap-validity bit mask + element offset of tap (0,0,0) channel 0.
template <bool FAST>
__device__ __forceinline__ void hdu_row_state(const ConvK& p, unsigned m, bool pointwise, int& rn, int& rid, int& rih, int& riw,
                                              int& rpix, unsigned& rmask) {
  rmask = 0u;
  if (FAST && pointwise) {                             // 1x1x1 stride 1 no padding: input pixel == output pixel
    rn = 0; rid = 0; rih = 0; riw = 0;
    rpix = (long long)m < p.M ? (int)m * (int)p.ldx : 0;
    rmask = (long long)m < p.M ? 1u : 0u;
  } else if ((long long)m < p.M) {
    unsigned t = hdu_fastdiv(m, p.div_wo_mul, p.div_wo_shr);
    const unsigned ow = m - t * (unsigned)p.Wo;
    unsigned t2 = hdu_fastdiv(t, p.div_ho_mul, p.div_ho_shr);
    const unsigned oh = t - t2 * (unsigned)p.Ho;
    const unsigned n = hdu_fastdiv(t2, p.div_do_mul, p.div_do_shr);
    const unsigned od = t2 - n * (unsigned)p.Do;
    rn = (int)n;
    rid = (int)od * p.sd - p.pd;
    rih = (int)oh * p.sh - p.ph;
    riw = (int)ow * p.sw - p.pw;
    rpix = ((rn * p.De + rid) * p.He + rih) * p.We + riw;
    if (FAST) {
      // tap-validity mask without loops: the valid taps of an axis are a contiguous range, so "one copy of the inner
      // mask per valid outer tap" is a multiplication by the outer range's bits of the spread constant
      // sum_q 2^(q * inner_taps) (from the host); the copies do not overlap and at most 32 taps exist: no carries
      const unsigned mw = hdu_tap_range_mask(riw, p.KW, p.We);
      const int lo_h = rih < 0 ? -rih : 0, hi_h = p.He - rih < p.KH ? p.He - rih : p.KH;
      const int lo_d = rid < 0 ? -rid : 0, hi_d = p.De - rid < p.KD ? p.De - rid : p.KD;
      const unsigned sh = hi_h > lo_h ? p.spread_h & hdu_bit_range(lo_h * p.KW, hi_h * p.KW) : 0u;
      const int thw = p.KH * p.KW;
      const unsigned sd = hi_d > lo_d ? p.spread_d & hdu_bit_range(lo_d * thw, hi_d * thw) : 0u;
      rmask = mw * sh * sd;
      rpix *= (int)p.ldx;                               // element offset of tap (0,0,0), channel 0
    }
  } else {
    rn = 0; rid = -(1 << 28); rih = -(1 << 28); riw = -(1 << 28); rpix = 0;
  }
}

// byte offset of 16-byte chunk `chunk` (0..7) of row `row` inside a [rows][128 B] LDS tile.
// XOR swizzle: the 16 lanes of a ds_read_b128 group (rows r..r+15, fixed logical chunk) hit 16
// distinct 16-B slots of the 256-B bank row.
__device__ __forceinline__ int lds_chunk_off(int row, int chunk) {
  return row * 128 + ((chunk ^ (row & 7)) << 4);
}

// XCD-aware tile order (MI355X: 8 XCDs, private 4 MiB L2 each; workgroup b is observed to run on XCD b % 8 -- used
// for speed only, never correctness).  Maps the hardware block index to a tile index such that each XCD walks a
// contiguous range of tiles: neighbouring output tiles share 3x3 halo rows and land in the same L2.  Bijective for
// any grid size.
__device__ __forceinline__ unsigned xcd_tile_index(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u;
  const unsigned xcd = bid & 7u, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Filter-gradient work grid -> this workgroup (1-D launch): k-column tiles x filter-row tiles x pixel splits, pixel split
// slowest.  (Rounds 1-2 measured an XCD-grouped order SLOWER -- with 8-step workgroups the launches were bound by their float
// atomics.  Round 4 sized the pixel splits from the whole launch (64-128 steps per workgroup), after which the bodies DO gain
// from the grouped order: they pass xcd_tile_index(bid) here.)
__device__ __forceinline__ bool wgrad_block(const ConvK& p, unsigned bid, unsigned* bx, unsigned* by, unsigned* bz) {
  const unsigned per = (unsigned)(p.wg_gx * p.wg_gy);
  *bz = bid / per;
  const unsigned w = bid % per;
  *by = w % (unsigned)p.wg_gy;
  *bx = w / (unsigned)p.wg_gy;
  return *bz < (unsigned)p.wg_gz;
}

// One layer of a batched filter-gradient launch (hdu_wgrad_plan_*): the launch covers the workgroups of MANY layers;
// a workgroup finds its layer by binary search over the first-workgroup table and then runs the ordinary kernel body.
struct WgradEntry {
  ConvK k;
  float* dw;
  long long per;        // pixel rows (DMA form) / spatial tiles (halo form) per workgroup split
};
