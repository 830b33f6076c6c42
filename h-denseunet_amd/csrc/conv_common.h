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
  long long ldx, ldskip, ldy;
  long long M;                // N*Do*Ho*Wo
  int N, Di, Hi, Wi, Cin;     // stored input dims
  int De, He, We;             // effective (up-sampled) input dims
  int ud, uh, uw;
  int KD, KH, KW, sd, sh, sw, pd, ph, pw;
  int Do, Ho, Wo, Cout;
  int Ktot;                   // KD*KH*KW*Cin
  int pro_relu, accumulate;
  float drop_scale;           // 1/keep (0 => dropout disabled)
  unsigned drop_thresh;       // keep iff hash < thresh
  unsigned drop_seed;
  const unsigned* drop_seed_dev;
  int xcd_swizzle;
  int debug_flags;            // developer experiments only (HDU_TUNE_DEBUG): 1 = skip operand DMA, 2 = skip MFMA
  int vec_out;                // output rows are 16-byte addressable (DMA kernels' vector epilogue)
};

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
