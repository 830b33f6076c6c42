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
};

// byte offset of 16-byte chunk `chunk` (0..7) of row `row` inside a [rows][128 B] LDS tile.
// XOR swizzle: the 16 lanes of a ds_read_b128 group (rows r..r+15, fixed logical chunk) hit 16
// distinct 16-B slots of the 256-B bank row.
__device__ __forceinline__ int lds_chunk_off(int row, int chunk) {
  return row * 128 + ((chunk ^ (row & 7)) << 4);
}
