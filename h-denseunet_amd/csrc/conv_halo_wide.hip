// conv_halo_wide.hip -- halo-tile forward / data-gradient kernel for the WIDE 3x3 and 3x3x3 stride-1 "same" convolutions
// (bf16): the decoder layers conv_up0..4 (denseunet.py:189-218), 3dconv_up0..4 / fianl_conv (denseunet3d.py:158-184,430) and
// their data gradients -- the launches that the im2col implicit GEMM (conv_igemm.hip) runs at 0.19-0.31 of the MFMA roof
// because it re-loads every input pixel once per tap.
//
// Work decomposition.  A workgroup owns TH x 32 output pixels of ONE output plane (image n, depth od) and BN = 32 * NT output
// channels.  The contraction is walked in stages of (depth tap kd, 16 input channels): per stage ONE async DMA of the
// (TH + 2) x 34 halo tile of input plane od + kd - pd (32 B per pixel) and of the filter rows' 9 in-plane taps for these 16
// channels (288 B per output channel); the 9 taps are then formed from LDS -- the A fragment of tap (kh, kw) is the halo
// tile read at a shifted pixel window.  Per MAC the L2 -> LDS traffic is 3-6 x below the im2col tiling's (e.g. TH = 8,
// BN = 128: 47 KB per 4.7 M MACs against 32 KB per 1.0 M).  A depth tap whose plane lies outside the volume is skipped, not
// loaded as zeros.  The decoder's nearest-neighbour up-sampling in front of the conv is address arithmetic of the halo-tile
// DMA (stored pixel = effective pixel >> u per axis).
//
// MFMA: v_mfma_f32_32x32x16_bf16 -- its k extent IS the 16-channel stage, and a 32 x 32 fragment needs half the LDS operand
// reads per FLOP of the 16 x 16 x 32 form.  Operands are swapped (D rows = output channels, D columns = pixels) so that a lane
// holds 4 consecutive output channels of one pixel per accumulator quad (8-byte LDS stores in the epilogue).
//
// LDS: NS ring stages of [halo tile | filter tile], lane-linear 1-KiB DMA pieces; the two 16-byte chunks of a 32-byte row are
// swapped on odd 8-row groups (chunk ^ ((row >> 3) & 1)) so that the 16 lanes of a ds_read_b128 group hit 16 distinct bank
// slots for every tap shift.  One counted s_waitcnt + one raw barrier per stage (the ring of conv_igemm_ring_kernel).
#include "conv_common.h"
#include "hdu_host.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 hdu_mfma_32x32x16_bf16(u32x4 a, u32x4 b, f32x16 c) {
#ifdef HDU_EMU
  return hipemu_mfma_32x32x16_bf16(__builtin_bit_cast(hipemu_u16x8, a), __builtin_bit_cast(hipemu_u16x8, b), c);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hdu_bf16x8, a), __builtin_bit_cast(hdu_bf16x8, b), c, 0, 0, 0);
#endif
}

template <int N> __device__ __forceinline__ void hw_wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt immediate");
#ifndef HDU_EMU
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
#endif
}

template <int TH_, int NT_, int WAVES_M_, int WAVES_N_, int NS_> struct HaloWide {
  static constexpr int TH = TH_, NT = NT_, WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, NS = NS_;
  static constexpr int NW = WAVES_M * WAVES_N, NTHR = NW * 64;
  static constexpr int TW = 32, HC = TW + 2, HR = TH + 2, HP = HR * HC;   // halo tile: HR rows of HC pixels
  static constexpr int BN = NT * 32, BM = TH * TW;
  static constexpr int XI = (HP * 2 + 63) / 64;            // 1-KiB DMA pieces of the halo tile (2 chunks per pixel)
  static constexpr int WI = (BN * 18 + 63) / 64;           // ... of the filter tile (9 taps x 2 chunks per row)
  static constexpr int LPW = (XI + WI + NW - 1) / NW;      // pieces per wave and stage (the last round is padded: counted vmcnt)
  static constexpr int STAGE = LPW * NW * 1024;
  static constexpr int XBYTES = XI * 1024;
  static constexpr int ROWB = BN * 2 + 16;                 // staged output row (+16 B: consecutive rows start on different banks)
  static constexpr int SMEM = NS * STAGE > BM * ROWB ? NS * STAGE : BM * ROWB;
  static constexpr int MT = TH / WAVES_M, NTW = NT / WAVES_N;   // 32 x 32 fragments per wave: tile rows x channel groups
  static_assert(TH % WAVES_M == 0 && NT % WAVES_N == 0, "wave layout");
  static_assert(SMEM <= 160 * 1024, "LDS");
  static_assert((NS - 2) * LPW < 64, "vmcnt");
};

// per-channel moments of the staged output tile (ConvK::stats_partial): epilogue_stats of conv_igemm.hip for NTHR threads and
// halo-tile rows (tile pixel -> image pixel validity), rows requested 8 at a time
template <typename C, typename RowValid>
__device__ __forceinline__ void hw_epilogue_stats(const ConvK& p, const char* smem, int n0, int tid, unsigned slot, RowValid valid) {
  constexpr int NCC = C::BN / 8, CPI = C::NTHR / 16;
  float* dst = p.stats_partial + (long long)(slot % (unsigned)p.stats_slots) * 2 * p.Cout;
  const int rl = tid & 15;
#pragma unroll 1
  for (int it = 0; it < (NCC + CPI - 1) / CPI; ++it) {      // every lane takes every trip (the row sum is wave-wide)
    const int cc = (tid >> 4) + it * CPI;
    const int nbase = n0 + cc * 8;
    const bool col = cc < NCC && nbase < p.Cout;
    const int ccq = col ? cc : 0, nc = col ? nbase : 0;
    float s1[8], s2[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; j += 4) {
      const f32x4 v4 = *(const f32x4*)(p.stats_shift + nc + j);
      sh[j] = v4[0]; sh[j + 1] = v4[1]; sh[j + 2] = v4[2]; sh[j + 3] = v4[3];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll 1
    for (int q0 = 0; q0 < C::BM / 16; q0 += 8) {
      u32x4 rv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) rv[q] = *(const u32x4*)(smem + (rl + (q0 + q) * 16) * C::ROWB + ccq * 16);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float keep = (col && valid(rl + (q0 + q) * 16)) ? 1.f : 0.f;
        float f[8];
        Chunk<bf16_t>::unpack(rv[q], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = (f[j] - sh[j]) * keep; s1[j] += d; s2[j] += d * d; }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = hdu_row16_sum(s1[j]); s2[j] = hdu_row16_sum(s2[j]); }
    float mine = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      mine = rl == j ? s1[j] : mine;
      mine = rl == 8 + j ? s2[j] : mine;
    }
    if (col) atomicAdd(dst + (rl < 8 ? 0 : p.Cout) + nbase + (rl & 7), mine);
  }
}

// ---- epilogue shared by the kernels of this file: bias / dropout / output affine in registers, the tile through LDS, 16-byte
// row stores (accumulate mode: read-modify-write), conv-epilogue statistics.  Tile pixel `row` = (y0 + row / 32, x0 + row % 32) of
// output plane `mplane` (= (n * Do + od) * Ho); acc[i][j] = 32 x 32 fragment (tile row wm * MT + i, channel group wn * NTW + j).
template <typename C>
__device__ __forceinline__ void hw_epilogue(const ConvK& p, f32x16 (&acc)[C::MT][C::NTW], char* smem, long long mplane, int y0, int x0,
                                            int n0, int wm, int wn, int lane, int tid, unsigned slot) {
  typedef bf16_t T;
  constexpr int BN = C::BN, BM = C::BM, NTHR = C::NTHR, ROWB = C::ROWB, MT = C::MT, NTW = C::NTW;
  const int l31 = lane & 31, lh = lane >> 5;
  // ---- epilogue: bias / dropout / output affine in registers, the tile through LDS, 16-byte row stores
  hw_wait_vmcnt<0>();                                         // (the dead pieces of the last iterations too)
  __syncthreads();                                            // every wave is done with the operand stages, all DMAs have landed
  const unsigned dseed = p.drop_seed + (p.drop_seed_dev ? *p.drop_seed_dev : 0u);
  const bool has_bias = p.bias != nullptr, drop = p.drop_scale != 0.f, has_epi = p.epi_a != nullptr;
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    f32x4 bias_v[4], epa_v[4], epb_v[4];                      // unconditional loads at a clamped channel (see igemm_epilogue)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nn = n0 + (wn * NTW + j) * 32 + 8 * g + 4 * lh;
      const int nc = nn < p.Cout ? nn : 0;
      bias_v[g] = has_bias ? *(const f32x4*)(p.bias + nc) : f32x4{0.f, 0.f, 0.f, 0.f};
      epa_v[g] = has_epi ? *(const f32x4*)(p.epi_a + nc) : f32x4{1.f, 1.f, 1.f, 1.f};
      epb_v[g] = has_epi ? *(const f32x4*)(p.epi_b + nc) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int trow = wm * MT + i;
      const int row = trow * 32 + l31;
      const long long m = (mplane + y0 + trow) * p.Wo + x0 + l31;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = (wn * NTW + j) * 32 + 8 * g + 4 * lh;
        const int nn = n0 + col;
        float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        if (has_bias) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += bias_v[g][r];
        }
        if (drop) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const unsigned h = hdu_hash32((unsigned long long)m * (unsigned)p.Cout + (unsigned)(nn + r), dseed);
            v[r] = h < p.drop_thresh ? v[r] * p.drop_scale : 0.f;
          }
        }
        if (has_epi) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = epa_v[g][r] * v[r] + epb_v[g][r];
            if (p.epi_relu) v[r] = v[r] > 0.f ? v[r] : 0.f;
          }
        }
        Chunk<T>::store4((T*)(smem + row * ROWB) + col, v);
      }
    }
  }
  __syncthreads();
  T* __restrict__ yp = (T*)p.y;
  constexpr int NCC = BN / 8;
  constexpr int NIT = (BM * NCC + NTHR - 1) / NTHR;
  auto out_ptr = [&](int q, bool& ok) -> T* {
    const int row = q / NCC, cc = q - row * NCC;
    const int oy = y0 + (row >> 5), ox = x0 + (row & 31);
    const int nn = n0 + cc * 8;
    ok = q < BM * NCC && oy < p.Ho && ox < p.Wo && nn < p.Cout;
    return ok ? yp + ((mplane + oy) * p.Wo + ox) * p.ldy + nn : yp;
  };
  if (p.accumulate) {
#pragma unroll 1
    for (int it0 = 0; it0 < NIT; it0 += 4) {
      u32x4 old[4];
      T* dst[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        dst[u] = out_ptr(tid + (it0 + u) * NTHR, ok[u]);
        ok[u] = ok[u] && it0 + u < NIT;
        old[u] = *(const u32x4*)dst[u];                      // unconditional (a lane without a chunk re-reads element 0)
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!ok[u]) continue;
        const int q = tid + (it0 + u) * NTHR;
        const int row = q / NCC, cc = q - row * NCC;
        float f[8], g[8];
        Chunk<T>::unpack(*(const u32x4*)(smem + row * ROWB + cc * 16), f);
        Chunk<T>::unpack(old[u], g);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) f[jj] += g[jj];
        *(u32x4*)dst[u] = Chunk<T>::pack(f);
      }
    }
  } else {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = tid + it * NTHR;
      bool ok;
      T* dst = out_ptr(q, ok);
      if (!ok) continue;
      const int row = q / NCC, cc = q - row * NCC;
      *(u32x4*)dst = *(const u32x4*)(smem + row * ROWB + cc * 16);
    }
  }
  if (p.stats_partial)
    hw_epilogue_stats<C>(p, smem, n0, tid, slot, [&](int row) { return y0 + (row >> 5) < p.Ho && x0 + (row & 31) < p.Wo; });
}

template <typename C>
__global__ __launch_bounds__(C::NTHR) void conv_halo_wide_kernel(ConvK p, int tiles_x, int tiles_y, int ngroups) {
  typedef bf16_t T;
  constexpr int TH = C::TH, HC = C::HC, HP = C::HP, BN = C::BN, BM = C::BM, NW = C::NW, NTHR = C::NTHR, NS = C::NS;
  constexpr int XI = C::XI, WI = C::WI, LPW = C::LPW, STAGE = C::STAGE, XBYTES = C::XBYTES, ROWB = C::ROWB;
  constexpr int MT = C::MT, NTW = C::NTW;
  __shared__ __attribute__((aligned(16))) char smem[C::SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
#ifdef HDU_EMU
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;
  const T* __restrict__ xp = (const T*)p.x;
  const int He = p.He, We = p.We;

  // workgroup -> (channel group, output plane, tile): channel groups fastest (they share the halo tiles), then the planes of
  // one tile position (neighbouring planes share two of their three input planes); each XCD walks a contiguous range
  unsigned t = (p.xcd_swizzle & 1) ? xcd_tile_index(blockIdx.x, gridDim.x) : blockIdx.x;
  const int ng = (int)(t % (unsigned)ngroups); t /= (unsigned)ngroups;
  const int od = (int)(t % (unsigned)p.Do); t /= (unsigned)p.Do;
  const int txi = (int)(t % (unsigned)tiles_x); t /= (unsigned)tiles_x;
  const int tyi = (int)(t % (unsigned)tiles_y);
  const int n = (int)(t / (unsigned)tiles_y);
  const int y0 = tyi * TH, x0 = txi * C::TW;
  const int n0 = ng * BN;

  // ---- fixed DMA roles of this lane: piece jj = j * NW + wave; pieces [0, XI) are the halo tile, [XI, XI + WI) the filters
  unsigned roff[LPW];                // byte offset inside the input plane / the filter (stage-independent part); HDU_OOB = zeros
  int rlc[LPW];                      // first channel of the lane's 16-byte chunk inside the 16-channel stage (0 / 8)
#pragma unroll
  for (int j = 0; j < LPW; ++j) {
    const int jj = j * NW + wave;
    if (jj < XI) {
      const int q = jj * 64 + lane;
      const int hp = q >> 1;
      const int lc = (q & 1) ^ ((hp >> 3) & 1);
      const int hr = hp / HC, hc = hp - hr * HC;
      const int iy = y0 - 1 + hr, ix = x0 - 1 + hc;
      const bool ok = hp < HP && (unsigned)iy < (unsigned)He && (unsigned)ix < (unsigned)We;
      roff[j] = ok ? (unsigned)(((iy >> p.uh) * p.Wi + (ix >> p.uw)) * (int)p.ldx + lc * 8) * 2u : HDU_OOB;
      rlc[j] = lc * 8;
    } else if (jj < XI + WI) {
      const int q = (jj - XI) * 64 + lane;
      const int row = q / 18, rem = q - row * 18;
      const int tap = rem >> 1;
      const int lc = (rem & 1) ^ ((row >> 3) & 1);
      const int co = n0 + row;
      const bool ok = row < BN && co < p.Cout;
      roff[j] = ok ? (unsigned)((co * p.KD * 9 + tap) * p.Cin + lc * 8) * 2u : HDU_OOB;
      rlc[j] = lc * 8;
    } else {
      roff[j] = HDU_OOB;
      rlc[j] = 0;
    }
  }
  const hdu_bufsrd wsrd = hdu_make_srd(p.w, p.w_bytes);
  const long long plane_elems = (long long)p.Hi * p.Wi * p.ldx;
  const unsigned plane_bytes = (unsigned)((((long long)p.Hi * p.Wi - 1) * p.ldx + p.Cin) * 2);

  // one stage = LPW DMA pieces per wave.  The stage's wave-uniform state (plane resource, filter offset) is set up once; the
  // pieces themselves are issued BETWEEN the taps of the stage that is being multiplied (issue_piece), not in a burst after the
  // barrier: both waves of a SIMD leave the barrier together, and a burst of LPW x ~100-180 issue cycles at that point leaves
  // the MFMA pipe idle (first build, rocprof + ISA: 52 % of the MFMA roof with the burst)
  // wave-uniform part of a stage's sources: input-plane pointer (+ first channel), bytes addressable behind it, filter offset
  auto stage_plane = [&](int kd, int c0) -> const T* {
    const int pz = od + kd - p.pd;                            // effective input plane (inside the volume: see kd_lo / kd_hi)
    return xp + (long long)(n * p.Di + (pz >> p.ud)) * plane_elems + c0;
  };
  auto issue_piece = [&](const hdu_bufsrd& xsrd, unsigned woff, int c0, char* base, int j, bool live) {
    const int jj = j * NW + wave;
    const bool ok = live && roff[j] != HDU_OOB && c0 + rlc[j] < p.Cin;   // (a ragged last stage: channels >= Cin read as zeros on both sides)
    if (jj < XI) hdu_bufload_lds16(xsrd, ok ? roff[j] : HDU_OOB, base + jj * 1024);
    else hdu_bufload_lds16(wsrd, ok ? roff[j] + woff : HDU_OOB, base + jj * 1024);
  };

  f32x16 acc[MT][NTW];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // stages: depth taps whose plane lies inside the volume x 16-channel chunks
  const int kd_lo = p.pd - od > 0 ? p.pd - od : 0;
  const int kd_hi = p.De + p.pd - od < p.KD ? p.De + p.pd - od : p.KD;
  const int nch = (p.Cin + 15) >> 4;
  const int nst = kd_hi > kd_lo ? (kd_hi - kd_lo) * nch : 0;
  int ikd = kd_lo, ic = 0;                                    // the next stage to issue
#pragma unroll
  for (int pre = 0; pre < NS - 1; ++pre) {
    const bool live = pre < nst;
    const int c0 = live ? ic * 16 : 0;
    const hdu_bufsrd xsrd = hdu_make_srd(stage_plane(live ? ikd : kd_lo, c0), plane_bytes - (unsigned)c0 * 2u);
    const unsigned woff = (unsigned)(ikd * 9 * p.Cin + c0) * 2u;
#pragma unroll
    for (int j = 0; j < LPW; ++j) issue_piece(xsrd, woff, c0, smem + pre * STAGE, j, live);
    if (live && ++ic == nch) { ic = 0; ++ikd; }
  }

  // per-lane fragment addressing
  const int l31 = lane & 31, lh = lane >> 5;
  int wrow_off[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int row = (wn * NTW + j) * 32 + l31;
    wrow_off[j] = XBYTES + row * 288 + ((lh ^ ((row >> 3) & 1)) << 4);
  }
  const int hp_base = wm * MT * HC + l31;

  int slot = 0;
  for (int s = 0; s < nst; ++s) {
    // every iteration issues LPW pieces per wave (the last NS - 1 iterations issue dead ones: out-of-range lanes, zeros into the
    // slot nobody reads again), so ONE counted wait serves every iteration and the taps below carry no branch
    hw_wait_vmcnt<(NS - 2) * LPW>();
    HDU_RAW_BARRIER();
    // the stage issued during this iteration (into the slot the previous iteration multiplied from): its DMA pieces go out
    // BETWEEN the taps below, not in a burst after the barrier -- both waves of a SIMD leave the barrier together, and a burst
    // of LPW x ~100-180 issue cycles there leaves the MFMA pipe idle (first build: 0.52 of the MFMA roof with the burst)
    const bool more = s + NS - 1 < nst;
    const int nc0 = more ? ic * 16 : 0;
    const hdu_bufsrd nxsrd = hdu_make_srd(stage_plane(more ? ikd : kd_lo, nc0), plane_bytes - (unsigned)nc0 * 2u);
    const unsigned nwoff = (unsigned)(ikd * 9 * p.Cin + nc0) * 2u;
    char* nbase = smem + (slot == 0 ? NS - 1 : slot - 1) * STAGE;
    if (more && ++ic == nch) { ic = 0; ++ikd; }
    const char* Xs = smem + slot * STAGE;
    constexpr int PER_TAP = (LPW + 8) / 9;                    // pieces issued behind each tap's MFMAs
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap - kh * 3;
      u32x4 af[MT], bf[NTW];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int hp = hp_base + (i + kh) * HC + kw;
        af[i] = *(const u32x4*)(Xs + hp * 32 + ((lh ^ ((hp >> 3) & 1)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NTW; ++j) bf[j] = *(const u32x4*)(Xs + wrow_off[j] + tap * 32);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = hdu_mfma_32x32x16_bf16(bf[j], af[i], acc[i][j]);
#pragma unroll
      for (int u = 0; u < PER_TAP; ++u)
        if (tap * PER_TAP + u < LPW) issue_piece(nxsrd, nwoff, nc0, nbase, tap * PER_TAP + u, more);
    }
    slot = slot == NS - 1 ? 0 : slot + 1;
  }

  const long long mplane = ((long long)n * p.Do + od) * p.Ho;
  hw_epilogue<C>(p, acc, smem, mplane, y0, x0, n0, wm, wn, lane, tid, blockIdx.x);
}

// =====================================================================================
// Stem: 7 x 7 (x 7) stride-2 convolution over an 8-channel (16 B per pixel) input, Cout <= 96 per group -- `conv1` / `3dconv1`
// (denseunet.py:163-164, denseunet3d.py:129-130).  The im2col GEMM gathers ONE 16-byte chunk per (row, tap): 343 taps x 8 padded
// channels, 0.26 of the MFMA roof at the shard shape (2.2 of 41.8 ms).  Here the contraction is ordered (kd, kh | kw, c): for one
// (kd, kh) the seven kw taps of an output pixel are 7 ADJACENT input pixels = 112 contiguous bytes, so a 16-element MFMA k-step is
// two neighbouring input pixels and the A fragment of output pixel ow is a 16-byte read at input pixel 2 * ow - 3 + kw (lane
// stride 32 B) of a row segment staged once per stage.  kw is padded to 8 (the 8th tap's filter chunk is read out of range =
// zeros).  Workgroup = 16 x 32 output pixels of one output plane x 96 channels, 8 waves (2 tile rows x 3 channel groups each);
// stage = (kd, kh): 16 row segments of 70 input pixels + the filter rows' 8 chunks; ring of 4 stages, counted waits as above.
// LDS: input pixel slot q = r * 70 + j stored at q ^ ((q >> 4) & 1) (the lanes of a ds_read_b128 group are 2 slots apart: pairs
// 16 / 48 slots apart would share a bank slot); filter rows at a 9-chunk (144 B) stride (9 * co mod 16 is a permutation).
struct StemCfg {
  static constexpr int TH = 16, NT = 3, WAVES_M = 8, WAVES_N = 1, NS = 4;
  static constexpr int NW = 8, NTHR = 512, TW = 32;
  static constexpr int BN = 96, BM = TH * TW;
  static constexpr int SEG = 70;                            // input pixels per row segment: 2 * 31 + 8 taps
  static constexpr int XSLOTS = TH * SEG;                   // 1120
  static constexpr int XI = (XSLOTS + 63) / 64;             // 18 pieces
  static constexpr int WSLOTS = BN * 9;                     // 864
  static constexpr int WI = (WSLOTS + 63) / 64;             // 14 pieces
  static constexpr int LPW = (XI + WI + NW - 1) / NW;       // 4
  static constexpr int STAGE = LPW * NW * 1024;
  static constexpr int XBYTES = XI * 1024;
  static constexpr int ROWB = BN * 2 + 16;
  static constexpr int SMEM = NS * STAGE > BM * ROWB ? NS * STAGE : BM * ROWB;
  static constexpr int MT = 2, NTW = 3;
  static_assert(SMEM <= 160 * 1024, "LDS");
};

__global__ __launch_bounds__(StemCfg::NTHR) void conv_stem_s2_kernel(ConvK p, int tiles_x, int tiles_y, int ngroups) {
  typedef StemCfg C;
  typedef bf16_t T;
  constexpr int NW = C::NW, NS = C::NS, XI = C::XI, WI = C::WI, LPW = C::LPW, STAGE = C::STAGE, XBYTES = C::XBYTES, SEG = C::SEG;
  constexpr int MT = C::MT, NTW = C::NTW, BN = C::BN;
  __shared__ __attribute__((aligned(16))) char smem[C::SMEM];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
#ifdef HDU_EMU
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const T* __restrict__ xp = (const T*)p.x;
  unsigned t = (p.xcd_swizzle & 1) ? xcd_tile_index(blockIdx.x, gridDim.x) : blockIdx.x;
  const int ng = (int)(t % (unsigned)ngroups); t /= (unsigned)ngroups;
  const int od = (int)(t % (unsigned)p.Do); t /= (unsigned)p.Do;
  const int txi = (int)(t % (unsigned)tiles_x); t /= (unsigned)tiles_x;
  const int tyi = (int)(t % (unsigned)tiles_y);
  const int n = (int)(t / (unsigned)tiles_y);
  const int y0 = tyi * C::TH, x0 = txi * C::TW;
  const int n0 = ng * BN;

  // ---- fixed DMA roles: pieces [0, XI) = input row segments (stage part: + kh rows), [XI, XI + WI) = filter chunks
  unsigned roff[LPW];
  int riy[LPW];                      // input row of the lane's pixel at kh = 0 (x pieces); rows outside the image read as zeros
#pragma unroll
  for (int j = 0; j < LPW; ++j) {
    const int jj = j * NW + wave;
    riy[j] = -(1 << 28);
    if (jj < XI) {
      const int ps = jj * 64 + lane;
      const int q = ps ^ ((ps >> 4) & 1);
      const int r = q / SEG, c = q - r * SEG;
      const int iy = 2 * (y0 + r) - p.ph, ix = 2 * x0 - p.pw + c;
      const bool ok = q < C::XSLOTS && (unsigned)ix < (unsigned)p.Wi;
      roff[j] = ok ? (unsigned)((iy * p.Wi + ix) * (int)p.ldx) * 2u : HDU_OOB;      // (iy may be negative: checked per stage with kh)
      riy[j] = ok ? iy : -(1 << 28);
    } else if (jj < XI + WI) {
      const int ps = (jj - XI) * 64 + lane;
      const int row = ps / 9, kw = ps - row * 9;
      const int co = n0 + row;
      const bool ok = row < BN && co < p.Cout && kw < 7;
      roff[j] = ok ? (unsigned)((co * p.KD * 49 + kw) * 8) * 2u : HDU_OOB;
    } else {
      roff[j] = HDU_OOB;
    }
  }
  const hdu_bufsrd wsrd = hdu_make_srd(p.w, p.w_bytes);
  const long long plane_elems = (long long)p.Hi * p.Wi * p.ldx;
  const unsigned plane_bytes = (unsigned)((((long long)p.Hi * p.Wi - 1) * p.ldx + 8) * 2);
  const int row_bytes = p.Wi * (int)p.ldx * 2;

  auto stage_plane = [&](int kd) -> const T* {
    const int pz = od * p.sd + kd - p.pd;
    return xp + (long long)(n * p.Di + pz) * plane_elems;
  };
  auto issue_piece = [&](const hdu_bufsrd& xsrd, int kd, int kh, char* base, int j, bool live) {
    const int jj = j * NW + wave;
    if (jj < XI) {
      const bool ok = live && (unsigned)(riy[j] + kh) < (unsigned)p.Hi;
      hdu_bufload_lds16(xsrd, ok ? roff[j] + (unsigned)(kh * row_bytes) : HDU_OOB, base + jj * 1024);
    } else {
      const bool ok = live && roff[j] != HDU_OOB;
      hdu_bufload_lds16(wsrd, ok ? roff[j] + (unsigned)((kd * 7 + kh) * 7 * 8 * 2) : HDU_OOB, base + jj * 1024);
    }
  };

  f32x16 acc[MT][NTW];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // stages: (kd with its plane inside the volume) x kh
  const int pz0 = od * p.sd - p.pd;
  const int kd_lo = pz0 < 0 ? -pz0 : 0;
  const int kd_hi = p.Di - pz0 < p.KD ? p.Di - pz0 : p.KD;
  const int nst = kd_hi > kd_lo ? (kd_hi - kd_lo) * 7 : 0;
  int ikd = kd_lo, ikh = 0;
#pragma unroll
  for (int pre = 0; pre < NS - 1; ++pre) {
    const bool live = pre < nst;
    const hdu_bufsrd xsrd = hdu_make_srd(stage_plane(live ? ikd : kd_lo), plane_bytes);
#pragma unroll
    for (int j = 0; j < LPW; ++j) issue_piece(xsrd, ikd, ikh, smem + pre * STAGE, j, live);
    if (live && ++ikh == 7) { ikh = 0; ++ikd; }
  }

  const int l31 = lane & 31, lh = lane >> 5;
  int wrow_off[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) wrow_off[j] = XBYTES + ((j * 32 + l31) * 9 + lh) * 16;
  int xq[MT];                        // slot of (tile row, input pixel 2 * l31 + lh) at k-step 0
#pragma unroll
  for (int i = 0; i < MT; ++i) xq[i] = (wave * MT + i) * SEG + 2 * l31 + lh;

  int slot = 0;
  for (int s = 0; s < nst; ++s) {
    hw_wait_vmcnt<(NS - 2) * LPW>();
    HDU_RAW_BARRIER();
    const bool more = s + NS - 1 < nst;
    const hdu_bufsrd nxsrd = hdu_make_srd(stage_plane(more ? ikd : kd_lo), plane_bytes);
    const int nkd = ikd, nkh = ikh;
    char* nbase = smem + (slot == 0 ? NS - 1 : slot - 1) * STAGE;
    if (more && ++ikh == 7) { ikh = 0; ++ikd; }
    const char* Xs = smem + slot * STAGE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {                          // k-step = input pixels 2 ks, 2 ks + 1 of the 8-pixel window
      u32x4 af[MT], bf[NTW];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int q = xq[i] + 2 * ks;
        af[i] = *(const u32x4*)(Xs + ((q ^ ((q >> 4) & 1)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NTW; ++j) bf[j] = *(const u32x4*)(Xs + wrow_off[j] + ks * 32);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = hdu_mfma_32x32x16_bf16(bf[j], af[i], acc[i][j]);
      if (ks < LPW) issue_piece(nxsrd, nkd, nkh, nbase, ks, more);
    }
    slot = slot == NS - 1 ? 0 : slot + 1;
  }
  const long long mplane = ((long long)n * p.Do + od) * p.Ho;
  hw_epilogue<C>(p, acc, smem, mplane, y0, x0, n0, wave, 0, lane, tid, blockIdx.x);
}

// =====================================================================================
// Stem filter gradient: dW[co][kd][kh][kw][c] of the 7 x 7 (x 7) stride-2 stem over 8 stored channels (the im2col
// filter-gradient kernel runs it at 0.08 of the MFMA roof -- 2.9 of 41.8 ms at the shard shape -- because a k-chunk of its GEMM is ONE
// 16-byte pixel).  Here a workgroup owns one depth tap kd and a range of 4 x 32-pixel output tiles; per tile ONE DMA of the dy tile
// ([128 px][96 co]) and of the 13 input row segments (70 pixels of 16 B) that the 7 kh taps of its 4 output rows touch.  Wave w < 7
// owns kh = w: D[co][n = kw * 8 + c] += dy^T[co][px] . X[px][n] with px as the MFMA k axis -- for a fixed (kd, kh) the (kw, c)
// columns of output pixel ow are the 112 contiguous bytes at input pixel 2 * ow - 3 (kw padded to 8: columns 56..63 are discarded).
// Both operands are read with the transposing LDS load (8 consecutive pixels of one column per lane).  The 7 x 6 accumulator
// fragments live in registers for the whole range; one float atomic per element at the end.
template <int NS_> struct StemWT {
  static constexpr int TH = 4, TW = 32, PX = TH * TW, SEG = 70, NR = 2 * TH + 5, NW = 8, NTHR = 512, NS = NS_;
  static constexpr int DSLOTS = PX * 12, DI = DSLOTS / 64;            // dy tile: 12 chunks (96 channels) per pixel, 24 pieces
  static constexpr int XSLOTS = NR * SEG, XI = (XSLOTS + 63) / 64;    // 910 slots, 15 pieces
  static constexpr int LPW = (DI + XI + NW - 1) / NW;                 // 5
  static constexpr int STAGE = LPW * NW * 1024, DBYTES = DI * 1024;
  static_assert(NS * STAGE <= 160 * 1024, "LDS");
};

typedef StemWT<2> StemW;
template <int NS_>
__global__ __launch_bounds__(StemW::NTHR) void conv_stem_wgrad_kernel(ConvK p, float* __restrict__ dw, int tiles_x, int tiles_y,
                                                                      int nsplit, int ngroups) {
  typedef StemWT<NS_> C;
  typedef bf16_t T;
  constexpr int NW = C::NW, NS = C::NS, DI = C::DI, XI = C::XI, LPW = C::LPW, STAGE = C::STAGE, DBYTES = C::DBYTES, SEG = C::SEG;
  __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
#ifdef HDU_EMU
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  // workgroup -> (depth tap kd, split of the tile list, channel group); the KD workgroups of a split are neighbours in the launch
  // order (they read the same dy tiles) and the XCD-major order keeps them on one L2
  unsigned t = xcd_tile_index(blockIdx.x, gridDim.x);
  const int kd = (int)(t % (unsigned)p.KD); t /= (unsigned)p.KD;
  const int split = (int)(t % (unsigned)nsplit);
  const int ng = (int)(t / (unsigned)nsplit);
  const int n0 = ng * 96;
  // output planes whose input plane od * sd + kd - pd lies inside the volume
  const int num = p.pd - kd;
  int od_lo = num > 0 ? (num + p.sd - 1) / p.sd : 0;
  int od_hi = (p.Di - 1 + p.pd - kd) >= 0 ? (p.Di - 1 + p.pd - kd) / p.sd + 1 : 0;
  if (od_hi > p.Do) od_hi = p.Do;
  const int nod = od_hi > od_lo ? od_hi - od_lo : 0;
  const int ntiles = p.N * nod * tiles_y * tiles_x;
  const int per = (ntiles + nsplit - 1) / nsplit;
  const int t0 = split * per;
  const int t1 = t0 + per < ntiles ? t0 + per : ntiles;
  const int nt = t1 > t0 ? t1 - t0 : 0;

  // ---- fixed DMA roles: pieces [0, DI) = dy tile, [DI, DI + XI) = input row segments
  unsigned loff[LPW];
  int la[LPW], lb[LPW];              // dy: tile row / column of the lane's pixel (la < 0: no chunk); x: segment row / pixel
#pragma unroll
  for (int j = 0; j < LPW; ++j) {
    const int jj = j * NW + wave;
    la[j] = -1; lb[j] = 0; loff[j] = 0u;
    if (jj < DI) {
      const int ps = jj * 64 + lane;
      const int px = ps / 12, ch = ps - px * 12;
      const int r = px >> 5, c = px & 31;
      if (n0 + ch * 8 < p.Cout) {
        la[j] = r; lb[j] = c;
        loff[j] = (unsigned)((r * p.Wo + c) * (int)p.ldy + n0 + ch * 8) * 2u;
      }
    } else if (jj < DI + XI) {
      const int ps = (jj - DI) * 64 + lane;
      const int rr = ps / SEG, cc = ps - rr * SEG;
      if (ps < C::XSLOTS) {
        la[j] = rr; lb[j] = cc;
        loff[j] = (unsigned)((rr * p.Wi + cc) * (int)p.ldx) * 2u;
      }
    }
  }
  const long long dplane_elems = (long long)p.Ho * p.Wo * p.ldy, xplane_elems = (long long)p.Hi * p.Wi * p.ldx;
  const unsigned dplane_bytes = (unsigned)((dplane_elems - p.ldy + p.Cout) * 2), xplane_bytes = (unsigned)((xplane_elems - p.ldx + 8) * 2);

  // tile cursor (issue side), advanced incrementally: (tx, ty, odr, n)
  int i_tx, i_ty, i_od, i_n;
  {
    int q = t0;
    i_tx = q % tiles_x; q /= tiles_x;
    i_ty = q % tiles_y; q /= tiles_y;
    i_od = nod > 0 ? q % nod : 0;
    i_n = nod > 0 ? q / nod : 0;
  }
  auto issue_tile = [&](int slot, bool live) {
    char* base = smem + slot * STAGE;
    const int y0 = i_ty * C::TH, x0 = i_tx * C::TW;
    const int od = od_lo + i_od;
    const int n = live ? i_n : 0;
    const hdu_bufsrd dsrd = hdu_make_srd((const T*)p.y + (long long)(n * p.Do + (live ? od : 0)) * dplane_elems, dplane_bytes);
    const hdu_bufsrd xsrd = hdu_make_srd((const T*)p.x + (long long)(n * p.Di + (live ? od * p.sd + kd - p.pd : 0)) * xplane_elems, xplane_bytes);
    const unsigned dbase = (unsigned)((y0 * p.Wo + x0) * (int)p.ldy) * 2u;
    const int iy0 = 2 * y0 - p.ph, ix0 = 2 * x0 - p.pw;
    const unsigned xbase = (unsigned)((iy0 * p.Wi + ix0) * (int)p.ldx) * 2u;      // (modular: the sum with a valid lane offset is in range)
#pragma unroll
    for (int j = 0; j < LPW; ++j) {
      const int jj = j * NW + wave;
      if (jj < DI) {
        const bool ok = live && la[j] >= 0 && y0 + la[j] < p.Ho && x0 + lb[j] < p.Wo;
        hdu_bufload_lds16(dsrd, ok ? dbase + loff[j] : HDU_OOB, base + jj * 1024);
      } else {
        const bool ok = live && la[j] >= 0 && (unsigned)(iy0 + la[j]) < (unsigned)p.Hi && (unsigned)(ix0 + lb[j]) < (unsigned)p.Wi;
        hdu_bufload_lds16(xsrd, ok ? xbase + loff[j] : HDU_OOB, base + jj * 1024);
      }
    }
    if (live) {
      if (++i_tx == tiles_x) {
        i_tx = 0;
        if (++i_ty == tiles_y) {
          i_ty = 0;
          if (++i_od == nod) { i_od = 0; ++i_n; }
        }
      }
    }
  };

  f32x16 acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int pre = 0; pre < NS - 1; ++pre) issue_tile(pre, pre < nt);

  // per-lane fragment addressing (tile-invariant): transposing reads, 16-lane groups = 4 pixel rows x 16 columns
  const int li = lane & 15, g16 = (lane >> 4) & 1, lh = lane >> 5;
  const int kh = wave;                                   // wave 7 only moves data
  const int kpix = 8 * lh + (li >> 2);                   // pixel (k index) of the first half; + 4 for the second
  int doff[3], xoff[2];
#pragma unroll
  for (int i = 0; i < 3; ++i) doff[i] = kpix * 192 + (i * 32 + g16 * 16 + (li & 3) * 4) * 2;
#pragma unroll
  for (int j = 0; j < 2; ++j) xoff[j] = DBYTES + (kh * SEG + 2 * kpix) * 16 + (j * 32 + g16 * 16 + (li & 3) * 4) * 2;

  int slot = 0;
  for (int s = 0; s < nt; ++s) {
    hw_wait_vmcnt<(NS - 2) * LPW>();
    HDU_RAW_BARRIER();
    issue_tile(slot == 0 ? NS - 1 : slot - 1, s + NS - 1 < nt);
    if (kh < 7) {
      const char* St = smem + slot * STAGE;
#pragma unroll
      for (int r = 0; r < C::TH; ++r)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          u32x4 af[3], bf[2];
          const int dbase = (r * 32 + ks * 16) * 192;                   // dy rows of these 16 pixels
          const int xb = (2 * r * SEG + 32 * ks) * 16;                  // input row 2 r (+ kh), input pixel 2 * (16 ks)
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const u32x2 lo = hdu_lds_tr16_b64(St + dbase + doff[i]);
            const u32x2 hi = hdu_lds_tr16_b64(St + dbase + doff[i] + 4 * 192);
            af[i] = u32x4{lo.x, lo.y, hi.x, hi.y};
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const u32x2 lo = hdu_lds_tr16_b64(St + xb + xoff[j]);
            const u32x2 hi = hdu_lds_tr16_b64(St + xb + xoff[j] + 8 * 16);
            bf[j] = u32x4{lo.x, lo.y, hi.x, hi.y};
          }
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = hdu_mfma_32x32x16_bf16(af[i], bf[j], acc[i][j]);
        }
    }
    slot = slot == NS - 1 ? 0 : slot + 1;
  }
  hw_wait_vmcnt<0>();
  if (kh >= 7 || nt == 0) return;
  const int l31 = lane & 31;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nn = j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = n0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (nn < 56 && co < p.Cout) atomicAdd(dw + ((long long)(co * p.KD + kd) * 7 + kh) * 56 + nn, acc[i][j][r]);
      }
    }
}

// ------------------------------------------------------------------ host side
// configurations: (tile rows, 32-channel groups, waves along the tile rows, waves along the channel groups, ring stages)
typedef HaloWide<8, 4, 4, 2, 3> HW_8x128;        // 512 threads, 144 KB: one workgroup per CU
typedef HaloWide<16, 2, 8, 1, 3> HW_16x64;       // 512 threads, 120 KB
typedef HaloWide<16, 3, 8, 1, 3> HW_16x96;       // 512 threads, 144 KB
typedef HaloWide<8, 2, 4, 1, 2> HW_8x64;         // 256 threads, 64 KB: two workgroups per CU
typedef HaloWide<8, 3, 4, 1, 2> HW_8x96;         // 256 threads, 80 KB: two workgroups per CU
typedef HaloWide<16, 4, 8, 1, 2> HW_16x128;      // 512 threads, 112 KB, two stages
typedef HaloWide<16, 2, 4, 1, 2> HW_16x64p;      // 256 threads, 80 KB, two stages: two workgroups per CU, 4 x 2 fragments per wave (large grids)

struct HwChoice { int cfg; };                    // 1..5 in the order above, 0 = not taken

template <typename C>
static void hw_launch(const ConvK& k, hipStream_t s) {
  const int tiles_x = (k.We + C::TW - 1) / C::TW, tiles_y = (k.He + C::TH - 1) / C::TH;
  const int ngroups = (k.Cout + C::BN - 1) / C::BN;
  const unsigned grid = (unsigned)((long long)k.N * k.Do * tiles_x * tiles_y * ngroups);
  HDU_LAUNCH((conv_halo_wide_kernel<C>), dim3(grid), dim3(C::NTHR), 0, s, k, tiles_x, tiles_y, ngroups);
}

// geometry the kernel covers
static bool hw_shape_ok(const ConvK& k, int dtype) {
  if (dtype != HDU_BF16 || k.bnb_u != nullptr || k.pro_a != nullptr || k.skip != nullptr || !k.vec_out) return false;
  if (k.KH != 3 || k.KW != 3 || (k.KD != 1 && k.KD != 3) || k.sd != 1 || k.sh != 1 || k.sw != 1 || k.ph != 1 || k.pw != 1) return false;
  if (k.Ho != k.He || k.Wo != k.We || k.Do != k.De + 2 * k.pd - k.KD + 1 || k.Do < 1) return false;
  if (k.KD == 1 && k.pd != 0) return false;
  if (k.Cin % 8 || k.Cout % 8) return false;
  if ((long long)k.Hi * k.Wi * k.ldx * 2 >= (1ll << 32)) return false;        // one input plane within a 32-bit byte offset
  if ((long long)k.N * k.Do * ((k.He + 7) / 8) * ((k.We + 31) / 32) * ((k.Cout + 31) / 32) >= (1ll << 31)) return false;
  return true;
}

// which configuration (0 = leave the launch to the im2col kernels).  Decided for the WHOLE layer (hdu_conv_desc.layer_rows),
// so that a depth shard sums every output element in the same order as the unsharded launch.
// Cost model, calibrated on the per-layer A/B of profiles/r05_experiment_halo_wide_*.txt (unit: one 32 x 32 x 16 MFMA per SIMD):
// a compute unit is handed load = ceil(workgroups / 256) workgroups; each needs stages x TH x NT x 9 / 4 MFMA slots of the pipe
// the co-resident workgroups share (measured 0.58-0.74 busy in the K loop), and a fixed prologue + epilogue per workgroup that a
// co-resident partner mostly hides (two-workgroup configurations).  Padded output channels and tile rows are paid in full.
static constexpr int NCFG = 7;
static int hw_choose(const ConvK& k, int dtype) {
  const int force = g_tuning[HDU_TUNE_HALO_WIDE];
  if (force == 1 || !hw_shape_ok(k, dtype)) return 0;
  if (force >= 2) return force - 1 <= NCFG ? force - 1 : 0;
  if (k.Cin < 32 || k.Cout < 48 || k.We < 24) return 0;
  const double scale = (double)k.M_layer / (double)k.M;                        // planes of the whole layer per plane of this launch
  constexpr int NC = 7;
  const int th[NC] = {8, 16, 16, 8, 8, 16, 16}, nt[NC] = {4, 2, 3, 2, 3, 4, 2}, occ[NC] = {1, 1, 1, 2, 2, 1, 2};
  const double busy[NC] = {0.60, 0.68, 0.68, 0.58, 0.58, 0.74, 0.76};
  const double nst = (double)((k.Cin + 15) / 16) * k.KD;
  // deep contractions (>= 24 stages) on grids that a 16-row configuration can spread over half the chip: the 8-row two-per-CU forms
  // re-stream the filter tile twice as often per MAC and lose (per-layer A/B: conv_up0 data gradient 741 vs 1014 TF) whatever the model says
  bool deep16 = false;
  if (nst >= 24. && !(g_tuning[HDU_TUNE_DEBUG] & 1024))                          // (bit 10: A/B without this rule)
    for (int c = 0; c < NC; ++c)
      if (th[c] == 16 && (double)k.N * k.Do * ((k.He + 15) / 16) * ((k.We + 31) / 32) * scale * (double)((k.Cout + nt[c] * 32 - 1) / (nt[c] * 32)) >= 128.)
        deep16 = true;
  int best = 0;
  double best_cost = 0., best_wgs = 0.;
  for (int c = 0; c < NC; ++c) {
    if (deep16 && th[c] == 8) continue;
    if (c == 6 && (g_tuning[HDU_TUNE_DEBUG] & 512)) continue;                  // A/B: without the two-per-CU 16 x 64 form
    const int bn = nt[c] * 32;
    const double wgs = (double)k.N * k.Do * ((k.He + th[c] - 1) / th[c]) * ((k.We + 31) / 32) * scale * (double)((k.Cout + bn - 1) / bn);
    const double load = __builtin_ceil(wgs / 256.0);
    const double work = (double)th[c] * nt[c];
    // (16x64p: 4 x 2 fragments per wave, ONE wave per SIMD and workgroup -- it needs a co-resident partner: measured 0.50 alone)
    const double main_ = nst * work / ((c == 6 && load < 4.) ? 0.50 : busy[c]);
    const double fixed = 40.0 + work;
    const double cost = load * main_ + __builtin_ceil(load / occ[c]) * fixed * ((occ[c] == 2 && load >= 2.) ? 0.3 : 1.0);
    if (best == 0 || cost < best_cost) { best = c + 1; best_cost = cost; best_wgs = wgs; }
  }
  // too small to fill the chip with halo tiles: the im2col ring kernels with split-K spread such layers over more compute units
  if (best_wgs < 128.) return 0;
  // narrow outputs (the 48-channel dense-block layers) pay 25 % padding here: only where the layer is large
  if (k.Cout < 64 && best_wgs < 1024.) return 0;
  return best;
}

// the stem kernels' geometry: 7 x 7 (x 7), stride 2 in the plane (and in depth for 7 x 7 x 7), 8 stored channels, plain launch
static bool stem_geom_ok(const ConvK& k, int dtype) {
  if (g_tuning[HDU_TUNE_HALO_WIDE] == 1 || dtype != HDU_BF16) return false;
  if (k.bnb_u != nullptr || k.pro_a != nullptr || k.skip != nullptr || !k.vec_out || (k.ud | k.uh | k.uw) != 0) return false;
  if (k.KH != 7 || k.KW != 7 || k.sh != 2 || k.sw != 2 || k.ph != 3 || k.pw != 3 || k.Cin != 8) return false;
  if (!((k.KD == 1 && k.sd == 1 && k.pd == 0) || (k.KD == 7 && k.sd == 2 && k.pd >= 0 && k.pd <= 3))) return false;
  if (k.Ho != (k.Hi + 6 - 7) / 2 + 1 || k.Wo != (k.Wi + 6 - 7) / 2 + 1 || k.Do != (k.Di + 2 * k.pd - k.KD) / k.sd + 1) return false;
  if (k.Cout % 8) return false;
  return (long long)k.Hi * k.Wi * k.ldx * 2 < (1ll << 31);
}

static bool stem_ok(const ConvK& k, int dtype) {
  if (!stem_geom_ok(k, dtype)) return false;
  const long long wgs = (long long)k.N * k.Do * ((k.Ho + 15) / 16) * ((k.Wo + 31) / 32) * ((k.Cout + 95) / 96);
  if (wgs >= (1ll << 31)) return false;
  if (g_tuning[HDU_TUNE_HALO_WIDE] >= 2) return true;                          // tests: every geometry the kernel covers
  return k.Wo >= 24 && wgs * k.M_layer / k.M >= 128;
}

static void stem_launch(const ConvK& k, hipStream_t s) {
  const int tiles_x = (k.Wo + 31) / 32, tiles_y = (k.Ho + 15) / 16, ngroups = (k.Cout + 95) / 96;
  const unsigned grid = (unsigned)((long long)k.N * k.Do * tiles_x * tiles_y * ngroups);
  HDU_LAUNCH(conv_stem_s2_kernel, dim3(grid), dim3(StemCfg::NTHR), 0, s, k, tiles_x, tiles_y, ngroups);
}

// filter gradient of the same layers (dy is `y` of the descriptor): same geometry test, any size (its im2col form is the slowest
// kernel family of the library), planes within 32-bit byte offsets
bool hdu_stem_wgrad_taken(const ConvK& k, int dtype) {
  return stem_geom_ok(k, dtype) && k.ldy % 8 == 0 && (long long)k.Ho * k.Wo * k.ldy * 2 < (1ll << 31) &&
         (long long)k.Cout * k.KD * 49 * 8 < (1ll << 31);
}

bool hdu_stem_wgrad_launch(const ConvK& k, int dtype, float* dw, hipStream_t s) {
  if (!hdu_stem_wgrad_taken(k, dtype)) return false;
  const int tiles_x = (k.Wo + 31) / 32, tiles_y = (k.Ho + 3) / 4, ngroups = (k.Cout + 95) / 96;
  const long long tiles = (long long)k.N * k.Do * tiles_x * tiles_y;
  // two stages (80 KB): two workgroups per CU, ~512 over the launch, at least 32 tiles each -- a workgroup ends with 96 x 56 x 7 float
  // atomics, which dominate the small stems (measured: 8 -> 32 tiles per workgroup and 3 -> 2 stages: 2D 134 -> 104 us, 224 x 224 x 12
  // 123 -> 99 us, shard shape 1274 -> 1270 us; profiles/r05_experiment_stem_kernels_ab.txt)
  long long nsplit = 512 / ((long long)k.KD * ngroups);
  if (nsplit > tiles / 32) nsplit = tiles / 32;
  if (nsplit < 1) nsplit = 1;
  const unsigned grid = (unsigned)(nsplit * k.KD * ngroups);
  HDU_LAUNCH((conv_stem_wgrad_kernel<2>), dim3(grid), dim3(StemW::NTHR), 0, s, k, dw, tiles_x, tiles_y, (int)nsplit, ngroups);
  return true;
}

bool hdu_halo_wide_taken(const ConvK& k, int dtype) { return stem_ok(k, dtype) || hw_choose(k, dtype) != 0; }

const char* hdu_halo_wide_name(const ConvK& k, int dtype) {
  if (stem_ok(k, dtype)) return "conv_stem_s2_kernel";
  static const char* names[8] = {"", "conv_halo_wide_kernel<8x128>", "conv_halo_wide_kernel<16x64>", "conv_halo_wide_kernel<16x96>",
                                 "conv_halo_wide_kernel<8x64>", "conv_halo_wide_kernel<8x96>", "conv_halo_wide_kernel<16x128>",
                                 "conv_halo_wide_kernel<16x64p>"};
  return names[hw_choose(k, dtype)];
}

bool hdu_halo_wide_launch(const ConvK& k, int dtype, hipStream_t s) {
  if (stem_ok(k, dtype)) { stem_launch(k, s); return true; }
  switch (hw_choose(k, dtype)) {
    case 1: hw_launch<HW_8x128>(k, s); return true;
    case 2: hw_launch<HW_16x64>(k, s); return true;
    case 3: hw_launch<HW_16x96>(k, s); return true;
    case 4: hw_launch<HW_8x64>(k, s); return true;
    case 5: hw_launch<HW_8x96>(k, s); return true;
    case 6: hw_launch<HW_16x128>(k, s); return true;
    case 7: hw_launch<HW_16x64p>(k, s); return true;
    default: return false;
  }
}
