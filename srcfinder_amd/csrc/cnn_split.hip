// CNN tile scorer: the float32 convolutions on the float16 matrix cores by OPERAND SPLITTING (round 5; VERDICT r4 "missing" 4).
//
// a = a_hi + a_lo with a_hi = fp16(a), a_lo = fp16(a - a_hi): 22 bits of a's 24-bit mantissa travel in two halves, and
//     a * w  =  a_hi w_hi + a_hi w_lo + a_lo w_hi   (+ a_lo w_lo ~ 2^-22 |a w|, dropped)
// is three v_mfma_f32_32x32x16_f16 with float32 accumulation -- each at 16 x the rate of the fp32 instruction.  The products
// carry a relative error of ~2^-22 (fp32: exact products, 2^-24 accumulate), so this is the same tolerance class as the fp32
// path (the reference goldens at 1e-4), NOT the fp16-storage option of cnn_f16.hip (operands rounded to 11 bits).
//
// float16's exponent range is the catch: a_lo ~ 2^-12 |a| is subnormal (absolute error 2^-25) once |a| < 0.25.  The weights
// are therefore scaled per output channel by a power of two that puts the channel's largest weight at ~2^13 (sf_cnn_split_weights;
// the epilogue multiplies by 2^-e: exact); the activations take a power of two PER LAYER (ascale; cnn_driver.hip's calibration
// puts the layer's largest activation of a sample of windows at 2^9..2^10: an activation's low half is subnormal -- an ABSOLUTE
// error of 2^-25 -- only below 2^-13 of that maximum, and 64 x headroom is left above it).
// An activation with |a ascale| >= 65504 has no float16: the launch stores 1 into the CALLER's device int (`overflow`, one per
// call / batch / stream -- the library keeps no flag of its own) and the caller repeats that batch on the fp32 kernels
// (sf_cnn_score_rows and srcfinder_amd.cnn do) -- the result is never silently wrong.
//
// Implicit GEMM as in cnn_f16.hip: 128 pixels x BN channels per workgroup, 32 input channels of one tap per chunk, operand
// tiles [row][k] with 80-byte rows -- here four of them (A hi / lo, B hi / lo); the activations are split on their way
// from global memory into LDS.
#include "cmf_common.h"
#include "cnn_ring.h"
#include "cnn_internal.h"
#include <type_traits>

typedef _Float16 sp_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 sp_h2 __attribute__((ext_vector_type(2)));
typedef float sp_f16 __attribute__((ext_vector_type(16)));
typedef float sp_f4 __attribute__((ext_vector_type(4)));
typedef unsigned sp_u4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SP_LD = 40;   // halves per LDS row (32 + 8 pad = 80 bytes)

struct ConvDstS {
  float *p[3];
  int ld[3], off[3], end[3];
  int fmt[3];      // 0: float32 [pixel][channel]; 1: the split format below (dense tensors: ld == the segment's channels, off == 0);
                   // 2: float32, the 2 x 2 / 2 max pool of the 16 x 16 grid taken in the epilogue -- [window][8][8][channel] (maxpool4)
  float oscale[3]; // split-format segments: the power of two their CONSUMER expects (its ascale), applied before the split
};
// The SPLIT FORMAT of an activation tensor [M][C], C a multiple of 8: per pixel and 8-channel group 16 halves -- the eight high halves,
// then the eight low halves -- i.e. [M][C / 8][2][8] float16 in the bytes of the [M][C] float32 tensor, the group of channel c at the
// byte offset of float32 channel 8 (c / 8).  A convolution whose only consumers are split-operand convolutions (the 3 x 3 reducers'
// outputs, conv2) writes it from its epilogue, and the consumer's tile fetch -- the same addresses as for float32 -- lands the two
// halves ready for LDS: the split, a third of a chunk's time in the kernel below, is done once per value instead of once per
// (value, tap, channel tile).

// The RING form of the convolution (round 6, trunk sharing; cnn_ring.h has the geometry): the rows of the implicit GEMM are the ring
// positions of the OUTPUT tensor's frame, per window (conv3: 496 of 4096), and a tap of such a row is fetched from one of two places
// -- the window's own ring tensor of the INPUT (the positions that see the window's zero padding) or the SHARED phase map of the
// input over the whole plane (every other position: identical for all windows of a phase) -- or is zero (outside the window's grid).
struct RingArgs {
  SfGather in;         // where the input lives
  int olo, ohi, nout;  // the rows: ring positions of the frame (olo, ohi) on the same grid, nout of them per window
  int nin;             // ring positions of the input frame
  int side, nfull;     // side = 1 (band sharing, cnn_ring.h): the rows are the frame's SIDE positions (nout = G (olo + ohi)) and row
                       // (window, j) is written to row window * nfull + its ring index of the output tensors (nfull = the whole ring)
};

// hi / lo halves of eight floats (scaled by s); big: the largest magnitude seen (float16 ends at 65504)
__device__ __forceinline__ void sp_split8(const sp_f4 &x0, const sp_f4 &x1, float s, sp_h8 &hi, sp_h8 &lo, float &big) {
  const float v[8] = {x0.x * s, x0.y * s, x0.z * s, x0.w * s, x1.x * s, x1.y * s, x1.z * s, x1.w * s};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const _Float16 h = (_Float16)v[k];
    hi[k] = h;
    lo[k] = (_Float16)(v[k] - (float)h);
    big = fmaxf(big, fabsf(v[k]));      // (a NaN activation stays NaN through the products: nothing to flag)
  }
}

// (A form with two LDS sets and two register sets -- loads two chunks ahead, one barrier per chunk -- needed 306 registers, ran one
//  workgroup per CU and was 1.7 x SLOWER: what hides the trips to L2 here is three workgroups per CU, not a deeper pipeline.)
template <int BN, bool ASPLIT = false, int BMT = 128, bool RING = false>      // ASPLIT: the input is in the split format; BMT: pixels per tile
__global__ __launch_bounds__(256, (BN == 64 && BMT == 128) ? 4 : (((BN == 128 && BMT == 128) || (BN == 64 && BMT == 256)) ? 3 : 2)) void k_conv_split(const float *__restrict__ in, int M, int H, int W, int Cin, int ld_in,
                                                    const _Float16 *__restrict__ whi, const _Float16 *__restrict__ wlo,
                                                    const float *__restrict__ wscale, const float *__restrict__ bias, int Cout,
                                                    int ks, float ascale, ConvDstS dst, int *__restrict__ overflow, RingArgs ring) {
  constexpr int BM = BMT, BK = 32, NPA = BM / 64;
  constexpr int WN = (BN >= 128 && BN != 160) ? 2 : 1;   // waves along N: 2 x 2 waves of 64 x BN/2, or 4 x 1 waves of 32 x 64 (32 x 160 at BN = 160)
  constexpr int WM = 4 / WN;
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
  constexpr int NPB = (BN + 63) / 64;            // weight-tile passes of 64 rows (the last one partly used at BN = 96)
  static_assert(BN == 64 || BN == 96 || BN == 128 || BN == 160 || BN == 192 || BN == 256, "channel tile");
  constexpr int NSET = 1;
  constexpr int SETH = (2 * BM + 2 * BN) * SP_LD;     // halves per set: A hi | A lo | B hi | B lo
  __shared__ __attribute__((aligned(16))) _Float16 sm[NSET * SETH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int q = tid & 3, ri = tid >> 2;
  // XCD-aware order: workgroup id -> (XCD = id & 7, slot = id >> 3); an XCD walks ITS contiguous eighth of the pixel tiles with the
  // channel tiles of a pixel tile in consecutive slots, so the activations a pixel tile fetches -- once per channel tile and tap,
  // and the rows above / below are its neighbours' -- come from that XCD's L2 after the first touch (the kernel is bound by
  // exactly this traffic: 128 -> 192 channels at 32 x 32 re-reads its 268 MB input 27 times)
  const int MT = (M + BM - 1) / BM, NT = (Cout + BN - 1) / BN, per = (MT + 7) / 8;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int mt = xcd * per + slot / NT, nt_ = slot - (slot / NT) * NT;
  if (mt >= MT || slot / NT >= per) return;
  const int m0 = mt * BM, n0 = nt_ * BN;
  const int pad = ks >> 1, taps = ks * ks, nchunk = (Cin + BK - 1) / BK, nit = taps * nchunk;

  constexpr unsigned OOB = 0x80000000u;
  unsigned rowoff[NPA], vmask[NPA], woff[NPB];
  const size_t shift = RING ? 0 : ((size_t)pad * W + pad) * ld_in;          // taps are addressed from (y - pad, x - pad): offsets >= 0
  // (RING: `in` = the phase maps with the batch's border tensor behind them: M / 496 windows x 252 positions)
  const size_t ring_floats = RING ? (size_t)ring.in.ring_off + (size_t)(M / (RING ? ring.nout : 1) + 1) * ring.nin * ld_in : 0;
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in) - shift, 0,
                                                                 RING ? (unsigned)(ring_floats * 4)
                                                                      : (unsigned)(((size_t)M * ld_in + 2 * shift) * 4 + 64), 0x00020000);
  unsigned rpos[NPA], rring[NPA];      // RING: (y << 8 | x) of the row's position, byte offset of its window's border tensor (+ 32 q)
  __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(whi), 0, (unsigned)((size_t)Cout * taps * Cin * 2), 0x00020000);
  __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(wlo), 0, (unsigned)((size_t)Cout * taps * Cin * 2), 0x00020000);
#pragma unroll
  for (int a = 0; a < NPA; ++a) {
    const int m = m0 + ri + 64 * a;
    const bool ok = m < M;
    const int mm = ok ? m : 0;
    if constexpr (RING) {
      const SfGather &gi = ring.in;
      const int n = mm / ring.nout, j = mm - n * ring.nout;
      int y3, x3;
      if (ring.side) sf_side_position(gi.G, ring.olo, ring.ohi, j, y3, x3);
      else sf_frame_position(gi.G, ring.olo, ring.ohi, j, y3, x3);
      const long long t = gi.tile0 + n;
      const int r = (int)(t / gi.W), c = (int)(t - (long long)r * gi.W);
      const int pm = (1 << gi.shift) - 1;
      const int ph = ((r & pm) << gi.shift) + (c & pm);
      rowoff[a] = (unsigned)((((size_t)(ph * gi.Hq + ((r >> gi.shift) - gi.Rb)) * gi.Wq + (c >> gi.shift)) * ld_in + 8 * q) * 4);   // the window's origin in its phase map
      rring[a] = (unsigned)(((size_t)gi.ring_off + (size_t)n * ring.nin * ld_in + 8 * q) * 4);
      rpos[a] = ok ? (unsigned)((y3 << 8) | x3) : 0xffffu;           // (a row past M: every tap outside)
      vmask[a] = 0;
    } else {
    const int py = ok ? (mm / W) % H : -100000, px = mm % W;
    rowoff[a] = (unsigned)(((size_t)mm * ld_in + 8 * q) * 4);
    unsigned vm = 0;
    for (int tp = 0; tp < taps; ++tp) {
      const int yy = py + tp / ks - pad, xx = px + tp % ks - pad;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) vm |= 1u << tp;
    }
    vmask[a] = vm;
    }
  }
#pragma unroll
  for (int b = 0; b < NPB; ++b) {
    const int co = n0 + ri + 64 * b;
    woff[b] = (co < Cout && ri + 64 * b < BN) ? (unsigned)(((size_t)co * taps * Cin + 8 * q) * 2) : OOB;
  }
  sp_f16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float big = 0.f;
  sp_f4 ra[NSET][NPA][2];
  sp_u4 rbh[NSET][NPB], rbl[NSET][NPB];
  int g_tap = 0, g_ty = 0, g_tx = 0, g_c0 = 0;
  auto as_f4 = [](sp_u4 v) { union { sp_u4 u; sp_f4 f; } c; c.u = v; return c.f; };
  auto as_h8 = [](sp_u4 v) { union { sp_u4 u; sp_h8 h; } c; c.u = v; return c.h; };
  auto gload = [&](auto setc) {
    constexpr int S = decltype(setc)::value;
    const bool kin = g_c0 + 8 * q < Cin;          // Cin is a multiple of 8: whole 8-channel piece in or out
    const unsigned sa = RING ? (unsigned)(g_c0 * 4) : (unsigned)(((g_ty * W + g_tx) * ld_in + g_c0) * 4);
    const unsigned sb = (unsigned)((g_tap * Cin + g_c0) * 2);
#pragma unroll
    for (int a = 0; a < NPA; ++a) {
      unsigned off;
      if constexpr (RING) {
        const SfGather &gi = ring.in;
        const int ys = (int)(rpos[a] >> 8) + g_ty - pad, xs = (int)(rpos[a] & 255u) + g_tx - pad;
        const bool inside = (unsigned)ys < (unsigned)gi.G && (unsigned)xs < (unsigned)gi.G;
        const bool border = sf_frame_ring(gi.G, gi.lo, gi.hi, ys, xs);
        const unsigned o_ring = rring[a] + (unsigned)(sf_frame_index(gi.G, gi.lo, gi.hi, ys, xs) * ld_in * 4);
        const unsigned o_map = rowoff[a] + (unsigned)((ys * gi.Wq + xs) * ld_in * 4);
        off = (kin && inside) ? (border ? o_ring : o_map) : OOB;
      } else {
        off = (kin && ((vmask[a] >> g_tap) & 1u)) ? rowoff[a] : OOB;
      }
      ra[S][a][0] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rsA, off, sa, 0));
      ra[S][a][1] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rsA, off == OOB ? OOB : off + 16, sa, 0));
    }
#pragma unroll
    for (int b = 0; b < NPB; ++b) {
      rbh[S][b] = __builtin_amdgcn_raw_buffer_load_b128(rsH, kin ? woff[b] : OOB, sb, 0);
      rbl[S][b] = __builtin_amdgcn_raw_buffer_load_b128(rsL, kin ? woff[b] : OOB, sb, 0);
    }
    g_c0 += BK;
    if (g_c0 >= Cin) {
      g_c0 = 0;
      ++g_tap;
      if (++g_tx == ks) { g_tx = 0; ++g_ty; }
    }
  };
  auto lstore = [&](auto setc) {     // register set S -> LDS set S
    constexpr int S = decltype(setc)::value;
    _Float16 *Ah = sm + S * SETH, *Al = Ah + BM * SP_LD, *Bh = Al + BM * SP_LD, *Bl = Bh + BN * SP_LD;
#pragma unroll
    for (int a = 0; a < NPA; ++a) {
      if constexpr (ASPLIT) {
        *reinterpret_cast<sp_f4 *>(Ah + (ri + 64 * a) * SP_LD + 8 * q) = ra[S][a][0];
        *reinterpret_cast<sp_f4 *>(Al + (ri + 64 * a) * SP_LD + 8 * q) = ra[S][a][1];
      } else {
        sp_h8 hi, lo;
        sp_split8(ra[S][a][0], ra[S][a][1], ascale, hi, lo, big);
        *reinterpret_cast<sp_h8 *>(Ah + (ri + 64 * a) * SP_LD + 8 * q) = hi;
        *reinterpret_cast<sp_h8 *>(Al + (ri + 64 * a) * SP_LD + 8 * q) = lo;
      }
    }
#pragma unroll
    for (int b = 0; b < NPB; ++b) {
      if (BN % 64 == 0 || ri + 64 * b < BN) {
        *reinterpret_cast<sp_u4 *>(Bh + (ri + 64 * b) * SP_LD + 8 * q) = rbh[S][b];
        *reinterpret_cast<sp_u4 *>(Bl + (ri + 64 * b) * SP_LD + 8 * q) = rbl[S][b];
      }
    }
  };
  const int aoff = (32 * TM * wm + (lane & 31)) * SP_LD + 8 * (lane >> 5);
  const int boff = (32 * TN * wn + (lane & 31)) * SP_LD + 8 * (lane >> 5);
  auto compute = [&](auto setc) {
    constexpr int S = decltype(setc)::value;
    const _Float16 *Ah = sm + S * SETH, *Al = Ah + BM * SP_LD, *Bh = Al + BM * SP_LD, *Bl = Bh + BN * SP_LD;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      sp_h8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const sp_h8 *>(Ah + aoff + i * 32 * SP_LD + 16 * kk);
        al[i] = *reinterpret_cast<const sp_h8 *>(Al + aoff + i * 32 * SP_LD + 16 * kk);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const sp_h8 *>(Bh + boff + j * 32 * SP_LD + 16 * kk);
        bl[j] = *reinterpret_cast<const sp_h8 *>(Bl + boff + j * 32 * SP_LD + 16 * kk);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);   // the small terms first
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  };
  using S0 = std::integral_constant<int, 0>;
  gload(S0{});
  lstore(S0{});
  __syncthreads();
  for (int it = 0; it < nit; ++it) {
    if (it + 1 < nit) gload(S0{});
    compute(S0{});
    __syncthreads();
    if (it + 1 < nit) {
      lstore(S0{});
      __syncthreads();
    }
  }
  // epilogue: acc[r] = D[row = (r&3) + 8(r>>2) + 4(lane>>5)][col = lane&31]; unscale (powers of two), bias, ReLU
  int *orow = reinterpret_cast<int *>(sm);           // side rows: where each of the tile's rows goes (the operand tiles are done with)
  if (ring.side) {
    for (int i = tid; i < BM; i += 256) {
      const int m = m0 + i;
      int o = 0;
      if (m < M) {
        const int n = m / ring.nout, j = m - n * ring.nout;
        int y, x;
        sf_side_position(ring.in.G, ring.olo, ring.ohi, j, y, x);
        o = n * ring.nfull + sf_frame_index(ring.in.G, ring.olo, ring.ohi, y, x);
      }
      orow[i] = o;
    }
    __syncthreads();
  }
  const float ia = 1.0f / ascale;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = n0 + 32 * (TN * wn + j) + (lane & 31);
    if (co < Cout) {
      const float bb = bias[co], sc = wscale[co] * ia;
      const int sg = (co < dst.end[0]) ? 0 : ((co < dst.end[1]) ? 1 : 2);
      const int cbase = (sg == 0) ? 0 : dst.end[sg - 1];
      float *op = dst.p[sg] + dst.off[sg] + (co - cbase);
      const int ld = dst.ld[sg];
      if (dst.fmt[sg] == 1) {      // split format: halves (m 2 ld + 16 (c / 8) + c % 8) and + 8
        const int cl = co - cbase;
        _Float16 *hp = reinterpret_cast<_Float16 *>(dst.p[sg]) + 16 * (cl >> 3) + (cl & 7);
        const float os = dst.oscale[sg];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + 32 * (TM * wm + i) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const float v = fmaxf(__builtin_fmaf(acc[i][j][r], sc, bb), 0.f) * os;
            const _Float16 h = (_Float16)v;
            big = fmaxf(big, v);
            if (m < M) {
              const size_t om = ring.side ? (size_t)orow[m - m0] : (size_t)m;
              hp[om * 2 * ld] = h;
              hp[om * 2 * ld + 8] = (_Float16)(v - (float)h);
            }
          }
      } else if (dst.fmt[sg] == 2) {
        // maxpool4 (googlenet1.py:75: 2 x 2 stride 2 on 16 x 16) in registers: a 32-row block of the tile is two rows of one window's grid
        // (row = x + 16 dy) and a lane holds, for px = a + 4 b + 2 (lane >> 5), all four of r = (2 a + e) + 4 (b + 2 dy).  Bias, scale (> 0)
        // and ReLU are monotone: applied to the largest accumulator they give the largest output -- the bits of pooling afterwards.
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int mb = m0 + 32 * (TM * wm + i);
          if (mb < M) {
            const size_t prow = (size_t)(mb >> 8) * 64 + (size_t)((mb >> 5) & 7) * 8 + 2 * (lane >> 5);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int b = 0; b < 2; ++b) {
                const float v = fmaxf(fmaxf(acc[i][j][2 * a + 4 * b], acc[i][j][2 * a + 1 + 4 * b]),
                                      fmaxf(acc[i][j][2 * a + 4 * (b + 2)], acc[i][j][2 * a + 1 + 4 * (b + 2)]));
                op[(prow + a + 4 * b) * ld] = fmaxf(__builtin_fmaf(v, sc, bb), 0.f);
              }
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + 32 * (TM * wm + i) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m < M) op[(ring.side ? (size_t)orow[m - m0] : (size_t)m) * ld] = fmaxf(__builtin_fmaf(acc[i][j][r], sc, bb), 0.f);
          }
      }
    }
  }
  if (!(big < 65504.f)) *overflow = 1;      // (every writer stores 1; NaN counts)
}

// per output channel: e = the power of two that puts max |w| into [2^12, 2^13); hi / lo halves of w 2^e; wscale = 2^-e
__global__ void k_split_weights(const float *__restrict__ w, int Cout, int K, _Float16 *__restrict__ hi, _Float16 *__restrict__ lo,
                                float *__restrict__ wscale) {
  __shared__ float red[256];
  const int co = blockIdx.x, tid = threadIdx.x;
  const float *wr = w + (size_t)co * K;
  float mx = 0.f;
  for (int i = tid; i < K; i += 256) mx = fmaxf(mx, fabsf(wr[i]));
  red[tid] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
    __syncthreads();
  }
  mx = red[0];
  int e = 0;
  if (mx > 0.f && mx < 3.0e38f) {
    int ex;
    frexpf(mx, &ex);        // mx = f 2^ex, f in [0.5, 1)
    e = 13 - ex;            // mx 2^e in [2^12, 2^13)
    e = e > 60 ? 60 : (e < -60 ? -60 : e);
  }
  const float s = ldexpf(1.0f, e);
  if (tid == 0) wscale[co] = ldexpf(1.0f, -e);
  for (int i = tid; i < K; i += 256) {
    const float v = wr[i] * s;
    const _Float16 h = (_Float16)v;
    hi[(size_t)co * K + i] = h;
    lo[(size_t)co * K + i] = (_Float16)(v - (float)h);
  }
}

// ---- inception branch 4 by operand splitting: 3 x 3 stride-1 pad-1 max pool + 1 x 1 convolution in one launch (googlenet1.py:213-214).
// The staging of cnn_kernels.hip's k_poolconv -- the tile's 128 pixels (whole image rows: W divides 128) plus one image row above and
// below, raw float32, linear in LDS; every thread pools its (pixel, channel quad) items from there -- feeding the split operand tiles
// of k_conv_split: the pooled values are split on their way into the A tiles.
template <int BN>
__global__ __launch_bounds__(256, 2) void k_poolconv_split(const float *__restrict__ in, int M, int H, int W, int Cin,
                                                           const _Float16 *__restrict__ whi, const _Float16 *__restrict__ wlo,
                                                           const float *__restrict__ wscale, const float *__restrict__ bias, int Cout,
                                                           float ascale, float *__restrict__ out, int ld_out, int ch_off,
                                                           int *__restrict__ overflow, int pool2) {
  constexpr int BM = 128, BK = 32, KQ = BK / 4;
  constexpr int WN = (BN >= 128) ? 2 : 1, WM = 4 / WN, TM = BM / (32 * WM), TN = BN / (32 * WN), NPB = BN / 64;
  constexpr int RMAX = BM + 2 * 32;                         // raw pixels at W = 32
  constexpr int NRAW = (RMAX * KQ + 255) / 256, NPOOL = BM * KQ / 256;
  __shared__ __attribute__((aligned(16))) float raw[RMAX * BK + BK];      // (+ a pixel of zeros: where a masked tap reads)
  __shared__ __attribute__((aligned(16))) _Float16 sm[(2 * BM + 2 * BN) * SP_LD];
  _Float16 *Ah = sm, *Al = Ah + BM * SP_LD, *Bh = Al + BM * SP_LD, *Bl = Bh + BN * SP_LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int q = tid & 3, ri = tid >> 2;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int R = BM + 2 * W, nchunk = (Cin + BK - 1) / BK;
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (unsigned)((size_t)M * Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(whi), 0, (unsigned)((size_t)Cout * Cin * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(wlo), 0, (unsigned)((size_t)Cout * Cin * 2), 0x00020000);
  unsigned rawoff[NRAW];
  int rawq[NRAW];
#pragma unroll
  for (int r = 0; r < NRAW; ++r) {
    const int idx = tid + 256 * r, rpx = idx / KQ, quad = idx - rpx * KQ;
    const long long m = (long long)m0 - W + rpx;
    rawq[r] = 4 * quad;
    rawoff[r] = (rpx < R && m >= 0 && m < M) ? (unsigned)(((size_t)m * Cin + 4 * quad) * 4) : OOB;
  }
  unsigned woff[NPB];
#pragma unroll
  for (int b = 0; b < NPB; ++b) {
    const int co = n0 + ri + 64 * b;
    woff[b] = (co < Cout) ? (unsigned)(((size_t)co * Cin + 8 * q) * 2) : OOB;
  }
  const int pq = tid % KQ, pblk = tid / KQ;                 // pooled items: quad pq of pixel 32 a + permuted block (k_poolconv's bank order)
  int ppx[NPOOL];
  int ptap[NPOOL][9];          // quad index of every tap of every pooled item in `raw` (the zero pixel for a tap outside the image):
                               // unconditional LDS reads -- a predicated read compiles to a branch and a wait per tap (193 branches
                               // and 87 full LDS waits per chunk loop: 3.7 us per chunk; round 6)
#pragma unroll
  for (int a = 0; a < NPOOL; ++a) {
    const int b8 = pblk & 7;
    const int perm = (b8 & 4) | ((b8 & 1) << 1) | ((b8 >> 1) & 1);             // 0 2 1 3 4 6 5 7
    const int px = (256 / KQ) * a + (pblk & ~7) + perm;
    ppx[a] = px;
    const int m = m0 + px;
    const bool ok = m < M;
    const int mm = ok ? m : 0;
    const int x = mm % W, y = (mm / W) % H;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
      const bool in = ok && yy >= 0 && yy < H && xx >= 0 && xx < W;
      ptap[a][tp] = in ? (px + W + (tp / 3 - 1) * W + (tp % 3 - 1)) * KQ + pq : RMAX * KQ + pq;      // (in 16-byte quads)
    }
  }
  if (tid < BK / 4) *reinterpret_cast<sp_f4 *>(raw + RMAX * BK + 4 * tid) = sp_f4{0.f, 0.f, 0.f, 0.f};
  sp_f16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float big = 0.f;
  sp_u4 rr[NRAW], rbh[NPB], rbl[NPB];
  auto gload = [&](int it) {
    const int c0 = it * BK;
    const unsigned sa = (unsigned)(c0 * 4), sb = (unsigned)(c0 * 2);
#pragma unroll
    for (int r = 0; r < NRAW; ++r) rr[r] = __builtin_amdgcn_raw_buffer_load_b128(rsA, (c0 + rawq[r] < Cin) ? rawoff[r] : OOB, sa, 0);
    const bool kin = c0 + 8 * q < Cin;
#pragma unroll
    for (int b = 0; b < NPB; ++b) {
      rbh[b] = __builtin_amdgcn_raw_buffer_load_b128(rsH, kin ? woff[b] : OOB, sb, 0);
      rbl[b] = __builtin_amdgcn_raw_buffer_load_b128(rsL, kin ? woff[b] : OOB, sb, 0);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int r = 0; r < NRAW; ++r)
      if (tid + 256 * r < RMAX * KQ) *reinterpret_cast<sp_u4 *>(raw + 4 * (tid + 256 * r)) = rr[r];
#pragma unroll
    for (int b = 0; b < NPB; ++b) {
      *reinterpret_cast<sp_u4 *>(Bh + (ri + 64 * b) * SP_LD + 8 * q) = rbh[b];
      *reinterpret_cast<sp_u4 *>(Bl + (ri + 64 * b) * SP_LD + 8 * q) = rbl[b];
    }
  };
  auto pool = [&]() {             // A tiles = hi / lo halves of the max over the 3 x 3 neighbourhood
#pragma unroll
    for (int a = 0; a < NPOOL; ++a) {
      sp_f4 mx = {0.f, 0.f, 0.f, 0.f};      // (0 is the identity: the activations are ReLU outputs)
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const sp_f4 v = reinterpret_cast<const sp_f4 *>(raw)[ptap[a][tp]];
        mx.x = fmaxf(mx.x, v.x); mx.y = fmaxf(mx.y, v.y); mx.z = fmaxf(mx.z, v.z); mx.w = fmaxf(mx.w, v.w);
      }
      const float v4[4] = {mx.x * ascale, mx.y * ascale, mx.z * ascale, mx.w * ascale};
      _Float16 h4[4], l4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        h4[k] = (_Float16)v4[k];
        l4[k] = (_Float16)(v4[k] - (float)h4[k]);
        big = fmaxf(big, v4[k]);
      }
      typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
      *reinterpret_cast<h4_t *>(Ah + ppx[a] * SP_LD + 4 * pq) = h4_t{h4[0], h4[1], h4[2], h4[3]};
      *reinterpret_cast<h4_t *>(Al + ppx[a] * SP_LD + 4 * pq) = h4_t{l4[0], l4[1], l4[2], l4[3]};
    }
  };
  const int aoff = (32 * TM * wm + (lane & 31)) * SP_LD + 8 * (lane >> 5);
  const int boff = (32 * TN * wn + (lane & 31)) * SP_LD + 8 * (lane >> 5);
  gload(0);
  stage();
  __syncthreads();
  pool();
  __syncthreads();
  for (int it = 0; it < nchunk; ++it) {
    if (it + 1 < nchunk) gload(it + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      sp_h8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const sp_h8 *>(Ah + aoff + i * 32 * SP_LD + 16 * kk);
        al[i] = *reinterpret_cast<const sp_h8 *>(Al + aoff + i * 32 * SP_LD + 16 * kk);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const sp_h8 *>(Bh + boff + j * 32 * SP_LD + 16 * kk);
        bl[j] = *reinterpret_cast<const sp_h8 *>(Bl + boff + j * 32 * SP_LD + 16 * kk);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (it + 1 < nchunk) {
      stage();
      __syncthreads();
      pool();
      __syncthreads();
    }
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = n0 + 32 * (TN * wn + j) + (lane & 31);
    if (co < Cout) {
      const float bb = bias[co], sc = wscale[co] * (1.0f / ascale);
      float *op = out + ch_off + co;
      if (pool2) {      // maxpool4 in registers (k_conv_split's fmt 2: 16 x 16 grids)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int mb = m0 + 32 * (TM * wm + i);
          if (mb < M) {
            const size_t prow = (size_t)(mb >> 8) * 64 + (size_t)((mb >> 5) & 7) * 8 + 2 * (lane >> 5);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int b = 0; b < 2; ++b) {
                const float v = fmaxf(fmaxf(acc[i][j][2 * a + 4 * b], acc[i][j][2 * a + 1 + 4 * b]),
                                      fmaxf(acc[i][j][2 * a + 4 * (b + 2)], acc[i][j][2 * a + 1 + 4 * (b + 2)]));
                op[(prow + a + 4 * b) * ld_out] = fmaxf(__builtin_fmaf(v, sc, bb), 0.f);
              }
          }
        }
      } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + 32 * (TM * wm + i) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (m < M) op[(size_t)m * ld_out] = fmaxf(__builtin_fmaf(acc[i][j][r], sc, bb), 0.f);
        }
      }
    }
  }
  if (!(big < 65504.f)) *overflow = 1;
}


template <int BN, int BM = 128>
int launch_split(const float *in, int in_split, int M, int H, int W, int Cin, int ld_in, const _Float16 *whi, const _Float16 *wlo,
                 const float *wscale, const float *bias, int Cout, int ks, float ascale, const ConvDstS &dst, int *overflow,
                 hipStream_t st, const RingArgs &rows = RingArgs{}) {
  dim3 grid(8 * sf_cdiv(sf_cdiv(M, BM), 8) * sf_cdiv(Cout, BN));
  if (in_split)
    hipLaunchKernelGGL((k_conv_split<BN, true, BM>), grid, dim3(256), 0, st, in, M, H, W, Cin, ld_in, whi, wlo, wscale, bias, Cout, ks, ascale, dst, overflow, rows);
  else
    hipLaunchKernelGGL((k_conv_split<BN, false, BM>), grid, dim3(256), 0, st, in, M, H, W, Cin, ld_in, whi, wlo, wscale, bias, Cout, ks, ascale, dst, overflow, rows);
  SF_LAUNCH_CHECK("k_conv_split");
  return 0;
}

// max |x| over n floats, OR-ed into *amax (a device float the caller zeroed; values are compared as their bit patterns, so the
// result does not depend on the order): the calibration of the split-operand layers' activation scales (cnn_driver.hip)
__global__ void k_absmax(const float *__restrict__ x, size_t n, float *__restrict__ amax) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = fabsf(x[i]);
    m = (v > m || v != v) ? v : m;                    // (a NaN wins: its pattern is above every finite one)
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float t = __shfl_xor(m, o);
    m = (__float_as_uint(t) > __float_as_uint(m)) ? t : m;
  }
  if ((threadIdx.x & 63) == 0 && m != 0.f) atomicMax(reinterpret_cast<unsigned *>(amax), __float_as_uint(m));
}

}  // namespace

extern "C" {

int sf_cnn_split_weights(const float *w, int Cout, int K, void *hi, void *lo, float *wscale, void *stream) {
  if (!w || !hi || !lo || !wscale || Cout < 1 || K < 1) { sf_set_error("sf_cnn_split_weights: bad argument"); return -1; }
  hipLaunchKernelGGL(k_split_weights, dim3(Cout), dim3(256), 0, (hipStream_t)stream, w, Cout, K, reinterpret_cast<_Float16 *>(hi),
                     reinterpret_cast<_Float16 *>(lo), wscale);
  SF_LAUNCH_CHECK("k_split_weights");
  return 0;
}

static int split_go(const float *in, int in_split, int N, int H, int W, int Cin, int ld_in, const void *whi, const void *wlo, const float *wscale,
                    const float *bias, int Cout, int ksize, float ascale, const ConvDstS &d, int *overflow, hipStream_t st,
                    const RingArgs &rows = RingArgs{}) {
  const long long Ml = (long long)N * H * W;
  const size_t halo = 2 * ((size_t)(ksize >> 1) * W + (ksize >> 1)) * ld_in * 4 + 64, img = (size_t)H * W * ld_in * 4;
  if ((size_t)Cout * ksize * ksize * Cin * 2 >= 0x7ff00000u || img + halo >= 0x7ff00000u) {
    sf_set_error("sf_cnn_conv_split: weights or one image of 2 GB or more (use sf_cnn_conv)");
    return -2;
  }
  if ((size_t)N * img + halo >= 0x7ff00000u && rows.side) { sf_set_error("sf_cnn_conv_split: scattered rows of 2 GB or more"); return -2; }
  if ((size_t)N * img + halo >= 0x7ff00000u) {   // the buffer descriptors address < 2 GB: images are independent, run them in pieces
    const int per = (int)((0x7ff00000u - 1 - halo) / img);
    for (int n0 = 0; n0 < N; n0 += per) {
      const int nn = (N - n0 < per) ? N - n0 : per;
      ConvDstS dd = d;
      for (int k = 0; k < 3; ++k) dd.p[k] = d.p[k] + (size_t)n0 * H * W * d.ld[k];
      if (int rc = split_go(in + (size_t)n0 * H * W * ld_in, in_split, nn, H, W, Cin, ld_in, whi, wlo, wscale, bias, Cout, ksize, ascale, dd, overflow, st))
        return rc;
    }
    return 0;
  }
  const _Float16 *h = reinterpret_cast<const _Float16 *>(whi), *l = reinterpret_cast<const _Float16 *>(wlo);
  // The channel tile: 128 where it pads the output channels no more than 64-channel tiles do.  (128 throughout: 41-42.3 k
  // windows/s against 43.2-43.7 k in alternating runs; 192 / 256-channel tiles need 228 / 270 registers, run two / one workgroup
  // per CU instead of three and were slower still: what hides this kernel's trips to L2 is occupancy; 96-channel tiles of 256
  // pixels for 96 / 192 / 288 channels -- 236 registers, two workgroups per CU -- were the same within the noise.)
  // 160-channel tiles of 128 pixels (round 6; 4 x 1 waves of 32 x 160): at equal padding they run at the speed of the others when the
  // launch fills many rounds of the 256 CUs (A/B on inception4a-4e: 181 vs 181 us for the stacked 1 x 1, the 3 x 3 layers 5 % slower),
  // so they are taken only where they balance the launch better -- the per-CU load ceil(workgroups / 256) x tile area:
  // inception5a's 3 x 3 124 -> 107 us, 5b's stacked 1 x 1 142 -> 137 (profiles/r06_conv_bound.md)
  {
    const bool wide = Cout > 64 && sf_cdiv(Cout, 128) * 128 <= sf_cdiv(Cout, 64) * 64;
    const long long bm = wide ? 128 : 256, bn = wide ? 128 : 64;
    auto cd = [](long long a, long long b) { return (a + b - 1) / b; };
    const long long load_now = cd(cd(Ml, bm) * cd(Cout, bn), 256) * bm * bn;
    const long long load_160 = cd(cd(Ml, 128) * cd(Cout, 160), 256) * 128 * 160;
    if (Cout > 128 && load_160 * 20 < load_now * 19 && sf_tune().cnn_variant != 2)
      return launch_split<160>(in, in_split, (int)Ml, H, W, Cin, ld_in, h, l, wscale, bias, Cout, ksize, ascale, d, overflow, st, rows);
  }
  if (Cout > 64 && sf_cdiv(Cout, 128) * 128 <= sf_cdiv(Cout, 64) * 64)
    return launch_split<128>(in, in_split, (int)Ml, H, W, Cin, ld_in, h, l, wscale, bias, Cout, ksize, ascale, d, overflow, st, rows);
  // 64-channel tiles take 256 pixels: wave tiles of 64 x 64 as in the 128 x 128 form -- 8 fragment reads per 12 matrix instructions
  // instead of 6 per 6 (the kernel sits at the LDS's bandwidth): 48.9 k -> 49.9 k windows/s
  return launch_split<64, 256>(in, in_split, (int)Ml, H, W, Cin, ld_in, h, l, wscale, bias, Cout, ksize, ascale, d, overflow, st, rows);
}

static bool sp_pow2(float v) { int e; return v > 0.f && v < 3.0e38f && frexpf(v, &e) == 0.5f; }

static int conv_split_go(const float *in, int in_split, int N, int H, int W, int Cin, int ld_in, const void *whi, const void *wlo,
                      const float *wscale, const float *bias, int Cout, int ksize, float ascale, float *out, int out_split, float oscale,
                      int ld_out, int ch_off, int *overflow, void *stream, int pool2) {
  if (pool2 && (out_split || H != 16 || W != 16)) { sf_set_error("sf_cnn_conv_split: the pooled epilogue takes float32 16 x 16 grids"); return -1; }
  if (!in || !whi || !wlo || !wscale || !bias || !out || !overflow || N < 1 || (ksize != 1 && ksize != 3) || (Cin & 7) || (ld_in & 3) ||
      Cin > ld_in || ch_off < 0 || ch_off + Cout > ld_out || !sp_pow2(ascale) || (in_split && ld_in != Cin) ||
      (out_split && (ld_out != Cout || ch_off != 0 || (Cout & 7) || !sp_pow2(oscale)))) {
    sf_set_error("sf_cnn_conv_split: bad argument (ksize 1|3, Cin multiple of 8; split-format tensors are dense; ascale / oscale powers "
                 "of two; overflow: a device int of the caller)");
    return -1;
  }
  ConvDstS d{};
  d.p[0] = d.p[1] = d.p[2] = out;
  d.ld[0] = d.ld[1] = d.ld[2] = ld_out;
  d.off[0] = d.off[1] = d.off[2] = ch_off;
  d.end[0] = d.end[1] = d.end[2] = Cout;
  d.fmt[0] = d.fmt[1] = d.fmt[2] = out_split ? 1 : (pool2 ? 2 : 0);
  d.oscale[0] = d.oscale[1] = d.oscale[2] = out_split ? oscale : 1.0f;
  return split_go(in, in_split, N, H, W, Cin, ld_in, whi, wlo, wscale, bias, Cout, ksize, ascale, d, overflow, (hipStream_t)stream);
}
int sf_cnn_conv_split(const float *in, int in_split, int N, int H, int W, int Cin, int ld_in, const void *whi, const void *wlo,
                      const float *wscale, const float *bias, int Cout, int ksize, float ascale, float *out, int out_split, float oscale,
                      int ld_out, int ch_off, int *overflow, void *stream) {
  return conv_split_go(in, in_split, N, H, W, Cin, ld_in, whi, wlo, wscale, bias, Cout, ksize, ascale, out, out_split, oscale, ld_out, ch_off,
                       overflow, stream, 0);
}

static int conv_split3_go(const float *in, int N, int H, int W, int Cin, int ld_in, const void *whi, const void *wlo,
                             const float *wscale, const float *bias, int c0, int c1, int c2, float ascale, float *out0, int ld0,
                             int off0, float *out1, int ld1, int off1, float *out2, int ld2, int off2, int out12_split, float oscale1,
                             float oscale2, int *overflow, void *stream, int pool2) {
  if (pool2 && (H != 16 || W != 16)) { sf_set_error("sf_cnn_conv_split3_split: the pooled epilogue takes 16 x 16 grids"); return -1; }
  if (!in || !whi || !wlo || !wscale || !bias || !out0 || !out1 || !out2 || !overflow || N < 1 || (Cin & 7) || (ld_in & 3) || Cin > ld_in ||
      c0 < 1 || c1 < 1 || c2 < 1 || off0 + c0 > ld0 || off1 + c1 > ld1 || off2 + c2 > ld2 || !sp_pow2(ascale) ||
      (out12_split && (ld1 != c1 || off1 != 0 || (c1 & 7) || ld2 != c2 || off2 != 0 || (c2 & 7) || !sp_pow2(oscale1) || !sp_pow2(oscale2)))) {
    sf_set_error("sf_cnn_conv_split3_split: bad argument");
    return -1;
  }
  ConvDstS d{};
  d.p[0] = out0; d.p[1] = out1; d.p[2] = out2;
  d.ld[0] = ld0; d.ld[1] = ld1; d.ld[2] = ld2;
  d.off[0] = off0; d.off[1] = off1; d.off[2] = off2;
  d.end[0] = c0; d.end[1] = c0 + c1; d.end[2] = c0 + c1 + c2;
  d.fmt[0] = pool2 ? 2 : 0; d.fmt[1] = d.fmt[2] = out12_split ? 1 : 0;
  d.oscale[0] = 1.0f; d.oscale[1] = out12_split ? oscale1 : 1.0f; d.oscale[2] = out12_split ? oscale2 : 1.0f;
  return split_go(in, 0, N, H, W, Cin, ld_in, whi, wlo, wscale, bias, c0 + c1 + c2, 1, ascale, d, overflow, (hipStream_t)stream);
}
int sf_cnn_conv_split3_split(const float *in, int N, int H, int W, int Cin, int ld_in, const void *whi, const void *wlo,
                             const float *wscale, const float *bias, int c0, int c1, int c2, float ascale, float *out0, int ld0,
                             int off0, float *out1, int ld1, int off1, float *out2, int ld2, int off2, int out12_split, float oscale1,
                             float oscale2, int *overflow, void *stream) {
  return conv_split3_go(in, N, H, W, Cin, ld_in, whi, wlo, wscale, bias, c0, c1, c2, ascale, out0, ld0, off0, out1, ld1, off1, out2, ld2, off2,
                        out12_split, oscale1, oscale2, overflow, stream, 0);
}

// A convolution (1 x 1 or 3 x 3) at the ring positions of the frame (olo, ohi) of every window of a batch, its input gathered from the
// window's own ring tensor or the shared phase maps (trunk sharing, cnn_share.hip / cnn_ring.h): `maps` = the input's phase maps
// [P * P][Hq][Wq][Cin] with the batch's ring tensor [N][count(G, ilo, ihi)][Cin] `ring_off` floats behind their start (float32, or
// both in the split format scaled by ascale when in_split); rows m = window * nout + ring index.  Output channels as
// sf_cnn_conv_split3_split: [0, c0) -> out0 (float32), [c0, c0 + c1) -> out1, the rest -> out2 (c1 = c2 = 0: one segment).
static int conv_ring_go(const float *maps, int in_split, long long tile0, int N, int W, int Rb, int Hq, int Wq, size_t ring_off, int shift,
                     int G, int ilo, int ihi, int olo, int ohi, int Cin, const void *whi, const void *wlo, const float *wscale,
                     const float *bias, int c0, int c1, int c2, int ksize, float ascale, float *out0, int ld0, int off0, float *out1,
                     int ld1, int off1, float *out2, int ld2, int off2, int out12_split, float oscale1, float oscale2, int *overflow,
                     void *stream, int side) {
  const int nin = sf_frame_count(G, ilo, ihi), nfull = sf_frame_count(G, olo, ohi), nout = side ? sf_side_count(G, olo, ohi) : nfull;
  const int Cout = c0 + c1 + c2;
  const size_t total = (ring_off + (size_t)(N + 1) * nin * Cin) * 4;
  if (!maps || !whi || !wlo || !wscale || !bias || !out0 || !overflow || N < 1 || W < 1 || tile0 < 0 || (ksize != 1 && ksize != 3) ||
      (Cin & 7) || c0 < 1 || c1 < 0 || c2 < 0 || (c2 > 0 && c1 < 1) || off0 + c0 > ld0 || (c1 > 0 && (!out1 || off1 + c1 > ld1)) ||
      (c2 > 0 && (!out2 || off2 + c2 > ld2)) || G < 4 || G > 128 || shift < 1 || shift > 4 || ilo < 0 || ihi < 0 || olo < 0 || ohi < 0 ||
      ilo + ihi >= G || olo + ohi >= G || nout < 1 || Hq < G || Wq < G || !sp_pow2(ascale) ||
      ring_off < ((size_t)1 << (2 * shift)) * Hq * Wq * Cin || total >= 0x7ff00000u ||
      (out12_split && ((c1 > 0 && (ld1 != c1 || off1 != 0 || (c1 & 7) || !sp_pow2(oscale1))) ||
                       (c2 > 0 && (ld2 != c2 || off2 != 0 || (c2 & 7) || !sp_pow2(oscale2)))))) {
    sf_set_error("sf_cnn_conv_ring: bad argument (maps + ring tensor below 2 GB; Cin a multiple of 8)");
    return -1;
  }
  ConvDstS d{};
  d.p[0] = out0; d.p[1] = c1 > 0 ? out1 : out0; d.p[2] = c2 > 0 ? out2 : out0;
  d.ld[0] = ld0; d.ld[1] = c1 > 0 ? ld1 : ld0; d.ld[2] = c2 > 0 ? ld2 : ld0;
  d.off[0] = off0; d.off[1] = c1 > 0 ? off1 : off0; d.off[2] = c2 > 0 ? off2 : off0;
  d.end[0] = c0; d.end[1] = c0 + c1; d.end[2] = Cout;
  d.fmt[0] = 0; d.fmt[1] = (out12_split && c1 > 0) ? 1 : 0; d.fmt[2] = (out12_split && c2 > 0) ? 1 : 0;
  d.oscale[0] = 1.0f; d.oscale[1] = d.fmt[1] ? oscale1 : 1.0f; d.oscale[2] = d.fmt[2] ? oscale2 : 1.0f;
  RingArgs ra{};
  ra.in = SfGather{tile0, W, Rb, Hq, Wq, shift, G, ilo, ihi, (unsigned)ring_off};
  ra.olo = olo; ra.ohi = ohi; ra.nout = nout; ra.nin = nin; ra.side = side; ra.nfull = nfull;
  const long long Ml = (long long)N * nout;
  if (Ml * (long long)(ld0 > Cout ? ld0 : Cout) * 4 >= (1ll << 40) || Ml >= 0x7fffffff) { sf_set_error("sf_cnn_conv_ring: batch too large"); return -2; }
  const int M = (int)Ml;
  const _Float16 *h = reinterpret_cast<const _Float16 *>(whi), *l = reinterpret_cast<const _Float16 *>(wlo);
  hipStream_t st = (hipStream_t)stream;
#define SF_RING_LAUNCH(BN, BM)                                                                                                          \
  {                                                                                                                                     \
    dim3 grid(8 * sf_cdiv(sf_cdiv(M, BM), 8) * sf_cdiv(Cout, BN));                                                                     \
    if (in_split)                                                                                                                       \
      hipLaunchKernelGGL((k_conv_split<BN, true, BM, true>), grid, dim3(256), 0, st, maps, M, G, G, Cin, Cin, h, l, wscale, bias, Cout, \
                         ksize, ascale, d, overflow, ra);                                                                               \
    else                                                                                                                                \
      hipLaunchKernelGGL((k_conv_split<BN, false, BM, true>), grid, dim3(256), 0, st, maps, M, G, G, Cin, Cin, h, l, wscale, bias,     \
                         Cout, ksize, ascale, d, overflow, ra);                                                                         \
  }
  if (Cout > 64 && sf_cdiv(Cout, 128) * 128 <= sf_cdiv(Cout, 64) * 64) SF_RING_LAUNCH(128, 128)
  else SF_RING_LAUNCH(64, 256)
#undef SF_RING_LAUNCH
  SF_LAUNCH_CHECK("k_conv_split<ring>");
  return 0;
}

int sf_cnn_conv_ring(const float *maps, int in_split, long long tile0, int N, int W, int Rb, int Hq, int Wq, size_t ring_off, int shift,
                     int G, int ilo, int ihi, int olo, int ohi, int Cin, const void *whi, const void *wlo, const float *wscale,
                     const float *bias, int c0, int c1, int c2, int ksize, float ascale, float *out0, int ld0, int off0, float *out1,
                     int ld1, int off1, float *out2, int ld2, int off2, int out12_split, float oscale1, float oscale2, int *overflow,
                     void *stream) {
  return conv_ring_go(maps, in_split, tile0, N, W, Rb, Hq, Wq, ring_off, shift, G, ilo, ihi, olo, ohi, Cin, whi, wlo, wscale, bias, c0, c1, c2,
                      ksize, ascale, out0, ld0, off0, out1, ld1, off1, out2, ld2, off2, out12_split, oscale1, oscale2, overflow, stream, 0);
}

// 1 when sf_cnn_pool_conv_split takes this geometry (whole image rows per 128-pixel tile, operands below 2 GB); else sf_cnn_pool_conv
int sf_cnn_pool_conv_split_ok(int N, int H, int W, int Cin, int Cout) {
  return (W >= 1 && W <= 32 && 128 % W == 0 && (Cin & 7) == 0 && (size_t)N * H * W * Cin * 4 < 0x7ff00000u &&
          (size_t)Cout * Cin * 2 < 0x7ff00000u) ? 1 : 0;
}
static int pool_conv_go(const float *in, int N, int H, int W, int Cin, const void *whi, const void *wlo, const float *wscale,
                           const float *bias, int Cout, float ascale, float *out, int ld_out, int ch_off, int *overflow, void *stream, int pool2) {
  if (pool2 && (H != 16 || W != 16)) { sf_set_error("sf_cnn_pool_conv_split: the pooled epilogue takes 16 x 16 grids"); return -1; }
  if (!in || !whi || !wlo || !wscale || !bias || !out || !overflow || N < 1 || ch_off < 0 || ch_off + Cout > ld_out || !sp_pow2(ascale) ||
      !sf_cnn_pool_conv_split_ok(N, H, W, Cin, Cout)) {
    sf_set_error("sf_cnn_pool_conv_split: bad argument (dense input, W dividing 128, Cin multiple of 8, operands < 2 GB, ascale a power of two)");
    return -1;
  }
  const int M = N * H * W;
  const _Float16 *h = reinterpret_cast<const _Float16 *>(whi), *l = reinterpret_cast<const _Float16 *>(wlo);
  if (Cout > 64)
    hipLaunchKernelGGL((k_poolconv_split<128>), dim3(sf_cdiv(M, 128), sf_cdiv(Cout, 128)), dim3(256), 0, (hipStream_t)stream, in, M, H, W,
                       Cin, h, l, wscale, bias, Cout, ascale, out, ld_out, ch_off, overflow, pool2);
  else
    hipLaunchKernelGGL((k_poolconv_split<64>), dim3(sf_cdiv(M, 128), sf_cdiv(Cout, 64)), dim3(256), 0, (hipStream_t)stream, in, M, H, W,
                       Cin, h, l, wscale, bias, Cout, ascale, out, ld_out, ch_off, overflow, pool2);
  SF_LAUNCH_CHECK("k_poolconv_split");
  return 0;
}
int sf_cnn_pool_conv_split(const float *in, int N, int H, int W, int Cin, const void *whi, const void *wlo, const float *wscale,
                           const float *bias, int Cout, float ascale, float *out, int ld_out, int ch_off, int *overflow, void *stream) {
  return pool_conv_go(in, N, H, W, Cin, whi, wlo, wscale, bias, Cout, ascale, out, ld_out, ch_off, overflow, stream, 0);
}

int sf_cnn_absmax(const float *x, size_t n, float *amax, void *stream) {
  if (!x || !amax || n < 1) { sf_set_error("sf_cnn_absmax: bad argument"); return -1; }
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(k_absmax, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, amax);
  SF_LAUNCH_CHECK("k_absmax");
  return 0;
}

}  // extern "C"

// ---- internal entry points (cnn_internal.h): the side-row forms for the band-sharing driver
int sfi_cnn_conv_side(const float *maps, int in_split, long long tile0, int N, int W, int Rb, int Hq, int Wq, size_t ring_off, int shift, int G,
                      int ilo, int ihi, int olo, int ohi, int Cin, const void *whi, const void *wlo, const float *wscale, const float *bias,
                      int c0, int c1, int c2, int ksize, float ascale, float *out0, int ld0, int off0, float *out1, int ld1, int off1,
                      float *out2, int ld2, int off2, int out12_split, float oscale1, float oscale2, int *overflow, void *stream) {
  if (olo + ohi < 1) { sf_set_error("sfi_cnn_conv_side: the frame has no side"); return -1; }
  return conv_ring_go(maps, in_split, tile0, N, W, Rb, Hq, Wq, ring_off, shift, G, ilo, ihi, olo, ohi, Cin, whi, wlo, wscale, bias, c0, c1, c2,
                      ksize, ascale, out0, ld0, off0, out1, ld1, off1, out2, ld2, off2, out12_split, oscale1, oscale2, overflow, stream, 1);
}

int sfi_cnn_conv_rows_side(const float *in, int N, int G, int olo, int ohi, int Cin, const void *whi, const void *wlo, const float *wscale,
                           const float *bias, int Cout, float ascale, float *out, int ld_out, int ch_off, int *overflow, void *stream) {
  if (!in || !whi || !wlo || !wscale || !bias || !out || !overflow || N < 1 || G < 4 || olo < 0 || ohi < 0 || olo + ohi < 1 || olo + ohi >= G ||
      (Cin & 7) || ch_off < 0 || ch_off + Cout > ld_out || !sp_pow2(ascale)) {
    sf_set_error("sfi_cnn_conv_rows_side: bad argument");
    return -1;
  }
  ConvDstS d{};
  d.p[0] = d.p[1] = d.p[2] = out;
  d.ld[0] = d.ld[1] = d.ld[2] = ld_out;
  d.off[0] = d.off[1] = d.off[2] = ch_off;
  d.end[0] = d.end[1] = d.end[2] = Cout;
  d.oscale[0] = d.oscale[1] = d.oscale[2] = 1.0f;
  RingArgs rows{};
  rows.in.G = G; rows.olo = olo; rows.ohi = ohi; rows.nout = sf_side_count(G, olo, ohi); rows.nfull = sf_frame_count(G, olo, ohi); rows.side = 1;
  return split_go(in, 0, 1, 1, N * rows.nout, Cin, Cin, whi, wlo, wscale, bias, Cout, 1, ascale, d, overflow, (hipStream_t)stream, rows);
}

// the block in front of maxpool4 (inception4e) with the pool taken in the epilogues: the float32 outputs land as [window][8][8][ld]
int sfi_cnn_conv_split_pool2(const float *in, int in_split, int N, int Cin, int ld_in, const void *whi, const void *wlo, const float *wscale,
                             const float *bias, int Cout, int ksize, float ascale, float *out, int ld_out, int ch_off, int *overflow, void *stream) {
  return conv_split_go(in, in_split, N, 16, 16, Cin, ld_in, whi, wlo, wscale, bias, Cout, ksize, ascale, out, 0, 1.0f, ld_out, ch_off, overflow,
                       stream, 1);
}
int sfi_cnn_conv_split3_pool2(const float *in, int N, int Cin, int ld_in, const void *whi, const void *wlo, const float *wscale, const float *bias,
                              int c0, int c1, int c2, float ascale, float *out0, int ld0, int off0, float *out1, float *out2, float oscale1,
                              float oscale2, int *overflow, void *stream) {
  return conv_split3_go(in, N, 16, 16, Cin, ld_in, whi, wlo, wscale, bias, c0, c1, c2, ascale, out0, ld0, off0, out1, c1, 0, out2, c2, 0, 1, oscale1,
                        oscale2, overflow, stream, 1);
}
int sfi_cnn_pool_conv_split_pool2(const float *in, int N, int Cin, const void *whi, const void *wlo, const float *wscale, const float *bias, int Cout,
                                  float ascale, float *out, int ld_out, int ch_off, int *overflow, void *stream) {
  return pool_conv_go(in, N, 16, 16, Cin, whi, wlo, wscale, bias, Cout, ascale, out, ld_out, ch_off, overflow, stream, 1);
}

