// Stage 3, production window: the per-column covariance on v_mfma_f64_4x4x4_f64 (cmf_cov.hip serves the others).
//
// Replaces numpy.cov as called by looshrinkage (cmf/robust_mf.py:52-70, :98, :130).  The 4x4x4 fp64 MFMA issues
// 1.5x the flops per cycle of the 16x16x4 one on MI355X (tools/microbench/mfma64.hip), and for X^T X it needs no
// LDS at all.  With lane = 16 q + 4 m + n the instruction computes, per block m,  D_m[i][j] += sum_k A_m[i][k] B_m[k][j]
// with A[i][k] at (q = k, n = i), B[k][j] at (q = k, n = j), D[i][j] at (q = i, n = j).  Take k = a row of the 4-row
// group m and i, j = bands inside 4-band groups I, J: ONE register per band group,
//     f[I] at lane (q, m, n) = x[row 4m + q][band(I, n)] - mu,
// is both the A operand of tile row I and the B operand of tile column J.  A wave keeps all 171 upper-triangular
// 4x4 tiles (x 4 blocks = 4 row groups, summed at the end) in registers and issues 171 MFMAs per 16 rows with
// every operand already in registers; band(I, n) = 18 n + I makes a lane's 18 values contiguous in the row.
// One 256-thread workgroup per CU (512-register waves), (column, row split) per workgroup as in the sweep.
#include "cmf_common.h"
#include <type_traits>

namespace {

constexpr int C4_NG = SF_SW4_NJ;                 // 18 band groups of 4
constexpr int C4_NTRI = C4_NG * (C4_NG + 1) / 2;  // 171 tiles

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
// index of tile (I, J), I <= J, in the row-major upper-triangular list
constexpr int tri_index(int I, int J) { return I * C4_NG - I * (I - 1) / 2 + (J - I); }

// 171 accumulators are 342 registers: more than either register file holds.  Left to the compiler they migrate
// between the files inside the loop (~200 v_accvgpr moves per tile, each one a stall of the MFMA stream), so
// the file of every accumulator is pinned: the first C4_NACC_A tiles live in AGPRs, the rest in VGPRs.
constexpr int C4_NACC_A = 112;
template <int T>
__device__ __forceinline__ void mfma_acc(double &acc, double a, double b) {
  if constexpr (T < C4_NACC_A) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

template <int CTRL>
__device__ __forceinline__ double dpp_row(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

template <int EXP>  // 0 = production; 1 = no loads in the loop, 2 = no MFMAs (timing experiments, wrong results)
__global__ __launch_bounds__(256, 1) void k_syrk4(const float *__restrict__ xt, const uint8_t *__restrict__ mask_t,
                                                   const double *__restrict__ mu, int L, int p, int rows_per_wg,
                                                   double *__restrict__ part) {
  constexpr int NG = C4_NG, PS = 4 * C4_NG;
  __shared__ double red[4][C4_NTRI][16];   // 87.5 KB: the four waves' tiles before the final sum
  __shared__ double mus[PS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, m = (lane >> 2) & 3, n = lane & 3;
  const int c = blockIdx.x, split = blockIdx.y;
  for (int i = tid; i < PS; i += 256) mus[i] = (i < p) ? mu[(size_t)c * p + i] : 0.0;
  __syncthreads();
  const int rbeg = split * rows_per_wg, rend = min(L, rbeg + rows_per_wg);
  const uint8_t *mp = mask_t + (size_t)c * L;
  const float *xc = xt + (size_t)c * L * PS + NG * n;
  const int rlane = 4 * m + q;   // this lane's row inside a 16-row tile

  double acc[C4_NTRI];
#pragma unroll
  for (int t = 0; t < C4_NTRI; ++t) acc[t] = 0.0;

  // Row tiles are fetched DEPTH tiles ahead: one tile is only ~1.2 us of MFMA work, less than the HBM latency
  // under load, and a wave has nothing else to hide it behind (one wave per SIMD).  (DEPTH 3 does not fit: the
  // allocator then has to move pinned accumulators around the asm MFMAs, which no hazard recogniser sees.)
  constexpr int DEPTH = 2;
  float xraw[DEPTH][NG];
  uint8_t mk[DEPTH];
  __shared__ double zeros[NG];
  if (tid < NG) zeros[tid] = 0.0;
  // nothing in a fetch depends on loaded data (the validity byte travels with the row and is looked at when the
  // tile is used): a dependent address would make every fetch wait for all the tiles in flight
  auto fetch = [&](int r0, auto sc) {
    constexpr int sl = decltype(sc)::value;
    const int row = r0 + rlane;
    const int rr = row < rend ? row : rbeg;
    mk[sl] = mp[rr];
    const float *xp = xc + (size_t)rr * PS;
#pragma unroll
    for (int s = 0; s < NG; s += 2) sf_load2(xp + s, xraw[sl][s], xraw[sl][s + 1]);
  };
  unsigned colm[3];   // lane n = 3 holds bands 54..71: those beyond the window are switched off
#pragma unroll
  for (int i = 0; i < 3; ++i) colm[i] = (NG * n + (NG - 3) + i < p) ? 0xffffffffu : 0u;
  int r0 = rbeg + 16 * wave;
  static_for<0, DEPTH>([&](auto sc) { fetch(r0 + 64 * decltype(sc)::value, sc); });
  __syncthreads();   // zeros[]
  for (; r0 < rend; r0 += 64 * DEPTH) {
    static_for<0, DEPTH>([&](auto sc) {
      constexpr int sl = decltype(sc)::value;
      const int rt = r0 + 64 * sl;
      {   // no branch in here: tiles past the end are clamped reads of zero weight (the split is a multiple of 64 DEPTH rows)
        const bool ok = (rt + rlane < rend) && mk[sl] != 0;
        int opq = 0;
        asm volatile("" : "+v"(opq));   // keep the 18 mean reads inside the iteration (see k_sweep)
        // invalid row: raw bits -> 0 and mean -> 0, so the operand is exactly 0 whatever the row held.  The raw bits are
        // switched off with ONE v_and_b32 per value before the conversion (written as a select the compiler converts first
        // and selects the two halves of the double: 38 v_cndmask per tile; every VALU instruction of this single wave per
        // SIMD is ~12 cycles the matrix pipe stands still, tools/microbench/mix4w.hip)
        const double *musl = (ok ? mus + NG * n : zeros) + opq;
        unsigned okm = ok ? 0xffffffffu : 0u;
        asm volatile("" : "+v"(okm));   // opaque: otherwise the and is turned back into a select of the CONVERTED value
        double f[NG];
#pragma unroll
        for (int I = 0; I < NG; ++I) {
          unsigned msk = okm;
          if (4 * NG - 3 <= 3 * NG + I) msk &= colm[I - (NG - 3)];   // only bands 69..71 can lie beyond a window of >= 69
          f[I] = (double)__uint_as_float(__float_as_uint(xraw[sl][I]) & msk) - musl[I];
        }
        if (EXP != 1) fetch(rt + 64 * DEPTH, sc);
        // the asm MFMAs below are invisible to the hazard recogniser: a VALU result (the last v_add_f64 of the centring)
        // needs wait states before an MFMA may read it as SrcA/B; nothing pads them when the scheduler puts the two
        // back to back (round 3: with the shorter prologue it did, and tile (0, 0) of every 16 rows came out wrong)
        static_assert(NG == 18, "operand list of the hazard pad");
        asm volatile("s_nop 4" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]),
                     "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]), "+v"(f[12]), "+v"(f[13]), "+v"(f[14]), "+v"(f[15]),
                     "+v"(f[16]), "+v"(f[17]));   // (the operands pin it between the last conversion and the first MFMA)
        if constexpr (EXP == 2) {
#pragma unroll
          for (int I = 0; I < NG; ++I) acc[I] += f[I];
        } else
        static_for<0, NG>([&](auto ic) {
          constexpr int I = decltype(ic)::value;
          static_for<I, NG>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            mfma_acc<tri_index(I, J)>(acc[tri_index(I, J)], f[I], f[J]);
          });
        });
      }
    });
  }
  // ---- sum the 4 blocks (row groups) of every tile, then the 4 waves
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // the asm MFMAs are invisible to the hazard recogniser: let the last ones retire
#pragma unroll
  for (int t = 0; t < C4_NTRI; ++t) {
    double v = acc[t];
    v += dpp_row<0x124>(v);   // row_ror:4
    v += dpp_row<0x128>(v);   // row_ror:8
    if (m == 0) red[wave][t][4 * q + n] = v;
  }
  __syncthreads();
  double *po = part + ((size_t)c * gridDim.y + split) * (C4_NTRI * 16);
  for (int i = tid; i < C4_NTRI * 16; i += 256) {
    const int t = i >> 4, e = i & 15;
    po[i] = (red[0][t][e] + red[1][t][e]) + (red[2][t][e] + red[3][t][e]);
  }
}

// k_syrk4d: the same product with TWO waves per SIMD in one workgroup.  Waves w and w+4 (the same SIMD) take the same
// 16 rows but one half each of the 171 tiles (86 / 85): 172 accumulator registers fit the 256 a wave may have at
// two waves per SIMD, so nothing is pinned and nothing lives in AGPRs, and one wave's loads, mean reads and
// conversions run under the other's MFMAs.  Price: both waves convert the rows (54 VALU per 86 MFMAs).
constexpr int C4_HALF = (C4_NTRI + 1) / 2;   // 86

template <int H>
__device__ __forceinline__ void syrk4d_tile(double (&acc)[C4_HALF], const double (&f)[C4_NG]) {
  static_for<0, C4_NG>([&](auto ic) {
    constexpr int I = decltype(ic)::value;
    static_for<I, C4_NG>([&](auto jc) {
      constexpr int J = decltype(jc)::value;
      constexpr int t = tri_index(I, J);
      if constexpr ((t < C4_HALF) == (H == 0))
        acc[t - H * C4_HALF] = __builtin_amdgcn_mfma_f64_4x4x4f64(f[I], f[J], acc[t - H * C4_HALF], 0, 0, 0);
    });
  });
}

__global__ __launch_bounds__(512, 1) void k_syrk4d(const float *__restrict__ xt, const uint8_t *__restrict__ mask_t,
                                                    const double *__restrict__ mu, int L, int p, int rows_per_wg,
                                                    double *__restrict__ part) {
  constexpr int NG = C4_NG, PS = 4 * C4_NG;
  __shared__ double red[4][C4_NTRI][16];
  __shared__ double mus[PS];
  __shared__ double zeros[NG];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wave = wv & 3, half = wv >> 2;
  const int q = lane >> 4, m = (lane >> 2) & 3, n = lane & 3;
  const int c = blockIdx.x, split = blockIdx.y;
  for (int i = tid; i < PS; i += 512) mus[i] = (i < p) ? mu[(size_t)c * p + i] : 0.0;
  if (tid < NG) zeros[tid] = 0.0;
  __syncthreads();
  const int rbeg = split * rows_per_wg, rend = min(L, rbeg + rows_per_wg);
  const uint8_t *mp = mask_t + (size_t)c * L;
  const float *xc = xt + (size_t)c * L * PS + NG * n;
  const int rlane = 4 * m + q;
  double acc[C4_HALF];
#pragma unroll
  for (int t = 0; t < C4_HALF; ++t) acc[t] = 0.0;
  constexpr int DEPTH = 1;   // the SIMD's other wave covers the latency; a second tile in flight would spill
  float xraw[DEPTH][NG];
  uint8_t mk[DEPTH];
  auto fetch = [&](int r0, auto sc) {
    constexpr int sl = decltype(sc)::value;
    const int row = r0 + rlane;
    const int rr = row < rend ? row : rbeg;
    mk[sl] = mp[rr];
    const float *xp = xc + (size_t)rr * PS;
#pragma unroll
    for (int s = 0; s < NG; s += 2) sf_load2(xp + s, xraw[sl][s], xraw[sl][s + 1]);
  };
  int r0 = rbeg + 16 * wave;
  static_for<0, DEPTH>([&](auto sc) { fetch(r0 + 64 * decltype(sc)::value, sc); });
  for (; r0 < rend; r0 += 64 * DEPTH) {
    static_for<0, DEPTH>([&](auto sc) {
      constexpr int sl = decltype(sc)::value;
      const int rt = r0 + 64 * sl;
      const bool ok = (rt + rlane < rend) && mk[sl] != 0;
      int opq = 0;
      asm volatile("" : "+v"(opq));
      const double *musl = (ok ? mus + NG * n : zeros) + opq;
      double f[NG];
#pragma unroll
      for (int I = 0; I < NG; ++I) {
        const float xv = (ok && NG * n + I < p) ? xraw[sl][I] : 0.0f;
        f[I] = (double)xv - musl[I];
      }
      fetch(rt + 64 * DEPTH, sc);
      if (half == 0) syrk4d_tile<0>(acc, f);   // wave-uniform
      else syrk4d_tile<1>(acc, f);
    });
  }
#pragma unroll
  for (int t = 0; t < C4_HALF; ++t) {
    double v = acc[t];
    v += dpp_row<0x124>(v);
    v += dpp_row<0x128>(v);
    const int tt = t + half * C4_HALF;
    if (m == 0 && tt < C4_NTRI) red[wave][tt][4 * q + n] = v;
  }
  __syncthreads();
  double *po = part + ((size_t)c * gridDim.y + split) * (C4_NTRI * 16);
  for (int i = tid; i < C4_NTRI * 16; i += 512) {
    const int t = i >> 4, e = i & 15;
    po[i] = (red[0][t][e] + red[1][t][e]) + (red[2][t][e] + red[3][t][e]);
  }
}

// part[c][split][tile][4 i + j] -> cov[c][band(I,i)][band(J,j)] (and its mirror), band(I, i) = 18 i + I
__global__ __launch_bounds__(256) void k_syrk4_reduce(const double *__restrict__ part, int nsplit,
                                                       const int32_t *__restrict__ nuse, int p, double *__restrict__ cov) {
  const int c = blockIdx.x;
  const double denom = (double)nuse[c] - 1.0;
  for (int idx = blockIdx.y * 256 + threadIdx.x; idx < C4_NTRI * 16; idx += 256 * gridDim.y) {
    const int t = idx >> 4, i = (idx >> 2) & 3, j = idx & 3;
    int I = 0, rem = t, rowlen = C4_NG;
    while (rem >= rowlen) { rem -= rowlen; ++I; --rowlen; }
    const int J = I + rem;
    double s = 0;
    for (int sp = 0; sp < nsplit; ++sp) s += part[((size_t)c * nsplit + sp) * (C4_NTRI * 16) + idx];
    s /= denom;
    const int a = C4_NG * i + I, b = C4_NG * j + J;
    if (a < p && b < p) {
      if (I != J) {
        cov[((size_t)c * p + a) * p + b] = s;
        cov[((size_t)c * p + b) * p + a] = s;
      } else if (i <= j) {   // diagonal tile: keep one of the two (identical) halves
        cov[((size_t)c * p + a) * p + b] = s;
        cov[((size_t)c * p + b) * p + a] = s;
      }
    }
  }
}

// ---- windows of 81..84 and 93..96 bands (NG = 21 / 24 band groups; the CO2 window is 309..391, p = 83: robust_mf.py:190-191)
// 231 (300) upper-triangular tiles are 462 (600) accumulator registers: more than a wave's 512.  A 16-row tile is therefore
// taken by TWO waves, each with half of the tiles (116 / 150 accumulators, pinned to the AGPR file as far as it reaches), on
// different SIMDs: waves (2 rg, 2 rg + 1) share row group rg of the workgroup's 32 rows per step.  Both convert the rows (3 NG
// vector instructions per NTRI / 2 MFMAs instead of per NTRI); everything else -- operand layout band(I, n) = NG n + I, the
// depth-2 row prefetch whose addresses do not depend on loaded data, exact zeros for invalid rows and for bands beyond the
// window (only the last three of lane group n = 3 can be: p >= 4 NG - 3), the split-wise partial sums -- is k_syrk4's.
template <int NG>
struct Cov4H {
  static constexpr int NTRI = NG * (NG + 1) / 2, HALF = (NTRI + 1) / 2;
  static constexpr int NACC_A = HALF < 120 ? HALF : 120;           // accumulators in AGPRs; the rest in VGPRs
  static constexpr int tri(int I, int J) { return I * NG - I * (I - 1) / 2 + (J - I); }
};
template <int T, int NA>
__device__ __forceinline__ void mfma_acc_h(double &acc, double a, double b) {
  if constexpr (T < NA) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
template <int I, int N>
__device__ __forceinline__ void tie_operands(double (&f)[N]) {   // one empty volatile asm per operand: it is computed before what follows
  if constexpr (I < N) {
    asm volatile("" : "+v"(f[I]));
    tie_operands<I + 1, N>(f);
  }
}
template <int NG, int H>
__device__ __forceinline__ void syrk4h_tile(double (&acc)[Cov4H<NG>::HALF], const double (&f)[NG]) {
  using C = Cov4H<NG>;
  static_for<0, NG>([&](auto ic) {
    constexpr int I = decltype(ic)::value;
    static_for<I, NG>([&](auto jc) {
      constexpr int J = decltype(jc)::value;
      constexpr int t = C::tri(I, J);
      if constexpr ((t < C::HALF) == (H == 0)) mfma_acc_h<t - H * C::HALF, C::NACC_A>(acc[t - H * C::HALF], f[I], f[J]);
    });
  });
}

template <int NG>
__global__ __launch_bounds__(256, 1) void k_syrk4h(const float *__restrict__ xt, const uint8_t *__restrict__ mask_t,
                                                    const double *__restrict__ mu, int L, int p, int rows_per_wg,
                                                    double *__restrict__ part) {
  using C = Cov4H<NG>;
  constexpr int PS = 4 * NG, NTRI = C::NTRI, HALF = C::HALF;
  extern __shared__ __attribute__((aligned(16))) double c4h_lds[];
  double *red = c4h_lds;                       // [2 row groups][NTRI][16]: the waves' tiles before the final sum
  double *mus = red + 2 * NTRI * 16;           // [PS]
  double *zeros = mus + PS;                    // [NG]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = wave & 1, rg = wave >> 1;
  const int q = lane >> 4, m = (lane >> 2) & 3, n = lane & 3;
  const int c = blockIdx.x, split = blockIdx.y;
  for (int i = tid; i < PS; i += 256) mus[i] = (i < p) ? mu[(size_t)c * p + i] : 0.0;
  if (tid < NG) zeros[tid] = 0.0;
  __syncthreads();
  const int rbeg = split * rows_per_wg, rend = min(L, rbeg + rows_per_wg);
  const uint8_t *mp = mask_t + (size_t)c * L;
  const float *xc = xt + (size_t)c * L * PS + NG * n;
  const int rlane = 4 * m + q;   // this lane's row inside a 16-row tile

  double acc[HALF];
#pragma unroll
  for (int t = 0; t < HALF; ++t) acc[t] = 0.0;

  constexpr int DEPTH = 2;
  float xraw[DEPTH][NG];
  uint8_t mk[DEPTH];
  auto fetch = [&](int r0, auto sc) {
    constexpr int sl = decltype(sc)::value;
    const int row = r0 + rlane;
    const int rr = row < rend ? row : rbeg;
    mk[sl] = mp[rr];
    const float *xp = xc + (size_t)rr * PS;
#pragma unroll
    for (int s = 0; s + 1 < NG; s += 2) sf_load2(xp + s, xraw[sl][s], xraw[sl][s + 1]);
    if constexpr (NG & 1) xraw[sl][NG - 1] = xp[NG - 1];
  };
  unsigned colm[3];   // lane n = 3 holds bands 3 NG .. 4 NG - 1: those beyond the window are switched off
#pragma unroll
  for (int i = 0; i < 3; ++i) colm[i] = (NG * n + (NG - 3) + i < p) ? 0xffffffffu : 0u;
  int r0 = rbeg + 16 * rg;
  static_for<0, DEPTH>([&](auto sc) { fetch(r0 + 32 * decltype(sc)::value, sc); });
  for (; r0 < rend; r0 += 32 * DEPTH) {
    static_for<0, DEPTH>([&](auto sc) {
      constexpr int sl = decltype(sc)::value;
      const int rt = r0 + 32 * sl;
      const bool ok = (rt + rlane < rend) && mk[sl] != 0;
      int opq = 0;
      asm volatile("" : "+v"(opq));   // keep the mean reads inside the iteration (see k_sweep)
      const double *musl = (ok ? mus + NG * n : zeros) + opq;
      unsigned okm = ok ? 0xffffffffu : 0u;
      asm volatile("" : "+v"(okm));   // opaque: otherwise the and is turned back into a select of the CONVERTED value
      double f[NG];
#pragma unroll
      for (int I = 0; I < NG; ++I) {
        unsigned msk = okm;
        if (I >= NG - 3) msk &= colm[I - (NG - 3)];
        f[I] = (double)__uint_as_float(__float_as_uint(xraw[sl][I]) & msk) - musl[I];
      }
      fetch(rt + 32 * DEPTH, sc);
      // the asm MFMAs are invisible to the hazard recogniser: every operand is tied down first, then the wait states a VALU
      // result needs before an MFMA may read it (volatile asm statements keep their order)
      tie_operands<0, NG>(f);
      asm volatile("s_nop 4");
      if (half == 0) syrk4h_tile<NG, 0>(acc, f);   // (wave-uniform)
      else syrk4h_tile<NG, 1>(acc, f);
    });
  }
  // ---- sum the 4 blocks (row groups of the MFMA) of every tile, then the two row groups of the workgroup
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // let the last asm MFMAs retire
#pragma unroll
  for (int t = 0; t < HALF; ++t) {
    double v = acc[t];
    v += dpp_row<0x124>(v);   // row_ror:4
    v += dpp_row<0x128>(v);   // row_ror:8
    const int tt = t + half * HALF;
    if (m == 0 && tt < NTRI) red[(rg * NTRI + tt) * 16 + 4 * q + n] = v;
  }
  __syncthreads();
  double *po = part + ((size_t)c * gridDim.y + split) * (NTRI * 16);
  for (int i = tid; i < NTRI * 16; i += 256) po[i] = red[i] + red[NTRI * 16 + i];
}

// part[c][split][tile][4 i + j] -> cov[c][band(I,i)][band(J,j)] (and its mirror), band(I, i) = NG i + I
template <int NG>
__global__ __launch_bounds__(256) void k_syrk4h_reduce(const double *__restrict__ part, int nsplit,
                                                        const int32_t *__restrict__ nuse, int p, double *__restrict__ cov) {
  constexpr int NTRI = NG * (NG + 1) / 2;
  const int c = blockIdx.x;
  const double denom = (double)nuse[c] - 1.0;
  for (int idx = blockIdx.y * 256 + threadIdx.x; idx < NTRI * 16; idx += 256 * gridDim.y) {
    const int t = idx >> 4, i = (idx >> 2) & 3, j = idx & 3;
    int I = 0, rem = t, rowlen = NG;
    while (rem >= rowlen) { rem -= rowlen; ++I; --rowlen; }
    const int J = I + rem;
    double s = 0;
    for (int sp = 0; sp < nsplit; ++sp) s += part[((size_t)c * nsplit + sp) * (NTRI * 16) + idx];
    s /= denom;
    const int a = NG * i + I, b = NG * j + J;
    if (a < p && b < p && (I != J || i <= j)) {
      cov[((size_t)c * p + a) * p + b] = s;
      cov[((size_t)c * p + b) * p + a] = s;
    }
  }
}

template <int NG>
int launch_cov4h(const float *xt, const uint8_t *mask_t, const int32_t *nuse, const double *mu, const SfGeom &g, double *cov,
                 void *scratch, hipStream_t st) {
  using C = Cov4H<NG>;
  const int nsplit = sf_sweep_splits(g.lines, g.ncols);
  int rows = sf_cdiv(g.lines, nsplit);
  rows = (rows + 63) / 64 * 64;   // whole prefetch rings: 2 row groups x 16 rows x depth 2
  double *part = reinterpret_cast<double *>(scratch);
  const size_t lds = ((size_t)2 * C::NTRI * 16 + 4 * NG + NG) * sizeof(double);
  if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_syrk4h<NG>), lds)) return rc;
  hipLaunchKernelGGL(k_syrk4h<NG>, dim3(g.ncols, nsplit), dim3(256), lds, st, xt, mask_t, mu, g.lines, g.p, rows, part);
  SF_LAUNCH_CHECK("k_syrk4h");
  hipLaunchKernelGGL(k_syrk4h_reduce<NG>, dim3(g.ncols, 4), dim3(256), 0, st, part, nsplit, nuse, g.p, cov);
  SF_LAUNCH_CHECK("k_syrk4h_reduce");
  return 0;
}

}  // namespace

size_t sf_cov4_scratch_bytes(const SfGeom &g) {
  const int ng = sf_sw4_groups(g.p) ? sf_sw4_groups(g.p) : C4_NG;
  return sf_align((size_t)g.ncols * sf_sweep_splits(g.lines, g.ncols) * (ng * (ng + 1) / 2) * 16 * sizeof(double));
}

int sf_launch_cov4(const float *xt, const uint8_t *mask_t, const int32_t *nuse, const double *mu, const SfGeom &g,
                   double *cov, void *scratch, hipStream_t st) {
  if (g.s4 == 21) return launch_cov4h<21>(xt, mask_t, nuse, mu, g, cov, scratch, st);
  if (g.s4 == 24) return launch_cov4h<24>(xt, mask_t, nuse, mu, g, cov, scratch, st);
  const int nsplit = sf_sweep_splits(g.lines, g.ncols);
  int rows = sf_cdiv(g.lines, nsplit);
  rows = (rows + 127) / 128 * 128;   // whole prefetch rings: 4 waves x 16 rows x depth 2
  double *part = reinterpret_cast<double *>(scratch);
#ifdef SF_SWEEP_EXPERIMENTS
  if (sf_tune().cov_variant == 11)
    hipLaunchKernelGGL(k_syrk4<1>, dim3(g.ncols, nsplit), dim3(256), 0, st, xt, mask_t, mu, g.lines, g.p, rows, part);
  else if (sf_tune().cov_variant == 12)
    hipLaunchKernelGGL(k_syrk4<2>, dim3(g.ncols, nsplit), dim3(256), 0, st, xt, mask_t, mu, g.lines, g.p, rows, part);
  else
#endif
  if (sf_tune().cov_variant == 3) {   // two waves per SIMD, each half of the tiles: measured slower (1.62 vs 1.45 ms), kept as an option
    hipLaunchKernelGGL(k_syrk4d, dim3(g.ncols, nsplit), dim3(512), 0, st, xt, mask_t, mu, g.lines, g.p, rows, part);
  } else
  hipLaunchKernelGGL(k_syrk4<0>, dim3(g.ncols, nsplit), dim3(256), 0, st, xt, mask_t, mu, g.lines, g.p, rows, part);
  SF_LAUNCH_CHECK("k_syrk4");
  hipLaunchKernelGGL(k_syrk4_reduce, dim3(g.ncols, 4), dim3(256), 0, st, part, nsplit, nuse, g.p, cov);
  SF_LAUNCH_CHECK("k_syrk4_reduce");
  return 0;
}
