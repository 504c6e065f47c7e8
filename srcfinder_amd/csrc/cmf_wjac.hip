// Wide-window eigensolver, round 4: the blocked one-sided Jacobi with its rotations on the matrix cores.
//
// cmf_wide.hip (k_blockjac / k_blockjac_x) visits a pair of 16-column blocks of the factor G (R = G G^T, p2 x p2, column
// major) per workgroup and applies its 256 plane rotations one by one to 426-row columns: 16 dependent steps of
// dot product -> 16-lane reduction -> rotation parameters -> column update, ~2 us each, 35 us a visit, 7000 launches and
// 0.28 s per 598-column flightline at p = 425 -- a sixteenth of the fp64 vector rate, bound by the latency of that chain.
//
// Here a visit does the SAME rotations in the SAME order, but on the 32 x 32 Gram matrix of the two blocks:
//   1. A = [G_a | G_b] (p2 x 32) -> LDS;
//   2. M = A^T A on v_mfma_f64_4x4x4_f64 (36 upper-triangular 4x4 tiles per wave, K split over the four waves);
//   3. the step's rotations (15 steps of 8 + 8 pairs inside the blocks in the first launch of a sweep, then 16 steps
//      of 16 cross pairs (a_i, b_(i+t) mod 16)) as TWO-sided rotations of M in LDS, M <- J^T M J, accumulated in
//      Q <- Q J: 256 threads, one 2x2 block of M each, 32-element rows instead of 426-element columns;
//   4. A <- A Q on the MFMA (27 row groups x 8 column groups x K = 32), back to global memory.
// A rotation is decided by the same test on the same quantities as before (|a.b| > tol |a||b|, with a.b, |a|^2, |b|^2
// now read from M, which the earlier rotations of the visit have updated by the rotation formula instead of by fresh dot
// products), so the sweep count and the converged state are those of the scalar kernels; a visit whose Gram matrix is
// already diagonal to the tolerance ends after step 2 without writing anything.
#include "cmf_common.h"

namespace {

constexpr int JM_B = 16;         // columns per block
constexpr int JM_P = 2 * JM_B;   // columns per visit
constexpr int JM_NT = 256;
constexpr int JM_LDM = JM_P + 1; // row stride of M and Q in LDS (doubles)

__device__ __forceinline__ void jm_pair(int s, int k, int m, int &a, int &b) {   // circle method, m + 1 players
  int x = s + k;
  x = x >= m ? x - m : x;
  int y = s - k;
  y = y < 0 ? y + m : y;
  a = x;
  b = (k == 0) ? m : y;
}
__device__ __forceinline__ double jm_rsqrt(double x) {   // hardware estimate + two Newton steps
  double y = __builtin_amdgcn_rsq(x);
  y = y * __builtin_fma(-0.5 * x * y, y, 1.5);
  y = y * __builtin_fma(-0.5 * x * y, y, 1.5);
  return y;
}
// the rotation of cmf_wide.hip (bj_rotation): columns a, b with squared norms aa, bb and inner product ab;
// a' = cs a - sn b, b' = sn a + cs b are orthogonal.  false (cs = 1, sn = 0): nothing to do.
__device__ __forceinline__ bool jm_rotation(double aa, double bb, double ab, double tol2, double &cs, double &sn) {
  cs = 1.0;
  sn = 0.0;
  const double ab2 = aa * bb;
  if (!(ab2 > 0.0 && ab * ab > tol2 * ab2)) return false;
  const double tau = bb - aa, gam = 2.0 * ab;
  const double rinv = jm_rsqrt(__builtin_fma(tau, tau, gam * gam));
  const double c2 = fabs(tau) * rinv;              // |cos 2 theta|
  const double h = __builtin_fma(0.5, c2, 0.5);    // cos^2 theta in [0.5, 1]
  const double rh = jm_rsqrt(h);
  cs = h * rh;
  sn = fabs(gam) * rinv * 0.5 * rh;
  sn = ((tau < 0.0) != (gam < 0.0)) ? -sn : sn;
  return true;
}
template <int CTRL>
__device__ __forceinline__ double jm_dpp_row(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// pair k (0..15) of rotation step t of a visit.  INNER: steps 0..14 are the pairs inside each block (8 + 8), steps
// 15..30 the cross pairs; otherwise steps 0..15 are the cross pairs.  Columns 0..15 = block a, 16..31 = block b.
template <bool INNER>
__device__ __forceinline__ void jm_step_pair(int t, int k, int &a, int &b) {
  if (INNER && t < JM_B - 1) {
    const int blk = k >> 3;
    jm_pair(t, k & 7, JM_B - 1, a, b);
    a += JM_B * blk;
    b += JM_B * blk;
  } else {
    const int tt = INNER ? t - (JM_B - 1) : t;
    a = k;
    b = JM_B + ((k + tt) & (JM_B - 1));
  }
}

// phase clocks of the visits (sf_debug_set(22, 1), tools/ab_wjac.py --stamps): s_memtime ticks summed over all workgroups
// [0] visits, [1] load, [2] Gram + reduction, [3] rotations, [4] update, [5] store, [6] visits that ended after the Gram test
__device__ unsigned long long g_jm_stamps[8];

// One visit per workgroup: blockIdx.x = pair slot of the step (circle method over mblk block slots), blockIdx.y = matrix.
// LDS: A [32][LDr] | M0 [32][33] | { M1 [32][33] | Q [32][33] } aliased with part [4][36][16], the Gram partials of the
// four waves (18 KB, dead before M1 / Q come to life): 148.3 KB at p2 = 426, 158.3 KB at p2 = 512.
template <bool INNER>
__global__ __launch_bounds__(JM_NT) void k_bjm(double *__restrict__ gscratch, int p2, int R16, int LDr, int nblk, int mblk,
                                                int step, const int32_t *__restrict__ cflag,
                                                const int32_t *__restrict__ done, int32_t *__restrict__ rot, int stamp) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0, tk4 = 0;
  if (stamp) tk0 = __builtin_readcyclecounter();
  double *M0 = sm + (size_t)JM_P * LDr;
  double *M1 = M0 + JM_P * JM_LDM;
  double *Q = M1 + JM_P * JM_LDM;
  double *part = M1;
  __shared__ int flags[2];   // [0]: some pair of the visit needs a rotation, [1]: a rotation was applied
  const int mtx = blockIdx.y;
  if (cflag[mtx] != 0 || done[mtx]) return;
  int ba, bb;
  jm_pair(step, blockIdx.x, mblk - 1, ba, bb);
  const bool has_a = ba < nblk, has_b = bb < nblk;
  if (!has_a && !has_b) return;
  if (!has_a) { ba = bb; }
  const bool lone = !(has_a && has_b);       // a block without partner: only its inner pairs (first launch of a sweep)
  if (lone && !INNER) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, m = (lane >> 2) & 3, n = lane & 3;
  double *G = gscratch + (size_t)mtx * 2 * p2 * p2;
  const double tol = (double)p2 * 2.220446049250313e-16, tol2 = tol * tol;
  if (tid < 2) flags[tid] = 0;

  // ---- 1. load: columns [ba*16, +16) -> A[0..15], [bb*16, +16) -> A[16..31]; rows >= p2 and columns >= p2 are zero
  // (all loads of both blocks are issued before the first LDS store: one memory round trip per visit instead of four --
  //  a workgroup is alone on its CU and has nothing else to hide the latency behind)
  constexpr int JM_U = 16;
  const int half = p2 >> 1, halfR = R16 >> 1;
  const int c0a = ba * JM_B, c0b = bb * JM_B;
  const int ncva = min(JM_B, p2 - c0a), ncvb = lone ? 0 : min(JM_B, p2 - c0b);
  const double2 *srca = reinterpret_cast<const double2 *>(G + (size_t)c0a * p2);
  const double2 *srcb = reinterpret_cast<const double2 *>(G + (size_t)c0b * p2);
  for (int base = tid; base < JM_B * halfR; base += JM_NT * JM_U) {
    double2 va[JM_U], vb[JM_U];
#pragma unroll
    for (int u = 0; u < JM_U; ++u) {
      const int idx = base + JM_NT * u;
      const int cc = idx / halfR, r2 = idx - cc * halfR;
      const bool in = idx < JM_B * halfR && r2 < half;
      va[u] = (in && cc < ncva) ? srca[(size_t)cc * half + r2] : make_double2(0.0, 0.0);
      vb[u] = (in && cc < ncvb) ? srcb[(size_t)cc * half + r2] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int u = 0; u < JM_U; ++u) {
      const int idx = base + JM_NT * u;
      if (idx < JM_B * halfR) {
        const int cc = idx / halfR, r2 = idx - cc * halfR;
        *reinterpret_cast<double2 *>(sm + (size_t)cc * LDr + 2 * r2) = va[u];
        *reinterpret_cast<double2 *>(sm + (size_t)(JM_B + cc) * LDr + 2 * r2) = vb[u];
      }
    }
  }
  __syncthreads();
  if (stamp) tk1 = __builtin_readcyclecounter();

  // ---- 2. M = A^T A.  f[I] at lane (q, m, n) = A[row 16 ks + 4 m + q][col 4 I + n] is the A operand of tile row I and
  //         the B operand of tile column J (cmf_cov4.hip); the four blocks of the instruction are four row quads of the
  //         16-row step, summed at the end; the 16-row steps are dealt round-robin to the waves.
  {
    double acc[36];
#pragma unroll
    for (int t = 0; t < 36; ++t) acc[t] = 0.0;
    const int nks = R16 >> 4;
    const double *ap = sm + (size_t)n * LDr + 4 * m + q;
    for (int ks = wave; ks < nks; ks += 4) {
      double f[8];
#pragma unroll
      for (int I = 0; I < 8; ++I) f[I] = ap[(size_t)4 * I * LDr + 16 * ks];
      int t = 0;
#pragma unroll
      for (int I = 0; I < 8; ++I)
#pragma unroll
        for (int J = I; J < 8; ++J) {
          acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(f[I], f[J], acc[t], 0, 0, 0);
          ++t;
        }
    }
#pragma unroll
    for (int t = 0; t < 36; ++t) {
      double v = acc[t];
      v += jm_dpp_row<0x124>(v);   // row_ror:4
      v += jm_dpp_row<0x128>(v);   // row_ror:8
      if (m == 0) part[((size_t)wave * 36 + t) * 16 + 4 * q + n] = v;   // D[i][j] at (q = i, n = j)
    }
  }
  __syncthreads();
  for (int e = tid; e < 36 * 16; e += JM_NT) {
    const int t = e >> 4, i = (e >> 2) & 3, j = e & 3;
    int I = 0, rem = t, rowlen = 8;
    while (rem >= rowlen) { rem -= rowlen; ++I; --rowlen; }
    const int J = I + rem;
    const double v = (part[(size_t)t * 16 + (e & 15)] + part[((size_t)36 + t) * 16 + (e & 15)]) +
                     (part[((size_t)72 + t) * 16 + (e & 15)] + part[((size_t)108 + t) * 16 + (e & 15)]);
    const int r = 4 * I + i, c = 4 * J + j;
    if (I != J || i <= j) {
      M0[r * JM_LDM + c] = v;
      M0[c * JM_LDM + r] = v;
    }
  }
  __syncthreads();   // part is dead: its bytes become M1 and Q
  for (int e = tid; e < JM_P * JM_P; e += JM_NT) {
    const int r = e >> 5, c = e & 31;
    Q[r * JM_LDM + c] = (r == c) ? 1.0 : 0.0;
  }
  __syncthreads();
  // ---- is there anything to rotate?  (the pairs this visit is responsible for: cross pairs, and the pairs inside the
  //      blocks in the first launch of a sweep)
  {
    const int r = tid >> 4, c = tid & 15;
    bool need = false;
    {
      const double aa = M0[r * JM_LDM + r], bbn = M0[(JM_B + c) * JM_LDM + JM_B + c], ab = M0[r * JM_LDM + JM_B + c];
      const double ab2 = aa * bbn;
      need = need || (ab2 > 0.0 && ab * ab > tol2 * ab2);
    }
    if (INNER && r != c) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const int a = JM_B * blk + r, b = JM_B * blk + c;
        const double aa = M0[a * JM_LDM + a], bbn = M0[b * JM_LDM + b], ab = M0[a * JM_LDM + b];
        const double ab2 = aa * bbn;
        need = need || (ab2 > 0.0 && ab * ab > tol2 * ab2);
      }
    }
    if (need) flags[0] = 1;
  }
  __syncthreads();
  if (stamp) tk2 = __builtin_readcyclecounter();
  if (!flags[0]) {         // uniform
    if (stamp && tid == 0) {
      atomicAdd(&g_jm_stamps[0], 1ull);
      atomicAdd(&g_jm_stamps[6], 1ull);
      atomicAdd(&g_jm_stamps[1], tk1 - tk0);
      atomicAdd(&g_jm_stamps[2], tk2 - tk1);
    }
    return;
  }

  // ---- 3. the visit's rotations on M (ping-pong between M0 and M1: one barrier per step), accumulated in Q.
  //         thread (ka, kb): the 2x2 block of M at rows pair ka, columns pair kb, and rows 2 kb, 2 kb + 1 of Q at the
  //         columns of pair ka.  Both rotations are recomputed by every thread that needs them (no broadcast step).
  {
    const int ka = tid >> 4, kb = tid & 15;
    constexpr int NSTEP = INNER ? (JM_B - 1) + JM_B : JM_B;
    const double *Mc = M0;
    double *Mn = M1;
    bool rotated = false;
    for (int t = 0; t < NSTEP; ++t) {
      int a1, b1, a2, b2;
      jm_step_pair<INNER>(t, ka, a1, b1);
      jm_step_pair<INNER>(t, kb, a2, b2);
      const double aa1 = Mc[a1 * JM_LDM + a1], bb1 = Mc[b1 * JM_LDM + b1], ab1 = Mc[a1 * JM_LDM + b1];
      const double aa2 = Mc[a2 * JM_LDM + a2], bb2 = Mc[b2 * JM_LDM + b2], ab2 = Mc[a2 * JM_LDM + b2];
      const double x00 = Mc[a1 * JM_LDM + a2], x01 = Mc[a1 * JM_LDM + b2];
      const double x10 = Mc[b1 * JM_LDM + a2], x11 = Mc[b1 * JM_LDM + b2];
      double c1, s1, c2, s2;
      const bool r1 = jm_rotation(aa1, bb1, ab1, tol2, c1, s1);
      const bool r2 = jm_rotation(aa2, bb2, ab2, tol2, c2, s2);
      // Y = X J2, Z = J1^T Y with J = [[c, s], [-s, c]]
      const double y00 = c2 * x00 - s2 * x01, y01 = s2 * x00 + c2 * x01;
      const double y10 = c2 * x10 - s2 * x11, y11 = s2 * x10 + c2 * x11;
      double z00 = c1 * y00 - s1 * y10, z01 = c1 * y01 - s1 * y11;
      double z10 = s1 * y00 + c1 * y10, z11 = s1 * y01 + c1 * y11;
      if (ka == kb && r1) {   // the rotated pair itself: orthogonal by construction
        z01 = 0.0;
        z10 = 0.0;
        rotated = true;
      }
      Mn[a1 * JM_LDM + a2] = z00;
      Mn[a1 * JM_LDM + b2] = z01;
      Mn[b1 * JM_LDM + a2] = z10;
      Mn[b1 * JM_LDM + b2] = z11;
      if (r1) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int row = 2 * kb + rr;
          const double u = Q[row * JM_LDM + a1], v = Q[row * JM_LDM + b1];
          Q[row * JM_LDM + a1] = c1 * u - s1 * v;
          Q[row * JM_LDM + b1] = s1 * u + c1 * v;
        }
      }
      (void)r2;
      __syncthreads();
      const double *tmp = Mc;
      Mc = Mn;
      Mn = const_cast<double *>(tmp);
    }
    if (rotated) flags[1] = 1;
  }
  __syncthreads();
  if (stamp) tk3 = __builtin_readcyclecounter();
  if (!flags[1]) return;   // (cannot happen after flags[0]; kept for safety: nothing to write)

  // ---- 4. A <- A Q on the MFMA.  Blocks of the instruction = the four row quads of a 16-row group, K = 4 columns per
  //         instruction: a[kk] at lane (q, m, n) = A[row 16 I' + 4 m + n][col 4 kk + q], B operand Q[4 kk + q][4 J + n]
  //         (the same for every block), result at lane (q, m, n) = (A Q)[row 16 I' + 4 m + q][col 4 J + n].
  {
    double bq[8][8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
      for (int J = 0; J < 8; ++J) bq[kk][J] = Q[(4 * kk + q) * JM_LDM + 4 * J + n];
    const int ngrp = R16 >> 4;
    for (int Ig = wave; Ig < ngrp; Ig += 4) {
      double a[8];
      const double *ap = sm + (size_t)q * LDr + 16 * Ig + 4 * m + n;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) a[kk] = ap[(size_t)4 * kk * LDr];
      double acc[8];
#pragma unroll
      for (int J = 0; J < 8; ++J) acc[J] = 0.0;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int J = 0; J < 8; ++J) acc[J] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[kk], bq[kk][J], acc[J], 0, 0, 0);
      double *op = sm + (size_t)n * LDr + 16 * Ig + 4 * m + q;
#pragma unroll
      for (int J = 0; J < 8; ++J) op[(size_t)4 * J * LDr] = acc[J];
    }
  }
  __syncthreads();
  if (stamp) tk4 = __builtin_readcyclecounter();

  // ---- 5. store
  for (int cblk = 0; cblk < (lone ? 1 : 2); ++cblk) {
    const int c0 = (cblk == 0 ? ba : bb) * JM_B;
    const int n2 = min(JM_B, p2 - c0) * half;
    double2 *dst = reinterpret_cast<double2 *>(G + (size_t)c0 * p2);
#pragma unroll 4
    for (int idx = tid; idx < n2; idx += JM_NT) {
      const int cc = idx / half, r2 = idx - cc * half;
      dst[idx] = *reinterpret_cast<const double2 *>(sm + (size_t)(cblk * JM_B + cc) * LDr + 2 * r2);
    }
  }
  if (tid == 0) rot[mtx] = 1;
  if (stamp) {
    __syncthreads();
    if (tid == 0) {
      const unsigned long long tk5 = __builtin_readcyclecounter();
      atomicAdd(&g_jm_stamps[0], 1ull);
      atomicAdd(&g_jm_stamps[1], tk1 - tk0);
      atomicAdd(&g_jm_stamps[2], tk2 - tk1);
      atomicAdd(&g_jm_stamps[3], tk3 - tk2);
      atomicAdd(&g_jm_stamps[4], tk4 - tk3);
      atomicAdd(&g_jm_stamps[5], tk5 - tk4);
    }
  }
}

// ---- k_bjm8: the same visit with EIGHT waves (two per SIMD) -- what the phase clocks of k_bjm asked for (tools/ab_wjac.py:
// per visit 12.9 k cycles load, 12.0 k Gram, 25.5 k rotations, 11.8 k update, 5.8 k store with four waves):
//   * Gram: the two waves of a SIMD take the same 16-row steps but 18 of the 36 tiles each: half the DPP / store epilogue
//     per wave, and one wave's LDS reads run under the other's MFMAs;
//   * rotations: a lane PAIR owns a 2x2 block of M; lane h = 0 works out the rotation of the block's row pair, lane h = 1
//     that of its column pair, one DPP swap hands each the other's (cos, sin), and each lane forms one of the two rows:
//     ~70 fp64 instructions per lane and step instead of ~120 (the step is bound by the instruction stream of a wave);
//   * update: B operands held for half the columns at a time (fits 256 registers), 27 row groups over eight waves.
// Same arithmetic per matrix entry as k_bjm except the order of the four partial Gram sums (identical: same tree).
constexpr int JM8_NT = 512;
__device__ __forceinline__ double jm_swap1(double v) {   // value of the neighbouring lane (lane ^ 1)
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int H>
__device__ __forceinline__ void jm_gram_half(double (&acc)[18], const double (&f)[8]) {
  int t = 0;
#pragma unroll
  for (int I = 0; I < 8; ++I)
#pragma unroll
    for (int J = I; J < 8; ++J) {
      if ((t < 18) == (H == 0)) acc[t - 18 * H] = __builtin_amdgcn_mfma_f64_4x4x4f64(f[I], f[J], acc[t - 18 * H], 0, 0, 0);
      ++t;
    }
}
__device__ __forceinline__ void jm_tile_ij(int t, int &I, int &J) {   // tile t of the row-major upper-triangular list
  int rem = t, rowlen = 8;
  I = 0;
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const bool go = rem >= rowlen;
    rem -= go ? rowlen : 0;
    I += go ? 1 : 0;
    rowlen -= go ? 1 : 0;
  }
  J = I + rem;
}

template <bool INNER>
__global__ __launch_bounds__(JM8_NT) void k_bjm8(double *__restrict__ gscratch, int p2, int R16, int LDr, int nblk, int mblk,
                                                  int step, const int32_t *__restrict__ cflag,
                                                  const int32_t *__restrict__ done, int32_t *__restrict__ rot, int stamp) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0, tk4 = 0;
  if (stamp) tk0 = __builtin_readcyclecounter();
  double *M0 = sm + (size_t)JM_P * LDr;
  double *M1 = M0 + JM_P * JM_LDM;
  double *Q = M1 + JM_P * JM_LDM;
  double *part = M1;
  __shared__ int flags[2];
  const int mtx = blockIdx.y;
  if (cflag[mtx] != 0 || done[mtx]) return;
  int ba, bb;
  jm_pair(step, blockIdx.x, mblk - 1, ba, bb);
  const bool has_a = ba < nblk, has_b = bb < nblk;
  if (!has_a && !has_b) return;
  if (!has_a) { ba = bb; }
  const bool lone = !(has_a && has_b);
  if (lone && !INNER) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, m = (lane >> 2) & 3, n = lane & 3;
  double *G = gscratch + (size_t)mtx * 2 * p2 * p2;
  const double tol = (double)p2 * 2.220446049250313e-16, tol2 = tol * tol;
  if (tid < 2) flags[tid] = 0;

  // ---- 1. load (every load of both blocks in flight before the first LDS store)
  constexpr int JM_U = 8;
  const int half = p2 >> 1, halfR = R16 >> 1;
  {
    const int c0a = ba * JM_B, c0b = bb * JM_B;
    const int ncva = min(JM_B, p2 - c0a), ncvb = lone ? 0 : min(JM_B, p2 - c0b);
    const double2 *srca = reinterpret_cast<const double2 *>(G + (size_t)c0a * p2);
    const double2 *srcb = reinterpret_cast<const double2 *>(G + (size_t)c0b * p2);
    for (int base = tid; base < JM_B * halfR; base += JM8_NT * JM_U) {
      double2 va[JM_U], vb[JM_U];
#pragma unroll
      for (int u = 0; u < JM_U; ++u) {
        const int idx = base + JM8_NT * u;
        const int cc = idx / halfR, r2 = idx - cc * halfR;
        const bool in = idx < JM_B * halfR && r2 < half;
        va[u] = (in && cc < ncva) ? srca[(size_t)cc * half + r2] : make_double2(0.0, 0.0);
        vb[u] = (in && cc < ncvb) ? srcb[(size_t)cc * half + r2] : make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int u = 0; u < JM_U; ++u) {
        const int idx = base + JM8_NT * u;
        if (idx < JM_B * halfR) {
          const int cc = idx / halfR, r2 = idx - cc * halfR;
          *reinterpret_cast<double2 *>(sm + (size_t)cc * LDr + 2 * r2) = va[u];
          *reinterpret_cast<double2 *>(sm + (size_t)(JM_B + cc) * LDr + 2 * r2) = vb[u];
        }
      }
    }
  }
  __syncthreads();
  if (stamp) tk1 = __builtin_readcyclecounter();

  // ---- 2. M = A^T A: wave = (16-row-step group ksg, tile half th)
  {
    const int ksg = wave & 3, th = wave >> 2;
    double acc[18];
#pragma unroll
    for (int t = 0; t < 18; ++t) acc[t] = 0.0;
    const int nks = R16 >> 4;
    const double *ap = sm + (size_t)n * LDr + 4 * m + q;
    for (int ks = ksg; ks < nks; ks += 4) {
      double f[8];
#pragma unroll
      for (int I = 0; I < 8; ++I) f[I] = ap[(size_t)4 * I * LDr + 16 * ks];
      if (th == 0) jm_gram_half<0>(acc, f);   // wave-uniform
      else jm_gram_half<1>(acc, f);
    }
#pragma unroll
    for (int t = 0; t < 18; ++t) {
      double v = acc[t];
      v += jm_dpp_row<0x124>(v);   // row_ror:4
      v += jm_dpp_row<0x128>(v);   // row_ror:8
      if (m == 0) part[((size_t)ksg * 36 + 18 * th + t) * 16 + 4 * q + n] = v;
    }
  }
  __syncthreads();
  for (int e = tid; e < 36 * 16; e += JM8_NT) {
    const int t = e >> 4, i = (e >> 2) & 3, j = e & 3;
    int I, J;
    jm_tile_ij(t, I, J);
    const double v = (part[(size_t)t * 16 + (e & 15)] + part[((size_t)36 + t) * 16 + (e & 15)]) +
                     (part[((size_t)72 + t) * 16 + (e & 15)] + part[((size_t)108 + t) * 16 + (e & 15)]);
    const int r = 4 * I + i, c = 4 * J + j;
    if (I != J || i <= j) {
      M0[r * JM_LDM + c] = v;
      M0[c * JM_LDM + r] = v;
    }
  }
  __syncthreads();   // part is dead: its bytes become M1 and Q
  for (int e = tid; e < JM_P * JM_P; e += JM8_NT) {
    const int r = e >> 5, c = e & 31;
    Q[r * JM_LDM + c] = (r == c) ? 1.0 : 0.0;
  }
  if (tid < 256) {
    const int r = tid >> 4, c = tid & 15;
    bool need = false;
    {
      const double aa = M0[r * JM_LDM + r], bbn = M0[(JM_B + c) * JM_LDM + JM_B + c], ab = M0[r * JM_LDM + JM_B + c];
      const double ab2 = aa * bbn;
      need = need || (ab2 > 0.0 && ab * ab > tol2 * ab2);
    }
    if (INNER && r != c) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const int a = JM_B * blk + r, b = JM_B * blk + c;
        const double aa = M0[a * JM_LDM + a], bbn = M0[b * JM_LDM + b], ab = M0[a * JM_LDM + b];
        const double ab2 = aa * bbn;
        need = need || (ab2 > 0.0 && ab * ab > tol2 * ab2);
      }
    }
    if (need) flags[0] = 1;
  }
  __syncthreads();
  if (stamp) tk2 = __builtin_readcyclecounter();
  if (!flags[0]) {         // uniform
    if (stamp && tid == 0) {
      atomicAdd(&g_jm_stamps[0], 1ull);
      atomicAdd(&g_jm_stamps[6], 1ull);
      atomicAdd(&g_jm_stamps[1], tk1 - tk0);
      atomicAdd(&g_jm_stamps[2], tk2 - tk1);
    }
    return;
  }

  // ---- 3. rotations: lanes (2i, 2i+1) = (h = 0, 1) own the block at rows pair ka, columns pair kb
  {
    const int ka = tid >> 5, kb = (tid >> 1) & 15, h = tid & 1;
    constexpr int NSTEP = INNER ? (JM_B - 1) + JM_B : JM_B;
    const double *Mc = M0;
    double *Mn = M1;
    bool rotated = false;
    for (int t = 0; t < NSTEP; ++t) {
      int a1, b1, a2, b2;
      jm_step_pair<INNER>(t, ka, a1, b1);
      jm_step_pair<INNER>(t, kb, a2, b2);
      const int am = h ? a2 : a1, bm = h ? b2 : b1;          // the pair whose rotation this lane works out
      const double aam = Mc[am * JM_LDM + am], bbm = Mc[bm * JM_LDM + bm], abm = Mc[am * JM_LDM + bm];
      const double x00 = Mc[a1 * JM_LDM + a2], x01 = Mc[a1 * JM_LDM + b2];
      const double x10 = Mc[b1 * JM_LDM + a2], x11 = Mc[b1 * JM_LDM + b2];
      const int qrow = 2 * kb + h;
      const double qu = Q[qrow * JM_LDM + a1], qv = Q[qrow * JM_LDM + b1];
      double cm, snm;
      jm_rotation(aam, bbm, abm, tol2, cm, snm);
      const double co = jm_swap1(cm), so = jm_swap1(snm);
      const double c1 = h ? co : cm, s1 = h ? so : snm;        // rotation of the row pair (ka)
      const double c2 = h ? cm : co, s2 = h ? snm : so;        // rotation of the column pair (kb)
      const double y00 = c2 * x00 - s2 * x01, y01 = s2 * x00 + c2 * x01;
      const double y10 = c2 * x10 - s2 * x11, y11 = s2 * x10 + c2 * x11;
      // row a1 of J1^T Y (h = 0): c1 y0. - s1 y1. ; row b1 (h = 1): s1 y0. + c1 y1.
      const double pc = h ? s1 : c1, qc = h ? c1 : -s1;
      double z0 = pc * y00 + qc * y10, z1 = pc * y01 + qc * y11;
      const bool did = (s1 != 0.0) || (c1 != 1.0);
      if (ka == kb && did) {   // the rotated pair itself: orthogonal by construction
        if (h) z0 = 0.0; else z1 = 0.0;
        rotated = true;
      }
      const int zr = h ? b1 : a1;
      Mn[zr * JM_LDM + a2] = z0;
      Mn[zr * JM_LDM + b2] = z1;
      Q[qrow * JM_LDM + a1] = c1 * qu - s1 * qv;
      Q[qrow * JM_LDM + b1] = s1 * qu + c1 * qv;
      __syncthreads();
      const double *tmp = Mc;
      Mc = Mn;
      Mn = const_cast<double *>(tmp);
    }
    if (rotated) flags[1] = 1;
  }
  __syncthreads();
  if (stamp) tk3 = __builtin_readcyclecounter();
  if (!flags[1]) return;

  // ---- 4. A <- A Q: row groups dealt to the eight waves, the B operands (Q) held for four of the eight 4-column K steps at
  //         a time; a wave's row groups are written only after both halves have been accumulated (in place: no other wave
  //         reads or writes these rows).
  {
    const int ngrp = R16 >> 4;
    double acc[4][8];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int J = 0; J < 8; ++J) acc[g][J] = 0.0;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      double bq[4][8];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int J = 0; J < 8; ++J) bq[kk][J] = Q[(4 * (4 * kh + kk) + q) * JM_LDM + 4 * J + n];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int Ig = wave + 8 * g;
        if (Ig < ngrp) {   // wave-uniform
          double a[4];
          const double *ap = sm + (size_t)(16 * kh + q) * LDr + 16 * Ig + 4 * m + n;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) a[kk] = ap[(size_t)4 * kk * LDr];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int J = 0; J < 8; ++J) acc[g][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[kk], bq[kk][J], acc[g][J], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int Ig = wave + 8 * g;
      if (Ig < ngrp) {
        double *op = sm + (size_t)n * LDr + 16 * Ig + 4 * m + q;
#pragma unroll
        for (int J = 0; J < 8; ++J) op[(size_t)4 * J * LDr] = acc[g][J];
      }
    }
  }
  __syncthreads();
  if (stamp) tk4 = __builtin_readcyclecounter();

  // ---- 5. store
  for (int cblk = 0; cblk < (lone ? 1 : 2); ++cblk) {
    const int c0 = (cblk == 0 ? ba : bb) * JM_B;
    const int n2 = min(JM_B, p2 - c0) * half;
    double2 *dst = reinterpret_cast<double2 *>(G + (size_t)c0 * p2);
#pragma unroll 4
    for (int idx = tid; idx < n2; idx += JM8_NT) {
      const int cc = idx / half, r2 = idx - cc * half;
      dst[idx] = *reinterpret_cast<const double2 *>(sm + (size_t)(cblk * JM_B + cc) * LDr + 2 * r2);
    }
  }
  if (tid == 0) rot[mtx] = 1;
  if (stamp) {
    __syncthreads();
    if (tid == 0) {
      const unsigned long long tk5 = __builtin_readcyclecounter();
      atomicAdd(&g_jm_stamps[0], 1ull);
      atomicAdd(&g_jm_stamps[1], tk1 - tk0);
      atomicAdd(&g_jm_stamps[2], tk2 - tk1);
      atomicAdd(&g_jm_stamps[3], tk3 - tk2);
      atomicAdd(&g_jm_stamps[4], tk4 - tk3);
      atomicAdd(&g_jm_stamps[5], tk5 - tk4);
    }
  }
}

// after a sweep: a matrix without a rotation is finished
__global__ void k_bjm_flags(int nb, int32_t *__restrict__ done, int32_t *__restrict__ rot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nb) return;
  if (!rot[i]) done[i] = 1;
  rot[i] = 0;
}

}  // namespace

// The sweeps of the blocked Jacobi on the nb factors in gv ([nb][2][p2][p2], column major; cflag != 0: not this
// solver's matrix).  done / rot: nb flags each, zeroed here.  After `sweeps` sweeps a matrix that still rotates keeps
// done == 0 (the caller hands it to the single-workgroup solver).
int sf_launch_wide_blockjac_mfma(double *gv, int p2, int nb, const int32_t *cflag, int32_t *done, int32_t *rot, int sweeps,
                                 hipStream_t st) {
  SF_HIP(hipMemsetAsync(done, 0, (size_t)nb * sizeof(int32_t), st));
  SF_HIP(hipMemsetAsync(rot, 0, (size_t)nb * sizeof(int32_t), st));
  const int nblk = sf_cdiv(p2, JM_B), mblk = nblk + (nblk & 1);
  const int R16 = sf_cdiv(p2, 16) * 16;
  int LDr = R16;
  while ((LDr % 32) != 16) LDr += 2;
  static_assert(4 * 36 * 16 >= 2 * JM_P * JM_LDM, "part covers M1 and Q");
  const size_t lds = ((size_t)JM_P * LDr + JM_P * JM_LDM + 4 * 36 * 16) * sizeof(double);
  if (lds > 160 * 1024) {
    sf_set_error("blocked Jacobi: %d rows need %zu bytes of LDS", p2, lds);
    return -2;
  }
  const bool eight = sf_tune().wide_eigh_variant != 3;   // 4 = eight waves, 3 = the four-wave form (first version of the round)
  if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_bjm<true>), lds)) return rc;
  if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_bjm<false>), lds)) return rc;
  if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_bjm8<true>), lds)) return rc;
  if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_bjm8<false>), lds)) return rc;
  const int stamp = sf_tune().wjac_stamps;
  const int nsteps = (mblk > 1) ? mblk - 1 : 1;
  const int npair = mblk / 2 > 0 ? mblk / 2 : 1;
  for (int sweep = 0; sweep < sweeps; ++sweep) {   // converged matrices drop out by their flag; no host round trip
    if (eight) {
      hipLaunchKernelGGL(k_bjm8<true>, dim3(npair, nb), dim3(JM8_NT), lds, st, gv, p2, R16, LDr, nblk, mblk > 1 ? mblk : 2, 0,
                         cflag, done, rot, stamp);
      for (int s = 1; s < nsteps; ++s)
        hipLaunchKernelGGL(k_bjm8<false>, dim3(npair, nb), dim3(JM8_NT), lds, st, gv, p2, R16, LDr, nblk, mblk, s, cflag, done, rot,
                           stamp);
    } else {
      hipLaunchKernelGGL(k_bjm<true>, dim3(npair, nb), dim3(JM_NT), lds, st, gv, p2, R16, LDr, nblk, mblk > 1 ? mblk : 2, 0, cflag,
                         done, rot, stamp);
      for (int s = 1; s < nsteps; ++s)
        hipLaunchKernelGGL(k_bjm<false>, dim3(npair, nb), dim3(JM_NT), lds, st, gv, p2, R16, LDr, nblk, mblk, s, cflag, done, rot, stamp);
    }
    hipLaunchKernelGGL(k_bjm_flags, dim3(sf_cdiv(nb, 256)), dim3(256), 0, st, nb, done, rot);
  }
  SF_LAUNCH_CHECK("k_bjm");
  return 0;
}

extern "C" int sf_debug_wjac_stamps(unsigned long long *out8, int reset) {
  if (out8) SF_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_jm_stamps), 8 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    SF_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_jm_stamps), z, sizeof(z)));
  }
  return 0;
}
