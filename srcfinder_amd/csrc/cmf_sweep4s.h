// The streamed rank-factored LOO sweep kernel k_sweep4s and its LDS layout (round 3; templates over the rank group count NK and the
// band group count NJ).  A header since round 5: its nine instantiations are compiled in three translation units side by side
// (cmf_loocv4.hip: NJ = 18; cmf_loocv4_21.hip / cmf_loocv4_24.hip: the CO2 family) -- one file took 7.5 of the build's 8 minutes.
#pragma once
#include "cmf_common.h"
#include <type_traits>

namespace {

__device__ __forceinline__ double shfl_xor_d(double v, int m) { return __shfl_xor(v, m, 64); }

constexpr int S4J = SF_SW4_NJ, S4M = SF_SW4_NM;

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int CTRL>
__device__ __forceinline__ double dpp_row(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------------------------
// k_sweep4s: the rank-factored sweep of k_sweep4r as ONE hand-scheduled instruction stream per 16-row tile (round 3).
// Same three products, same accumulation order inside every chain (GEMM1: band steps ascending, GEMM2a: eigen groups
// ascending, GEMM2b: factor groups ascending), but
//   * every MFMA operand that comes from LDS travels through one register ring filled by in-order asm reads that run a
//     fixed number of reads AHEAD of their use across phase boundaries, awaited with counted s_waitcnt lgkmcnt(N) --
//     k_sweep4r waited for lgkmcnt(0) at each of its 18 + 18 band / eigen steps with the last read issued two MFMAs
//     earlier (an LDS round trip exposed 36 times per tile);
//   * operands of two consecutive MFMAs sit side by side in LDS and arrive as one ds_read_b128 (234 + 42 + 7 reads per
//     tile instead of 541);
//   * the tile prologue is branch-free: the column mean comes in nine unconditional 16-byte reads (k_sweep4r's
//     `rowok ? x - mu : 0` compiled to 18 exec-masked blocks, each an LDS read + lgkmcnt(0)), the next tile's rows are
//     fetched from an address that does not depend on the validity byte (which k_sweep4r loaded and WAITED for before
//     it could issue the row loads), and an invalid row is switched off where the data is narrowest: its 7 (9) values
//     of t are set to 0, so q = 1 exactly as with x = 0 (a NaN row stays inside its own column of every product);
//   What it does NOT do is get rid of the vector instructions, and they are what is left: on gfx950 every VALU instruction
//   issued beside the 4x4x4 fp64 MFMA stream costs ~6 cycles of matrix time even with two waves per SIMD
//   (tools/microbench/mix4w.hip: 3 VALU per 8 MFMA take the pipe from 72 to 56 TFLOP/s -- this kernel's rate; LDS reads,
//   s_waitcnt and s_nop are free).  Measured by leaving a class out (SF_SWEEP_EXPERIMENTS, results wrong, MFMAs kept):
//   the row reduction (221 of the 323) 0.40 ms, conversion + centring (36) 0.19 ms, the 42 DPP moves 0.05 ms, squares
//   and validity selects 0.03 + 0.05 ms; with every one of them gone the launch is 4.9 ms, not 4.2: profiles/r03_sweep_ablation.txt.
//   (The hardware's A-block broadcast -- cbsz / abid, which would replace the three DPP-rotated copies of t -- assembles
//   for v_mfma_f64_4x4x4_4b_f64 but is ignored by gfx950: tools/microbench/mfma4_layout.hip modes 1-3.)
template <int NK, int NJT = S4J>
struct SwS {
  // NJ band steps (= band groups of four: a lane group holds NJ consecutive bands), NJE eigen groups (NJ rounded up to even:
  // the W blocks of two eigen groups travel as one 16-byte pair; the padding group of an odd NJ is zero)
  static constexpr int NJ = NJT, NJE = NJT + (NJT & 1), NM = S4M, NKP = (NK + 1) / 2, NG = NM / 4;   // NG full groups of 4 alpha tiles + one tile
  static constexpr int NMU = (NJ + 1) / 2;   // 16-byte pairs of the column mean per lane group
  static constexpr int R1 = NJ * (NJE / 2);  // W block pairs (GEMM1 A operands), ds_read_b128
  static constexpr int R2 = NJE * NKP;       // -U block pairs (GEMM2a A operands), ds_read_b128 (odd NK: last pair half empty)
  static constexpr int R3 = NG * NK * 2;     // W fragment pairs of the full alpha-tile groups (GEMM2b B operands), ds_read_b128
  static constexpr int R4 = NK;              // W fragments of the 13th alpha tile, ds_read_b64
  static constexpr int NR = R1 + R2 + R3 + R4;
  static constexpr int LEAD12 = 8, LEAD3 = 4, RING = 10;
  static_assert(NM == 4 * NG + 1, "tile structure");
  static constexpr int lead(int c) { return c < R1 + R2 ? LEAD12 : LEAD3; }
  // reads issued when unit c (= read c and the MFMAs it feeds) is awaited
  static constexpr int issued(int c) {
    int m = 0;
    for (int i = 0; i <= c; ++i) {
      int t = i + lead(i);
      if (t > NR) t = NR;
      if (t > m) m = t;
    }
    return m;
  }
  // LDS layout (doubles)
  static constexpr int OW = 0;                               // [R1][16][2]
  static constexpr int OU = OW + R1 * 32;                    // [R2][16][2]
  static constexpr int OF = OU + R2 * 32;                    // [R3][64][2]
  static constexpr int OL = OF + R3 * 128;                   // [R4][64]
  static constexpr int OM = OL + R4 * 64;                    // mu [4 lane groups][2 NMU] (16-byte aligned slices)
  static constexpr int OS = OM + 8 * NMU;                    // 1/sqrt(lam) [4 NJE] (prologue only)
  static constexpr int TOTAL = OS + 4 * NJE;
  static constexpr size_t lds_bytes() { return (size_t)TOTAL * sizeof(double); }
};

typedef double d2_t __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ d2_t lds_ld128(unsigned addr) {
  d2_t r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
template <int OFF>
__device__ __forceinline__ double lds_ld64(unsigned addr) {
  double r;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
template <int CNT>
__device__ __forceinline__ void lds_await(d2_t &a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(CNT)); }
template <int CNT>
__device__ __forceinline__ void lds_await(d2_t &a, d2_t &b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(CNT)); }
template <int CNT>
__device__ __forceinline__ void lds_await(double &a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(CNT)); }
template <int I, int N>
__device__ __forceinline__ void lds_tie(d2_t (&m)[N]) {
  if constexpr (I < N) {
    asm volatile("" : "+v"(m[I]));
    lds_tie<I + 1, N>(m);
  }
}
// wait, then tie every register of the group to a point behind the wait (volatile asm statements keep their order)
template <int CNT, int N>
__device__ __forceinline__ void lds_await_all(d2_t (&m)[N]) {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(CNT));
  lds_tie<0, N>(m);
}

// EXP (timing experiments, wrong results, -DSF_SWEEP_EXPERIMENTS): bit 0 no row reduction, 1 no conversion / centring,
// 2 no DPP rotations, 3 no squares, 4 no validity selects -- the MFMAs and LDS reads stay; bit 5 no tiles at all (what the
// table prologue + epilogue of every workgroup cost), bit 6 no table prologue (tiles on whatever the LDS holds)
// RN: the running products are renormalised (mantissa / exponent split) after every RN-th tile of a wave.  A nonzero finite q of this
// sweep lies in [~1e-17, ~1e3] (1 + a sum of O(1) terms in fp64), so the product of the 4 RN = 16 values a lane folds in between two
// splits stays inside [1e-272, 1e48]: scaling by powers of two is exact there, the results are bit-identical to RN = 1.
template <int NK, int EXP = 0, int RN = 4, int NJT = S4J>
__global__ __launch_bounds__(512, 1) void k_sweep4s(const float *__restrict__ xt, const uint8_t *__restrict__ mask_t,
                                                    const int32_t *__restrict__ nuse, const double *__restrict__ mu,
                                                    const double *__restrict__ ufrag_g, const double *__restrict__ wfrag2_g,
                                                    const int32_t *__restrict__ lrok, const double *__restrict__ lam,
                                                    const double *__restrict__ wfrag,
                                                    size_t wstride, const int32_t *__restrict__ status,
                                                    const double *__restrict__ alphas, int nalpha, int L, int p,
                                                    int PS, int rows_per_wg, double *__restrict__ part, int split_fastest) {
  using S = SwS<NK, NJT>;
  constexpr int NJ = S::NJ, NJE = S::NJE, NMU = S::NMU, NM = S::NM, NA16 = NM * 16, NW = 8, NKP = S::NKP, NG = S::NG;
  constexpr int R1 = S::R1, R2 = S::R2, R3 = S::R3, NR = S::NR, RING = S::RING;
  constexpr int NK2 = SF_LR_K2 / 4;          // stride of the global fragment layout (shared by both ranks)
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  // grid (column, split) or (split, column): with the splits fastest the workgroups that load one column's tables run at the
  // same time (the tables come from HBM once and from L2 for the rest); results do not depend on the order
  const int c = split_fastest ? blockIdx.y : blockIdx.x, split = split_fastest ? blockIdx.x : blockIdx.y;
  const int nsplit = split_fastest ? gridDim.x : gridDim.y;
  double *po = part + ((size_t)c * nsplit + split) * 2 * NA16;
  if (status[c] != 0 || lrok[c] != sf_lr_code(NK)) return;   // another instantiation / k_sweep4 takes these columns
  // ---- prologue: the tables, permuted into the pair layouts.  Every global load of the workgroup is issued first (16-byte
  // loads, compile-time trip counts: ~17 per thread in flight at once), then the LDS stores: with one workgroup per CU
  // nothing else runs on the CU meanwhile, and a load-store loop paid one L2 round trip per iteration
  if constexpr ((EXP & 64) == 0) {
    constexpr int NT = 64 * NW;
    constexpr int NW2 = NJ * NJE * 8, NU2 = NJE * NKP * 16, NF2 = R3 * 64, NL2 = NK * 32;   // 16-byte pieces of the four tables
    constexpr int IW = (NW2 + NT - 1) / NT, IU = (NU2 + NT - 1) / NT, IF = (NF2 + NT - 1) / NT, IL = (NL2 + NT - 1) / NT;
    static_assert(IL == 1 && 4 * NJE <= NT, "one piece per thread");
    double *mus = sm + S::OM, *scl = sm + S::OS;
    const d2_t *wsrc = reinterpret_cast<const d2_t *>(wfrag + (size_t)c * wstride);            // [(s*NJE + jg)*16 + 4q + n]
    const d2_t *us = reinterpret_cast<const d2_t *>(ufrag_g + (size_t)c * (NJE * NK2 * 16));   // [(jg*NK2 + mg)*16 + 4q + n]
    const double *ws = wfrag2_g + (size_t)c * (NM * NK2 * 64);                                 // [(M*NK2 + mg)*64 + lane]
    const d2_t zero2 = {0.0, 0.0};
    d2_t vw[IW], vu[IU], vf[IF], vl;
    double lamv = 1.0, muv = 0.0;
    if (tid < p) { lamv = lam[(size_t)c * p + tid]; muv = mu[(size_t)c * p + tid]; }
#pragma unroll
    for (int k = 0; k < IW; ++k) { const int i = tid + k * NT; vw[k] = (i < NW2) ? wsrc[i] : zero2; }
#pragma unroll
    for (int k = 0; k < IU; ++k) {
      const int i = tid + k * NT, sl2 = i & 7, h = (i >> 3) & 1, pr = i >> 4, jg = pr / NKP, mg = 2 * (pr - jg * NKP) + h;
      vu[k] = (i < NU2 && mg < NK) ? us[(jg * NK2 + mg) * 8 + sl2] : zero2;
    }
#pragma unroll
    for (int k = 0; k < IF; ++k) {
      const int i = tid + k * NT, ln2 = i & 31, kk = (i >> 5) & 1, r = i >> 6;   // r = (gr*NK + mg)*2 + h
      const int h = r & 1, gm = r >> 1, gr = gm / NK, mg = gm - gr * NK, M = 4 * gr + 2 * h + kk;
      vf[k] = (i < NF2) ? *reinterpret_cast<const d2_t *>(ws + (M * NK2 + mg) * 64 + 2 * ln2) : zero2;
    }
    vl = (tid < NL2) ? *reinterpret_cast<const d2_t *>(ws + ((NM - 1) * NK2 + (tid >> 5)) * 64 + 2 * (tid & 31)) : zero2;
    // W blocks scaled by 1/sqrt(lam_j): GEMM1 then yields the whitened coordinates y_j/sqrt(lam_j) (unit variance) and
    // z their squares, which is what the row-scaled factorisation of cmf_lowrank.hip multiplies
    // (band b = NJ g + s of lane group g sits at mus[2 NMU g + s]: every group's slice starts on a 16-byte boundary)
    if (tid < 4 * NJE) {
      scl[tid] = (tid < p) ? 1.0 / sqrt(lamv) : 1.0;
      if (tid < 4 * NJ) mus[2 * NMU * (tid / NJ) + tid % NJ] = muv;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < IW; ++k) {
      const int i = tid + k * NT;
      if (i < NW2) {
        const int blk = i >> 3, sl = 2 * (i & 7), s = blk / NJE, jg = blk - s * NJE;
        double *dst = sm + S::OW + ((s * (NJE / 2) + (jg >> 1)) * 16 + sl) * 2 + (jg & 1);
        dst[0] = vw[k].x * scl[4 * jg + (sl & 3)];
        dst[2] = vw[k].y * scl[4 * jg + (sl & 3) + 1];
      }
    }
#pragma unroll
    for (int k = 0; k < IU; ++k) {
      const int i = tid + k * NT, sl2 = i & 7, h = (i >> 3) & 1, pr = i >> 4;
      if (i < NU2) {
        double *dst = sm + S::OU + pr * 32 + 4 * sl2 + h;
        dst[0] = vu[k].x;
        dst[2] = vu[k].y;
      }
    }
#pragma unroll
    for (int k = 0; k < IF; ++k) {
      const int i = tid + k * NT, ln2 = i & 31, kk = (i >> 5) & 1, r = i >> 6;
      if (i < NF2) {
        double *dst = sm + S::OF + r * 128 + 4 * ln2 + kk;
        dst[0] = vf[k].x;
        dst[2] = vf[k].y;
      }
    }
    if (tid < NL2) *reinterpret_cast<d2_t *>(sm + S::OL + 2 * tid) = vl;
  }
  __syncthreads();

  double P[NM], N[NM];
  int E[NM], sg[NM];   // sg: OR of the sign words of every q seen (a negative q makes log(q), hence the NLL, NaN in the reference)
#pragma unroll
  for (int u = 0; u < NM; ++u) { P[u] = 1.0; N[u] = 0.0; E[u] = 0; sg[u] = 0; }
  int ntile = 0;
  int nrowok = 0;   // valid rows seen by this lane (lanes with g == 0 cover every row of the wave's tiles once)

  const int rbeg = split * rows_per_wg, rend = (EXP & 32) ? rbeg : min(L, rbeg + rows_per_wg);
  const uint8_t *mp = mask_t + (size_t)c * L;
  const float *xc = xt + (size_t)c * L * PS + NJ * g;
  const unsigned smb = (unsigned)(size_t)sm;
  const unsigned wadr = smb + S::OW * 8 + (4 * g + (lane & 3)) * 16;
  const unsigned uadr = smb + S::OU * 8 + (4 * g + (lane & 3)) * 16;
  const unsigned fadr = smb + S::OF * 8 + lane * 16;
  const unsigned ladr = smb + S::OL * 8 + lane * 8;
  const unsigned madr = smb + S::OM * 8 + g * (NMU * 16);
  const double qnan = __builtin_nan("");

  float xraw[NJ];
  bool rowok_next;
  auto fetch = [&](int r0, float (&dst)[NJ], bool &ok) {
    const int row = r0 + li, rowc = row < rend ? row : rend - 1;   // the address does not depend on the validity byte
    ok = (mp[rowc] != 0) && (row < rend);
    const float *xp = xc + (size_t)rowc * PS;
#pragma unroll
    for (int s = 0; s + 1 < NJ; s += 2) sf_load2(xp + s, dst[s], dst[s + 1]);
    if constexpr (NJ & 1) dst[NJ - 1] = xp[NJ - 1];
  };
  int r0 = rbeg + 16 * wave;
  if (r0 < rend) fetch(r0, xraw, rowok_next);

  for (; r0 < rend;) {
  for (int it = 0; it < RN && r0 < rend; ++it, r0 += 16 * NW) {
    const bool rowok = rowok_next;
    nrowok += (rowok && g == 0) ? 1 : 0;
    d2_t ring[RING];
    double ringl[RING];
    double x[NJ], z[NJE], t[4][NK], acc[4][4];
    d2_t mur[NMU];
    auto issue = [&](auto rc) {   // read r of the tile's stream into its ring slot
      constexpr int r = decltype(rc)::value;
      if constexpr (r < R1 && r * 256 >= 65536) ring[r % RING] = lds_ld128<r * 256 - 32768>(wadr + 32768);   // (16-bit offset field)
      else if constexpr (r < R1) ring[r % RING] = lds_ld128<r * 256>(wadr);
      else if constexpr (r < R1 + R2) ring[r % RING] = lds_ld128<(r - R1) * 256>(uadr);
      else if constexpr (r < R1 + R2 + R3) ring[r % RING] = lds_ld128<(r - R1 - R2) * 1024>(fadr);
      else ringl[r % RING] = lds_ld64<(r - R1 - R2 - R3) * 512>(ladr);
    };
    auto cvt = [&](auto sc) {     // x[s] = row value - column mean  (fp64, as the reference centres)
      constexpr int s = decltype(sc)::value;
      double v = (double)xraw[s] - ((s & 1) ? mur[s >> 1].y : mur[s >> 1].x);
      if constexpr ((EXP & 2) != 0) {
        v = (s & 1) ? mur[s >> 1].y : mur[s >> 1].x;
        asm volatile("" :: "v"(xraw[s]));
      }
      if constexpr (3 * NJ + s >= 4 * NJ - 3) v = (NJ * g + s < p) ? v : 0.0;   // p >= 69: only bands 69..71 can lie beyond the window
      x[s] = v;
    };
    static_for<0, NMU>([&](auto ic) { mur[decltype(ic)::value] = lds_ld128<decltype(ic)::value * 16>(madr); });
    static_for<0, S::issued(0)>(issue);
    lds_await_all<S::issued(0)>(mur);
    cvt(std::integral_constant<int, 0>{});
    cvt(std::integral_constant<int, 1>{});

    // Row reduction of alpha tile u from acc[k][0..3] (the q of 4 rows), in four stages of independent instructions:
    //   m = q0 q1 q2 q3,  nu = (q0+q1) q2 q3 + (q2+q3) q0 q1  (= m sum 1/q_s),  N <- N m + nu P,  P <- P m  (P: mantissa, exponent in E)
    double rm01[4], rm23[4], rs01[4], rs23[4], rm[4], rnu[4];
    auto reduce_stage = [&](auto grc, auto stagec) {
      constexpr int gq = decltype(grc)::value, stage = decltype(stagec)::value;
      constexpr int ntl = (NM - gq * 4) < 4 ? (NM - gq * 4) : 4;
#pragma unroll
      for (int k = 0; k < ntl; ++k) {
        const int u = gq * 4 + k;
        if constexpr (stage == 0) {
          const double q0 = acc[k][0], q1 = acc[k][1], q2 = acc[k][2], q3 = acc[k][3];
          rm01[k] = q0 * q1; rm23[k] = q2 * q3; rs01[k] = q0 + q1; rs23[k] = q2 + q3;
          int sw = sg[u];                                        // any q < 0 so far (two v_or3_b32)
          sw = (sw | __double2hiint(q0)) | __double2hiint(q1);
          sg[u] = (sw | __double2hiint(q2)) | __double2hiint(q3);
        } else if constexpr (stage == 1) {
          rm[k] = rm01[k] * rm23[k];
          rnu[k] = __builtin_fma(rs01[k], rm23[k], rs23[k] * rm01[k]);
        } else if constexpr (stage == 2) {
          N[u] = __builtin_fma(N[u], rm[k], rnu[k] * P[u]);   // N m + nu P
          P[u] = P[u] * rm[k];                                // P m
        } else {
          const int e = __builtin_amdgcn_frexp_exp(P[u]);
          P[u] = __builtin_amdgcn_frexp_mant(P[u]);
          N[u] = __builtin_amdgcn_ldexp(N[u], -e);
          E[u] += e;
        }
      }
    };
    auto reduce_group = [&](auto grc) {
      if constexpr ((EXP & 1) != 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int q = 0; q < 4; ++q) asm volatile("" :: "v"(acc[k][q]));
        return;
      }
      static_for<0, (RN == 1 ? 4 : 3)>([&](auto sc) { reduce_stage(grc, sc); });
    };

    static_for<0, NR>([&](auto cc) {
      constexpr int u = decltype(cc)::value;   // unit = read u + the MFMAs it feeds
      constexpr int slot = u % RING;
      constexpr bool pairwait = (u < R1 + R2);                                  // phases 1, 2: one wait per two units
      if constexpr (!pairwait || (u & 1) == 0) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (u >= R1 + R2 + R3) lds_await<S::issued(u) - u - 1>(ringl[slot]);
        else if constexpr (!pairwait) lds_await<S::issued(u) - u - 1>(ring[slot]);
        else lds_await<S::issued(u) - u - 2>(ring[slot], ring[(u + 1) % RING]);   // R1, R2 are even: u + 1 is in the same phase
      }
      if constexpr (u < R1) {
        constexpr int s = u / (NJE / 2), jg = 2 * (u % (NJE / 2));
        if constexpr (s == 0) {
          z[jg] = __builtin_amdgcn_mfma_f64_4x4x4f64(ring[slot].x, x[s], 0.0, 0, 0, 0);
          z[jg + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(ring[slot].y, x[s], 0.0, 0, 0, 0);
        } else {
          z[jg] = __builtin_amdgcn_mfma_f64_4x4x4f64(ring[slot].x, x[s], z[jg], 0, 0, 0);
          z[jg + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(ring[slot].y, x[s], z[jg + 1], 0, 0, 0);
        }
        if constexpr (jg == 0 && s + 2 < NJ) cvt(std::integral_constant<int, s + 2>{});   // two band steps ahead
        if constexpr (u == (NJ - 2) * (NJE / 2)) {   // every raw value is converted: the next tile's rows may land in xraw
          if (r0 + 16 * NW < rend) fetch(r0 + 16 * NW, xraw, rowok_next);
        }
        if constexpr (u == R1 - 1 && (EXP & 8) == 0) z[0] = z[0] * z[0];
      } else if constexpr (u < R1 + R2) {
        constexpr int v = u - R1, jg = v / NKP, mg = 2 * (v % NKP);
        if constexpr (jg == 0) {
          t[0][mg] = __builtin_amdgcn_mfma_f64_4x4x4f64(ring[slot].x, z[jg], 0.0, 0, 0, 0);
          if constexpr (mg + 1 < NK) t[0][mg + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(ring[slot].y, z[jg], 0.0, 0, 0, 0);
        } else {
          t[0][mg] = __builtin_amdgcn_mfma_f64_4x4x4f64(ring[slot].x, z[jg], t[0][mg], 0, 0, 0);
          if constexpr (mg + 1 < NK) t[0][mg + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(ring[slot].y, z[jg], t[0][mg + 1], 0, 0, 0);
        }
        if constexpr (mg == 2 && jg + 1 < NJE && (EXP & 8) == 0) z[jg + 1] = z[jg + 1] * z[jg + 1];   // the square the next eigen group multiplies
        if constexpr (u == R1 + R2 - 1) {   // an invalid row leaves every q of the tile at 1
          if constexpr ((EXP & 16) == 0) {
#pragma unroll
            for (int m = 0; m < NK; ++m) t[0][m] = rowok ? t[0][m] : 0.0;
          }
#pragma unroll
          for (int m = 0; m < NK; ++m) {   // block m of rotation s meets row group (m + s) % 4
            if constexpr ((EXP & 4) != 0) {
              t[1][m] = t[2][m] = t[3][m] = t[0][m];
            } else {
              t[1][m] = dpp_row<0x124>(t[0][m]);  // row_ror:4
              t[2][m] = dpp_row<0x128>(t[0][m]);  // row_ror:8
              t[3][m] = dpp_row<0x12C>(t[0][m]);  // row_ror:12
            }
          }
        }
      } else if constexpr (u < R1 + R2 + R3) {
        constexpr int v = u - R1 - R2, h = v & 1, gm = v >> 1, gr = gm / NK, jg = gm - gr * NK;
        static_for<0, 2>([&](auto kc) {
          constexpr int kk = decltype(kc)::value, k = 2 * h + kk;
          const double b = kk ? ring[slot].y : ring[slot].x;
          static_for<0, 4>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if constexpr (jg == 0 && (EXP & 4) != 0)   // distinct chains although the four A operands are the same register
              acc[k][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(t[s][jg], b, s == 0 ? 1.0 : (s == 1 ? 2.0 : (s == 2 ? 0.5 : 4.0)), 0, 0, 0);
            else if constexpr (jg == 0) acc[k][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(t[s][jg], b, 1.0, 0, 0, 0);
            else acc[k][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(t[s][jg], b, acc[k][s], 0, 0, 0);
          });
        });
        if constexpr (h == 1 && jg == NK - 1) reduce_group(std::integral_constant<int, gr>{});   // the group is complete
      } else {
        constexpr int jg = u - R1 - R2 - R3;
        static_for<0, 4>([&](auto sc) {
          constexpr int s = decltype(sc)::value;
          if constexpr (jg == 0 && (EXP & 4) != 0)
            acc[0][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(t[s][jg], ringl[slot], s == 0 ? 1.0 : (s == 1 ? 2.0 : (s == 2 ? 0.5 : 4.0)), 0, 0, 0);
          else if constexpr (jg == 0) acc[0][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(t[s][jg], ringl[slot], 1.0, 0, 0, 0);
          else acc[0][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(t[s][jg], ringl[slot], acc[0][s], 0, 0, 0);
        });
        if constexpr (jg == NK - 1) {
          __builtin_amdgcn_sched_barrier(0);
          reduce_group(std::integral_constant<int, NG>{});
        }
      }
      // keep the ring `lead` reads ahead
      if constexpr (u + 1 < NR) static_for<S::issued(u), S::issued(u + 1)>(issue);
    });
    ntile += 1;
  }
    if constexpr (RN > 1) {
#pragma unroll
      for (int u = 0; u < NM; ++u) {
        const int e = __builtin_amdgcn_frexp_exp(P[u]);
        P[u] = __builtin_amdgcn_frexp_mant(P[u]);
        N[u] = __builtin_amdgcn_ldexp(N[u], -e);
        E[u] += e;
      }
    }
  }

  __syncthreads();
  double *redP = sm;
  double *redR = redP + NW * NA16;
  int *redE = reinterpret_cast<int *>(redR + NW * NA16);
#pragma unroll
  for (int u = 0; u < NM; ++u) {
    double pv = P[u], rv = N[u] / P[u] - 4.0 * (double)ntile;   // sum over this lane's rows of (1/q - 1) = beta r/q
    if (sg[u] < 0) rv = qnan;                                      // applied once, here, instead of per 16-row tile
    int ev = E[u];
#pragma unroll
    for (int msk = 16; msk <= 32; msk <<= 1) {
      const double po2 = shfl_xor_d(pv, msk);
      const int eo = __shfl_xor(ev, msk, 64);
      rv += shfl_xor_d(rv, msk);
      const double pm = pv * po2;
      ev += eo + __builtin_amdgcn_frexp_exp(pm);
      pv = __builtin_amdgcn_frexp_mant(pm);
    }
    if (g == 0) {
      redP[wave * NA16 + 16 * u + li] = pv;
      redR[wave * NA16 + 16 * u + li] = rv;
      redE[wave * NA16 + 16 * u + li] = ev;
    }
  }
  __syncthreads();
  for (int i = tid; i < NA16; i += 64 * NW) {
    double pv = 1.0, rv = 0.0;
    int ev = 0;
    for (int w = 0; w < NW; ++w) {
      const double pm = pv * redP[w * NA16 + i];
      ev += redE[w * NA16 + i] + __builtin_amdgcn_frexp_exp(pm);
      pv = __builtin_amdgcn_frexp_mant(pm);
      rv += redR[w * NA16 + i];
    }
    po[i] = log(pv) + (double)ev * 0.6931471805599453094;
    po[NA16 + i] = rv;   // = beta_i sum_k r_k/q_k: k_nll divides (rq_scaled = 1)
  }
  // the number of rows this workgroup accumulated, for the beta = 0 term of k_nll: kept in the last padding slot
  // of the alpha axis (the grid has 201 points, the tiles 208)
  __syncthreads();
  int *cred = reinterpret_cast<int *>(sm);
  for (int off = 32; off > 0; off >>= 1) nrowok += __shfl_xor(nrowok, off, 64);
  if (lane == 0) cred[wave] = nrowok;
  __syncthreads();
  if (tid == 0 && nalpha < NA16) {
    int tot = 0;
    for (int w = 0; w < NW; ++w) tot += cred[w];
    po[2 * NA16 - 1] = (double)tot;
  }
}

}  // namespace
