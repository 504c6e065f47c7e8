// Stage 5, production window: the LOO sweep on v_mfma_f64_4x4x4_f64 (see cmf_loocv.hip for the algorithm and for
// the 16x16x4 kernels that serve every other window).  Built with -mllvm -amdgpu-mfma-vgpr-form=1: the results of
// both products feed VALU code (square, row reduction), and every VALU instruction stalls the 4x4x4 fp64 MFMA
// stream (tools/microbench/mix4.hip: +4.5 cycles per instruction on a 16.5-cycle slot), so AGPR accumulators and
// their v_accvgpr_read copies are pure loss here.
#include "cmf_common.h"
#include <type_traits>
#include "cmf_sweep4s.h"


namespace {


// ---------------------------------------------------------------------------------------------------------------
// k_sweep4: the production window (p <= 72, 201-point grid) on v_mfma_f64_4x4x4_f64.
//
// Measured on MI355X (tools/microbench/mfma64.hip): the 16x16x4 fp64 MFMA issues at ~100 cycles (46-48 TFLOP/s),
// the 4x4x4 (4 blocks) one at 17 cycles (71 TFLOP/s, ~90 % of the DP unit) -- 1.5x the throughput for 4x the
// operand traffic.  Register layout of the 4x4x4 instruction (tools/microbench/mfma4_layout.hip), with
// lane = 16 q + 4 m + n:   block = m;   A[i][k] at (q = k, n = i);   B[k][j] at (q = k, n = j);   D[i][j] at (q = i, n = j).
//
//   GEMM1  D1[jidx][row] += W^T[jidx][band] . X^T[band][row]     blocks m = the four 4-row groups of the 16-row tile,
//          A = W blocks (LDS, 16 values broadcast to the 4 blocks), B = this lane's 18 consecutive bands of its row
//   z = D1^2 sits at (q = jidx, n = row): exactly the A layout of the next product.  Rotating z by 0/4/8/12 lanes
//   inside each 16-lane row (DPP row_ror) lets block m meet row group (m + s) % 4:
//   GEMM2  D2_s[row][alpha] += z_s[row][jidx] . C[jidx][alpha]    blocks m = four 4-alpha groups of a 16-alpha tile,
//          B = c_ij fragments (LDS, 512 B per read, reused by the 4 rotations)
// so every (row, alpha) pair is produced exactly once and a lane owns ONE alpha (16 M + (lane & 15)) for 4 rows of
// the tile per alpha tile M -- the same reduction shape as k_sweep.  324 + 936 MFMAs per 16 rows.
// LDS: c fragments 119,808 B + W blocks 41,472 B + mu = 161,856 B of the 163,840.
// LDS reads of GEMM2 step t (one per alpha tile of its group of 4; none past the last step)
constexpr int sw4r_reads(int t, int nk) {   // k_sweep4r: nk steps per group of 4 alpha tiles
  const int gr = t / nk, left = S4M - 4 * gr;
  return left <= 0 ? 0 : (left < 4 ? left : 4);
}
constexpr int sw4_reads(int t) {
  const int gr = t / S4J, left = S4M - 4 * gr;
  return left <= 0 ? 0 : (left < 4 ? left : 4);
}

__global__ void k_wfrag4(const double *__restrict__ evec, const double *__restrict__ d, int p, int nj, int nje, size_t stride,
                         double *__restrict__ wfrag) {
  // wfrag[c][(ig*nje + jg)*16 + 4 q + n] = V[b][j] / d[b],   b = nj q + ig,   j = 4 jg + n   (nje = nj rounded up to even: the
  // streamed sweep reads the blocks of two eigen groups as one pair; the padding group is zero)
  const int c = blockIdx.x;
  const double *ev = evec + (size_t)c * p * p;
  const double *dd = d + (size_t)c * p;
  double *w = wfrag + (size_t)c * stride;
  for (int i = threadIdx.x; i < nj * nje * 16; i += blockDim.x) {
    const int n = i & 3, q = (i >> 2) & 3, blk = i >> 4;
    const int ig = blk / nje, jg = blk - ig * nje;
    const int b = nj * q + ig, j = 4 * jg + n;
    w[i] = (j < p && b < p) ? ev[(size_t)j * p + b] / dd[b] : 0.0;
  }
}

// LDS reads (into AGPRs: MFMA operands only, keeps the VALU-visible file free) the compiler may not move or merge: issue order = source order, completion is awaited explicitly with
// lds_wait<N>() whose operands tie the loaded registers to the wait (nothing can read them before it).
template <int OFF, bool VG = false>
__device__ __forceinline__ double lds_ld(unsigned addr) {
  double r;
  if constexpr (VG) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  else asm volatile("ds_read_b64 %0, %1 offset:%2" : "=a"(r) : "v"(addr), "n"(OFF));
  return r;
}
template <int CNT>
__device__ __forceinline__ void lds_wait(double &a, double &b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(CNT));
}
template <int CNT, bool VG = false>
__device__ __forceinline__ void lds_wait1(double &a) {
  if constexpr (VG) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(CNT));
  else asm volatile("s_waitcnt lgkmcnt(%1)" : "+a"(a) : "n"(CNT));
}
template <int CNT, bool VG = false>
__device__ __forceinline__ void lds_wait4(double &a, double &b, double &c, double &d) {
  if constexpr (VG) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(CNT));
  else asm volatile("s_waitcnt lgkmcnt(%4)" : "+a"(a), "+a"(b), "+a"(c), "+a"(d) : "n"(CNT));
}
template <int CNT, bool VG = false>
__device__ __forceinline__ void lds_wait6(double &a, double &b, double &c, double &d, double &e, double &f) {
  if constexpr (VG) asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(CNT));
  else asm volatile("s_waitcnt lgkmcnt(%6)" : "+a"(a), "+a"(b), "+a"(c), "+a"(d), "+a"(e), "+a"(f) : "n"(CNT));
}

template <int EXP>  // 0 = production; 1..3 = timing experiments (skip GEMM1 / epilogue / GEMM2), wrong results
__global__ __launch_bounds__(256, 1) void k_sweep4(const float *__restrict__ xt, const uint8_t *__restrict__ mask_t,
                                                    const int32_t *__restrict__ nuse, const double *__restrict__ mu,
                                                    const double *__restrict__ lam, const double *__restrict__ wfrag,
                                                    size_t wstride, const int32_t *__restrict__ status,
                                                    const double *__restrict__ alphas, int nalpha, int L, int p,
                                                    int PS, int rows_per_wg, double *__restrict__ part,
                                                    const int32_t *__restrict__ lrok) {
  constexpr int NJ = S4J, NM = S4M, NA16 = NM * 16;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double *cfrag = sm;                        // [NM][NJ][64]
  double *wblk = cfrag + NM * NJ * 64;       // [NJ ig][NJ jg][16]
  double *mus = wblk + NJ * NJ * 16;         // [72]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int c = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
  double *po = part + ((size_t)c * nsplit + split) * 2 * NA16;
  if (lrok && lrok[c] != 0) return;   // this column is swept in rank-factored form (k_sweep4r)
  if (status[c] != 0) {
    for (int i = tid; i < 2 * NA16; i += 256) po[i] = 0.0;
    return;
  }
  const double n = (double)nuse[c];
  // ---- prologue.  The alpha grid and the eigenvalues are staged in LDS first (the fragment loop would otherwise
  //      wait on two dependent global loads per entry), and the global -> LDS copies are issued in batches.
  for (int i = tid; i < 4 * NJ; i += 256) mus[i] = (i < p) ? mu[(size_t)c * p + i] : 0.0;
  {
    double *ta = wblk, *tl = wblk + NA16;   // temporaries in the W area (filled afterwards)
    for (int i = tid; i < NA16; i += 256) ta[i] = (i < nalpha) ? alphas[i] : 1.0;
    for (int i = tid; i < 4 * NJ; i += 256) tl[i] = (i < p) ? lam[(size_t)c * p + i] : 1.0;
    __syncthreads();
    // fragment position idx = tid + 256 k: the lane is fixed, so alpha = 16 u + (lane & 15), eigen index = 4 jg + (lane >> 4)
#pragma unroll 2
    for (int idx = tid; idx < NM * NJ * 64; idx += 256) {
      const int us = idx >> 6;
      const int u = us / NJ, jg = us - u * NJ;
      const double a = ta[16 * u + li], lj = tl[4 * jg + g];
      const double beta = (1.0 - a) / (n - 1.0);
      const double v = -beta / ((n * beta) * lj + a);   // GEMM2 accumulates q = 1 - beta r directly (accumulator starts at 1)
      cfrag[idx] = (16 * u + li < nalpha && 4 * jg + g < p) ? v : 0.0;
    }
    __syncthreads();
    const double *wsrc = wfrag + (size_t)c * wstride;
    constexpr int WN = NJ * NJ * 16;   // 5184 = 20.25 x 256
#pragma unroll
    for (int k0 = 0; k0 < 5; ++k0) {
      double t[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] = wsrc[tid + 256 * (4 * k0 + k)];
#pragma unroll
      for (int k = 0; k < 4; ++k) wblk[tid + 256 * (4 * k0 + k)] = t[k];
    }
    if (tid < WN - 20 * 256) wblk[tid + 20 * 256] = wsrc[tid + 20 * 256];
  }
  __syncthreads();

  double P[NM], N[NM];
  int E[NM], sg[NM];   // sg: OR of the sign words of every q seen (a negative q makes log(q), hence the NLL, NaN in the reference)
#pragma unroll
  for (int u = 0; u < NM; ++u) { P[u] = 1.0; N[u] = 0.0; E[u] = 0; sg[u] = 0; }
  int ntile = 0;
  int nrowok = 0;   // valid rows seen by this lane (lanes with g == 0 cover every row of the wave's tiles once)

  const int rbeg = split * rows_per_wg, rend = min(L, rbeg + rows_per_wg);
  const uint8_t *mp = mask_t + (size_t)c * L;
  const float *xc = xt + (size_t)c * L * PS + NJ * g;
  const double *cf = cfrag + lane;
  const double *wf = wblk + 4 * g + (lane & 3);
  const double qnan = __builtin_nan("");

  float xraw[NJ];
  bool rowok_next;
  auto fetch = [&](int r0, float (&dst)[NJ], bool &ok) {
    const int row = r0 + li, rowc = row < rend ? row : rend - 1;   // round 3: the address does not wait for the validity byte
    ok = (mp[rowc] != 0) && (row < rend);
    const float *xp = xc + (size_t)rowc * PS;
#pragma unroll
    for (int s = 0; s < NJ; s += 2) sf_load2(xp + s, dst[s], dst[s + 1]);
  };
  int r0 = rbeg + 16 * wave;
  if (r0 < rend) fetch(r0, xraw, rowok_next);
  unsigned colm4[3];   // lane group g = 3 holds bands 54..71: those beyond the window are switched off
#pragma unroll
  for (int i = 0; i < 3; ++i) colm4[i] = (NJ * g + (NJ - 3) + i < p) ? 0xffffffffu : 0u;
  double *zeros4 = mus + 4 * NJ;   // 18 zeros behind the mean (the LDS block has room: see SW4_LDS)
  if (tid < NJ) zeros4[tid] = 0.0;
  __syncthreads();

  if (EXP == 4) r0 = rend;   // timing experiment: prologue + final reduction only
  for (; r0 < rend; r0 += 16 * 4) {
    const bool rowok = rowok_next;
    nrowok += (rowok && g == 0) ? 1 : 0;
    int opq = 0;
    asm volatile("" : "+v"(opq));  // keeps the loop-invariant LDS operand reads inside the iteration (see k_sweep)
    const double *cfl = cf + opq;
    const double *wfl = wf + opq;
    // branch-free (round 3; the select form compiled to 18 exec-masked blocks, each an LDS read + lgkmcnt(0)): an invalid
    // row's raw bits and its mean are switched to 0, so its operand is exactly 0 whatever the row held
    const double *musl = (rowok ? mus + NJ * g : zeros4) + opq;
    unsigned okm = rowok ? 0xffffffffu : 0u;
    asm volatile("" : "+v"(okm));
    double x[NJ];
#pragma unroll
    for (int s = 0; s < NJ; ++s) {
      unsigned msk = okm;
      if (s >= NJ - 3) msk &= colm4[s - (NJ - 3)];
      x[s] = (double)__uint_as_float(__float_as_uint(xraw[s]) & msk) - musl[s];
    }
    if (r0 + 64 < rend) fetch(r0 + 64, xraw, rowok_next);
    // ---- GEMM1: 18 independent chains, one 4-band step at a time.  The A blocks of step s+1 are read (in-order
    //      asm reads, two per two MFMAs) while step s multiplies; left to itself the scheduler sinks every read
    //      next to its use and waits for it: 2 MFMAs per LDS round trip.
    double z[4][NJ];
    double wa[2][NJ];
    const unsigned wadr = (unsigned)(size_t)wfl;  // LDS byte address of this lane's slot in block (0, 0)
    static_for<0, NJ>([&](auto jc) {
      constexpr int jg = decltype(jc)::value;
      z[0][jg] = 0.0;
      wa[0][jg] = lds_ld<jg * 128>(wadr);
    });
    if constexpr (EXP == 1) {
#pragma unroll
      for (int jg = 0; jg < NJ; ++jg) z[0][jg] = x[jg] + wa[0][jg];
    } else
    static_for<0, NJ>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      static_for<0, NJ / 6>([&](auto kc) {
        constexpr int k = decltype(kc)::value * 6;
        lds_wait6<0>(wa[s & 1][k], wa[s & 1][k + 1], wa[s & 1][k + 2], wa[s & 1][k + 3], wa[s & 1][k + 4], wa[s & 1][k + 5]);
      });
      static_for<0, NJ / 2>([&](auto jc) {
        constexpr int jg = decltype(jc)::value * 2;
        if constexpr (s + 1 < NJ) {
          wa[(s + 1) & 1][jg] = lds_ld<((s + 1) * NJ + jg) * 128>(wadr);
          wa[(s + 1) & 1][jg + 1] = lds_ld<((s + 1) * NJ + jg + 1) * 128>(wadr);
        }
        z[0][jg] = __builtin_amdgcn_mfma_f64_4x4x4f64(wa[s & 1][jg], x[s], z[0][jg], 0, 0, 0);
        z[0][jg + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(wa[s & 1][jg + 1], x[s], z[0][jg + 1], 0, 0, 0);
      });
    });
#pragma unroll
    for (int jg = 0; jg < NJ; ++jg) {
      const double zz = z[0][jg] * z[0][jg];
      z[0][jg] = zz;
      z[1][jg] = dpp_row<0x124>(zz);  // row_ror:4
      z[2][jg] = dpp_row<0x128>(zz);  // row_ror:8
      z[3][jg] = dpp_row<0x12C>(zz);  // row_ror:12
    }
    // ---- GEMM2: alpha tiles in groups of 4 (16 independent chains); the c fragments are read DEPTH-1 steps
    //      ahead through a register ring that runs across group boundaries.  The fragments carry -beta_i, the
    //      accumulators start at 1: what comes out is q = 1 - beta r, and with r/q = (1/q - 1)/beta the row
    //      reduction needs only  prod q  and  sum 1/q.  Both are kept as a fraction: P = prod q (mantissa, the
    //      exponent split off into E) and N with N/P = sum 1/q, updated per 4 rows without a division:
    //        m = q0 q1 q2 q3,  nu = (q0+q1) q2 q3 + (q2+q3) q0 q1  (= m sum 1/q_s),  N <- N m + nu P,  P <- P m.
    constexpr int TG = 4, NG = (NM + TG - 1) / TG, NSTEP = NG * NJ, DEPTH = 4;
    double br[TG][DEPTH];
    const unsigned cadr = (unsigned)(size_t)cfl;        // fragment (tile 0, jg 0) of this lane
    auto loadb = [&](auto tc) {
      constexpr int t = decltype(tc)::value;
      constexpr int gr = t / NJ, jg = t - gr * NJ;
      static_for<0, TG>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (gr * TG + k < NM) br[k][t % DEPTH] = lds_ld<(k * NJ + jg) * 512>(cadr + gr * (TG * NJ * 512));
      });
    };
    if constexpr (EXP == 3) {
#pragma unroll
      for (int u = 0; u < NM; ++u) N[u] += z[u & 3][u];
    } else {
    static_for<0, DEPTH - 1>(loadb);
    double acc[2][TG][4];
    // Row reduction of the TG alpha tiles of group `gr` from accumulator set `st`, in four short stages of
    // independent instructions (one stage per MFMA step of the NEXT group: a tile's update is a chain of ~10
    // dependent fp64 operations and, issued in one piece, idles the matrix pipe for its whole latency).
    double rm01[TG], rm23[TG], rs01[TG], rs23[TG], rm[TG], rnu[TG];
    auto reduce_stage = [&](auto grc, auto stc, auto stagec) {
      constexpr int gq = decltype(grc)::value, st = decltype(stc)::value, stage = decltype(stagec)::value;
      constexpr int ntile = (NM - gq * TG) < TG ? (NM - gq * TG) : TG;
#pragma unroll
      for (int k = 0; k < ntile; ++k) {
        const int u = gq * TG + k;
        // acc[st][k][s] = q for alpha i = 16u + li and row (group (m + s) % 4, index g) of this tile
        if constexpr (EXP == 2) {
          if (stage == 0) N[u] += (acc[st][k][0] + acc[st][k][1]) + (acc[st][k][2] + acc[st][k][3]);
        } else if constexpr (stage == 0) {
          const double q0 = acc[st][k][0], q1 = acc[st][k][1], q2 = acc[st][k][2], q3 = acc[st][k][3];
          rm01[k] = q0 * q1; rm23[k] = q2 * q3; rs01[k] = q0 + q1; rs23[k] = q2 + q3;
          sg[u] |= __double2hiint(q0) | __double2hiint(q1) | __double2hiint(q2) | __double2hiint(q3);   // any q < 0 so far
        } else if constexpr (stage == 1) {
          rm[k] = rm01[k] * rm23[k];
          rnu[k] = __builtin_fma(rs01[k], rm23[k], rs23[k] * rm01[k]);
        } else if constexpr (stage == 2) {
          rnu[k] = __builtin_fma(N[u], rm[k], rnu[k] * P[u]);   // N m + nu P
          rm[k] = P[u] * rm[k];                                 // P m
        } else {
          const int e = __builtin_amdgcn_frexp_exp(rm[k]);
          P[u] = __builtin_amdgcn_frexp_mant(rm[k]);
          const double nn = __builtin_amdgcn_ldexp(rnu[k], -e);
          N[u] = nn;
          E[u] += e;
        }
      }
    };
    static_for<0, NSTEP>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      constexpr int gr = t / NJ, jg = t - gr * NJ, st = gr & 1;
      constexpr int nt = sw4_reads(t);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (t + DEPTH - 1 < NSTEP) loadb(std::integral_constant<int, t + DEPTH - 1>{});
      // in-order returns: everything but the reads of the newer steps t+1 .. t+DEPTH-1 has landed
      constexpr int newer = sw4_reads(t + 1) + sw4_reads(t + 2) + sw4_reads(t + 3);
      static_assert(DEPTH == 4 && newer <= 15, "lgkmcnt is a 4-bit counter");
      if constexpr (nt == 4) lds_wait4<newer>(br[0][t % DEPTH], br[1][t % DEPTH], br[2][t % DEPTH], br[3][t % DEPTH]);
      else lds_wait1<newer>(br[0][t % DEPTH]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int k = 0; k < nt; ++k) {
          if constexpr (jg == 0) acc[st][k][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(z[s][jg], br[k][t % DEPTH], 1.0, 0, 0, 0);
          else acc[st][k][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(z[s][jg], br[k][t % DEPTH], acc[st][k][s], 0, 0, 0);
        }
      // the previous group's tiles are reduced underneath this group's MFMAs (stages at steps 1, 3, 5, 7)
      if constexpr (gr > 0 && (jg & 1) == 1 && jg < 8)
        reduce_stage(std::integral_constant<int, gr - 1>{}, std::integral_constant<int, 1 - st>{},
                     std::integral_constant<int, jg / 2>{});
      if constexpr (t == NSTEP - 1) {   // the last group (one tile) has nothing to hide under
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 4>([&](auto sc) {
          reduce_stage(std::integral_constant<int, gr>{}, std::integral_constant<int, st>{}, sc);
        });
      }
    });
    }
    ntile += 1;
  }

  __syncthreads();
  double *redP = sm;
  double *redR = redP + 4 * NA16;
  int *redE = reinterpret_cast<int *>(redR + 4 * NA16);
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
  for (int i = tid; i < NA16; i += 256) {
    double pv = 1.0, rv = 0.0;
    int ev = 0;
    for (int w = 0; w < 4; ++w) {
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
  if (tid == 0 && nalpha < NA16) po[2 * NA16 - 1] = (double)(cred[0] + cred[1] + cred[2] + cred[3]);
}

// k_sweep4r: the same sweep with GEMM2 through the rank-28 factorisation of its coefficient matrix (cmf_lowrank.hip):
//   t[row][m] = - sum_j z[row][j] U[j][m]        (126 MFMA per 16 rows; A = U blocks broadcast from LDS, B = z as GEMM1 left it)
//   q[row][alpha] = 1 + sum_m t[row][m] W[m][alpha]   (364 MFMA; A = t and its three row rotations, B = W fragments in LDS)
// 814 MFMAs per 16 rows instead of 1260; q differs from the full product by one ulp.  Columns whose factorisation
// was not accepted (lrok == 0) are left to k_sweep4.
// NW waves per workgroup share the LDS tables: with NW = 8 every SIMD holds two waves of the same workgroup and
// one wave's LDS reads, DPP moves and waits run under the other's MFMAs (the tables allow only one workgroup per
// CU, and with a single wave per SIMD every non-MFMA instruction is a bubble in the matrix pipe).
template <int EXP, int NW, int NK>   // NK = rank / 4: 7 (rank 28, lrok == 1) or 9 (rank 36, lrok == 2)
__global__ __launch_bounds__(64 * NW, 1) void k_sweep4r(const float *__restrict__ xt, const uint8_t *__restrict__ mask_t,
                                                    const int32_t *__restrict__ nuse, const double *__restrict__ mu,
                                                    const double *__restrict__ ufrag_g, const double *__restrict__ wfrag2_g,
                                                    const int32_t *__restrict__ lrok, const double *__restrict__ lam,
                                                    const double *__restrict__ wfrag,
                                                    size_t wstride, const int32_t *__restrict__ status,
                                                    const double *__restrict__ alphas, int nalpha, int L, int p,
                                                    int PS, int rows_per_wg, double *__restrict__ part) {
  constexpr int NJ = S4J, NM = S4M, NA16 = NM * 16;
  constexpr bool VG = (NW == 8);   // two waves per SIMD: 256 registers per wave, all of them arch VGPRs (no 'a' operands)
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int NK2 = SF_LR_K2 / 4;          // stride of the global fragment layout (shared by both ranks)
  double *wfr = sm;                          // [NM][NK][64]  W fragments (GEMM2b B operand)
  double *wblk = wfr + NM * NK * 64;         // [NJ ig][NJ jg][16]
  double *ufr = wblk + NJ * NJ * 16;         // [NJ jg][NK][16]  -U blocks (GEMM2a A operand)
  double *mus = ufr + NJ * NK * 16;          // [72]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int c = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
  double *po = part + ((size_t)c * nsplit + split) * 2 * NA16;
  if (status[c] != 0 || lrok[c] != sf_lr_code(NK)) return;   // another instantiation / k_sweep4 takes these columns
  // ---- prologue: three table copies
  {
    auto copy = [&](double *dst, const double *src, int nel) {
      constexpr int NT = 64 * NW;
      int i = tid;
      for (; i + 3 * NT < nel; i += 4 * NT) {
        const double t0 = src[i], t1 = src[i + NT], t2 = src[i + 2 * NT], t3 = src[i + 3 * NT];
        dst[i] = t0; dst[i + NT] = t1; dst[i + 2 * NT] = t2; dst[i + 3 * NT] = t3;
      }
      for (; i < nel; i += NT) dst[i] = src[i];
    };
    // W blocks scaled by 1/sqrt(lam_j): GEMM1 then yields the whitened coordinates y_j/sqrt(lam_j) (unit variance) and
    // z their squares, which is what the row-scaled factorisation of cmf_lowrank.hip multiplies
    for (int i = tid; i < 4 * NJ; i += 64 * NW) mus[i] = (i < p) ? 1.0 / sqrt(lam[(size_t)c * p + i]) : 1.0;   // (mus: scratch here)
    __syncthreads();
    {
      const double *wsrc = wfrag + (size_t)c * wstride;
      for (int i = tid; i < NJ * NJ * 16; i += 64 * NW) wblk[i] = wsrc[i] * mus[4 * ((i >> 4) % NJ) + (i & 3)];
    }
    __syncthreads();
    for (int i = tid; i < 4 * NJ; i += 64 * NW) mus[i] = (i < p) ? mu[(size_t)c * p + i] : 0.0;
    {   // the first NK of the NK2 factor groups of every (jg) / (M) block
      const double *us = ufrag_g + (size_t)c * (NJ * NK2 * 16);
      for (int i = tid; i < NJ * NK * 16; i += 64 * NW) {
        const int blk = i / (NK * 16), r = i - blk * (NK * 16);
        ufr[i] = us[blk * (NK2 * 16) + r];
      }
      const double *ws = wfrag2_g + (size_t)c * (NM * NK2 * 64);
      for (int i = tid; i < NM * NK * 64; i += 64 * NW) {
        const int blk = i / (NK * 64), r = i - blk * (NK * 64);
        wfr[i] = ws[blk * (NK2 * 64) + r];
      }
    }
  }
  __syncthreads();

  double P[NM], N[NM];
  int E[NM], sg[NM];   // sg: OR of the sign words of every q seen (a negative q makes log(q), hence the NLL, NaN in the reference)
#pragma unroll
  for (int u = 0; u < NM; ++u) { P[u] = 1.0; N[u] = 0.0; E[u] = 0; sg[u] = 0; }
  int ntile = 0;
  int nrowok = 0;   // valid rows seen by this lane (lanes with g == 0 cover every row of the wave's tiles once)

  const int rbeg = split * rows_per_wg, rend = min(L, rbeg + rows_per_wg);
  const uint8_t *mp = mask_t + (size_t)c * L;
  const float *xc = xt + (size_t)c * L * PS + NJ * g;
  const double *cf = wfr + lane;
  const double *uf = ufr + 4 * g + (lane & 3);
  const double *wf = wblk + 4 * g + (lane & 3);
  const double qnan = __builtin_nan("");

  float xraw[NJ];
  bool rowok_next;
  auto fetch = [&](int r0, float (&dst)[NJ], bool &ok) {
    const int row = r0 + li;
    ok = (row < rend) && (mp[row < rend ? row : rbeg] != 0);
    const float *xp = xc + (size_t)(ok ? row : rbeg) * PS;
#pragma unroll
    for (int s = 0; s < NJ; s += 2) sf_load2(xp + s, dst[s], dst[s + 1]);
  };
  int r0 = rbeg + 16 * wave;
  if (r0 < rend) fetch(r0, xraw, rowok_next);

  if (EXP == 4) r0 = rend;   // timing experiment: prologue + final reduction only
  for (; r0 < rend; r0 += 16 * NW) {
    const bool rowok = rowok_next;
    nrowok += (rowok && g == 0) ? 1 : 0;
    int opq = 0;
    asm volatile("" : "+v"(opq));  // keeps the loop-invariant LDS operand reads inside the iteration (see k_sweep)
    const double *cfl = cf + opq;
    const double *wfl = wf + opq;
    const double *musl = mus + opq;
    double x[NJ];
#pragma unroll
    for (int s = 0; s < NJ; ++s) {
      const int b = NJ * g + s;
      x[s] = (rowok && b < p) ? (double)xraw[s] - musl[b] : 0.0;
    }
    if (r0 + 16 * NW < rend) fetch(r0 + 16 * NW, xraw, rowok_next);
    // ---- GEMM1: 18 independent chains, one 4-band step at a time.  The A blocks of step s+1 are read (in-order
    //      asm reads, two per two MFMAs) while step s multiplies; left to itself the scheduler sinks every read
    //      next to its use and waits for it: 2 MFMAs per LDS round trip.
    double z[4][NJ];
    double wa[2][NJ];
    const unsigned wadr = (unsigned)(size_t)wfl;  // LDS byte address of this lane's slot in block (0, 0)
    static_for<0, NJ>([&](auto jc) {
      constexpr int jg = decltype(jc)::value;
      z[0][jg] = 0.0;
      wa[0][jg] = lds_ld<jg * 128, VG>(wadr);
    });
    if constexpr (EXP == 1) {
#pragma unroll
      for (int jg = 0; jg < NJ; ++jg) z[0][jg] = x[jg] + wa[0][jg];
    } else
    static_for<0, NJ>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      static_for<0, NJ / 6>([&](auto kc) {
        constexpr int k = decltype(kc)::value * 6;
        lds_wait6<0, VG>(wa[s & 1][k], wa[s & 1][k + 1], wa[s & 1][k + 2], wa[s & 1][k + 3], wa[s & 1][k + 4], wa[s & 1][k + 5]);
      });
      static_for<0, NJ / 2>([&](auto jc) {
        constexpr int jg = decltype(jc)::value * 2;
        if constexpr (s + 1 < NJ) {
          wa[(s + 1) & 1][jg] = lds_ld<((s + 1) * NJ + jg) * 128, VG>(wadr);
          wa[(s + 1) & 1][jg + 1] = lds_ld<((s + 1) * NJ + jg + 1) * 128, VG>(wadr);
        }
        z[0][jg] = __builtin_amdgcn_mfma_f64_4x4x4f64(wa[s & 1][jg], x[s], z[0][jg], 0, 0, 0);
        z[0][jg + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(wa[s & 1][jg + 1], x[s], z[0][jg + 1], 0, 0, 0);
      });
    });
#pragma unroll
    for (int jg = 0; jg < NJ; ++jg) z[0][jg] = z[0][jg] * z[0][jg];
    // ---- GEMM2a: t = -z U, 7 independent chains; the U blocks of step jg+1 are read while step jg multiplies
    double t4[4][NK];
    {
      const unsigned uadr = (unsigned)(size_t)(uf + opq);
      double ua[2][NK];
      static_for<0, NK>([&](auto mc) {
        constexpr int mg = decltype(mc)::value;
        t4[0][mg] = 0.0;
        ua[0][mg] = lds_ld<mg * 128, VG>(uadr);
      });
      static_for<0, NJ>([&](auto jc) {
        constexpr int jg = decltype(jc)::value;
        lds_wait4<0, VG>(ua[jg & 1][0], ua[jg & 1][1], ua[jg & 1][2], ua[jg & 1][3]);
        lds_wait4<0, VG>(ua[jg & 1][3], ua[jg & 1][4], ua[jg & 1][5], ua[jg & 1][6]);
        if constexpr (NK > 7) lds_wait4<0, VG>(ua[jg & 1][NK - 4], ua[jg & 1][NK - 3], ua[jg & 1][NK - 2], ua[jg & 1][NK - 1]);
        static_for<0, NK>([&](auto mc) {
          constexpr int mg = decltype(mc)::value;
          if constexpr (jg + 1 < NJ) ua[(jg + 1) & 1][mg] = lds_ld<((jg + 1) * NK + mg) * 128, VG>(uadr);
          t4[0][mg] = __builtin_amdgcn_mfma_f64_4x4x4f64(ua[jg & 1][mg], z[0][jg], t4[0][mg], 0, 0, 0);
        });
      });
    }
#pragma unroll
    for (int mg = 0; mg < NK; ++mg) {
      const double tv = t4[0][mg];
      t4[1][mg] = dpp_row<0x124>(tv);  // row_ror:4
      t4[2][mg] = dpp_row<0x128>(tv);  // row_ror:8
      t4[3][mg] = dpp_row<0x12C>(tv);  // row_ror:12
    }
    // ---- GEMM2b: q = 1 + t W.  alpha tiles in groups of 4 (16 independent chains), W fragments read DEPTH-1 steps
    //      ahead through a register ring that runs across group boundaries; row reduction as in k_sweep4.
    constexpr int TG = 4, NG = (NM + TG - 1) / TG, NSTEP = NG * NK, DEPTH = 4;
    double br[TG][DEPTH];
    const unsigned cadr = (unsigned)(size_t)cfl;        // fragment (tile 0, jg 0) of this lane
    auto loadb = [&](auto tc) {
      constexpr int t = decltype(tc)::value;
      constexpr int gr = t / NK, jg = t - gr * NK;
      static_for<0, TG>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (gr * TG + k < NM) br[k][t % DEPTH] = lds_ld<(k * NK + jg) * 512, VG>(cadr + gr * (TG * NK * 512));
      });
    };
    if constexpr (EXP == 3) {
#pragma unroll
      for (int u = 0; u < NM; ++u) N[u] += t4[u & 3][u % NK];
    } else {
    static_for<0, DEPTH - 1>(loadb);
    double acc[2][TG][4];
    // Row reduction of the TG alpha tiles of group `gr` from accumulator set `st`, in four short stages of
    // independent instructions (one stage per MFMA step of the NEXT group: a tile's update is a chain of ~10
    // dependent fp64 operations and, issued in one piece, idles the matrix pipe for its whole latency).
    double rm01[TG], rm23[TG], rs01[TG], rs23[TG], rm[TG], rnu[TG];
    auto reduce_stage = [&](auto grc, auto stc, auto stagec) {
      constexpr int gq = decltype(grc)::value, st = decltype(stc)::value, stage = decltype(stagec)::value;
      constexpr int ntile = (NM - gq * TG) < TG ? (NM - gq * TG) : TG;
#pragma unroll
      for (int k = 0; k < ntile; ++k) {
        const int u = gq * TG + k;
        // acc[st][k][s] = q for alpha i = 16u + li and row (group (m + s) % 4, index g) of this tile
        if constexpr (EXP == 2) {
          if (stage == 0) N[u] += (acc[st][k][0] + acc[st][k][1]) + (acc[st][k][2] + acc[st][k][3]);
        } else if constexpr (stage == 0) {
          const double q0 = acc[st][k][0], q1 = acc[st][k][1], q2 = acc[st][k][2], q3 = acc[st][k][3];
          rm01[k] = q0 * q1; rm23[k] = q2 * q3; rs01[k] = q0 + q1; rs23[k] = q2 + q3;
          sg[u] |= __double2hiint(q0) | __double2hiint(q1) | __double2hiint(q2) | __double2hiint(q3);   // any q < 0 so far
        } else if constexpr (stage == 1) {
          rm[k] = rm01[k] * rm23[k];
          rnu[k] = __builtin_fma(rs01[k], rm23[k], rs23[k] * rm01[k]);
        } else if constexpr (stage == 2) {
          rnu[k] = __builtin_fma(N[u], rm[k], rnu[k] * P[u]);   // N m + nu P
          rm[k] = P[u] * rm[k];                                 // P m
        } else {
          const int e = __builtin_amdgcn_frexp_exp(rm[k]);
          P[u] = __builtin_amdgcn_frexp_mant(rm[k]);
          const double nn = __builtin_amdgcn_ldexp(rnu[k], -e);
          N[u] = nn;
          E[u] += e;
        }
      }
    };
    static_for<0, NSTEP>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      constexpr int gr = t / NK, jg = t - gr * NK, st = VG ? 0 : (gr & 1);
      constexpr int nt = sw4r_reads(t, NK);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (t + DEPTH - 1 < NSTEP) loadb(std::integral_constant<int, t + DEPTH - 1>{});
      // in-order returns: everything but the reads of the newer steps t+1 .. t+DEPTH-1 has landed
      constexpr int newer = sw4r_reads(t + 1, NK) + sw4r_reads(t + 2, NK) + sw4r_reads(t + 3, NK);
      static_assert(DEPTH == 4 && newer <= 15, "lgkmcnt is a 4-bit counter");
      if constexpr (nt == 4) lds_wait4<newer, VG>(br[0][t % DEPTH], br[1][t % DEPTH], br[2][t % DEPTH], br[3][t % DEPTH]);
      else lds_wait1<newer, VG>(br[0][t % DEPTH]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int k = 0; k < nt; ++k) {
          if constexpr (jg == 0) acc[st][k][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(t4[s][jg], br[k][t % DEPTH], 1.0, 0, 0, 0);
          else acc[st][k][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(t4[s][jg], br[k][t % DEPTH], acc[st][k][s], 0, 0, 0);
        }
      // the previous group's tiles are reduced underneath this group's MFMAs (stages at steps 1, 2, 4, 5)
      if constexpr (!VG && gr > 0 && (jg == 1 || jg == 2 || jg == 4 || jg == 5))
        reduce_stage(std::integral_constant<int, gr - 1>{}, std::integral_constant<int, 1 - st>{},
                     std::integral_constant<int, (jg == 1 ? 0 : (jg == 2 ? 1 : (jg == 4 ? 2 : 3)))>{});
      if constexpr (VG && jg == NK - 1 && t != NSTEP - 1) {   // one accumulator set: reduce at once (the SIMD's other wave fills the pipe)
        static_for<0, 4>([&](auto sc) {
          reduce_stage(std::integral_constant<int, gr>{}, std::integral_constant<int, 0>{}, sc);
        });
      }
      if constexpr (t == NSTEP - 1) {   // the last group (one tile) has nothing to hide under
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 4>([&](auto sc) {
          reduce_stage(std::integral_constant<int, gr>{}, std::integral_constant<int, st>{}, sc);
        });
      }
    });
    }
    ntile += 1;
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

constexpr size_t sw4r_lds(int nk) { return ((size_t)S4M * nk * 64 + S4J * S4J * 16 + S4J * nk * 16 + 4 * S4J) * sizeof(double); }
constexpr size_t SW4_LDS = ((size_t)S4M * S4J * 64 + S4J * S4J * 16 + 4 * S4J + S4J) * sizeof(double);   // c fragments, W blocks, mean, 18 zeros

template <int EXP>
int launch_sweep4_t(const float *xt, const uint8_t *mask_t, const int32_t *nuse, const double *mu, const double *lam,
                    const double *wfrag, size_t wstride, const int32_t *status, const double *alphas, const SfGeom &g,
                    int nsplit, double *part, hipStream_t st, const int32_t *lrok = nullptr) {
  if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_sweep4<EXP>), SW4_LDS)) return rc;
  int rows = sf_cdiv(g.lines, nsplit);
  rows = (rows + 63) / 64 * 64;
  hipLaunchKernelGGL(k_sweep4<EXP>, dim3(g.ncols, nsplit), dim3(256), SW4_LDS, st, xt, mask_t, nuse, mu, lam, wfrag,
                     wstride, status, alphas, g.nalpha, g.lines, g.p, g.ps, rows, part, lrok);
  SF_LAUNCH_CHECK("k_sweep4");
  return 0;
}

int launch_sweep4r(const float *xt, const uint8_t *mask_t, const int32_t *nuse, const double *mu, const double *ufrag,
                   const double *wfrag2, const int32_t *lrok, const double *lam, const double *wfrag, size_t wstride, const int32_t *status,
                   const double *alphas, const SfGeom &g, int nsplit, double *part, hipStream_t st) {
  constexpr int NK1 = SF_LR_K / 4, NK2 = SF_LR_K2 / 4;
  if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_sweep4r<0, 8, NK1>), sw4r_lds(NK1))) return rc;
  if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_sweep4r<0, 4, NK1>), sw4r_lds(NK1))) return rc;
  if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_sweep4r<0, 4, NK2>), sw4r_lds(NK2))) return rc;
  int rows = sf_cdiv(g.lines, nsplit);
  rows = (rows + 127) / 128 * 128;
#define SW4R_ARGS xt, mask_t, nuse, mu, ufrag, wfrag2, lrok, lam, wfrag, wstride, status, alphas, g.nalpha, g.lines, g.p, g.ps, rows, part
  const int form = sf_tune().sweep4_form;
  if (form != 1) {   // the streamed kernel (round 3), both ranks
    if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_sweep4s<NK1>), SwS<NK1>::lds_bytes())) return rc;
    const int sfast = sf_tune().sweep_grid != 1;
    const dim3 grid = sfast ? dim3(nsplit, g.ncols) : dim3(g.ncols, nsplit);
    if (form == 0 || form == 3 || form == 5) hipLaunchKernelGGL((k_sweep4s<NK1>), grid, dim3(512), SwS<NK1>::lds_bytes(), st, SW4R_ARGS, sfast);
    if (form == 0) {   // rank 24 (lrok == 3; the factorisation offers it only to this form: sf_launch_sweep4)
      constexpr int NK0 = SF_LR_K0 / 4;
      if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_sweep4s<NK0>), SwS<NK0>::lds_bytes())) return rc;
      hipLaunchKernelGGL((k_sweep4s<NK0>), grid, dim3(512), SwS<NK0>::lds_bytes(), st, SW4R_ARGS, sfast);
    }
    if (form == 4) {   // renormalisation after every tile (the form of the round's first half): A/B and the bit-identity test
      if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_sweep4s<NK1, 0, 1>), SwS<NK1>::lds_bytes())) return rc;
      hipLaunchKernelGGL((k_sweep4s<NK1, 0, 1>), grid, dim3(512), SwS<NK1>::lds_bytes(), st, SW4R_ARGS, sfast);
    }
#ifdef SF_SWEEP_EXPERIMENTS
#define SW4S_EXP(E) if (form == 100 + E) { \
      if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_sweep4s<NK1, E>), SwS<NK1>::lds_bytes())) return rc; \
      hipLaunchKernelGGL((k_sweep4s<NK1, E>), grid, dim3(512), SwS<NK1>::lds_bytes(), st, SW4R_ARGS, sfast); }
    SW4S_EXP(1) SW4S_EXP(2) SW4S_EXP(4) SW4S_EXP(5) SW4S_EXP(8) SW4S_EXP(16) SW4S_EXP(31) SW4S_EXP(32) SW4S_EXP(64) SW4S_EXP(96)
#endif
    SF_LAUNCH_CHECK("k_sweep4s");
    // rank 36 (lrok == 2): the streamed kernel fits two waves per SIMD there too (250 registers; round 2's k_sweep4r needed
    // one wave per SIMD for its wider t copies): 2.90 against 3.41 ms per stage-5 call on 256 rank-36 columns
    // (tools/tune_sweep_rank36.py); sf_debug_set(20, 3) keeps k_sweep4r<0,4,9> for these columns
    if (form != 3) {
      if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_sweep4s<NK2>), SwS<NK2>::lds_bytes())) return rc;
      hipLaunchKernelGGL((k_sweep4s<NK2>), grid, dim3(512), SwS<NK2>::lds_bytes(), st, SW4R_ARGS, sfast);
    } else {
      hipLaunchKernelGGL((k_sweep4r<0, 4, NK2>), dim3(g.ncols, nsplit), dim3(256), sw4r_lds(NK2), st, SW4R_ARGS);
    }
    SF_LAUNCH_CHECK("k_sweep4s(rank 36)");
    return 0;
  }
  if (sf_tune().sweep4r_waves == 4)
    hipLaunchKernelGGL((k_sweep4r<0, 4, NK1>), dim3(g.ncols, nsplit), dim3(256), sw4r_lds(NK1), st, SW4R_ARGS);
  else
    hipLaunchKernelGGL((k_sweep4r<0, 8, NK1>), dim3(g.ncols, nsplit), dim3(512), sw4r_lds(NK1), st, SW4R_ARGS);
  SF_LAUNCH_CHECK("k_sweep4r");
  // rank 36 (lrok == 2): one wave per SIMD (the wider t registers do not fit two)
  hipLaunchKernelGGL((k_sweep4r<0, 4, NK2>), dim3(g.ncols, nsplit), dim3(256), sw4r_lds(NK2), st, SW4R_ARGS);
#undef SW4R_ARGS
  SF_LAUNCH_CHECK("k_sweep4r");
  return 0;
}

}  // namespace

int sf_launch_wfrag4(const double *evec, const double *d, const SfGeom &g, size_t wstride, double *wfrag, hipStream_t st) {
  const int nj = sf_sw4_groups(g.p), nje = nj + (nj & 1);
  hipLaunchKernelGGL(k_wfrag4, dim3(g.ncols), dim3(256), 0, st, evec, d, g.p, nj, nje, wstride, wfrag);
  SF_LAUNCH_CHECK("k_wfrag4");
  return 0;
}

int sf_launch_sweep4(const float *xt, const uint8_t *mask_t, const int32_t *nuse, const double *mu, const double *lam,
                     const double *wfrag, size_t wstride, const int32_t *status, const double *alphas, const SfGeom &g,
                     int nsplit, double *part, int variant, void *lr_scratch, hipStream_t st, const int32_t **lrok_out) {
#define SW4_ARGS xt, mask_t, nuse, mu, lam, wfrag, wstride, status, alphas, g, nsplit, part, st
  const int nj = sf_sw4_groups(g.p), nje = nj + (nj & 1);
  if ((variant == 0 && lr_scratch) || nj > S4J) {
    // rank-factored sweep for the columns whose factorisation is accepted, the full-rank kernel for the rest
    // (it returns at once for the others: a column is swept by exactly one of the two)
    if (!lr_scratch) { sf_set_error("sf_launch_sweep4: windows of 21 / 24 band groups need the rank-factorisation scratch"); return -2; }
    char *base = reinterpret_cast<char *>(lr_scratch);
    double *ufrag = reinterpret_cast<double *>(base);
    double *wfrag2 = reinterpret_cast<double *>(base + sf_align((size_t)g.ncols * nje * (SF_LR_K2 / 4) * 16 * sizeof(double)));
    int32_t *lrok = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(wfrag2) +
                                                sf_align((size_t)g.ncols * S4M * (SF_LR_K2 / 4) * 64 * sizeof(double)));
    if (lrok_out) *lrok_out = lrok;
    // (the rank-24 tier exists in the streamed kernel only: the debug forms of sf_debug_set(20, .) sweep ranks 28 / 36; form 5 is
    //  the default with that tier switched off -- what the bit-for-bit comparisons against forms 1 and 4 run)
    if (int rc = sf_launch_lowrank(lam, nuse, status, alphas, g, ufrag, wfrag2, lrok, st, nj > S4J || sf_tune().sweep4_form == 0)) return rc;
    if (nj == 21)
      return sf_launch_sweep4s_21(xt, mask_t, nuse, mu, ufrag, wfrag2, lrok, lam, wfrag, wstride, status, alphas, g, nsplit, part, st);
    if (nj == 24)
      return sf_launch_sweep4s_24(xt, mask_t, nuse, mu, ufrag, wfrag2, lrok, lam, wfrag, wstride, status, alphas, g, nsplit, part, st);
    if (int rc = launch_sweep4r(xt, mask_t, nuse, mu, ufrag, wfrag2, lrok, lam, wfrag, wstride, status, alphas, g, nsplit, part, st))
      return rc;
    return launch_sweep4_t<0>(SW4_ARGS, lrok);
  }
#ifdef SF_SWEEP_EXPERIMENTS
  if (variant == 11) return launch_sweep4_t<1>(SW4_ARGS);
  if (variant == 12) return launch_sweep4_t<2>(SW4_ARGS);
  if (variant == 13) return launch_sweep4_t<3>(SW4_ARGS);
  if (variant == 14) return launch_sweep4_t<4>(SW4_ARGS);
#endif
  return launch_sweep4_t<0>(SW4_ARGS);
#undef SW4_ARGS
}
