// Stage 5, production window: the LOO sweep on v_mfma_f64_4x4x4_f64 (see cmf_loocv.hip for the algorithm and for
// the 16x16x4 kernels that serve every other window).  Built with -mllvm -amdgpu-mfma-vgpr-form=1: the results of
// both products feed VALU code (square, row reduction), and every VALU instruction stalls the 4x4x4 fp64 MFMA
// stream (tools/microbench/mix4.hip: +4.5 cycles per instruction on a 16.5-cycle slot), so AGPR accumulators and
// their v_accvgpr_read copies are pure loss here.
#include "cmf_common.h"
#include <type_traits>


namespace {

__device__ __forceinline__ double shfl_xor_d(double v, int m) { return __shfl_xor(v, m, 64); }

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
constexpr int S4J = SF_SW4_NJ, S4M = SF_SW4_NM;
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

// windows of 21 / 24 band groups (CO2: p = 83): the streamed kernel only -- rank 28, and rank 36 where its tables fit the LDS
// (NJ = 21: 145 KB; NJ = 24: 162 KB + the mean: does not fit, k_lowrank<24> refuses those columns).  The full-rank 4x4x4
// kernel has no room for its 13 x NJ x 64 coefficient fragments beside the W blocks beyond NJ = 18: a refused column is
// swept by the 16x16x4 kernel (sf_launch_loocv).
template <int NJ>
int launch_sweep4s_nj(const float *xt, const uint8_t *mask_t, const int32_t *nuse, const double *mu, const double *ufrag,
                      const double *wfrag2, const int32_t *lrok, const double *lam, const double *wfrag, size_t wstride,
                      const int32_t *status, const double *alphas, const SfGeom &g, int nsplit, double *part, hipStream_t st) {
  constexpr int NK1 = SF_LR_K / 4, NK2 = SF_LR_K2 / 4;
  int rows = sf_cdiv(g.lines, nsplit);
  rows = (rows + 127) / 128 * 128;
  const dim3 grid(nsplit, g.ncols);
  using S1 = SwS<NK1, NJ>;
  if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_sweep4s<NK1, 0, 4, NJ>), S1::lds_bytes())) return rc;
  hipLaunchKernelGGL((k_sweep4s<NK1, 0, 4, NJ>), grid, dim3(512), S1::lds_bytes(), st, xt, mask_t, nuse, mu, ufrag, wfrag2, lrok, lam,
                     wfrag, wstride, status, alphas, g.nalpha, g.lines, g.p, g.ps, rows, part, 1);
  SF_LAUNCH_CHECK("k_sweep4s");
  {
    constexpr int NK0 = SF_LR_K0 / 4;
    using S0 = SwS<NK0, NJ>;
    if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_sweep4s<NK0, 0, 4, NJ>), S0::lds_bytes())) return rc;
    hipLaunchKernelGGL((k_sweep4s<NK0, 0, 4, NJ>), grid, dim3(512), S0::lds_bytes(), st, xt, mask_t, nuse, mu, ufrag, wfrag2, lrok, lam,
                       wfrag, wstride, status, alphas, g.nalpha, g.lines, g.p, g.ps, rows, part, 1);
    SF_LAUNCH_CHECK("k_sweep4s(rank 24)");
  }
  if constexpr (SwS<NK2, NJ>::lds_bytes() <= 160 * 1024) {
    using S2 = SwS<NK2, NJ>;
    if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_sweep4s<NK2, 0, 4, NJ>), S2::lds_bytes())) return rc;
    hipLaunchKernelGGL((k_sweep4s<NK2, 0, 4, NJ>), grid, dim3(512), S2::lds_bytes(), st, xt, mask_t, nuse, mu, ufrag, wfrag2, lrok,
                       lam, wfrag, wstride, status, alphas, g.nalpha, g.lines, g.p, g.ps, rows, part, 1);
    SF_LAUNCH_CHECK("k_sweep4s(rank 36)");
  }
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
      return launch_sweep4s_nj<21>(xt, mask_t, nuse, mu, ufrag, wfrag2, lrok, lam, wfrag, wstride, status, alphas, g, nsplit, part, st);
    if (nj == 24)
      return launch_sweep4s_nj<24>(xt, mask_t, nuse, mu, ufrag, wfrag2, lrok, lam, wfrag, wstride, status, alphas, g, nsplit, part, st);
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
