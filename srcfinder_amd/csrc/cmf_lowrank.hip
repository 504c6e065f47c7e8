// Low-rank form of the LOO sweep's coefficient matrix (production window, used by k_sweep4r in cmf_loocv4.hip).
//
// GEMM2 of the sweep evaluates  q_k(alpha_i) = 1 - sum_j z_kj B_ji,   B_ji = beta_i / (n beta_i lam_j + alpha_i),
// for every row k: 72 x 208 coefficients.  B is a Cauchy matrix in disguise, B_ji = (1/(n-1)) / (x_j + g_i) with
// x_j = n lam_j/(n-1), g_i = alpha_i/(1-alpha_i), and its numerical rank is ~25: the singular values fall by a decade
// every 1.6 (sigma_26/sigma_0 < 1e-16 on flightline-like spectra).  So per column
//        B' = diag(lam) B = U W + E,   U (72 x K), W (K x 208) with orthonormal rows,   K = 28 (36),   |E| <= ~1e-15 |B'|,
// and the sweep multiplies z by U (72 x 28) and then by W (28 x 208) instead of by B: 44 % fewer flops in the
// product that is 3/4 of the kernel, with q changing by one ulp (numpy model: max |dq| = 1.1e-16, identical NLL
// argmin on every test spectrum).
//
// The factorisation is a Householder QR with column pivoting of B^T (208 x 72), stopped after K steps:
//   B^T P = Q R   =>   B = (P R_K^T) (Q_K^T) + E,   |E| = the largest remaining column norm,
// backward stable whatever the conditioning (B is numerically singular by design -- no Gram matrix, no
// inverse).  One 576-thread workgroup per column, 8 lanes per column with DPP reductions, the matrix in their registers.  A column whose remaining norm
// after K steps is not below 1e-14 |R_00| (q then moves by <= 72 x 1e-14 x |R_00| ~ 5e-16; the rounding floor is ~7e-16 |R_00|), or whose
// correlation matrix is not safely positive definite (condition > 1e10), is flagged and swept by the full-rank kernel.
#include "cmf_common.h"

namespace {

constexpr int LR_NA = 16 * SF_SW4_NM;      // 208
constexpr int LR_K0 = SF_LR_K0;            // 24: enough for a noise-floor cluster + a few signal directions (round 5)
constexpr int LR_K = SF_LR_K;              // 28: the fast rank
constexpr int LR_K2 = SF_LR_K2;            // 36: second chance for spectra with a wider eigenvalue range
constexpr int LR_NK2 = LR_K2 / 4;          // fragment layout stride (both ranks share the 36-wide layout)
constexpr int LR_TPC = 8;                  // lanes per column in the update (8-lane DPP reductions, no LDS partials)
// threads: 8 lanes per column of B^T (4 NJ columns: 576 at NJ = 18, 672 at 21, 768 at 24); the Q_K phase uses the first 576
constexpr int LR_PK = 5;                   // row slots (of 8 rows) that can touch the diagonal: 8 * 5 = 40 > K2
static_assert(LR_TPC * LR_PK >= LR_K2 + 1 && LR_NA % LR_TPC == 0, "row slots");

template <int CTRL>
__device__ __forceinline__ double lr_dpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lr_sum8(double v) {   // all 8 lanes of an aligned group get the sum
  v += lr_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
  v += lr_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
  v += lr_dpp<0x141>(v);   // row_half_mirror
  return v;
}
// one exchange of an 8-lane argmax: keep the larger value, on equal values the lower index
template <int CTRL>
__device__ __forceinline__ void lr_argmax_step(double &v, int &i) {
  const double ov = lr_dpp<CTRL>(v);
  const int oi = __builtin_amdgcn_update_dpp(0, i, CTRL, 0xF, 0xF, true);
  const bool take = (ov > v) || (ov == v && oi < i);
  v = take ? ov : v;
  i = take ? oi : i;
}
__device__ __forceinline__ double lr_sum16(double v) {
  v = lr_sum8(v);
  v += lr_dpp<0x140>(v);   // row_mirror
  return v;
}

// Round 3: the matrix lives in REGISTERS.  An 8-lane group owns one column (26 values per lane) for the whole
// factorisation; what the groups share is the reflector of the step, which the pivot's group writes once -- scaled, with its
// unit diagonal and the zeros above it -- into row s of the reflector store V[36][208] (60 KB of LDS instead of the
// 120 KB matrix: two workgroups per CU, 598 columns in two rounds instead of three, and a step moves a third of the
// LDS bytes it used to).  Q_K is formed in registers as well (16 lanes per column, reflectors read from V).  Same
// arithmetic in the same order as the LDS-resident form of round 2: the fragments are bit-identical.
template <int NJ>
__global__ __launch_bounds__(32 * NJ)
void k_lowrank(const double *__restrict__ lam, const int32_t *__restrict__ nuse,
               const int32_t *__restrict__ status, const double *__restrict__ alphas,
               int nalpha, int p, double *__restrict__ ufrag, double *__restrict__ wfrag,
               int32_t *__restrict__ lrok, int allow_k0) {
  constexpr int LR_P = 4 * NJ, LR_NT = 8 * LR_P, NJE = NJ + (NJ & 1);   // NJE: eigen groups of the sweep's pair layout (even)
  static_assert(LR_NT >= 576 && LR_NT <= 1024, "the Q_K phase needs 36 x 16 threads");
  __shared__ double V[LR_K2][LR_NA];            // reflector s: 0 above row s, 1 at row s, x_i * scale below
  __shared__ double cnorm[LR_P];                // squared norms of the remaining columns, rows >= current (-1: pivoted)
  __shared__ double cnorm2[LR_P];               // the same over the rows below the current one
  __shared__ double tau_s[LR_K2];
  __shared__ double s_al[LR_NA], s_be[LR_NA], s_lam[LR_P];
  __shared__ double s_r00;
  constexpr int NV = LR_NA / LR_TPC;            // 26 values of its column per lane
  const int c = blockIdx.x, tid = threadIdx.x;
  if (status[c] != 0) {
    if (tid == 0) lrok[c] = 0;
    return;
  }
  const double n = (double)nuse[c];
  // ---- B^T: a[j][i] = beta_i / (n beta_i lam_j + alpha_i)   (zero for the padding alpha / eigen indices).
  //      alpha, beta and lam are staged in LDS first: two dependent global loads per entry would dominate the kernel.
  for (int i = tid; i < LR_NA; i += LR_NT) {
    const double a = (i < nalpha) ? alphas[i] : 1.0;
    s_al[i] = a;
    s_be[i] = (i < nalpha) ? (1.0 - a) / (n - 1.0) : 0.0;
  }
  for (int j = tid; j < LR_P; j += LR_NT) s_lam[j] = (j < p) ? lam[(size_t)c * p + j] : 1.0;
  if (tid == 0) s_r00 = 0.0;
  __syncthreads();
  const int col = tid >> 3, sub = tid & 7;   // column `col` (0..71), rows sub, sub+8, ...
  // (row-scaled: what is factored is B'_ji = lam_j B_ji, all entries in [0, 1/n); the sweep feeds it the whitened
  //  squares z_j / lam_j -- of order one whatever the spectrum -- so the factorisation error is relative to what each
  //  eigen-direction actually contributes to q, not to the largest entry of B, which is 1/(n lam_min).)
  double av[NV];
  {
    const double lj = s_lam[col];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = sub + LR_TPC * k;
      const double a = s_al[i], beta = s_be[i];
      const double v = lj * beta / ((n * beta) * lj + a);
      av[k] = (i < nalpha && col < p) ? v : 0.0;
    }
  }
  bool lam_ok = true;
  {
    double lmin = 1.7976931348623157e308, lmax = 0.0;
    for (int j = 0; j < p; ++j) { lmin = fmin(lmin, s_lam[j]); lmax = fmax(lmax, s_lam[j]); }
    lam_ok = (lmin > 1e-10 * lmax) && (lmax <= 1.7976931348623157e308);   // positive definite, condition < 1e10
  }
  {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const double v = av[k];
      s0 = __builtin_fma(v, v, s0);
      if (sub + LR_TPC * k > 0) s1 = __builtin_fma(v, v, s1);
    }
    s0 = lr_sum8(s0);
    s1 = lr_sum8(s1);
    if (sub == 0) { cnorm[col] = s0; cnorm2[col] = s1; }
  }
  __syncthreads();
  // Two barriers per step.  Every wave finds the pivot for itself (same data, same tie-break); the pivot's group forms the
  // reflector's scalars from the carried norms -- cnorm = rows >= s, cnorm2 = rows > s, both summed afresh by the update
  // of the previous step, so nothing is obtained by subtraction -- and publishes the reflector; the others apply it.
  int mypos = -1;                               // step at which this group's column was pivoted
  int kuse = 0;                                 // accepted rank: LR_K, LR_K2 or 0 (full-rank sweep)
  for (int s = 0; s < LR_K2; ++s) {
    if ((s == LR_K0 && allow_k0) || s == LR_K) {   // rank 24 / 28 reached: is the trailing block already at the rounding floor?  (same answer in every thread)
      double bn = 0.0;
      for (int j = 0; j < LR_P; ++j) bn = fmax(bn, cnorm[j]);
      const double r00 = s_r00;
      if (sqrt(bn) <= 1e-14 * r00 && r00 > 0.0) { kuse = s; break; }
    }
    // pivot = the remaining column of largest norm (lowest index on ties; pivoted columns carry -1).  Every 8-lane
    // group scans all 72 candidates (9 per lane) and finishes with three DPP exchanges: no cross-wave traffic, no
    // ds_bpermute chain (six dependent LDS-crossbar round trips were the longest part of the step).
    double bv = cnorm[sub];
    int bi = sub;
#pragma unroll
    for (int k = 1; k < (LR_P + LR_TPC - 1) / LR_TPC; ++k) {
      const int jc = sub + LR_TPC * k;
      const double v2 = (LR_P % LR_TPC == 0 || jc < LR_P) ? cnorm[jc < LR_P ? jc : 0] : -2.0;
      if (v2 > bv) { bv = v2; bi = jc; }
    }
    lr_argmax_step<0xB1>(bv, bi);
    lr_argmax_step<0x4E>(bv, bi);
    lr_argmax_step<0x141>(bv, bi);
    const int pv = bi;
    if (col == pv) {
      // this group's column is the pivot: alpha = its entry in row s (held by lane s % 8, slot s / 8)
      double alpha = 0.0;
#pragma unroll
      for (int k = 0; k < LR_PK; ++k) alpha = (sub + LR_TPC * k == s) ? av[k] : alpha;
      alpha = lr_sum8(alpha);                                    // exactly one lane holds it: the sum is the value
      const double xn2 = cnorm2[pv];
      double tau = 0.0, scale = 0.0, betah = alpha;
      if (xn2 > 0.0) {
        const double h2 = __builtin_fma(alpha, alpha, xn2);
        double y = __builtin_amdgcn_rsq(h2);                     // 1/sqrt(h2): hardware estimate + two Newton steps
        y = y * __builtin_fma(-0.5 * h2 * y, y, 1.5);
        y = y * __builtin_fma(-0.5 * h2 * y, y, 1.5);
        betah = -copysign(h2 * y, alpha);
        const double den = alpha - betah;                        // same sign as alpha, |den| >= |alpha|: no cancellation
        double rd = __builtin_amdgcn_rcp(den);
        rd = rd * __builtin_fma(-den, rd, 2.0);
        rd = rd * __builtin_fma(-den, rd, 2.0);
        scale = rd;
        tau = den * (y * (alpha < 0.0 ? -1.0 : 1.0));            // (betah - alpha)/betah = den / (sign(alpha) sqrt(h2))
      }
      // v = x * scale (v_s = 1); rows >= 40 are below every diagonal position (s < 36): only the first five row slots
      // need the predicates.  R_ss = betah takes the diagonal entry's place in the column.
#pragma unroll
      for (int k = 0; k < LR_PK; ++k) {
        const int i = sub + LR_TPC * k;
        V[s][i] = (i > s) ? av[k] * scale : ((i == s) ? 1.0 : 0.0);
        av[k] = (i == s) ? betah : av[k];
      }
#pragma unroll
      for (int k = LR_PK; k < NV; ++k) V[s][sub + LR_TPC * k] = av[k] * scale;
      if (sub == 0) {
        tau_s[s] = tau;
        if (s == 0) s_r00 = fabs(betah);
      }
      mypos = s;
    }
    __syncthreads();   // the reflector is published
    if (col == pv) {
      if (sub == 0) cnorm[pv] = -1.0;   // (after the barrier: the other groups read the norms for their pivot search before it)
    } else if (mypos < 0) {
      // ---- apply H = I - tau v v^T; squared norms of what is left in rows > s and in rows > s+1
      // (the reflector is read from LDS twice rather than kept: 26 + 26 values per lane would not leave room for two
      //  workgroups per CU)
      const double tau = tau_s[s];
      const double *vs = V[s] + sub;
      double w = 0.0;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int i = sub + LR_TPC * k;
        const double a = (k >= LR_PK || i >= s) ? av[k] : 0.0;
        w = __builtin_fma(vs[LR_TPC * k], a, w);
      }
      const double wt = tau * lr_sum8(w);
      double nn = 0.0, nn2 = 0.0;
#pragma unroll
      for (int k = 0; k < LR_PK; ++k) {
        const int i = sub + LR_TPC * k;
        if (i >= s) {
          const double nv = __builtin_fma(-wt, vs[LR_TPC * k], av[k]);
          av[k] = nv;
          if (i > s) nn = __builtin_fma(nv, nv, nn);
          if (i > s + 1) nn2 = __builtin_fma(nv, nv, nn2);
        }
      }
      double nb = 0.0;
#pragma unroll
      for (int k = LR_PK; k < NV; ++k) {
        const double nv = __builtin_fma(-wt, vs[LR_TPC * k], av[k]);
        av[k] = nv;
        nb = __builtin_fma(nv, nv, nb);
      }
      nn += nb;
      nn2 += nb;
      nn = lr_sum8(nn);
      nn2 = lr_sum8(nn2);
      if (sub == 0) { cnorm[col] = nn; cnorm2[col] = nn2; }
    }
    __syncthreads();
  }
  if (kuse == 0) {   // what is left after K2 steps
    double bn = 0.0;
    for (int j = 0; j < LR_P; ++j) bn = fmax(bn, cnorm[j]);
    const double r00 = s_r00;
    if (sqrt(bn) <= 1e-14 * r00 && r00 > 0.0) kuse = LR_K2;
  }
  if (!lam_ok) kuse = 0;   // (near-)singular or indefinite correlation matrix: the full-rank kernel
  if (NJ == 24 && kuse == LR_K2) kuse = 0;   // the rank-36 tables of 24 band groups do not fit the LDS (cmf_loocv4.hip)
  if (tid == 0) lrok[c] = (kuse == 0) ? 0 : sf_lr_code(kuse / 4);
  if (kuse == 0) return;
  // ---- U fragments: ufrag[(jg*NK + mg)*16 + 4q + n] = -U[4jg+q][4mg+n];  U[j][m] = R[m][column j]: rows m <= its step
  //      of a pivoted column (below them sat its reflector), all K rows of the others -- straight from the registers
  {
    double *uo = ufrag + (size_t)c * (NJE * LR_NK2 * 16);
    if (NJE > NJ)                                  // the padding eigen group of an odd NJ multiplies nothing
      for (int i = tid; i < LR_NK2 * 16; i += LR_NT) uo[NJ * LR_NK2 * 16 + i] = 0.0;
    const int jg = col >> 2, q = col & 3;
#pragma unroll
    for (int k = 0; k < LR_PK; ++k) {
      const int m = sub + LR_TPC * k;
      if (m < LR_K2) {
        const double r = (m < kuse && (mypos < 0 || m <= mypos)) ? av[k] : 0.0;
        uo[(jg * LR_NK2 + (m >> 2)) * 16 + 4 * q + (m & 3)] = -r;
      }
    }
  }
  // ---- Q_K = H_0 ... H_{K-1} [I_K; 0]: column m in the registers of a 16-lane group (13 values per lane), reflectors
  //      from V.  The columns are independent: no barrier between reflectors.
  constexpr int NQ = LR_NA / 16;
  const int qcol = tid >> 4, qsub = tid & 15;   // 36 columns x 16 lanes = 576 threads
  double qv[NQ];
#pragma unroll
  for (int k = 0; k < NQ; ++k) qv[k] = (qsub + 16 * k == qcol) ? 1.0 : 0.0;
  if (qcol < kuse && tid < 576) {
    for (int s = kuse - 1; s >= 0; --s) {
      const double tau = tau_s[s];
      double vv[NQ];
      double w = 0.0;
#pragma unroll
      for (int k = 0; k < NQ; ++k) {
        const int i = qsub + 16 * k;
        vv[k] = V[s][i];
        const double qm = (i >= s) ? qv[k] : 0.0;
        w = __builtin_fma(vv[k], qm, w);
      }
      const double wt = tau * lr_sum16(w);
#pragma unroll
      for (int k = 0; k < NQ; ++k) {
        const int i = qsub + 16 * k;
        qv[k] = (i >= s) ? __builtin_fma(-wt, vv[k], qv[k]) : qv[k];
      }
    }
  }
  // ---- W fragments: wfrag[(M*NK + mg)*64 + lane], lane = 16q + 4mm + n  ->  W[4mg+q][16M + 4mm + n] = Q[alpha][m]
  if (tid < 576) {
    double *wo = wfrag + (size_t)c * (SF_SW4_NM * LR_NK2 * 64);
    const int mg = qcol >> 2, q = qcol & 3;
#pragma unroll
    for (int k = 0; k < NQ; ++k)    // alpha = 16 k + qsub: alpha tile M = k
      wo[(k * LR_NK2 + mg) * 64 + 16 * q + qsub] = (qcol < kuse) ? qv[k] : 0.0;
  }
}

}  // namespace

size_t sf_lowrank_bytes(const SfGeom &g) {
  const int nj = sf_sw4_groups(g.p) ? sf_sw4_groups(g.p) : SF_SW4_NJ, nje = nj + (nj & 1);
  return sf_align((size_t)g.ncols * nje * LR_NK2 * 16 * sizeof(double)) +
         sf_align((size_t)g.ncols * SF_SW4_NM * LR_NK2 * 64 * sizeof(double)) + sf_align((size_t)g.ncols * sizeof(int32_t));
}

int sf_launch_lowrank(const double *lam, const int32_t *nuse, const int32_t *status, const double *alphas, const SfGeom &g,
                      double *ufrag, double *wfrag, int32_t *lrok, hipStream_t st, int allow_k0) {
  if (g.s4 == 21)
    hipLaunchKernelGGL(k_lowrank<21>, dim3(g.ncols), dim3(32 * 21), 0, st, lam, nuse, status, alphas, g.nalpha, g.p, ufrag, wfrag, lrok, allow_k0);
  else if (g.s4 == 24)
    hipLaunchKernelGGL(k_lowrank<24>, dim3(g.ncols), dim3(32 * 24), 0, st, lam, nuse, status, alphas, g.nalpha, g.p, ufrag, wfrag, lrok, allow_k0);
  else
    hipLaunchKernelGGL(k_lowrank<18>, dim3(g.ncols), dim3(32 * 18), 0, st, lam, nuse, status, alphas, g.nalpha, g.p, ufrag, wfrag, lrok, allow_k0);
  SF_LAUNCH_CHECK("k_lowrank");
  return 0;
}

// test hook (tests/test_cmf_gpu.py): the factorisation of one launch, fragments as the sweep reads them
extern "C" int sf_debug_lowrank(const double *lam, const int32_t *nuse, const int32_t *status, const double *alphas,
                                int nalpha, int p, int ncols, double *ufrag, double *wfrag, int32_t *lrok, void *stream) {
  const SfGeom g = sf_geom(64, p, ncols, nalpha);
  if (!sf_sw4_groups(p) || g.nu != SF_SW4_NM) { sf_set_error("sf_debug_lowrank: windows of 69..72, 81..84, 93..96 bands only"); return -2; }
  return sf_launch_lowrank(lam, nuse, status, alphas, g, ufrag, wfrag, lrok, (hipStream_t)stream);
}
