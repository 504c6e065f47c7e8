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
// inverse).  One 576-thread workgroup per column, the matrix in LDS, 8 lanes per column with DPP reductions.  A column whose remaining norm
// after K steps is not below 1e-14 |R_00| (q then moves by <= 72 x 1e-14 x |R_00| ~ 5e-16; the rounding floor is ~7e-16 |R_00|), or whose
// correlation matrix is not safely positive definite (condition > 1e10), is flagged and swept by the full-rank kernel.
#include "cmf_common.h"

namespace {

constexpr int LR_P = 4 * SF_SW4_NJ;        // 72
constexpr int LR_NA = 16 * SF_SW4_NM;      // 208
constexpr int LR_K = SF_LR_K;              // 28: the fast rank
constexpr int LR_K2 = SF_LR_K2;            // 36: second chance for spectra with a wider eigenvalue range
constexpr int LR_NK2 = LR_K2 / 4;          // fragment layout stride (both ranks share the 36-wide layout)
constexpr int LR_LDA = LR_NA + 1;          // column stride in LDS (odd: threads on different columns, same row)
constexpr int LR_TPC = 8;                  // lanes per column in the update (8-lane DPP reductions, no LDS partials)
constexpr int LR_NT = 576;                 // 72 columns x 8 lanes = 9 waves
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

__global__ __launch_bounds__(LR_NT) void k_lowrank(const double *__restrict__ lam, const int32_t *__restrict__ nuse,
                                                    const int32_t *__restrict__ status, const double *__restrict__ alphas,
                                                    int nalpha, int p, double *__restrict__ ufrag, double *__restrict__ wfrag,
                                                    int32_t *__restrict__ lrok) {
  extern __shared__ double A[];                 // [LR_P][LR_LDA] column-major: column j = the 208 coefficients of eigen index j
  __shared__ double cnorm[LR_P];                // squared norms of the remaining columns, rows >= current (-1: pivoted)
  __shared__ double cnorm2[LR_P];               // the same over the rows below the current one
  __shared__ int posof[LR_P];                   // step at which a column was pivoted, -1 if never
  __shared__ double tau_s[LR_K2];
  __shared__ double s_al[LR_NA], s_be[LR_NA], s_lam[LR_P];
  __shared__ int perm[LR_K2];                   // perm[s] = the column pivoted at step s
  __shared__ int freecol[LR_K2];                 // never-pivoted columns that hold Q_K afterwards
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  if (status[c] != 0) {
    if (tid == 0) lrok[c] = 0;
    return;
  }
  const double n = (double)nuse[c];
  // ---- B^T: A[j][i] = beta_i / (n beta_i lam_j + alpha_i)   (zero for the padding alpha / eigen indices).
  //      alpha, beta and lam are staged in LDS first: two dependent global loads per entry would dominate the kernel.
  for (int i = tid; i < LR_NA; i += LR_NT) {
    const double a = (i < nalpha) ? alphas[i] : 1.0;
    s_al[i] = a;
    s_be[i] = (i < nalpha) ? (1.0 - a) / (n - 1.0) : 0.0;
  }
  for (int j = tid; j < LR_P; j += LR_NT) s_lam[j] = (j < p) ? lam[(size_t)c * p + j] : 1.0;
  __syncthreads();
  for (int idx = tid; idx < LR_P * LR_NA; idx += LR_NT) {
    const int j = idx / LR_NA, i = idx - j * LR_NA;
    const double a = s_al[i], beta = s_be[i], lj = s_lam[j];
    const double v = lj * beta / ((n * beta) * lj + a);
    A[j * LR_LDA + i] = (i < nalpha && j < p) ? v : 0.0;
  }
  // (row-scaled: what is factored is B'_ji = lam_j B_ji, all entries in [0, 1/n); the sweep feeds it the whitened
  //  squares z_j / lam_j -- of order one whatever the spectrum -- so the factorisation error is relative to what each
  //  eigen-direction actually contributes to q, not to the largest entry of B, which is 1/(n lam_min).)
  bool lam_ok = true;
  {
    double lmin = 1.7976931348623157e308, lmax = 0.0;
    for (int j = 0; j < p; ++j) { lmin = fmin(lmin, s_lam[j]); lmax = fmax(lmax, s_lam[j]); }
    lam_ok = (lmin > 1e-10 * lmax) && (lmax <= 1.7976931348623157e308);   // positive definite, condition < 1e10
  }
  if (tid < LR_P) posof[tid] = -1;
  __syncthreads();
  const int col = tid >> 3, sub = tid & 7;   // update role: column `col` (0..71), rows sub, sub+8, ...
  {
    double s0 = 0.0, s1 = 0.0;
    for (int i = sub; i < LR_NA; i += LR_TPC) {
      const double v = A[col * LR_LDA + i];
      s0 = __builtin_fma(v, v, s0);
      if (i > 0) s1 = __builtin_fma(v, v, s1);
    }
    s0 = lr_sum8(s0);
    s1 = lr_sum8(s1);
    if (sub == 0) { cnorm[col] = s0; cnorm2[col] = s1; }
  }
  __syncthreads();
  // Two barriers per step.  No column is physically moved: a pivoted column keeps its place (posof[col] = its step),
  // every wave finds the pivot for itself (same data, same tie-break), every lane forms the reflector's scalars
  // from the carried norms -- cnorm = rows >= s, cnorm2 = rows > s, both summed afresh by the update of the
  // previous step, so nothing is obtained by subtraction.
  double r00 = 0.0;
  bool mine_done = false;                       // this group's column has been pivoted
  int kuse = 0;                                 // accepted rank: LR_K, LR_K2 or 0 (full-rank sweep)
  for (int s = 0; s < LR_K2; ++s) {
    if (s == LR_K) {   // rank 28 reached: is the trailing block already at the rounding floor?  (same answer in every thread)
      double bn = 0.0;
      for (int j = 0; j < LR_P; ++j) bn = fmax(bn, cnorm[j]);
      if (sqrt(bn) <= 1e-14 * r00 && r00 > 0.0) { kuse = LR_K; break; }
    }
    // pivot = the remaining column of largest norm (lowest index on ties; pivoted columns carry -1).  Every 8-lane
    // group scans all 72 candidates (9 per lane) and finishes with three DPP exchanges: no cross-wave traffic, no
    // ds_bpermute chain (six dependent LDS-crossbar round trips were the longest part of the step).
    double bv = cnorm[sub];
    int bi = sub;
#pragma unroll
    for (int k = 1; k < LR_P / LR_TPC; ++k) {
      const double v2 = cnorm[sub + LR_TPC * k];
      if (v2 > bv) { bv = v2; bi = sub + LR_TPC * k; }
    }
    lr_argmax_step<0xB1>(bv, bi);
    lr_argmax_step<0x4E>(bv, bi);
    lr_argmax_step<0x141>(bv, bi);
    const int pv = bi;
    const double *xs = A + pv * LR_LDA;
    const double alpha = xs[s], xn2 = cnorm2[pv];
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
    if (s == 0) r00 = fabs(betah);
    // v = x * scale (v_s = 1 implicit), in registers of every lane for its rows
    double vr[(LR_NA + LR_TPC - 1) / LR_TPC];
    // rows >= 32 are below every diagonal position (s < 28): only the first four row slots need the predicates
#pragma unroll
    for (int k = 0; k < LR_PK; ++k) {
      const int i = sub + LR_TPC * k;
      vr[k] = (i > s) ? xs[i] * scale : ((i == s) ? 1.0 : 0.0);
    }
#pragma unroll
    for (int k = LR_PK; k < LR_NA / LR_TPC; ++k) vr[k] = xs[sub + LR_TPC * k] * scale;
    __syncthreads();   // every lane has read column pv (and the norms) before they change
    if (col == pv) {   // the pivot's own group stores v, R_ss and the bookkeeping while the others update
#pragma unroll
      for (int k = 0; k < LR_NA / LR_TPC; ++k) {
        const int i = sub + LR_TPC * k;
        if (k >= LR_PK || i > s) A[pv * LR_LDA + i] = vr[k];
      }
      if (sub == 0) {
        A[pv * LR_LDA + s] = betah;
        tau_s[s] = tau;
        perm[s] = pv;
        posof[pv] = s;
        cnorm[pv] = -1.0;
      }
      mine_done = true;
    } else if (!mine_done) {
      // ---- apply H = I - tau v v^T; squared norms of what is left in rows > s and in rows > s+1
      double *ac = A + col * LR_LDA;
      double av[(LR_NA + LR_TPC - 1) / LR_TPC];
      double w = 0.0;
#pragma unroll
      for (int k = 0; k < LR_NA / LR_TPC; ++k) {
        const int i = sub + LR_TPC * k;
        av[k] = (k >= LR_PK || i >= s) ? ac[i] : 0.0;
        w = __builtin_fma(vr[k], av[k], w);
      }
      const double wt = tau * lr_sum8(w);
      double nn = 0.0, nn2 = 0.0;
#pragma unroll
      for (int k = 0; k < LR_PK; ++k) {
        const int i = sub + LR_TPC * k;
        if (i >= s) {
          const double nv = __builtin_fma(-wt, vr[k], av[k]);
          ac[i] = nv;
          if (i > s) nn = __builtin_fma(nv, nv, nn);
          if (i > s + 1) nn2 = __builtin_fma(nv, nv, nn2);
        }
      }
      double nb = 0.0;
#pragma unroll
      for (int k = LR_PK; k < LR_NA / LR_TPC; ++k) {
        const double nv = __builtin_fma(-wt, vr[k], av[k]);
        ac[sub + LR_TPC * k] = nv;
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
    if (sqrt(bn) <= 1e-14 * r00 && r00 > 0.0) kuse = LR_K2;
  }
  if (!lam_ok) kuse = 0;   // (near-)singular or indefinite correlation matrix: the full-rank kernel
  if (tid == 0) lrok[c] = (kuse == LR_K) ? 1 : ((kuse == LR_K2) ? 2 : 0);
  if (kuse == 0) return;
  // ---- U fragments: ufrag[(jg*NK + mg)*16 + 4q + n] = -U[4jg+q][4mg+n];  U[j][m] = R[m][column j]: rows m <= posof[j]
  //      of a pivoted column (below them sits its reflector), all K rows of the others
  double *uo = ufrag + (size_t)c * (SF_SW4_NJ * LR_NK2 * 16);
  for (int idx = tid; idx < LR_P * LR_K2; idx += LR_NT) {
    const int j = idx / LR_K2, m = idx - j * LR_K2;
    const int pj = posof[j];
    const double r = (m < kuse && (pj < 0 || m <= pj)) ? A[j * LR_LDA + m] : 0.0;
    const int jg = j >> 2, q = j & 3, mg = m >> 2, nn = m & 3;
    uo[(jg * LR_NK2 + mg) * 16 + 4 * q + nn] = -r;
  }
  __syncthreads();
  // ---- Q_K = H_0 ... H_{K-1} [I_K; 0], formed in K of the columns that were never pivoted (their R entries have
  //      been exported); reflector s sits in column perm[s].  16 lanes per column.
  if (tid == 0) {
    int m = 0;
    for (int j = 0; j < LR_P && m < kuse; ++j)
      if (posof[j] < 0) freecol[m++] = j;
  }
  __syncthreads();
  for (int idx = tid; idx < kuse * LR_NA; idx += LR_NT) {
    const int m = idx / LR_NA, i = idx - m * LR_NA;
    A[freecol[m] * LR_LDA + i] = (i == m) ? 1.0 : 0.0;
  }
  __syncthreads();
  const int qcol = tid >> 4, qsub = tid & 15;   // up to 36 columns x 16 lanes = 576 threads
  for (int s = kuse - 1; s >= 0; --s) {
    const double *v = A + perm[s] * LR_LDA;
    const double tau = tau_s[s];
    if (qcol < kuse) {
      double *qc = A + freecol[qcol] * LR_LDA;
      double vv[(LR_NA + 15) / 16], qv[(LR_NA + 15) / 16];
      double w = 0.0;
#pragma unroll
      for (int k = 0; k < (LR_NA + 15) / 16; ++k) {
        const int i = qsub + 16 * k;
        vv[k] = (i > s && i < LR_NA) ? v[i] : ((i == s) ? 1.0 : 0.0);
        qv[k] = (i >= s && i < LR_NA) ? qc[i] : 0.0;
        w = __builtin_fma(vv[k], qv[k], w);
      }
      const double wt = tau * lr_sum16(w);
#pragma unroll
      for (int k = 0; k < (LR_NA + 15) / 16; ++k) {
        const int i = qsub + 16 * k;
        if (i >= s && i < LR_NA) qc[i] = __builtin_fma(-wt, vv[k], qv[k]);
      }
    }
    // columns are independent: no barrier between reflectors
  }
  __syncthreads();
  // ---- W fragments: wfrag[(M*NK + mg)*64 + lane], lane = 16q + 4mm + n  ->  W[4mg+q][16M + 4mm + n] = Q[alpha][m]
  double *wo = wfrag + (size_t)c * (SF_SW4_NM * LR_NK2 * 64);
  for (int idx = tid; idx < SF_SW4_NM * LR_NK2 * 64; idx += LR_NT) {
    const int ln = idx & 63, blk = idx >> 6;
    const int M = blk / LR_NK2, mg = blk - M * LR_NK2;
    const int q = ln >> 4, a = 16 * M + (ln & 15);
    wo[idx] = (4 * mg + q < kuse) ? A[freecol[4 * mg + q] * LR_LDA + a] : 0.0;
  }
}

}  // namespace

size_t sf_lowrank_bytes(const SfGeom &g) {
  return sf_align((size_t)g.ncols * SF_SW4_NJ * LR_NK2 * 16 * sizeof(double)) +
         sf_align((size_t)g.ncols * SF_SW4_NM * LR_NK2 * 64 * sizeof(double)) + sf_align((size_t)g.ncols * sizeof(int32_t));
}

int sf_launch_lowrank(const double *lam, const int32_t *nuse, const int32_t *status, const double *alphas, const SfGeom &g,
                      double *ufrag, double *wfrag, int32_t *lrok, hipStream_t st) {
  const size_t lds = (size_t)LR_P * LR_LDA * sizeof(double);
  if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_lowrank), lds)) return rc;
  hipLaunchKernelGGL(k_lowrank, dim3(g.ncols), dim3(LR_NT), lds, st, lam, nuse, status, alphas, g.nalpha, g.p, ufrag, wfrag, lrok);
  SF_LAUNCH_CHECK("k_lowrank");
  return 0;
}

// test hook (tests/test_cmf_gpu.py): the factorisation of one launch, fragments as the sweep reads them
extern "C" int sf_debug_lowrank(const double *lam, const int32_t *nuse, const int32_t *status, const double *alphas,
                                int nalpha, int p, int ncols, double *ufrag, double *wfrag, int32_t *lrok, void *stream) {
  const SfGeom g = sf_geom(64, p, ncols, nalpha);
  if (g.s4 != SF_SW4_NJ || g.nu != SF_SW4_NM) { sf_set_error("sf_debug_lowrank: production window only"); return -2; }
  return sf_launch_lowrank(lam, nuse, status, alphas, g, ufrag, wfrag, lrok, (hipStream_t)stream);
}
