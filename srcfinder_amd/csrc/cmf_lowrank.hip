// Low-rank form of the LOO sweep's coefficient matrix (production window, used by k_sweep4r in cmf_loocv4.hip).
//
// GEMM2 of the sweep evaluates  q_k(alpha_i) = 1 - sum_j z_kj B_ji,   B_ji = beta_i / (n beta_i lam_j + alpha_i),
// for every row k: 72 x 208 coefficients.  B is a Cauchy matrix in disguise, B_ji = (1/(n-1)) / (x_j + g_i) with
// x_j = n lam_j/(n-1), g_i = alpha_i/(1-alpha_i), and its numerical rank is ~25: the singular values fall by a decade
// every 1.6 (sigma_26/sigma_0 < 1e-16 on flightline-like spectra).  So per column
//        B = U W + E,   U (72 x K), W (K x 208) with orthonormal rows,   K = 28,   |E| <= ~1e-16 |B|,
// and the sweep multiplies z by U (72 x 28) and then by W (28 x 208) instead of by B: 44 % fewer flops in the
// product that is 3/4 of the kernel, with q changing by one ulp (numpy model: max |dq| = 1.1e-16, identical NLL
// argmin on every test spectrum).
//
// The factorisation is a Householder QR with column pivoting of B^T (208 x 72), stopped after K steps:
//   B^T P = Q R   =>   B = (P R_K^T) (Q_K^T) + E,   |E| = the largest remaining column norm,
// backward stable whatever the conditioning (B is numerically singular by design -- no Gram matrix, no
// inverse).  One 256-thread workgroup per column, the matrix in LDS; ~30 us.  A column whose remaining norm
// after K steps is not below 3e-15 |R_00| (the rounding floor of the trailing block is ~7e-16 |R_00|) is flagged and swept by the full-rank kernel instead.
#include "cmf_common.h"

namespace {

constexpr int LR_P = 4 * SF_SW4_NJ;        // 72
constexpr int LR_NA = 16 * SF_SW4_NM;      // 208
constexpr int LR_K = SF_LR_K;              // 28
constexpr int LR_LDA = LR_NA + 1;          // column stride in LDS (odd: threads on different columns, same row)
constexpr int LR_TPC = 3;                  // threads per column in the update

__global__ __launch_bounds__(256) void k_lowrank(const double *__restrict__ lam, const int32_t *__restrict__ nuse,
                                                  const int32_t *__restrict__ status, const double *__restrict__ alphas,
                                                  int nalpha, int p, double *__restrict__ ufrag, double *__restrict__ wfrag,
                                                  int32_t *__restrict__ lrok) {
  extern __shared__ double A[];                 // [LR_P][LR_LDA] column-major: column j = the 208 coefficients of eigen index j
  __shared__ double part[LR_P][LR_TPC];         // partial dot products / partial squared norms
  __shared__ double qpart[LR_K * 9];
  __shared__ double wred[4];
  __shared__ double tau_s[LR_K];
  __shared__ double sc[4];                      // [0] tau, [1] 1/(alpha - beta_h)
  __shared__ int perm[LR_P];                    // perm[pos] = original column at position pos
  __shared__ int piv;
  const int c = blockIdx.x, tid = threadIdx.x;
  if (status[c] != 0) {
    if (tid == 0) lrok[c] = 0;
    return;
  }
  const double n = (double)nuse[c];
  // ---- B^T: A[j][i] = beta_i / (n beta_i lam_j + alpha_i)   (zero for the padding alpha / eigen indices)
  for (int idx = tid; idx < LR_P * LR_NA; idx += 256) {
    const int j = idx / LR_NA, i = idx - j * LR_NA;
    double v = 0.0;
    if (i < nalpha && j < p) {
      const double a = alphas[i];
      const double beta = (1.0 - a) / (n - 1.0);
      v = beta / ((n * beta) * lam[(size_t)c * p + j] + a);
    }
    A[j * LR_LDA + i] = v;
  }
  if (tid < LR_P) perm[tid] = tid;
  __syncthreads();
  const int col = tid / LR_TPC, sub = tid - col * LR_TPC;   // update role: column `col`, rows sub, sub+3, ...
  const bool worker = col < LR_P;
  // squared norms of all columns
  if (worker) {
    double s = 0.0;
    for (int i = sub; i < LR_NA; i += LR_TPC) { const double v = A[col * LR_LDA + i]; s = __builtin_fma(v, v, s); }
    part[col][sub] = s;
  }
  __syncthreads();
  double r00 = 0.0, resid = 0.0;
  for (int s = 0; s < LR_K; ++s) {
    // ---- pivot: the remaining column of largest norm
    if (tid == 0) {
      int best = s;
      double bn = -1.0;
      for (int j = s; j < LR_P; ++j) {
        const double v = (part[j][0] + part[j][1]) + part[j][2];
        if (v > bn) { bn = v; best = j; }
      }
      piv = best;
    }
    __syncthreads();
    const int pv = piv;
    if (pv != s) {
      for (int i = tid; i < LR_NA; i += 256) {
        const double t = A[s * LR_LDA + i];
        A[s * LR_LDA + i] = A[pv * LR_LDA + i];
        A[pv * LR_LDA + i] = t;
      }
      if (tid == 0) { const int t = perm[s]; perm[s] = perm[pv]; perm[pv] = t; }
    }
    __syncthreads();
    // ---- Householder reflector of column s below the diagonal (LAPACK dlarfg); the norm of the part below the
    //      diagonal is summed directly (the carried column norm minus alpha^2 would cancel)
    {
      double x2 = 0.0;
      if (tid < LR_NA && tid > s) { const double xv = A[s * LR_LDA + tid]; x2 = xv * xv; }
      for (int off = 32; off > 0; off >>= 1) x2 += __shfl_xor(x2, off, 64);
      if ((tid & 63) == 0) wred[tid >> 6] = x2;
    }
    __syncthreads();
    if (tid == 0) {
      const double alpha = A[s * LR_LDA + s];
      const double xn2 = (wred[0] + wred[1]) + (wred[2] + wred[3]);
      double tau = 0.0, scale = 0.0, betah = alpha;
      if (xn2 > 0.0) {
        betah = -copysign(sqrt(alpha * alpha + xn2), alpha);
        tau = (betah - alpha) / betah;
        scale = 1.0 / (alpha - betah);
      }
      sc[0] = tau;
      sc[1] = scale;
      tau_s[s] = tau;
      A[s * LR_LDA + s] = betah;   // R_ss
    }
    __syncthreads();
    const double tau = sc[0], scale = sc[1];
    for (int i = s + 1 + tid; i < LR_NA; i += 256) A[s * LR_LDA + i] *= scale;   // v (v_s = 1 implicit)
    __syncthreads();
    // ---- apply H = I - tau v v^T to the remaining columns; partial squared norms of what is left below row s
    const double *v = A + s * LR_LDA;
    double w = 0.0;
    if (worker && col > s) {
      const double *ac = A + col * LR_LDA;
      for (int i = s + sub; i < LR_NA; i += LR_TPC) w = __builtin_fma(i == s ? 1.0 : v[i], ac[i], w);
      part[col][sub] = w;
    }
    __syncthreads();
    double wt = 0.0;
    if (worker && col > s) wt = tau * ((part[col][0] + part[col][1]) + part[col][2]);
    __syncthreads();   // every partial has been read before part[] is reused for the norms
    if (worker && col > s) {
      double *ac = A + col * LR_LDA;
      double nn = 0.0;
      for (int i = s + sub; i < LR_NA; i += LR_TPC) {
        const double nv = __builtin_fma(-wt, i == s ? 1.0 : v[i], ac[i]);
        ac[i] = nv;
        if (i > s) nn = __builtin_fma(nv, nv, nn);
      }
      part[col][sub] = nn;
    } else if (worker) {
      part[col][sub] = 0.0;
    }
    __syncthreads();
    if (s == 0) r00 = fabs(A[0]);
  }
  {   // what is left after K steps
    double bn = 0.0;
    for (int j = LR_K; j < LR_P; ++j) bn = fmax(bn, (part[j][0] + part[j][1]) + part[j][2]);
    resid = sqrt(bn);
  }
  if (tid == 0) lrok[c] = (resid <= 3e-15 * r00 && r00 > 0.0) ? 1 : 0;
  // ---- U fragments: ufrag[(jg*NK + mg)*16 + 4q + n] = -U[4jg+q][4mg+n],  U[perm[pos]][m] = R[m][pos]
  double *uo = ufrag + (size_t)c * (SF_SW4_NJ * (LR_K / 4) * 16);
  for (int idx = tid; idx < LR_P * LR_K; idx += 256) {
    const int pos = idx / LR_K, m = idx - pos * LR_K;
    const double r = (m <= pos) ? A[pos * LR_LDA + m] : 0.0;
    const int j = perm[pos];
    const int jg = j >> 2, q = j & 3, mg = m >> 2, nn = m & 3;
    uo[(jg * (LR_K / 4) + mg) * 16 + 4 * q + nn] = -r;
  }
  __syncthreads();
  // ---- Q_K = H_0 ... H_{K-1} [I_K; 0], formed in the (now free) columns K .. 2K-1 of A
  double *Q = A + LR_K * LR_LDA;
  for (int idx = tid; idx < LR_K * LR_NA; idx += 256) {
    const int m = idx / LR_NA, i = idx - m * LR_NA;
    Q[m * LR_LDA + i] = (i == m) ? 1.0 : 0.0;
  }
  __syncthreads();
  const int qcol = tid / 9, qsub = tid - qcol * 9;   // 28 columns x 9 threads = 252
  for (int s = LR_K - 1; s >= 0; --s) {
    const double *v = A + s * LR_LDA;
    const double tau = tau_s[s];
    double w = 0.0;
    if (qcol < LR_K) {
      const double *qc = Q + qcol * LR_LDA;
      for (int i = s + qsub; i < LR_NA; i += 9) w = __builtin_fma(i == s ? 1.0 : v[i], qc[i], w);
    }
    double *pp = qpart;
    if (qcol < LR_K) pp[qcol * 9 + qsub] = w;
    __syncthreads();
    if (qcol < LR_K) {
      double t = 0.0;
      for (int e = 0; e < 9; ++e) t += pp[qcol * 9 + e];
      const double wt = tau * t;
      double *qc = Q + qcol * LR_LDA;
      for (int i = s + qsub; i < LR_NA; i += 9) qc[i] = __builtin_fma(-wt, i == s ? 1.0 : v[i], qc[i]);
    }
    __syncthreads();
  }
  // ---- W fragments: wfrag[(M*NK + mg)*64 + lane], lane = 16q + 4mm + n  ->  W[4mg+q][16M + 4mm + n] = Q[alpha][m]
  double *wo = wfrag + (size_t)c * (SF_SW4_NM * (LR_K / 4) * 64);
  for (int idx = tid; idx < SF_SW4_NM * (LR_K / 4) * 64; idx += 256) {
    const int ln = idx & 63, blk = idx >> 6;
    const int M = blk / (LR_K / 4), mg = blk - M * (LR_K / 4);
    const int q = ln >> 4, a = 16 * M + (ln & 15);
    wo[idx] = Q[(4 * mg + q) * LR_LDA + a];
  }
}

}  // namespace

size_t sf_lowrank_bytes(const SfGeom &g) {
  return sf_align((size_t)g.ncols * SF_SW4_NJ * (LR_K / 4) * 16 * sizeof(double)) +
         sf_align((size_t)g.ncols * SF_SW4_NM * (LR_K / 4) * 64 * sizeof(double)) + sf_align((size_t)g.ncols * sizeof(int32_t));
}

int sf_launch_lowrank(const double *lam, const int32_t *nuse, const int32_t *status, const double *alphas, const SfGeom &g,
                      double *ufrag, double *wfrag, int32_t *lrok, hipStream_t st) {
  const size_t lds = (size_t)LR_P * LR_LDA * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    SF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lowrank), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_lowrank, dim3(g.ncols), dim3(256), lds, st, lam, nuse, status, alphas, g.nalpha, g.p, ufrag, wfrag, lrok);
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
