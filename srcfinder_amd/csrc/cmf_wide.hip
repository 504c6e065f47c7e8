// Wide-window statistics path (active window of more than 96 bands, e.g. the reference's reflectance window
// 5..420, p = 416, cmf/robust_mf.py:186-187).  Same algorithm as the LDS-resident path, but the
// per-column matrices no longer fit LDS / registers, so the stages become batched float64 GEMMs over
// global-memory operands plus a global-memory Jacobi:
//   X~ = valid rows - mean                      k_center     (float64 copy, zero rows where masked)
//   S  = X~^T X~ / (n-1)                        k_dgemm<TA>  (robust_mf.py:52-70)
//   R  = D^-1 S D^-1 = V diag(lam) V^T          k_eigh_global (Cholesky + one-sided Jacobi, fallback with V)
//   Z  = (X~ W)^2,  W = D^-1 V                  k_dgemm<SQUARE>
//   r  = Z C,  C[j][i] = 1/(n beta_i lam_j + a_i)   k_dgemm
//   sum_k log q, sum_k r/q                      k_nllrows -> k_nll (shared with the fused path)
// Columns are processed in batches so the float64 scratch (X~, Z, r: 166 MB per column at 20000 x 416) stays a
// few GB.  Correctness-first: v_mfma_f64_16x16x4_f64 64x64 block tiles without software pipelining.
#include "cmf_common.h"

namespace {

constexpr int WD_BM = 64, WD_BN = 64, WD_BK = 16, WD_LD = 80;  // LDS row stride 16 (mod 32) doubles

template <typename XT>
__global__ void k_center(const XT *__restrict__ xt, const uint8_t *__restrict__ mask_t, const double *__restrict__ mu,
                         int L, int p, int PS, int c0, double *__restrict__ xc) {
  const int c = blockIdx.y;  // column within the batch
  const size_t tot = (size_t)L * p;
  const XT *xs = xt + (size_t)(c0 + c) * L * PS;
  const uint8_t *mp = mask_t + (size_t)(c0 + c) * L;
  const double *m = mu + (size_t)(c0 + c) * p;
  double *o = xc + (size_t)c * tot;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / p), b = (int)(i - (size_t)r * p);
    o[i] = mp[r] ? (double)xs[(size_t)r * PS + b] - m[b] : 0.0;
  }
}

// C[M x N] = op(A) B, float64, batched over blockIdx.z.  TA: A is stored [K x M] (row-major) instead of [M x K].
// SQUARE: C = (A B).^2.  Scalar predicated loads (any size); 64x64 block, 4 waves of 32x32, BK = 16.
template <bool TA, bool SQUARE>
__global__ __launch_bounds__(256) void k_dgemm(const double *__restrict__ A, int lda, size_t sA,
                                                const double *__restrict__ B, int ldb, size_t sB,
                                                double *__restrict__ Cm, int ldc, size_t sC, int M, int N, int K,
                                                double scale) {
  __shared__ double As[WD_BK * WD_LD];
  __shared__ double Bs[WD_BK * WD_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * WD_BM, n0 = blockIdx.y * WD_BN;
  A += (size_t)blockIdx.z * sA;
  B += (size_t)blockIdx.z * sB;
  Cm += (size_t)blockIdx.z * sC;
  d4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};

  for (int k0 = 0; k0 < K; k0 += WD_BK) {
    double ra[4], rb[4];
    if (TA) {  // A[k][m]: thread (k = tid/16, 4 consecutive m)
      const int k = tid >> 4, mq = (tid & 15) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = m0 + mq + j;
        ra[j] = (k0 + k < K && m < M) ? A[(size_t)(k0 + k) * lda + m] : 0.0;
      }
    } else {   // A[m][k]: thread (m = tid/4, 4 consecutive k)
      const int m = m0 + (tid >> 2), kq = (tid & 3) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) ra[j] = (m < M && k0 + kq + j < K) ? A[(size_t)m * lda + k0 + kq + j] : 0.0;
    }
    {
      const int k = tid >> 4, nq = (tid & 15) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + nq + j;
        rb[j] = (k0 + k < K && n < N) ? B[(size_t)(k0 + k) * ldb + n] : 0.0;
      }
    }
    __syncthreads();  // previous tile consumed
    if (TA) {
      const int k = tid >> 4, mq = (tid & 15) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) As[k * WD_LD + mq + j] = ra[j];
    } else {
      const int m = tid >> 2, kq = (tid & 3) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) As[(kq + j) * WD_LD + m] = ra[j];
    }
    {
      const int k = tid >> 4, nq = (tid & 15) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) Bs[k * WD_LD + nq + j] = rb[j];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < WD_BK / 4; ++kk) {
      double a[2], b[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = As[(4 * kk + g) * WD_LD + 32 * wm + 16 * t + li];
        b[t] = Bs[(4 * kk + g) * WD_LD + 32 * wn + 16 * t + li];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + 32 * wm + 16 * i + g + 4 * r, n = n0 + 32 * wn + 16 * j + li;
        if (m < M && n < N) {
          double v = acc[i][j][r] * scale;
          if (SQUARE) v = v * v;
          Cm[(size_t)m * ldc + n] = v;
        }
      }
}

// cov = XtX / (n - 1)  (in place, per column of the batch)
__global__ void k_scale_cov(double *__restrict__ cov, const int32_t *__restrict__ nuse, int p, int c0) {
  const int c = blockIdx.y;
  const double inv = 1.0 / ((double)nuse[c0 + c] - 1.0);
  double *o = cov + (size_t)(c0 + c) * p * p;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < p * p; i += gridDim.x * blockDim.x) o[i] *= inv;
}

// W[b][j] = V_j[b] / d_b  (row-major p x p, the B operand of Y = X~ W)
__global__ void k_wmat(const double *__restrict__ evec, const double *__restrict__ d, int p, int c0,
                       double *__restrict__ W) {
  const int c = blockIdx.y;
  const double *ev = evec + (size_t)(c0 + c) * p * p, *dd = d + (size_t)(c0 + c) * p;
  double *o = W + (size_t)c * p * p;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < p * p; i += gridDim.x * blockDim.x) {
    const int b = i / p, j = i - b * p;
    o[i] = ev[(size_t)j * p + b] / dd[b];
  }
}

// C[j][i] = 1 / (n beta_i lam_j + alpha_i)   (p x NA16, zero-padded alpha columns)
__global__ void k_cmat(const double *__restrict__ lam, const int32_t *__restrict__ nuse, const int32_t *__restrict__ status,
                       const double *__restrict__ alphas, int nalpha, int NA16, int p, int c0, double *__restrict__ Cm) {
  const int c = blockIdx.y;
  const double n = (double)nuse[c0 + c];
  const bool ok = status[c0 + c] == 0;
  const double *lc = lam + (size_t)(c0 + c) * p;
  double *o = Cm + (size_t)c * p * NA16;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < p * NA16; i += gridDim.x * blockDim.x) {
    const int j = i / NA16, a = i - j * NA16;
    double v = 0.0;
    if (ok && a < nalpha) {
      const double al = alphas[a];
      const double beta = (1.0 - al) / (n - 1.0);
      v = 1.0 / ((n * beta) * lc[j] + al);
    }
    o[i] = v;
  }
}

// per (column, row chunk): sum_k log q_ki and sum_k r_ki / q_ki for every alpha, q = 1 - beta r
// (robust_mf.py:115-117); same partial layout as the fused sweep so k_nll finishes both paths.
__global__ __launch_bounds__(256) void k_nllrows(const double *__restrict__ Rm, const int32_t *__restrict__ nuse,
                                                  const int32_t *__restrict__ status, const double *__restrict__ alphas,
                                                  int nalpha, int NA16, int L, int rows_per_wg, int c0, int nsplit,
                                                  double *__restrict__ part) {
  const int c = blockIdx.x, split = blockIdx.y, i = threadIdx.x;
  double *po = part + ((size_t)(c0 + c) * nsplit + split) * 2 * NA16;
  if (i >= NA16) return;
  if (status[c0 + c] != 0 || i >= nalpha) {
    po[i] = 0.0;
    po[NA16 + i] = 0.0;
    return;
  }
  const double n = (double)nuse[c0 + c];
  const double beta = (1.0 - alphas[i]) / (n - 1.0);
  const int rbeg = split * rows_per_wg, rend = min(L, rbeg + rows_per_wg);
  const double *rp = Rm + (size_t)c * L * NA16 + i;
  double P = 1.0, S = 0.0;
  int E = 0;
  bool neg = false;
  for (int k = rbeg; k < rend; ++k) {
    const double r = rp[(size_t)k * NA16];
    const double q = __builtin_fma(-beta, r, 1.0);
    neg = neg | (q < 0.0);
    S += r / q;
    const double pm = P * q;
    E += __builtin_amdgcn_frexp_exp(pm);
    P = __builtin_amdgcn_frexp_mant(pm);
  }
  po[i] = log(P) + (double)E * 0.6931471805599453094;
  po[NA16 + i] = neg ? __builtin_nan("") : S;
}

// ---- one-sided Jacobi with the matrix in global memory (one 1024-thread workgroup per column) -------------------
// Same method as k_eigh (cmf_eigh.hip): Cholesky R = L L^T, Jacobi on the columns of L, eigenvectors = normalised
// columns; if R is not positive definite, Jacobi on R with V accumulated alongside.  All waves of the workgroup
// run on one CU and share its L1, so plain loads/stores ordered by __syncthreads() (which drains vmcnt) are
// coherent; 16 lanes per pair keep a pair's two columns in registers between the dot product and the rotation.
__device__ __forceinline__ double vshfl_sum16(double v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}
__device__ __forceinline__ void rr_pair_w(int s, int k, int m, int &a, int &b) {
  int x = s + k;
  x = x >= m ? x - m : x;
  int y = s - k;
  y = y < 0 ? y + m : y;
  a = x;
  b = (k == 0) ? m : y;
}

constexpr int EG_RMAX = 32;  // rows per lane at 16 lanes per pair: p2 <= 512

__global__ __launch_bounds__(512) void k_eigh_global(const double *__restrict__ cov, const int32_t *__restrict__ nuse, int p,
                                                       int p2, int c0, double *__restrict__ d_out,
                                                       double *__restrict__ lam_out, double *__restrict__ evec_out,
                                                       int32_t *__restrict__ status, double *__restrict__ gscratch) {
  __shared__ double dv[512], nrm[512];
  __shared__ int flag[2];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int c = c0 + blockIdx.x;
  const double *S = cov + (size_t)c * p * p;
  double *G = gscratch + (size_t)blockIdx.x * 2 * p2 * p2;  // [p2][p2] column-major
  double *V = G + (size_t)p2 * p2;
  const int n = nuse[c];
  const int LD = p2;
  if (tid < 2) flag[tid] = 0;
  for (int i = tid; i < p2; i += nthr) dv[i] = (i < p) ? sqrt(S[(size_t)i * p + i]) : 0.0;
  __syncthreads();
  for (int i = tid; i < p; i += nthr) {
    const double v = dv[i];
    if (!(v > 0.0) || !(v <= 1.79769313486231570e+308)) atomicOr(&flag[0], 1);
  }
  __syncthreads();
  int st = 0;
  if (n <= 0) st = 1;
  else if (n < 2 || flag[0]) st = 2;
  if (tid == 0) status[c] = st;
  for (int i = tid; i < p; i += nthr) d_out[(size_t)c * p + i] = dv[i];
  if (st != 0) {
    for (int i = tid; i < p; i += nthr) lam_out[(size_t)c * p + i] = 0.0;
    for (int i = tid; i < p * p; i += nthr) evec_out[(size_t)c * p * p + i] = ((i / p) == (i % p)) ? 1.0 : 0.0;
    return;
  }
  auto load_R = [&]() {
    for (int i = tid; i < p2 * p2; i += nthr) {
      const int col = i / p2, row = i - col * p2;
      double r = 0.0;
      if (col < p && row < p) r = S[(size_t)row * p + col] / (dv[row] * dv[col]);
      G[col * LD + row] = r;
    }
  };
  load_R();
  __syncthreads();
  bool chol_ok = true;
  for (int kk = 0; kk < p; ++kk) {
    const double dk = G[kk * LD + kk];
    if (!(dk > 0.0) || !(dk <= 1.79769313486231570e+308)) { chol_ok = false; break; }
    const double rk = 1.0 / sqrt(dk);
    __syncthreads();
    for (int i = kk + tid; i < p; i += nthr) G[kk * LD + i] = (i == kk) ? dk * rk : G[kk * LD + i] * rk;
    __syncthreads();
    const int rem = p - kk - 1;
    for (int e = tid; e < rem * rem; e += nthr) {
      const int jj = e / rem, ii = e - jj * rem;
      if (ii >= jj) {
        const int j = kk + 1 + jj, i = kk + 1 + ii;
        G[j * LD + i] = G[j * LD + i] - G[kk * LD + i] * G[kk * LD + j];
      }
    }
    __syncthreads();
  }
  if (chol_ok) {
    for (int i = tid; i < p2 * p2; i += nthr) {
      const int col = i / p2, row = i - col * p2;
      if (row < col || col >= p || row >= p) G[col * LD + row] = 0.0;
    }
  } else {
    __syncthreads();
    load_R();
    for (int i = tid; i < p2 * p2; i += nthr) V[i] = ((i / p2) == (i % p2)) ? 1.0 : 0.0;
  }
  __syncthreads();

  const int npairs = p2 >> 1, m = p2 - 1;
  const int kloc = tid >> 4, sub = tid & 15;
  const int ppp = nthr >> 4;  // pairs per pass
  const int nr = (p2 - sub + 15) >> 4;
  const double tol = (double)p2 * 2.220446049250313e-16, tol2 = tol * tol;
  for (int sweep = 0; sweep < 40; ++sweep) {
    bool rotated = false;
    for (int j = tid; j < p2; j += nthr) {
      double sacc = 0;
      for (int r = 0; r < p2; ++r) { const double x = G[j * LD + r]; sacc += x * x; }
      nrm[j] = sacc;
    }
    __syncthreads();
    for (int s = 0; s < m; ++s) {
      for (int k = kloc; k < npairs; k += ppp) {  // disjoint pairs: no ordering needed between passes
        int a, b;
        rr_pair_w(s, k, m, a, b);
        double *ga = G + (size_t)a * LD + sub, *gb = G + (size_t)b * LD + sub;
        const double aa = nrm[a], bb = nrm[b];
        double xa[EG_RMAX], xb[EG_RMAX];
#pragma unroll
        for (int i = 0; i < EG_RMAX; ++i) {
          const int ii = min(i, nr - 1);
          const double u = ga[16 * ii], v = gb[16 * ii];
          xa[i] = i < nr ? u : 0.0;
          xb[i] = i < nr ? v : 0.0;
        }
        double ab = 0;
#pragma unroll
        for (int i = 0; i < EG_RMAX; ++i) ab = __builtin_fma(xa[i], xb[i], ab);
        ab = vshfl_sum16(ab);
        const double ab2 = aa * bb;
        if (ab2 > 0.0 && ab * ab > tol2 * ab2) {
          rotated = true;
          const double tau = bb - aa, gam = 2.0 * ab;
          const double rinv = 1.0 / sqrt(tau * tau + gam * gam);
          const double c2 = fabs(tau) * rinv;
          const double h = 0.5 + 0.5 * c2;
          const double cs = sqrt(h);
          double sn = fabs(gam) * rinv * 0.5 / cs;
          sn = ((tau < 0.0) != (gam < 0.0)) ? -sn : sn;
#pragma unroll
          for (int i = 0; i < EG_RMAX; ++i) {
            if (i < nr) {
              ga[16 * i] = cs * xa[i] - sn * xb[i];
              gb[16 * i] = sn * xa[i] + cs * xb[i];
            }
          }
          if (!chol_ok) {
            double *va = V + (size_t)a * LD + sub, *vb = V + (size_t)b * LD + sub;
            for (int i = 0; i < nr; ++i) {
              const double vx = va[16 * i], vy = vb[16 * i];
              va[16 * i] = cs * vx - sn * vy;
              vb[16 * i] = sn * vx + cs * vy;
            }
          }
          if (sub == 0) {
            const double cc = cs * cs, ss = sn * sn, x2 = 2.0 * cs * sn * ab;
            nrm[a] = cc * aa - x2 + ss * bb;
            nrm[b] = ss * aa + x2 + cc * bb;
          }
        }
      }
      __syncthreads();
    }
    if (rotated) flag[1] = 1;
    __syncthreads();
    const int any = flag[1];
    __syncthreads();
    if (tid == 0) flag[1] = 0;
    if (!any) break;
  }
  __syncthreads();
  for (int j = tid; j < p; j += nthr) {
    double sacc = 0;
    for (int r = 0; r < p2; ++r) { const double x = G[j * LD + r]; sacc += x * x; }
    nrm[j] = sacc;
    lam_out[(size_t)c * p + j] = chol_ok ? sacc : sqrt(sacc);
  }
  __syncthreads();
  for (int i = tid; i < p * p; i += nthr) {
    const int j = i / p, b = i - j * p;
    double v;
    if (chol_ok) {
      const double s2 = nrm[j];
      v = s2 > 0.0 ? G[j * LD + b] / sqrt(s2) : ((j == b) ? 1.0 : 0.0);
    } else {
      v = V[j * LD + b];
    }
    evec_out[(size_t)c * p * p + i] = v;
  }
}

}  // namespace

// scratch per column of a batch: X~ + Z (L x p each), r (L x NA16), W (p x p), C (p x NA16), G|V (2 p2^2)
static size_t wide_col_bytes(const SfGeom &g) {
  const size_t L = g.lines, p = g.p, na = (size_t)g.nu * 16, p2 = g.p + (g.p & 1);
  return sf_align((2 * L * p + L * na + p * p + p * na + 2 * p2 * p2) * sizeof(double));
}
int sf_wide_batch(const SfGeom &g) {
  const size_t per = wide_col_bytes(g);
  size_t b = ((size_t)8 << 30) / per;
  if (b < 1) b = 1;
  if (b > (size_t)g.ncols) b = g.ncols;
  return (int)b;
}
static int wide_nll_splits(const SfGeom &g) { return sf_cdiv(g.lines, 512) > 64 ? 64 : sf_cdiv(g.lines, 512); }
size_t sf_wide_scratch_bytes(const SfGeom &g) {
  return (size_t)sf_wide_batch(g) * wide_col_bytes(g) +
         sf_align((size_t)g.ncols * wide_nll_splits(g) * 2 * g.nu * 16 * sizeof(double));
}

// stages 3-5 (covariance, eigendecomposition, LOO sweep + argmin) for windows too wide for the fused kernels
int sf_launch_wide_stats(const void *xt, int xt_f64, const uint8_t *mask_t, const int32_t *nuse, const int32_t *nloo,
                         const double *mu, const double *alphas, const SfGeom &g, double *cov, double *d, double *lam,
                         double *evec, int32_t *status, double *nll, int32_t *alphaidx, void *scratch, hipStream_t st);

int sf_launch_nll_finish(const double *part, int nsplit, const int32_t *nuse, const double *d, const double *lam,
                         const int32_t *status, const double *alphas, const SfGeom &g, double *nll, int32_t *alphaidx,
                         hipStream_t st);

int sf_launch_wide_stats(const void *xt, int xt_f64, const uint8_t *mask_t, const int32_t *nuse, const int32_t *nloo,
                         const double *mu, const double *alphas, const SfGeom &g, double *cov, double *d, double *lam,
                         double *evec, int32_t *status, double *nll, int32_t *alphaidx, void *scratch, hipStream_t st) {
  if (g.p > 512 || g.nalpha > SF_NALPHA_MAX) {
    sf_set_error("wide statistics path supports up to 512 bands and %d alphas", SF_NALPHA_MAX);
    return -2;
  }
  if (!nloo) nloo = nuse;
  const int L = g.lines, p = g.p, NA16 = g.nu * 16, p2 = g.p + (g.p & 1);
  const int bc = sf_wide_batch(g);
  const int nsplit = wide_nll_splits(g);
  const int rows = sf_cdiv(L, nsplit);
  const size_t per = wide_col_bytes(g);
  char *base = reinterpret_cast<char *>(scratch);
  double *part = reinterpret_cast<double *>(base + (size_t)bc * per);
  // batch-strided views (column c of the batch at base + c*per would break the GEMM batch stride, so each
  // array is laid out contiguously over the batch instead)
  double *xc = reinterpret_cast<double *>(base);
  double *z = xc + (size_t)bc * L * p;
  double *rm = z + (size_t)bc * L * p;
  double *W = rm + (size_t)bc * L * NA16;
  double *Cm = W + (size_t)bc * p * p;
  double *gv = Cm + (size_t)bc * p * NA16;
  for (int c0 = 0; c0 < g.ncols; c0 += bc) {
    const int nb = (g.ncols - c0 < bc) ? g.ncols - c0 : bc;
    // nuse: the rows the covariance is made of (ddof 1); nloo: the n of beta and of 1/(2n) (robust_mf.py:109, :116) --
    // the same number in the column loop, separate in the function-level looshrinkage(I_zm, alphas, nll, n)
    if (xt_f64)
      hipLaunchKernelGGL(k_center<double>, dim3(256, nb), dim3(256), 0, st, reinterpret_cast<const double *>(xt), mask_t, mu,
                         L, p, g.ps, c0, xc);
    else
      hipLaunchKernelGGL(k_center<float>, dim3(256, nb), dim3(256), 0, st, reinterpret_cast<const float *>(xt), mask_t, mu,
                         L, p, g.ps, c0, xc);
    SF_LAUNCH_CHECK("k_center");
    // S = X~^T X~  (A = X~ stored [K = L][M = p] -> TA)
    hipLaunchKernelGGL((k_dgemm<true, false>), dim3(sf_cdiv(p, WD_BM), sf_cdiv(p, WD_BN), nb), dim3(256), 0, st, xc, p,
                       (size_t)L * p, xc, p, (size_t)L * p, cov + (size_t)c0 * p * p, p, (size_t)p * p, p, p, L, 1.0);
    SF_LAUNCH_CHECK("k_dgemm(syrk)");
    hipLaunchKernelGGL(k_scale_cov, dim3(64, nb), dim3(256), 0, st, cov, nuse, p, c0);
    SF_LAUNCH_CHECK("k_scale_cov");
    hipLaunchKernelGGL(k_eigh_global, dim3(nb), dim3(512), 0, st, cov, nuse, p, p2, c0, d, lam, evec, status, gv);
    SF_LAUNCH_CHECK("k_eigh_global");
    hipLaunchKernelGGL(k_wmat, dim3(64, nb), dim3(256), 0, st, evec, d, p, c0, W);
    SF_LAUNCH_CHECK("k_wmat");
    hipLaunchKernelGGL(k_cmat, dim3(64, nb), dim3(256), 0, st, lam, nloo, status, alphas, g.nalpha, NA16, p, c0, Cm);
    SF_LAUNCH_CHECK("k_cmat");
    hipLaunchKernelGGL((k_dgemm<false, true>), dim3(sf_cdiv(L, WD_BM), sf_cdiv(p, WD_BN), nb), dim3(256), 0, st, xc, p,
                       (size_t)L * p, W, p, (size_t)p * p, z, p, (size_t)L * p, L, p, p, 1.0);
    SF_LAUNCH_CHECK("k_dgemm(Y^2)");
    hipLaunchKernelGGL((k_dgemm<false, false>), dim3(sf_cdiv(L, WD_BM), sf_cdiv(NA16, WD_BN), nb), dim3(256), 0, st, z, p,
                       (size_t)L * p, Cm, NA16, (size_t)p * NA16, rm, NA16, (size_t)L * NA16, L, NA16, p, 1.0);
    SF_LAUNCH_CHECK("k_dgemm(r)");
    hipLaunchKernelGGL(k_nllrows, dim3(nb, nsplit), dim3(256), 0, st, rm, nloo, status, alphas, g.nalpha, NA16, L, rows,
                       c0, nsplit, part);
    SF_LAUNCH_CHECK("k_nllrows");
  }
  return sf_launch_nll_finish(part, nsplit, nloo, d, lam, status, alphas, g, nll, alphaidx, st);
}
