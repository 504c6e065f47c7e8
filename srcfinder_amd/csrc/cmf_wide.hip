// Wide-window statistics path (active window of more than 96 bands, e.g. the reference's reflectance window
// 5..420, p = 416, cmf/robust_mf.py:186-187).  Same algorithm as the LDS-resident path, but the
// per-column matrices no longer fit LDS / registers:
//   S = X~^T X~ / (n-1)                  k_wsyrk   (cmf_wgemm.hip: fused centring / promotion, 4x4x4 fp64 MFMA; robust_mf.py:52-70)
//   R = D^-1 S D^-1 = V diag(lam) V^T    this file + cmf_wtri.hip: blocked Cholesky, tridiagonal preconditioner, blocked one-sided
//                                        Jacobi on the factor in global memory (k_blockjac / k_blockjac_q), k_eigh_global as the
//                                        single-workgroup fallback
//   NLL(alpha)                           k_wsweep8 (cmf_wgemm.hip) -> k_nll, exact determinants where the total leaves the range (linalg.hip)
// k_dgemm (below) is the batched float64 GEMM of the preconditioner.  Round 3's route (float64 copies of X~, Z and r in
// batches of 36 columns: k_center, three k_dgemm, k_nllrows) was removed in round 5 together with its ~3 GB larger scratch.
#include "cmf_common.h"

namespace {

constexpr int WD_BM = 64, WD_BN = 64, WD_BK = 16, WD_LD = 80;  // LDS row stride 16 (mod 32) doubles

// C[M x N] = op(A) B, float64, batched over blockIdx.z.  TA: A is stored [K x M] (row-major) instead of [M x K].
// SQUARE: C = (A B).^2.  Scalar predicated loads (any size); 64x64 block, 4 waves of 32x32, BK = 16.
// SYM (the covariance, C = A^T A with B = A, M = N): only the tiles on and below the diagonal are computed, each
// written to both triangles.
// TB: B is stored [N x K] (row-major) instead of [K x N].  skip1 / skip2: per-matrix flags, a non-zero one leaves the matrix alone.
template <bool TA, bool SQUARE, bool SYM = false, bool TB = false>
__global__ __launch_bounds__(256) void k_dgemm(const double *__restrict__ A, int lda, size_t sA,
                                                const double *__restrict__ B, int ldb, size_t sB,
                                                double *__restrict__ Cm, int ldc, size_t sC, int M, int N, int K,
                                                double scale, const int32_t *__restrict__ skip1 = nullptr,
                                                const int32_t *__restrict__ skip2 = nullptr, int tri = 0) {
  if ((skip1 && skip1[blockIdx.z] != 0) || (skip2 && skip2[blockIdx.z] != 0)) return;
  __shared__ double As[WD_BK * WD_LD];
  __shared__ double Bs[WD_BK * WD_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * WD_BM, n0 = blockIdx.y * WD_BN;
  if (SYM && n0 > m0) return;
  A += (size_t)blockIdx.z * sA;
  B += (size_t)blockIdx.z * sB;
  Cm += (size_t)blockIdx.z * sC;
  d4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};

  // tri: the B operand is triangular -- 1: B'[k][n] = 0 for k > n (the K loop ends with the tile's last column), 2: B'[k][n] = 0 for
  // k < n (it starts at the tile's first column): the products with the Cholesky factor skip their zero half
  const int kbeg = (tri == 2) ? (blockIdx.y * WD_BN) / WD_BK * WD_BK : 0;
  const int kend = (tri == 1) ? min(K, (int)(blockIdx.y * WD_BN + WD_BN)) : K;
  for (int k0 = kbeg; k0 < kend; k0 += WD_BK) {
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
    if (TB) {  // B[n][k]: thread (n = tid/4, 4 consecutive k)
      const int n = n0 + (tid >> 2), kq = (tid & 3) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) rb[j] = (n < N && k0 + kq + j < K) ? B[(size_t)n * ldb + k0 + kq + j] : 0.0;
    } else {
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
    if (TB) {
      const int n = tid >> 2, kq = (tid & 3) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) Bs[(kq + j) * WD_LD + n] = rb[j];
    } else {
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
          if (SYM && n0 < m0) Cm[(size_t)n * ldc + m] = v;
        }
      }
}


// ---- one-sided Jacobi with the matrix in global memory (one 1024-thread workgroup per column) -------------------
// Same method as k_eigh (cmf_eigh.hip): Cholesky R = L L^T, Jacobi on the columns of L, eigenvectors = normalised
// columns; if R is not positive definite, Jacobi on R with V accumulated alongside.  All waves of the workgroup
// run on one CU and share its L1, so plain loads/stores ordered by __syncthreads() (which drains vmcnt) are
// coherent; 16 lanes per pair keep a pair's two columns in registers between the dot product and the rotation.
// 16-lane butterfly sum with DPP lane swaps (no LDS crossbar round trips, cf. cmf_eigh.hip)
template <int CTRL>
__device__ __forceinline__ double wd_dpp_swap(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double dpp_sum16(double v) {
  v += wd_dpp_swap<0xB1>(v);   // quad_perm [1,0,3,2]
  v += wd_dpp_swap<0x4E>(v);   // quad_perm [2,3,0,1]
  v += wd_dpp_swap<0x141>(v);  // row_half_mirror
  v += wd_dpp_swap<0x140>(v);  // row_mirror
  return v;
}
__device__ __forceinline__ double wd_rsqrt(double x) {   // hardware estimate + two Newton steps
  double y = __builtin_amdgcn_rsq(x);
  y = y * __builtin_fma(-0.5 * x * y, y, 1.5);
  y = y * __builtin_fma(-0.5 * x * y, y, 1.5);
  return y;
}
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
                                                       int32_t *__restrict__ status, double *__restrict__ gscratch,
                                                       int mode, int32_t *__restrict__ cflag, int unit) {
  // unit: `cov` holds the batch's already whitened matrices (full shrinkage target, k_wg_* below): matrix blockIdx.x,
  // no diagonal scaling, d is not written.
  // mode 0: the whole decomposition here.  mode 1: status, d and the Cholesky factor only (G = L left in gscratch for
  // the blocked Jacobi below; cflag = 0 ok, 1 not positive definite -> mode 2, 2 nothing to do).  mode 2: the whole
  // decomposition, only for the matrices mode 1 flagged 1 (Jacobi on R with V accumulated alongside).
  // mode 3: like mode 1 without the factorisation -- R is left in gscratch for the blocked Cholesky (k_chol_*).
  __shared__ double dv[512], nrm[512];
  __shared__ int flag[2];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int c = c0 + blockIdx.x;
  const double *S = cov + (size_t)(unit ? (int)blockIdx.x : c) * p * p;
  double *G = gscratch + (size_t)blockIdx.x * 2 * p2 * p2;  // [p2][p2] column-major
  double *V = G + (size_t)p2 * p2;
  const int n = nuse[c];
  const int LD = p2;
  if (mode == 2 && cflag[blockIdx.x] != 1) return;
  if (tid < 2) flag[tid] = 0;
  for (int i = tid; i < p2; i += nthr) dv[i] = (i < p) ? (unit ? 1.0 : sqrt(S[(size_t)i * p + i])) : 0.0;
  __syncthreads();
  for (int i = tid; i < p; i += nthr) {
    const double v = dv[i];
    if (!(v > 0.0) || !(v <= 1.79769313486231570e+308)) atomicOr(&flag[0], 1);
  }
  __syncthreads();
  int st = 0;
  if (n <= 0) st = 1;
  else if (n == 1) st = 3;   // one valid row: numpy.cov (ddof 1) is NaN, every NLL is NaN, argmin = 0, C and the score are NaN
  else if (flag[0]) st = 2;
  if (tid == 0) status[c] = st;
  if (!unit) for (int i = tid; i < p; i += nthr) d_out[(size_t)c * p + i] = dv[i];
  if (st != 0) {
    for (int i = tid; i < p; i += nthr) lam_out[(size_t)c * p + i] = 0.0;
    for (int i = tid; i < p * p; i += nthr) evec_out[(size_t)c * p * p + i] = ((i / p) == (i % p)) ? 1.0 : 0.0;
    if ((mode == 1 || mode == 3) && tid == 0) cflag[blockIdx.x] = 2;
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
  if (mode == 3) {
    if (tid == 0) cflag[blockIdx.x] = 0;
    return;
  }
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
  if (mode == 1) {
    if (tid == 0) cflag[blockIdx.x] = chol_ok ? 0 : 1;
    return;
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


// ---- blocked right-looking Cholesky of the batch (R = L L^T in place, lower triangle of the column-major buffer) ----
// k_eigh_global factors a 425 x 425 matrix with ONE workgroup and three barriers per column (13.7 ms per batch).  Here:
// per 16-column panel, k_chol_panel (one workgroup per matrix, the panel in LDS) factors the diagonal block and
// solves the rows below it, then k_chol_trail applies the rank-16 update to the trailing lower triangle with one
// workgroup per 64 x 64 tile and matrix.  A non-positive pivot flags the matrix (cflag = 1 -> k_eigh_global, mode 2).
constexpr int CH_B = 16;
// (gbn, gstride: matrix mtx of the launch is matrix mtx % gbn of column group mtx / gbn, whose buffers sit gstride bytes behind the
//  previous group's -- the ~850 launches of a flightline's four groups become ~210 over all its matrices)
__device__ __forceinline__ char *ch_goff(void *base, int grp, size_t gstride) { return reinterpret_cast<char *>(base) + (size_t)grp * gstride; }
__global__ __launch_bounds__(256) void k_chol_panel(double *__restrict__ gscratch, int p, int p2, int kb,
                                                     int32_t *__restrict__ cflag, int gbn, size_t gstride) {
  extern __shared__ __attribute__((aligned(16))) double pan[];   // [CH_B][H]: column j of the panel, rows k0 .. p-1
  __shared__ int bad;
  const int grp = blockIdx.x / gbn, mtx = blockIdx.x - grp * gbn, tid = threadIdx.x;
  gscratch = reinterpret_cast<double *>(ch_goff(gscratch, grp, gstride));
  cflag = reinterpret_cast<int32_t *>(ch_goff(cflag, grp, gstride));
  if (cflag[mtx] != 0) return;
  double *G = gscratch + (size_t)mtx * 2 * p2 * p2;
  const int k0 = kb * CH_B, nc = min(CH_B, p - k0), H = p - k0;
  if (tid == 0) bad = 0;
  for (int i = tid; i < nc * H; i += 256) {
    const int j = i / H, r = i - j * H;
    pan[j * H + r] = G[(size_t)(k0 + j) * p2 + k0 + r];
  }
  __syncthreads();
  for (int j = 0; j < nc; ++j) {
    const double dk = pan[j * H + j];
    if (!(dk > 0.0) || !(dk <= 1.79769313486231570e+308)) { if (tid == 0) bad = 1; break; }   // uniform
    const double rk = 1.0 / sqrt(dk);
    __syncthreads();
    for (int r = j + tid; r < H; r += 256) pan[j * H + r] = (r == j) ? dk * rk : pan[j * H + r] * rk;
    __syncthreads();
    const int rem = nc - j - 1;                      // columns j+1 .. nc-1 of the panel, rows >= their own index
    for (int e = tid; e < rem * (H - j - 1); e += 256) {
      const int jj = e / (H - j - 1), rr = e - jj * (H - j - 1);
      const int c2 = j + 1 + jj, r = j + 1 + rr;
      if (r >= c2) pan[c2 * H + r] = __builtin_fma(-pan[j * H + r], pan[j * H + c2], pan[c2 * H + r]);
    }
    __syncthreads();
  }
  __syncthreads();
  if (bad) {
    if (tid == 0) cflag[mtx] = 1;
    return;
  }
  for (int i = tid; i < nc * H; i += 256) {
    const int j = i / H, r = i - j * H;
    if (r >= j) G[(size_t)(k0 + j) * p2 + k0 + r] = pan[j * H + r];
  }
}

// A[i][j] -= sum_k L[i][k] L[j][k] for i >= j in the trailing block (rows / columns >= k1 = (kb + 1) * 16)
__global__ __launch_bounds__(256) void k_chol_trail(double *__restrict__ gscratch, int p, int p2, int kb,
                                                     const int32_t *__restrict__ cflag, int gbn, size_t gstride) {
  __shared__ double Li[64][CH_B + 1], Lj[64][CH_B + 1];
  const int grp = blockIdx.z / gbn, mtx = blockIdx.z - grp * gbn, tid = threadIdx.x;
  gscratch = reinterpret_cast<double *>(ch_goff(gscratch, grp, gstride));
  cflag = reinterpret_cast<const int32_t *>(ch_goff(const_cast<int32_t *>(cflag), grp, gstride));
  if (cflag[mtx] != 0) return;
  const int k0 = kb * CH_B, k1 = k0 + CH_B;
  const int ti = blockIdx.y, tj = blockIdx.x;
  if (tj > ti) return;                               // lower triangle of tiles
  const int i0 = k1 + ti * 64, j0 = k1 + tj * 64;
  if (i0 >= p || j0 >= p) return;
  double *G = gscratch + (size_t)mtx * 2 * p2 * p2;
  for (int e = tid; e < 64 * CH_B; e += 256) {
    const int k = e / 64, r = e - k * 64;
    Li[r][k] = (i0 + r < p) ? G[(size_t)(k0 + k) * p2 + i0 + r] : 0.0;
    Lj[r][k] = (j0 + r < p) ? G[(size_t)(k0 + k) * p2 + j0 + r] : 0.0;
  }
  __syncthreads();
  // thread (ri, cj): rows ri, ri+16, ri+32, ri+48 of columns cj, cj+16, ... (consecutive lanes: consecutive rows)
  const int ri = tid & 15, cj = tid >> 4;
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) {
    const int c = cj + 16 * cc, col = j0 + c;
    if (col >= p) continue;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = ri + 16 * rr, row = i0 + r;
      if (row >= p || row < col) continue;
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < CH_B; ++k) acc = __builtin_fma(Li[r][k], Lj[c][k], acc);
      G[(size_t)col * p2 + row] -= acc;
    }
  }
}

// after the factorisation: zero the strict upper triangle and the padding (G = L)
__global__ void k_chol_clean(double *__restrict__ gscratch, int p, int p2, const int32_t *__restrict__ cflag, int gbn,
                             size_t gstride) {
  const int grp = blockIdx.y / gbn, mtx = blockIdx.y - grp * gbn;
  gscratch = reinterpret_cast<double *>(ch_goff(gscratch, grp, gstride));
  cflag = reinterpret_cast<const int32_t *>(ch_goff(const_cast<int32_t *>(cflag), grp, gstride));
  if (cflag[mtx] != 0) return;
  double *G = gscratch + (size_t)mtx * 2 * p2 * p2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < p2 * p2; i += gridDim.x * blockDim.x) {
    const int col = i / p2, row = i - col * p2;
    if (row < col || col >= p || row >= p) G[i] = 0.0;
  }
}

// ---- blocked one-sided Jacobi for the wide windows -----------------------------------------------------------------
// k_eigh_global moves the whole p2 x p2 factor through ONE CU's memory pipe every step (p2 - 1 steps a sweep, 2.9 MB
// each at p = 425): ~30 GB/s per workgroup, 3 s for a 598-column flightline.  Here the factor is cut into blocks of 16
// columns; a workgroup takes a PAIR of blocks into LDS (2 x 16 x LDr doubles, 111 KB at p = 425) and rotates the 256
// cross pairs (a_i, b_j) in 16 conflict-free steps -- group i of 16 lanes keeps column a_i in registers for the whole
// visit, the b columns travel through LDS -- then writes both blocks back.  The block pairs of a step are disjoint
// (circle method over the blocks), so one launch per step does all of them for all matrices: a sweep is nblk - 1
// (or nblk) launches instead of p2 - 1 passes over the matrix, and the pairs INSIDE a block are swept in the first
// launch of the sweep, where every block appears exactly once.  Every pair is met once per sweep: a cyclic Jacobi in
// a blocked order.  Same rotation formula, threshold and carried norms as k_eigh_global.
constexpr int BJ_B = 16;        // columns per block
constexpr double BJ_TINY2 = 1e-18;   // (1e-9)^2: a rotation below this ratio leaves only second-order (1e-18) residue
constexpr int BJ_NT = 256;      // 16 groups of 16 lanes

__device__ __forceinline__ bool bj_rotation(double aa, double bb, double ab, double tol2, double &cs, double &sn) {
  const double ab2 = aa * bb;
  if (!(ab2 > 0.0 && ab * ab > tol2 * ab2)) return false;
  const double tau = bb - aa, gam = 2.0 * ab;
  const double rinv = wd_rsqrt(__builtin_fma(tau, tau, gam * gam));
  const double c2 = fabs(tau) * rinv;              // |cos 2 theta|
  const double h = __builtin_fma(0.5, c2, 0.5);    // cos^2 theta in [0.5, 1]
  const double rh = wd_rsqrt(h);
  cs = h * rh;
  sn = fabs(gam) * rinv * 0.5 * rh;
  sn = ((tau < 0.0) != (gam < 0.0)) ? -sn : sn;
  return true;
}

// RM: rows per lane the unrolled loops cover (16 RM >= p2): the loops of the first version always ran EG_RMAX = 32 iterations,
// five of them on zeros at p = 425 (0.672 -> 0.659 s per full-band flightline, same bits).  The kernels are bound by the latency of a
// step's dependent chain, not by its instruction count: EIGHT lanes per pair (two waves per workgroup, half the replicated
// rotation-parameter work) take 0.89 s, thirty-two (eight waves, half the rows per lane, one LDS-crossbar exchange in the dot
// product) 0.735 s (tools/ab_wide.py): sixteen is the optimum of this structure.
template <int RM>
__global__ __launch_bounds__(BJ_NT) void k_blockjac(double *__restrict__ gscratch, int p2, int LDr, int nblk, int mblk, int step,
                                                     const int32_t *__restrict__ cflag, const int32_t *__restrict__ done,
                                                     int32_t *__restrict__ rot) {
  extern __shared__ __attribute__((aligned(16))) double sm[];   // [2][BJ_B][LDr], then nrm[2 * BJ_B]
  double *nrm = sm + (size_t)2 * BJ_B * LDr;
  __shared__ int any;
  const int mtx = blockIdx.y;
  if (cflag[mtx] != 0 || done[mtx]) return;
  // circle method over mblk (even) block slots; slot >= nblk is the dummy
  int ba, bb;
  rr_pair_w(step, blockIdx.x, mblk - 1, ba, bb);
  const bool has_a = ba < nblk, has_b = bb < nblk;
  if (!has_a && !has_b) return;
  if (!has_a) { ba = bb; }                       // a lone block: only its inner sweep (step 0)
  const bool lone = !(has_a && has_b);
  if (lone && step != 0) return;
  const int tid = threadIdx.x, grp = tid >> 4, sub = tid & 15;
  double *G = gscratch + (size_t)mtx * 2 * p2 * p2;
  const int nr = (p2 - sub + 15) >> 4;
  const double tol = (double)p2 * 2.220446049250313e-16, tol2 = tol * tol;
  if (tid == 0) any = 0;
  // ---- load: columns [ba*16, +16) -> sm[0..15], [bb*16, +16) -> sm[16..31]; columns past p2 are zero
  // (a block is BJ_B * p2 CONTIGUOUS doubles of the column-major factor; p2 and LDr are even: 16-byte moves, 8 in
  //  flight per lane -- with one 8-byte load at a time a workgroup pulls ~4 GB/s and the copy dominates the visit)
  constexpr int BJ_U = 8;
  const int half = p2 >> 1;
  for (int cblk = 0; cblk < (lone ? 1 : 2); ++cblk) {
    const int c0 = (cblk == 0 ? ba : bb) * BJ_B;
    const int ncv = min(BJ_B, p2 - c0);                    // real columns of this block
    const int n2 = ncv * half;
    const double2 *src = reinterpret_cast<const double2 *>(G + (size_t)c0 * p2);
    for (int base = tid; base < BJ_B * half; base += BJ_NT * BJ_U) {
      double2 v[BJ_U];
#pragma unroll
      for (int u = 0; u < BJ_U; ++u) {
        const int idx = base + BJ_NT * u;
        v[u] = (idx < n2) ? src[idx] : make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int u = 0; u < BJ_U; ++u) {
        const int idx = base + BJ_NT * u;
        if (idx < BJ_B * half) {
          const int cc = idx / half, r2 = idx - cc * half;
          *reinterpret_cast<double2 *>(sm + (size_t)(cblk * BJ_B + cc) * LDr + 2 * r2) = v[u];
        }
      }
    }
  }
  __syncthreads();
  const int ncolw = lone ? BJ_B : 2 * BJ_B;
  for (int j = grp; j < ncolw; j += 16) {       // exact squared norms of the columns in LDS
    double sacc = 0.0;
    for (int i = 0; i < nr; ++i) { const double x = sm[(size_t)j * LDr + sub + 16 * i]; sacc = __builtin_fma(x, x, sacc); }
    sacc = dpp_sum16(sacc);
    if (sub == 0) nrm[j] = sacc;
  }
  __syncthreads();
  bool rotated = false, big = false;
  // ---- the pairs inside each block, once per sweep: 15 steps of 8 pairs per block (groups 0-7: block a, 8-15: block b)
  if (step == 0) {
    const int blk = grp >> 3, k = grp & 7;
    const bool work = !(lone && blk == 1);
    for (int s = 0; s < BJ_B - 1; ++s) {
      if (work) {
        int a, b;
        rr_pair_w(s, k, BJ_B - 1, a, b);
        double *ga = sm + (size_t)(blk * BJ_B + a) * LDr + sub, *gb = sm + (size_t)(blk * BJ_B + b) * LDr + sub;
        double xa[RM], xb[RM];
#pragma unroll
        for (int i = 0; i < RM; ++i) {
          const int ii = min(i, nr - 1);
          const double u = ga[16 * ii], v = gb[16 * ii];
          xa[i] = i < nr ? u : 0.0;
          xb[i] = i < nr ? v : 0.0;
        }
        double ab = 0.0;
#pragma unroll
        for (int i = 0; i < RM; ++i) ab = __builtin_fma(xa[i], xb[i], ab);
        ab = dpp_sum16(ab);
        const double aa = nrm[blk * BJ_B + a], bbn = nrm[blk * BJ_B + b];
        double cs, sn;
        if (bj_rotation(aa, bbn, ab, tol2, cs, sn)) {
          rotated = true;
          big = big || (ab * ab > BJ_TINY2 * (aa * bbn));
#pragma unroll
          for (int i = 0; i < RM; ++i) {
            if (i < nr) {
              ga[16 * i] = cs * xa[i] - sn * xb[i];
              gb[16 * i] = sn * xa[i] + cs * xb[i];
            }
          }
          if (sub == 0) {
            const double cc = cs * cs, ss = sn * sn, x2 = 2.0 * cs * sn * ab;
            nrm[blk * BJ_B + a] = cc * aa - x2 + ss * bbn;
            nrm[blk * BJ_B + b] = ss * aa + x2 + cc * bbn;
          }
        }
      }
      __syncthreads();
    }
  }
  // ---- the 256 cross pairs: group i owns a_i (registers), meets b_(i + t) mod 16 at step t
  if (!lone) {
    double xa[RM];
    double *ga = sm + (size_t)grp * LDr + sub;
#pragma unroll
    for (int i = 0; i < RM; ++i) { const int ii = min(i, nr - 1); const double u = ga[16 * ii]; xa[i] = i < nr ? u : 0.0; }
    double aa = nrm[grp];
    for (int t = 0; t < BJ_B; ++t) {
      const int j = (grp + t) & (BJ_B - 1);
      double *gb = sm + (size_t)(BJ_B + j) * LDr + sub;
      double xb[RM];
#pragma unroll
      for (int i = 0; i < RM; ++i) { const int ii = min(i, nr - 1); const double v = gb[16 * ii]; xb[i] = i < nr ? v : 0.0; }
      double ab = 0.0;
#pragma unroll
      for (int i = 0; i < RM; ++i) ab = __builtin_fma(xa[i], xb[i], ab);
      ab = dpp_sum16(ab);
      const double bbn = nrm[BJ_B + j];
      double cs, sn;
      if (bj_rotation(aa, bbn, ab, tol2, cs, sn)) {
        rotated = true;
        big = big || (ab * ab > BJ_TINY2 * (aa * bbn));
#pragma unroll
        for (int i = 0; i < RM; ++i) {
          const double na = cs * xa[i] - sn * xb[i], nb = sn * xa[i] + cs * xb[i];
          xa[i] = na;
          if (i < nr) gb[16 * i] = nb;
        }
        const double cc = cs * cs, ss = sn * sn, x2 = 2.0 * cs * sn * ab;
        if (sub == 0) nrm[BJ_B + j] = ss * aa + x2 + cc * bbn;
        aa = cc * aa - x2 + ss * bbn;
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < RM; ++i) if (i < nr) ga[16 * i] = xa[i];
  }
  if (rotated) any = 1;
  __syncthreads();   // (a lane's "1" must not land on another lane's "2")
  if (big) any = 2;
  __syncthreads();
  // ---- store
  for (int cblk = 0; cblk < (lone ? 1 : 2); ++cblk) {
    const int c0 = (cblk == 0 ? ba : bb) * BJ_B;
    const int n2 = min(BJ_B, p2 - c0) * half;
    double2 *dst = reinterpret_cast<double2 *>(G + (size_t)c0 * p2);
#pragma unroll 4
    for (int idx = tid; idx < n2; idx += BJ_NT) {
      const int cc = idx / half, r2 = idx - cc * half;
      dst[idx] = *reinterpret_cast<const double2 *>(sm + (size_t)(cblk * BJ_B + cc) * LDr + 2 * r2);
    }
  }
  if (tid == 0 && any) atomicMax(&rot[mtx], any);
}

// Round 4: TWO block pairs per workgroup, four blocks per visit.  The blocks of step 0's pairs form "super-blocks" K = (a_K, b_K)
// (their inner and mutual pairs are k_blockjac's, the first launch of a sweep); the remaining pairs are those between
// super-blocks, and a circle method over the SUPER-blocks schedules them: a visit of the super-pair (K, L) holds a_K and b_K
// in the registers of its two 256-thread teams and a_L, b_L in LDS, rotates (a_K, a_L) | (b_K, b_L) and then, the teams swapping
// their LDS block, (a_K, b_L) | (b_K, a_L).  Four block pairs per four blocks moved instead of one per two: half the memory
// traffic per rotation (the visits of a step move 5 TB/s, profiles/r04_wjac_phase_clocks.txt), half the launches per sweep
// (14 instead of 27 at p = 425).  Every pair of columns still meets exactly once per sweep; same rotation formula, threshold
// and carried norms as k_blockjac_x.
template <int RM>
__global__ __launch_bounds__(2 * BJ_NT) void k_blockjac_q(double *__restrict__ gscratch, int p2, int LDr, int nblk, int mblk, int sstep,
                                                           const int32_t *__restrict__ cflag, const int32_t *__restrict__ done,
                                                           int32_t *__restrict__ rot) {
  extern __shared__ __attribute__((aligned(16))) double sm[];   // [2][BJ_B][LDr] (the blocks of L), then nrm[2][BJ_B]
  double *nrm = sm + (size_t)2 * BJ_B * LDr;
  __shared__ int any;
  const int mtx = blockIdx.y;
  if (cflag[mtx] != 0 || done[mtx]) return;
  const int msb = mblk >> 1, msbE = msb + (msb & 1);     // super-blocks; an even number of circle slots
  int K, Ls;
  rr_pair_w(sstep, blockIdx.x, msbE - 1, K, Ls);
  if (K >= msb || Ls >= msb) return;
  int blkA[2], blkB[2];
  rr_pair_w(0, K, mblk - 1, blkA[0], blkA[1]);
  rr_pair_w(0, Ls, mblk - 1, blkB[0], blkB[1]);
  const int tid = threadIdx.x, team = tid >> 8, grp = (tid >> 4) & 15, sub = tid & 15;
  double *G = gscratch + (size_t)mtx * 2 * p2 * p2;
  const int nr = (p2 - sub + 15) >> 4;
  const double tol = (double)p2 * 2.220446049250313e-16, tol2 = tol * tol;
  if (tid == 0) any = 0;
  constexpr int BJ_U = 8;
  const int half = p2 >> 1;
  // ---- both blocks of L -> LDS (a block index past the last block: zeros, never stored)
  for (int slot = 0; slot < 2; ++slot) {
    const int c0 = blkB[slot] * BJ_B;
    const int n2 = (blkB[slot] < nblk ? min(BJ_B, p2 - c0) : 0) * half;
    const double2 *src = reinterpret_cast<const double2 *>(G + (size_t)min(c0, p2 - 1) * p2);
    for (int base = tid; base < BJ_B * half; base += 2 * BJ_NT * BJ_U) {
      double2 v[BJ_U];
#pragma unroll
      for (int u = 0; u < BJ_U; ++u) {
        const int idx = base + 2 * BJ_NT * u;
        v[u] = (idx < n2) ? src[idx] : make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int u = 0; u < BJ_U; ++u) {
        const int idx = base + 2 * BJ_NT * u;
        if (idx < BJ_B * half) {
          const int cc = idx / half, r2 = idx - cc * half;
          *reinterpret_cast<double2 *>(sm + (size_t)(slot * BJ_B + cc) * LDr + 2 * r2) = v[u];
        }
      }
    }
  }
  // ---- this team's block of K: group grp keeps column a_grp in registers for the whole visit
  const int ablk = blkA[team];
  const int acol = ablk * BJ_B + grp;
  const bool areal = ablk < nblk && acol < p2;
  double *ga = G + (size_t)min(acol, p2 - 1) * p2 + sub;
  double xa[RM];
#pragma unroll
  for (int i = 0; i < RM; ++i) {
    const int ii = min(i, nr - 1);
    const double u = ga[16 * ii];
    xa[i] = (i < nr && areal) ? u : 0.0;
  }
  double aa = 0.0;
#pragma unroll
  for (int i = 0; i < RM; ++i) aa = __builtin_fma(xa[i], xa[i], aa);
  aa = dpp_sum16(aa);
  __syncthreads();
  {
    // exact squared norms of the LDS columns: team t sums slot t
    double sacc = 0.0;
    for (int i = 0; i < nr; ++i) { const double x = sm[(size_t)(team * BJ_B + grp) * LDr + sub + 16 * i]; sacc = __builtin_fma(x, x, sacc); }
    sacc = dpp_sum16(sacc);
    if (sub == 0) nrm[team * BJ_B + grp] = sacc;
  }
  __syncthreads();
  bool rotated = false, big = false;
#pragma unroll 1
  for (int ss = 0; ss < 2; ++ss) {
    const int slot = team ^ ss;                 // sub-step 0: (a_K, a_L) | (b_K, b_L); sub-step 1: the teams swap their LDS block
    double *bbase = sm + (size_t)slot * BJ_B * LDr;
    double *nb = nrm + slot * BJ_B;
    for (int t = 0; t < BJ_B; ++t) {
      const int j = (grp + t) & (BJ_B - 1);
      double *gb = bbase + (size_t)j * LDr + sub;
      double xb[RM];
#pragma unroll
      for (int i = 0; i < RM; ++i) { const int ii = min(i, nr - 1); const double v = gb[16 * ii]; xb[i] = i < nr ? v : 0.0; }
      double ab = 0.0;
#pragma unroll
      for (int i = 0; i < RM; ++i) ab = __builtin_fma(xa[i], xb[i], ab);
      ab = dpp_sum16(ab);
      const double bbn = nb[j];
      double cs, sn;
      if (bj_rotation(aa, bbn, ab, tol2, cs, sn)) {
        rotated = true;
        big = big || (ab * ab > BJ_TINY2 * (aa * bbn));
#pragma unroll
        for (int i = 0; i < RM; ++i) {
          const double na = cs * xa[i] - sn * xb[i], nbv = sn * xa[i] + cs * xb[i];
          xa[i] = na;
          if (i < nr) gb[16 * i] = nbv;
        }
        const double cc = cs * cs, s2 = sn * sn, x2 = 2.0 * cs * sn * ab;
        if (sub == 0) nb[j] = s2 * aa + x2 + cc * bbn;
        aa = cc * aa - x2 + s2 * bbn;
      }
      __syncthreads();
    }
  }
  if (rotated) any = 1;
  __syncthreads();   // (a lane's "1" must not land on another lane's "2")
  if (big) any = 2;
  if (areal) {
#pragma unroll
    for (int i = 0; i < RM; ++i) if (i < nr) ga[16 * i] = xa[i];
  }
  __syncthreads();
  for (int slot = 0; slot < 2; ++slot) {
    if (blkB[slot] >= nblk) continue;
    const int c0 = blkB[slot] * BJ_B;
    const int n2 = min(BJ_B, p2 - c0) * half;
    double2 *dst = reinterpret_cast<double2 *>(G + (size_t)c0 * p2);
#pragma unroll 4
    for (int idx = tid; idx < n2; idx += 2 * BJ_NT) {
      const int cc = idx / half, r2 = idx - cc * half;
      dst[idx] = *reinterpret_cast<const double2 *>(sm + (size_t)(slot * BJ_B + cc) * LDr + 2 * r2);
    }
  }
  if (tid == 0 && any) atomicMax(&rot[mtx], any);
}

// after a sweep: a matrix without a rotation is finished -- and so is one whose rotations were all TINY (|a.b| <= 1e-9 |a||b|
// for every pair that rotated; rot = 1): each of those pairs is now orthogonal to working precision and disturbed the others by
// the square of that ratio, so the sweep that would follow finds nothing above the tolerance (it is the verification sweep a
// cyclic Jacobi otherwise pays: one of the 12 at p = 425).  rot = 2: some rotation was larger -- another sweep.
__global__ void k_blockjac_flags(int nb, int32_t *__restrict__ done, int32_t *__restrict__ rot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nb) return;
  if (rot[i] < 2) done[i] = 1;
  rot[i] = 0;
}

// a matrix still rotating after the last sweep goes to the single-workgroup kernel (mode 2), which has its own cap
__global__ void k_blockjac_leftover(int nb, int32_t *__restrict__ cflag, const int32_t *__restrict__ done) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nb && cflag[i] == 0 && !done[i]) cflag[i] = 1;
}

// eigenvalues = squared column norms of the orthogonalised factor, eigenvectors = normalised columns
__global__ __launch_bounds__(512) void k_blockjac_finish(const double *__restrict__ gscratch, int p, int p2, int c0,
                                                          const int32_t *__restrict__ cflag, double *__restrict__ lam_out,
                                                          double *__restrict__ evec_out) {
  __shared__ double nrm[512];
  if (cflag[blockIdx.x] != 0) return;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int c = c0 + blockIdx.x;
  const double *G = gscratch + (size_t)blockIdx.x * 2 * p2 * p2;
  for (int j = tid; j < p; j += nthr) {
    double sacc = 0;
    for (int r = 0; r < p2; ++r) { const double x = G[(size_t)j * p2 + r]; sacc += x * x; }
    nrm[j] = sacc;
    lam_out[(size_t)c * p + j] = sacc;
  }
  __syncthreads();
  for (int i = tid; i < p * p; i += nthr) {
    const int j = i / p, b = i - j * p;
    const double s2 = nrm[j];
    evec_out[(size_t)c * p * p + i] = s2 > 0.0 ? G[(size_t)j * p2 + b] / sqrt(s2) : ((j == b) ? 1.0 : 0.0);
  }
}

// ---- full shrinkage target on a wide window (-f, robust_mf.py:354; looshrinkage's T = cov(I_reg), :99, :131) ---------
// cmf_general.hip has the algebra and the 96-band version with everything in LDS.  Here the matrices stay in global
// memory (L2-resident, 1.4 MB each at p = 425):  T = L L^T by the blocked Cholesky above,  R = L^-1 S L^-T  by two
// forward substitutions with one thread per right-hand side (the right-hand sides are the ROWS of the row-major buffer,
// so a wave reads 64 consecutive doubles per step and the L entry is wave-uniform),  eigenpairs of R by the blocked
// Jacobi in `unit` mode,  evec_j <- D L^-T v_j  by one backward substitution, d = diag(L).
__global__ void k_wg_load(const double *__restrict__ target, const int32_t *__restrict__ nuse, int p, int p2, int c0,
                          double *__restrict__ gscratch, int32_t *__restrict__ cflag) {
  const int mtx = blockIdx.y, c = c0 + mtx;
  const double *T = target + (size_t)c * p * p;
  double *G = gscratch + (size_t)mtx * 2 * p2 * p2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < p2 * p2; i += gridDim.x * blockDim.x) {
    const int col = i / p2, row = i - col * p2;
    G[i] = (col < p && row < p) ? T[(size_t)row * p + col] : 0.0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) cflag[mtx] = nuse[c] > 0 ? 0 : 2;
}
// Lc[k*p + i] = L[i][k] (column-major), B = S, d = diag(L); a target that is not positive definite makes every G_a
// singular together with S (fewer rows than bands): L = B = I, d = 1 and tflag = 1 marks the column for k_wg_back_out
__global__ void k_wg_prep(const double *__restrict__ gscratch, const int32_t *__restrict__ cflag, const double *__restrict__ cov,
                          int p, int p2, int c0, double *__restrict__ Lc, double *__restrict__ B, double *__restrict__ d_out,
                          int32_t *__restrict__ tflag) {
  const int mtx = blockIdx.y, c = c0 + mtx;
  const double *G = gscratch + (size_t)mtx * 2 * p2 * p2, *S = cov + (size_t)c * p * p;
  double *Lm = Lc + (size_t)mtx * p * p, *Bm = B + (size_t)mtx * p * p;
  const int flag = cflag[mtx];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < p * p; i += gridDim.x * blockDim.x) {
    const int a = i / p, b = i - a * p;
    if (flag == 0) {
      Lm[i] = G[(size_t)a * p2 + b];      // column a, row b (upper part zeroed by k_chol_clean)
      Bm[i] = S[i];
    } else {
      Lm[i] = Bm[i] = (a == b) ? 1.0 : 0.0;
    }
    if (a == b) d_out[(size_t)c * p + a] = (flag == 0) ? G[(size_t)a * p2 + a] : 1.0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) tflag[mtx] = flag;
}
// B <- L^-1 B (UPPER = false) or L^-T B (true), column t of B by thread t; B row-major [p][p]
template <bool UPPER>
__global__ __launch_bounds__(64) void k_wg_solve(const double *__restrict__ Lc, double *__restrict__ B, int p) {
  const int mtx = blockIdx.y, t = blockIdx.x * 64 + threadIdx.x;
  if (t >= p) return;
  const double *Lm = Lc + (size_t)mtx * p * p;
  double *Bm = B + (size_t)mtx * p * p + t;
  if (!UPPER) {
    for (int i = 0; i < p; ++i) {         // x_i = (b_i - sum_{k < i} L[i][k] x_k) / L[i][i]
      double a0 = Bm[(size_t)i * p], a1 = 0.0, a2 = 0.0, a3 = 0.0;
      int k = 0;
      for (; k + 4 <= i; k += 4) {
        a0 = __builtin_fma(-Lm[(size_t)k * p + i], Bm[(size_t)k * p], a0);
        a1 = __builtin_fma(-Lm[(size_t)(k + 1) * p + i], Bm[(size_t)(k + 1) * p], a1);
        a2 = __builtin_fma(-Lm[(size_t)(k + 2) * p + i], Bm[(size_t)(k + 2) * p], a2);
        a3 = __builtin_fma(-Lm[(size_t)(k + 3) * p + i], Bm[(size_t)(k + 3) * p], a3);
      }
      for (; k < i; ++k) a0 = __builtin_fma(-Lm[(size_t)k * p + i], Bm[(size_t)k * p], a0);
      Bm[(size_t)i * p] = ((a0 + a1) + (a2 + a3)) / Lm[(size_t)i * p + i];
    }
  } else {
    for (int i = p - 1; i >= 0; --i) {    // w_i = (v_i - sum_{k > i} L[k][i] w_k) / L[i][i]
      double a0 = Bm[(size_t)i * p], a1 = 0.0, a2 = 0.0, a3 = 0.0;
      const double *Li = Lm + (size_t)i * p;
      int k = i + 1;
      for (; k + 4 <= p; k += 4) {
        a0 = __builtin_fma(-Li[k], Bm[(size_t)k * p], a0);
        a1 = __builtin_fma(-Li[k + 1], Bm[(size_t)(k + 1) * p], a1);
        a2 = __builtin_fma(-Li[k + 2], Bm[(size_t)(k + 2) * p], a2);
        a3 = __builtin_fma(-Li[k + 3], Bm[(size_t)(k + 3) * p], a3);
      }
      for (; k < p; ++k) a0 = __builtin_fma(-Li[k], Bm[(size_t)k * p], a0);
      Bm[(size_t)i * p] = ((a0 + a1) + (a2 + a3)) / Li[i];
    }
  }
}
// in place: B <- B^T (sym = 0) or (B + B^T) / 2 (sym = 1)
__global__ void k_wg_transpose(double *__restrict__ B, int p, int sym) {
  double *Bm = B + (size_t)blockIdx.y * p * p;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < p * p; i += gridDim.x * blockDim.x) {
    const int a = i / p, b = i - a * p;
    if (b >= a) continue;
    const double x = Bm[(size_t)a * p + b], y = Bm[(size_t)b * p + a];
    Bm[(size_t)a * p + b] = sym ? 0.5 * (x + y) : y;
    Bm[(size_t)b * p + a] = sym ? 0.5 * (x + y) : x;
  }
}
// B[b][j] = evec[c][j][b]: the eigenvectors as the right-hand sides of L^T w = v
__global__ void k_wg_back_in(const double *__restrict__ evec, int p, int c0, double *__restrict__ B) {
  const int mtx = blockIdx.y, c = c0 + mtx;
  const double *ev = evec + (size_t)c * p * p;
  double *Bm = B + (size_t)mtx * p * p;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < p * p; i += gridDim.x * blockDim.x) {
    const int b = i / p, j = i - b * p;
    Bm[i] = ev[(size_t)j * p + b];
  }
}
// evec[c][j][b] = d_b w_j[b];  status 2 (and lam = 0) where the Cholesky of the target failed
__global__ void k_wg_back_out(const double *__restrict__ B, int p, int c0, const double *__restrict__ d, double *__restrict__ evec,
                              double *__restrict__ lam, int32_t *__restrict__ status, const int32_t *__restrict__ tflag) {
  const int mtx = blockIdx.y, c = c0 + mtx;
  if (tflag[mtx] == 1) {
    if (blockIdx.x == 0) {
      if (threadIdx.x == 0 && status[c] == 0) status[c] = 2;
      for (int i = threadIdx.x; i < p; i += blockDim.x) lam[(size_t)c * p + i] = 0.0;
    }
    return;
  }
  if (status[c] != 0) return;
  const double *Bm = B + (size_t)mtx * p * p, *dc = d + (size_t)c * p;
  double *ev = evec + (size_t)c * p * p;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < p * p; i += gridDim.x * blockDim.x) {
    const int j = i / p, b = i - j * p;
    ev[i] = Bm[(size_t)b * p + j] * dc[b];
  }
}

}  // namespace

// Exact determinants (linalg.hip): every grid point while that is cheap (the function-level looshrinkage(): one
// column, 201 factorisations side by side on 256 CUs), else the 24 grid points on the finite side of each crossing of
// the float64 range (the prefix products of p <= 512 pivots were never seen to run further ahead of the total than 16
// grid points, tests/test_cmf_gpu.py::test_looshrinkage_function_512_band_golden).
int sf_exact_det_window(const SfGeom &g) { return ((size_t)g.ncols * g.nalpha <= 2048) ? 0 : 24; }
// ---- the fused route of round 4 (cmf_wgemm.hip): no float64 copies of X~, Z, r -- a column needs its eigensolver work
// matrices (2 p2^2), the sweep operands W and C, and, on the full-target route, L and B.  All columns of a flightline in one group when 8 GB hold them.
static size_t fused_col_bytes(const SfGeom &g) {
  const size_t p = g.p, p2 = g.p + (g.p & 1);
  SfGeom one = g;
  one.ncols = 1;
  return sf_align(2 * p2 * p2 * sizeof(double)) + sf_wgemm_operand_bytes(one) + sf_align(2 * p * p * sizeof(double)) +
         sf_wtri_small_bytes(g.p, 1) + 256;
}
int sf_wide_group(const SfGeom &g) {
  // The eigensolver streams every factor (2 p2^2 doubles are allocated, p2^2 are live) once per Jacobi step: a group whose
  // factors fit the 256 MB Infinity Cache runs its visits ~15 % faster than one that spills to HBM (598 matrices of p = 425:
  // 38.5 us per round of 512 workgroups against 33 with 200), and 13 x 150 workgroups are 3.8 rounds, paid as 4.
  const size_t p2 = g.p + (g.p & 1);
  size_t b = (size_t)230e6 / (p2 * p2 * sizeof(double));
  const size_t bmem = ((size_t)8 << 30) / fused_col_bytes(g);
  if (b > bmem) b = bmem;
  if (b < 1) b = 1;
  if (b > (size_t)g.ncols) b = g.ncols;
  const int ngroups = sf_cdiv(g.ncols, (int)b);
  return sf_cdiv(g.ncols, ngroups);   // balanced groups
}
static size_t fused_scratch_bytes(const SfGeom &g) {
  const size_t gb = sf_wide_group(g), ngr = sf_cdiv(g.ncols, (int)gb);
  return ngr * gb * fused_col_bytes(g) + sf_wgemm_part_bytes(g) + sf_align((size_t)g.ncols * g.nalpha * sizeof(double)) +
         sf_exact_det_scratch_bytes(g, sf_exact_det_window(g)) + 4096;
}
size_t sf_wide_scratch_bytes(const SfGeom &g) { return fused_scratch_bytes(g); }   // (round 5: round 3's route and its ~3 GB larger scratch are gone)

// Batched n x n float64 GEMM on COLUMN-major matrices through k_dgemm (whose operands are row-major: a column-major X is the
// row-major X^T):  tb = 0: C = Bm Am;  tb = 1: C = Bm^T Am ... with Am / Bm the column-major matrices in A / B -- i.e. the
// row-major product C^T = A^T-view x B-view.  Used by the tridiagonal preconditioner (cmf_wtri.hip):
//   W = L^T U:  A = U, B = L, tb = 1      G = W^T W:  A = W, B = W, tb = 1      W' = W M:  A = M, B = W, tb = 0
int sf_wide_dgemm(const double *A, int lda, size_t sA, const double *B, int ldb, size_t sB, int tb, double *C, int ldc, size_t sC,
                  int n, int nb, const int32_t *skip1, const int32_t *skip2, hipStream_t st, int lower) {
  // lower: Bm is lower-triangular (the Cholesky factor): its zero half is skipped (tb = 0: C = Bm Am; tb = 1: C = Bm^T Am)
  const dim3 grid(sf_cdiv(n, WD_BM), sf_cdiv(n, WD_BN), nb);
  if (lower && tb)
    hipLaunchKernelGGL((k_dgemm<false, false, false, true>), grid, dim3(256), 0, st, A, lda, sA, B, ldb, sB, C, ldc, sC, n, n, n, 1.0,
                       skip1, skip2, 2);
  else if (lower)
    hipLaunchKernelGGL((k_dgemm<false, false, false, false>), grid, dim3(256), 0, st, A, lda, sA, B, ldb, sB, C, ldc, sC, n, n, n, 1.0,
                       skip1, skip2, 1);
  else if (tb && A == B)   // C = Am^T Am: symmetric, the tiles above the diagonal are mirrored
    hipLaunchKernelGGL((k_dgemm<false, false, true, true>), grid, dim3(256), 0, st, A, lda, sA, B, ldb, sB, C, ldc, sC, n, n, n, 1.0,
                       skip1, skip2);
  else if (tb)
    hipLaunchKernelGGL((k_dgemm<false, false, false, true>), grid, dim3(256), 0, st, A, lda, sA, B, ldb, sB, C, ldc, sC, n, n, n, 1.0,
                       skip1, skip2);
  else
    hipLaunchKernelGGL((k_dgemm<false, false, false, false>), grid, dim3(256), 0, st, A, lda, sA, B, ldb, sB, C, ldc, sC, n, n, n, 1.0,
                       skip1, skip2);
  SF_LAUNCH_CHECK("k_dgemm(wtri)");
  return 0;
}

// blocked Cholesky of the nb matrices in gv whose flag is 0 (flag -> 1 where a pivot is not positive)
static int wide_chol(double *gv, int p, int p2, int nb, int32_t *cflag, hipStream_t st, int gbn = 0, size_t gstride = 0) {
  if (gbn <= 0) { gbn = nb > 0 ? nb : 1; gstride = 0; }
  const size_t plds = (size_t)CH_B * p * sizeof(double);
  if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_chol_panel), plds)) return rc;
  const int npan = sf_cdiv(p, CH_B);
  for (int kb = 0; kb < npan; ++kb) {
    hipLaunchKernelGGL(k_chol_panel, dim3(nb), dim3(256), plds, st, gv, p, p2, kb, cflag, gbn, gstride);
    const int rem = p - (kb + 1) * CH_B;
    if (rem > 0) {
      const int nt = sf_cdiv(rem, 64);
      hipLaunchKernelGGL(k_chol_trail, dim3(nt, nt, nb), dim3(256), 0, st, gv, p, p2, kb, cflag, gbn, gstride);
    }
  }
  hipLaunchKernelGGL(k_chol_clean, dim3(64, nb), dim3(256), 0, st, gv, p, p2, cflag, gbn, gstride);
  SF_LAUNCH_CHECK("k_chol");
  return 0;
}

// eigendecomposition of a batch of nb correlation matrices (columns c0 .. c0+nb-1)
// (unit: `cov` = the batch's nb whitened matrices, no diagonal scaling, d untouched -- the full-target route)
// pre != nullptr: the tridiagonal preconditioner of cmf_wtri.hip between the Cholesky and the sweeps (work matrices and flags of
// the group: pre->B2, B3 [nb][p^2], small, pflag)
// The route is a function of p ONLY: the preconditioned and the plain sweeps apply different rotation sequences, so a
// column's eigenvectors -- and with them its NLL and scores -- must not depend on how many columns share the call (cmf_common.h:
// a column is bit-identical as part of the flightline or of any shard, however narrow).  A call of a handful of matrices pays the
// tridiagonalisation's ~9 ms of latency where the plain sweeps would be launch-bound at ~6 ms.  sf_debug_set(10, 6) = plain sweeps.
static bool wide_precond_on(int p, int ncols) {
  (void)ncols;
  const int v = sf_tune().wide_eigh_variant;
  return p >= 128 && (v == 0 || v == 7 || v == 8);   // (8: every preconditioner is refused afterwards: the fallback's test)
}
// stage: 0 = all of it; 1 = only the work matrices (R, d, flags: k_eigh_global mode 3) -- the caller then runs the preconditioner's
// first half and the Cholesky over ALL groups at once (sf_launch_wtri_prepare, wide_chol) -- and 2 = the rest (second half, sweeps, finish)
struct WidePre { double *B2, *B3, *small; int32_t *pflag; };
static int wide_eigh(const double *cov, const int32_t *nuse, int p, int p2, int c0, int nb, double *d, double *lam, double *evec,
                     int32_t *status, double *gv, int32_t *cflag, int32_t *done, int32_t *rot, hipStream_t st, int unit = 0,
                     const WidePre *pre = nullptr, int stage = 0) {
  if (sf_tune().wide_eigh_variant == 1) {
    hipLaunchKernelGGL(k_eigh_global, dim3(nb), dim3(512), 0, st, cov, nuse, p, p2, c0, d, lam, evec, status, gv, 0, cflag, unit);
    SF_LAUNCH_CHECK("k_eigh_global");
    return 0;
  }
  if (stage != 2) {
    hipLaunchKernelGGL(k_eigh_global, dim3(nb), dim3(512), 0, st, cov, nuse, p, p2, c0, d, lam, evec, status, gv, 3, cflag, unit);
    SF_LAUNCH_CHECK("k_eigh_global(prep)");
  }
  if (stage == 1) return 0;
  const bool precond = pre && (stage == 2 || wide_precond_on(p, nb));
  if (precond && stage == 0)
    if (int rc = sf_launch_wtri_prepare(gv, p, p2, nb, pre->B2, pre->B3, pre->small, cflag, pre->pflag, st, 0, 0)) return rc;
  if (stage != 2)
    if (int rc = wide_chol(gv, p, p2, nb, cflag, st)) return rc;
  if (precond)
    if (int rc = sf_launch_wtri_apply(gv, p, p2, nb, pre->B2, pre->B3, pre->small, cflag, pre->pflag, st)) return rc;
  SF_HIP(hipMemsetAsync(done, 0, (size_t)nb * sizeof(int32_t), st));
  SF_HIP(hipMemsetAsync(rot, 0, (size_t)nb * sizeof(int32_t), st));
  const int nblk = sf_cdiv(p2, BJ_B), mblk = nblk + (nblk & 1);
  int LDr = p2;
  while ((LDr % 32) != 16) ++LDr;
  const size_t lds = ((size_t)2 * BJ_B * LDr + 2 * BJ_B) * sizeof(double);
  const int nrmax = sf_cdiv(p2, 16);
#define BJ_SWEEPS(RM)                                                                                                            \
  {                                                                                                                              \
    if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_blockjac<RM>), lds)) return rc;                                    \
    if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_blockjac_q<RM>), lds)) return rc;                                  \
    for (int sweep = 0; sweep < nsweeps; ++sweep) { /* converged matrices drop out by their flag; no host round trip */          \
      /* (10-12 sweeps on flightline-like spectra; a matrix still rotating after 16 is redone by k_eigh_global, mode 2) */        \
      hipLaunchKernelGGL(k_blockjac<RM>, dim3(mblk / 2 > 0 ? mblk / 2 : 1, nb), dim3(BJ_NT), lds, st, gv, p2, LDr, nblk,         \
                         mblk > 1 ? mblk : 2, 0, cflag, done, rot);                                                              \
      for (int s = 1; s < msbE; ++s)                                                                                             \
        hipLaunchKernelGGL(k_blockjac_q<RM>, dim3(msbE / 2, nb), dim3(2 * BJ_NT), lds, st, gv, p2, LDr, nblk, mblk, s, cflag,    \
                           done, rot);                                                                                           \
      hipLaunchKernelGGL(k_blockjac_flags, dim3(sf_cdiv(nb, 256)), dim3(256), 0, st, nb, done, rot);                             \
    }                                                                                                                            \
  }
  // behind the preconditioner a matrix needs ONE sweep (two or three if its preconditioner was poor: measured cosines 1e-11, a
  // 6-decade spectrum 1e-9); the launches of sweeps nobody needs are ~4.5 us each, 15 a sweep and group.  A matrix still rotating
  // after the last sweep -- one whose preconditioner was refused (pflag) and whose Cholesky still succeeded -- is redone by
  // k_eigh_global (mode 2), as after 16 sweeps without the preconditioner.
  const int nsweeps = precond ? 5 : 16;
  // quad visits (k_blockjac_q) for the cross steps of a sweep (round 3's pair visits, k_blockjac_x, were removed in round 5)
  const int msb = mblk / 2, msbE = msb + (msb & 1);
  if (nrmax <= 8) BJ_SWEEPS(8)
  else if (nrmax <= 16) BJ_SWEEPS(16)
  else if (nrmax <= 24) BJ_SWEEPS(24)
  else if (nrmax <= 28) BJ_SWEEPS(28)
  else BJ_SWEEPS(32)
#undef BJ_SWEEPS
  SF_LAUNCH_CHECK("k_blockjac");
  hipLaunchKernelGGL(k_blockjac_leftover, dim3(sf_cdiv(nb, 256)), dim3(256), 0, st, nb, cflag, done);
  hipLaunchKernelGGL(k_blockjac_finish, dim3(nb), dim3(512), 0, st, gv, p, p2, c0, cflag, lam, evec);
  SF_LAUNCH_CHECK("k_blockjac_finish");
  hipLaunchKernelGGL(k_eigh_global, dim3(nb), dim3(512), 0, st, cov, nuse, p, p2, c0, d, lam, evec, status, gv, 2, cflag, unit);
  SF_LAUNCH_CHECK("k_eigh_global(fallback)");
  return 0;
}

// Round 4: covariance and sweep by the fused 4x4x4 kernels of cmf_wgemm.hip, the eigensolver on groups of up to 200 columns.
static int wide_stats_fused(const void *xt, int xt_f64, const uint8_t *mask_t, const int32_t *nuse, const int32_t *nloo,
                            const double *mu, const double *alphas, const SfGeom &g, double *cov, double *d, double *lam,
                            double *evec, int32_t *status, double *nll, int32_t *alphaidx, void *scratch, hipStream_t st,
                            const double *target) {
  const int p = g.p, p2 = g.p + (g.p & 1);
  const int gb = sf_wide_group(g), ngr = sf_cdiv(g.ncols, gb);
  const size_t per = fused_col_bytes(g);
  SfGeom one = g;
  one.ncols = 1;
  const size_t gv_b = sf_align(2 * (size_t)p2 * p2 * sizeof(double)), op_b = sf_wgemm_operand_bytes(one);
  char *base = reinterpret_cast<char *>(scratch);
  double *part = reinterpret_cast<double *>(base + (size_t)ngr * gb * per);
  double *rest = reinterpret_cast<double *>(reinterpret_cast<char *>(part) + sf_wgemm_part_bytes(g));
  void *det_scratch = reinterpret_cast<char *>(rest) + sf_align((size_t)g.ncols * g.nalpha * sizeof(double));
  // Unimodal route with the tridiagonal preconditioner: covariances and work matrices of every group first, then the
  // preconditioner's first half (tridiagonalisation, bisection: latency-bound per workgroup) over ALL columns in one set of
  // launches, then group by group the Cholesky, the second half, the sweeps and the LOO sweep.  phase 0: one pass as before.
  const bool phased = !target && wide_precond_on(p, g.ncols) && ngr > 1;
  for (int phase = phased ? 1 : 0; phase <= (phased ? 2 : 0); ++phase) {
  if (phase == 2) {
    char *g0 = base;
    double *gv0 = reinterpret_cast<double *>(g0);
    double *Lc0 = reinterpret_cast<double *>(g0 + (size_t)gb * (gv_b + op_b));
    int32_t *fl0 = reinterpret_cast<int32_t *>(g0 + (size_t)gb * (per - 256));
    double *small0 = reinterpret_cast<double *>(g0 + (size_t)gb * (per - 256 - sf_wtri_small_bytes(g.p, 1)));
    if (int rc = sf_launch_wtri_prepare(gv0, p, p2, ngr * gb, Lc0, Lc0 + (size_t)gb * p * p, small0, fl0, fl0 + 4 * gb, st, gb,
                                        (size_t)gb * per))
      return rc;
    if (int rc = wide_chol(gv0, p, p2, ngr * gb, fl0, st, gb, (size_t)gb * per)) return rc;
  }
  // (phased route: ONE covariance launch over all columns instead of one per group -- fewer partly filled last rounds of the
  //  chip's 512 workgroup slots: 70.0 -> 68.4 ms a flightline, same bits)
  if (phase == 1)
    if (int rc = sf_launch_wsyrk(xt, xt_f64, mask_t, nuse, mu, g, 0, g.ncols, cov, st)) return rc;
  for (int gi = 0; gi < ngr; ++gi) {
    const int c0 = gi * gb, nb = (g.ncols - c0 < gb) ? g.ncols - c0 : gb;
    char *gbase = base + (size_t)gi * gb * per;
    // group layout: gv [nb][2 p2^2] | operands (W | Ct for nb columns) | Lc, Bw [nb][p^2] | flags
    double *gv = reinterpret_cast<double *>(gbase);
    void *opnd = gbase + (size_t)gb * gv_b;
    double *Lc = reinterpret_cast<double *>(gbase + (size_t)gb * (gv_b + op_b));
    double *Bw = Lc + (size_t)gb * p * p;
    int32_t *flags = reinterpret_cast<int32_t *>(gbase + (size_t)gb * (per - 256));   // cflag | done | rot | tflag | pflag, gb each
    int32_t *tflag = flags + 3 * gb;
    static_assert(sizeof(int32_t) == 4, "flags");
    if ((size_t)5 * gb * sizeof(int32_t) > (size_t)gb * 256) return -2;   // (20 bytes of flags per column: always true)
    // the preconditioner's work: the full-target route's L / B matrices (free on the unimodal route) and d, e, t, scales
    double *small = reinterpret_cast<double *>(gbase + (size_t)gb * (per - 256 - sf_wtri_small_bytes(g.p, 1)));
    const WidePre pre{Lc, Bw, small, flags + 4 * gb};
    // gv of matrix i must sit at gv + i * 2 p2^2 (the eigensolver kernels index it that way): gv_b may be padded, so the
    // group's gv block is addressed densely and simply has to fit
    if (phase == 0)
      if (int rc = sf_launch_wsyrk(xt, xt_f64, mask_t, nuse, mu, g, c0, nb, cov, st)) return rc;
    if (!target) {
      if (phase == 1 && nb < gb) {   // (the all-groups launches walk gb slots per group: the unused ones of a short group are flagged)
        SF_HIP(hipMemsetAsync(flags + nb, 0x01, (size_t)(gb - nb) * sizeof(int32_t), st));
      }
      if (int rc = wide_eigh(cov, nuse, p, p2, c0, nb, d, lam, evec, status, gv, flags, flags + gb, flags + 2 * gb, st, 0, &pre, phase))
        return rc;
      if (phase == 1) continue;
    } else {
      hipLaunchKernelGGL(k_wg_load, dim3(64, nb), dim3(256), 0, st, target, nuse, p, p2, c0, gv, flags);
      SF_LAUNCH_CHECK("k_wg_load");
      if (int rc = wide_chol(gv, p, p2, nb, flags, st)) return rc;
      hipLaunchKernelGGL(k_wg_prep, dim3(64, nb), dim3(256), 0, st, gv, flags, cov, p, p2, c0, Lc, Bw, d, tflag);
      hipLaunchKernelGGL(k_wg_solve<false>, dim3(sf_cdiv(p, 64), nb), dim3(64), 0, st, Lc, Bw, p);
      hipLaunchKernelGGL(k_wg_transpose, dim3(64, nb), dim3(256), 0, st, Bw, p, 0);
      hipLaunchKernelGGL(k_wg_solve<false>, dim3(sf_cdiv(p, 64), nb), dim3(64), 0, st, Lc, Bw, p);
      hipLaunchKernelGGL(k_wg_transpose, dim3(64, nb), dim3(256), 0, st, Bw, p, 1);
      SF_LAUNCH_CHECK("k_wg_whiten");
      if (int rc = wide_eigh(Bw, nuse, p, p2, c0, nb, d, lam, evec, status, gv, flags, flags + gb, flags + 2 * gb, st, 1)) return rc;
      hipLaunchKernelGGL(k_wg_back_in, dim3(64, nb), dim3(256), 0, st, evec, p, c0, Bw);
      hipLaunchKernelGGL(k_wg_solve<true>, dim3(sf_cdiv(p, 64), nb), dim3(64), 0, st, Lc, Bw, p);
      hipLaunchKernelGGL(k_wg_back_out, dim3(64, nb), dim3(256), 0, st, Bw, p, c0, d, evec, lam, status, tflag);
      SF_LAUNCH_CHECK("k_wg_back");
    }
    if (int rc = sf_launch_wsweep(xt, xt_f64, mask_t, nloo, mu, d, lam, evec, status, alphas, g, c0, nb, opnd, part, st)) return rc;
  }
  }
  if (int rc = sf_launch_nll_finish(part, sf_wgemm_splits(g), nloo, d, lam, status, alphas, g, nll, alphaidx, st, rest)) return rc;
  return sf_launch_exact_det(cov, nloo, status, alphas, g, sf_exact_det_window(g), rest, nll, alphaidx, det_scratch, st, target);
}

// Eigenpairs of symmetric positive definite matrices of 97 .. 512 rows AS THEY ARE (no correlation scaling): the wide eigensolver
// (blocked Cholesky, blocked one-sided Jacobi on the factor; the single-workgroup solver for a matrix whose Cholesky fails) on
// caller-provided matrices -- the host-side eig() of the reference's PCA (cmf/robust_mf.py:78-84, :310-312 with -R -k 2: p = 416).
extern "C" size_t sf_cmf_eigh_wide_scratch_bytes(int p, int ncols) {
  if (p < 1 || ncols < 1) return 0;
  const int p2 = p + (p & 1);
  return sf_align((size_t)ncols * 2 * p2 * p2 * sizeof(double)) + sf_align((size_t)4 * ncols * sizeof(int32_t));
}
extern "C" int sf_cmf_eigh_wide(const double *A, int p, int ncols, double *lam, double *evec, int32_t *status, void *scratch,
                                void *stream) {
  if (!A || !lam || !evec || !status || !scratch || p <= SF_MAX_ACTIVE_FUSED || p > 512 || ncols < 1) {
    sf_set_error("sf_cmf_eigh_wide: bad argument (97 .. 512 rows; sf_cmf_eigh_general with an identity target serves the smaller ones)");
    return -1;
  }
  hipStream_t st = (hipStream_t)stream;
  const int p2 = p + (p & 1);
  double *gv = reinterpret_cast<double *>(scratch);
  int32_t *flags = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(scratch) + sf_align((size_t)ncols * 2 * p2 * p2 * sizeof(double)));
  int32_t *nrows = flags + 3 * ncols;      // (the kernels' "rows behind the matrix": any count above 1 says "a real matrix")
  SF_HIP(hipMemsetAsync(nrows, 0x01, (size_t)ncols * sizeof(int32_t), st));      // 0x01010101 > 1
  return wide_eigh(A, nrows, p, p2, 0, ncols, nullptr, lam, evec, status, gv, flags, flags + ncols, flags + 2 * ncols, st, 1);
}

int sf_launch_wide_stats(const void *xt, int xt_f64, const uint8_t *mask_t, const int32_t *nuse, const int32_t *nloo,
                         const double *mu, const double *alphas, const SfGeom &g, double *cov, double *d, double *lam,
                         double *evec, int32_t *status, double *nll, int32_t *alphaidx, void *scratch, hipStream_t st,
                         const double *target) {
  if (g.p > 512 || g.nalpha > SF_NALPHA_MAX) {
    sf_set_error("wide statistics path supports up to 512 bands and %d alphas", SF_NALPHA_MAX);
    return -2;
  }
  if (!nloo) nloo = nuse;
  return wide_stats_fused(xt, xt_f64, mask_t, nuse, nloo, mu, alphas, g, cov, d, lam, evec, status, nll, alphaidx, scratch, st, target);
}
