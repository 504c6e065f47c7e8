// Generalised whitening for the full-column regulariser of the multimodal branch (-f; cmf/robust_mf.py:354 and
// looshrinkage's  T = cov(I_reg)  at :99, :131) -- gfx950 only.
//
// With a diagonal target the sweep works in the eigenbasis of R = D^-1 S D^-1 (cmf_eigh.hip).  A full target
// T = L L^T (Cholesky) gives the same structure one congruence further:
//        G_a = n b S + a T = L (n b R + a I) L^T,     R = L^-1 S L^-T = V diag(lam) V^T,
//        r_k = y^T (n b lam + a)^-1 y,  y = W^T x,    W = L^-T V,      log det G_a = log det T + sum_j log(n b lam_j + a),
//        C^-1 = W ((1-a) lam + a)^-1 W^T.
// Downstream kernels take (d, lam, evec) and use them only as  W = D^-1 evec^T  and  2 sum log d:  handing them
// d = diag(L) and evec_j = D (L^-T v_j) reproduces W and log det T exactly, so stages 5-7 run unchanged.
//
//   k_gen_whiten   one workgroup per column: Cholesky of T in LDS, two triangular solves -> R (symmetrised), L, d
//   (k_eigh, unit) eigendecomposition of R WITHOUT its own diagonal scaling
//   k_gen_back     evec_j <- D (L^-T v_j): one back-substitution per eigenvector
#include "cmf_common.h"

int sf_launch_eigh_unit(const double *cov, const int32_t *nuse, const SfGeom &g, double *d, double *lam, double *evec,
                        int32_t *status, void *scratch, hipStream_t st);

namespace {

constexpr int GW_NT = 256;

__global__ __launch_bounds__(GW_NT) void k_gen_whiten(const double *__restrict__ S, const double *__restrict__ T,
                                                       const int32_t *__restrict__ nuse, int p, int LD,
                                                       double *__restrict__ R, double *__restrict__ Lfac,
                                                       double *__restrict__ d_out) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double *A = sm;                       // [p][LD]  A[col*LD + row]: T, then its Cholesky factor (lower triangle)
  double *B = A + (size_t)p * LD;       // [p][LD]  B[row*LD + col]: S, then X = L^-1 S, then R
  double *inv = B + (size_t)p * LD;     // [p]      1 / L_ii
  __shared__ int bad;
  const int c = blockIdx.x, tid = threadIdx.x;
  const double *Sc = S + (size_t)c * p * p, *Tc = T + (size_t)c * p * p;
  if (tid == 0) bad = 0;
  for (int i = tid; i < p * p; i += GW_NT) {
    const int a = i / p, b = i - a * p;
    A[a * LD + b] = Tc[(size_t)b * p + a];
    B[a * LD + b] = Sc[(size_t)a * p + b];
  }
  __syncthreads();
  // ---- Cholesky T = L L^T in place (right-looking; the same recurrence as cmf_eigh.hip's)
  for (int kk = 0; kk < p; ++kk) {
    const double dk = A[kk * LD + kk];
    if (!(dk > 0.0) || !(dk <= 1.79769313486231570e+308)) { if (tid == 0) bad = 1; break; }   // uniform
    const double rk = 1.0 / sqrt(dk);
    __syncthreads();
    for (int i = kk + tid; i < p; i += GW_NT) A[kk * LD + i] = (i == kk) ? sqrt(dk) : A[kk * LD + i] * rk;
    __syncthreads();
    const int rem = p - kk - 1;
    for (int e = tid; e < rem * rem; e += GW_NT) {
      const int jj = e / rem, ii = e - jj * rem;
      if (ii >= jj) {
        const int j = kk + 1 + jj, i = kk + 1 + ii;
        A[j * LD + i] = __builtin_fma(-A[kk * LD + i], A[kk * LD + j], A[j * LD + i]);
      }
    }
    __syncthreads();
  }
  __syncthreads();
  if (bad || nuse[c] <= 0) {     // T not positive definite: G_a is singular for every a together with S (n_col < p + 1)
    const bool fail = nuse[c] > 0;
    for (int i = tid; i < p * p; i += GW_NT) {
      R[(size_t)c * p * p + i] = ((i / p) == (i % p)) ? 1.0 : 0.0;
      Lfac[(size_t)c * p * p + i] = ((i / p) == (i % p)) ? 1.0 : 0.0;
    }
    for (int i = tid; i < p; i += GW_NT) d_out[(size_t)c * p + i] = (fail && i == 0) ? -1.0 : 1.0;   // d[0] < 0: flag for k_gen_back
    return;
  }
  for (int i = tid; i < p; i += GW_NT) inv[i] = 1.0 / A[i * LD + i];
  __syncthreads();
  // ---- X = L^-1 S: thread t substitutes down column t of S (B[row][t]: consecutive threads, consecutive banks)
  if (tid < p) {
    for (int i = 0; i < p; ++i) {
      double acc = B[i * LD + tid];
      for (int k = 0; k < i; ++k) acc = __builtin_fma(-A[k * LD + i], B[k * LD + tid], acc);
      B[i * LD + tid] = acc * inv[i];
    }
  }
  __syncthreads();
  // ---- R^T = L^-1 X^T: thread t substitutes along row t of X (B[t][col], stride LD: odd multiple of 2 banks)
  if (tid < p) {
    for (int i = 0; i < p; ++i) {
      double acc = B[tid * LD + i];
      for (int k = 0; k < i; ++k) acc = __builtin_fma(-A[k * LD + i], B[tid * LD + k], acc);
      B[tid * LD + i] = acc * inv[i];
    }
  }
  __syncthreads();
  for (int i = tid; i < p * p; i += GW_NT) {
    const int a = i / p, b = i - a * p;
    R[(size_t)c * p * p + i] = 0.5 * (B[a * LD + b] + B[b * LD + a]);
    Lfac[(size_t)c * p * p + i] = (b <= a) ? A[b * LD + a] : 0.0;     // Lfac[row a][col b], lower triangle
  }
  for (int i = tid; i < p; i += GW_NT) d_out[(size_t)c * p + i] = A[i * LD + i];
}

// evec[c][j][:] <- D (L^-T v_j);  status 2 when the Cholesky of T failed
__global__ __launch_bounds__(GW_NT) void k_gen_back(const double *__restrict__ Lfac, int p, int LD, double *__restrict__ d,
                                                     double *__restrict__ evec, double *__restrict__ lam,
                                                     int32_t *__restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double *A = sm;                       // [p][LD]  A[row*LD + col] = L[row][col]
  double *V = A + (size_t)p * LD;       // [p][LD]  V[b*LD + j] = component b of eigenvector j
  const int c = blockIdx.x, tid = threadIdx.x;
  const bool fail = d[(size_t)c * p] < 0.0;
  __syncthreads();
  if (fail) {
    if (tid == 0) { if (status[c] == 0) status[c] = 2; d[(size_t)c * p] = 1.0; }
    for (int i = tid; i < p; i += GW_NT) lam[(size_t)c * p + i] = 0.0;
    return;
  }
  if (status[c] != 0) return;
  double *ev = evec + (size_t)c * p * p;
  for (int i = tid; i < p * p; i += GW_NT) {
    const int a = i / p, b = i - a * p;
    A[a * LD + b] = Lfac[(size_t)c * p * p + i];
    V[b * LD + a] = ev[i];              // ev[j = a][b]
  }
  __syncthreads();
  if (tid < p) {
    for (int i = p - 1; i >= 0; --i) {   // L^T w = v:  w_i = (v_i - sum_{k > i} L_ki w_k) / L_ii
      double acc = V[i * LD + tid];
      for (int k = i + 1; k < p; ++k) acc = __builtin_fma(-A[k * LD + i], V[k * LD + tid], acc);
      V[i * LD + tid] = acc / A[i * LD + i];
    }
  }
  __syncthreads();
  for (int i = tid; i < p * p; i += GW_NT) {
    const int a = i / p, b = i - a * p;
    ev[i] = V[b * LD + a] * A[b * LD + b];
  }
}

}  // namespace

extern "C" int sf_cmf_eigh_general(const double *cov, const double *target, const int32_t *nuse, int p, int ncols,
                                   double *r_tmp, double *l_tmp, double *d, double *lam, double *evec, int32_t *status,
                                   void *scratch, void *stream) {
  if (!cov || !target || !nuse || !r_tmp || !l_tmp || !d || !lam || !evec || !status || !scratch) {
    sf_set_error("null pointer");
    return -1;
  }
  if (p < 1 || p > SF_MAX_ACTIVE_FUSED || ncols < 1) { sf_set_error("sf_cmf_eigh_general: bad geometry"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  const int LD = p | 1;
  const size_t lds = ((size_t)2 * p * LD + p) * sizeof(double);
  if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_gen_whiten), lds)) return rc;
  if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_gen_back), lds)) return rc;
  hipLaunchKernelGGL(k_gen_whiten, dim3(ncols), dim3(GW_NT), lds, st, cov, target, nuse, p, LD, r_tmp, l_tmp, d);
  SF_LAUNCH_CHECK("k_gen_whiten");
  if (int rc = sf_launch_eigh_unit(r_tmp, nuse, sf_geom(1, p, ncols, 1), nullptr, lam, evec, status, scratch, st)) return rc;
  hipLaunchKernelGGL(k_gen_back, dim3(ncols), dim3(GW_NT), lds, st, l_tmp, p, LD, d, evec, lam, status);
  SF_LAUNCH_CHECK("k_gen_back");
  return 0;
}
