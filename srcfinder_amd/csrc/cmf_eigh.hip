// Stage 4: batched symmetric eigendecomposition of the per-column correlation matrix.
//
// Why it exists: looshrinkage (cmf/robust_mf.py:105-117) inverts G_a = n*beta*S + a*diag(S) and takes its
// determinant for each of the 201 alphas.  With d = sqrt(diag S) and R = S/(d d^T) = V diag(lam) V^T,
//   G_a = D (n*beta*R + a*I) D,  so  x^T G_a^-1 x = sum_j y_j^2 / (n*beta*lam_j + a),  y = V^T D^-1 x,
//   log det G_a = 2 sum log d_j + sum_j log(n*beta*lam_j + a),
// i.e. ONE eigendecomposition per column replaces 201 LU factorisations + inversions.
//
// Method: one-sided (Hestenes) Jacobi on G = R with accumulated V, both resident in LDS, one workgroup per
// column, 8 lanes per column pair (dot products reduced with wave shuffles), round-robin pair ordering so
// the p/2 rotations of a step touch disjoint columns (one barrier per step).  At convergence the columns of
// G = R V are orthogonal, lam_j = |G[:,j]|, V[:,j] the eigenvector.  Latency-bound (barrier per step), ~70
// steps per sweep, 6-10 sweeps.
#include "cmf_common.h"

namespace {

constexpr int EIG_MAXSWEEP = 40;

__device__ __forceinline__ double shfl_xor_d(double v, int m) { return __shfl_xor(v, m, 64); }

// pair k of round-robin step s over p2 (even) players; m = p2 - 1
__device__ __forceinline__ void rr_pair(int s, int k, int m, int &a, int &b) {
  if (k == 0) {
    a = s % m;
    b = m;
  } else {
    a = (s + k) % m;
    b = (s - k + m) % m;
  }
}

__global__ void k_eigh(const double *__restrict__ cov, const int32_t *__restrict__ nuse, int p, int p2, int LD,
                       double *__restrict__ d_out, double *__restrict__ lam_out, double *__restrict__ evec_out,
                       int32_t *__restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double *G = sm;                 // [p2][LD] column-major: G[col*LD + row]
  double *V = sm + (size_t)p2 * LD;
  double *dv = V + (size_t)p2 * LD;  // [p2]
  __shared__ int flag[2];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int c = blockIdx.x;
  const double *S = cov + (size_t)c * p * p;
  const int n = nuse[c];

  if (tid < 2) flag[tid] = 0;
  for (int i = tid; i < p2; i += nthr) {
    double v = 0.0;
    if (i < p) v = sqrt(S[(size_t)i * p + i]);
    dv[i] = v;
  }
  __syncthreads();
  // a band with zero / non-finite variance makes every G_a exactly singular (det == 0 -> all NLL inf,
  // then inv(C) raises LinAlgError, robust_mf.py:112-113,:123-127,:371-374)
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
  // G = R, V = I
  for (int i = tid; i < p2 * p2; i += nthr) {
    const int col = i / p2, row = i - col * p2;
    double r = 0.0;
    if (col < p && row < p) r = S[(size_t)row * p + col] / (dv[row] * dv[col]);
    G[col * LD + row] = r;
    V[col * LD + row] = (col == row) ? 1.0 : 0.0;
  }
  __syncthreads();

  const int npairs = p2 >> 1, m = p2 - 1;
  const int k = tid >> 3, sub = tid & 7;
  const bool active = k < npairs;
  const double tol = 4.0 * 2.220446049250313e-16;
  for (int sweep = 0; sweep < EIG_MAXSWEEP; ++sweep) {
    bool rotated = false;
    for (int s = 0; s < m; ++s) {
      if (active) {
        int a, b;
        rr_pair(s, k, m, a, b);
        double *ga = G + a * LD, *gb = G + b * LD;
        double aa = 0, bb = 0, ab = 0;
        for (int r = sub; r < p2; r += 8) {
          const double x = ga[r], y = gb[r];
          aa += x * x; bb += y * y; ab += x * y;
        }
        aa += shfl_xor_d(aa, 1); bb += shfl_xor_d(bb, 1); ab += shfl_xor_d(ab, 1);
        aa += shfl_xor_d(aa, 2); bb += shfl_xor_d(bb, 2); ab += shfl_xor_d(ab, 2);
        aa += shfl_xor_d(aa, 4); bb += shfl_xor_d(bb, 4); ab += shfl_xor_d(ab, 4);
        const double lim = tol * sqrt(aa * bb);
        if (aa * bb > 0.0 && fabs(ab) > lim) {
          rotated = true;
          const double zeta = (bb - aa) / (2.0 * ab);
          const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
          double *va = V + a * LD, *vb = V + b * LD;
          for (int r = sub; r < p2; r += 8) {
            const double x = ga[r], y = gb[r];
            ga[r] = cs * x - sn * y;
            gb[r] = sn * x + cs * y;
            const double vx = va[r], vy = vb[r];
            va[r] = cs * vx - sn * vy;
            vb[r] = sn * vx + cs * vy;
          }
        }
      }
      __syncthreads();
    }
    if (rotated) flag[1] = 1;  // benign race: every writer stores 1
    __syncthreads();
    const int any = flag[1];
    __syncthreads();
    if (tid == 0) flag[1] = 0;
    if (!any) break;
  }
  __syncthreads();
  // eigenvalues = column norms of G; eigenvectors = columns of V
  for (int j = tid; j < p; j += nthr) {
    double s = 0;
    for (int r = 0; r < p2; ++r) {
      const double x = G[j * LD + r];
      s += x * x;
    }
    lam_out[(size_t)c * p + j] = sqrt(s);
  }
  for (int i = tid; i < p * p; i += nthr) {
    const int j = i / p, b = i - j * p;
    evec_out[(size_t)c * p * p + i] = V[j * LD + b];
  }
}

}  // namespace

int sf_launch_eigh(const double *cov, const int32_t *nuse, const SfGeom &g, double *d, double *lam, double *evec,
                   int32_t *status, hipStream_t st) {
  const int p2 = g.p + (g.p & 1);
  int LD = p2;
  while ((LD % 32) != 8 && (LD % 32) != 24) ++LD;
  const size_t lds = ((size_t)2 * p2 * LD + p2) * sizeof(double);
  if (lds > 160 * 1024 - 64) {
    sf_set_error("active window of %d bands exceeds the LDS-resident eigensolver (max %d)", g.p, SF_MAX_ACTIVE_FUSED);
    return -2;
  }
  int threads = (p2 / 2) * 8;
  threads = (threads + 63) / 64 * 64;
  if (threads < 64) threads = 64;
  if (threads > 1024) threads = 1024;  // unreachable for p <= 96
  static size_t lds_set = 0;
  if (lds > lds_set) {
    SF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_eigh), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
    lds_set = lds;
  }
  hipLaunchKernelGGL(k_eigh, dim3(g.ncols), dim3(threads), lds, st, cov, nuse, g.p, p2, LD, d, lam, evec, status);
  SF_LAUNCH_CHECK("k_eigh");
  return 0;
}
