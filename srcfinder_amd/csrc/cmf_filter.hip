// Stage 6: shrunk covariance, solve and normalisation -> the per-column matched filter vector.
//
// Replaces cmf/robust_mf.py:124-134 (alpha from the argmin, C = (1-a) S + a diag S on UNSCALED data),
// :363 (inv(C)) and :378-381 (target, normaliser).  With S = D R D and R = V diag(lam) V^T:
//   C^-1 t = D^-1 V diag(1/((1-a) lam_j + a)) V^T D^-1 t,
// so the explicit inverse becomes two p x p mat-vecs per column.  The per-pixel score of stage 7 is then
//   mf * scale = (x - mu) . filt = x . filt - bias,   filt = scale * C^-1 t / (t^T C^-1 t).
// Tiny kernel (one workgroup per column); latency-bound.
#include "cmf_common.h"

namespace {

__global__ __launch_bounds__(128) void k_filter(const double *__restrict__ mu, const double *__restrict__ d,
                                                 const double *__restrict__ lam, const double *__restrict__ evec,
                                                 const double *__restrict__ alphas, int32_t *__restrict__ alphaidx,
                                                 const double *__restrict__ abscf, int reflectance, int p,
                                                 int32_t *__restrict__ status, double *__restrict__ filt,
                                                 double *__restrict__ bias) {
  extern __shared__ double sm[];
  double *tt = sm;       // [p]  D^-1 t
  double *eu = tt + p;   // [p]  e_j * u_j
  double *wv = eu + p;   // [p]  filter before the final scaling
  __shared__ double s_norm;
  __shared__ int s_bad;
  const int c = blockIdx.x, tid = threadIdx.x;
  const double *muc = mu + (size_t)c * p, *dc = d + (size_t)c * p, *lc = lam + (size_t)c * p;
  const double *ev = evec + (size_t)c * p * p;
  double *fo = filt + (size_t)c * p;
  if (status[c] == 3) {
    // exactly one valid row (robust_mf.py:52-70, :92-127): cov() divides by n - 1 = 0 -> a NaN covariance, every NLL is
    // NaN, numpy.argmin of an all-NaN vector is 0 and NaN != inf, so alpha = alphas[0]; C is NaN, scipy's inv
    // (check_finite=False) does not raise, and the row's score is NaN.  Reproduced with a NaN filter on a status-0 column.
    const double qn = __builtin_nan("");
    for (int b = tid; b < p; b += 128) fo[b] = qn;
    if (tid == 0) { bias[c] = qn; alphaidx[c] = 0; status[c] = 0; }
    return;
  }
  if (status[c] != 0) {  // 1: no valid rows; 2: singular -> the reference writes 0 for the valid rows
    for (int b = tid; b < p; b += 128) fo[b] = 0.0;
    if (tid == 0) bias[c] = 0.0;
    return;
  }
  if (tid == 0) s_bad = 0;
  const int ai = alphaidx[c];
  const double alpha = (ai >= 0) ? alphas[ai] : 0.0;  // robust_mf.py:123-127
  for (int b = tid; b < p; b += 128) {
    const double t = reflectance ? (abscf[b] - muc[b]) : (abscf[b] * muc[b]);  // :378-379
    tt[b] = t / dc[b];
  }
  __syncthreads();
  for (int j = tid; j < p; j += 128) {
    double u = 0.0;
    for (int b = 0; b < p; ++b) u += ev[(size_t)j * p + b] * tt[b];
    const double den = (1.0 - alpha) * lc[j] + alpha;
    if (!(den > 0.0)) s_bad = 1;
    const double e = 1.0 / den;
    eu[j] = e * u;
    wv[j] = e * u * u;  // temporarily: terms of the normaliser
  }
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int j = 0; j < p; ++j) s += wv[j];
    s_norm = s;  // t^T C^-1 t
    if (!s_bad && s == 0.0) s_bad = 2;
    else if (!(fabs(s) > 0.0) || !(fabs(s) <= 1.79769313486231570e+308)) s_bad = 1;
  }
  __syncthreads();
  if (s_bad == 2) {
    // t^T C^-1 t == 0 with a positive definite C: the target is the zero vector (a library without absorption in the
    // window, a zero column mean).  The reference divides by it (robust_mf.py:380-381): 0 / 0, every valid row of the
    // column scores NaN -- a NaN filter on a status-0 column, as for status 3 above
    const double qn = __builtin_nan("");
    for (int b = tid; b < p; b += 128) fo[b] = qn;
    if (tid == 0) bias[c] = qn;
    return;
  }
  if (s_bad) {
    for (int b = tid; b < p; b += 128) fo[b] = 0.0;
    if (tid == 0) { bias[c] = 0.0; status[c] = 2; }
    return;
  }
  const double scale = (reflectance ? 1.0 : 100000.0) / s_norm;  // ppmscaling, :38, :383-386
  for (int b = tid; b < p; b += 128) {
    double v = 0.0;
    for (int j = 0; j < p; ++j) v += ev[(size_t)j * p + b] * eu[j];
    const double w = v / dc[b] * scale;
    wv[b] = w;  // distinct slot per thread; the normaliser terms in wv are dead after the barrier above
    fo[b] = w;
  }
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int b = 0; b < p; ++b) s += muc[b] * wv[b];
    bias[c] = s;
  }
}

}  // namespace

int sf_launch_filter(const double *mu, const double *d, const double *lam, const double *evec, const double *alphas,
                     int32_t *alphaidx, const double *abscf, int reflectance, const SfGeom &g, int32_t *status,
                     double *filt, double *bias, hipStream_t st) {
  const size_t lds = (size_t)3 * g.p * sizeof(double);
  hipLaunchKernelGGL(k_filter, dim3(g.ncols), dim3(128), lds, st, mu, d, lam, evec, alphas, alphaidx, abscf, reflectance,
                     g.p, status, filt, bias);
  SF_LAUNCH_CHECK("k_filter");
  return 0;
}
