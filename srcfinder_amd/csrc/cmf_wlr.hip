// Rank factorisation of the wide window's sweep coefficients (round 5; used by k_wsweep8<.., FACT> in cmf_wgemm.hip).
//
// The second product of the sweep, r_k(alpha_a) = sum_j z_kj / (n beta_a lam_j + alpha_a) (cmf/robust_mf.py:105-117 restated,
// DESIGN.md section 4), multiplies the tile's squares by a p x 201 matrix whose numerical rank is far below p: the rows are
// samples of ONE smooth family f_lam(alpha), and a flightline column's spectrum is a noise-floor cluster of ~p - 5 nearly
// equal eigenvalues plus a few signal directions -- rank ~25 at 1e-15, whatever the range.  As on the narrow windows
// (cmf_lowrank.hip) what is factored is the row- and column-scaled matrix
//        B'_ja = lam_j beta_a / (n beta_a lam_j + alpha_a) = sum_m G_jm W_ma + E,       W: K <= 31 rows, orthonormal to rounding,
// and the sweep multiplies by Uc = diag(1 / lam) G (p x K) and then by W (K x 208): beta_a r_k(a).
//
// k_lowrank factors 72 spectra per workgroup in registers and has no room for 425, so the basis W comes from <= 72 PROXY
// eigenvalues of the column -- the sorted spectrum thinned greedily so that no eigenvalue is further than delta (in log lam)
// from a proxy, delta found by a 64-way search for at most 72 proxies: isolated eigenvalues are proxies themselves, a
// cluster is covered densely -- and is then checked and completed against the column's REAL eigenvalues (k_wlr_build): the
// coefficients G = B' W^T and the residual of every row are computed from the formula; while the largest residual exceeds the
// narrow windows' bar (1e-14 of the largest row norm) the normalised residual row -- orthogonalised twice against the rows already
// there -- joins the basis.  On flightline-like spectra the proxies leave 2e-14 .. 8e-14 (the interpolation error between them) and
// ONE such row takes it to 1.5e-15 (numpy model of this file).  What the sweep uses is exactly what was checked.  A column that
// does not reach the bar with 31 rows (a spectrum spread densely over decades), or whose proxies k_lowrank refuses (condition
// > 1e10, rank > 28), keeps wlr = 0 and is swept by the unfactored kernel.
#include "cmf_common.h"

namespace {

constexpr int WL_M = 72;       // proxy eigenvalues = the columns k_lowrank<18> factors
constexpr int WL_K = 32;       // factor slots of the sweep: 31 basis rows + the all-ones column
constexpr int WL_NA = 208;     // alpha slots (13 tiles of 16)
constexpr int WL_PMAX = 512;

__global__ __launch_bounds__(256) void k_wlr_proxy(const double *__restrict__ lam, const int32_t *__restrict__ status, int p,
                                                   double *__restrict__ lamp) {
  __shared__ double sv[WL_PMAX];
  __shared__ double lg[WL_PMAX];
  __shared__ int sel[WL_M];
  __shared__ int nsel;
  const int c = blockIdx.x, tid = threadIdx.x;
  double *out = lamp + (size_t)c * WL_M;
  if (status[c] != 0) {
    if (tid < WL_M) out[tid] = 1.0;
    return;
  }
  for (int i = tid; i < WL_PMAX; i += 256) sv[i] = (i < p) ? lam[(size_t)c * p + i] : 1.7976931348623157e308;
  __syncthreads();
  for (int k = 2; k <= WL_PMAX; k <<= 1)          // bitonic sort, ascending
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < WL_PMAX; i += 256) {
        const int l = i ^ j;
        if (l > i) {
          const double a = sv[i], b = sv[l];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { sv[i] = b; sv[l] = a; }
        }
      }
      __syncthreads();
    }
  const bool pos = sv[0] > 0.0 && sv[p - 1] < 1.7976931348623157e308;   // (NaN sorts nowhere in particular: caught here or by k_lowrank)
  if (!pos) {   // not positive definite: k_lowrank's own test refuses the column (lam_min <= 1e-10 lam_max)
    if (tid < WL_M) out[tid] = (tid == 0) ? sv[0] : 1.0;
    return;
  }
  for (int i = tid; i < p; i += 256) lg[i] = log(sv[i]);
  __syncthreads();
  if (tid < 64) {
    // greedy cover with spacing delta: index 0, then every eigenvalue more than delta above the last one taken, then p - 1
    auto count = [&](double delta) {
      int cnt = 1;
      double last = lg[0];
      for (int i = 1; i < p; ++i)
        if (lg[i] - last > delta) { ++cnt; last = lg[i]; }
      return cnt + ((lg[p - 1] - last > 0.0) ? 1 : 0);
    };
    double lo = 0.0, hi = (lg[p - 1] - lg[0]) + 1.0;   // hi: two proxies (both ends)
    for (int round = 0; round < 3; ++round) {
      const double d = lo + (hi - lo) * (double)(tid + 1) / 64.0;
      const bool ok = count(d) <= WL_M;
      const unsigned long long okm = __ballot(ok);     // (lane 63 tests hi itself: always set)
      const int first = __ffsll((long long)okm) - 1;
      const double nhi = lo + (hi - lo) * (double)(first + 1) / 64.0;
      const double nlo = lo + (hi - lo) * (double)first / 64.0;
      hi = nhi;
      lo = nlo;
    }
    if (tid == 0) {
      int cnt = 1;
      double last = lg[0];
      sel[0] = 0;
      for (int i = 1; i < p; ++i)
        if (lg[i] - last > hi && cnt < WL_M) { sel[cnt++] = i; last = lg[i]; }
      if (lg[p - 1] - last > 0.0) {
        if (cnt < WL_M) sel[cnt++] = p - 1; else sel[WL_M - 1] = p - 1;
      }
      nsel = cnt;
    }
  }
  __syncthreads();
  if (tid < WL_M) out[tid] = sv[sel[tid < nsel ? tid : nsel - 1]];   // (unused slots repeat the largest eigenvalue: a zero column after its pivot)
}

// One workgroup per column, a thread per eigenvalue.  W: the basis, rows K .. 30 appended here.
__global__ __launch_bounds__(512) void k_wlr_build(const double *__restrict__ lam, const int32_t *__restrict__ nloo,
                                                   const int32_t *__restrict__ status, const double *__restrict__ alphas, int nalpha,
                                                   int p, const double *__restrict__ wfrag, const int32_t *__restrict__ lrok,
                                                   int njw4, int njl4, double *__restrict__ U8, double *__restrict__ T8,
                                                   int32_t *__restrict__ wlr) {
  __shared__ double W[WL_K - 1][WL_NA];
  __shared__ double s_al[WL_NA], s_be[WL_NA];
  __shared__ double gsel[WL_K];
  __shared__ double wred[8];
  __shared__ int ired[8];
  __shared__ double s_r00, s_lsel, s_norm;
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int code = lrok[c];
  int K = (code == 3) ? SF_LR_K0 : ((code == 1) ? SF_LR_K : 0);
  constexpr int USZ = 4 * 8 * 56 * 8, TSZ = 8 * 8 * 7 * 16;   // doubles per column: Uc image (4 chunks x 8 waves x 56 x 8), T image
  double *uo = U8 + (size_t)c * USZ, *to = T8 + (size_t)c * TSZ;
  if (status[c] != 0 || K == 0) {
    if (tid == 0) wlr[c] = 0;
    return;
  }
  const double n = (double)nloo[c];
  for (int i = tid; i < WL_NA; i += 512) {
    const double a = (i < nalpha) ? alphas[i] : 1.0;
    s_al[i] = a;
    s_be[i] = (i < nalpha) ? (1.0 - a) / (n - 1.0) : 0.0;
  }
  {   // W[m][a] = wfrag[((a >> 4) * 9 + (m >> 2)) * 64 + 16 (m & 3) + (a & 15)]   (cmf_lowrank.hip)
    const double *wf = wfrag + (size_t)c * (13 * 9 * 64);
    for (int i = tid; i < (WL_K - 1) * WL_NA; i += 512) {
      const int m = i / WL_NA, a = i - m * WL_NA;
      W[m][a] = (m < K) ? wf[((a >> 4) * 9 + (m >> 2)) * 64 + 16 * (m & 3) + (a & 15)] : 0.0;
    }
  }
  for (int i = tid; i < USZ; i += 512) uo[i] = 0.0;
  __syncthreads();
  const int j = tid;
  const bool have = j < p;
  const double lj = have ? lam[(size_t)c * p + j] : 1.0;
  auto bval = [&](int a) { const double be = s_be[a]; return (have && be > 0.0) ? lj * be / ((n * be) * lj + s_al[a]) : 0.0; };
  double g[WL_K - 1];
#pragma unroll
  for (int m = 0; m < WL_K - 1; ++m) g[m] = 0.0;
  double b2 = 0.0;
  for (int a = 0; a < WL_NA; ++a) {
    const double b = bval(a);
    b2 = __builtin_fma(b, b, b2);
#pragma unroll
    for (int m = 0; m < WL_K - 1; ++m) g[m] = __builtin_fma(b, W[m][a], g[m]);
  }
  auto block_max = [&](double v, int idx, double &vmax, int &imax) {   // largest value, lowest index on ties
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double ov = __hiloint2double(__shfl_xor(__double2hiint(v), off, 64), __shfl_xor(__double2loint(v), off, 64));
      const int oi = __shfl_xor(idx, off, 64);
      const bool take = (ov > v) || (ov == v && oi < idx);
      v = take ? ov : v;
      idx = take ? oi : idx;
    }
    if (lane == 0) { wred[wave] = v; ired[wave] = idx; }
    __syncthreads();
    vmax = wred[0];
    imax = ired[0];
    for (int w = 1; w < 8; ++w)
      if (wred[w] > vmax || (wred[w] == vmax && ired[w] < imax)) { vmax = wred[w]; imax = ired[w]; }
    __syncthreads();
  };
  {
    double vmax;
    int imax;
    block_max(have ? b2 : -1.0, j, vmax, imax);
    if (tid == 0) s_r00 = sqrt(vmax);
  }
  bool pass = false;
  for (;;) {
    double e2 = 0.0;
    for (int a = 0; a < WL_NA; ++a) {
      double e = bval(a);
#pragma unroll
      for (int m = 0; m < WL_K - 1; ++m) e = __builtin_fma(-g[m], W[m][a], e);
      e2 = __builtin_fma(e, e, e2);
    }
    double vmax;
    int imax;
    block_max(have ? e2 : -1.0, j, vmax, imax);
    const double r00 = s_r00;
    if (!(vmax >= 0.0) || !(r00 > 0.0)) break;                 // NaN somewhere: not factored
    if (sqrt(vmax) <= 1e-14 * r00) { pass = true; break; }
    if (K == WL_K - 1) break;
    // the residual row of eigenvalue imax, normalised, becomes basis row K
    if (j == imax) {
#pragma unroll
      for (int m = 0; m < WL_K - 1; ++m) gsel[m] = g[m];
      s_lsel = lj;
    }
    __syncthreads();
    double ea = 0.0;
    if (tid < WL_NA) {
      const double be = s_be[tid], ls = s_lsel;
      ea = (be > 0.0) ? ls * be / ((n * be) * ls + s_al[tid]) : 0.0;
      for (int m = 0; m < WL_K - 1; ++m) ea = __builtin_fma(-gsel[m], W[m][tid], ea);
    }
    double sq = (tid < WL_NA) ? ea * ea : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __hiloint2double(__shfl_xor(__double2hiint(sq), off, 64), __shfl_xor(__double2loint(sq), off, 64));
    if (lane == 0) wred[wave] = sq;
    __syncthreads();
    if (tid == 0) s_norm = sqrt(((wred[0] + wred[1]) + (wred[2] + wred[3])) + ((wred[4] + wred[5]) + (wred[6] + wred[7])));
    __syncthreads();
    if (tid < WL_NA) W[K][tid] = ea / s_norm;
    __syncthreads();
    // the residual is a difference of O(1) numbers at the 1e-14 level: its direction carries ~1 % of rounding noise, i.e. overlaps
    // of that size with the rows already there -- and G = B' W^T is only the right set of coefficients for ORTHONORMAL rows.
    // Orthogonalised against them again, twice ("twice is enough"), the new row is orthonormal to rounding.
    for (int rep = 0; rep < 2; ++rep) {
      if (tid < WL_K - 1) {
        double dsum = 0.0;
        if (tid < K)
          for (int a = 0; a < WL_NA; ++a) dsum = __builtin_fma(W[tid][a], W[K][a], dsum);
        gsel[tid] = dsum;
      }
      __syncthreads();
      double wa = 0.0;
      if (tid < WL_NA) {
        wa = W[K][tid];
        for (int m = 0; m < WL_K - 1; ++m) wa = __builtin_fma(-gsel[m], W[m][tid], wa);
      }
      double s2 = (tid < WL_NA) ? wa * wa : 0.0;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s2 += __hiloint2double(__shfl_xor(__double2hiint(s2), off, 64), __shfl_xor(__double2loint(s2), off, 64));
      if (lane == 0) wred[wave] = s2;
      __syncthreads();
      if (tid == 0) s_norm = sqrt(((wred[0] + wred[1]) + (wred[2] + wred[3])) + ((wred[4] + wred[5]) + (wred[6] + wred[7])));
      __syncthreads();
      if (tid < WL_NA) W[K][tid] = wa / s_norm;
      __syncthreads();
    }
    double gn = 0.0;
    for (int a = 0; a < WL_NA; ++a) gn = __builtin_fma(bval(a), W[K][a], gn);
#pragma unroll
    for (int m = 0; m < WL_K - 1; ++m) g[m] = (m == K) ? gn : g[m];
    ++K;
  }
  if (tid == 0) wlr[c] = pass ? K : 0;
  if (!pass) return;
  // ---- the sweep's operand images (layouts of k_cmat_t8 / a W slice, cmf_wgemm.hip):
  //      U8 [4 chunks][8 waves][56][8]: eigenvalue j of wave w's slice, factor 8 ch + a8;   T8 [8 waves][8 kk][7 ag][16]: T[4 kk + q][28 w + 4 ag + n]
  if (have) {
    const int w = (j < 4 * njw4) ? j / njw4 : 4 + (j - 4 * njw4) / njl4;
    const int jl = (j < 4 * njw4) ? j - w * njw4 : (j - 4 * njw4) - (w - 4) * njl4;
#pragma unroll
    for (int m = 0; m < WL_K; ++m) {
      const double v = (m == WL_K - 1) ? 1.0 : g[m < WL_K - 1 ? m : 0] / lj;
      uo[(((m >> 3) * 8 + w) * 56 + jl) * 8 + (m & 7)] = v;
    }
  }
  for (int i = tid; i < TSZ; i += 512) {
    const int w = i / 896, r = i - w * 896, kk = r / 112, r2 = r - kk * 112, ag = r2 >> 4, q = (r2 >> 2) & 3, nn = r2 & 3;
    const int m = 4 * kk + q, a = 28 * w + 4 * ag + nn;
    double v = 0.0;
    if (a < nalpha) {
      const bool bz = !(s_be[a] > 0.0);
      v = (m == WL_K - 1) ? (bz ? 1.0 : 0.0) : (bz ? 0.0 : W[m][a]);
    }
    to[i] = v;
  }
}

}  // namespace

// scratch of one launch over nb columns: proxies, k_lowrank's fragments and verdicts
size_t sf_wlr_scratch_bytes(int nb) {
  const SfGeom g72 = sf_geom(64, WL_M, nb, 201);
  return sf_align((size_t)nb * WL_M * sizeof(double)) + sf_lowrank_bytes(g72);
}
size_t sf_wlr_image_bytes(int nb) {   // U8 (+ 1 KB: the sweep's last 16-byte pieces run past a slice), T8, wlr
  return sf_align((size_t)nb * 4 * 8 * 56 * 8 * sizeof(double) + 1024) + sf_align((size_t)nb * 8 * 896 * sizeof(double)) +
         sf_align((size_t)nb * sizeof(int32_t));
}

int sf_launch_wlr(const double *lam, const int32_t *nloo, const int32_t *status, const double *alphas, int nalpha, int p, int nb,
                  int njw, int njl, void *scratch, void *images, const double **U8, const double **T8, const int32_t **wlr,
                  hipStream_t st) {
  if (p > WL_PMAX || nalpha > WL_NA || 16 * njw + 16 * njl < p) { sf_set_error("sf_launch_wlr: geometry"); return -2; }
  const SfGeom g72 = sf_geom(64, WL_M, nb, nalpha);
  char *sb = reinterpret_cast<char *>(scratch);
  double *lamp = reinterpret_cast<double *>(sb);
  char *lr = sb + sf_align((size_t)nb * WL_M * sizeof(double));
  double *ufrag = reinterpret_cast<double *>(lr);
  double *wfrag = reinterpret_cast<double *>(lr + sf_align((size_t)nb * 18 * (SF_LR_K2 / 4) * 16 * sizeof(double)));
  int32_t *lrok = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(wfrag) + sf_align((size_t)nb * 13 * (SF_LR_K2 / 4) * 64 * sizeof(double)));
  char *ib = reinterpret_cast<char *>(images);
  double *u8 = reinterpret_cast<double *>(ib);
  double *t8 = reinterpret_cast<double *>(ib + sf_align((size_t)nb * 4 * 8 * 56 * 8 * sizeof(double) + 1024));
  int32_t *wl = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(t8) + sf_align((size_t)nb * 8 * 896 * sizeof(double)));
  hipLaunchKernelGGL(k_wlr_proxy, dim3(nb), dim3(256), 0, st, lam, status, p, lamp);
  SF_LAUNCH_CHECK("k_wlr_proxy");
  if (int rc = sf_launch_lowrank(lamp, nloo, status, alphas, g72, ufrag, wfrag, lrok, st, 1)) return rc;
  hipLaunchKernelGGL(k_wlr_build, dim3(nb), dim3(512), 0, st, lam, nloo, status, alphas, nalpha, p, wfrag, lrok, 4 * njw, 4 * njl, u8, t8, wl);
  SF_LAUNCH_CHECK("k_wlr_build");
  *U8 = u8;
  *T8 = t8;
  *wlr = wl;
  return 0;
}

// test hook (srcfinder_amd.cmf.sweep_routes, tests/test_cmf_gpu.py): the verdict per column -- 0 = unfactored, else the rank
extern "C" size_t sf_debug_wlr_bytes(int ncols) { return sf_wlr_scratch_bytes(ncols) + sf_wlr_image_bytes(ncols); }
extern "C" int sf_debug_wlr(const double *lam, const int32_t *nloo, const int32_t *status, const double *alphas, int nalpha, int p,
                            int ncols, void *scratch, int32_t *wlr_out, void *stream) {
  if (!lam || !nloo || !status || !alphas || !scratch || !wlr_out || ncols < 1) { sf_set_error("sf_debug_wlr: bad argument"); return -1; }
  if (p <= 256 || p > 432) { sf_set_error("sf_debug_wlr: windows of 257..432 bands"); return -2; }
  const double *u8, *t8;
  const int32_t *wl;
  char *sb = reinterpret_cast<char *>(scratch);
  if (int rc = sf_launch_wlr(lam, nloo, status, alphas, nalpha, p, ncols, 14, 13, sb, sb + sf_wlr_scratch_bytes(ncols), &u8, &t8, &wl,
                             (hipStream_t)stream))
    return rc;
  SF_HIP(hipMemcpyAsync(wlr_out, wl, (size_t)ncols * sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}
