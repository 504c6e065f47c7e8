// Stage 5: Theiler leave-one-out negative log-likelihood for every alpha of the grid, and its argmin.
//
// Replaces the alpha loop of looshrinkage (cmf/robust_mf.py:105-127).  Per valid row k and alpha i:
//   r_ki = x_k^T G_i^-1 x_k = sum_j y_kj^2 * c_ij,   y_k = V^T D^-1 (x_k - mu),   c_ij = 1/(n*beta_i*lam_j + a_i)
//   q_ki = 1 - beta_i r_ki,    nll_i = 0.5 (p log 2pi + log det G_i) + (1/2n) sum_k (log q_ki + r_ki/q_ki)
// (stability scaling 100 of :94 cancels in y and enters log det as 2 p log 100).
//
// k_sweep: one 256-thread workgroup = (column, row split).  A wave owns 16 rows at a time and chains two
// fp64 MFMA products without leaving registers:
//   GEMM1  Y^T(16j x 16k) += W^T(j,b) . X^T(b,k)      A = W = D^-1 V (fragment-ordered, global/L2), B = rows
//   Z = Y^T .^ 2  (the accumulator registers ARE the next A operand: lane&15 = row k, lane>>4 = j slot)
//   GEMM2  r(16k x 16i) += Z(k,j) . C(j,i)            B = c_ij fragments resident in LDS (117 KB at p=72)
// and reduces over rows in registers: per (lane, alpha tile) a running product of q (exponent split off
// with frexp so it cannot underflow) and a running sum of r/q formed with ONE division per 4 rows.
// Zeroed (invalid) rows give r = 0, q = 1 and contribute exactly nothing.  MFMA-bound:
// (NT*S4 + NU*S4) MFMAs per 16 rows = 324 at p = 72, A = 201.
#include "cmf_common.h"

namespace {

constexpr int SW_NUMAX = SF_NALPHA_MAX / 16;  // 13 alpha tiles

__device__ __forceinline__ double shfl_xor_d(double v, int m) { return __shfl_xor(v, m, 64); }

// W fragments: wfrag[c][(t*S4 + s)*64 + lane] = V[b][j] / d[b],  j = 16t + (lane&15),  b = S4*(lane>>4) + s
__global__ void k_wfrag(const double *__restrict__ evec, const double *__restrict__ d, int p, int S4, int nt,
                        size_t stride, double *__restrict__ wfrag) {
  const int c = blockIdx.x;
  const double *ev = evec + (size_t)c * p * p;
  const double *dd = d + (size_t)c * p;
  double *w = wfrag + (size_t)c * stride;
  const int total = nt * S4 * 64;
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const int lane = i & 63, ts = i >> 6;
    const int t = ts / S4, s = ts - t * S4;
    const int j = 16 * t + (lane & 15), b = S4 * (lane >> 4) + s;
    double v = 0.0;
    if (j < p && b < p) v = ev[(size_t)j * p + b] / dd[b];
    w[i] = v;
  }
}

// NT  : 16-wide tiles over the band / eigen axis (p <= 16 NT)
// S4C : compile-time number of 4-deep k-steps ceil(p/4) (0 = runtime; the production windows p = 72 and
//       p = 83 get their own instantiation so every LDS offset is an immediate and dead steps vanish)
// NUC : compile-time number of alpha tiles (0 = runtime)
// WREG: hold the GEMM1 A operand (W fragments, NT*S4 doubles per lane) in registers for the whole kernel --
//       one wave per SIMD (256-thread workgroup, 512-register budget); otherwise stream them from L2.
template <int NT, int S4C, int NUC, bool WREG, typename XT>
__global__ __launch_bounds__(256, 1) void k_sweep(const XT *__restrict__ xt, const uint8_t *__restrict__ mask_t,
                                                   const int32_t *__restrict__ nuse, const double *__restrict__ mu,
                                                   const double *__restrict__ lam, const double *__restrict__ wfrag,
                                                   size_t wstride, const int32_t *__restrict__ status,
                                                   const double *__restrict__ alphas, int nalpha, int L, int p,
                                                   int PS, int rows_per_wg, double *__restrict__ part,
                                                   const int32_t *__restrict__ taken) {
  constexpr int NS = S4C ? S4C : 4 * NT;
  constexpr int NUM = NUC ? NUC : SW_NUMAX;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int S4 = S4C ? S4C : ((p + 3) >> 2);
  const int NU = NUC ? NUC : ((nalpha + 15) >> 4);
  double *cfrag = sm;                            // [NU][S4][64]
  double *mus = cfrag + (size_t)NU * S4 * 64;    // [PS]
  double *betas = mus + PS;                      // [NU*16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int c = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
  const int NA16 = NU * 16;
  double *po = part + ((size_t)c * nsplit + split) * 2 * NA16;

  if (taken && taken[c] != 0) return;   // a rank-factored 4x4x4 sweep serves this column (cmf_loocv4.hip)
  if (status[c] != 0) {  // no valid rows / singular: every NLL is +inf, nothing to accumulate
    for (int i = tid; i < 2 * NA16; i += 256) po[i] = 0.0;
    return;
  }
  const double n = (double)nuse[c];
  // ---- prologue: c_ij fragments, column mean, beta_i
  for (int i = tid; i < NA16; i += 256) {
    const double a = (i < nalpha) ? alphas[i] : 1.0;
    betas[i] = (i < nalpha) ? (1.0 - a) / (n - 1.0) : 0.0;
  }
  for (int i = tid; i < PS; i += 256) mus[i] = (i < p) ? mu[(size_t)c * p + i] : 0.0;
  {
    const double *lc = lam + (size_t)c * p;
    const int total = NU * S4 * 64;
    for (int idx = tid; idx < total; idx += 256) {
      const int ln = idx & 63, us = idx >> 6;
      const int u = us / S4, s2 = us - u * S4;
      const int i = 16 * u + (ln & 15), j = 4 * s2 + (ln >> 4);
      double v = 0.0;
      if (i < nalpha && j < p) {
        const double a = alphas[i];
        const double beta = (1.0 - a) / (n - 1.0);
        v = 1.0 / ((n * beta) * lc[j] + a);
      }
      cfrag[idx] = v;
    }
  }
  const double *wf = wfrag + (size_t)c * wstride + lane;
  double wreg[WREG ? NT : 1][WREG ? NS : 1];
  if (WREG) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < NS; ++s) wreg[t][s] = (s < S4) ? wf[((size_t)t * S4 + s) * 64] : 0.0;
  }
  __syncthreads();

  double P[NUM], R[NUM];
  int E[NUM];
#pragma unroll
  for (int u = 0; u < NUM; ++u) { P[u] = 1.0; R[u] = 0.0; E[u] = 0; }

  const int rbeg = split * rows_per_wg, rend = min(L, rbeg + rows_per_wg);
  const uint8_t *mp = mask_t + (size_t)c * L;
  const XT *xc = xt + (size_t)c * L * PS + S4 * g;
  const double *cf = cfrag + lane;
  const double qnan = __builtin_nan("");
  const bool pairload = (S4 & 1) == 0;

  // raw (float) B operand of the NEXT row tile is fetched one tile ahead
  XT xraw[NS + 1];
  bool rowok_next;
  auto fetch = [&](int r0, XT (&dst)[NS + 1], bool &ok) {
    const int row = r0 + li;
    ok = (row < rend) && (mp[row < rend ? row : rbeg] != 0);
    const XT *xp = xc + (size_t)(ok ? row : rbeg) * PS;
    if (pairload) {
#pragma unroll
      for (int s = 0; s < NS; s += 2) {
        dst[s] = (XT)0;
        dst[s + 1] = (XT)0;
        if (s < S4) sf_load2(xp + s, dst[s], dst[s + 1]);
      }
    } else {
#pragma unroll
      for (int s = 0; s < NS; ++s) dst[s] = (s < S4) ? xp[s] : (XT)0;
    }
  };
  int r0 = rbeg + 16 * wave;
  if (r0 < rend) fetch(r0, xraw, rowok_next);

  for (; r0 < rend; r0 += 16 * 4) {
    const bool rowok = rowok_next;
    // The LDS / L2 operand tables are loop-invariant; without this the compiler hoists all ~270 operand
    // loads out of the row loop and spills.  An opaque zero keeps them inside the iteration.
    int opq = 0;
    asm volatile("" : "+v"(opq));
    const double *cfl = cf + opq;
    const double *wfl = wf + opq;
    const double *musl = mus + opq;
    const double *betl = betas + opq;
    // ---- B operand of GEMM1: this lane's S4 consecutive bands of its row, centred, as float64
    double x[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int b = S4 * g + s;
      x[s] = (rowok && s < S4 && b < p) ? (double)xraw[s] - musl[b] : 0.0;
    }
    if (r0 + 64 < rend) fetch(r0 + 64, xraw, rowok_next);
    // ---- GEMM1 (NT independent accumulator chains) + square:
    //      z[t][reg] = y^2 for eigen index j = 16t + 4reg + g, row = li
    d4_t z[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) z[t] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (s < S4) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const double a = WREG ? wreg[WREG ? t : 0][WREG ? s : 0] : wfl[((size_t)t * S4 + s) * 64];
          z[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x[s], z[t], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) z[t] = z[t] * z[t];
    // ---- GEMM2 per pair of alpha tiles (two independent chains) + row reduction
#pragma unroll
    for (int u0 = 0; u0 < NUM; u0 += 2) {
      if (u0 < NU) {
        const bool two = (u0 + 1 < NUM) && (u0 + 1 < NU);
        d4_t acc0 = d4_t{0.0, 0.0, 0.0, 0.0}, acc1 = d4_t{0.0, 0.0, 0.0, 0.0};
        const double *cu0 = cfl + (size_t)u0 * S4 * 64;
        const double *cu1 = cfl + (size_t)(u0 + 1) * S4 * 64;
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
          if (s2 < S4) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(z[s2 >> 2][s2 & 3], cu0[s2 * 64], acc0, 0, 0, 0);
            if (two) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(z[s2 >> 2][s2 & 3], cu1[s2 * 64], acc1, 0, 0, 0);
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (h == 0 || two) {
            const int u = u0 + h;
            const d4_t acc = h ? acc1 : acc0;
            // acc[k] = r for alpha i = 16u + li and rows 4k + g of this tile
            const double be = betl[16 * u + li];
            const double q0 = __builtin_fma(-be, acc[0], 1.0), q1 = __builtin_fma(-be, acc[1], 1.0);
            const double q2 = __builtin_fma(-be, acc[2], 1.0), q3 = __builtin_fma(-be, acc[3], 1.0);
            const double m01 = q0 * q1, m23 = q2 * q3, m = m01 * m23;
            const double num = (acc[0] * q1 + acc[1] * q0) * m23 + (acc[2] * q3 + acc[3] * q2) * m01;
            double rq = num / m;  // = sum_k r_k / q_k over the 4 rows
            const int sgn = __double2hiint(q0) | __double2hiint(q1) | __double2hiint(q2) | __double2hiint(q3);
            rq = (sgn < 0) ? qnan : rq;  // some q < 0: log(q) is NaN in the reference
            R[u < NUM ? u : 0] += rq;
            const double pm = P[u < NUM ? u : 0] * m;
            E[u < NUM ? u : 0] += __builtin_amdgcn_frexp_exp(pm);
            P[u < NUM ? u : 0] = __builtin_amdgcn_frexp_mant(pm);
          }
        }
      }
    }
  }

  // ---- combine the 4 lane groups that share an alpha (lanes l, l^16, l^32, l^48), then the 4 waves
  __syncthreads();  // cfrag no longer needed -> reuse as reduction scratch
  double *redP = sm;              // [4][NA16]
  double *redR = redP + 4 * NA16;
  int *redE = reinterpret_cast<int *>(redR + 4 * NA16);
#pragma unroll
  for (int u = 0; u < NUM; ++u) {
    if (u < NU) {
      double pv = P[u], rv = R[u];
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
    po[i] = log(pv) + (double)ev * 0.6931471805599453094;  // sum_k log q_k
    po[NA16 + i] = rv;                                      // sum_k r_k / q_k
  }
}

// NLL assembly, det over/underflow emulation (rule (i)), NaN-first argmin (numpy.argmin).
// The 201 x p logarithms of a column are spread over all threads into an LDS table when it fits (TBL), and
// summed per alpha in the original order: same bits as the one-thread-per-alpha loop, a third of its time.
template <bool TBL>
__global__ __launch_bounds__(1024) void k_nll(const double *__restrict__ part, int nsplit, const int32_t *__restrict__ nuse,
                                              const double *__restrict__ d, const double *__restrict__ lam,
                                              const int32_t *__restrict__ status, const double *__restrict__ alphas,
                                              int nalpha, int p, int NA16, int rq_scaled_all,
                                              double *__restrict__ nll_out, int32_t *__restrict__ alphaidx,
                                              double *__restrict__ rest_out, const int32_t *__restrict__ rq_col) {
  extern __shared__ double tbl[];   // TBL: [nalpha][p] log(n beta_i lam_j + alpha_i), then [p] log(100 d_j)
  __shared__ double snll[SF_NALPHA_MAX];
  __shared__ double slogd;
  const int c = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const int st = status[c];
  const double n = (double)nuse[c];
  const double inf = __builtin_inf();
  // which sweep wrote this column's partial sums: the 4x4x4 kernels hand over beta sum r/q (and count their rows), the
  // 16x16x4 kernel sum r/q; rq_col (the rank-factorisation's verdict per column) when the two share a launch
  const int rq_scaled = rq_col ? (rq_col[c] != 0) : rq_scaled_all;
  if (TBL && st == 0) {
    double *tld = tbl + (size_t)nalpha * p;
    for (int j = tid; j < p; j += nthr) tld[j] = log(d[(size_t)c * p + j] * 100.0);
    for (int idx = tid; idx < nalpha * p; idx += nthr) {
      const int i = idx / p, j = idx - i * p;
      const double a = alphas[i];
      const double nb = n * ((1.0 - a) / (n - 1.0));
      tbl[idx] = log(nb * lam[(size_t)c * p + j] + a);
    }
    __syncthreads();
  }
  if (tid == 0) {
    double s = 0.0;
    if (st == 0)
      for (int j = 0; j < p; ++j)
        s += TBL ? tbl[(size_t)nalpha * p + j] : log(d[(size_t)c * p + j] * 100.0);  // diag of cov(100 x), robust_mf.py:94-99
    slogd = 2.0 * s;
  }
  __syncthreads();
  for (int i = tid; i < nalpha; i += nthr) {
    double v = inf;
    if (st == 0) {
      const double a = alphas[i];
      const double beta = (1.0 - a) / (n - 1.0);
      const double nb = n * beta;
      double ld = slogd;
      for (int j = 0; j < p; ++j) ld += TBL ? tbl[(size_t)i * p + j] : log(nb * lam[(size_t)c * p + j] + a);
      double lsum = 0.0, rsum = 0.0;
      for (int sp = 0; sp < nsplit; ++sp) {
        const double *pp = part + ((size_t)c * nsplit + sp) * 2 * NA16;
        lsum += pp[i];
        rsum += pp[NA16 + i];
      }
      // k_sweep4 hands over beta sum r/q.  beta <= 0 (the last grid point, alpha = 1 + 3e-13): q = 1 to rounding and
      // sum_k r_k = sum_j (sum_k y_kj^2) / (n beta lam_j + alpha) = (rows - 1) sum_j lam_j / (n beta lam_j + alpha)
      // (= (rows - 1) p / alpha for the diagonal target, where sum lam = trace R = p), rows = the rows the covariance
      // was made of (the sweep counts them: a cluster's statistics are swept with the COLUMN's n, robust_mf.py:355-356).
      if (rq_scaled) {
        double rows = n;
        if (nalpha < NA16) {
          rows = 0.0;
          for (int sp = 0; sp < nsplit; ++sp) rows += part[((size_t)c * nsplit + sp) * 2 * NA16 + 2 * NA16 - 1];
        }
        if (beta > 0.0) {
          rsum = rsum / beta;
        } else {
          double tr = 0.0;
          for (int j = 0; j < p; ++j) { const double lj = lam[(size_t)c * p + j]; tr += lj / (nb * lj + a); }
          rsum = (rows - 1.0) * tr;
        }
      }
      // det in the denormal range: the reference's product of LU pivots is ROUNDED to a multiple of 2^-1074 before
      // log() sees it -- a true determinant in [2^-1075, 2^-1074) rounds up to the smallest denormal (a finite NLL with
      // log det = -744.44), below 2^-1075 it rounds to 0 (skipped).  exp() here rounds the same way; the rounding of the
      // reference's INTERMEDIATE products (at most the last two factors) is not reproduced.
      if (ld < -708.0) {
        const double dt = exp(ld);
        ld = (dt > 0.0) ? log(dt) : -inf;
      }
      // everything but the determinant term: the exact-determinant pass of the wide windows (linalg.hip) adds its own
      if (rest_out) rest_out[(size_t)c * nalpha + i] = 0.5 * ((double)p * 1.8378770664093453) + 1.0 / (2.0 * n) * (lsum + rsum);
      if (ld < -745.2) {
        v = inf;  // det underflowed to 0 -> the reference skips this alpha (robust_mf.py:112-113)
      } else {
        if (ld >= 709.782712893384) ld = inf;  // det overflowed -> log(inf)
        v = 0.5 * ((double)p * 1.8378770664093453 + ld) + 1.0 / (2.0 * n) * (lsum + rsum);
      }
    }
    snll[i] = v;
    if (nll_out) nll_out[(size_t)c * nalpha + i] = v;
  }
  __syncthreads();
  if (tid == 0) {
    int idx = -1;
    double best = inf;
    for (int i = 0; i < nalpha; ++i) {
      const double v = snll[i];
      if (v != v) { idx = i; break; }          // numpy.argmin returns the first NaN
      if (v < best) { best = v; idx = i; }
    }
    alphaidx[c] = idx;                          // -1: every NLL is +inf (robust_mf.py:123-127)
  }
}

template <int NT, int S4C, int NUC, bool WREG, typename XT>
int launch_sweep_t(const XT *xt, const uint8_t *mask_t, const int32_t *nuse, const double *mu, const double *lam,
                   const double *wfrag, size_t wstride, const int32_t *status, const double *alphas, const SfGeom &g,
                   int nsplit, double *part, hipStream_t st, const int32_t *taken = nullptr) {
  const size_t lds = ((size_t)g.nu * g.s4 * 64 + g.ps + g.nu * 16) * sizeof(double);
  const size_t lds_red = (size_t)4 * g.nu * 16 * (2 * sizeof(double) + sizeof(int));
  const size_t need = lds > lds_red ? lds : lds_red;
  if (need > 160 * 1024) {
    sf_set_error("alpha grid x active window (%d x %d) does not fit the LDS-resident sweep", g.nalpha, g.p);
    return -2;
  }
  if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_sweep<NT, S4C, NUC, WREG, XT>), need)) return rc;
  int rows = sf_cdiv(g.lines, nsplit);
  rows = (rows + 63) / 64 * 64;
  hipLaunchKernelGGL((k_sweep<NT, S4C, NUC, WREG, XT>), dim3(g.ncols, nsplit), dim3(256), need, st, xt, mask_t, nuse, mu,
                     lam, wfrag, wstride, status, alphas, g.nalpha, g.lines, g.p, g.ps, rows, part, taken);
  SF_LAUNCH_CHECK("k_sweep");
  return 0;
}

#define SW_ARGS xt, mask_t, nuse, mu, lam, wfrag, wstride, status, alphas, g, nsplit, part, st, taken
template <typename XT>
int launch_sweep(const XT *xt, const uint8_t *mask_t, const int32_t *nuse, const double *mu, const double *lam,
                 const double *wfrag, size_t wstride, const int32_t *status, const double *alphas, const SfGeom &g,
                 int nsplit, double *part, hipStream_t st, const int32_t *taken = nullptr) {
  if constexpr (sizeof(XT) == 4) {
    const bool std_grid = g.nu == 13;
    // production windows get fully specialised kernels (CH4 radiance p = 72; CO2 p = 83)
    if (std_grid && g.s4 == 18) return launch_sweep_t<5, 18, 13, true, XT>(SW_ARGS);
    if (std_grid && g.s4 == 21) return launch_sweep_t<6, 21, 13, false, XT>(SW_ARGS);
  }
  switch (g.nt) {
    case 1: return launch_sweep_t<1, 0, 0, true, XT>(SW_ARGS);
    case 2: return launch_sweep_t<2, 0, 0, true, XT>(SW_ARGS);
    case 3: return launch_sweep_t<3, 0, 0, true, XT>(SW_ARGS);
    case 4: return launch_sweep_t<4, 0, 0, true, XT>(SW_ARGS);
    case 5: return launch_sweep_t<5, 0, 0, false, XT>(SW_ARGS);
    case 6: return launch_sweep_t<6, 0, 0, false, XT>(SW_ARGS);
    default:
      sf_set_error("active window of %d bands exceeds the fused statistics path (max %d)", g.p, SF_MAX_ACTIVE_FUSED);
      return -2;
  }
}
#undef SW_ARGS

int launch_nll(const double *part, int nsplit, const int32_t *nuse, const double *d, const double *lam, const int32_t *status,
               const double *alphas, const SfGeom &g, int rq_scaled, double *nll, int32_t *alphaidx, hipStream_t st,
               double *rest = nullptr, const int32_t *rq_col = nullptr) {
  const size_t lds = ((size_t)g.nalpha * g.p + g.p) * sizeof(double);
  if (lds <= 150 * 1024 && g.ncols <= 256) {   // one 116 KB workgroup per CU: only worth it when the launch is a single round
    if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_nll<true>), lds)) return rc;
    hipLaunchKernelGGL(k_nll<true>, dim3(g.ncols), dim3(1024), lds, st, part, nsplit, nuse, d, lam, status, alphas, g.nalpha,
                       g.p, g.nu * 16, rq_scaled, nll, alphaidx, rest, rq_col);
  } else {
    hipLaunchKernelGGL(k_nll<false>, dim3(g.ncols), dim3(256), 0, st, part, nsplit, nuse, d, lam, status, alphas, g.nalpha,
                       g.p, g.nu * 16, rq_scaled, nll, alphaidx, rest, rq_col);
  }
  SF_LAUNCH_CHECK("k_nll");
  return 0;
}

}  // namespace


int sf_launch_nll_finish(const double *part, int nsplit, const int32_t *nuse, const double *d, const double *lam,
                         const int32_t *status, const double *alphas, const SfGeom &g, double *nll, int32_t *alphaidx,
                         hipStream_t st, double *rest) {
  return launch_nll(part, nsplit, nuse, d, lam, status, alphas, g, 0, nll, alphaidx, st, rest);
}

static bool sweep4_ok(const SfGeom &g, int xt_f64) {
  return !xt_f64 && sf_tune().sweep_variant != 1 && g.nu == SF_SW4_NM && sf_sw4_groups(g.p) != 0;
}

// per column: the W fragments of the 16x16x4 sweep ([nt][s4][64]) or of the 4x4x4 sweeps ([nj][nje][16], nje = nj rounded up
// to even); windows of 21 / 24 band groups keep BOTH (the columns whose rank factorisation is refused take the 16x16x4 kernel)
size_t sf_wfrag_elems(const SfGeom &g) {
  const int nj = sf_sw4_groups(g.p);
  const size_t a = (size_t)g.nt * g.s4 * 64, b = (size_t)nj * (nj + (nj & 1)) * 16;
  if (nj > SF_SW4_NJ) return a + b;
  return a > b ? a : b;
}

size_t sf_loocv_scratch_bytes(const SfGeom &g) {
  const int nsplit = sf_sweep_splits(g.lines, g.ncols);
  return sf_align((size_t)g.ncols * sf_wfrag_elems(g) * sizeof(double)) +
         sf_align((size_t)g.ncols * nsplit * 2 * g.nu * 16 * sizeof(double)) + sf_lowrank_bytes(g);
}

int sf_launch_loocv(const void *xt, int xt_f64, const uint8_t *mask_t, const int32_t *nuse, const double *mu, const double *d,
                    const double *lam, const double *evec, const int32_t *status, const double *alphas,
                    const SfGeom &g, double *nll, int32_t *alphaidx, void *scratch, hipStream_t st) {
  if (g.nalpha < 1 || g.nalpha > SF_NALPHA_MAX) {
    sf_set_error("alpha grid of %d points is outside 1..%d", g.nalpha, SF_NALPHA_MAX);
    return -2;
  }
  const int nsplit = sf_sweep_splits(g.lines, g.ncols);
  const size_t wstride = sf_wfrag_elems(g);
  double *wfrag = reinterpret_cast<double *>(scratch);
  double *part = reinterpret_cast<double *>(reinterpret_cast<char *>(scratch) +
                                            sf_align((size_t)g.ncols * wstride * sizeof(double)));
  if (sweep4_ok(g, xt_f64)) {
    if (int rcw = sf_launch_wfrag4(evec, d, g, wstride, wfrag, st)) return rcw;
    void *lr = reinterpret_cast<char *>(part) + sf_align((size_t)g.ncols * nsplit * 2 * g.nu * 16 * sizeof(double));
    const int nj = sf_sw4_groups(g.p);
    if (nj > SF_SW4_NJ) {
      // 21 / 24 band groups (CO2: p = 83): the streamed rank-factored kernels; a column whose factorisation is refused (lrok == 0)
      // is swept by the 16x16x4 kernel from its own fragments (behind the 4x4x4 ones in the column's slot), and k_nll reads the
      // factorisation's verdict to tell the two forms of partial sums apart
      const int32_t *lrok = nullptr;
      int rc4 = sf_launch_sweep4((const float *)xt, mask_t, nuse, mu, lam, wfrag, wstride, status, alphas, g, nsplit, part, 0, lr, st,
                                 &lrok);
      if (rc4) return rc4;
      double *wfragG = wfrag + (size_t)nj * (nj + (nj & 1)) * 16;
      hipLaunchKernelGGL(k_wfrag, dim3(g.ncols), dim3(256), 0, st, evec, d, g.p, g.s4, g.nt, wstride, wfragG);
      SF_LAUNCH_CHECK("k_wfrag");
      if (int rc = launch_sweep((const float *)xt, mask_t, nuse, mu, lam, wfragG, wstride, status, alphas, g, nsplit, part, st, lrok))
        return rc;
      return launch_nll(part, nsplit, nuse, d, lam, status, alphas, g, 1, nll, alphaidx, st, nullptr, lrok);
    }
    int rc4 = sf_launch_sweep4((const float *)xt, mask_t, nuse, mu, lam, wfrag, wstride, status, alphas, g, nsplit, part,
                               sf_tune().sweep_variant, sf_tune().sweep_variant == 2 ? nullptr : lr, st);
    if (rc4) return rc4;
    return launch_nll(part, nsplit, nuse, d, lam, status, alphas, g, 1, nll, alphaidx, st);
  }
  hipLaunchKernelGGL(k_wfrag, dim3(g.ncols), dim3(256), 0, st, evec, d, g.p, g.s4, g.nt, wstride, wfrag);
  SF_LAUNCH_CHECK("k_wfrag");
  int rc = xt_f64 ? launch_sweep((const double *)xt, mask_t, nuse, mu, lam, wfrag, wstride, status, alphas, g, nsplit, part, st)
                  : launch_sweep((const float *)xt, mask_t, nuse, mu, lam, wfrag, wstride, status, alphas, g, nsplit, part, st);
  if (rc) return rc;
  return launch_nll(part, nsplit, nuse, d, lam, status, alphas, g, 0, nll, alphaidx, st);
}
