// Wide windows: a tridiagonal PRECONDITIONER for the blocked Jacobi eigensolver (round 4).
//
// The one-sided Jacobi of cmf_wide.hip orthogonalises the columns of the Cholesky factor L of a column's correlation
// matrix R = L L^T (eigenvalues = squared column norms, eigenvectors = normalised columns: relative accuracy on every
// eigenvalue, what the 201 det / inv of robust_mf.py:105-117 need at p = 425).  From L itself it takes 11-12 sweeps of
// 16 ms per flightline: the noise floor of a flightline column is a cluster of ~420 nearly equal eigenvalues, every
// rotation inside it is a large-angle one (profiles/r04_wjac_phase_clocks.txt).  Here the factor is rotated FIRST into
// nearly orthogonal columns by a cheap O(n^3) route whose own accuracy does not matter:
//
//   R = Q T Q^T          k_tridiag   Householder tridiagonalisation, panels of 8 columns with delayed rank-16 updates
//   T z_k = t_k z_k      k_tri_eig   eigenvalues by bisection (Sturm counts), vectors by a twisted factorisation each
//   U0 = Q Z S^-1        k_tri_back  the reflectors applied to the columns of Z (S = diag sqrt t_k)
//   W  = L^T U0          (GEMM)      ~ the right singular vectors of L: orthogonal to ~1e-11
//   W' = W (3 I - W^T W) / 2   (two GEMMs) one Newton-Schulz step: orthogonal to 1e-15, so that (L W')(L W')^T = R to rounding
//   F  = L W'            (GEMM)      columns orthogonal to ~1e-11 (cosines)
//
// and the Jacobi sweeps start from F instead of L: their first sweep finds only tiny rotations (|cos| <= 1e-9) and is the
// last (k_blockjac_flags).  Whatever the preconditioner gets wrong -- a tight pair of eigenvalues, a poor inverse-iteration
// vector -- costs Jacobi sweeps, never accuracy: F F^T = R holds to rounding because W' is orthogonal to rounding, and the
// result is still the Jacobi's.  A matrix whose tridiagonal eigenvalues are not all positive and finite keeps F = L.
// Measured (profiles/r04_wtri_ab.txt): the eigensolver of a 598 x 425 x 425 flightline 198 -> 85-92 ms, the flightline 450 -> 334 ms,
// alpha indices identical.  Used from 32 columns a call (cmf_wide.hip: below that its latency exceeds the plain sweeps').
#include "cmf_common.h"
#include "sf_tune.h"
#include <type_traits>

namespace {

constexpr int TR_NT = 512;      // threads of k_tridiag (one row each)
constexpr int TR_NB = 8;        // panel width
constexpr int TR_MG = TR_NT / 64;   // row groups of 64 a lane may own in the symv
constexpr int TR_CU = 6;            // columns a wave has in flight there (x 7 row groups: 42 loads per lane)

template <int CTRL>
__device__ __forceinline__ double tr_dpp_swap(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// sum over the 64 lanes, the same value in every lane: DPP butterflies inside the rows of 16, the four rows by readlane (no
// LDS crossbar round trips: the corrections of a tridiagonalisation step reduce up to 18 values)
__device__ __forceinline__ double tr_wave_sum(double v) {
  v += tr_dpp_swap<0xB1>(v);
  v += tr_dpp_swap<0x4E>(v);
  v += tr_dpp_swap<0x141>(v);
  v += tr_dpp_swap<0x140>(v);
  auto rdl = [](double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
  };
  return (rdl(v, 0) + rdl(v, 16)) + (rdl(v, 32) + rdl(v, 48));
}

// block-wide sums of NV values per thread: partial sums per wave in red[NV][8], every thread gets the totals
template <int NV>
__device__ __forceinline__ void tr_block_sum(double (&v)[NV], double *red, int nv) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (i < nv) {
      const double s = tr_wave_sum(v[i]);
      if (lane == 0) red[i * 8 + wave] = s;
    }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (i < nv) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < TR_NT / 64; ++w) s += red[i * 8 + w];
      v[i] = s;
    }
  __syncthreads();
}

// Two-level batch: matrix mtx of a launch is matrix mtx % gbn of column group mtx / gbn; a group's buffers sit gstride BYTES behind
// the previous group's (cmf_wide.hip lays a group's work out in one block) -- so the launches that are latency-bound per
// workgroup (tridiagonalisation, bisection) can run over all the groups of a flightline at once.
template <typename T>
__device__ __forceinline__ T *tr_goff(T *base, int grp, size_t gstride) {
  return reinterpret_cast<T *>(reinterpret_cast<char *>(const_cast<typename std::remove_const<T>::type *>(base)) + (size_t)grp * gstride);
}

// A (column-major, ld) <- copy of the n x n matrix in G (column-major, ldg) for matrices with flag 0
__global__ void k_tri_copy(const double *__restrict__ G, size_t sG, int ldg, double *__restrict__ A, size_t sA, int lda, int n,
                           const int32_t *__restrict__ cflag, int gbn, size_t gstride) {
  const int grp = blockIdx.y / gbn, mtx = blockIdx.y - grp * gbn;
  G = tr_goff(G, grp, gstride); A = tr_goff(A, grp, gstride); cflag = tr_goff(cflag, grp, gstride);
  if (cflag[mtx] != 0) return;
  const double *g = G + (size_t)mtx * sG;
  double *a = A + (size_t)mtx * sA;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * n; i += gridDim.x * blockDim.x) {
    const int c = i / n, r = i - c * n;
    a[(size_t)c * lda + r] = g[(size_t)c * ldg + r];
  }
}

// Householder tridiagonalisation of the symmetric n x n matrix A (column-major, ld, BOTH triangles stored and kept).
// On exit: de[0..n) = diagonal of T, de[pl..pl+n-1) = subdiagonal, column j of A holds reflector j in rows j+1.. (v[j+1] = 1
// stored explicitly), A[j][j] = tau_j.  Q = H_0 H_1 ... H_{n-2}, H_j = I - tau_j v_j v_j^T, T = Q^T A Q.
// One workgroup per matrix, thread = row.  Panels of TR_NB columns (LAPACK dlatrd's scheme): inside a panel the trailing matrix
// in memory stays as it was at the panel's start, a column is brought up to date by the panel's vectors V, W (LDS) when its
// turn comes, p = A v reads the stale trailing matrix and is corrected by V (W^T v) + W (V^T v); after the panel the rank-16
// update A -= V W^T + W V^T is applied to the rows / columns behind it.  Traffic: the trailing matrix once per column (the symv)
// plus once per panel -- served by the Infinity Cache for a group of ~150 matrices.
__device__ unsigned long long g_tr_stamps[8];   // phase clocks of workgroup 0 (sf_debug_wtri_stamps): reflector, symv, corrections, update
__global__ __launch_bounds__(TR_NT) void k_tridiag(double *__restrict__ Aall, size_t sA, int ld, int n, double *__restrict__ deall,
                                                   size_t sDE, int pl, const int32_t *__restrict__ cflag, int gbn, size_t gstride) {
  extern __shared__ __attribute__((aligned(16))) double trs[];
  const int grp = blockIdx.x / gbn, mtx = blockIdx.x - grp * gbn;
  Aall = tr_goff(Aall, grp, gstride); deall = tr_goff(deall, grp, gstride); cflag = tr_goff(cflag, grp, gstride);
  if (cflag[mtx] != 0) return;
  double *A = Aall + (size_t)mtx * sA;
  double *de = deall + (size_t)mtx * sDE;
  const int tid = threadIdx.x, r = tid;
  const int nl = n;                       // LDS row length
  double *V = trs;                        // [TR_NB][nl]
  double *W = V + TR_NB * nl;             // [TR_NB][nl]
  double *red = W + TR_NB * nl;           // [2 * TR_NB + 2][8]
  double *bc = red + (2 * TR_NB + 2) * 8; // broadcast scalars
  double *part = bc + 8;                  // [8 waves][nl]: the symv's partial sums
  const bool rin = r < n;
  const int mg = (n + 63) >> 6;           // row groups of 64
  for (int j0 = 0; j0 < n - 1; j0 += TR_NB) {
    const int nbp = min(TR_NB, n - 1 - j0);
    for (int i = 0; i < nbp; ++i) {
      const int j = j0 + i;
      unsigned long long tq0 = 0, tq1 = 0, tq2 = 0, tq3 = 0;
      const bool stampw = blockIdx.x == 0 && tid == 0;
      if (stampw) tq0 = __builtin_readcyclecounter();
      // (a) column j brought up to date (rows >= j)
      double a = 0.0;
      if (rin && r >= j) {
        a = A[(size_t)j * ld + r];
        for (int t = 0; t < i; ++t) a -= V[t * nl + r] * W[t * nl + j] + W[t * nl + r] * V[t * nl + j];
      }
      // (b) the reflector of x = a[j+1 ..]
      double s1[1] = {(rin && r >= j + 2) ? a * a : 0.0};
      if (rin && r == j) bc[0] = a;
      if (rin && r == j + 1) bc[1] = a;
      tr_block_sum<1>(s1, red, 1);        // (its barriers also publish bc)
      const double xn2 = s1[0], alpha = bc[1];
      double tau = 0.0, beta = alpha, scal = 0.0;
      if (xn2 > 0.0) {
        beta = -copysign(sqrt(alpha * alpha + xn2), alpha);
        tau = (beta - alpha) / beta;
        scal = 1.0 / (alpha - beta);
      }
      double v = 0.0;
      if (rin && r == j + 1) v = 1.0;
      else if (rin && r >= j + 2) v = a * scal;
      if (rin) {
        V[i * nl + r] = v;                       // (zero in rows <= j)
        if (r >= j + 1) A[(size_t)j * ld + r] = v;
        if (r == j) { de[j] = bc[0]; de[pl + j] = beta; A[(size_t)j * ld + j] = tau; }
      }
      __syncthreads();
      // (c) p = A_stale v over rows / columns >= j + 1, and (d) the partial dots W_t . v, V_t . v of the correction
      if (stampw) tq1 = __builtin_readcyclecounter();
      //     The symv is latency-bound if a thread walks its row alone (8 loads in flight); so wave w takes the columns
      //     k = j + 1 + w (mod 8), its lanes the rows lane + 64 m of every row group still alive: 6 x 7 loads in flight per lane,
      //     the eight partial sums of a row meet in LDS.  (Reading only the lower triangle -- a = A[r][k] serving p_r and p_k --
      //     halves the traffic and was 45 % SLOWER: the per-column wave reductions cost more than the bytes.)
      {
        const int lane = tid & 63, wave = tid >> 6;
        const int m0 = (j + 1) >> 6;
        const double *vi = V + i * nl;
        double ps[TR_MG];
#pragma unroll
        for (int m = 0; m < TR_MG; ++m) ps[m] = 0.0;
        int k = j + 1 + wave;
        for (; k + 8 * (TR_CU - 1) < n; k += 8 * TR_CU) {
          double x[TR_CU][TR_MG];
#pragma unroll
          for (int u = 0; u < TR_CU; ++u)
#pragma unroll
            for (int m = 0; m < TR_MG; ++m)
              if (m >= m0 && m < mg) x[u][m] = (A + (size_t)(k + 8 * u) * ld + lane)[64 * m];   // (rows past n - 1: the next column's, unused)
#pragma unroll
          for (int u = 0; u < TR_CU; ++u) {
            const double vk = vi[k + 8 * u];
#pragma unroll
            for (int m = 0; m < TR_MG; ++m)
              if (m >= m0 && m < mg) ps[m] = __builtin_fma(x[u][m], vk, ps[m]);
          }
        }
        for (; k < n; k += 8) {
          const double vk = vi[k];
#pragma unroll
          for (int m = 0; m < TR_MG; ++m)
            if (m >= m0 && m < mg) ps[m] = __builtin_fma((A + (size_t)k * ld + lane)[64 * m], vk, ps[m]);
        }
#pragma unroll
        for (int m = 0; m < TR_MG; ++m)
          if (m < mg && lane + 64 * m < n) part[wave * nl + lane + 64 * m] = (m >= m0) ? ps[m] : 0.0;
      }
      __syncthreads();
      double p = 0.0;
      if (rin && r >= j + 1) {
#pragma unroll
        for (int w = 0; w < TR_NT / 64; ++w) p += part[w * nl + r];
      }
      if (stampw) tq2 = __builtin_readcyclecounter();
      double dots[2 * TR_NB];
#pragma unroll
      for (int t = 0; t < TR_NB; ++t) {
        dots[2 * t] = (t < i && rin) ? W[t * nl + r] * v : 0.0;
        dots[2 * t + 1] = (t < i && rin) ? V[t * nl + r] * v : 0.0;
      }
      if (i > 0) tr_block_sum<2 * TR_NB>(dots, red, 2 * i);
      if (rin && r >= j + 1) {
#pragma unroll
        for (int t = 0; t < TR_NB; ++t)
          if (t < i) p -= V[t * nl + r] * dots[2 * t] + W[t * nl + r] * dots[2 * t + 1];
      }
      // (e) w = tau p - (tau / 2)(tau p . v) v
      p *= tau;
      double s2[1] = {(rin && r >= j + 1) ? p * v : 0.0};
      tr_block_sum<1>(s2, red, 1);
      const double al2 = -0.5 * tau * s2[0];
      if (rin) W[i * nl + r] = (r >= j + 1) ? p + al2 * v : 0.0;
      __syncthreads();
      if (stampw) {
        tq3 = __builtin_readcyclecounter();
        g_tr_stamps[0] += tq1 - tq0; g_tr_stamps[1] += tq2 - tq1; g_tr_stamps[2] += tq3 - tq2; g_tr_stamps[4] += 1;
      }
    }
    unsigned long long tu0 = 0;
    if (blockIdx.x == 0 && tid == 0) tu0 = __builtin_readcyclecounter();
    // the rank-2nb update of what lies behind the panel: rows and columns >= j0 + nbp (both triangles).  On the matrix cores:
    // as scalar code every multiply-add fetched a broadcast value of V / W from LDS and the update -- 16 flops per element --
    // was LDS-issue-bound at 360 k cycles a panel, 85 % of the kernel.  A 16 x 16 tile of A^T (the block is symmetric: rows
    // and columns swap roles so that a lane's four elements are 16-lane contiguous runs of a column) takes four
    // v_mfma_f64_16x16x4 with operands X = [V W], Y = [W V] read conflict-free from the t-major LDS panels.
    const int c1 = j0 + nbp;
    if (c1 < n && nbp == TR_NB && n - c1 > 16) {
      typedef double d4_t __attribute__((ext_vector_type(4)));
      const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
      const int nt = (n - c1 + 15) >> 4;
      for (int q = wave; q < nt * nt; q += TR_NT / 64) {
        const int tc = q / nt, trw = q - tc * nt;
        const int C0 = c1 + 16 * tc, R0 = c1 + 16 * trw;
        const int row = R0 + li;
        d4_t acc;
        double *ap = A + (size_t)(C0 + g) * ld + row;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = (row < n && C0 + g + 4 * e < n) ? ap[(size_t)4 * e * ld] : 0.0;
        const int ci = min(C0 + li, n - 1), ri = min(row, n - 1);
        // D'[m = column g + 4 e][n = row li] -= sum_k Y[C0 + m][k] X[R0 + n][k]; A' operand: lane (g, li) = -Y[C0 + li][k0 + g],
        // B' operand: lane (g, li) = X[R0 + li][k0 + g]; X[r][k] = k < 8 ? V[k][r] : W[k - 8][r], Y[c][k] = k < 8 ? W[k][c] : V[k - 8][c]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-W[g * nl + ci], V[g * nl + ri], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-W[(4 + g) * nl + ci], V[(4 + g) * nl + ri], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-V[g * nl + ci], W[g * nl + ri], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-V[(4 + g) * nl + ci], W[(4 + g) * nl + ri], acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (row < n && C0 + g + 4 * e < n) ap[(size_t)4 * e * ld] = acc[e];
      }
    } else if (c1 < n) {
      double vr[TR_NB], wr[TR_NB];
#pragma unroll
      for (int t = 0; t < TR_NB; ++t) {
        vr[t] = (t < nbp && rin) ? V[t * nl + r] : 0.0;
        wr[t] = (t < nbp && rin) ? W[t * nl + r] : 0.0;
      }
      if (rin && r >= c1) {
        double *ar = A + r;
        for (int c = c1; c < n; ++c) {
          double acc = ar[(size_t)c * ld];
#pragma unroll
          for (int t = 0; t < TR_NB; ++t) acc -= vr[t] * W[t * nl + c] + wr[t] * V[t * nl + c];
          ar[(size_t)c * ld] = acc;
        }
      }
    }
    __syncthreads();
    __threadfence_block();
    if (blockIdx.x == 0 && tid == 0) g_tr_stamps[3] += __builtin_readcyclecounter() - tu0;
  }
  // the last diagonal element (no reflector past column n - 2; its column was updated as part of the last panel's trailing
  // block, or, when the last panel ended at n - 1, is row n - 1 of the stale matrix corrected here)
  if (tid == 0) {   // (the last panel ends at column n - 2: its trailing update covered row / column n - 1)
    const int jl = n - 1;
    de[jl] = A[(size_t)jl * ld + jl];
    de[pl + jl] = 0.0;
    A[(size_t)jl * ld + jl] = 0.0;        // tau of the missing reflector
  }
}

// Eigenvalues of the tridiagonal T (de: diagonal, subdiagonal) by bisection on the Sturm count, thread k the k-th smallest;
// then its eigenvector by ONE twisted factorisation (forward L D L^T, backward U D U^T of T - t_k I, the twist at the smallest
// |gamma|).  D+ goes to row-major scratch Zt (element i of thread k at Zt[i * ldz + k]: coalesced), D- to Dm likewise; the vector
// overwrites D+ in place.  de[2 pl + k] = t_k, de[3 pl + k] = 1 / (|z_k| sqrt(t_k)); pflag = 1 when some t_k is not positive and finite.
constexpr int TE_NT = 512;
constexpr int TE_PF = 8;      // scratch elements fetched together in the twisted-factorisation passes
__global__ __launch_bounds__(TE_NT) void k_tri_eig(double *__restrict__ deall, size_t sDE, int n, double *__restrict__ Ztall,
                                                   size_t sZ, int ldz, double *__restrict__ Dmall, size_t sD, int ldd,
                                                   int pl, const int32_t *__restrict__ cflag, int32_t *__restrict__ pflag, int gbn,
                                                   size_t gstride) {
  extern __shared__ __attribute__((aligned(16))) double tes[];
  const int grp = blockIdx.x / gbn, mtx = blockIdx.x - grp * gbn;
  deall = tr_goff(deall, grp, gstride); Ztall = tr_goff(Ztall, grp, gstride); Dmall = tr_goff(Dmall, grp, gstride);
  cflag = tr_goff(cflag, grp, gstride); pflag = tr_goff(pflag, grp, gstride);
  if (cflag[mtx] != 0) return;
  const double *de = deall + (size_t)mtx * sDE;
  double *Zt = Ztall + (size_t)mtx * sZ, *Dm = Dmall + (size_t)mtx * sD;
  double *dd = tes, *ee = dd + pl, *e2 = ee + pl;
  double *lamo = deall + (size_t)mtx * sDE + 2 * pl, *sclo = lamo + pl;
  __shared__ double gl[2];
  __shared__ int bad;
  const int tid = threadIdx.x;
  for (int i = tid; i < n; i += TE_NT) {
    const double d = de[i], e = (i < n - 1) ? de[pl + i] : 0.0;
    dd[i] = d;
    ee[i] = e;
    e2[i] = e * e;
  }
  if (tid == 0) bad = 0;
  __syncthreads();
  if (tid == 0) {   // Gershgorin bounds
    double lo = dd[0], hi = dd[0];
    for (int i = 0; i < n; ++i) {
      const double rad = (i > 0 ? fabs(ee[i - 1]) : 0.0) + (i < n - 1 ? fabs(ee[i]) : 0.0);
      lo = fmin(lo, dd[i] - rad);
      hi = fmax(hi, dd[i] + rad);
    }
    const double w = hi - lo;
    gl[0] = lo - 1e-3 * w - 1e-300;
    gl[1] = hi + 1e-3 * w + 1e-300;
  }
  __syncthreads();
  const int k = tid;
  double lam = 0.0;
  const double tnorm = fmax(fabs(gl[0]), fabs(gl[1]));
  const double tiny = 2.220446049250313e-16 * tnorm * 1e-3 + 1e-300;
  if (k < n) {
    // bisection; the quotient of the Sturm recurrence by reciprocal + two Newton steps (the launch is bound by the vector
    // pipes: a quadrisection with three independent chains a round does 1.5 x the evaluations and took as long), the last
    // rounds with the IEEE division
    double lo = gl[0], hi = gl[1];
    auto rcp = [](double q) {
      double y = __builtin_amdgcn_rcp(q);
      y = __builtin_fma(y, __builtin_fma(-q, y, 1.0), y);
      return __builtin_fma(y, __builtin_fma(-q, y, 1.0), y);
    };
    for (int it = 0; it < 120; ++it) {
      const double mid = 0.5 * (lo + hi);
      if (!(mid > lo && mid < hi)) break;   // two adjacent floats
      const bool exact = (hi - lo) <= 64.0 * 2.220446049250313e-16 * fmax(fabs(lo), fabs(hi));
      int cnt = 0;
      double q = dd[0] - mid;
      if (q == 0.0) q = -tiny;
      cnt += q < 0.0;
      if (exact) {
        for (int i = 1; i < n; ++i) {
          q = dd[i] - mid - e2[i - 1] / q;
          if (q == 0.0) q = -tiny;
          cnt += q < 0.0;
        }
      } else {
        for (int i = 1; i < n; ++i) {
          q = __builtin_fma(-e2[i - 1], rcp(q), dd[i] - mid);
          if (q == 0.0) q = -tiny;
          cnt += q < 0.0;
        }
      }
      if (cnt > k) hi = mid; else lo = mid;   // cnt = number of eigenvalues < mid
    }
    lam = 0.5 * (lo + hi);
    if (!(lam > 0.0) || !(lam <= 1.79769313486231570e+308)) bad = 1;
    // ---- twisted factorisation of T - lam I
    double dp = dd[0] - lam;
    if (dp == 0.0) dp = tiny;
    Zt[k] = dp;
    for (int i = 0; i < n - 1; ++i) {         // D+_{i+1} = (d_{i+1} - lam) - e_i^2 / D+_i
      dp = (dd[i + 1] - lam) - e2[i] / dp;
      if (dp == 0.0) dp = tiny;
      Zt[(size_t)(i + 1) * ldz + k] = dp;
    }
    // (the three passes below walk this thread's scratch columns in blocks of TE_PF elements, loaded together: a dependent
    //  global load per recurrence step is 2 us of latency each -- 2.5 ms a launch)
    double dm = dd[n - 1] - lam;
    if (dm == 0.0) dm = tiny;
    Dm[(size_t)(n - 1) * ldd + k] = dm;
    double gbest = fabs(dp + dm - (dd[n - 1] - lam));
    int rtw = n - 1;
    for (int ib = n - 2; ib >= 0; ib -= TE_PF) {          // D-_i = (d_i - lam) - e_i^2 / D-_{i+1}
      double dpl[TE_PF];
#pragma unroll
      for (int u = 0; u < TE_PF; ++u) dpl[u] = Zt[(size_t)max(ib - u, 0) * ldz + k];
#pragma unroll
      for (int u = 0; u < TE_PF; ++u) {
        const int i = ib - u;
        if (i >= 0) {
          dm = (dd[i] - lam) - e2[i] / dm;
          if (dm == 0.0) dm = tiny;
          Dm[(size_t)i * ldd + k] = dm;
          const double g = fabs(dpl[u] + dm - (dd[i] - lam));
          if (g < gbest) { gbest = g; rtw = i; }
        }
      }
    }
    // ---- z: z_r = 1, upward z_i = -(e_i / D+_i) z_{i+1}, downward z_{i+1} = -(e_i / D-_{i+1}) z_i; in place of D+
    double z = 1.0, nrm2 = 1.0;
    for (int ib = rtw - 1; ib >= 0; ib -= TE_PF) {
      double dpl[TE_PF];
#pragma unroll
      for (int u = 0; u < TE_PF; ++u) dpl[u] = Zt[(size_t)max(ib - u, 0) * ldz + k];
#pragma unroll
      for (int u = 0; u < TE_PF; ++u) {
        const int i = ib - u;
        if (i >= 0) {
          z = -(ee[i] / dpl[u]) * z;
          Zt[(size_t)i * ldz + k] = z;
          nrm2 += z * z;
        }
      }
    }
    z = 1.0;
    Zt[(size_t)rtw * ldz + k] = 1.0;
    for (int ib = rtw; ib < n - 1; ib += TE_PF) {
      double dml[TE_PF];
#pragma unroll
      for (int u = 0; u < TE_PF; ++u) dml[u] = Dm[(size_t)min(ib + u + 1, n - 1) * ldd + k];
#pragma unroll
      for (int u = 0; u < TE_PF; ++u) {
        const int i = ib + u;
        if (i < n - 1) {
          z = -(ee[i] / dml[u]) * z;
          Zt[(size_t)(i + 1) * ldz + k] = z;
          nrm2 += z * z;
        }
      }
    }
    lamo[k] = lam;
    const double sc = 1.0 / sqrt(nrm2 * fabs(lam));
    sclo[k] = sc;
    if (!(sc > 0.0) || !(sc <= 1.79769313486231570e+308) || !(nrm2 <= 1.79769313486231570e+308)) bad = 1;
  }
  __syncthreads();
  if (tid == 0) pflag[mtx] = bad;
}

// Z (column-major, ld) = Zt^T with column k scaled by scl[k]  (32 x 32 tiles through LDS)
__global__ __launch_bounds__(256) void k_tri_transpose(const double *__restrict__ Ztall, size_t sZt, int ldzt, double *__restrict__ Zall,
                                                       size_t sZ, int ldz, int n, const double *__restrict__ small, size_t sS, int pl,
                                                       const int32_t *__restrict__ cflag, const int32_t *__restrict__ pflag) {
  __shared__ double tile[32][33];
  const int mtx = blockIdx.z;
  if (cflag[mtx] != 0 || pflag[mtx] != 0) return;
  const double *Zt = Ztall + (size_t)mtx * sZt;
  double *Z = Zall + (size_t)mtx * sZ;
  const double *scl = small + (size_t)mtx * sS + 3 * pl;
  const int i0 = blockIdx.x * 32, k0 = blockIdx.y * 32;   // rows i (vector elements), columns k (eigenvalues)
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int y = ty; y < 32; y += 8) {
    const int i = i0 + y, k = k0 + tx;
    tile[y][tx] = (i < n && k < n) ? Zt[(size_t)i * ldzt + k] * scl[k] : 0.0;
  }
  __syncthreads();
  for (int y = ty; y < 32; y += 8) {
    const int k = k0 + y, i = i0 + tx;
    if (i < n && k < n) Z[(size_t)k * ldz + i] = tile[tx][y];
  }
}

// U0 = H_0 H_1 ... H_{n-2} Z, column by column: 16 lanes per column (row = sub + 16 t, RM rows a lane), 32 columns per workgroup, the
// reflectors staged through LDS eight at a time, zero above their first row and padded to 16 RM rows -- so the dot product and
// the update run over all RM rows without a predicate.  Reflector j lives in column j of A (rows j+1.., v[j+1] = 1), tau_j at A[j][j].
constexpr int TBK_NT = 512, TBK_CH = 8;
template <int RM>
__global__ __launch_bounds__(TBK_NT) void k_tri_back(const double *__restrict__ Aall, size_t sA, int lda, double *__restrict__ Zall,
                                                     size_t sZ, int ldz, int n, const int32_t *__restrict__ cflag,
                                                     const int32_t *__restrict__ pflag) {
  extern __shared__ __attribute__((aligned(16))) double tbs[];   // [2][TBK_CH][nl] reflectors + [2][TBK_CH] taus
  const int mtx = blockIdx.y;
  if (cflag[mtx] != 0 || pflag[mtx] != 0) return;
  const double *A = Aall + (size_t)mtx * sA;
  double *Z = Zall + (size_t)mtx * sZ;
  const int tid = threadIdx.x, grp = tid >> 4, sub = tid & 15;
  const int col = blockIdx.x * 32 + grp;
  const bool creal = col < n;
  const int nr = (n - sub + 15) >> 4;
  constexpr int nl = 16 * RM;
  double *taus = tbs + 2 * TBK_CH * nl;
  double x[RM];
  double *zc = Z + (size_t)min(col, n - 1) * ldz + sub;
#pragma unroll
  for (int t = 0; t < RM; ++t) {
    const double z = zc[16 * min(t, nr - 1)];
    x[t] = (t < nr && creal) ? z : 0.0;
  }
  const int nref = n - 1;                                  // reflectors 0 .. n-2
  const int nch = (nref + TBK_CH - 1) / TBK_CH;
  auto stage = [&](int ch, int buf) {                      // reflectors jhi-1 .. jhi-TBK_CH (descending), jhi = nref - ch * TBK_CH
    const int jhi = nref - ch * TBK_CH;
    double *dst = tbs + (size_t)buf * TBK_CH * nl;
    for (int e = tid; e < TBK_CH * nl; e += TBK_NT) {
      const int u = e / nl, rr = e - u * nl, j = jhi - 1 - u;
      dst[e] = (j >= 0 && rr > j && rr < n) ? A[(size_t)j * lda + rr] : 0.0;
    }
    if (tid < TBK_CH) { const int j = jhi - 1 - tid; taus[buf * TBK_CH + tid] = (j >= 0) ? A[(size_t)j * lda + j] : 0.0; }
  };
  stage(0, 0);
  __syncthreads();
  for (int ch = 0; ch < nch; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nch) stage(ch + 1, buf ^ 1);
    const double *vb = tbs + (size_t)buf * TBK_CH * nl + sub;
    // the reflectors of this chunk are zero in rows 0 .. jlo: the row slots below (jlo + 1) / 16 are skipped, in steps of RM / 4
    // (compile-time loop bounds: the column stays in registers)
    const int jlo = max(nref - (ch + 1) * TBK_CH, 0);
    auto run = [&](auto t0c) {
      constexpr int T0 = decltype(t0c)::value;
#pragma unroll 2
      for (int u = 0; u < TBK_CH; ++u) {                   // (a reflector past the first is all zeros with tau = 0: harmless)
        const double *vj = vb + (size_t)u * nl;
        double vv[RM];
#pragma unroll
        for (int t = T0; t < RM; ++t) vv[t] = vj[16 * t];
        double d0 = 0.0, d1 = 0.0;
#pragma unroll
        for (int t = T0; t + 1 < RM; t += 2) { d0 = __builtin_fma(vv[t], x[t], d0); d1 = __builtin_fma(vv[t + 1], x[t + 1], d1); }
        if ((RM - T0) & 1) d0 = __builtin_fma(vv[RM - 1], x[RM - 1], d0);
        double dot = d0 + d1;
        dot += tr_dpp_swap<0xB1>(dot);
        dot += tr_dpp_swap<0x4E>(dot);
        dot += tr_dpp_swap<0x141>(dot);
        dot += tr_dpp_swap<0x140>(dot);
        const double f = -taus[buf * TBK_CH + u] * dot;
#pragma unroll
        for (int t = T0; t < RM; ++t) x[t] = __builtin_fma(f, vv[t], x[t]);
      }
    };
    constexpr int Q = RM / 4;
    const int tz = (jlo + 1) >> 4;
    if (tz >= 3 * Q) run(std::integral_constant<int, 3 * Q>{});
    else if (tz >= 2 * Q) run(std::integral_constant<int, 2 * Q>{});
    else if (tz >= Q) run(std::integral_constant<int, Q>{});
    else run(std::integral_constant<int, 0>{});
    __syncthreads();
  }
  if (creal) {
#pragma unroll
    for (int t = 0; t < RM; ++t)
      if (t < nr) zc[16 * t] = x[t];
  }
}

// M = 1.5 I - 0.5 G  (n x n, in place; G = W^T W)
__global__ void k_tri_nsm(double *__restrict__ Gall, size_t sG, int ld, int n, const int32_t *__restrict__ cflag,
                          const int32_t *__restrict__ pflag) {
  const int mtx = blockIdx.y;
  if (cflag[mtx] != 0 || pflag[mtx] != 0) return;
  double *G = Gall + (size_t)mtx * sG;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * n; i += gridDim.x * blockDim.x) {
    const int c = i / n, r = i - c * n;
    const double g = G[(size_t)c * ld + r];
    G[(size_t)c * ld + r] = (r == c ? 1.5 : 0.0) - 0.5 * g;
  }
}

// G (the Jacobi's work matrix, column-major ldg) <- F where the preconditioner succeeded and F is finite
__global__ void k_tri_select(const double *__restrict__ Fall, size_t sF, int ldf, double *__restrict__ Gall, size_t sG, int ldg, int n,
                             const int32_t *__restrict__ cflag, const int32_t *__restrict__ pflag) {
  const int mtx = blockIdx.y;
  if (cflag[mtx] != 0 || pflag[mtx] != 0) return;
  const double *F = Fall + (size_t)mtx * sF;
  double *G = Gall + (size_t)mtx * sG;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * n; i += gridDim.x * blockDim.x) {
    const int c = i / n, r = i - c * n;
    G[(size_t)c * ldg + r] = F[(size_t)c * ldf + r];
  }
}

// a matrix whose preconditioner was refused goes the way of one whose Cholesky failed: the single-workgroup eigensolver from
// the covariance itself (k_eigh_global, mode 2) -- the few sweeps launched behind the preconditioner would not finish it
__global__ void k_tri_demote(int nb, int32_t *__restrict__ cflag, const int32_t *__restrict__ pflag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nb && cflag[i] == 0 && pflag[i] != 0) cflag[i] = 1;
}

// pflag |= 1 where F has a non-finite entry (checked before the copy)
__global__ void k_tri_check(const double *__restrict__ Fall, size_t sF, int ldf, int n, const int32_t *__restrict__ cflag,
                            int32_t *__restrict__ pflag) {
  const int mtx = blockIdx.y;
  if (cflag[mtx] != 0 || pflag[mtx] != 0) return;
  const double *F = Fall + (size_t)mtx * sF;
  bool bad = false;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * n; i += gridDim.x * blockDim.x) {
    const int c = i / n, r = i - c * n;
    const double f = F[(size_t)c * ldf + r];
    bad = bad || !(fabs(f) <= 1.79769313486231570e+308);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&pflag[mtx], 2);
}

}  // namespace

// gv: [nb][2][p2 * p2]: half 0 = the correlation matrix R on entry of `prepare` (before the Cholesky), the factor L on entry of
// `apply`; half 1 = work (R's copy -> reflectors -> G / M).  B2, B3: [nb][p * p] work matrices.  small: [nb][4 * pl] doubles
// (d, e | t_k | scales), pl = p rounded up to 16.  pflag: [nb].
size_t sf_wtri_small_bytes(int p, int nb) { return sf_align((size_t)nb * 4 * ((p + 15) & ~15) * sizeof(double)); }

int sf_launch_wtri_prepare(double *gv, int p, int p2, int nb, double *B2, double *B3, double *small, const int32_t *cflag,
                           int32_t *pflag, hipStream_t st, int gbn, size_t gstride) {
  if (gbn <= 0) { gbn = nb; gstride = 0; }
  if (p > TR_NT || p < 4) { sf_set_error("tridiagonal preconditioner: %d bands unsupported", p); return -2; }
  const size_t sG = (size_t)2 * p2 * p2, sB = (size_t)p * p;
  const int pl = (p + 15) & ~15;
  const size_t sS = (size_t)4 * pl;
  double *A1 = gv + (size_t)p2 * p2;
  hipLaunchKernelGGL(k_tri_copy, dim3(64, nb), dim3(256), 0, st, gv, sG, p2, A1, sG, p2, p, cflag, gbn, gstride);
  SF_LAUNCH_CHECK("k_tri_copy");
  const size_t lds1 = ((size_t)2 * TR_NB * p + (2 * TR_NB + 2) * 8 + 8 + (size_t)(TR_NT / 64) * p) * sizeof(double);
  if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_tridiag), lds1)) return rc;
  hipLaunchKernelGGL(k_tridiag, dim3(nb), dim3(TR_NT), lds1, st, A1, sG, p2, p, small, sS, pl, cflag, gbn, gstride);
  SF_LAUNCH_CHECK("k_tridiag");
  const size_t lds2 = (size_t)3 * pl * sizeof(double);
  hipLaunchKernelGGL(k_tri_eig, dim3(nb), dim3(TE_NT), lds2, st, small, sS, p, B3, sB, p, B2, sB, p, pl, cflag, pflag, gbn, gstride);
  SF_LAUNCH_CHECK("k_tri_eig");
  return 0;
}

// after the Cholesky (gv half 0 = L): Z -> U0 S^-1 -> W -> W' -> F -> gv half 0 where everything stayed finite
int sf_launch_wtri_apply(double *gv, int p, int p2, int nb, double *B2, double *B3, double *small, int32_t *cflag,
                         int32_t *pflag, hipStream_t st) {
  const size_t sG = (size_t)2 * p2 * p2, sB = (size_t)p * p;
  const int pl = (p + 15) & ~15;
  const size_t sS = (size_t)4 * pl;
  double *A1 = gv + (size_t)p2 * p2;
  const int nt = sf_cdiv(p, 32);
  hipLaunchKernelGGL(k_tri_transpose, dim3(nt, nt, nb), dim3(256), 0, st, B3, sB, p, B2, sB, p, p, small, sS, pl, cflag, pflag);
  SF_LAUNCH_CHECK("k_tri_transpose");
#define TBK_GO(RM)                                                                                                              \
  {                                                                                                                             \
    const size_t lds3 = ((size_t)2 * TBK_CH * 16 * RM + 2 * TBK_CH) * sizeof(double);                                           \
    if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_tri_back<RM>), lds3)) return rc;                                  \
    hipLaunchKernelGGL(k_tri_back<RM>, dim3(sf_cdiv(p, 32), nb), dim3(TBK_NT), lds3, st, A1, sG, p2, B2, sB, p, p, cflag, pflag); \
  }
  const int rm = sf_cdiv(p, 16);
  if (rm <= 12) TBK_GO(12) else if (rm <= 16) TBK_GO(16) else if (rm <= 20) TBK_GO(20) else if (rm <= 24) TBK_GO(24)
  else if (rm <= 27) TBK_GO(27) else TBK_GO(32)
#undef TBK_GO
  SF_LAUNCH_CHECK("k_tri_back");
  // W (B3) = L^T U0s
  if (int rc = sf_wide_dgemm(B2, p, sB, gv, p2, sG, 1, B3, p, sB, p, nb, cflag, pflag, st, 1)) return rc;
  // G (A1) = W^T W, M = 1.5 I - 0.5 G
  if (int rc = sf_wide_dgemm(B3, p, sB, B3, p, sB, 1, A1, p2, sG, p, nb, cflag, pflag, st)) return rc;
  hipLaunchKernelGGL(k_tri_nsm, dim3(64, nb), dim3(256), 0, st, A1, sG, p2, p, cflag, pflag);
  SF_LAUNCH_CHECK("k_tri_nsm");
  // W' (B2) = W M;  F (B3) = L W'
  if (int rc = sf_wide_dgemm(A1, p2, sG, B3, p, sB, 0, B2, p, sB, p, nb, cflag, pflag, st)) return rc;
  if (int rc = sf_wide_dgemm(B2, p, sB, gv, p2, sG, 0, B3, p, sB, p, nb, cflag, pflag, st, 1)) return rc;
  hipLaunchKernelGGL(k_tri_check, dim3(64, nb), dim3(256), 0, st, B3, sB, p, p, cflag, pflag);
  hipLaunchKernelGGL(k_tri_select, dim3(64, nb), dim3(256), 0, st, B3, sB, p, gv, sG, p2, p, cflag, pflag);
  if (sf_tune().wide_eigh_variant == 8) SF_HIP(hipMemsetAsync(pflag, 0x01, (size_t)nb * sizeof(int32_t), st));   // (test: every preconditioner refused)
  hipLaunchKernelGGL(k_tri_demote, dim3(sf_cdiv(nb, 256)), dim3(256), 0, st, nb, cflag, pflag);
  SF_LAUNCH_CHECK("k_tri_select");
  return 0;
}

namespace {
// debug helper: column-major n x n (ld) <- row-major / column-major source [n][n] (symmetric R: the same; L: given column-major)
__global__ void k_tri_load(const double *__restrict__ src, int n, double *__restrict__ dst, size_t sD, int ld) {
  const int mtx = blockIdx.y;
  const double *s = src + (size_t)mtx * n * n;
  double *d = dst + (size_t)mtx * sD;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * n; i += gridDim.x * blockDim.x) {
    const int c = i / n, r = i - c * n;
    d[(size_t)c * ld + r] = s[i];
  }
}
__global__ void k_tri_store(const double *__restrict__ src, size_t sS, int ld, int n, double *__restrict__ dst) {
  const int mtx = blockIdx.y;
  const double *s = src + (size_t)mtx * sS;
  double *d = dst + (size_t)mtx * n * n;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * n; i += gridDim.x * blockDim.x) {
    const int c = i / n, r = i - c * n;
    d[i] = s[(size_t)c * ld + r];
  }
}
__global__ void k_tri_store_small(const double *__restrict__ small, size_t sS, int off, int n, double *__restrict__ dst) {
  const int mtx = blockIdx.x;
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[(size_t)mtx * n + i] = small[(size_t)mtx * sS + off + i];
}
}  // namespace

extern "C" int sf_debug_wtri_stamps(unsigned long long *out8, int reset) {
  if (out8) SF_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_tr_stamps), 8 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    SF_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_tr_stamps), z, sizeof(z)));
  }
  return 0;
}
extern "C" {
/* Test entry of the tridiagonal preconditioner (tests/test_cmf_gpu.py): R [nb][p][p] symmetric positive definite, Lc [nb][p][p] its
 * lower Cholesky factor in COLUMN-major order (zeros above the diagonal); F [nb][p][p] column-major <- the preconditioned factor
 * (F F^T = R, nearly orthogonal columns), tlam [nb][p] <- the tridiagonal route's eigenvalues (ascending), pflag [nb] <- 0 where it
 * was applied.  scratch >= sf_debug_wtri_scratch_bytes(p, nb). */
size_t sf_debug_wtri_scratch_bytes(int p, int nb) {
  const size_t p2 = p + (p & 1);
  return sf_align((size_t)nb * 2 * p2 * p2 * sizeof(double)) + 2 * sf_align((size_t)nb * p * p * sizeof(double)) +
         sf_wtri_small_bytes(p, nb) + sf_align((size_t)2 * nb * sizeof(int32_t));
}
int sf_debug_wtri(const double *R, const double *Lc, int p, int nb, double *F, double *tlam, int32_t *pflag_out, void *scratch,
                  void *stream) {
  hipStream_t st = (hipStream_t)stream;
  const int p2 = p + (p & 1), pl = (p + 15) & ~15;
  char *b = reinterpret_cast<char *>(scratch);
  double *gv = reinterpret_cast<double *>(b); b += sf_align((size_t)nb * 2 * p2 * p2 * sizeof(double));
  double *B2 = reinterpret_cast<double *>(b); b += sf_align((size_t)nb * p * p * sizeof(double));
  double *B3 = reinterpret_cast<double *>(b); b += sf_align((size_t)nb * p * p * sizeof(double));
  double *small = reinterpret_cast<double *>(b); b += sf_wtri_small_bytes(p, nb);
  int32_t *cflag = reinterpret_cast<int32_t *>(b), *pflag = cflag + nb;
  SF_HIP(hipMemsetAsync(gv, 0, (size_t)nb * 2 * p2 * p2 * sizeof(double), st));
  SF_HIP(hipMemsetAsync(cflag, 0, (size_t)2 * nb * sizeof(int32_t), st));
  const size_t sG = (size_t)2 * p2 * p2;
  hipLaunchKernelGGL(k_tri_load, dim3(64, nb), dim3(256), 0, st, R, p, gv, sG, p2);
  if (int rc = sf_launch_wtri_prepare(gv, p, p2, nb, B2, B3, small, cflag, pflag, st, 0, 0)) return rc;
  hipLaunchKernelGGL(k_tri_load, dim3(64, nb), dim3(256), 0, st, Lc, p, gv, sG, p2);
  if (int rc = sf_launch_wtri_apply(gv, p, p2, nb, B2, B3, small, cflag, pflag, st)) return rc;
  hipLaunchKernelGGL(k_tri_store, dim3(64, nb), dim3(256), 0, st, gv, sG, p2, p, F);
  hipLaunchKernelGGL(k_tri_store_small, dim3(nb), dim3(256), 0, st, small, (size_t)4 * pl, 2 * pl, p, tlam);
  SF_HIP(hipMemcpyAsync(pflag_out, pflag, (size_t)nb * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  SF_LAUNCH_CHECK("sf_debug_wtri");
  return 0;
}
}
