// Small dense linear algebra with the reference's LAPACK semantics: batched LU with partial pivoting in float64.
//
//  * det / inv of the Python surface (cmf/robust_mf.py:72-90: scipy.linalg.det / inv, check_finite=False);
//  * the EXACT over/underflow behaviour of the determinants inside looshrinkage (:111-113).  scipy's det is getrf
//    followed by the running product of the LU diagonal in index order: once a prefix of that product has reached inf
//    it stays inf (log -> inf: the alpha can never win), once it has reached 0 it stays 0 (the alpha is skipped) --
//    whatever the value of the full determinant.  The restated sweep (one eigendecomposition per column) knows the
//    TOTAL log-determinant only, which classifies an alpha wrongly when a prefix leaves the float64 range and the total
//    does not (1-2 grid points at p = 425, up to 16 at p = 512: SURVEY.md §7.3 item 3).  For windows wider than 96 bands
//    the grid points in question are factorised for real: G = n beta S' + alpha diag(S') (S' the covariance of the
//    data scaled by 100, :94-110), partial-pivot LU, running product.
// One workgroup per matrix, the matrix in global memory (a 425 x 425 float64 matrix is 1.4 MB: L2-resident), right-
// looking elimination.  Latency-bound (~2 ms per 425 x 425 matrix), used where it is the only exact way.
#include "cmf_common.h"
#include <algorithm>

namespace {

constexpr int LU_NT = 1024;

// A[n][n] row-major, overwritten by L \ U (unit lower).  piv[k] = row swapped with k.  Returns through shared state:
// first zero pivot (LAPACK info, 0 = none), number of swaps.  det (optional): the running product of U's diagonal in
// index order with the sign of the permutation applied at the end (scipy: find_det_from_lu; 0 when info > 0).
__device__ void lu_factor(double *A, int n, int *piv, double *det_out, int *info_out, int tid) {
  __shared__ double sval[LU_NT / 64];
  __shared__ int sidx[LU_NT / 64];
  __shared__ int s_p;
  __shared__ double s_piv;
  int info = 0, nswap = 0;
  double det = 1.0;
  for (int k = 0; k < n; ++k) {
    // ---- pivot: first row of max |A[i][k]|, i >= k (idamax)
    double best = -1.0;
    int bi = 0x7fffffff;
    for (int i = k + tid; i < n; i += LU_NT) {
      const double v = fabs(A[(size_t)i * n + k]);
      if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { sval[tid >> 6] = best; sidx[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
      double b = sval[0];
      int p = sidx[0];
      for (int w = 1; w < LU_NT / 64; ++w)
        if (sval[w] > b || (sval[w] == b && sidx[w] < p)) { b = sval[w]; p = sidx[w]; }
      if (p == 0x7fffffff) p = k;                 // (a column of NaNs: idamax returns the first row)
      s_p = p;
      s_piv = A[(size_t)p * n + k];
    }
    __syncthreads();
    const int p = s_p;
    const double pv = s_piv;
    if (piv && tid == 0) piv[k] = p;
    if (p != k) {
      ++nswap;
      for (int j = tid; j < n; j += LU_NT) {
        const double t = A[(size_t)k * n + j];
        A[(size_t)k * n + j] = A[(size_t)p * n + j];
        A[(size_t)p * n + j] = t;
      }
    }
    det *= pv;
    if (pv == 0.0) {
      if (!info) info = k + 1;                    // dgetf2: record, skip the elimination of this column
      __syncthreads();
      continue;
    }
    __syncthreads();
    const int m = n - k - 1;
    for (int i = tid; i < m; i += LU_NT) A[(size_t)(k + 1 + i) * n + k] /= pv;
    __syncthreads();
    // ---- trailing update: A[i][j] -= l_i * u_j, i, j > k (64 columns x 16 rows per pass, coalesced along j)
    for (int i = k + 1 + (tid >> 6); i < n; i += LU_NT / 64) {
      const double li = A[(size_t)i * n + k];
      for (int j = k + 1 + (tid & 63); j < n; j += 64) A[(size_t)i * n + j] -= li * A[(size_t)k * n + j];
    }
    __syncthreads();
  }
  if (tid == 0) {
    if (det_out) *det_out = info ? 0.0 : ((nswap & 1) ? -det : det);
    if (info_out) *info_out = info;
  }
}

// ---- determinant only, blocked: the workhorse of the exact-determinant pass ------------------------------------------
// lu_factor above re-reads and re-writes the whole trailing matrix for every column (410 MB of L2 traffic and 1275
// barriers for one 425 x 425 matrix: 11.6 ms per matrix and workgroup).  Here a panel of 16 columns is factorised in LDS
// (row-major [H][18]; the pivot search, the row swap and the pivot bookkeeping are done by wave 0 between two
// barriers per column), then every thread takes ONE column of the part right of the panel: it gathers the 16 pivot
// rows of its column (the net row permutation of the panel, all reads before the writes), solves the unit-lower 16 x 16
// system for its column of U in registers and applies the rank-16 update to its column, reading the L rows as LDS
// broadcasts.  Neither L nor the rows of U are written back: only the running product of the pivots is wanted.
// Same pivoting rule (first row of maximal magnitude) and the same left-to-right product as lu_factor.
// (A 425 x 425 factorisation alone on the chip, sf_debug_lu_stamps: 3.9 M cycles = panel 1.8 M -- 425 columns of pivot search by
// wave 0 + elimination between two barriers --, permutation + solve 0.8 M, update 1.3 M.  A panel form with a thread per row, the
// row in registers and a block-wide argmax was measured SLOWER (2.0 M) and dropped.)
// MF (n <= LB_MF_MAX: the 16 rows of U fit LDS beside the panel): the rank-16 update on the matrix cores.  As scalar code every
// multiply-add fetched a broadcast L value from LDS -- 16 per element: the update was LDS-issue-bound and the pass one LU per
// 1.5 ms and CU.  Here a thread still permutes and solves its column of U, puts it into LDS, and the waves then take 16 x 16 tiles
// of the trailing block with four v_mfma_f64_16x16x4 each (A operand -L, B operand U).
constexpr int LB_NT = 512, LB_NB = 16, LB_LD = 18, LB_MF_MAX = 560;
static size_t lb_lds_bytes(int n) {
  return (size_t)n * LB_LD * sizeof(double) + (n <= LB_MF_MAX ? (size_t)LB_NB * n * sizeof(double) : 0) + (size_t)n * sizeof(int) +
         64 * sizeof(int);
}
static bool lb_fits(int n) { return lb_lds_bytes(n) <= 150 * 1024; }

__device__ unsigned long long g_lu_stamps[8];   // phase clocks of workgroup 0's factorisations: panel load, panel, solve, update, count
// Off in production: the accumulation is a non-atomic read-modify-write of a device global, racing across the streams
// of the in-flight pipeline.  sf_debug_lu_stamps(out, 1) zeroes the clocks and turns them on, (out, 2) reads and turns them off.
__device__ int g_lu_stamps_on = 0;
template <bool MF>
__device__ void lu_det_blocked(double *__restrict__ A, int n, double *det_out, double *pan, int *rowof, int *plist,
                               double *Us = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int info = 0, nswap = 0;
  double det = 1.0;
  for (int k0 = 0; k0 < n; k0 += LB_NB) {
    const int nc = min(LB_NB, n - k0), H = n - k0, k1 = k0 + nc, rest = n - k1;
    const bool stw = blockIdx.x == 0 && tid == 0 && g_lu_stamps_on;
    unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    if (stw) q0 = __builtin_readcyclecounter();
    for (int e = tid; e < H * LB_NB; e += LB_NT) {
      const int r = e >> 4, c = e & 15;
      pan[r * LB_LD + c] = (c < nc) ? A[(size_t)(k0 + r) * n + k0 + c] : 0.0;
    }
    for (int r = tid; r < H; r += LB_NT) rowof[r] = r;
    __syncthreads();
    if (stw) q1 = __builtin_readcyclecounter();
    for (int j = 0; j < nc; ++j) {
      if (wave == 0) {
        double best = -1.0;
        int bi = 0x7fffffff;
        for (int r = j + lane; r < H; r += 64) {
          const double v = fabs(pan[r * LB_LD + j]);
          if (v > best || (v == best && r < bi)) { best = v; bi = r; }
        }
        for (int o = 32; o > 0; o >>= 1) {
          const double ov = __shfl_xor(best, o, 64);
          const int oi = __shfl_xor(bi, o, 64);
          if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        const int p = (bi == 0x7fffffff) ? j : bi;       // (a column of NaNs: idamax returns the first row)
        if (p != j) {
          if (lane < nc) {
            const double t = pan[j * LB_LD + lane];
            pan[j * LB_LD + lane] = pan[p * LB_LD + lane];
            pan[p * LB_LD + lane] = t;
          }
          if (lane == 0) { const int t = rowof[j]; rowof[j] = rowof[p]; rowof[p] = t; }
        }
        if (lane == 0) plist[j] = p;
      }
      __syncthreads();
      const double pv = pan[j * LB_LD + j];
      if (tid == 0) {
        det *= pv;
        if (plist[j] != j) ++nswap;
      }
      if (pv == 0.0) {                                   // dgetf2: record, no elimination with this column
        info = 1;
        __syncthreads();
        continue;
      }
      for (int r = j + 1 + tid; r < H; r += LB_NT) {
        const double l = pan[r * LB_LD + j] / pv;
        pan[r * LB_LD + j] = l;
        for (int c = j + 1; c < nc; ++c) pan[r * LB_LD + c] = __builtin_fma(-l, pan[j * LB_LD + c], pan[r * LB_LD + c]);
      }
      __syncthreads();
    }
    if (stw) q2 = __builtin_readcyclecounter();
    for (int jc = tid; jc < rest; jc += LB_NT) {
      double *col = A + (size_t)k0 * n + k1 + jc;        // column jc of the right part, panel row 0
      double u[LB_NB], moved[LB_NB];
#pragma unroll
      for (int t = 0; t < LB_NB; ++t) u[t] = (t < nc) ? col[(size_t)rowof[t] * n] : 0.0;
#pragma unroll
      for (int t = 0; t < LB_NB; ++t) {
        const int q = (t < nc) ? plist[t] : 0;
        moved[t] = (q >= nc) ? col[(size_t)rowof[q] * n] : 0.0;
      }
#pragma unroll
      for (int t = 0; t < LB_NB; ++t) {
        const int q = (t < nc) ? plist[t] : 0;
        if (q >= nc) col[(size_t)q * n] = moved[t];
      }
#pragma unroll
      for (int t = 1; t < LB_NB; ++t)
#pragma unroll
        for (int sx = 0; sx < t; ++sx)
          if (t < nc) u[t] = __builtin_fma(-pan[t * LB_LD + sx], u[sx], u[t]);
      if (MF) {
#pragma unroll
        for (int t = 0; t < LB_NB; ++t) Us[t * n + jc] = u[t];
        continue;
      }
      int r = nc;
      for (; r + 4 <= H; r += 4) {
        double a0 = col[(size_t)r * n], a1 = col[(size_t)(r + 1) * n], a2 = col[(size_t)(r + 2) * n], a3 = col[(size_t)(r + 3) * n];
#pragma unroll
        for (int t = 0; t < LB_NB; ++t) {
          a0 = __builtin_fma(-pan[r * LB_LD + t], u[t], a0);
          a1 = __builtin_fma(-pan[(r + 1) * LB_LD + t], u[t], a1);
          a2 = __builtin_fma(-pan[(r + 2) * LB_LD + t], u[t], a2);
          a3 = __builtin_fma(-pan[(r + 3) * LB_LD + t], u[t], a3);
        }
        col[(size_t)r * n] = a0; col[(size_t)(r + 1) * n] = a1; col[(size_t)(r + 2) * n] = a2; col[(size_t)(r + 3) * n] = a3;
      }
      for (; r < H; ++r) {
        double a0 = col[(size_t)r * n];
#pragma unroll
        for (int t = 0; t < LB_NB; ++t) a0 = __builtin_fma(-pan[r * LB_LD + t], u[t], a0);
        col[(size_t)r * n] = a0;
      }
    }
    if (MF && rest > 0 && H > nc) {
      __syncthreads();   // U's rows are in LDS, the right part's rows are permuted
      if (stw) q3 = __builtin_readcyclecounter();
      typedef double d4_t __attribute__((ext_vector_type(4)));
      const int g = lane >> 4, li = lane & 15;
      const int nrt = (H - nc + 15) >> 4, nct = (rest + 15) >> 4;
      for (int q = wave; q < nrt * nct; q += LB_NT / 64) {
        const int tr = q / nct, tc = q - tr * nct;
        const int R0 = nc + 16 * tr, C0 = 16 * tc, cc = C0 + li;
        double *ap = A + (size_t)(k0 + R0 + g) * n + k1 + cc;
        d4_t acc;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = (R0 + g + 4 * e < H && cc < rest) ? ap[(size_t)4 * e * n] : 0.0;
        const double *lp = pan + (size_t)min(R0 + li, H - 1) * LB_LD + g;
        const double *up = Us + (size_t)g * n + min(cc, rest - 1);
#pragma unroll
        for (int kk = 0; kk < LB_NB; kk += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-lp[kk], up[(size_t)kk * n], acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (R0 + g + 4 * e < H && cc < rest) ap[(size_t)4 * e * n] = acc[e];
      }
    }
    __syncthreads();
    if (stw) {
      const unsigned long long q4 = __builtin_readcyclecounter();
      g_lu_stamps[0] += q1 - q0; g_lu_stamps[1] += q2 - q1; g_lu_stamps[2] += (q3 ? q3 : q4) - q2; g_lu_stamps[3] += q3 ? q4 - q3 : 0;
    }
  }
  if (blockIdx.x == 0 && tid == 0 && g_lu_stamps_on) g_lu_stamps[4] += 1;
  if (tid == 0 && det_out) *det_out = info ? 0.0 : ((nswap & 1) ? -det : det);
}

__global__ __launch_bounds__(LU_NT) void k_lu_det(const double *__restrict__ A, int n, double *__restrict__ work,
                                                  double *__restrict__ det, int32_t *__restrict__ info) {
  const int b = blockIdx.x, tid = threadIdx.x;
  double *W = work + (size_t)b * n * n;
  for (size_t e = tid; e < (size_t)n * n; e += LU_NT) W[e] = A[(size_t)b * n * n + e];
  __syncthreads();
  lu_factor(W, n, nullptr, det + b, info ? info + b : nullptr, tid);
}

__global__ __launch_bounds__(LB_NT) void k_lu_det_blocked(const double *__restrict__ A, int n, double *__restrict__ work,
                                                          double *__restrict__ det) {
  extern __shared__ __attribute__((aligned(16))) double lb_lds[];
  const bool mf = n <= LB_MF_MAX;
  double *Us = lb_lds + (size_t)n * LB_LD;
  int *rowof = reinterpret_cast<int *>(Us + (mf ? (size_t)LB_NB * n : 0));
  const int b = blockIdx.x, tid = threadIdx.x;
  double *W = work + (size_t)b * n * n;
  for (size_t e = tid; e < (size_t)n * n; e += LB_NT) W[e] = A[(size_t)b * n * n + e];
  __syncthreads();
  if (mf) lu_det_blocked<true>(W, n, det + b, lb_lds, rowof, rowof + n, Us);
  else lu_det_blocked<false>(W, n, det + b, lb_lds, rowof, rowof + n);
}

// inverse from the factors: solve (P A) X = P I column by column (forward with unit L, backward with U)
__global__ __launch_bounds__(LU_NT) void k_lu_inv(const double *__restrict__ A, int n, double *__restrict__ work,
                                                  int32_t *__restrict__ pivs, double *__restrict__ Ainv,
                                                  int32_t *__restrict__ info) {
  __shared__ int s_info;
  const int b = blockIdx.x, tid = threadIdx.x;
  double *W = work + (size_t)b * n * n;
  int *piv = pivs + (size_t)b * n;
  for (size_t e = tid; e < (size_t)n * n; e += LU_NT) W[e] = A[(size_t)b * n * n + e];
  __syncthreads();
  lu_factor(W, n, piv, nullptr, &s_info, tid);
  __syncthreads();
  if (tid == 0 && info) info[b] = s_info;
  if (s_info) return;                              // singular: scipy raises LinAlgError
  double *X = Ainv + (size_t)b * n * n;
  // X = P (identity rows permuted as the factorisation did)
  for (size_t e = tid; e < (size_t)n * n; e += LU_NT) X[e] = ((e / n) == (e % n)) ? 1.0 : 0.0;
  __syncthreads();
  for (int k = 0; k < n; ++k) {
    const int p = piv[k];
    if (p != k)
      for (int j = tid; j < n; j += LU_NT) {
        const double t = X[(size_t)k * n + j];
        X[(size_t)k * n + j] = X[(size_t)p * n + j];
        X[(size_t)p * n + j] = t;
      }
    __syncthreads();
  }
  // forward: L Y = P, row by row (all right-hand sides j in parallel)
  for (int i = 1; i < n; ++i) {
    for (int j = tid; j < n; j += LU_NT) {
      double s = X[(size_t)i * n + j];
      for (int k = 0; k < i; ++k) s -= W[(size_t)i * n + k] * X[(size_t)k * n + j];
      X[(size_t)i * n + j] = s;
    }
    __syncthreads();
  }
  // backward: U X = Y
  for (int i = n - 1; i >= 0; --i) {
    for (int j = tid; j < n; j += LU_NT) {
      double s = X[(size_t)i * n + j];
      for (int k = i + 1; k < n; ++k) s -= W[(size_t)i * n + k] * X[(size_t)k * n + j];
      X[(size_t)i * n + j] = s / W[(size_t)i * n + i];
    }
    __syncthreads();
  }
}

// ---- exact determinants of the shrinkage grid ---------------------------------------------------------------------
// job list: (column, alpha index) pairs whose determinant is factorised for real
//   full   every alpha of every column (function-level looshrinkage: one column)
//   window the finite grid points within W points of one where the TOTAL log-determinant (rule (i)) has left the
//          float64 range
__global__ void k_det_jobs(const double *__restrict__ nll, const double *__restrict__ rest, const int32_t *__restrict__ status,
                           int ncols, int nalpha, int window, int32_t *__restrict__ jobs, int32_t *__restrict__ njobs, int cap) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols || status[c] != 0) return;
  auto push = [&](int i) {
    const int k = atomicAdd(njobs, 1);
    if (k < cap) jobs[k] = c * nalpha + i;
  };
  if (window <= 0) {
    for (int i = 0; i < nalpha; ++i) push(i);
    return;
  }
  // rest is finite wherever the sweep is; nll = +inf there means the TOTAL log-determinant left the float64 range
  // (rule (i)).  A prefix of the pivot product can leave it a few grid points earlier: every grid point that is still
  // finite and lies within `window` points of a lost one is factorised.
  for (int i = 0; i < nalpha; ++i) {
    const double v = nll[(size_t)c * nalpha + i];
    if (v == __builtin_inf() || v != v) continue;
    bool near = false;
    for (int j = max(0, i - window); j <= min(nalpha - 1, i + window) && !near; ++j) {
      const double vj = nll[(size_t)c * nalpha + j], rj = rest ? rest[(size_t)c * nalpha + j] : 0.0;
      near = (vj == __builtin_inf()) && (rj == rj) && (rj != __builtin_inf());
    }
    if (near) push(i);
  }
}

// The windowed rule in rounds ("deepening"): the grid points whose pivot product leaves the range although the total does
// not sit next to the crossing -- one grid step moves log det by up to p log(10^0.05) (49 at p = 425), the excess of the
// largest prefix over the total changes slowly with alpha.  So the finite points are factorised in groups of `group`,
// walking away from each crossing: round r takes the next `group` points of every side that is still open; a side closes
// when a whole group came back unchanged (or when the walk has covered `window` points or met the other side).  On the
// benchmark flightline this factorises ~8 points per column instead of 48.  Columns whose lost points are not a run at
// the start and / or a run at the end of the grid take the plain window rule in round 0.
// state[c] = { next candidate of the low side (ascending), of the high side (descending), low side open, high side open }
__global__ void k_det_round(const double *__restrict__ nll, const double *__restrict__ rest, const int32_t *__restrict__ status,
                            int ncols, int nalpha, int window, int group, int round, int32_t *__restrict__ jobs,
                            int32_t *__restrict__ njobs, int cap, int32_t *__restrict__ state) {
  // one wave per column: the common case -- no lost grid point at all (round 0), both sides closed (later rounds) -- is
  // settled by one coalesced pass of the 64 lanes; only a column with work to do goes on, on lane 0
  const int c = blockIdx.x;
  if (c >= ncols || status[c] != 0) return;
  const double *v = nll + (size_t)c * nalpha, *rs = rest ? rest + (size_t)c * nalpha : nullptr;
  const double inf = __builtin_inf();
  if (round == 0) {
    bool any = false;
    for (int i = threadIdx.x; i < nalpha; i += 64) any = any || (v[i] == inf);
    if (!__any(any)) {
      if (threadIdx.x == 0) { int32_t *s0 = state + 4 * (size_t)c; s0[0] = s0[1] = s0[2] = s0[3] = 0; }
      return;
    }
  } else if (state[4 * (size_t)c + 2] == 0 && state[4 * (size_t)c + 3] == 0) {
    return;
  }
  if (threadIdx.x != 0) return;
  auto push = [&](int i) {
    const int k = atomicAdd(njobs, 1);
    if (k < cap) jobs[k] = c * nalpha + i;
  };
  auto lost = [&](int i) { return v[i] == inf && (!rs || (rs[i] == rs[i] && rs[i] != inf)); };
  auto finite = [&](int i) { return v[i] == v[i] && v[i] != inf && v[i] != -inf; };
  int32_t *s = state + 4 * (size_t)c;
  if (round == 0) {
    int lo = 0;
    while (lo < nalpha && lost(lo)) ++lo;
    int hi = nalpha;
    while (hi > lo && lost(hi - 1)) --hi;
    bool interior = false;
    for (int i = lo; i < hi; ++i) interior = interior || lost(i);
    if (interior) {
      for (int i = 0; i < nalpha; ++i) {
        if (!finite(i)) continue;
        bool near = false;
        for (int j = max(0, i - window); j <= min(nalpha - 1, i + window) && !near; ++j) near = lost(j);
        if (near) push(i);
      }
      s[0] = s[1] = s[2] = s[3] = 0;
      return;
    }
    s[0] = lo;
    s[1] = hi - 1;
    s[2] = lo > 0;
    s[3] = hi < nalpha;
  } else {
    if (s[2]) {                                    // the low side's last group: s[0]-group .. s[0]-1
      bool changed = false;
      for (int i = max(0, s[0] - group); i < s[0]; ++i) changed = changed || !finite(i);
      if (!changed) s[2] = 0;
    }
    if (s[3]) {
      bool changed = false;
      for (int i = s[1] + 1; i <= min(nalpha - 1, s[1] + group); ++i) changed = changed || !finite(i);
      if (!changed) s[3] = 0;
    }
  }
  if (s[2])
    for (int k = 0; k < group; ++k) {
      const int i = s[0];
      if (i > s[1]) { s[2] = 0; break; }
      s[0] = i + 1;
      if (finite(i)) push(i);
    }
  if (s[3])
    for (int k = 0; k < group; ++k) {
      const int i = s[1];
      if (i < s[0]) { s[3] = 0; break; }
      s[1] = i - 1;
      if (finite(i)) push(i);
    }
}

// G = n (beta S') + alpha T, S' = 1e4 S (the covariance of 100 x), T = diag S' (robust_mf.py:94-110), then LU
__global__ __launch_bounds__(LU_NT) void k_det_grid(const double *__restrict__ cov, const int32_t *__restrict__ nloo,
                                                    const double *__restrict__ alphas, int nalpha, int p,
                                                    const int32_t *__restrict__ jobs, const int32_t *__restrict__ njobs,
                                                    int job0, double *__restrict__ work, double *__restrict__ det,
                                                    const double *__restrict__ target) {
  const int slot = blockIdx.x, tid = threadIdx.x;
  const int nj = min(*njobs, job0);                  // (job0 carries the capacity of the job list)
  double *G = work + (size_t)slot * p * p;
  for (int jb = slot; jb < nj; jb += gridDim.x) {    // one launch per round: a workgroup walks the job list
    const int job = jobs[jb], c = job / nalpha, ai = job - c * nalpha;
    const double n = (double)nloo[c], a = alphas[ai];
    const double beta = (1.0 - a) / (n - 1.0);
    const double *S = cov + (size_t)c * p * p;
    const double *T = target ? target + (size_t)c * p * p : nullptr;   // full shrinkage target (:99), else diag(S)
    for (size_t e = tid; e < (size_t)p * p; e += LU_NT) {
      const int i = (int)(e / p), j = (int)(e % p);
      const double s = S[e] * 1e4;
      double gij = n * (beta * s);
      if (T) gij += a * (T[e] * 1e4);
      else if (i == j) gij += a * s;
      G[e] = gij;
    }
    __syncthreads();
    lu_factor(G, p, nullptr, det + jb, nullptr, tid);
    __syncthreads();
  }
}
// the same through the blocked factorisation (p <= 1000: the panel fits LDS)
__global__ __launch_bounds__(LB_NT) void k_det_grid_blocked(const double *__restrict__ cov, const int32_t *__restrict__ nloo,
                                                            const double *__restrict__ alphas, int nalpha, int p,
                                                            const int32_t *__restrict__ jobs, const int32_t *__restrict__ njobs,
                                                            int job0, double *__restrict__ work, double *__restrict__ det,
                                                            const double *__restrict__ target) {
  extern __shared__ __attribute__((aligned(16))) double lb_lds[];
  const bool mf = p <= LB_MF_MAX;
  double *Us = lb_lds + (size_t)p * LB_LD;
  int *rowof = reinterpret_cast<int *>(Us + (mf ? (size_t)LB_NB * p : 0));
  const int slot = blockIdx.x, tid = threadIdx.x;
  const int nj = min(*njobs, job0);                  // (job0 carries the capacity of the job list)
  double *G = work + (size_t)slot * p * p;
  for (int jb = slot; jb < nj; jb += gridDim.x) {    // one launch per round: a workgroup walks the job list
    const int job = jobs[jb], c = job / nalpha, ai = job - c * nalpha;
    const double n = (double)nloo[c], a = alphas[ai];
    const double beta = (1.0 - a) / (n - 1.0);
    const double *S = cov + (size_t)c * p * p;
    const double *T = target ? target + (size_t)c * p * p : nullptr;
    for (size_t e = tid; e < (size_t)p * p; e += LB_NT) {
      const int i = (int)(e / p), j = (int)(e % p);
      const double s = S[e] * 1e4;
      double gij = n * (beta * s);
      if (T) gij += a * (T[e] * 1e4);
      else if (i == j) gij += a * s;
      G[e] = gij;
    }
    __syncthreads();
    if (mf) lu_det_blocked<true>(G, p, det + jb, lb_lds, rowof, rowof + p, Us);
    else lu_det_blocked<false>(G, p, det + jb, lb_lds, rowof, rowof + p);
    __syncthreads();
  }
}

// nll[job] = 0.5 log(det) + rest, +inf where det == 0 (the reference skips the alpha, :112-113); then numpy.argmin
__global__ void k_det_apply(const int32_t *__restrict__ jobs, const int32_t *__restrict__ njobs, const double *__restrict__ det,
                            const double *__restrict__ rest, double *__restrict__ nll, int cap) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= *njobs || k >= cap) return;
  const int job = jobs[k];
  const double dt = det[k];
  if (!rest) {      // no split of the NLL at hand (narrow windows): the factorisation only decides finite / lost
    if (dt == 0.0) nll[job] = __builtin_inf();
    else if (!(dt > 0.0 && dt < __builtin_inf())) nll[job] = 0.5 * log(dt);     // +inf -> inf, -inf / NaN -> NaN
    return;
  }
  nll[job] = (dt == 0.0) ? __builtin_inf() : 0.5 * log(dt) + rest[job];
}
__global__ void k_argmin_nan_first(const double *__restrict__ nll, const int32_t *__restrict__ status, int ncols, int nalpha,
                                   int32_t *__restrict__ alphaidx) {
  // numpy.argmin: the first NaN if there is one, else the first occurrence of the minimum; -1 when every entry is +inf
  // (robust_mf.py:119-125).  One wave per column, coalesced loads, the (value, index) pairs reduced with shuffles.
  const int c = blockIdx.x, lane = threadIdx.x;
  if (c >= ncols || status[c] != 0) return;
  int nanidx = 0x7fffffff, idx = 0x7fffffff;
  double best = __builtin_inf();
  for (int i = lane; i < nalpha; i += 64) {
    const double v = nll[(size_t)c * nalpha + i];
    if (v != v) { if (i < nanidx) nanidx = i; }
    else if (v < best) { best = v; idx = i; }           // (ascending i per lane: the first occurrence stays)
  }
  for (int o = 32; o > 0; o >>= 1) {
    const int on = __shfl_xor(nanidx, o, 64), oi = __shfl_xor(idx, o, 64);
    const double ob = __shfl_xor(best, o, 64);
    nanidx = min(nanidx, on);
    if (ob < best || (ob == best && oi < idx)) { best = ob; idx = oi; }
  }
  if (lane == 0) alphaidx[c] = nanidx != 0x7fffffff ? nanidx : (best < __builtin_inf() ? idx : -1);
}

}  // namespace

constexpr int DET_SLOTS = 512;   // matrices factorised per launch (two rounds of 256 CUs) ...
constexpr size_t DET_SLOT_BYTES = (size_t)1 << 30;     // ... within this many bytes of p x p float64 work matrices (p = 425: all 512, p = 512: 512;
                                                       // 290 slots at p = 425 cost the pass 13.4 instead of 9.1 ms per launch: the workgroups pair up on a CU)
constexpr int DET_GROUP = 4;     // grid points per side and round of the deepening rule (k_det_round), windows of <= 96 bands
constexpr int DET_GROUP_WIDE = 2;   // wide windows (an LU of 425 x 425 is 3 ms of a workgroup): the misclassified points are a run of
                                    // 1-2 next to the crossing; a side closes after two unchanged points, in twice as many (mostly empty) rounds
// The job list holds every grid point of every column whatever the window: a column whose lost points are not two runs at
// the ends of the grid takes the plain window rule in round 0 and may push up to nalpha jobs (with ncols x 2 x
// window entries such a column overflowed the list and which jobs survived depended on the order of the atomics).  An int
// and a double per entry: 2.4 KB per column.  The p x p work matrices are what costs memory; the job loop strides over them.
static size_t det_maxjobs(const SfGeom &g) { return (size_t)g.ncols * g.nalpha; }
static size_t det_slots(const SfGeom &g, size_t maxjobs) {
  size_t slots = DET_SLOT_BYTES / ((size_t)g.p * g.p * sizeof(double));
  slots = slots < 64 ? 64 : (slots > DET_SLOTS ? DET_SLOTS : slots);
  return maxjobs < slots ? maxjobs : slots;
}
size_t sf_exact_det_scratch_bytes(const SfGeom &g, int window) {
  (void)window;
  const size_t maxjobs = det_maxjobs(g);
  const size_t slots = det_slots(g, maxjobs);
  return sf_align(slots * g.p * g.p * sizeof(double)) + sf_align(maxjobs * sizeof(int32_t)) + sf_align(maxjobs * sizeof(double)) +
         sf_align(sizeof(int32_t)) + sf_align((size_t)g.ncols * 4 * sizeof(int32_t));
}
// window <= 0: every grid point; else the `window` points on the safe side of each range crossing (see k_det_jobs)
int sf_launch_exact_det(const double *cov, const int32_t *nloo, const int32_t *status, const double *alphas, const SfGeom &g,
                        int window, const double *rest, double *nll, int32_t *alphaidx, void *scratch, hipStream_t st,
                        const double *target) {
  const size_t maxjobs = det_maxjobs(g);
  const size_t slots_alloc = det_slots(g, maxjobs);
  const size_t slots = sf_tune().det_slots > 0 ? std::min(slots_alloc, (size_t)sf_tune().det_slots) : slots_alloc;
  char *p = reinterpret_cast<char *>(scratch);
  double *work = reinterpret_cast<double *>(p); p += sf_align(slots_alloc * g.p * g.p * sizeof(double));
  int32_t *jobs = reinterpret_cast<int32_t *>(p); p += sf_align(maxjobs * sizeof(int32_t));
  double *det = reinterpret_cast<double *>(p); p += sf_align(maxjobs * sizeof(double));
  int32_t *njobs = reinterpret_cast<int32_t *>(p); p += sf_align(sizeof(int32_t));
  int32_t *state = reinterpret_cast<int32_t *>(p);
  const bool blocked = lb_fits(g.p) && sf_tune().lu_variant == 0;
  if (blocked)
    if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_det_grid_blocked), lb_lds_bytes(g.p))) return rc;
  // window > 0: rounds of DET_GROUP points per open side (k_det_round); det_variant 1 = the plain window rule in one round
  const bool rounds = window > 0 && sf_tune().det_variant == 0;
  const int group = g.p > SF_MAX_ACTIVE_FUSED ? DET_GROUP_WIDE : DET_GROUP;
  const int nrounds = rounds ? sf_cdiv(window, group) : 1;
  for (int r = 0; r < nrounds; ++r) {
    // round 0 may hold every grid point of a plain-window column, later rounds at most 2 x DET_GROUP per column
    const size_t cap = (r == 0) ? maxjobs : std::min(maxjobs, (size_t)g.ncols * 2 * group);
    SF_HIP(hipMemsetAsync(njobs, 0, sizeof(int32_t), st));
    if (rounds)
      hipLaunchKernelGGL(k_det_round, dim3(g.ncols), dim3(64), 0, st, nll, rest, status, g.ncols, g.nalpha, window,
                         group, r, jobs, njobs, (int)maxjobs, state);
    else
      hipLaunchKernelGGL(k_det_jobs, dim3(sf_cdiv(g.ncols, 128)), dim3(128), 0, st, nll, rest, status, g.ncols, g.nalpha, window, jobs,
                         njobs, (int)maxjobs);
    SF_LAUNCH_CHECK("k_det_jobs");
    if (blocked)
      hipLaunchKernelGGL(k_det_grid_blocked, dim3((unsigned)slots), dim3(LB_NT), lb_lds_bytes(g.p), st, cov, nloo, alphas,
                         g.nalpha, g.p, jobs, njobs, (int)cap, work, det, target);
    else
      hipLaunchKernelGGL(k_det_grid, dim3((unsigned)slots), dim3(LU_NT), 0, st, cov, nloo, alphas, g.nalpha, g.p, jobs, njobs,
                         (int)cap, work, det, target);
    SF_LAUNCH_CHECK("k_det_grid");
    hipLaunchKernelGGL(k_det_apply, dim3(sf_cdiv((int)cap, 256)), dim3(256), 0, st, jobs, njobs, det, rest, nll, (int)cap);
    SF_LAUNCH_CHECK("k_det_apply");
  }
  hipLaunchKernelGGL(k_argmin_nan_first, dim3(g.ncols), dim3(64), 0, st, nll, status, g.ncols, g.nalpha, alphaidx);
  SF_LAUNCH_CHECK("k_argmin_nan_first");
  return 0;
}

extern "C" int sf_debug_lu_stamps(unsigned long long *out8, int reset) {
  if (out8) SF_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_lu_stamps), 8 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int on = reset == 1 ? 1 : 0;
    SF_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_lu_stamps), z, sizeof(z)));
    SF_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_lu_stamps_on), &on, sizeof(on)));
  }
  return 0;
}

extern "C" {

/* The exact-determinant pass as its own entry (windows of up to 96 bands: the fused stage-5 kernels keep no split of the
 * NLL into log det + rest): the grid points next to one whose total log-determinant left the float64 range -- window > 0:
 * at most `window` per crossing, in rounds of four; window <= 0: every grid point -- are factorised for real
 * (G = n beta 1e4 S + alpha 1e4 T, T = target or diag S) and those whose running pivot product is 0 or not finite
 * (robust_mf.py:111-113) are marked in nll; alphaidx is recomputed with numpy.argmin's rule.
 * scratch >= sf_cmf_exact_det_scratch_bytes(p, ncols, nalpha, window). */
size_t sf_cmf_exact_det_scratch_bytes(int p, int ncols, int nalpha, int window) {
  return sf_exact_det_scratch_bytes(sf_geom(1, p, ncols, nalpha), window);
}
int sf_cmf_exact_det(const double *cov, const double *target, const int32_t *nloo, const int32_t *status, const double *alphas,
                     int nalpha, int p, int ncols, int window, double *nll, int32_t *alphaidx, void *scratch, void *stream) {
  if (!cov || !nloo || !status || !alphas || !nll || !alphaidx || !scratch || p < 1 || ncols < 1 || nalpha < 1) {
    sf_set_error("sf_cmf_exact_det: bad argument");
    return -1;
  }
  return sf_launch_exact_det(cov, nloo, status, alphas, sf_geom(1, p, ncols, nalpha), window, nullptr, nll, alphaidx, scratch,
                             (hipStream_t)stream, target);
}

/* scipy.linalg.det semantics (cmf/robust_mf.py:86-90) for a batch of n x n float64 matrices: LU with partial pivoting,
 * the running product of the diagonal in index order (prefixes that reach inf or 0 stay there), 0 for an exactly
 * singular matrix.  work: batch * n * n doubles. */
int sf_linalg_det(const double *A, int n, int batch, double *work, double *det, void *stream) {
  if (!A || !work || !det || n < 1 || batch < 1) { sf_set_error("sf_linalg_det: bad argument"); return -1; }
  if (lb_fits(n) && sf_tune().lu_variant == 0) {
    if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_lu_det_blocked), lb_lds_bytes(n))) return rc;
    hipLaunchKernelGGL(k_lu_det_blocked, dim3(batch), dim3(LB_NT), lb_lds_bytes(n), (hipStream_t)stream, A, n, work, det);
  } else {
    hipLaunchKernelGGL(k_lu_det, dim3(batch), dim3(LU_NT), 0, (hipStream_t)stream, A, n, work, det, nullptr);
  }
  SF_LAUNCH_CHECK("k_lu_det");
  return 0;
}
/* scipy.linalg.inv semantics (cmf/robust_mf.py:72-76): LU with partial pivoting + solves; info[b] > 0 = exactly
 * singular (the reference catches LinAlgError, :371).  work: batch * n * n doubles, piv: batch * n int32. */
int sf_linalg_inv(const double *A, int n, int batch, double *work, int32_t *piv, double *Ainv, int32_t *info, void *stream) {
  if (!A || !work || !piv || !Ainv || !info || n < 1 || batch < 1) { sf_set_error("sf_linalg_inv: bad argument"); return -1; }
  hipLaunchKernelGGL(k_lu_inv, dim3(batch), dim3(LU_NT), 0, (hipStream_t)stream, A, n, work, piv, Ainv, info);
  SF_LAUNCH_CHECK("k_lu_inv");
  return 0;
}

}  // extern "C"
