// Stage 4: batched symmetric eigendecomposition of the per-column correlation matrix.
//
// Why it exists: looshrinkage (cmf/robust_mf.py:105-117) inverts G_a = n*beta*S + a*diag(S) and takes its
// determinant for each of the 201 alphas.  With d = sqrt(diag S) and R = S/(d d^T) = V diag(lam) V^T,
//   G_a = D (n*beta*R + a*I) D,  so  x^T G_a^-1 x = sum_j y_j^2 / (n*beta*lam_j + a),  y = V^T D^-1 x,
//   log det G_a = 2 sum log d_j + sum_j log(n*beta*lam_j + a),
// i.e. ONE eigendecomposition per column replaces 201 LU factorisations + inversions.
//
// Method: one-sided (Hestenes) Jacobi, one workgroup per column, 8 lanes per column pair (dot products reduced
// with DPP lane swaps), round-robin pair ordering so the p/2 rotations of a step touch disjoint columns (one
// LDS-only barrier per step), ONE p x p matrix in LDS (41 KB at p = 72 -> three columns per CU).
//   * Normal path: Cholesky R = L L^T in LDS, then Jacobi on the columns of G = L.  At convergence
//     G = U diag(sigma): lam_j = |g_j|^2 and the eigenvector is g_j/|g_j| -- no eigenvector accumulation at all,
//     ~10 sweeps, and small eigenvalues keep high relative accuracy (Demmel-Veselic).
//   * Fallback (R not positive definite, e.g. fewer valid rows than bands): Jacobi on G = R, every (cos, sin)
//     recorded in a global scratch list and replayed on V = I in the same LDS buffer.
// The kernel is latency-bound: ~70 steps per sweep, each ~200 dependent fp64 instructions of a single wave.
#include "cmf_common.h"

namespace {

constexpr int EIG_MAXSWEEP = 30;
// -DSF_EIGH_STAMPS: thread 0 of a workgroup sums the s_memtime deltas of the phases of a Jacobi step (diagnostic build only;
// the sums go to the unused rotation log of the column, tools/probe_eigh.py --stamps prints them).  Round 3, one 72 x 72
// matrix alone on the chip, 710 steps: 2220 cycles per step = operands landed 610 + dot product reduced 260 + rotation
// parameters 270 + rotation and stores issued 700 + barrier 270 + sweep prologue 10 -- every phase scales with the
// instructions a WAVE issues (~8 cycles per fp64 instruction), not with the SIMD's load: seven lanes per pair (nine pairs
// per wave = exactly four waves instead of 4.5, eleven rows per lane, dot product summed through LDS) halves the barrier
// phase and loses more in the others (0.70 against 0.65 ms); three partial sums in the dot product change nothing.
#ifdef SF_EIGH_STAMPS
#define EIG_STAMP(i) do { if (tid == 0) { const long long t_ = __builtin_readcyclecounter(); stamp[i] += t_ - tprev; tprev = t_; } } while (0)
#else
#define EIG_STAMP(i) do { } while (0)
#endif

// 8-lane butterfly sum with DPP lane swaps (no LDS traffic): xor 1, xor 2, then half-row mirror (the
// partner sits in the other quad, whose four lanes already hold the same partial).
template <int CTRL>
__device__ __forceinline__ double dpp_swap(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double sum4(double v) {
  v += dpp_swap<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_swap<0x4E>(v);   // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ double sum8(double v) {
  v += dpp_swap<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_swap<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_swap<0x141>(v);  // row_half_mirror
  return v;
}
__device__ __forceinline__ double sum16(double v) {
  v = sum8(v);
  v += dpp_swap<0x140>(v);  // row_mirror: the other 8-lane half of the 16-lane row
  return v;
}
// 1/sqrt(x) to double precision: hardware estimate + two Newton steps
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * __builtin_fma(-0.5 * x * y, y, 1.5);
  y = y * __builtin_fma(-0.5 * x * y, y, 1.5);
  return y;
}

#include "cmf_eigh_pre.h"   // the tridiagonal preconditioner of the Jacobi sweeps (round 6)

// pair k of round-robin step s over p2 (even) players; m = p2 - 1
__device__ __forceinline__ void rr_pair(int s, int k, int m, int &a, int &b) {
  int x = s + k;
  x = x >= m ? x - m : x;
  int y = s - k;
  y = y < 0 ? y + m : y;
  a = x;
  b = (k == 0) ? m : y;
}


// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding
// global access (vmcnt(0)); the step loops below keep a rotation store (phase 1) or a rotation prefetch
// (phase 2) in flight across the barrier, and waiting for it costs a full memory round trip per step
// (measured: 4200 -> ~1000 cycles per step).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// RMAX = ceil(p2/8) rows per lane; FULL = (p2 % 8 == 0): every lane owns exactly RMAX rows and all row
// predicates fold away.  Loads are always unconditional (clamped row, value zeroed afterwards): predicated
// LDS loads compile to one exec-masked branch + wait EACH and serialise the step (measured 2300 -> ~900 cycles).
// PRE: 0, or p2 (72 / 84) -- the sweeps start from the preconditioned factor F of cmf_eigh_pre.h instead of L (a second p2 x LD
// matrix and 8 p2 + 16 doubles of LDS behind nrm; 4 p2 threads).
template <int EIG_RMAX, bool FULL, int LPP, int PRE = 0>
__global__ __launch_bounds__(PRE > 0 ? 4 * PRE : 1024) void k_eigh(const double *__restrict__ cov, const int32_t *__restrict__ nuse, int p, int p2, int LD,
                       double *__restrict__ d_out, double *__restrict__ lam_out, double *__restrict__ evec_out,
                       int32_t *__restrict__ status, double2 *__restrict__ rot, size_t rot_stride, int unit) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double *M = sm;                    // [p2][LD] column-major: M[col*LD + row]; G in phase 1, V in phase 2
  double *dv = sm + (size_t)p2 * LD;  // [p2]
  double *nrm = dv + p2;              // [p2] squared column norms of G, refreshed every sweep
  __shared__ int flag[3];   // [0] bad diagonal, [1] a rotation this sweep, [2] a rotation above the tiny ratio this sweep
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int c = blockIdx.x;
  const double *S = cov + (size_t)c * p * p;
  const int n = nuse[c];

  if (tid < 3) flag[tid] = 0;
  for (int i = tid; i < p2; i += nthr) {
    double v = 0.0;
    if (i < p) v = unit ? 1.0 : sqrt(S[(size_t)i * p + i]);   // unit: the matrix is already whitened (cmf_general.hip)
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
  else if (n == 1) st = 3;   // one valid row: numpy.cov (ddof 1) is NaN, every NLL is NaN, argmin = 0, C and the score are NaN
  else if (flag[0]) st = 2;
  if (tid == 0) status[c] = st;
  if (!unit) for (int i = tid; i < p; i += nthr) d_out[(size_t)c * p + i] = dv[i];
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
      M[col * LD + row] = r;
    }
  };
  bool pre_ok = false, pre_done = false;
  if constexpr (PRE > 0) {
    __shared__ int preflag[2];
    double *Bm = nrm + p2, *sml = Bm + (size_t)p2 * LD;
    const int pr = eig_precondition<PRE>(M, Bm, sml, preflag, S, dv, p, LD, rot + (size_t)c * rot_stride);
    pre_ok = pr != 0;
    pre_done = pr == 2;        // F's columns already orthogonal to the sweeps' tolerance: no sweep would rotate anything
    __syncthreads();
  }
  if (!pre_ok) load_R();
  __syncthreads();

  // ---------------- Cholesky R = L L^T in place (lower triangle of the column-major buffer)
  bool chol_ok = true;
  for (int kk = 0; kk < p && !pre_ok; ++kk) {
    const double dk = M[kk * LD + kk];  // final after the previous trailing update + barrier
    if (!(dk > 0.0) || !(dk <= 1.79769313486231570e+308)) { chol_ok = false; break; }  // uniform
    const double rk = rsqrt_nr(dk);
    __syncthreads();                    // everyone has read dk before it is overwritten
    for (int i = kk + tid; i < p; i += nthr) M[kk * LD + i] = (i == kk) ? dk * rk : M[kk * LD + i] * rk;
    __syncthreads();
    const int rem = p - kk - 1;         // trailing update of the lower triangle, columns kk+1 .. p-1
    for (int e = tid; e < rem * rem; e += nthr) {
      const int jj = e / rem, ii = e - jj * rem;
      if (ii >= jj) {
        const int j = kk + 1 + jj, i = kk + 1 + ii;
        M[j * LD + i] = __builtin_fma(-M[kk * LD + i], M[kk * LD + j], M[j * LD + i]);
      }
    }
    __syncthreads();
  }
  if (pre_ok) {
    // M = F already (F F^T = R, columns orthogonal to ~1e-12: the first sweep below is the last)
  } else if (chol_ok) {
    for (int i = tid; i < p2 * p2; i += nthr) {  // G = L: zero the strict upper triangle and the padding
      const int col = i / p2, row = i - col * p2;
      if (row < col || col >= p || row >= p) M[col * LD + row] = 0.0;
    }
  } else {
    __syncthreads();
    load_R();
  }
  __syncthreads();

  const int npairs = p2 >> 1, m = p2 - 1;
  const int k = tid / LPP, sub = tid % LPP;   // LPP lanes share a column pair (rows sub, sub + LPP, ...)
  const bool active = k < npairs;
  const int nr = (p2 - sub + LPP - 1) / LPP;  // rows owned by this lane
  const double2 *myrot = rot + (size_t)c * rot_stride + k;
  // |g_a . g_b| <= tol |g_a||g_b|: the dot product itself carries ~sqrt(p) eps of rounding, so a
  // threshold below that never settles; p*eps leaves off-diagonals <= 1e-14 * min(lam_a, lam_b).
  const double tol = (double)p2 * 2.220446049250313e-16;
  const double tol2 = tol * tol;
  int nsteps = 0;  // fallback only: steps whose rotations have to be replayed on V
#ifdef SF_EIGH_STAMPS
  long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
  const long long tstart = tprev;
#endif
  // ---------------- phase 1: orthogonalise the columns of G (record the rotations in the fallback)
  for (int sweep = 0; sweep < (pre_done ? 0 : EIG_MAXSWEEP); ++sweep) {
    bool rotated = false, big = false;
    // exact squared norms once per sweep; inside the sweep they are carried through the rotations
    // (|a'|^2 = c^2 aa - 2cs ab + s^2 bb), so a step needs ONE dot product instead of three
    for (int j = tid; j < p2; j += nthr) {
      double sacc = 0;
      for (int r = 0; r < p2; ++r) {
        const double x = M[j * LD + r];
        sacc = __builtin_fma(x, x, sacc);
      }
      nrm[j] = sacc;
    }
    __syncthreads();
    // Column residency.  In the circle method pair k of step s is (a, b) = (s + k, s - k) mod m (k = 0: b = m), so the
    // a-columns of step s+1 are those of step s minus column s plus column s+36: a lane group that FOLLOWS its
    // a-column keeps it in registers from one step to the next, and only the b-columns (and the one a-column that
    // changes sides) go through LDS.  Same pairs, same rotations, same results as re-reading both columns every
    // step -- but half the LDS traffic, and the step is LDS-store bound (ds_write_b64 moves ~85 B/clk per CU).
    int acol = k;               // this group's a-column: pair index (acol - s) mod m at step s
    bool reload = true;         // xa has to come from LDS (sweep start, or the column changed sides)
    double xa[EIG_RMAX];
#pragma unroll
    for (int i = 0; i < EIG_RMAX; ++i) xa[i] = 0.0;
    EIG_STAMP(0);   // sweep prologue (norms)
    for (int s = 0; s < m; ++s) {
      if (active) {
        int kk = acol - s;
        kk = kk < 0 ? kk + m : kk;                       // current pair index of this group, 0 .. npairs-1
        int bcol = 2 * s - acol;
        bcol = bcol < 0 ? bcol + m : (bcol >= m ? bcol - m : bcol);
        const int a = acol, b = (kk == 0) ? m : bcol;
        double *ga = M + a * LD + sub, *gb = M + b * LD + sub;
        double xb[EIG_RMAX];
#pragma unroll
        for (int i = 0; i < EIG_RMAX; ++i) {
          const int ii = FULL ? i : min(i, nr - 1);
          const double vb_ = gb[LPP * ii];
          xb[i] = (FULL || i < nr) ? vb_ : 0.0;
        }
        if (reload) {
#pragma unroll
          for (int i = 0; i < EIG_RMAX; ++i) {
            const int ii = FULL ? i : min(i, nr - 1);
            const double va_ = ga[LPP * ii];
            xa[i] = (FULL || i < nr) ? va_ : 0.0;
          }
        }
        const bool leaving = (kk == 0) || (s == m - 1);  // after this step the a-column is someone's b-column / the sweep ends
        const double aa = nrm[a], bb = nrm[b];
#ifdef SF_EIGH_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        EIG_STAMP(1);   // operands landed
#endif
        double ab = 0;
#pragma unroll
        for (int i = 0; i < EIG_RMAX; ++i) ab = __builtin_fma(xa[i], xb[i], ab);
        ab = (LPP == 16) ? sum16(ab) : ((LPP == 8) ? sum8(ab) : sum4(ab));
        const double ab2 = aa * bb;
        double cs = 1.0, sn = 0.0;
#ifdef SF_EIGH_STAMPS
        asm volatile("" : "+v"(ab));
        EIG_STAMP(2);   // dot product reduced
#endif
        if (ab2 > 0.0 && ab * ab > tol2 * ab2) {  // uniform over the pair's 8 lanes
          rotated = true;
          big = big || (ab * ab > 1e-18 * ab2);
          // tan(2 theta) = 2ab / (bb - aa), small-angle branch, no division:
          const double tau = bb - aa, gam = 2.0 * ab;
          const double rinv = rsqrt_nr(__builtin_fma(tau, tau, gam * gam));
          const double c2 = fabs(tau) * rinv;             // |cos 2theta|
          const double h = __builtin_fma(0.5, c2, 0.5);   // cos^2 theta in [0.5, 1]
          const double rh = rsqrt_nr(h);
          cs = h * rh;
          sn = fabs(gam) * rinv * 0.5 * rh;
          sn = ((tau < 0.0) != (gam < 0.0)) ? -sn : sn;
#ifdef SF_EIGH_STAMPS
          asm volatile("" : "+v"(sn), "+v"(cs));
          EIG_STAMP(3);   // rotation parameters
#endif
#pragma unroll
          for (int i = 0; i < EIG_RMAX; ++i) {
            const double na = cs * xa[i] - sn * xb[i];
            const double nb = sn * xa[i] + cs * xb[i];
            xa[i] = na;
            if (FULL || i < nr) gb[LPP * i] = nb;
          }
          if (sub == 0) {
            const double cc = cs * cs, ss = sn * sn, x2 = 2.0 * cs * sn * ab;
            nrm[a] = cc * aa - x2 + ss * bb;
            nrm[b] = ss * aa + x2 + cc * bb;
          }
        }
        if (leaving) {
#pragma unroll
          for (int i = 0; i < EIG_RMAX; ++i)
            if (FULL || i < nr) ga[LPP * i] = xa[i];
        }
        if (!chol_ok && sub == 0) rot[(size_t)c * rot_stride + (size_t)(sweep * m + s) * npairs + kk] = make_double2(cs, sn);
        reload = (kk == 0);                              // next step this group owns column s + npairs instead
        if (kk == 0) { acol = s + npairs; acol = acol >= m ? acol - m : acol; }
      }
      EIG_STAMP(4);     // rotation applied, stores issued
      lds_barrier();
      EIG_STAMP(5);     // barrier
    }
    if (rotated) flag[1] = 1;  // benign race: every writer stores 1
    if (big) flag[2] = 1;
    __syncthreads();
    const int any = flag[1], anybig = flag[2];
    __syncthreads();
    if (tid == 0) { flag[1] = 0; flag[2] = 0; }
    if (!any) break;           // this sweep was all identities: nothing of it to replay
    nsteps = (sweep + 1) * m;
    // every rotation of the sweep was TINY (|a.b| <= 1e-9 |a||b|): each such pair is now orthogonal to working precision and
    // disturbed the others by the square of that ratio -- the sweep that would follow is the all-identities verification sweep
    if (!anybig) break;
  }
  __syncthreads();
  if (chol_ok && tid == 0) rot[(size_t)c * rot_stride] = make_double2((double)(nsteps / m), 0.0);  // tools/probe_eigh.py
#ifdef SF_EIGH_STAMPS
  if (chol_ok && tid == 0) {
    for (int i = 0; i < 6; ++i) rot[(size_t)c * rot_stride + 1 + i] = make_double2((double)stamp[i], 0.0);
    rot[(size_t)c * rot_stride + 7] = make_double2((double)(__builtin_readcyclecounter() - tstart), 0.0);
  }
#endif
  if (chol_ok) {
    // G = U diag(sigma): lam = sigma^2, eigenvector = normalised column
    for (int j = tid; j < p; j += nthr) {
      double sacc = 0;
      for (int r = 0; r < p; ++r) {
        const double x = M[j * LD + r];
        sacc = __builtin_fma(x, x, sacc);
      }
      nrm[j] = sacc;
      lam_out[(size_t)c * p + j] = sacc;
    }
    __syncthreads();
    for (int i = tid; i < p * p; i += nthr) {
      const int j = i / p, b = i - j * p;
      const double s2 = nrm[j];
      evec_out[(size_t)c * p * p + i] = s2 > 0.0 ? M[j * LD + b] * rsqrt_nr(s2) : ((j == b) ? 1.0 : 0.0);
    }
    return;
  }
  // eigenvalues = column norms of G = R V
  for (int j = tid; j < p; j += nthr) {
    double sacc = 0;
    for (int r = 0; r < p2; ++r) {
      const double x = M[j * LD + r];
      sacc += x * x;
    }
    lam_out[(size_t)c * p + j] = sqrt(sacc);
  }
  __syncthreads();
  // ---------------- phase 2: V = I, replay
  for (int i = tid; i < p2 * p2; i += nthr) {
    const int col = i / p2, row = i - col * p2;
    M[col * LD + row] = (col == row) ? 1.0 : 0.0;
  }
  __syncthreads();
  double2 rnext = make_double2(1.0, 0.0);
  if (active && nsteps > 0) rnext = myrot[0];
  int s = 0;
  for (int t = 0; t < nsteps; ++t) {
    if (active) {
      const double2 r = rnext;
      if (t + 1 < nsteps) rnext = myrot[(size_t)(t + 1) * npairs];  // in flight across the barrier
      if (r.y != 0.0) {
        int a, b;
        rr_pair(s, k, m, a, b);
        double *va = M + a * LD + sub, *vb = M + b * LD + sub;
        double ya[EIG_RMAX], yb[EIG_RMAX];
#pragma unroll
        for (int i = 0; i < EIG_RMAX; ++i) {
          const int ii = FULL ? i : min(i, nr - 1);
          ya[i] = va[LPP * ii];
          yb[i] = vb[LPP * ii];
        }
#pragma unroll
        for (int i = 0; i < EIG_RMAX; ++i) {
          if (FULL || i < nr) {
            va[LPP * i] = r.x * ya[i] - r.y * yb[i];
            vb[LPP * i] = r.y * ya[i] + r.x * yb[i];
          }
        }
      }
    }
    s = (s + 1 == m) ? 0 : s + 1;
    lds_barrier();
  }
  for (int i = tid; i < p * p; i += nthr) {
    const int j = i / p, b = i - j * p;
    evec_out[(size_t)c * p * p + i] = M[j * LD + b];
  }
}

}  // namespace


size_t sf_eigh_scratch_bytes(const SfGeom &g) {
  const int p2 = g.p + (g.p & 1);
  return sf_align((size_t)g.ncols * EIG_MAXSWEEP * (p2 - 1) * (p2 / 2) * sizeof(double2));
}

static int launch_eigh(const double *cov, const int32_t *nuse, const SfGeom &g, double *d, double *lam, double *evec,
                       int32_t *status, void *scratch, int unit, hipStream_t st);
int sf_launch_eigh(const double *cov, const int32_t *nuse, const SfGeom &g, double *d, double *lam, double *evec,
                   int32_t *status, void *scratch, hipStream_t st) {
  return launch_eigh(cov, nuse, g, d, lam, evec, status, scratch, 0, st);
}
// the same solver on an already whitened matrix: no diagonal scaling, d is not written
int sf_launch_eigh_unit(const double *cov, const int32_t *nuse, const SfGeom &g, double *d, double *lam, double *evec,
                        int32_t *status, void *scratch, hipStream_t st) {
  return launch_eigh(cov, nuse, g, d, lam, evec, status, scratch, 1, st);
}
static int launch_eigh(const double *cov, const int32_t *nuse, const SfGeom &g, double *d, double *lam, double *evec,
                       int32_t *status, void *scratch, int unit, hipStream_t st) {
  const int p2 = g.p + (g.p & 1);
  // Lanes per column pair: 8 is the measured optimum (tools/probe_eigh.py, one 72x72 matrix: 8 lanes 0.65 ms,
  // 4 lanes -- 3 waves, one per SIMD, twice the rows per lane -- 0.78 ms, 16 lanes -- 9 waves -- 0.77 ms).
  // The other two stay reachable through sf_debug_set(7, .) for the production sizes.
  const int lpp = (sf_tune().eigh_lpp == 4 || sf_tune().eigh_lpp == 8 || sf_tune().eigh_lpp == 16) ? sf_tune().eigh_lpp : 8;
  const bool lpp4 = lpp == 4 && (p2 == 72 || p2 == 84);
  const bool lpp16 = lpp == 16 && (p2 == 72 || p2 == 84);
  int LD = p2;
  if (lpp4) { while ((LD % 32) != 12 && (LD % 32) != 20) ++LD; }     // 4 consecutive columns x 4 rows: 64 distinct banks
  else if (lpp16) { while ((LD % 32) != 16) ++LD; }
  else { while ((LD % 32) != 8 && (LD % 32) != 24) ++LD; }
  // The tridiagonal preconditioner (cmf_eigh_pre.h; the CH4 / CO2 window sizes, 8 lanes per pair) is built, correct and OFF by
  // default: sf_debug_set(7, 2) turns it on.  Measured (profiles/r06_eigh_precond.md): one sweep instead of nine, but the
  // preconditioner itself costs 1.34 M cycles of latency-bound float64 work on ONE workgroup (0.62 against 0.66 ms for a 72-band
  // matrix alone) and its second LDS matrix leaves one workgroup per CU instead of three (598 columns: 1.90 against 1.14 ms).
  const bool pre = !lpp4 && !lpp16 && (p2 == 72 || p2 == 84) && sf_tune().eigh_lpp == 2;
  const size_t lds = ((size_t)p2 * LD + 2 * p2 + (pre ? (size_t)p2 * LD + 8 * p2 + 16 : 0)) * sizeof(double);
  if (lds > 160 * 1024 - 64 || g.p > SF_MAX_ACTIVE_FUSED) {
    sf_set_error("active window of %d bands exceeds the LDS-resident eigensolver (max %d)", g.p, SF_MAX_ACTIVE_FUSED);
    return -2;
  }
  int threads = (p2 / 2) * (lpp4 ? 4 : (lpp16 ? 16 : 8));
  threads = (threads + 63) / 64 * 64;
  if (threads < 64) threads = 64;
  if (pre) threads = 4 * p2;                    // exactly four lanes per column (cmf_eigh_pre.h; __launch_bounds__ of the instantiation)
  const size_t rot_stride = (size_t)EIG_MAXSWEEP * (p2 - 1) * (p2 / 2);
  const int rmax = (p2 + 7) / 8;
  const bool full = (p2 % 8) == 0;
  auto go = [&](auto kern) -> int {
    if (int rc = sf_lds_attr(reinterpret_cast<const void *>(kern), lds)) return rc;
    hipLaunchKernelGGL(kern, dim3(g.ncols), dim3(threads), lds, st, cov, nuse, g.p, p2, LD, d, lam, evec, status,
                       reinterpret_cast<double2 *>(scratch), rot_stride, unit);
    return 0;
  };
  int rc = -2;
  if (pre) {
    rc = (p2 == 72) ? go(k_eigh<9, true, 8, 72>) : go(k_eigh<11, false, 8, 84>);
  } else if (lpp4) {
    rc = (p2 == 72) ? go(k_eigh<18, true, 4>) : go(k_eigh<21, true, 4>);
  } else if (lpp16) {
    rc = (p2 == 72) ? go(k_eigh<5, false, 16>) : go(k_eigh<6, false, 16>);
  } else {
  switch (rmax) {
#define EIG_CASE(R) case R: rc = full ? go(k_eigh<R, true, 8>) : go(k_eigh<R, false, 8>); break;
    EIG_CASE(1) EIG_CASE(2) EIG_CASE(3) EIG_CASE(4) EIG_CASE(5) EIG_CASE(6)
    EIG_CASE(7) EIG_CASE(8) EIG_CASE(9) EIG_CASE(10) EIG_CASE(11) EIG_CASE(12)
#undef EIG_CASE
    default: sf_set_error("eigensolver: unsupported size %d", g.p); return -2;
  }
  }
  if (rc) return rc;
  SF_LAUNCH_CHECK("k_eigh");
  return 0;
}
