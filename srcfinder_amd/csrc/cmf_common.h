// Shared declarations for the CMF HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "srcfinder_amd.h"

typedef double d4_t __attribute__((ext_vector_type(4)));

// ---- error plumbing (c_api.hip) -------------------------------------------------------------
void sf_set_error(const char *fmt, ...);
int sf_fail_hip(hipError_t e, const char *what);
#define SF_HIP(call)                                   \
  do {                                                 \
    hipError_t _e = (call);                            \
    if (_e != hipSuccess) return sf_fail_hip(_e, #call); \
  } while (0)
#define SF_LAUNCH_CHECK(name)                          \
  do {                                                 \
    hipError_t _e = hipGetLastError();                 \
    if (_e != hipSuccess) return sf_fail_hip(_e, name); \
  } while (0)

// Raise a kernel's dynamic-LDS limit to `bytes` on the CURRENT device (the attribute is per device and per function;
// remembered per (device, function) so the driver is asked only when the size grows).  Thread safe.  c_api.hip
int sf_lds_attr(const void *fn, size_t bytes);

#include "sf_tune.h"   // the per-thread tuning / experiment knobs (sf_debug_set)

static inline int sf_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t sf_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Row validity of the reference: ((~(x<0)) & isfinite(x))  (cmf/robust_mf.py:282).
// NaN fails the first compare, +inf the second, -0.0 passes like the reference.
__device__ __forceinline__ bool sf_valid(float x) { return (x >= 0.0f) & (x <= 3.402823466e+38f); }

// Column-major row data (xt) may be float32 (the extracted cube) or float64 (the function-level
// looshrinkage() entry, whose input is already centred float64): 4- and 2-element loads promoted to double.
__device__ __forceinline__ void sf_load4(const float *p, double &a, double &b, double &c, double &d) {
  const float4 f = *reinterpret_cast<const float4 *>(p);
  a = (double)f.x; b = (double)f.y; c = (double)f.z; d = (double)f.w;
}
__device__ __forceinline__ void sf_load4(const double *p, double &a, double &b, double &c, double &d) {
  const double2 u = *reinterpret_cast<const double2 *>(p), v = *reinterpret_cast<const double2 *>(p + 2);
  a = u.x; b = u.y; c = v.x; d = v.y;
}
__device__ __forceinline__ void sf_load2(const float *p, float &a, float &b) {
  const float2 f = *reinterpret_cast<const float2 *>(p);
  a = f.x; b = f.y;
}
__device__ __forceinline__ void sf_load2(const double *p, double &a, double &b) {
  const double2 f = *reinterpret_cast<const double2 *>(p);
  a = f.x; b = f.y;
}

// XCD-aware workgroup -> (column block, line chunk) map for the two kernels that stream the BIL cube in
// 64-sample column blocks.  A block's 256-byte row segments start at arbitrary offsets inside 128-byte lines,
// so neighbouring column blocks share their boundary lines.  Workgroups are dealt round-robin over the 8 XCDs
// (bid % 8), each with a private L2: with the natural (cb fastest) order the two sharers sit on different
// XCDs and both fetch the line (measured: FETCH_SIZE = 1.5x the algorithmic bytes).  Here all column blocks of a
// line chunk get consecutive slots of ONE XCD, so the second request is an L2 hit.  Speed only: any placement
// gives the same results.
__device__ __forceinline__ bool sf_xcd_map(int bid, int ncb, int nchunk, int &cb, int &chunk) {
  const int xcd = bid & 7, slot = bid >> 3;
  chunk = (slot / ncb) * 8 + xcd;
  cb = slot % ncb;
  return chunk < nchunk;
}
static inline int sf_xcd_grid(int ncb, int nchunk) { return sf_cdiv(nchunk, 8) * 8 * ncb; }

// ---- geometry shared by host launchers and kernels ---------------------------------------------
struct SfGeom {
  int lines, p, ps, nt, s4, ncols, nalpha, nu;
};
static inline SfGeom sf_geom(int lines, int p, int ncols, int nalpha) {
  SfGeom g;
  g.lines = lines; g.p = p; g.ncols = ncols; g.nalpha = nalpha;
  g.ps = (p + 3) / 4 * 4;        // xt row stride (floats), 16-byte rows
  g.nt = (p + 15) / 16;          // 16-wide MFMA tiles over the band axis
  g.s4 = (p + 3) / 4;            // 4-deep MFMA k-steps over the band / eigen axis
  g.nu = (nalpha + 15) / 16;     // 16-wide tiles over the alpha grid
  return g;
}

// split counts.  Every one is a function of the number of LINES only (the ncols argument is deliberately unused):
// the split boundaries fix the order in which a column's float64 sums (masked column sums, covariance, NLL terms,
// score statistics) are accumulated, and a column must come out bit-identical whether it is processed as part of the
// full 598-column flightline or of a 74-column shard on one of 8 GPUs (tests/test_cmf_gpu.py:
// test_shard_of_a_full_width_run_is_bit_identical).  The values are chosen for the standard widths 598 / 299 / 150 / 75.
static inline int sf_extract_lines_per_wg(int lines, int ncols) {
  // <= 500 line chunks (measured: 250 -> 2.05 ms, 500 -> 1.91 ms, 1000 -> 1.97 ms + a slower mean kernel): every
  // chunk leaves a partial-sum record per column that the mean kernel reads back.
  (void)ncols;
  int lpw = sf_cdiv(lines, 500);
  lpw = (lpw + 3) / 4 * 4;
  return lpw < 4 ? 4 : lpw;
}
static inline int sf_colsum_chunks(int lines, int ncols) {
  (void)ncols;
  const int maxc = sf_cdiv(lines, 256);
  return maxc < 4 ? (maxc < 1 ? 1 : maxc) : 4;
}
static inline int sf_syrk_splits(int lines, int ncols) {
  // the 16x16x4 covariance (windows other than the production one): 7 workgroups per CU are resident, so 598 columns
  // x 28 splits are ~9 rounds with a short last one
  (void)ncols;
  const int maxc = sf_cdiv(lines, 512);
  return maxc < 28 ? (maxc < 1 ? 1 : maxc) : 28;
}
static inline int sf_sweep_splits(int lines, int ncols) {
  // The 4x4x4 sweep and covariance run ONE workgroup per CU (LDS- / register-bound) and all workgroups take the same
  // time, so a launch proceeds in rounds of 256 workgroups.  2048-row splits: at 20000 lines 598 / 299 / 150 / 75
  // columns x 10 splits are 23.4 / 11.7 / 5.9 / 2.9 rounds -- the last round is nearly full for every standard shard
  // width (4096-row splits cost the same at 598 columns but leave a 75-column shard 1.5 rounds of 2: +33 %).
  (void)ncols;
  int ns = sf_cdiv(lines, 2048);
  if (ns > 24) ns = 24;
  return ns < 1 ? 1 : ns;
}
static inline int sf_score_lines_per_wg(int lines, int ncols) {
  // the column-block score kernel (the row kernel has its own fixed 8-line granularity): 32-line chunks -- one 8-line
  // batch per wave, no line loop at all -- measured best on the full flightline (tools/tune_score.py)
  (void)ncols;
  return lines >= 2048 ? 32 : 16;
}

// ---- stage launchers (each in its own .hip) -------------------------------------------------------
int sf_launch_extract(const float *cube, int lines, int bands, int samples, int s0, int ncols, int b0, int p,
                      float *xt, uint8_t *mask_t, double *sum_part, int *cnt_part, hipStream_t st);
size_t sf_extract_sum_bytes(const SfGeom &g);
bool sf_extract_fuses_sum(int p);
int sf_launch_extract_fused(const float *cube, int lines, int bands, int samples, int s0, int b0, const SfGeom &g,
                            float *xt, uint8_t *mask_t, void *scratch, hipStream_t st);
int sf_launch_mean_from_partials(const SfGeom &g, int32_t *nuse, double *mu, void *scratch, hipStream_t st);
size_t sf_mean_scratch_bytes(const SfGeom &g);
int sf_launch_mean(const void *xt, int xt_f64, const uint8_t *mask_t, const SfGeom &g, int32_t *nuse, double *mu,
                   void *scratch, hipStream_t st);
size_t sf_cov_scratch_bytes(const SfGeom &g);
int sf_launch_cov(const void *xt, int xt_f64, const uint8_t *mask_t, const int32_t *nuse, const double *mu,
                  const SfGeom &g, double *cov, void *scratch, hipStream_t st);
size_t sf_wfrag_elems(const SfGeom &g);   // per column
size_t sf_eigh_scratch_bytes(const SfGeom &g);
int sf_launch_eigh(const double *cov, const int32_t *nuse, const SfGeom &g, double *d, double *lam, double *evec,
                   int32_t *status, void *scratch, hipStream_t st);
size_t sf_loocv_scratch_bytes(const SfGeom &g);
int sf_launch_loocv(const void *xt, int xt_f64, const uint8_t *mask_t, const int32_t *nuse, const double *mu, const double *d,
                    const double *lam, const double *evec, const int32_t *status, const double *alphas,
                    const SfGeom &g, double *nll, int32_t *alphaidx, void *scratch, hipStream_t st);
int sf_launch_filter(const double *mu, const double *d, const double *lam, const double *evec, const double *alphas,
                     int32_t *alphaidx, const double *abscf, int reflectance, const SfGeom &g,
                     int32_t *status, double *filt, double *bias, hipStream_t st);
size_t sf_wide_scratch_bytes(const SfGeom &g);
int sf_launch_wide_stats(const void *xt, int xt_f64, const uint8_t *mask_t, const int32_t *nuse, const int32_t *nloo,
                         const double *mu, const double *alphas, const SfGeom &g, double *cov, double *d, double *lam,
                         double *evec, int32_t *status, double *nll, int32_t *alphaidx, void *scratch, hipStream_t st,
                         const double *target = nullptr);   // target [ncols][p][p]: full shrinkage target (-f), else diag(S)
// cmf_wgemm.hip: the wide-window covariance and sweep on the 4x4x4 fp64 MFMA, fused with centring / squaring / row reduction
int sf_wgemm_splits(const SfGeom &g);
size_t sf_wgemm_operand_bytes(const SfGeom &g);   // W and C of g.ncols columns
size_t sf_wgemm_part_bytes(const SfGeom &g);      // the sweep partials of the whole flightline
// rank factorisation of the wide sweep's coefficient matrix (cmf_wlr.hip): operand images U8 / T8 and the verdict per column
size_t sf_wlr_scratch_bytes(int nb);
size_t sf_wlr_image_bytes(int nb);
int sf_launch_wlr(const double *lam, const int32_t *nloo, const int32_t *status, const double *alphas, int nalpha, int p, int nb,
                  int njw, int njl, void *scratch, void *images, const double **U8, const double **T8, const int32_t **wlr,
                  hipStream_t st);
// cmf_wtri.hip: the tridiagonal preconditioner of the wide eigensolver; cmf_wide.hip: the batched float64 GEMM it uses
// (column-major n x n matrices: C = Y X for tb = 0, C = Y^T X ... see the definition), matrices with skip1 / skip2 != 0 untouched
size_t sf_wtri_small_bytes(int p, int nb);
int sf_launch_wtri_prepare(double *gv, int p, int p2, int nb, double *B2, double *B3, double *small, const int32_t *cflag,
                           int32_t *pflag, hipStream_t st, int gbn, size_t gstride);   // (nb matrices in groups of gbn, gstride bytes apart; gbn 0: one group)
int sf_launch_wtri_apply(double *gv, int p, int p2, int nb, double *B2, double *B3, double *small, int32_t *cflag,
                         int32_t *pflag, hipStream_t st);
int sf_wide_dgemm(const double *A, int lda, size_t sA, const double *B, int ldb, size_t sB, int tb, double *C, int ldc, size_t sC,
                  int n, int nb, const int32_t *skip1, const int32_t *skip2, hipStream_t st, int lower = 0);
int sf_launch_wsyrk(const void *xt, int xt_f64, const uint8_t *mask_t, const int32_t *nuse, const double *mu, const SfGeom &g,
                    int c0, int nb, double *cov, hipStream_t st);
int sf_launch_wsweep(const void *xt, int xt_f64, const uint8_t *mask_t, const int32_t *nloo, const double *mu, const double *d,
                     const double *lam, const double *evec, const int32_t *status, const double *alphas, const SfGeom &g, int c0,
                     int nb, void *opnd, double *part, hipStream_t st);
int sf_launch_nll_finish(const double *part, int nsplit, const int32_t *nuse, const double *d, const double *lam,
                         const int32_t *status, const double *alphas, const SfGeom &g, double *nll, int32_t *alphaidx,
                         hipStream_t st, double *rest = nullptr);   // rest: the NLL without its determinant term
// linalg.hip: the exact determinants (partial-pivot LU, running product) of the grid points whose classification by the
// total log-determinant is not certain (window > 0), or of every grid point (window <= 0)
size_t sf_exact_det_scratch_bytes(const SfGeom &g, int window);
int sf_exact_det_window(const SfGeom &g);   // cmf_wide.hip: the policy (0 = every grid point)
int sf_launch_exact_det(const double *cov, const int32_t *nloo, const int32_t *status, const double *alphas, const SfGeom &g,
                        int window, const double *rest, double *nll, int32_t *alphaidx, void *scratch, hipStream_t st,
                        const double *target = nullptr);
size_t sf_score_scratch_bytes(int lines, int ncols);
int sf_launch_score(const float *cube, int lines, int bands, int samples, int s0, int ncols, int b0, int p,
                    const double *filt, const double *bias, const int32_t *status, const int32_t *alphaidx,
                    int rgb0, int rgb1, int rgb2, double nodata, double *out, int out_samples, int out_s0,
                    int out_bands, int16_t *bgmeta, void *scratch, int want_stats, hipStream_t st,
                    hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr);   // events bracket the score kernel alone
int sf_launch_colstats(const void *stat_scratch, int lines, int samples, int s0, int ncols, int p, const int32_t *nuse,
                       const int32_t *status, double nodata, double *colstats, hipStream_t st);
// cmf_loocv4.hip: the production-window sweep (p in 69..72, 201-point grid) on the 4x4x4 fp64 MFMA
constexpr int SF_SW4_NJ = 18, SF_SW4_NM = 13;
// windows the 4x4x4 kernels serve: band groups of four (a lane group of the MFMA holds NJ consecutive bands, 4 NJ = the row
// stride of xt, and at most the last three lie beyond the window): 18 (p 69..72: CH4), 21 (81..84: CO2, p = 83), 24 (93..96)
static inline int sf_sw4_groups(int p) {
  const int s4 = (p + 3) / 4;
  return (s4 == 18 || s4 == 21 || s4 == 24) ? s4 : 0;
}
constexpr int SF_LR_K0 = 24, SF_LR_K = 28, SF_LR_K2 = 36;   // ranks of the factored sweep coefficients (cmf_lowrank.hip); fragments use the K2 layout
// verdict per column (lrok): 0 = not factored (full-rank sweep), 1 = rank 28, 2 = rank 36, 3 = rank 24
constexpr int sf_lr_code(int nk) { return nk == SF_LR_K0 / 4 ? 3 : (nk == SF_LR_K / 4 ? 1 : 2); }
size_t sf_lowrank_bytes(const SfGeom &g);
int sf_launch_lowrank(const double *lam, const int32_t *nuse, const int32_t *status, const double *alphas, const SfGeom &g,
                      double *ufrag, double *wfrag, int32_t *lrok, hipStream_t st, int allow_k0 = 1);
// cmf_cov4.hip: the production-window covariance on the 4x4x4 fp64 MFMA
size_t sf_cov4_scratch_bytes(const SfGeom &g);
int sf_launch_cov4(const float *xt, const uint8_t *mask_t, const int32_t *nuse, const double *mu, const SfGeom &g,
                   double *cov, void *scratch, hipStream_t st);
int sf_launch_wfrag4(const double *evec, const double *d, const SfGeom &g, size_t wstride, double *wfrag, hipStream_t st);
// k_sweep4s for the windows of 21 / 24 band groups: their own translation units (cmf_loocv4_21.hip, cmf_loocv4_24.hip)
int sf_launch_sweep4s_21(const float *xt, const uint8_t *mask_t, const int32_t *nuse, const double *mu, const double *ufrag,
                         const double *wfrag2, const int32_t *lrok, const double *lam, const double *wfrag, size_t wstride,
                         const int32_t *status, const double *alphas, const SfGeom &g, int nsplit, double *part, hipStream_t st);
int sf_launch_sweep4s_24(const float *xt, const uint8_t *mask_t, const int32_t *nuse, const double *mu, const double *ufrag,
                         const double *wfrag2, const int32_t *lrok, const double *lam, const double *wfrag, size_t wstride,
                         const int32_t *status, const double *alphas, const SfGeom &g, int nsplit, double *part, hipStream_t st);
int sf_launch_sweep4(const float *xt, const uint8_t *mask_t, const int32_t *nuse, const double *mu, const double *lam,
                     const double *wfrag, size_t wstride, const int32_t *status, const double *alphas, const SfGeom &g,
                     int nsplit, double *part, int variant, void *lr_scratch, hipStream_t st,
                     const int32_t **lrok_out = nullptr);   // lrok_out: where the rank factorisation left its verdict per column

