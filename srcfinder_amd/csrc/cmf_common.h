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

// split counts (deterministic functions of the geometry so that results do not depend on the GPU)
static inline int sf_extract_lines_per_wg(int lines, int ncols) {
  // <= 500 line chunks (measured: 250 -> 2.05 ms, 500 -> 1.91 ms, 1000 -> 1.97 ms + a slower mean kernel): every
  // chunk leaves a partial-sum record per column that the mean kernel reads back.
  // Deliberately independent of the number of columns: the chunk boundaries fix the order in which a column's
  // masked sum is accumulated, and that must not depend on how the columns are sharded over ranks.
  (void)ncols;
  int lpw = sf_cdiv(lines, 500);
  lpw = (lpw + 3) / 4 * 4;
  return lpw < 4 ? 4 : lpw;
}
static inline int sf_colsum_chunks(int lines, int ncols) {
  int want = sf_cdiv(2048, ncols);
  int maxc = sf_cdiv(lines, 256);
  int c = want < maxc ? want : maxc;
  return c < 1 ? 1 : c;
}
static inline int sf_syrk_splits(int lines, int ncols) {
  // many more workgroups than resident slots (7 per CU) so the last partial round costs < 1/8
  int want = sf_cdiv(16384, ncols);
  int maxc = sf_cdiv(lines, 512);
  int c = want < maxc ? want : maxc;
  return c < 1 ? 1 : c;
}
static inline int sf_sweep_splits(int lines, int ncols) {
  // The sweep runs ONE workgroup per CU (LDS-bound) and all workgroups take the same time, so the launch
  // proceeds in rounds of 256: pick the row split whose last round is fullest (598 columns x 3 splits
  // would leave 2 workgroups alone in an 8th round; x 5 fills 11.7 of 12).
  int maxs = sf_cdiv(lines, 1024);
  if (maxs > 24) maxs = 24;
  if (maxs < 1) maxs = 1;
  int best = 1;
  double best_cost = 1e300;
  for (int ns = 1; ns <= maxs; ++ns) {
    const long wgs = (long)ncols * ns;
    const long rounds = (wgs + 255) / 256;
    const double cost = (double)rounds / ns * (1.0 + 0.004 * ns);  // mild preference for fewer prologues
    if (cost < best_cost * 0.999) { best_cost = cost; best = ns; }
  }
  return best;
}
static inline int sf_score_lines_per_wg(int lines, int ncols) {
  // 32-line chunks (one 8-line batch per wave: the workgroup has no line loop at all) measured best on the full
  // flightline (tools/tune_score.py: 0.81 ms against 0.86 ms for 64 lines, many short workgroups beat fewer long
  // ones); keep at least ~1 round of 1024 workgroups on small shards
  int colblocks = sf_cdiv(ncols, 64);
  int lpw = 32;
  while (lpw > 16 && (long)colblocks * sf_cdiv(lines, lpw) < 1024) lpw -= 16;
  return lpw;
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
                         double *evec, int32_t *status, double *nll, int32_t *alphaidx, void *scratch, hipStream_t st);
int sf_launch_nll_finish(const double *part, int nsplit, const int32_t *nuse, const double *d, const double *lam,
                         const int32_t *status, const double *alphas, const SfGeom &g, double *nll, int32_t *alphaidx,
                         hipStream_t st);
size_t sf_score_scratch_bytes(int lines, int ncols);
int sf_launch_score(const float *cube, int lines, int bands, int samples, int s0, int ncols, int b0, int p,
                    const double *filt, const double *bias, const int32_t *status, const int32_t *alphaidx,
                    int rgb0, int rgb1, int rgb2, double nodata, double *out, int out_samples, int out_s0,
                    int out_bands, int16_t *bgmeta, void *scratch, int want_stats, hipStream_t st);
int sf_launch_colstats(const void *stat_scratch, int lines, int samples, int s0, int ncols, int p, const int32_t *nuse,
                       const int32_t *status, double nodata, double *colstats, hipStream_t st);
// cmf_loocv4.hip: the production-window sweep (p in 69..72, 201-point grid) on the 4x4x4 fp64 MFMA
constexpr int SF_SW4_NJ = 18, SF_SW4_NM = 13;
constexpr int SF_LR_K = 28, SF_LR_K2 = 36;   // ranks of the factored sweep coefficients (cmf_lowrank.hip); fragments use the K2 layout
size_t sf_lowrank_bytes(const SfGeom &g);
int sf_launch_lowrank(const double *lam, const int32_t *nuse, const int32_t *status, const double *alphas, const SfGeom &g,
                      double *ufrag, double *wfrag, int32_t *lrok, hipStream_t st);
// cmf_cov4.hip: the production-window covariance on the 4x4x4 fp64 MFMA
size_t sf_cov4_scratch_bytes(const SfGeom &g);
int sf_launch_cov4(const float *xt, const uint8_t *mask_t, const int32_t *nuse, const double *mu, const SfGeom &g,
                   double *cov, void *scratch, hipStream_t st);
int sf_launch_wfrag4(const double *evec, const double *d, const SfGeom &g, size_t wstride, double *wfrag, hipStream_t st);
int sf_launch_sweep4(const float *xt, const uint8_t *mask_t, const int32_t *nuse, const double *mu, const double *lam,
                     const double *wfrag, size_t wstride, const int32_t *status, const double *alphas, const SfGeom &g,
                     int nsplit, double *part, int variant, void *lr_scratch, hipStream_t st);

