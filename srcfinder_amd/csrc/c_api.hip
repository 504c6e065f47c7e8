// C ABI of libsrcfinder_amd.so (declared in include/srcfinder_amd.h): argument checks, workspace carving,
// stage sequencing.  No device allocation, no synchronisation, no process-wide mutable state: the error string, the
// optional score-kernel timing list and the tuning knobs of sf_debug_set are all per calling thread.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "cmf_common.h"

namespace {
thread_local char g_err[512] = "";
thread_local SfTune g_tune;

struct TimedLaunch {
  hipEvent_t a, b;
};
thread_local bool g_timing = false;
thread_local std::vector<TimedLaunch> *g_timed = nullptr;

// persistent (cross-stage) buffers of the fused driver, carved from the caller's workspace
struct Carve {
  char *base;
  size_t off;
  void *take(size_t bytes) {
    void *p = base ? base + off : nullptr;
    off += sf_align(bytes);
    return p;
  }
};
struct Ws {
  float *xt;
  uint8_t *mask_t;
  double *mu, *cov, *d, *lam, *evec, *filt, *bias, *nll;
  void *scratch;
  size_t scratch_bytes, total;
};

Ws carve(void *base, const SfGeom &g) {
  Carve c{reinterpret_cast<char *>(base), 0};
  Ws w;
  const size_t nc = g.ncols, p = g.p;
  w.xt = (float *)c.take(nc * g.lines * g.ps * sizeof(float));
  w.mask_t = (uint8_t *)c.take(nc * g.lines);
  w.mu = (double *)c.take(nc * p * sizeof(double));
  w.cov = (double *)c.take(nc * p * p * sizeof(double));
  w.d = (double *)c.take(nc * p * sizeof(double));
  w.lam = (double *)c.take(nc * p * sizeof(double));
  w.evec = (double *)c.take(nc * p * p * sizeof(double));
  w.filt = (double *)c.take(nc * p * sizeof(double));
  w.bias = (double *)c.take(nc * sizeof(double));
  w.nll = (double *)c.take(nc * g.nalpha * sizeof(double));
  size_t s = sf_mean_scratch_bytes(g);
  if (g.p <= SF_MAX_ACTIVE_FUSED) {
    if (sf_extract_sum_bytes(g) > s) s = sf_extract_sum_bytes(g);
    if (sf_cov_scratch_bytes(g) > s) s = sf_cov_scratch_bytes(g);
    if (sf_eigh_scratch_bytes(g) > s) s = sf_eigh_scratch_bytes(g);
    if (sf_loocv_scratch_bytes(g) > s) s = sf_loocv_scratch_bytes(g);
    if (sf_exact_det_scratch_bytes(g, SF_NARROW_DET_WINDOW) > s) s = sf_exact_det_scratch_bytes(g, SF_NARROW_DET_WINDOW);
  } else {
    if (sf_wide_scratch_bytes(g) > s) s = sf_wide_scratch_bytes(g);
    if (sf_extract_sum_bytes(g) > s) s = sf_extract_sum_bytes(g);   // (the fused column sums of the 416- / 425-band windows)
  }
  if (sf_score_scratch_bytes(g.lines, g.ncols) > s) s = sf_score_scratch_bytes(g.lines, g.ncols);
  w.scratch_bytes = s;
  w.scratch = c.take(s);
  w.total = c.off;
  return w;
}

int check_geom(int lines, int bands, int samples, int s0, int s1, int b0, int p) {
  if (lines < 1 || bands < 1 || samples < 1) { sf_set_error("empty cube (%d x %d x %d)", lines, bands, samples); return -1; }
  if (s0 < 0 || s1 > samples || s1 <= s0) { sf_set_error("bad column shard [%d,%d) of %d samples", s0, s1, samples); return -1; }
  if (b0 < 0 || p < 1 || b0 + p > bands) { sf_set_error("bad active window [%d,%d) of %d bands", b0, b0 + p, bands); return -1; }
  return 0;
}

// The streaming score kernel, optionally bracketed by HIP events on its own stream (bench.py roofline);
// the column statistics are finalised by a second, tiny kernel outside the bracket.
int timed_score(const float *cube, int lines, int bands, int samples, int s0, int ncols, int b0, int p,
                const double *filt, const double *bias, const int32_t *status, const int32_t *alphaidx,
                const int32_t *nuse, int rgb0, int rgb1, int rgb2, double nodata, double *out, int out_samples,
                int out_s0, int out_bands, int16_t *bgmeta, double *colstats, void *scratch, hipStream_t st) {
  TimedLaunch tl{};
  const bool timed = g_timing;
  if (timed) {
    SF_HIP(hipEventCreate(&tl.a));
    SF_HIP(hipEventCreate(&tl.b));
  }
  int rc = sf_launch_score(cube, lines, bands, samples, s0, ncols, b0, p, filt, bias, status, alphaidx, rgb0, rgb1, rgb2,
                           nodata, out, out_samples, out_s0, out_bands, bgmeta, scratch, colstats != nullptr, st,
                           timed ? tl.a : nullptr, timed ? tl.b : nullptr);
  if (timed) {
    if (rc) { (void)hipEventDestroy(tl.a); (void)hipEventDestroy(tl.b); return rc; }
    if (!g_timed) g_timed = new std::vector<TimedLaunch>();
    g_timed->push_back(tl);
  }
  if (rc) return rc;
  if (colstats) rc = sf_launch_colstats(scratch, lines, samples, s0, ncols, p, nuse, status, nodata, colstats, st);
  return rc;
}
}  // namespace

SfTune &sf_tune() { return g_tune; }

void sf_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int sf_fail_hip(hipError_t e, const char *what) {
  sf_set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
  return (int)e > 0 ? (int)e : 1;
}

#include <map>
#include <mutex>
#include <utility>
int sf_lds_attr(const void *fn, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void *>, size_t> have;
  int dev = 0;
  SF_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  size_t &cur = have[std::make_pair(dev, fn)];
  if (bytes > cur) {
    SF_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    cur = bytes;
  }
  return 0;
}

extern "C" {

int sf_version(void) { return 100; }
const char *sf_last_error_string(void) { return g_err; }

size_t sf_cmf_workspace_bytes(int lines, int p, int ncols, int nalpha) {
  if (lines < 1 || p < 1 || ncols < 1 || nalpha < 1) return 0;
  return carve(nullptr, sf_geom(lines, p, ncols, nalpha)).total;
}

int sf_cmf_extract_columns(const float *cube, int lines, int bands, int samples, int s0, int s1, int b0, int p,
                           float *xt, uint8_t *mask_t, void *stream) {
  if (int rc = check_geom(lines, bands, samples, s0, s1, b0, p)) return rc;
  if (!cube || !xt || !mask_t) { sf_set_error("null pointer"); return -1; }
  return sf_launch_extract(cube, lines, bands, samples, s0, s1 - s0, b0, p, xt, mask_t, nullptr, nullptr,
                           (hipStream_t)stream);
}

int sf_cmf_column_mean(const void *xt, int xt_f64, const uint8_t *mask_t, int lines, int p, int ncols, int32_t *nuse,
                       double *mu, void *scratch, void *stream) {
  if (!xt || !mask_t || !nuse || !mu || !scratch) { sf_set_error("null pointer"); return -1; }
  return sf_launch_mean(xt, xt_f64, mask_t, sf_geom(lines, p, ncols, 1), nuse, mu, scratch, (hipStream_t)stream);
}

int sf_cmf_covariance(const void *xt, int xt_f64, const uint8_t *mask_t, const int32_t *nuse, const double *mu, int lines,
                      int p, int ncols, double *cov, void *scratch, void *stream) {
  if (!xt || !mask_t || !nuse || !mu || !cov || !scratch) { sf_set_error("null pointer"); return -1; }
  return sf_launch_cov(xt, xt_f64, mask_t, nuse, mu, sf_geom(lines, p, ncols, 1), cov, scratch, (hipStream_t)stream);
}

int sf_cmf_eigh(const double *cov, const int32_t *nuse, int p, int ncols, double *d, double *lam, double *evec,
                int32_t *status, void *scratch, void *stream) {
  if (!cov || !nuse || !d || !lam || !evec || !status || !scratch) { sf_set_error("null pointer"); return -1; }
  return sf_launch_eigh(cov, nuse, sf_geom(1, p, ncols, 1), d, lam, evec, status, scratch, (hipStream_t)stream);
}

int sf_cmf_loocv(const void *xt, int xt_f64, const uint8_t *mask_t, const int32_t *nuse, const double *mu, const double *d,
                 const double *lam, const double *evec, const int32_t *status, const double *alphas, int nalpha,
                 int lines, int p, int ncols, double *nll, int32_t *alphaidx, void *scratch, void *stream) {
  if (!xt || !mask_t || !nuse || !mu || !d || !lam || !evec || !status || !alphas || !alphaidx || !scratch) {
    sf_set_error("null pointer");
    return -1;
  }
  return sf_launch_loocv(xt, xt_f64, mask_t, nuse, mu, d, lam, evec, status, alphas, sf_geom(lines, p, ncols, nalpha), nll,
                         alphaidx, scratch, (hipStream_t)stream);
}

int sf_cmf_wide_stats(const void *xt, int xt_f64, const uint8_t *mask_t, const int32_t *nrows, const int32_t *nloo,
                      const double *mu, const double *alphas, int nalpha, int lines, int p, int ncols, double *cov, double *d,
                      double *lam, double *evec, int32_t *status, double *nll, int32_t *alphaidx, void *scratch, void *stream) {
  if (!xt || !mask_t || !nrows || !mu || !alphas || !cov || !d || !lam || !evec || !status || !nll || !alphaidx || !scratch) {
    sf_set_error("null pointer");
    return -1;
  }
  if (lines < 1 || p < 1 || ncols < 1 || nalpha < 1) { sf_set_error("sf_cmf_wide_stats: bad geometry"); return -1; }
  return sf_launch_wide_stats(xt, xt_f64, mask_t, nrows, nloo, mu, alphas, sf_geom(lines, p, ncols, nalpha), cov, d, lam, evec,
                              status, nll, alphaidx, scratch, (hipStream_t)stream);
}

int sf_cmf_wide_stats_target(const void *xt, int xt_f64, const uint8_t *mask_t, const int32_t *nrows, const int32_t *nloo,
                             const double *mu, const double *alphas, int nalpha, int lines, int p, int ncols,
                             const double *target, double *cov, double *d, double *lam, double *evec, int32_t *status,
                             double *nll, int32_t *alphaidx, void *scratch, void *stream) {
  if (!xt || !mask_t || !nrows || !mu || !alphas || !target || !cov || !d || !lam || !evec || !status || !nll || !alphaidx ||
      !scratch) {
    sf_set_error("null pointer");
    return -1;
  }
  if (lines < 1 || p < 1 || ncols < 1 || nalpha < 1) { sf_set_error("sf_cmf_wide_stats_target: bad geometry"); return -1; }
  return sf_launch_wide_stats(xt, xt_f64, mask_t, nrows, nloo, mu, alphas, sf_geom(lines, p, ncols, nalpha), cov, d, lam, evec,
                              status, nll, alphaidx, scratch, (hipStream_t)stream, target);
}

int sf_cmf_filter(const double *mu, const double *d, const double *lam, const double *evec, const double *alphas,
                  int32_t *alphaidx, const double *abscf, int reflectance, int p, int ncols, int32_t *status,
                  double *filt, double *bias, void *stream) {
  if (!mu || !d || !lam || !evec || !alphas || !alphaidx || !abscf || !status || !filt || !bias) {
    sf_set_error("null pointer");
    return -1;
  }
  return sf_launch_filter(mu, d, lam, evec, alphas, alphaidx, abscf, reflectance, sf_geom(1, p, ncols, 1), status, filt,
                          bias, (hipStream_t)stream);
}

int sf_cmf_score(const float *cube, int lines, int bands, int samples, int s0, int s1, int b0, int p,
                 const double *filt, const double *bias, const int32_t *status, const int32_t *alphaidx,
                 const int32_t *nuse, int rgb0, int rgb1, int rgb2, double nodata, double *out, int out_samples,
                 int out_s0, int out_bands, int16_t *bgmeta, double *colstats, void *scratch, void *stream) {
  if (int rc = check_geom(lines, bands, samples, s0, s1, b0, p)) return rc;
  if (!cube || !filt || !bias || !status || !alphaidx || !nuse || !out) { sf_set_error("null pointer"); return -1; }
  if (nodata > 0) { sf_set_error("nodata value=%f > 0, values will not be masked", nodata); return -3; }  // robust_mf.py:232-234
  if (out_bands != 1 && out_bands != 4) { sf_set_error("out_bands must be 1 or 4 (got %d)", out_bands); return -1; }
  if (out_bands == 4 && (rgb0 < 0 || rgb1 < 0 || rgb2 < 0 || rgb0 >= bands || rgb1 >= bands || rgb2 >= bands)) {
    sf_set_error("rgb band out of range");
    return -1;
  }
  if (out_s0 < 0 || out_s0 + (s1 - s0) > out_samples) { sf_set_error("output column window out of range"); return -1; }
  if (colstats && !scratch) { sf_set_error("colstats needs scratch"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  return timed_score(cube, lines, bands, samples, s0, s1 - s0, b0, p, filt, bias, status, alphaidx, nuse, rgb0, rgb1,
                     rgb2, nodata, out, out_samples, out_s0, out_bands, bgmeta, colstats, scratch, st);
}

int sf_cmf_run(const float *cube, int lines, int bands, int samples, int s0, int s1, int b0, int p,
               const double *abscf, const double *alphas, int nalpha, int reflectance, int rgb0, int rgb1, int rgb2,
               double nodata, double *out, int out_samples, int out_s0, int out_bands, int32_t *alphaidx, int32_t *nuse,
               int32_t *status, double *colstats, int16_t *bgmeta, double *nll_out, void *workspace,
               size_t workspace_bytes, void *stream) {
  if (int rc = check_geom(lines, bands, samples, s0, s1, b0, p)) return rc;
  if (!cube || !abscf || !alphas || !out || !alphaidx || !nuse || !status || !workspace) {
    sf_set_error("null pointer");
    return -1;
  }
  if (nodata > 0) { sf_set_error("nodata value=%f > 0, values will not be masked", nodata); return -3; }
  if (p > 512) {
    sf_set_error("active window of %d bands exceeds the wide statistics path (max 512)", p);
    return -2;
  }
  const int ncols = s1 - s0;
  const SfGeom g = sf_geom(lines, p, ncols, nalpha);
  const Ws w = carve(workspace, g);
  if (w.total > workspace_bytes) {
    sf_set_error("workspace too small: need %zu bytes, got %zu", w.total, workspace_bytes);
    return -4;
  }
  hipStream_t st = (hipStream_t)stream;
  int rc;
  double *nll = nll_out ? nll_out : w.nll;
  if (p > SF_MAX_ACTIVE_FUSED) {  // wide window (e.g. reflectance 5..420): batched-GEMM statistics path
    if (sf_extract_fuses_sum(p)) {   // 1..425 and the -R window: the column sums ride along with the transpose (round 5)
      if ((rc = sf_launch_extract_fused(cube, lines, bands, samples, s0, b0, g, w.xt, w.mask_t, w.scratch, st))) return rc;
      if ((rc = sf_launch_mean_from_partials(g, nuse, w.mu, w.scratch, st))) return rc;
    } else {
      if ((rc = sf_launch_extract(cube, lines, bands, samples, s0, ncols, b0, p, w.xt, w.mask_t, nullptr, nullptr, st)))
        return rc;
      if ((rc = sf_launch_mean(w.xt, 0, w.mask_t, g, nuse, w.mu, w.scratch, st))) return rc;
    }
    if ((rc = sf_launch_wide_stats(w.xt, 0, w.mask_t, nuse, nullptr, w.mu, alphas, g, w.cov, w.d, w.lam, w.evec, status, nll,
                                   alphaidx, w.scratch, st)))
      return rc;
  } else {
    if (sf_extract_fuses_sum(p)) {  // column sums ride along with the transpose (per-chunk partials in scratch)
      if ((rc = sf_launch_extract_fused(cube, lines, bands, samples, s0, b0, g, w.xt, w.mask_t, w.scratch, st))) return rc;
      if ((rc = sf_launch_mean_from_partials(g, nuse, w.mu, w.scratch, st))) return rc;
    } else {
      if ((rc = sf_launch_extract(cube, lines, bands, samples, s0, ncols, b0, p, w.xt, w.mask_t, nullptr, nullptr, st)))
        return rc;
      if ((rc = sf_launch_mean(w.xt, 0, w.mask_t, g, nuse, w.mu, w.scratch, st))) return rc;
    }
    if ((rc = sf_launch_cov(w.xt, 0, w.mask_t, nuse, w.mu, g, w.cov, w.scratch, st))) return rc;
    if ((rc = sf_launch_eigh(w.cov, nuse, g, w.d, w.lam, w.evec, status, w.scratch, st))) return rc;
    if ((rc = sf_launch_loocv(w.xt, 0, w.mask_t, nuse, w.mu, w.d, w.lam, w.evec, status, alphas, g, nll, alphaidx, w.scratch,
                              st)))
      return rc;
    // det() at the edge of the float64 range (a column of fewer valid rows than bands): the finite grid points next to a lost
    // one are factorised for real, robust_mf.py:111-113 (linalg.hip; nine small launches that find nothing to do otherwise)
    if (sf_tune().det_variant != 2 &&
        (rc = sf_launch_exact_det(w.cov, nuse, status, alphas, g, SF_NARROW_DET_WINDOW, nullptr, nll, alphaidx, w.scratch, st)))
      return rc;
  }
  if ((rc = sf_launch_filter(w.mu, w.d, w.lam, w.evec, alphas, alphaidx, abscf, reflectance, g, status, w.filt, w.bias, st)))
    return rc;
  return timed_score(cube, lines, bands, samples, s0, ncols, b0, p, w.filt, w.bias, status, alphaidx, nuse, rgb0, rgb1,
                     rgb2, nodata, out, out_samples, out_s0, out_bands, bgmeta, colstats, w.scratch, st);
}


static int *sf_tune_slot(int key) {
  SfTune &t = sf_tune();
  switch (key) {
    case 1: return &t.score_variant;
    case 2: return &t.score_lpw;
    case 3: return &t.score_xcd;
    case 4: return &t.sweep_variant;
    case 5: return &t.cov_variant;
    case 6: return &t.extract_variant;
    case 7: return &t.eigh_lpp;
    case 8: return &t.sweep4r_waves;
    case 10: return &t.wide_eigh_variant;
    case 14: return &t.lu_variant;
    case 15: return &t.det_variant;
    case 16: return &t.cnn_variant;
    case 17: return &t.cnn_conv_variant;
    case 18: return &t.cnn_pool_variant;
    case 19: return &t.extract_nt;
    case 20: return &t.sweep4_form;
    case 21: return &t.sweep_grid;
    case 22: return &t.wjac_stamps;
    case 24: return &t.wsweep_variant;
    case 26: return &t.det_slots;
    default: return nullptr;
  }
}
int sf_debug_set(int key, int value) {
  int *slot = sf_tune_slot(key);
  if (!slot) { sf_set_error("sf_debug_set: unknown key %d", key); return -1; }
  *slot = value;
  return 0;
}
int sf_debug_get(int key, int *value) {
  int *slot = sf_tune_slot(key);
  if (!slot || !value) { sf_set_error("sf_debug_get: unknown key %d", key); return -1; }
  *value = *slot;
  return 0;
}

int sf_cmf_score_timing(int enable) {
  if (g_timed) {
    for (auto &t : *g_timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    g_timed->clear();
  }
  g_timing = enable != 0;
  return 0;
}

int sf_cmf_score_timing_read(double *total_ms, int *launches) {
  double tot = 0.0;
  int n = 0;
  if (g_timed) {
    for (auto &t : *g_timed) {
      SF_HIP(hipEventSynchronize(t.b));
      float ms = 0.f;
      SF_HIP(hipEventElapsedTime(&ms, t.a, t.b));
      tot += ms;
      ++n;
      (void)hipEventDestroy(t.a);
      (void)hipEventDestroy(t.b);
    }
    g_timed->clear();
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = n;
  return 0;
}

}  // extern "C"
