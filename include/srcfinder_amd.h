/* srcfinder_amd C ABI -- the drop-in boundary for the columnwise robust matched filter (CMF).
 *
 * The reference (dsmbgu8/srcfinder) is pure Python; its hot path has no FFI of its own.  What a
 * maintainer would bind with ctypes is exactly this header: each entry point names the piece of
 * `cmf/robust_mf.py` it replaces (file:line).  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - plain C, no C++/torch types; every pointer is a DEVICE pointer owned by the caller unless it
 *     says "host"; the library never allocates or frees result buffers; scratch comes from a caller
 *     workspace sized by sf_cmf_workspace_bytes().
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); all work is
 *     enqueued asynchronously on it; nothing synchronises the device.
 *   - return value: 0 ok, <0 argument error (see sf_last_error_string), >0 a hipError_t.
 *   - per-column soft failures are data, not errors: status[col] = 0 ok, 1 no valid rows
 *     (robust_mf.py:303-304), 2 singular covariance (robust_mf.py:371-374 -> scores := 0).
 *
 * Geometry: the cube is BIL float32 [lines][bands][samples] (robust_mf.py:206-208).  A call works on
 * the column shard [s0, s1) and the active band window [b0, b0+p) (0-based; the reference's
 * 1-based inclusive [a0,a1] is b0 = a0-1, p = a1-a0+1, robust_mf.py:185-194,:298).
 */
#ifndef SRCFINDER_AMD_H
#define SRCFINDER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the ABI is exactly the names declared between this push and its pop. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define SF_MAX_ACTIVE_FUSED 96   /* largest active window of the LDS-resident statistics path */
#define SF_NALPHA_MAX 208        /* alpha grid is padded to a multiple of 16 (201 -> 208) */

int sf_version(void);
const char *sf_last_error_string(void);

/* Largest column-major scratch the fused driver needs for a shard of `ncols` columns.  For windows wider than
 * SF_MAX_ACTIVE_FUSED it includes the exact-determinant pass's scratch: a job list of ncols * nalpha entries (12 bytes each)
 * and up to 512 p x p float64 work matrices within 1 GB (740 MB at p = 425); every stream slot of a host that keeps several
 * flightlines in flight (srcfinder_amd.inflight) holds its own workspace. */
size_t sf_cmf_workspace_bytes(int lines, int p, int ncols, int nalpha);

/* Stage 1 -- column extract (robust_mf.py:298), valid-row mask (:282,:299): transposes the active
 * window of the shard into column-major float32 xt[ncols][lines][ps] (ps = p rounded up to 4) and
 * writes mask_t[ncols][lines] (1 = every active value is >= 0 and finite). */
int sf_cmf_extract_columns(const float *cube, int lines, int bands, int samples, int s0, int s1,
                           int b0, int p, float *xt, uint8_t *mask_t, void *stream);

/* Stages 2, 3 and 5 read the column-major rows `xt` as float32 (xt_f64 = 0: what stage 1 wrote) or as
 * float64 (xt_f64 = 1: the function-level looshrinkage() entry, whose input is float64 already).
 *
 * Stage 2 -- masked column mean (robust_mf.py:301-302,:347): nuse[ncols], mu[ncols][p] (float64).
 * `scratch` >= sf_cmf_workspace_bytes(). */
int sf_cmf_column_mean(const void *xt, int xt_f64, const uint8_t *mask_t, int lines, int p, int ncols,
                       int32_t *nuse, double *mu, void *scratch, void *stream);

/* Stage 3 -- sample covariance of the centred valid rows, ddof = 1 (robust_mf.py:52-70 as called at
 * :130): cov[ncols][p][p] float64, computed with fp64 MFMA. */
int sf_cmf_covariance(const void *xt, int xt_f64, const uint8_t *mask_t, const int32_t *nuse,
                      const double *mu, int lines, int p, int ncols, double *cov, void *scratch, void *stream);

/* Stage 4 -- symmetric eigendecomposition of the correlation matrix R = D^-1 S D^-1 (the restated
 * form of the 201 det+inverse calls at robust_mf.py:105-117): d[ncols][p] = sqrt(diag S),
 * lam[ncols][p], evec[ncols][p][p] (evec[c][j][:] is eigenvector j), status[ncols] (2 when a band has
 * zero or non-finite variance, 1 when nuse == 0, 3 when nuse == 1 -- resolved by stage 6). */
int sf_cmf_eigh(const double *cov, const int32_t *nuse, int p, int ncols, double *d, double *lam,
                double *evec, int32_t *status, void *scratch, void *stream);

/* Stages 3-5 in one call for windows of 97..512 bands (reflectance -R 5..420, full-band 1..425), where the
 * LDS-resident kernels do not fit: batched float64 GEMMs + a global-memory Jacobi (cmf_wide.hip).  Same outputs as
 * sf_cmf_covariance + sf_cmf_eigh + sf_cmf_loocv.  nrows = the rows the covariance is made of, nloo = the n of beta
 * and 1/(2n) (NULL: nrows) -- the function-level looshrinkage(I_zm, alphas, nll, n) passes them separately
 * (robust_mf.py:92-117).  scratch >= sf_cmf_workspace_bytes(lines, p, ncols, nalpha). */
int sf_cmf_wide_stats(const void *xt, int xt_f64, const uint8_t *mask_t, const int32_t *nrows, const int32_t *nloo,
                      const double *mu, const double *alphas, int nalpha, int lines, int p, int ncols, double *cov, double *d,
                      double *lam, double *evec, int32_t *status, double *nll, int32_t *alphaidx, void *scratch, void *stream);

/* The same with a full shrinkage target[ncols][p][p] (multimodal -f on a wide window; looshrinkage(..., I_reg) with more
 * than 96 bands -- robust_mf.py:99, :131, :354): blocked Cholesky of the target, R = L^-1 S L^-T by substitution,
 * eigenpairs of R, outputs d = diag(L), evec_j = D (L^-T v_j) as sf_cmf_eigh_general returns them; the exact
 * determinants are those of n beta S + alpha T.  status 2 when the target is not positive definite. */
int sf_cmf_wide_stats_target(const void *xt, int xt_f64, const uint8_t *mask_t, const int32_t *nrows, const int32_t *nloo,
                             const double *mu, const double *alphas, int nalpha, int lines, int p, int ncols,
                             const double *target, double *cov, double *d, double *lam, double *evec, int32_t *status,
                             double *nll, int32_t *alphaidx, void *scratch, void *stream);

/* det() over/underflow exactly as scipy's running LU-pivot product has it (robust_mf.py:111-113) for windows of up to 96
 * bands, after sf_cmf_loocv: the finite grid points next to one whose total log-determinant left the float64 range
 * (window > 0: at most `window` per crossing, in rounds of four; window <= 0: every grid point) are factorised for real
 * -- G = n beta 1e4 S + alpha 1e4 T, T = target[ncols][p][p] or diag S when target is NULL -- and marked +inf (NaN) in nll
 * where the product is 0 (not finite); alphaidx is recomputed (numpy.argmin, first NaN wins).  Wide windows do this inside
 * sf_cmf_wide_stats.  scratch >= sf_cmf_exact_det_scratch_bytes(...). */
size_t sf_cmf_exact_det_scratch_bytes(int p, int ncols, int nalpha, int window);
int sf_cmf_exact_det(const double *cov, const double *target, const int32_t *nloo, const int32_t *status, const double *alphas,
                     int nalpha, int p, int ncols, int window, double *nll, int32_t *alphaidx, void *scratch, void *stream);

/* Stage 4, full shrinkage target (multimodal -f: T = cov(I_reg), robust_mf.py:99, :131, :354) -- the same restatement one
 * congruence further: target = L L^T (Cholesky), eigendecomposition of L^-1 S L^-T, and outputs d = diag(L),
 * evec_j = D (L^-T v_j) chosen so that stages 5-7 run unchanged (they only form D^-1 evec^T and 2 sum log d = log det T).
 * r_tmp, l_tmp: caller-owned [ncols][p][p] float64 temporaries.  status 2 when the target is not positive definite. */
int sf_cmf_eigh_general(const double *cov, const double *target, const int32_t *nuse, int p, int ncols, double *r_tmp,
                        double *l_tmp, double *d, double *lam, double *evec, int32_t *status, void *scratch, void *stream);

/* Eigenpairs of ncols symmetric positive definite matrices A[ncols][p][p] of 97 .. 512 rows, as they are (no correlation scaling):
 * lam[ncols][p], evec[ncols][p][p] (row j = eigenvector j), status[ncols] (2: a non-positive or non-finite diagonal).  The wide
 * eigensolver of sf_cmf_wide_stats on a caller's matrices: scipy.linalg.eig of a covariance matrix as the reference's PCA calls it
 * with -R -k 2 (p = 416; cmf/robust_mf.py:78-84, :310-312).  scratch >= sf_cmf_eigh_wide_scratch_bytes(p, ncols). */
size_t sf_cmf_eigh_wide_scratch_bytes(int p, int ncols);
int sf_cmf_eigh_wide(const double *A, int p, int ncols, double *lam, double *evec, int32_t *status, void *scratch, void *stream);

/* Stage 5 -- leave-one-out NLL for every alpha (robust_mf.py:105-117) and its argmin (:121-127):
 * nll[ncols][nalpha] (inf where the reference's det over/underflows), alphaidx[ncols] (-1 if none). */
int sf_cmf_loocv(const void *xt, int xt_f64, const uint8_t *mask_t, const int32_t *nuse, const double *mu,
                 const double *d, const double *lam, const double *evec, const int32_t *status,
                 const double *alphas, int nalpha, int lines, int p, int ncols,
                 double *nll, int32_t *alphaidx, void *scratch, void *stream);

/* Stage 6 -- shrunk covariance C = (1-a)S + a diag(S) (robust_mf.py:130-134), solve, normalise
 * (:363,:378-381): filt[ncols][p] = scale * C^-1 t / (t^T C^-1 t), bias[ncols] = mu . filt, with
 * t = abscf*mu (radiance) or abscf-mu (reflectance != 0); scale = 1e5 or 1 (:383-386).
 * status is updated to 2 where C is singular.  A column with exactly ONE valid row (status 3 from stage 4) ends as the
 * reference leaves it: numpy.cov divides by n - 1 = 0, every NLL is NaN, argmin picks index 0 and the score is NaN --
 * alphaidx := 0, filt = bias = NaN, status := 0. */
int sf_cmf_filter(const double *mu, const double *d, const double *lam, const double *evec,
                  const double *alphas, int32_t *alphaidx, const double *abscf, int reflectance,
                  int p, int ncols, int32_t *status, double *filt, double *bias, void *stream);

/* Stage 7 -- the per-pixel matched filter score and output assembly (robust_mf.py:377-397, :266):
 * streams the BIL cube once, recomputes row validity inline, writes
 *   out[(line*out_samples + out_s0 + c)*out_bands + out_bands-1] = x.filt - bias   (valid rows)
 *                                                               = nodata          (invalid rows)
 * and, when out_bands == 4, bands 0..2 = cube[line][rgb[k]][col] as float64 for every line of every
 * column that has at least one valid row (:303-304 skips the others).  Optional: bgmeta int16
 * [lines][out_samples][2] band 1 = alpha index on valid rows (:365); colstats[3][ncols] = npix, mean,
 * std (ddof 0) of the written scores (:388-392).  This is the HBM-roofline kernel. */
int sf_cmf_score(const float *cube, int lines, int bands, int samples, int s0, int s1, int b0, int p,
                 const double *filt, const double *bias, const int32_t *status, const int32_t *alphaidx,
                 const int32_t *nuse, int rgb0, int rgb1, int rgb2, double nodata,
                 double *out, int out_samples, int out_s0, int out_bands,
                 int16_t *bgmeta, double *colstats, void *scratch, void *stream);

/* Fused driver = stages 1..7 (the body of the column loop, robust_mf.py:297-397, unimodal k = 1).
 * Host arrays: none.  Device outputs: out (see sf_cmf_score), alphaidx/nuse/status[ncols],
 * colstats[3][ncols] (may be NULL), bgmeta (may be NULL), nll_out[ncols][nalpha] (may be NULL). */
int sf_cmf_run(const float *cube, int lines, int bands, int samples, int s0, int s1, int b0, int p,
               const double *abscf, const double *alphas, int nalpha, int reflectance,
               int rgb0, int rgb1, int rgb2, double nodata,
               double *out, int out_samples, int out_s0, int out_bands,
               int32_t *alphaidx, int32_t *nuse, int32_t *status, double *colstats,
               int16_t *bgmeta, double *nll_out, void *workspace, size_t workspace_bytes, void *stream);

/* ---- multimodal background (cmf/robust_mf.py:306-332, -k > 1) ------------------------------------------------
 * The per-cluster statistics are stages 2..6 above run with the row mask  valid & (label == k)  (stage 5 takes the
 * COLUMN's valid-row count as n, as the reference passes `nuse`, :355-356).  The three entry points below are what
 * the unimodal path does not have.
 *
 * Cluster labels (:309-313): rows of a column in the top-`pcadim` whitened principal coordinates of the
 * eigenbasis of stage 4, then deterministic Lloyd iterations from along-track quantile seeds.  The reference's
 * MiniBatchKMeans is unseeded and cannot be reproduced; labels_t[ncols][lines] uint8 (255 = invalid row) is a pure
 * function of (data, k, seed).  scratch >= ncols * lines * pcadim floats.  k <= 8, pcadim <= 8. */
int sf_cmf_kmeans(const float *xt, const uint8_t *mask_t, const double *mu, const double *d, const double *lam,
                  const double *evec, int lines, int p, int ncols, int k, int pcadim, unsigned long long seed, int iters,
                  uint8_t *labels_t, void *scratch, void *stream);

/* Matched filter of ONE cluster (:377-386): rows with rowmask_t[ncols][lines] != 0 get their score in the last
 * band of `out` (0 when status == 2, :371-374) and the (cluster, alpha index) pair in bgmeta (:327, :365); every
 * other pixel is left as it is (initialise the product with sf_cmf_score and an all-zero filter first).
 * cluster == -32768 leaves the cluster band of bgmeta alone (cluster rejection, :321-327, where the pass of a
 * rejected cluster re-scores the rows of the others). */
int sf_cmf_score_cluster(const float *cube, int lines, int bands, int samples, int s0, int s1, int b0, int p,
                         const double *filt, const double *bias, const int32_t *status, const int32_t *alphaidx,
                         const uint8_t *rowmask_t, int cluster, double *out, int out_samples, int out_s0, int out_bands,
                         int16_t *bgmeta, void *stream);

/* Column statistics of the finished product over a column's valid rows (:388-392): colstats[3][ncols]. */
int sf_cmf_colstats_rows(const double *out, int out_samples, int out_s0, int out_bands, const uint8_t *mask_t, int lines,
                         int ncols, double nodata, double *colstats, void *stream);

/* Column profile of a finished product (triage/cmf_profile.py:110-140, the default non-robust statistics):
 * over the pixels of band `band` of img[lines][samples][nbands] (float64) that are not NODATA/NaN and > 0 (after the
 * float32 cast the reference applies): profile[5][samples] = npix, mean, std (ddof 0), min, max (NaN where npix = 0).
 * scratch >= ceil(lines/256) * samples * 5 doubles. */
int sf_cmf_column_profile(const double *img, int lines, int samples, int nbands, int band, double nodata,
                          double *profile, void *scratch, void *stream);

/* Robust variant (triage/cmf_profile.py:124-127, use_robust_stats): profile[5][samples] = npix, median, MAD
 * (median of |x - median|), and the (1-p) / p percentiles with numpy's 'nearest' rule (srcfinder_util.py:647-653,
 * called with p = 0.95), all on the float32 cast of the valid positive pixels.  Any number of lines (up to 32768 the column is
 * sorted in LDS, beyond that the order statistics are selected by radix passes over the product: the same numbers). */
int sf_cmf_column_profile_robust(const double *img, int lines, int samples, int nbands, int band, double nodata,
                                 double p, double *profile, void *stream);

/* Test hook: the rank-24 / 28 / 36 factorisation B = U W of the sweep's coefficient matrix (cmf_lowrank.hip), in
 * the fragment order the sweep reads: ufrag[ncols][18*9*16], wfrag[ncols][13*9*64], lrok[ncols] (0 full rank,
 * 1 rank 28, 2 rank 36, 3 rank 24). */
int sf_debug_lowrank(const double *lam, const int32_t *nuse, const int32_t *status, const double *alphas,
                     int nalpha, int p, int ncols, double *ufrag, double *wfrag, int32_t *lrok, void *stream);

/* Test hook of the wide windows' rank factorisation (csrc/cmf_wlr.hip; 257..432 bands): wlr_out[c] = 0 when column c's sweep
 * runs unfactored, else the rank (24..31) of the factorisation of its coefficient matrix (cmf/robust_mf.py:105-117 restated).
 * scratch: sf_debug_wlr_bytes(ncols) bytes of device memory. */
size_t sf_debug_wlr_bytes(int ncols);
int sf_debug_wlr(const double *lam, const int32_t *nloo, const int32_t *status, const double *alphas, int nalpha, int p,
                 int ncols, void *scratch, int32_t *wlr_out, void *stream);

/* Test / tuning hook: phase clocks of the tiles of the fused wide-window sweep (cmf_wgemm.hip), accumulated while
 * sf_debug_set(22, 1) is on: out8 = tiles, Y = X~ W, r = Z C + rows, then (k_wsweep8) the r
 * phase's MFMAs, row reductions, first barrier, exchange + second barrier, 0 */
int sf_debug_wsweep_stamps(unsigned long long *out8, int reset);
/* test entry of the tridiagonal preconditioner of the wide-window eigensolver (csrc/cmf_wtri.hip; the eigendecomposition that
 * replaces the 201 det / inv of cmf/robust_mf.py:105-117 on windows of more than 96 bands): R [nb][p][p] symmetric positive
 * definite, Lc [nb][p][p] its lower Cholesky factor in column-major order -> F [nb][p][p] column-major with F F^T = R and nearly
 * orthogonal columns, tlam [nb][p] the tridiagonal route's eigenvalues, pflag [nb] 0 where the preconditioner was applied */
/* phase clocks of the tridiagonalisation's workgroup 0: out8 = cycles in the reflector, the symv, the corrections, the trailing
 * update, then the number of columns */
int sf_debug_wtri_stamps(unsigned long long *out8, int reset);
/* phase clocks of the blocked LU of the exact-determinant pass, workgroup 0: out8 = cycles in panel load, panel factorisation,
 * permutation + triangular solve, rank-16 update, then the number of factorisations.  The clocks are OFF unless asked for:
 * reset = 1 zeroes them and turns them on, reset = 2 (after the read into out8) zeroes them and turns them off */
int sf_debug_lu_stamps(unsigned long long *out8, int reset);
size_t sf_debug_wtri_scratch_bytes(int p, int nb);
int sf_debug_wtri(const double *R, const double *Lc, int p, int nb, double *F, double *tlam, int32_t *pflag, void *scratch,
                  void *stream);

/* Timing hook for bench.py's roofline line: while enabled, every sf_cmf_score launch (direct or
 * inside sf_cmf_run) is bracketed by a fresh pair of HIP events on the launch stream.
 * sf_cmf_score_timing_read() synchronises those events, returns the summed kernel time and the
 * number of launches since the last enable, and clears the list. */
int sf_cmf_score_timing(int enable);
/* Tuning knobs for experiments (tools/tune_*.py; the keys are listed at struct SfTune in csrc/sf_tune.h); the
 * defaults are the shipped choices.  The knobs are PER CALLING THREAD (the library holds no process-wide mutable
 * state): a knob set on one thread does not reach launches issued from another -- a host that fans work out over
 * threads copies them with sf_debug_get / sf_debug_set (srcfinder_amd.cnn._predict_multi_gpu does) -- and calls that
 * have to agree on a layout derived from a knob (the score launch and the column statistics that read its partial
 * sums, keys 1 / 2) must be issued from the same thread.  The reference has no counterpart (plain Python globals). */
int sf_debug_set(int key, int value);
int sf_debug_get(int key, int *value);
int sf_cmf_score_timing_read(double *total_ms, int *launches);


/* scipy.linalg.det / inv as the reference wraps them (cmf/robust_mf.py:72-76, :86-90) for a batch of n x n float64
 * matrices (row-major, device): LU with partial pivoting.  det = the running product of the pivots in index order (a
 * prefix that reaches inf or 0 stays there: what log(det) and the det == 0 test of looshrinkage see, :111-113), 0 for
 * an exactly singular matrix.  inv: info[b] > 0 = exactly singular (scipy raises LinAlgError; the column loop catches
 * it, :371).  work: batch * n * n doubles; piv: batch * n int32. */
int sf_linalg_det(const double *A, int n, int batch, double *work, double *det, void *stream);
int sf_linalg_inv(const double *A, int n, int batch, double *work, int32_t *piv, double *Ainv, int32_t *info, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Spectrometer masks on the resident cube (spectrometer_masks/masks_sds.py) and the image primitives they share with
 * the saliency -> detections step (salience_predictions.py): binary dilations, 8-connected component labelling.
 * Masks are uint8 [lines][samples] (0 / 1) device arrays owned by the caller.
 * ------------------------------------------------------------------------------------------------------------ */

/* Per-pixel rules over the BIL cube (masks_sds.py:133-232, :330): sat = any band of [sat_b0, sat_b1) > sat_thr;
 * cloud = x[cloud_b0] > cloud_thr and the slope cloud_b0 -> cloud_b1 is negative (dwl = wavelength[cloud_b1] -
 * wavelength[cloud_b0]; the second slope of :222 never enters the result, :230); spec = sat and x[vis_band] > vis_thr;
 * dark = x[dark_band] < dark_thr and not <= -9999; grow = sat and x[grow_band] < vis_thr (:314); border = x[0] == -9999. */
int sf_masks_pixel(const float *cube, int lines, int bands, int samples, int sat_b0, int sat_b1, float sat_thr, int cloud_b0,
                   int cloud_b1, float cloud_thr, float dwl, int vis_band, float vis_thr, int dark_band, float dark_thr,
                   int grow_band, uint8_t *sat, uint8_t *cloud, uint8_t *spec, uint8_t *dark, uint8_t *grow,
                   uint8_t *border, void *stream);
/* `iterations` passes of the 4-neighbour binary dilation, in place (tmp: a second H x W buffer) -- dilate_mask,
 * masks_sds.py:252-273 (skimage.morphology.binary_dilation's default structuring element, background border). */
int sf_image_dilate_cross(uint8_t *mask, uint8_t *tmp, int H, int W, int iterations, void *stream);
/* Binary dilation by skimage.morphology.disk(radius) = {x^2 + y^2 <= r^2} (masks_sds.py:280, :317). */
size_t sf_image_dilate_disk_scratch_bytes(int H, int W, int radius);
int sf_image_dilate_disk(const uint8_t *src, uint8_t *dst, int H, int W, int radius, void *scratch, void *stream);
/* 8-connected (skimage connectivity=2) component labelling: labels[H][W] int32, 0 = background, components numbered
 * 1..n in raster order of their first pixels (measure.label, masks_sds.py:309; srcfinder_util.imlabel,
 * salience_predictions.py:61); area (optional): pixels per id, area_cap >= n + 1 entries; *ncomp_dev = n. */
size_t sf_image_label8_scratch_bytes(int H, int W);
int sf_image_label8(const uint8_t *mask, int H, int W, int32_t *labels, int32_t *area, int area_cap, int32_t *ncomp_dev,
                    void *scratch, void *stream);
/* Clear the pixels of `sel` that lie in components of fewer than minarea pixels (region.area >= mingrowarea, :310). */
int sf_image_filter_small_components(const int32_t *labels, const int32_t *area, int minarea, uint8_t *sel, int H, int W,
                                     void *stream);
/* Product assembly (:336-341): out[lines][samples][4] int16 = cloud, specular, flare (2 where flare_buffer, 1 where sat
 * and not specular; flare_buffer NULL = no grow radius: 0), dark; all four -9999 where border. */
int sf_masks_compose(const uint8_t *cloud, const uint8_t *spec, const uint8_t *sat, const uint8_t *flare_buffer,
                     const uint8_t *dark, const uint8_t *border, int lines, int samples, int16_t *out, void *stream);

/* Saliency map -> detections, the region loop of salience_predictions.py:66-108 (salience2detections).  labels[H][W]:
 * 1..nregions from sf_image_label8 of (salience > threshold, :60-61); sal[H][W] float32; cmf[H][W][cmf_nb] float64
 * product, CMF in band cmf_band; nodata[H][W] uint8 (RGB band 0 == -9999, :45).  Per region, over pmsk = (label == id)
 * & ~nodata for the saliency and pmsk & (cmf > cmfthr) for the CMF: max, min, median, MAD (median of |x - median|),
 * truncated centre of mass of the pixels holding the maximum (:81-103).  bbox: scratch (nregions + 1) * 4 int32.
 * rec[nregions + 1][20] float64, row id: [0..3] bounding slices (row start, row stop, col start, col stop),
 * [4..10] saliency max, min, median, MAD, max row, max col, n; [11..17] the same for the CMF; [18] status, always 0 (a region
 * of any size: up to 32768 saliency / 16384 CMF values are sorted in LDS, larger sets are selected by radix passes). */
int sf_detect_region_stats(const int32_t *labels, int H, int W, int nregions, const float *sal, const double *cmf,
                           int cmf_nb, int cmf_band, const uint8_t *nodata, double cmfthr, int32_t *bbox, double *rec,
                           void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * CNN tile scorer (cnn/cnn_pred_pipeline.py + cnn/archs/googlenet1.py, eval graph).  Activations are NHWC
 * float32 device buffers owned by the caller; conv weights are [Cout][k*k][Cin] float32 with BatchNorm
 * (eps 1e-3, running statistics) folded in by the caller (srcfinder_amd/cnn.py does it on upload).
 * ------------------------------------------------------------------------------------------------------------ */

/* ClampCH4 -> Normalize -> Pad(dim/2, dim/2, dim/2-1, dim/2-1) (cnn_pred_pipeline.py:19-30, :39-47, :126-157):
 * plane[H][W] -> padded[H+dim-1][W+dim-1]. */
int sf_cnn_prepare_plane(const float *plane, int H, int W, float vmin, float vmax, float mean, float stdv, int dim,
                         float *padded, void *stream);

/* FlightlineConvolve.__getitem__ (cnn_pred_pipeline.py:53-58) fused with conv1 = BasicConv2d(1, 64, 7x7, stride 2,
 * pad 3) (googlenet1.py:60, :266-275): tiles tile0 .. tile0+ntiles-1 (row-major pixel index) of the padded plane
 * -> out[ntiles][128][128][64].  w = [64][49], bias = [64]. */
int sf_cnn_conv1(const float *padded, int Hp, int Wp, int W, long long tile0, int ntiles, const float *w,
                 const float *bias, float *out, void *stream);
/* The same fused with maxpool1 = MaxPool2d(3, stride 2, ceil_mode) (googlenet1.py:61): out[ntiles][64][64][64]; the
 * 128 x 128 x 64 conv1 activation (4 MB per tile) never leaves the CU (implicit GEMM on the fp32 matrix cores). */
int sf_cnn_conv1_pool(const float *padded, int Hp, int Wp, int W, long long tile0, int ntiles, const float *w,
                      const float *bias, float *out, void *stream);

/* nn.MaxPool2d(ksize, stride, pad, ceil_mode=True) (googlenet1.py:61,:64,:68,:75,:213); Ho/Wo are the caller's
 * ceil-mode output sizes; edge windows are clipped to the input. */
int sf_cnn_maxpool(const float *in, int N, int H, int W, int C, int ksize, int stride, int pad, float *out, int Ho,
                   int Wo, void *stream);

/* BasicConv2d with a 1x1 or 3x3 (pad 1) stride-1 kernel (googlenet1.py:266-275): conv + folded-BN bias + ReLU,
 * written into channels [ch_off, ch_off+Cout) of an output whose pixel stride is ld_out floats -- the inception
 * concat (googlenet1.py:223-228) costs nothing.  Implicit GEMM on v_mfma_f32_32x32x2_f32. */
int sf_cnn_conv(const float *in, int N, int H, int W, int Cin, int ld_in, const float *w, const float *bias, int Cout,
                int ksize, float *out, int ld_out, int ch_off, void *stream);

/* 3 x 3 stride-1 pad-1 convolution + folded-BatchNorm bias + ReLU by Winograd's minimal filtering F(2 x 2, 3 x 3)
 * (BasicConv2d, cnn/archs/googlenet1.py:266-275; conv3 and the inception branches' 3 x 3 convolutions, :62-78, :184-228): the same
 * float32 arithmetic class as sf_cnn_conv (float32 operands, float32 accumulate on the fp32 matrix cores) with 16 instead of 36
 * multiplications per 2 x 2 output block and channel pair; results agree with sf_cnn_conv to rounding (a few 1e-7 relative).
 * sf_cnn_wino_ok: the geometries served (square images of 8 or a multiple of 16 pixels, Cin >= 16 and a multiple of 8).
 * sf_cnn_wino_weights: U[16][Cout][Cin] = G g G^T of the folded weights w[Cout][9][Cin] (sf_cnn_wino_weight_floats floats; once per
 * weight upload).  sf_cnn_conv3x3_wino: arguments as sf_cnn_conv with U in place of w. */
int sf_cnn_wino_ok(int H, int W, int Cin);
size_t sf_cnn_wino_weight_floats(int Cout, int Cin);
int sf_cnn_wino_weights(const float *w, int Cout, int Cin, float *U, void *stream);
int sf_cnn_conv3x3_wino(const float *in, int N, int H, int W, int Cin, int ld_in, const float *U, const float *bias, int Cout,
                        float *out, int ld_out, int ch_off, void *stream);

/* The same float32 convolutions by OPERAND SPLITTING on the float16 matrix cores (csrc/cnn_split.hip): a = a_hi + a_lo (two fp16
 * halves = 22 mantissa bits), a w = a_hi w_hi + a_hi w_lo + a_lo w_hi as three fp16 MFMAs with float32 accumulation.  Same
 * tolerance class as sf_cnn_conv (errors against float64 of a few 1e-7 relative, as the fp32 kernel's; the reference goldens at
 * 1e-4) at 1.5-2.6 x its speed -- NOT the fp16-storage option (sf_cnn_*_f16).  sf_cnn_split_weights: folded weights
 * w[Cout][K = k*k*Cin] -> hi[Cout*K], lo[Cout*K] (float16) of w 2^e(co) and wscale[Cout] = 2^-e(co) (a power of two per output
 * channel that keeps the low halves normal; once per weight upload).  sf_cnn_conv_split / sf_cnn_conv_split3_split: arguments
 * as sf_cnn_conv / sf_cnn_conv_split3 with (hi, lo, wscale) in place of w and ascale, a power of two the layer's input
 * activations are (fp32 input) or were (split-format input: by their producer's oscale) multiplied by; the epilogue divides it out
 * exactly.  The two ends of float16's range are part of the contract.  UNDERFLOW: below |a ascale| = 0.125 an activation's low half
 * is subnormal (absolute error 2^-25) -- sf_cnn_calibrate measures every split layer's largest input on a sample of the plane's
 * windows and returns the powers of two that put it at 2^9..2^10.  OVERFLOW: an activation with |a ascale| >= 65504 has no
 * float16 -- the launch stores 1 into `overflow`, a device int OF THE CALLER (one per call, batch or stream; the library keeps no
 * flag of its own, so concurrent streams and threads cannot see each other's), and the caller repeats that work with sf_cnn_conv /
 * sf_cnn_conv3x3_wino (sf_cnn_score_rows and srcfinder_amd.cnn do it themselves).
 * in_split / out_split / out12_split: the tensor is in the SPLIT FORMAT instead of float32 -- per pixel and
 * 8-channel group sixteen float16, the eight high halves then the eight low halves, in the bytes of the dense float32 tensor
 * ([M][C/8][2][8]; C a multiple of 8, ld == C, no channel offset), scaled by oscale (= the consumer's ascale).  A tensor only
 * split-operand convolutions read (conv2's output, the 3 x 3 reducers' outputs of an Inception block) is written that way by its
 * producer's epilogue and fetched without conversion by its consumers: the split is done once per value, not once per (value,
 * tap, channel tile).  sf_cnn_absmax: max |x| of n floats folded into *amax (a device float the caller zeroed; order-independent).
 * BasicConv2d / Inception -- googlenet1.py:184-228, :266-275. */
int sf_cnn_split_weights(const float *w, int Cout, int K, void *hi, void *lo, float *wscale, void *stream);
int sf_cnn_conv_split(const float *in, int in_split, int N, int H, int W, int Cin, int ld_in, const void *whi, const void *wlo,
                      const float *wscale, const float *bias, int Cout, int ksize, float ascale, float *out, int out_split, float oscale,
                      int ld_out, int ch_off, int *overflow, void *stream);
int sf_cnn_conv_split3_split(const float *in, int N, int H, int W, int Cin, int ld_in, const void *whi, const void *wlo,
                             const float *wscale, const float *bias, int c0, int c1, int c2, float ascale, float *out0, int ld0,
                             int off0, float *out1, int ld1, int off1, float *out2, int ld2, int off2, int out12_split, float oscale1,
                             float oscale2, int *overflow, void *stream);
int sf_cnn_absmax(const float *x, size_t n, float *amax, void *stream);
/* branch4 of an Inception (MaxPool2d(3, 1, 1, ceil_mode) + 1x1 BasicConv2d, googlenet1.py:213-214) by operand splitting: arguments as
 * sf_cnn_pool_conv with (hi, lo, wscale); dense input, image width dividing 128 (sf_cnn_pool_conv_split_ok; else sf_cnn_pool_conv). */
int sf_cnn_pool_conv_split_ok(int N, int H, int W, int Cin, int Cout);
int sf_cnn_pool_conv_split(const float *in, int N, int H, int W, int Cin, const void *whi, const void *wlo, const float *wscale,
                           const float *bias, int Cout, float ascale, float *out, int ld_out, int ch_off, int *overflow, void *stream);

/* Inception branch 4 (googlenet1.py:213-214) in one call: MaxPool2d(3, stride 1, pad 1, ceil_mode) into pooled_scratch
 * (N*H*W*Cin floats), then the 1x1 BasicConv2d.  `in` is dense ([N][H][W][Cin]) and NON-NEGATIVE (a concatenation of
 * ReLU outputs, as every inception input is): the pool kernel relies on 0 being the identity of max.  (A form that takes the pool inside
 * the convolution's tile fetch exists behind sf_debug_set(18, 1); it reads the tile nine times through the L1 and
 * measured 3 % slower end to end.) */
int sf_cnn_pool_conv(const float *in, int N, int H, int W, int Cin, int ld_in, const float *w, const float *bias, int Cout,
                     float *out, int ld_out, int ch_off, float *pooled_scratch, void *stream);

/* The three 1x1 BasicConv2d that read the same inception input (branch1, branch2[0], branch3[0];
 * googlenet1.py:199-210) as ONE GEMM: w = [c0+c1+c2][Cin] (the three folded weight sets stacked), output channels
 * [0,c0) -> out0 (+off0, stride ld0), [c0,c0+c1) -> out1, the rest -> out2. */
int sf_cnn_conv_split3(const float *in, int N, int H, int W, int Cin, int ld_in, const float *w, const float *bias,
                       int c0, int c1, int c2, float *out0, int ld0, int off0, float *out1, int ld1, int off1,
                       float *out2, int ld2, int off2, void *stream);

/* AdaptiveAvgPool2d(1) -> Linear(C, 2) -> softmax[:,1] (googlenet1.py:87-89,:156-161; cnn_pred_pipeline.py:177-180)
 * and the NODATA rule (:185-189): out[tile0 + t] = plane[tile0 + t] == nodata ? nodata : p.  plane may be NULL. */
int sf_cnn_head(const float *in, int ntiles, int HW, int C, const float *fcw, const float *fcb, const float *plane,
                long long tile0, float nodata, float *out, void *stream);

/* The whole tile scorer in one call (cnn_pred_pipeline.py:159-189 around googlenet1.py's eval graph): scores the image
 * rows [r0, r1) of the H x W plane -- one 256 x 256 window per pixel of the padded plane [H+255][W+255] from
 * sf_cnn_prepare_plane -- in batches of `batch` windows and writes out[row * W + col] (float32; NODATA where plane is
 * NODATA, plane may be NULL).  blob: the BatchNorm-folded float32 weights in ONE array of sf_cnn_blob_floats() values:
 * for conv1, conv2, conv3, then per inception block {branch1|branch2.0|branch3.0 stacked, branch2.1, branch3.1,
 * branch4.1}, then fc: weights [Cout][k*k][Cin] followed by the bias [Cout].  workspace >=
 * sf_cnn_score_workspace_bytes(batch) (activations of one batch + the Winograd / split-operand forms of the weights + the overflow
 * slots).  All launches are enqueued on `stream`.
 * route (an ARGUMENT of the call -- nothing process- or thread-wide selects the arithmetic):
 *   0  operand splitting on the fp16 matrix cores (sf_cnn_conv_split: the float32 tolerance class; the product's default) with the
 *      trunk through inception3b SHARED between the overlapping windows (below: sf_cnn_ring_pool1 ...; since the second half of round 6
 *      also the ring rows that see only a window's top / bottom padding, from strip maps the call builds per 16 image rows -- the
 *      side-row forms of the ring kernels are internal, csrc/cnn_internal.h); 5 = shared through conv3
 *      only (round 6's first form); 3 = every window evaluated on its own (round 5's form; bit-identical to the kernels sequenced
 *      one batch at a time).  All three
 *      ends of float16's range are handled inside the call: `scales` = the sf_cnn_num_scales() per-layer activation scales (HOST
 *      floats, powers of two, e.g. from sf_cnn_calibrate) or NULL -- the call then calibrates itself on a fixed sample of the
 *      plane's windows (a function of the plane alone: every row range and batch size of a flightline gets the same scales, hence
 *      the same bits); every batch owns an overflow slot in the workspace, the slots are read back every 1024 batches and a batch
 *      that raised its slot is scored AGAIN on route 4 before the call returns (info, if given: int[2] -- [0] the batches scored again,
 *      [1] the batches that ran on the shared trunk; a strip whose phase maps leave float16's range runs unshared).  The call
 *      therefore synchronises `stream` before returning on this route;
 *   4  Winograd F(2 x 2, 3 x 3) + fp32 implicit GEMM on the fp32 matrix cores;   2 (1)  the direct fp32 kernel for everything.
 * sf_cnn_calibrate: the scales alone (host array of sf_cnn_num_scales() floats: [0] maxpool1's output, [1] conv2's output,
 * [2 + 3 i ...] inception block i's input, its 3 x 3 reducer's output, its "5 x 5" reducer's output); synchronises `stream`. */
size_t sf_cnn_blob_floats(void);
size_t sf_cnn_score_workspace_bytes(int batch, int H, int W);   /* H = W = 0: without the trunk-sharing buffers (routes 3, 4, 2, 1; sf_cnn_calibrate) */
int sf_cnn_num_scales(void);
int sf_cnn_calibrate(const float *padded, int H, int W, const float *blob, int batch, void *workspace, size_t workspace_bytes,
                     float *scales, void *stream);
int sf_cnn_score_rows(const float *padded, const float *plane, int H, int W, int r0, int r1, const float *blob, float *out,
                      int batch, int route, const float *scales, int *info, void *workspace, size_t workspace_bytes, void *stream);

/* Trunk sharing (csrc/cnn_share.hip, csrc/cnn_ring.h): cnn_pred_pipeline.py:53-58 scores one 256 x 256 window per pixel, so
 * neighbouring windows overlap by 255/256.  A layer's activation for one window lives on a G x G grid and depends on the window only
 * through its zero padding: with the layer's FRAME (lo, hi), rows / columns 0 .. lo - 1 and G - hi .. G - 1 -- the RING -- hold values
 * only this window has, the interior equals the layer evaluated fully convolutionally on the whole padded plane at the window's
 * phase (the phase MAPS, built once per strip of image rows with the FCN kernels).  Ring positions are enumerated: the lo top rows,
 * the hi bottom rows (G positions each), then for the rows between them the lo left and the hi right columns
 * (count = (lo + hi) G + (G - lo - hi)(lo + hi)).  Window t of a batch = image pixel ((tile0 + t) / W, (tile0 + t) % W); with P = 2^shift
 * phases per axis its phase is (r & (P - 1)) P + (c & (P - 1)) and its origin in that map ((r >> shift) - Rb, c >> shift).
 * Grid 64, shift 2 (googlenet1.py:60-64): maxpool1 / conv2 frame (1, 1), conv3 (2, 2).  Grid 32, shift 3 (:64-68): maxpool2 (1, 2),
 * inception3a (2, 3), inception3b (3, 4); maxpool3 assembles inception4a's whole 16 x 16 input.
 *   sf_cnn_phase_canvas  canvas[Hc][Wc] = padded[y0 + u][x0 + v] (zero outside): the plane shifted by a phase
 *   sf_cnn_ring_pool1    maxpool1(conv1(window)) at the 252 ring positions of frame (1, 1) on the 64 grid: out[ntiles][252][64]
 *   sf_cnn_conv_ring     a 1 x 1 / 3 x 3 BasicConv2d (operand splitting) at the ring positions of the frame (olo, ohi), its input
 *                        gathered from `maps` = the input's phase maps [P P][Hq][Wq][Cin] with the batch's ring tensor
 *                        [N][count(G, ilo, ihi)][Cin] ring_off floats behind their start (both float32, or both in the split format):
 *                        rows m = window * count(G, olo, ohi) + ring index; output segments as sf_cnn_conv_split3_split
 *                        (c1 = c2 = 0: one)
 *   sf_cnn_pool_gather   MaxPool2d(3, stride 1 pad 1 | stride 2 ceil_mode) of such a tensor, on the whole output grid (olo < 0:
 *                        out[N][Go][Go][C]) or at the ring positions of (olo, ohi) on Go (out[N][count][C])
 * sf_cnn_score_rows (routes 0 / 5) sequences all of it. */
int sf_cnn_phase_canvas(const float *padded, int Hp, int Wp, int y0, int x0, int Hc, int Wc, float *canvas, void *stream);
int sf_cnn_ring_pool1(const float *padded, int Hp, int Wp, int W, long long tile0, int ntiles, const float *w, const float *bias,
                      float *out, void *stream);
int sf_cnn_conv_ring(const float *maps, int in_split, long long tile0, int N, int W, int Rb, int Hq, int Wq, size_t ring_off, int shift,
                     int G, int ilo, int ihi, int olo, int ohi, int Cin, const void *whi, const void *wlo, const float *wscale,
                     const float *bias, int c0, int c1, int c2, int ksize, float ascale, float *out0, int ld0, int off0, float *out1,
                     int ld1, int off1, float *out2, int ld2, int off2, int out12_split, float oscale1, float oscale2, int *overflow,
                     void *stream);
int sf_cnn_pool_gather(const float *maps, long long tile0, int N, int W, int Rb, int Hq, int Wq, size_t ring_off, int shift, int G,
                       int ilo, int ihi, int C, int stride, int Go, int olo, int ohi, float *out, void *stream);

/* FCN shift-and-stitch, the reference's approximate fast mode (cnn/fcn_pred_pipeline.py).
 * sf_cnn_fcn_prepare: ClampCH4 + Normalize of the plane, embedded at (top, left) = divmod(shift, scale) in a zero canvas
 *   out[nshift][Hc][Wc], Hc = H + (scale - H % scale) + scale (FlightlineShiftStitch, :32-65).
 * sf_cnn_conv1_image: conv1 (7x7 s2 p3, folded BN + ReLU) over whole images img[N][Hc][Wc] -> NHWC [N][Ho][Wo][64]
 *   (float32, or float16 when out_f16); the rest of the trunk is sf_cnn_maxpool / sf_cnn_conv / sf_cnn_conv_split3,
 *   the 1x1 head (:158-160) is sf_cnn_head with HW = 1.
 * sf_cnn_fcn_stitch: stitch_stack (:67-92) + the NODATA mask (:249) for the prediction maps pred[nshift][Hq][Wq] of
 *   shifts shift0..: writes their pixels of out[H][W]. */
int sf_cnn_fcn_prepare(const float *plane, int H, int W, float vmin, float vmax, float mean, float stdv, int scale,
                       int shift0, int nshift, int Hc, int Wc, float *out, void *stream);
int sf_cnn_conv1_image(const float *img, int N, int Hc, int Wc, const float *w, const float *bias, void *out, int out_f16,
                       void *stream);
int sf_cnn_fcn_stitch(const float *pred, int nshift, int shift0, int scale, int Hq, int Wq, const float *plane, int H, int W,
                      float nodata, float *out, void *stream);

/* ---- reduced-precision option of the CNN scorer (NOT the parity path): float16 activations/weights, fp32 accumulate on
 * v_mfma_f32_32x32x16_f16 -- the precision class of the reference's own cuDNN-TF32 default on recent GPUs.  Same
 * operators, arguments as the fp32 entry points; activation / weight pointers are float16 (passed as void*). */
int sf_cnn_conv1_f16(const float *padded, int Hp, int Wp, int W, long long tile0, int ntiles, const float *w,
                     const float *bias, void *out, void *stream);
int sf_cnn_maxpool_f16(const void *in, int N, int H, int W, int C, int ksize, int stride, int pad, void *out, int Ho,
                       int Wo, void *stream);
int sf_cnn_conv_f16(const void *in, int N, int H, int W, int Cin, int ld_in, const void *w, const float *bias, int Cout,
                    int ksize, void *out, int ld_out, int ch_off, void *stream);
int sf_cnn_conv_split3_f16(const void *in, int N, int H, int W, int Cin, int ld_in, const void *w, const float *bias,
                           int c0, int c1, int c2, void *out0, int ld0, int off0, void *out1, int ld1, int off1,
                           void *out2, int ld2, int off2, void *stream);
int sf_cnn_head_f16(const void *in, int ntiles, int HW, int C, const float *fcw, const float *fcb, const float *plane,
                    long long tile0, float nodata, float *out, void *stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* SRCFINDER_AMD_H */
