// Multimodal background (cmf/robust_mf.py:306-332, -k > 1): the pieces the unimodal pipeline does not have.
//
// The per-cluster statistics run through the ordinary stage kernels with the row mask  valid & (label == k)
// (host side: srcfinder_amd/cmf.py); this file holds
//   k_pca_project / k_kmeans   cluster labels: the rows of a column in the top-`pcadim` whitened principal
//                              coordinates (the eigenbasis stage 4 computed anyway), deterministic Lloyd
//                              iterations from along-track quantile seeds.  The reference's clustering is an
//                              UNSEEDED MiniBatchKMeans on an unsorted general eig (:310-313) and cannot be
//                              reproduced; parity of everything downstream is tested with injected labels.
//   k_score_cluster            matched filter of ONE cluster: writes the score and the (cluster, alpha index)
//                              pair of the rows in the mask, touches nothing else (:327, :365, :377-386)
//   k_colstats_rows            npix / mean / std of a column's valid rows from the finished image (:388-392)
#include "cmf_common.h"

namespace {

constexpr int KM_MAXK = 8, KM_MAXD = 8, KM_NT = 1024;

// y[c][row][m] = sum_b (x[row][b] - mu_b) / d_b * V[j_m][b]   (float; invalid rows are left untouched)
__global__ __launch_bounds__(256) void k_pca_project(const float *__restrict__ xt, const uint8_t *__restrict__ mask_t,
                                                      const double *__restrict__ mu, const double *__restrict__ d,
                                                      const double *__restrict__ lam, const double *__restrict__ evec,
                                                      int L, int p, int PS, int pcadim, float *__restrict__ y) {
  extern __shared__ double wsm[];   // [pcadim][p] weights, then [p] mean
  __shared__ int top[KM_MAXD];
  const int c = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) {   // indices of the pcadim largest eigenvalues, descending (ties: lower index first)
    for (int m = 0; m < pcadim; ++m) {
      int best = -1;
      for (int j = 0; j < p; ++j) {
        bool used = false;
        for (int q = 0; q < m; ++q) used |= (top[q] == j);
        if (!used && (best < 0 || lam[(size_t)c * p + j] > lam[(size_t)c * p + best])) best = j;
      }
      top[m] = best;
    }
  }
  __syncthreads();
  for (int i = tid; i < pcadim * p; i += 256) {
    const int m = i / p, b = i - m * p;
    wsm[i] = evec[((size_t)c * p + top[m]) * p + b] / d[(size_t)c * p + b];
  }
  for (int b = tid; b < p; b += 256) wsm[pcadim * p + b] = mu[(size_t)c * p + b];
  __syncthreads();
  const double *mus = wsm + pcadim * p;
  for (int row = blockIdx.y * 256 + tid; row < L; row += 256 * gridDim.y) {
    if (!mask_t[(size_t)c * L + row]) continue;
    const float *xp = xt + ((size_t)c * L + row) * PS;
    double acc[KM_MAXD];
#pragma unroll
    for (int m = 0; m < KM_MAXD; ++m) acc[m] = 0.0;
    for (int b = 0; b < p; ++b) {
      const double xv = (double)xp[b] - mus[b];
#pragma unroll
      for (int m = 0; m < KM_MAXD; ++m)
        if (m < pcadim) acc[m] = __builtin_fma(xv, wsm[m * p + b], acc[m]);
    }
    float *yo = y + ((size_t)c * L + row) * pcadim;
#pragma unroll
    for (int m = 0; m < KM_MAXD; ++m)
      if (m < pcadim) yo[m] = (float)acc[m];
  }
}

// One workgroup per column.  Seeds: the valid rows of rank (i + 0.5 + jitter) n / k along the track, jitter in
// (-0.25, 0.25) from a hash of (seed, column, i).  Then `iters` Lloyd iterations (or until no label changes),
// all sums reduced in a fixed order: the labels are a pure function of (data, k, seed).
__device__ __forceinline__ uint64_t km_mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(KM_NT) void k_kmeans(const float *__restrict__ y, const uint8_t *__restrict__ mask_t, int L,
                                                   int dim, int k, uint64_t seed, int iters, uint8_t *__restrict__ labels_t) {
  __shared__ double cen[KM_MAXK][KM_MAXD];
  __shared__ double part[KM_NT / 64][KM_MAXK][KM_MAXD + 1];
  __shared__ int cnt[KM_NT];
  __shared__ int seedrow[KM_MAXK];
  __shared__ int changed;
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint8_t *mp = mask_t + (size_t)c * L;
  const float *yc = y + (size_t)c * L * dim;
  uint8_t *lab = labels_t + (size_t)c * L;
  // ---- rank of every thread's contiguous slab of rows among the valid rows
  const int per = (L + KM_NT - 1) / KM_NT;
  const int r0 = tid * per, r1 = min(L, r0 + per);
  int mine = 0;
  for (int r = r0; r < r1; ++r) mine += mp[r] ? 1 : 0;
  cnt[tid] = mine;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i < KM_NT; ++i) { const int v = cnt[i]; cnt[i] = run; run += v; }
    changed = run;   // = number of valid rows
  }
  __syncthreads();
  const int nvalid = changed;
  for (int r = r0 + 0; r < r1; ++r) lab[r] = 255;   // invalid rows (and everything when the column is empty)
  if (nvalid == 0) return;
  const int kk = min(k, nvalid);
  if (tid < kk) {
    const double u = (double)(km_mix(seed ^ km_mix(((uint64_t)c << 8) | (uint64_t)tid)) >> 11) * (1.0 / 9007199254740992.0);
    int rank = (int)(((double)tid + 0.5 + 0.5 * (u - 0.5)) * (double)nvalid / (double)kk);
    seedrow[tid] = rank < 0 ? 0 : (rank >= nvalid ? nvalid - 1 : rank);
  }
  __syncthreads();
  {   // the thread whose slab holds rank seedrow[i] publishes that row as centre i
    int run = cnt[tid];
    for (int r = r0; r < r1; ++r) {
      if (!mp[r]) continue;
      for (int i = 0; i < kk; ++i)
        if (seedrow[i] == run)
          for (int m = 0; m < dim; ++m) cen[i][m] = (double)yc[(size_t)r * dim + m];
      ++run;
    }
  }
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    double s[KM_MAXK][KM_MAXD + 1];
#pragma unroll
    for (int i = 0; i < KM_MAXK; ++i)
#pragma unroll
      for (int m = 0; m <= KM_MAXD; ++m) s[i][m] = 0.0;
    int chg = 0;
    if (tid == 0) changed = 0;
    __syncthreads();
    for (int r = tid; r < L; r += KM_NT) {
      if (!mp[r]) continue;
      double v[KM_MAXD];
#pragma unroll
      for (int m = 0; m < KM_MAXD; ++m) v[m] = (m < dim) ? (double)yc[(size_t)r * dim + m] : 0.0;
      int best = 0;
      double bd = 1.7976931348623157e308;
#pragma unroll
      for (int i = 0; i < KM_MAXK; ++i) {
        if (i < kk) {
          double dd = 0.0;
#pragma unroll
          for (int m = 0; m < KM_MAXD; ++m) {
            const double e = v[m] - ((m < dim) ? cen[i][m] : 0.0);
            dd = __builtin_fma(e, e, dd);
          }
          if (dd < bd) { bd = dd; best = i; }   // ties: the lower cluster id
        }
      }
      if (lab[r] != (uint8_t)best) { lab[r] = (uint8_t)best; chg = 1; }
#pragma unroll
      for (int i = 0; i < KM_MAXK; ++i) {
        if (i == best) {
#pragma unroll
          for (int m = 0; m < KM_MAXD; ++m) s[i][m] += v[m];
          s[i][KM_MAXD] += 1.0;
        }
      }
    }
    if (chg) changed = 1;   // benign race
    // ---- fixed-order reduction: lanes (xor butterfly is order-symmetric), then waves 0..15 in order
#pragma unroll
    for (int i = 0; i < KM_MAXK; ++i) {
      if (i < kk) {
#pragma unroll
        for (int m = 0; m <= KM_MAXD; ++m) {
          double a = s[i][m];
          for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
          if (lane == 0) part[wave][i][m] = a;
        }
      }
    }
    __syncthreads();
    if (tid < kk * (KM_MAXD + 1)) {
      const int i = tid / (KM_MAXD + 1), m = tid - i * (KM_MAXD + 1);
      double a = 0.0;
      for (int w = 0; w < KM_NT / 64; ++w) a += part[w][i][m];
      part[0][i][m] = a;
    }
    __syncthreads();
    if (tid < kk * KM_MAXD) {
      const int i = tid / KM_MAXD, m = tid - i * KM_MAXD;
      const double n = part[0][i][KM_MAXD];
      if (n > 0.0 && m < dim) cen[i][m] = part[0][i][m] / n;   // an emptied cluster keeps its centre
    }
    __syncthreads();
    if (!changed) break;
  }
}

// rowmask_t[c][l] != 0: score the pixel with this cluster's filter and stamp (cluster, alpha index)
// WGL: windows too wide for a [p][64] float64 LDS tile read the lane's filter vector from global memory (L1/L2 resident)
template <bool WGL>
__global__ __launch_bounds__(256) void k_score_cluster(const float *__restrict__ cube, int L, int B, int C, int s0, int Cs,
                                                        int b0, int p, const double *__restrict__ filt,
                                                        const double *__restrict__ bias, const int32_t *__restrict__ status,
                                                        const int32_t *__restrict__ alphaidx,
                                                        const uint8_t *__restrict__ rowmask_t, int cluster,
                                                        double *__restrict__ out, int oS, int os0, int ob,
                                                        int16_t *__restrict__ bgmeta, int lines_per_wg) {
  extern __shared__ double ws[];   // [p][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int colbase = blockIdx.x * 64;
  const int ncol = min(64, Cs - colbase);
  const bool colok = lane < ncol;
  const int col = colbase + (colok ? lane : ncol - 1);
  if (!WGL) {
    for (int idx = tid; idx < 64 * p; idx += 256) {
      const int cl = idx / p, b = idx - cl * p;
      ws[b * 64 + cl] = (cl < ncol) ? filt[(size_t)(colbase + cl) * p + b] : 0.0;
    }
  }
  const double *wg = filt + (size_t)col * p;
  const double mybias = bias[col];
  const int st = status[col], ai = alphaidx[col];
  __syncthreads();
  const int lbeg = blockIdx.y * lines_per_wg, lend = min(L, lbeg + lines_per_wg);
  for (int l = lbeg + wave; l < lend; l += 4) {
    if (!colok || st == 1 || !rowmask_t[(size_t)col * L + l]) continue;
    const float *xp = cube + ((size_t)l * B + b0) * C + s0 + col;
    double acc = 0.0;
    for (int b = 0; b < p; ++b) acc = __builtin_fma((double)xp[(size_t)b * C], WGL ? wg[b] : ws[b * 64 + lane], acc);
    const size_t pix = (size_t)l * oS + os0 + col;
    out[pix * ob + (ob - 1)] = (st == 2) ? 0.0 : (acc - mybias);   // singular C: the mode's rows get 0 (:373)
    if (bgmeta) {
      if (cluster != -32768) bgmeta[pix * 2] = (int16_t)cluster;   // -32768: the caller owns the cluster band (-r)
      if (st == 0) bgmeta[pix * 2 + 1] = (int16_t)ai;
    }
  }
}

__global__ __launch_bounds__(256) void k_colstats_rows(const double *__restrict__ out, int oS, int os0, int ob,
                                                        const uint8_t *__restrict__ mask_t, int L, double nodata,
                                                        double *__restrict__ colstats, int Cs) {
  __shared__ double red[256][3];
  const int c = blockIdx.x, tid = threadIdx.x;
  double n = 0, s1 = 0, s2 = 0;
  for (int l = tid; l < L; l += 256) {
    if (!mask_t[(size_t)c * L + l]) continue;
    const double v = out[((size_t)l * oS + os0 + c) * ob + (ob - 1)];
    n += 1.0; s1 += v; s2 += v * v;
  }
  red[tid][0] = n; red[tid][1] = s1; red[tid][2] = s2;
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < 256; ++i) { n += red[i][0]; s1 += red[i][1]; s2 += red[i][2]; }
    if (n > 0) {
      const double mean = s1 / n;
      double var = s2 / n - mean * mean;
      colstats[c] = n; colstats[Cs + c] = mean; colstats[2 * Cs + c] = sqrt(var < 0 ? 0 : var);
    } else {
      colstats[c] = nodata; colstats[Cs + c] = nodata; colstats[2 * Cs + c] = nodata;   // skipped column (:293-295)
    }
  }
}

}  // namespace

extern "C" {

int sf_cmf_kmeans(const float *xt, const uint8_t *mask_t, const double *mu, const double *d, const double *lam,
                  const double *evec, int lines, int p, int ncols, int k, int pcadim, unsigned long long seed, int iters,
                  uint8_t *labels_t, void *scratch, void *stream) {
  if (!xt || !mask_t || !mu || !d || !lam || !evec || !labels_t || !scratch || k < 1 || k > KM_MAXK || pcadim < 1 ||
      pcadim > KM_MAXD || pcadim > p || lines < 1 || ncols < 1) {
    sf_set_error("sf_cmf_kmeans: bad argument (k <= %d, pcadim <= %d)", KM_MAXK, KM_MAXD);
    return -1;
  }
  hipStream_t st = (hipStream_t)stream;
  const int PS = (p + 3) / 4 * 4;
  float *y = reinterpret_cast<float *>(scratch);   // [ncols][lines][pcadim]
  const size_t lds = ((size_t)pcadim * p + p) * sizeof(double);
  hipLaunchKernelGGL(k_pca_project, dim3(ncols, 8), dim3(256), lds, st, xt, mask_t, mu, d, lam, evec, lines, p, PS, pcadim, y);
  SF_LAUNCH_CHECK("k_pca_project");
  hipLaunchKernelGGL(k_kmeans, dim3(ncols), dim3(KM_NT), 0, st, y, mask_t, lines, pcadim, k, (uint64_t)seed, iters, labels_t);
  SF_LAUNCH_CHECK("k_kmeans");
  return 0;
}

int sf_cmf_score_cluster(const float *cube, int lines, int bands, int samples, int s0, int s1, int b0, int p,
                         const double *filt, const double *bias, const int32_t *status, const int32_t *alphaidx,
                         const uint8_t *rowmask_t, int cluster, double *out, int out_samples, int out_s0, int out_bands,
                         int16_t *bgmeta, void *stream) {
  const int ncols = s1 - s0;
  if (!cube || !filt || !bias || !status || !alphaidx || !rowmask_t || !out || ncols < 1 || s0 < 0 || s1 > samples ||
      b0 < 0 || b0 + p > bands || (out_bands != 1 && out_bands != 4)) {
    sf_set_error("sf_cmf_score_cluster: bad argument");
    return -1;
  }
  const int lpw = 64;
  const size_t lds = (size_t)p * 64 * sizeof(double);
  const dim3 grid(sf_cdiv(ncols, 64), sf_cdiv(lines, lpw));
  if (lds > 100 * 1024) {
    hipLaunchKernelGGL(k_score_cluster<true>, grid, dim3(256), 0, (hipStream_t)stream, cube, lines, bands, samples, s0, ncols,
                       b0, p, filt, bias, status, alphaidx, rowmask_t, cluster, out, out_samples, out_s0, out_bands, bgmeta, lpw);
  } else {
    if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_score_cluster<false>), lds)) return rc;
    hipLaunchKernelGGL(k_score_cluster<false>, grid, dim3(256), lds, (hipStream_t)stream, cube, lines, bands, samples, s0,
                       ncols, b0, p, filt, bias, status, alphaidx, rowmask_t, cluster, out, out_samples, out_s0, out_bands,
                       bgmeta, lpw);
  }
  SF_LAUNCH_CHECK("k_score_cluster");
  return 0;
}

int sf_cmf_colstats_rows(const double *out, int out_samples, int out_s0, int out_bands, const uint8_t *mask_t, int lines,
                         int ncols, double nodata, double *colstats, void *stream) {
  if (!out || !mask_t || !colstats || lines < 1 || ncols < 1) {
    sf_set_error("sf_cmf_colstats_rows: bad argument");
    return -1;
  }
  hipLaunchKernelGGL(k_colstats_rows, dim3(ncols), dim3(256), 0, (hipStream_t)stream, out, out_samples, out_s0, out_bands,
                     mask_t, lines, nodata, colstats, ncols);
  SF_LAUNCH_CHECK("k_colstats_rows");
  return 0;
}

}  // extern "C"
