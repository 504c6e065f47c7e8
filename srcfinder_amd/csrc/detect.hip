// Saliency map -> plume detections (first half): per-region statistics of the thresholded saliency
// map and of the CMF product under it.
//
// Replaces the region loop of salience_predictions.py:25-150 (salience2detections): salmask = salience > salthr,
// 8-connected labelling (srcfinder_util.imlabel: skimage label, connectivity 2 -> sf_image_label8 in masks.hip),
// bounding boxes (scipy find_objects, :66), and per region
//   pmsk = (label == id) & ~nodata                      ppix = salience[pmsk]   (float32)        :81-90
//   cmsk = pmsk & (cmf > cmfthr)                        cpix = cmf[cmsk]        (float64 here)   :96-103
//   median, MAD = median(|x - median|) (srcfinder_util.mad with medval), min, max (extrema), and the truncated centre
//   of mass of the pixels that hold the maximum (np.int32(center_of_mass(img * mask == max)) + bbox origin).
// One workgroup per region: the region's values are gathered from its bounding box into LDS, bitonic-sorted, the order
// statistics read off (numpy's median of an even count = mean of the two middle values in the array's dtype).  A region
// with more pixels than the LDS holds (32768 saliency / 16384 CMF values; the reference has no cap,
// salience_predictions.py:81-103) takes the same order statistics by radix selection on the values' bit patterns, byte by
// byte from the top, re-reading the bounding box each pass: exact, so both routes give the same numbers.
#include "cmf_common.h"

namespace {

constexpr int DT_NT = 1024;
constexpr int DT_CAP32 = 32768, DT_CAP64 = 16384;   // LDS-resident sort: pixels per region

__global__ __launch_bounds__(256) void k_region_bbox(const int32_t *__restrict__ labels, int H, int W, int32_t *__restrict__ bbox) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  const int id = labels[(size_t)y * W + x];
  if (id <= 0) return;
  int32_t *b = bbox + (size_t)id * 4;
  atomicMin(&b[0], y); atomicMax(&b[1], y + 1); atomicMin(&b[2], x); atomicMax(&b[3], x + 1);   // slices: start, stop
}
__global__ void k_bbox_init(int32_t *bbox, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  bbox[i * 4 + 0] = 0x7fffffff; bbox[i * 4 + 1] = 0; bbox[i * 4 + 2] = 0x7fffffff; bbox[i * 4 + 3] = 0;
}

template <typename T>
__device__ __forceinline__ void bitonic(T *a, int npow2, int tid) {
  for (int k = 2; k <= npow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npow2; i += DT_NT) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const T x = a[i], y = a[ixj];
          const bool up = (i & k) == 0;
          if ((x > y) == up) { a[i] = y; a[ixj] = x; }
        }
      }
      __syncthreads();
    }
  }
}

// order-preserving integer key of a float / double (negative values: all bits flipped; others: the sign bit set)
__device__ __forceinline__ unsigned long long dt_key(float v) {
  const unsigned b = __float_as_uint(v);
  return (unsigned long long)((b & 0x80000000u) ? ~b : (b | 0x80000000u));
}
__device__ __forceinline__ unsigned long long dt_key(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
template <typename T>
__device__ __forceinline__ T dt_unkey(unsigned long long k) {
  if constexpr (sizeof(T) == 4) {
    const unsigned b = (unsigned)k;
    return __uint_as_float((b & 0x80000000u) ? (b & 0x7fffffffu) : ~b);
  } else {
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
  }
}
// the k-th smallest (0-based) of the set {f(val(y, x)) : sel(y, x)} over the bounding box, f = identity (mode 0) or |v - med|
template <typename T, typename Sel, typename Val>
__device__ T region_select(const int32_t *bb, Sel sel, Val val, int mode, T med, unsigned k, int tid) {
  __shared__ unsigned hist[256];
  __shared__ unsigned long long s_prefix;
  __shared__ unsigned s_k;
  const int y0 = bb[0], x0 = bb[2], bw = bb[3] - bb[2], bn = (bb[1] - bb[0]) * bw;
  unsigned long long prefix = 0, mask = 0;
  for (int shift = 8 * (int)sizeof(T) - 8; shift >= 0; shift -= 8) {
    for (int i = tid; i < 256; i += DT_NT) hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < bn; i += DT_NT) {
      const int y = y0 + i / bw, x = x0 + i % bw;
      if (sel(y, x)) {
        T v = val(y, x);
        if (mode) v = v > med ? v - med : med - v;
        const unsigned long long key = dt_key(v);
        if ((key & mask) == prefix) atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
      }
    }
    __syncthreads();
    if (tid == 0) {
      unsigned cum = 0, bsel = 255;
      for (unsigned b = 0; b < 256; ++b) {
        if (k < cum + hist[b]) { bsel = b; break; }
        cum += hist[b];
      }
      s_prefix = prefix | ((unsigned long long)bsel << shift);
      s_k = k - cum;
    }
    __syncthreads();
    prefix = s_prefix;
    k = s_k;
    mask |= 255ull << shift;
    __syncthreads();
  }
  return dt_unkey<T>(prefix);
}

// statistics of one value set of one region.  sel(y, x) says whether the pixel belongs to the set; val(y, x) its value.
// out: max, min, median, mad, maxrow, maxcol, n.  Always true since round 5 (a set larger than the LDS sort is selected from
// global memory); the status word of the record stays for the ABI.
template <typename T, typename Sel, typename Val>
__device__ bool region_stats(T *buf, int cap, const int32_t *bb, Sel sel, Val val, double *out, int tid) {
  __shared__ int cnt;
  __shared__ unsigned long long srow, scol;
  __shared__ int smax_n;
  __shared__ T smed;
  const int y0 = bb[0], y1 = bb[1], x0 = bb[2], x1 = bb[3];
  const int bw = x1 - x0, bn = (y1 - y0) * bw;
  if (tid == 0) { cnt = 0; srow = 0; scol = 0; smax_n = 0; }
  __syncthreads();
  for (int i = tid; i < bn; i += DT_NT) {
    const int y = y0 + i / bw, x = x0 + i % bw;
    if (sel(y, x)) {
      const int k = atomicAdd(&cnt, 1);
      if (k < cap) buf[k] = val(y, x);
    }
  }
  __syncthreads();
  const int n = cnt;
  const double nanv = __builtin_nan("");
  if (n == 0) {
    if (tid == 0) { out[0] = out[1] = out[2] = out[3] = out[4] = out[5] = nanv; out[6] = 0; }
    __syncthreads();
    return true;
  }
  T vmin, vmax, med, madv;
  if (n > cap) {
    // more values than the LDS sort holds: the same order statistics by radix selection over the bounding box
    vmin = region_select<T>(bb, sel, val, 0, (T)0, 0u, tid);
    vmax = region_select<T>(bb, sel, val, 0, (T)0, (unsigned)(n - 1), tid);
    const T hi = region_select<T>(bb, sel, val, 0, (T)0, (unsigned)(n / 2), tid);
    med = (n & 1) ? hi : (T)((region_select<T>(bb, sel, val, 0, (T)0, (unsigned)(n / 2 - 1), tid) + hi) * (T)0.5);
    const T mhi = region_select<T>(bb, sel, val, 1, med, (unsigned)(n / 2), tid);
    madv = (n & 1) ? mhi : (T)((region_select<T>(bb, sel, val, 1, med, (unsigned)(n / 2 - 1), tid) + mhi) * (T)0.5);
  } else {
    int npow2 = 1;
    while (npow2 < n) npow2 <<= 1;
    const T inf = (T)__builtin_inf();
    for (int i = n + tid; i < npow2; i += DT_NT) buf[i] = inf;
    __syncthreads();
    bitonic(buf, npow2, tid);
    vmin = buf[0];
    vmax = buf[n - 1];
    if (tid == 0) smed = (n & 1) ? buf[n / 2] : (T)((buf[n / 2 - 1] + buf[n / 2]) * (T)0.5);
    __syncthreads();
    med = smed;
    __syncthreads();
    for (int i = tid; i < n; i += DT_NT) buf[i] = buf[i] > med ? buf[i] - med : med - buf[i];
    __syncthreads();
    bitonic(buf, npow2, tid);
    madv = (n & 1) ? buf[n / 2] : (T)((buf[n / 2 - 1] + buf[n / 2]) * (T)0.5);
  }
  // centre of mass of the pixels of the bounding box whose masked value equals the maximum (img * mask == max)
  for (int i = tid; i < bn; i += DT_NT) {
    const int y = y0 + i / bw, x = x0 + i % bw;
    if (sel(y, x) && val(y, x) == vmax) {
      atomicAdd(&srow, (unsigned long long)(y - y0));
      atomicAdd(&scol, (unsigned long long)(x - x0));
      atomicAdd(&smax_n, 1);
    }
  }
  __syncthreads();
  if (tid == 0) {
    out[0] = (double)vmax; out[1] = (double)vmin; out[2] = (double)med; out[3] = (double)madv;
    // scipy's center_of_mass: sum(index * w) / sum(w) in float64, then numpy's int32 cast (truncation)
    out[4] = (double)((int)((double)srow / (double)smax_n) + y0);
    out[5] = (double)((int)((double)scol / (double)smax_n) + x0);
    out[6] = (double)n;
  }
  __syncthreads();
  return true;
}

// rec[id][20]: bbox (minr, maxr, minc, maxc as slice start / stop), sal (max, min, med, mad, maxrow, maxcol, n),
//              cmf (max, min, med, mad, maxrow, maxcol, n), status (0 ok, 1 region larger than the LDS sort), -, -
__global__ __launch_bounds__(DT_NT) void k_region_stats(const int32_t *__restrict__ labels, int H, int W,
                                                        const float *__restrict__ sal, const double *__restrict__ cmf,
                                                        int cmf_nb, int cmf_band, const uint8_t *__restrict__ nodata,
                                                        double cmfthr, const int32_t *__restrict__ bbox, double *__restrict__ rec) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int id = blockIdx.x + 1, tid = threadIdx.x;
  const int32_t *bb = bbox + (size_t)id * 4;
  double *o = rec + (size_t)id * 20;
  if (tid < 4) o[tid] = (double)bb[tid];
  auto in_region = [&](int y, int x) { return labels[(size_t)y * W + x] == id && !nodata[(size_t)y * W + x]; };
  auto salv = [&](int y, int x) { return sal[(size_t)y * W + x]; };
  auto cmfv = [&](int y, int x) { return cmf[((size_t)y * W + x) * cmf_nb + cmf_band]; };
  auto in_cmf = [&](int y, int x) { return in_region(y, x) && cmfv(y, x) > cmfthr; };
  const bool ok1 = region_stats<float>(reinterpret_cast<float *>(lds), DT_CAP32, bb, in_region, salv, o + 4, tid);
  const bool ok2 = region_stats<double>(reinterpret_cast<double *>(lds), DT_CAP64, bb, in_cmf, cmfv, o + 11, tid);
  if (tid == 0) { o[18] = (ok1 && ok2) ? 0.0 : 1.0; o[19] = 0.0; }
  (void)H;
}

}  // namespace

extern "C" {

/* Region statistics of a labelled saliency map (salience_predictions.py:66-108).  labels[H][W]: 1..nregions from
 * sf_image_label8 of (salience > threshold); sal[H][W] float32; cmf[H][W][cmf_nb] float64 product, band cmf_band;
 * nodata[H][W] uint8 (RGB band 0 == -9999, :45).  bbox: scratch (nregions + 1) * 4 int32;
 * rec[nregions + 1][20] float64 (row 0 unused): see k_region_stats. */
int sf_detect_region_stats(const int32_t *labels, int H, int W, int nregions, const float *sal, const double *cmf,
                           int cmf_nb, int cmf_band, const uint8_t *nodata, double cmfthr, int32_t *bbox, double *rec,
                           void *stream) {
  if (!labels || !sal || !cmf || !nodata || !bbox || !rec || H < 1 || W < 1 || nregions < 0 || cmf_band < 0 ||
      cmf_band >= cmf_nb) {
    sf_set_error("sf_detect_region_stats: bad argument");
    return -1;
  }
  if (nregions == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_bbox_init, dim3(sf_cdiv(nregions + 1, 256)), dim3(256), 0, st, bbox, nregions);
  SF_LAUNCH_CHECK("k_bbox_init");
  hipLaunchKernelGGL(k_region_bbox, dim3(sf_cdiv(W, 64), sf_cdiv(H, 4)), dim3(256), 0, st, labels, H, W, bbox);
  SF_LAUNCH_CHECK("k_region_bbox");
  const size_t lds = (size_t)DT_CAP32 * sizeof(float);
  if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_region_stats), lds)) return rc;
  hipLaunchKernelGGL(k_region_stats, dim3(nregions), dim3(DT_NT), lds, st, labels, H, W, sal, cmf, cmf_nb, cmf_band, nodata,
                     cmfthr, bbox, rec);
  SF_LAUNCH_CHECK("k_region_stats");
  return 0;
}

}  // extern "C"
