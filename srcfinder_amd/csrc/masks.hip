// Spectrometer masks on the resident BIL cube and the image primitives they share with the
// saliency -> detections step (N4): binary dilations and 8-connected component labelling.
//
// Replaces spectrometer_masks/masks_sds.py:
//   per-pixel rules   get_saturation_mask :133-151, get_spec_mask :153-163, get_dark_mask :165-179,
//                     get_cloud_mask :181-232 (bright at band a AND falling slope a -> b; the slope b -> c is computed
//                     but never used: it is passed as numpy.logical_and's `out` argument, :230), border rule :330
//   morphology        dilate_mask :255-273 (N passes of skimage.morphology.binary_dilation's default cross),
//                     flare buffer :306-327 (binary_dilation with morphology.disk(radius) of the "grow" pixels:
//                     saturated pixels of regions of at least mingrowarea pixels whose 500 nm radiance is below the
//                     visible threshold; regions are 2-connected = 8-neighbour, measure.label :309)
// The cube is the BIL float32 [lines][bands][samples] array the matched filter reads (the reference opens the same file
// with interleave='bip'; the rules are per pixel, so the interleave only decides which loads coalesce: here lane = sample).
// Integer / boolean work throughout: results are bit-exact.
#include "cmf_common.h"

namespace {

// ---- per-pixel rules -------------------------------------------------------------------------------------------
// sat    any value of the saturation window > sat_thr                                  (:150)
// cloud  x[cb0] > cloud_thr and (x[cb1] - x[cb0]) / (wl[cb1] - wl[cb0]) < 0             (:196, :219-222, :230)
// spec   sat and x[vis_band] > vis_thr                                                  (:160-162)
// dark   x[dark_band] < dark_thr and not x[dark_band] <= -9999                          (:175-178)
// grow   sat and x[grow_band] < vis_thr   (the pixels a flare buffer grows from, :314)
// border x[0] == -9999                                                                  (:330)
__global__ __launch_bounds__(256) void k_masks_pixel(const float *__restrict__ cube, int L, int B, int S, int sat_b0,
                                                      int sat_b1, float sat_thr, int cb0, int cb1, float cloud_thr,
                                                      float dwl_sign, int vis_band, float vis_thr, int dark_band,
                                                      float dark_thr, int grow_band, uint8_t *__restrict__ sat,
                                                      uint8_t *__restrict__ cloud, uint8_t *__restrict__ spec,
                                                      uint8_t *__restrict__ dark, uint8_t *__restrict__ grow,
                                                      uint8_t *__restrict__ border) {
  const int s = blockIdx.x * 64 + (threadIdx.x & 63);
  const int l = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (s >= S || l >= L) return;
  const float *px = cube + (size_t)l * B * S + s;
  bool is_sat = false;
#pragma unroll 8
  for (int b = sat_b0; b < sat_b1; ++b) is_sat |= px[(size_t)b * S] > sat_thr;
  const float r0 = px[(size_t)cb0 * S], r1 = px[(size_t)cb1 * S];
  // der_a = (r1 - r0) / (-(wl[cb0] - wl[cb1])): only its sign is used; dwl_sign = sign(wl[cb1] - wl[cb0])
  const float diff = r1 - r0;                                   // float32, as numpy.diff of the float32 pair
  const bool slope_a = (dwl_sign > 0.f) ? (diff < 0.f) : (diff > 0.f);
  const float v = px[(size_t)vis_band * S], dk = px[(size_t)dark_band * S], g = px[(size_t)grow_band * S];
  const size_t i = (size_t)l * S + s;
  sat[i] = is_sat;
  cloud[i] = (r0 > cloud_thr) && slope_a;
  spec[i] = is_sat && (v > vis_thr);
  dark[i] = (dk < dark_thr) && !(dk <= -9999.0f);
  grow[i] = is_sat && (g < vis_thr);
  border[i] = px[0] == -9999.0f;
}

// one pass of the 4-neighbour (cross) binary dilation; outside the image is background
__global__ __launch_bounds__(256) void k_dilate_cross(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int H, int W) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  const size_t i = (size_t)y * W + x;
  uint8_t v = src[i];
  if (x > 0) v |= src[i - 1];
  if (x + 1 < W) v |= src[i + 1];
  if (y > 0) v |= src[i - W];
  if (y + 1 < H) v |= src[i + W];
  dst[i] = v != 0;
}

// distance (capped at cap) to the nearest foreground pixel of the same row
__global__ __launch_bounds__(256) void k_row_distance(const uint8_t *__restrict__ src, uint16_t *__restrict__ d, int H, int W, int cap) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  const uint8_t *row = src + (size_t)y * W;
  int best = cap;
  for (int k = 0; k < cap; ++k) {
    if ((x - k >= 0 && row[x - k]) || (x + k < W && row[x + k])) { best = k; break; }
  }
  d[(size_t)y * W + x] = (uint16_t)best;
}
// disk dilation from the row distances: a pixel is covered iff some row dy away has a foreground pixel within
// floor(sqrt(r^2 - dy^2)) columns (skimage.morphology.disk: x^2 + y^2 <= r^2)
__global__ __launch_bounds__(256) void k_dilate_disk(const uint16_t *__restrict__ d, uint8_t *__restrict__ dst, int H, int W, int r,
                                                      const int *__restrict__ halfw) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  bool hit = false;
  for (int dy = -r; dy <= r && !hit; ++dy) {
    const int yy = y + dy;
    if (yy < 0 || yy >= H) continue;
    hit = (int)d[(size_t)yy * W + x] <= halfw[dy + r];
  }
  dst[(size_t)y * W + x] = hit;
}
__global__ void k_disk_halfwidths(int r, int *halfw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > 2 * r) return;
  const long dy = i - r, rem = (long)r * r - dy * dy;
  long w = (long)floor(sqrt((double)rem));
  while ((w + 1) * (w + 1) <= rem) ++w;                         // exact integer square root
  while (w * w > rem) --w;
  halfw[i] = (int)w;
}

// ---- 8-connected component labelling (union-find on pixel indices, roots = smallest index of a component) ----------
__device__ __forceinline__ int cc_find(const int *lab, int i) {
  int r = lab[i];
  while (r != i) { i = r; r = lab[i]; }
  return r;
}
__device__ __forceinline__ void cc_union(int *lab, int a, int b) {
  for (;;) {
    a = cc_find(lab, a);
    b = cc_find(lab, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }               // a > b: hang the larger root under the smaller
    const int old = atomicMin(&lab[a], b);
    if (old == a) return;
    a = old;                                                     // somebody re-rooted a meanwhile: merge that root with b
  }
}
__global__ __launch_bounds__(256) void k_cc_init(const uint8_t *__restrict__ mask, int *__restrict__ lab, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) lab[i] = mask[i] ? i : -1;
}
__global__ __launch_bounds__(256) void k_cc_merge(const uint8_t *__restrict__ mask, int *__restrict__ lab, int H, int W) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  const int i = y * W + x;
  if (!mask[i]) return;
  if (x > 0 && mask[i - 1]) cc_union(lab, i, i - 1);
  if (y > 0) {
    if (mask[i - W]) cc_union(lab, i, i - W);
    if (x > 0 && mask[i - W - 1]) cc_union(lab, i, i - W - 1);
    if (x + 1 < W && mask[i - W + 1]) cc_union(lab, i, i - W + 1);
  }
}
// flatten, mark roots (1 at the first pixel of every component in raster order)
__global__ __launch_bounds__(256) void k_cc_flatten(int *__restrict__ lab, int *__restrict__ isroot, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int l = lab[i];
  int r = -1;
  if (l >= 0) r = cc_find(lab, i);
  isroot[i] = (r == i) ? 1 : 0;
  if (l >= 0) lab[i] = r;    // (a racing reader sees either the old parent or the root: both lead to the root)
}
// exclusive prefix sum of `isroot` in three passes (block sums, scan of the sums by one workgroup, apply)
constexpr int SCAN_B = 1024;
__global__ __launch_bounds__(256) void k_scan_blocksum(const int *__restrict__ v, int n, int *__restrict__ bsum) {
  __shared__ int red[256];
  const int base = blockIdx.x * SCAN_B;
  int s = 0;
  for (int k = threadIdx.x; k < SCAN_B; k += 256) s += (base + k < n) ? v[base + k] : 0;
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) bsum[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(1024) void k_scan_sums(int *__restrict__ bsum, int nb, int *__restrict__ total) {
  __shared__ int buf[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nb ? bsum[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int t = threadIdx.x >= o ? buf[threadIdx.x - o] : 0;
      __syncthreads();
      buf[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nb) bsum[i] = carry + buf[threadIdx.x] - v;          // exclusive
    __syncthreads();
    if (threadIdx.x == 0) carry += buf[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}
// labels[i] = 1-based component id in raster order of the components' first pixels (0 = background); area per id
__global__ __launch_bounds__(256) void k_cc_relabel(const int *__restrict__ lab, const int *__restrict__ isroot,
                                                     const int *__restrict__ bsum, int n, int *__restrict__ rootid) {
  __shared__ int wsum[4];
  const int base = blockIdx.x * SCAN_B;
  int run = bsum[blockIdx.x];
  for (int k0 = 0; k0 < SCAN_B; k0 += 256) {
    const int i = base + k0 + threadIdx.x;
    const int v = i < n ? isroot[i] : 0;
    // inclusive scan of v inside the 256-thread chunk: wave scan + wave offsets
    int x = v;
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(x, o, 64);
      if ((threadIdx.x & 63) >= o) x += t;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += wsum[w];
    const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (i < n && v) rootid[i] = run + off + x;                   // 1-based id of the component rooted at pixel i
    run += tot;
    __syncthreads();
  }
  (void)lab;
}
__global__ __launch_bounds__(256) void k_cc_assign(const int *__restrict__ lab, const int *__restrict__ rootid, int n,
                                                    int *__restrict__ labels, int *__restrict__ area) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int r = lab[i];
  const int id = r >= 0 ? rootid[r] : 0;
  labels[i] = id;
  if (id > 0 && area) atomicAdd(&area[id], 1);
}
// keep the pixels of `sel` that lie in components of at least minarea pixels
__global__ __launch_bounds__(256) void k_cc_filter(const int *__restrict__ labels, const int *__restrict__ area, int minarea,
                                                    uint8_t *__restrict__ sel, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int id = labels[i];
  if (id == 0 || area[id] < minarea) sel[i] = 0;
}

// product assembly (:336-341): int16 [lines][samples][4] = cloud (dilated), specular, flare (2 = buffer, 1 = flare
// and not specular), dark; -9999 on the image border (band 0 == -9999)
__global__ __launch_bounds__(256) void k_masks_compose(const uint8_t *__restrict__ cloud, const uint8_t *__restrict__ spec,
                                                        const uint8_t *__restrict__ sat, const uint8_t *__restrict__ buf,
                                                        const uint8_t *__restrict__ dark, const uint8_t *__restrict__ border,
                                                        int n, int16_t *__restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int16_t c = cloud[i], sp = spec[i], dk = dark[i];
  int16_t fl = 0;
  if (buf) {                                                    // only when a grow radius is set (:306)
    if (buf[i]) fl = 2;
    if (sat[i] && !spec[i]) fl = 1;
  }
  if (border[i]) c = sp = fl = dk = -9999;
  int16_t *o = out + (size_t)i * 4;
  o[0] = c; o[1] = sp; o[2] = fl; o[3] = dk;
}

dim3 grid2(int W, int H) { return dim3(sf_cdiv(W, 64), sf_cdiv(H, 4)); }

}  // namespace

extern "C" {

int sf_masks_pixel(const float *cube, int lines, int bands, int samples, int sat_b0, int sat_b1, float sat_thr, int cloud_b0,
                   int cloud_b1, float cloud_thr, float dwl, int vis_band, float vis_thr, int dark_band, float dark_thr,
                   int grow_band, uint8_t *sat, uint8_t *cloud, uint8_t *spec, uint8_t *dark, uint8_t *grow,
                   uint8_t *border, void *stream) {
  if (!cube || !sat || !cloud || !spec || !dark || !grow || !border) { sf_set_error("sf_masks_pixel: null pointer"); return -1; }
  const int bmax = bands - 1;
  if (lines < 1 || samples < 1 || sat_b0 < 0 || sat_b1 > bands || sat_b0 > sat_b1 || cloud_b0 < 0 || cloud_b0 > bmax ||
      cloud_b1 < 0 || cloud_b1 > bmax || vis_band < 0 || vis_band > bmax || dark_band < 0 || dark_band > bmax ||
      grow_band < 0 || grow_band > bmax || dwl == 0.f) {
    sf_set_error("sf_masks_pixel: band index out of range (cube has %d bands) or equal cloud wavelengths", bands);
    return -1;
  }
  hipLaunchKernelGGL(k_masks_pixel, grid2(samples, lines), dim3(256), 0, (hipStream_t)stream, cube, lines, bands, samples,
                     sat_b0, sat_b1, sat_thr, cloud_b0, cloud_b1, cloud_thr, dwl, vis_band, vis_thr, dark_band, dark_thr,
                     grow_band, sat, cloud, spec, dark, grow, border);
  SF_LAUNCH_CHECK("k_masks_pixel");
  return 0;
}

int sf_image_dilate_cross(uint8_t *mask, uint8_t *tmp, int H, int W, int iterations, void *stream) {
  if (!mask || !tmp || H < 1 || W < 1 || iterations < 0) { sf_set_error("sf_image_dilate_cross: bad argument"); return -1; }
  uint8_t *a = mask, *b = tmp;
  for (int it = 0; it < iterations; ++it) {
    hipLaunchKernelGGL(k_dilate_cross, grid2(W, H), dim3(256), 0, (hipStream_t)stream, a, b, H, W);
    SF_LAUNCH_CHECK("k_dilate_cross");
    uint8_t *t = a; a = b; b = t;
  }
  if (a != mask) SF_HIP(hipMemcpyAsync(mask, a, (size_t)H * W, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

size_t sf_image_dilate_disk_scratch_bytes(int H, int W, int radius) {
  return sf_align((size_t)H * W * sizeof(uint16_t)) + sf_align((size_t)(2 * radius + 1) * sizeof(int));
}
int sf_image_dilate_disk(const uint8_t *src, uint8_t *dst, int H, int W, int radius, void *scratch, void *stream) {
  if (!src || !dst || !scratch || H < 1 || W < 1 || radius < 0 || radius > 60000) {
    sf_set_error("sf_image_dilate_disk: bad argument");
    return -1;
  }
  uint16_t *d = reinterpret_cast<uint16_t *>(scratch);
  int *halfw = reinterpret_cast<int *>(reinterpret_cast<char *>(scratch) + sf_align((size_t)H * W * sizeof(uint16_t)));
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_disk_halfwidths, dim3(sf_cdiv(2 * radius + 1, 256)), dim3(256), 0, st, radius, halfw);
  SF_LAUNCH_CHECK("k_disk_halfwidths");
  hipLaunchKernelGGL(k_row_distance, grid2(W, H), dim3(256), 0, st, src, d, H, W, radius + 1);
  SF_LAUNCH_CHECK("k_row_distance");
  hipLaunchKernelGGL(k_dilate_disk, grid2(W, H), dim3(256), 0, st, d, dst, H, W, radius, halfw);
  SF_LAUNCH_CHECK("k_dilate_disk");
  return 0;
}

size_t sf_image_label8_scratch_bytes(int H, int W) {
  const size_t n = (size_t)H * W;
  return 3 * sf_align(n * sizeof(int)) + sf_align(((n + SCAN_B - 1) / SCAN_B + 1) * sizeof(int));
}
/* labels[H][W] int32: 0 background, 1..n components of mask != 0 (8-neighbour connectivity) numbered in raster order of
 * their first pixels (skimage.measure.label / scipy.ndimage.label order); area[id] (optional, >= n+1 ints, zeroed here
 * up to area_cap entries); *ncomp_dev receives n. */
int sf_image_label8(const uint8_t *mask, int H, int W, int32_t *labels, int32_t *area, int area_cap, int32_t *ncomp_dev,
                    void *scratch, void *stream) {
  if (!mask || !labels || !ncomp_dev || !scratch || H < 1 || W < 1 || (size_t)H * W > 0x7fffffffu) {
    sf_set_error("sf_image_label8: bad argument");
    return -1;
  }
  const int n = H * W, nb = sf_cdiv(n, SCAN_B);
  char *p = reinterpret_cast<char *>(scratch);
  int *lab = reinterpret_cast<int *>(p); p += sf_align((size_t)n * sizeof(int));
  int *isroot = reinterpret_cast<int *>(p); p += sf_align((size_t)n * sizeof(int));
  int *rootid = reinterpret_cast<int *>(p); p += sf_align((size_t)n * sizeof(int));
  int *bsum = reinterpret_cast<int *>(p);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_cc_init, dim3(sf_cdiv(n, 256)), dim3(256), 0, st, mask, lab, n);
  SF_LAUNCH_CHECK("k_cc_init");
  hipLaunchKernelGGL(k_cc_merge, grid2(W, H), dim3(256), 0, st, mask, lab, H, W);
  SF_LAUNCH_CHECK("k_cc_merge");
  hipLaunchKernelGGL(k_cc_flatten, dim3(sf_cdiv(n, 256)), dim3(256), 0, st, lab, isroot, n);
  SF_LAUNCH_CHECK("k_cc_flatten");
  hipLaunchKernelGGL(k_scan_blocksum, dim3(nb), dim3(256), 0, st, isroot, n, bsum);
  SF_LAUNCH_CHECK("k_scan_blocksum");
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st, bsum, nb, ncomp_dev);
  SF_LAUNCH_CHECK("k_scan_sums");
  hipLaunchKernelGGL(k_cc_relabel, dim3(nb), dim3(256), 0, st, lab, isroot, bsum, n, rootid);
  SF_LAUNCH_CHECK("k_cc_relabel");
  if (area) SF_HIP(hipMemsetAsync(area, 0, (size_t)area_cap * sizeof(int32_t), st));
  hipLaunchKernelGGL(k_cc_assign, dim3(sf_cdiv(n, 256)), dim3(256), 0, st, lab, rootid, n, labels, area);
  SF_LAUNCH_CHECK("k_cc_assign");
  return 0;
}

int sf_image_filter_small_components(const int32_t *labels, const int32_t *area, int minarea, uint8_t *sel, int H, int W,
                                     void *stream) {
  if (!labels || !area || !sel || H < 1 || W < 1) { sf_set_error("sf_image_filter_small_components: bad argument"); return -1; }
  const int n = H * W;
  hipLaunchKernelGGL(k_cc_filter, dim3(sf_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, labels, area, minarea, sel, n);
  SF_LAUNCH_CHECK("k_cc_filter");
  return 0;
}

int sf_masks_compose(const uint8_t *cloud, const uint8_t *spec, const uint8_t *sat, const uint8_t *flare_buffer,
                     const uint8_t *dark, const uint8_t *border, int lines, int samples, int16_t *out, void *stream) {
  if (!cloud || !spec || !sat || !dark || !border || !out || lines < 1 || samples < 1) {
    sf_set_error("sf_masks_compose: bad argument");
    return -1;
  }
  const int n = lines * samples;
  hipLaunchKernelGGL(k_masks_compose, dim3(sf_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, cloud, spec, sat, flare_buffer,
                     dark, border, n, out);
  SF_LAUNCH_CHECK("k_masks_compose");
  return 0;
}

}  // extern "C"
