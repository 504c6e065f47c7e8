// FCN shift-and-stitch: the reference's approximate fast mode for the saliency map (cnn/fcn_pred_pipeline.py) --
// gfx950 only.  The GoogLeNet trunk (conv1 .. inception5b, total stride 32) runs fully convolutionally over the WHOLE
// flightline once per (top, left) shift in [0, 32)^2, the 1x1 head gives one probability per 32x32 cell, and the 1024
// maps are interlaced (:67-92).  ~44x fewer flops than one 256x256 window per pixel; not the parity path of the tile
// scorer (zero padding happens at the flightline's border instead of every window's).
//
//   k_fcn_prepare    clamp + normalise (ClampCH4, Normalize) and embed at (top, left) in a zero canvas
//                    [(H + pad0 + 32), (W + pad1 + 32)], pad = scale - (n % scale)                    (:44-65)
//   k_conv1_img      conv1 7x7 s2 p3 (1 -> 64, folded BN + ReLU) over an arbitrary image, NHWC out    (googlenet1.py:60)
//   (trunk)          the same k_maxpool / k_conv_igemm kernels as the tile scorer
//   k_fcn_stitch     stitched[scale-top-1::scale, scale-left-1::scale] = pred, crop by scale/2, NODATA (:67-92, :249)
#include "cmf_common.h"
#include <hip/hip_fp16.h>

namespace {

__global__ __launch_bounds__(256) void k_fcn_prepare(const float *__restrict__ plane, int H, int W, float vmin, float vmax,
                                                      float mean, float stdv, int scale, int shift0, int Hc, int Wc,
                                                      float *__restrict__ out) {
  const int s = blockIdx.z;
  const int top = (shift0 + s) / scale, left = (shift0 + s) % scale;
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= Wc) return;
  const int iy = y - top, ix = x - left;
  float v = 0.f;
  if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
    const float t = plane[(size_t)iy * W + ix];
    v = (fminf(fmaxf(t, vmin), vmax) - mean) / stdv;      // torch.clamp, then Normalize: sub, div
  }
  out[((size_t)s * Hc + y) * Wc + x] = v;
}

constexpr int CI_PATCH = 37;  // 2*15 + 7
template <typename T>
__device__ __forceinline__ void store4(T *o, float a, float b, float c, float d);
template <>
__device__ __forceinline__ void store4<float>(float *o, float a, float b, float c, float d) {
  *reinterpret_cast<float4 *>(o) = make_float4(a, b, c, d);
}
template <>
__device__ __forceinline__ void store4<_Float16>(_Float16 *o, float a, float b, float c, float d) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  h4 v = {(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)d};
  *reinterpret_cast<h4 *>(o) = v;
}

// One workgroup = 16x16 outputs of one image; taps outside the image are zero (padding 3).  Same accumulation
// order as the tile kernel (ky, kx ascending, fp32 FMA).
template <typename T>
__global__ __launch_bounds__(256) void k_conv1_img(const float *__restrict__ img, int Hc, int Wc, int Ho, int Wo,
                                                    const float *__restrict__ w /*[64][49]*/, const float *__restrict__ bias,
                                                    T *__restrict__ out /*[n][Ho][Wo][64]*/) {
  __shared__ float patch[CI_PATCH][CI_PATCH + 1];
  __shared__ __attribute__((aligned(16))) float ws[49][64];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int n = blockIdx.z;
  const int oy0 = blockIdx.y * 16, ox0 = blockIdx.x * 16;
  const float *im = img + (size_t)n * Hc * Wc;
  for (int i = tid; i < 49 * 64; i += 256) ws[i / 64][i % 64] = w[(i % 64) * 49 + i / 64];
  for (int i = tid; i < CI_PATCH * CI_PATCH; i += 256) {
    const int py = i / CI_PATCH, px = i % CI_PATCH;
    const int iy = 2 * oy0 - 3 + py, ix = 2 * ox0 - 3 + px;
    float v = 0.f;
    if (iy >= 0 && iy < Hc && ix >= 0 && ix < Wc) v = im[(size_t)iy * Wc + ix];
    patch[py][px] = v;
  }
  __syncthreads();
  float acc[64];
#pragma unroll
  for (int c = 0; c < 64; ++c) acc[c] = 0.f;
  for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
      const float v = patch[2 * ty + ky][2 * tx + kx];
      const float4 *wr = reinterpret_cast<const float4 *>(&ws[ky * 7 + kx][0]);
#pragma unroll
      for (int c4 = 0; c4 < 16; ++c4) {
        const float4 ww = wr[c4];
        acc[4 * c4 + 0] = fmaf(v, ww.x, acc[4 * c4 + 0]);
        acc[4 * c4 + 1] = fmaf(v, ww.y, acc[4 * c4 + 1]);
        acc[4 * c4 + 2] = fmaf(v, ww.z, acc[4 * c4 + 2]);
        acc[4 * c4 + 3] = fmaf(v, ww.w, acc[4 * c4 + 3]);
      }
    }
  }
  const int oy = oy0 + ty, ox = ox0 + tx;
  if (oy >= Ho || ox >= Wo) return;
  T *o = out + (((size_t)n * Ho + oy) * Wo + ox) * 64;
  const float4 *b4 = reinterpret_cast<const float4 *>(bias);
#pragma unroll
  for (int c4 = 0; c4 < 16; ++c4) {
    const float4 bb = b4[c4];
    store4<T>(o + 4 * c4, fmaxf(acc[4 * c4] + bb.x, 0.f), fmaxf(acc[4 * c4 + 1] + bb.y, 0.f),
              fmaxf(acc[4 * c4 + 2] + bb.z, 0.f), fmaxf(acc[4 * c4 + 3] + bb.w, 0.f));
  }
}

// pred[ns][Hq][Wq] of shifts shift0.. -> out[H][W]: cell (i, j) of shift (top, left) is stitched pixel
// (scale-top-1 + scale i, scale-left-1 + scale j); the crop takes stitched[scale/2 + y][scale/2 + x]
__global__ __launch_bounds__(256) void k_fcn_stitch(const float *__restrict__ pred, int ns, int shift0, int scale, int Hq,
                                                     int Wq, const float *__restrict__ plane, int H, int W, float nodata,
                                                     float *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)ns * Hq * Wq;
  if (i >= total) return;
  const int s = (int)(i / ((size_t)Hq * Wq));
  const int r = (int)(i - (size_t)s * Hq * Wq);
  const int ci = r / Wq, cj = r - ci * Wq;
  const int top = (shift0 + s) / scale, left = (shift0 + s) % scale;
  const int y = scale - top - 1 + scale * ci - scale / 2, x = scale - left - 1 + scale * cj - scale / 2;
  if (y < 0 || y >= H || x < 0 || x >= W) return;
  const size_t o = (size_t)y * W + x;
  out[o] = (plane[o] == nodata) ? nodata : pred[i];
}

}  // namespace

extern "C" {

int sf_cnn_fcn_prepare(const float *plane, int H, int W, float vmin, float vmax, float mean, float stdv, int scale,
                       int shift0, int nshift, int Hc, int Wc, float *out, void *stream) {
  if (!plane || !out || H < 1 || W < 1 || scale < 1 || nshift < 1 || shift0 < 0 || shift0 + nshift > scale * scale ||
      Hc != H + (scale - H % scale) + scale || Wc != W + (scale - W % scale) + scale) {
    sf_set_error("sf_cnn_fcn_prepare: bad argument");
    return -1;
  }
  hipLaunchKernelGGL(k_fcn_prepare, dim3(sf_cdiv(Wc, 256), Hc, nshift), dim3(256), 0, (hipStream_t)stream, plane, H, W, vmin,
                     vmax, mean, stdv, scale, shift0, Hc, Wc, out);
  SF_LAUNCH_CHECK("k_fcn_prepare");
  return 0;
}

int sf_cnn_conv1_image(const float *img, int N, int Hc, int Wc, const float *w, const float *bias, void *out, int out_f16,
                       void *stream) {
  if (!img || !w || !bias || !out || N < 1 || Hc < 1 || Wc < 1) { sf_set_error("sf_cnn_conv1_image: bad argument"); return -1; }
  const int Ho = (Hc - 1) / 2 + 1, Wo = (Wc - 1) / 2 + 1;
  const dim3 grid(sf_cdiv(Wo, 16), sf_cdiv(Ho, 16), N);
  if (out_f16)
    hipLaunchKernelGGL(k_conv1_img<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, img, Hc, Wc, Ho, Wo, w, bias,
                       reinterpret_cast<_Float16 *>(out));
  else
    hipLaunchKernelGGL(k_conv1_img<float>, grid, dim3(256), 0, (hipStream_t)stream, img, Hc, Wc, Ho, Wo, w, bias,
                       reinterpret_cast<float *>(out));
  SF_LAUNCH_CHECK("k_conv1_img");
  return 0;
}

int sf_cnn_fcn_stitch(const float *pred, int nshift, int shift0, int scale, int Hq, int Wq, const float *plane, int H, int W,
                      float nodata, float *out, void *stream) {
  if (!pred || !plane || !out || nshift < 1 || scale < 1 || Hq < 1 || Wq < 1) { sf_set_error("sf_cnn_fcn_stitch: bad argument"); return -1; }
  const size_t total = (size_t)nshift * Hq * Wq;
  hipLaunchKernelGGL(k_fcn_stitch, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred, nshift,
                     shift0, scale, Hq, Wq, plane, H, W, nodata, out);
  SF_LAUNCH_CHECK("k_fcn_stitch");
  return 0;
}

}  // extern "C"
