// CNN tile scorer kernels (gfx950): the eval graph of cnn/archs/googlenet1.py evaluated on a 256x256 window
// around every pixel (cnn/cnn_pred_pipeline.py).  Activations are NHWC float32 so that the contraction axis
// (tap, input channel) of every convolution is contiguous; BatchNorm (eps 1e-3, running stats) is folded into
// the weights/bias on upload, ReLU is fused into the conv epilogue, inception branches write straight into their
// channel slice of the concatenated output.  Convolutions run as implicit GEMM on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains, no reduced precision) -- MFMA-bound: 3.7 GFLOP per tile.
#include "cmf_common.h"

typedef float f16_t __attribute__((ext_vector_type(16)));
typedef unsigned u4_t __attribute__((ext_vector_type(4)));

namespace {

// ---- clamp -> normalize -> zero pad (cnn_pred_pipeline.py:19-30, :39-47, :126-157) -------------------------------
__global__ void k_prepare(const float *__restrict__ plane, int H, int W, float vmin, float vmax, float mean, float stdv,
                          int padlo, int Hp, int Wp, float *__restrict__ padded) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Hp * Wp) return;
  const int y = i / Wp - padlo, x = i % Wp - padlo;
  float v = 0.f;
  if (y >= 0 && y < H && x >= 0 && x < W) {
    float t = plane[(size_t)y * W + x];
    t = fminf(fmaxf(t, vmin), vmax);  // torch.clamp (NaN stays NaN: fmaxf(NaN, a) = a would not -> handle below)
    if (plane[(size_t)y * W + x] != plane[(size_t)y * W + x]) t = plane[(size_t)y * W + x];
    v = (t - mean) / stdv;
  }
  padded[i] = v;
}

// ---- conv1: 7x7 stride 2 pad 3, 1 -> 64 channels, read straight from the padded plane (no tile copy) ---------------
// (googlenet1.py:60; tile window cnn_pred_pipeline.py:53-58).  The tile is its own image: taps outside
// [0,256) are zero even where the padded flightline has data.  One workgroup = 16x16 outputs of one tile.
constexpr int C1_PATCH = 37;  // 2*15 + 7
__global__ __launch_bounds__(256) void k_conv1(const float *__restrict__ padded, int Wp, int Wimg, long long tile0,
                                                const float *__restrict__ w /*[64][49]*/, const float *__restrict__ bias,
                                                float *__restrict__ out /*[n][128][128][64]*/) {
  __shared__ float patch[C1_PATCH][C1_PATCH + 1];
  __shared__ __attribute__((aligned(16))) float ws[49][64];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int t = blockIdx.y;
  const long long tile = tile0 + t;
  const int trow = (int)(tile / Wimg), tcol = (int)(tile % Wimg);
  const int oy0 = (blockIdx.x >> 3) * 16, ox0 = (blockIdx.x & 7) * 16;
  for (int i = tid; i < 49 * 64; i += 256) ws[i / 64][i % 64] = w[(i % 64) * 49 + i / 64];
  for (int i = tid; i < C1_PATCH * C1_PATCH; i += 256) {
    const int py = i / C1_PATCH, px = i % C1_PATCH;
    const int iy = 2 * oy0 - 3 + py, ix = 2 * ox0 - 3 + px;  // tile-local input coordinates
    float v = 0.f;
    if (iy >= 0 && iy < 256 && ix >= 0 && ix < 256) v = padded[(size_t)(trow + iy) * Wp + tcol + ix];
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
  float4 *o = reinterpret_cast<float4 *>(out + (((size_t)t * 128 + oy0 + ty) * 128 + ox0 + tx) * 64);
  const float4 *b4 = reinterpret_cast<const float4 *>(bias);
#pragma unroll
  for (int c4 = 0; c4 < 16; ++c4) {
    const float4 bb = b4[c4];
    o[c4] = make_float4(fmaxf(acc[4 * c4] + bb.x, 0.f), fmaxf(acc[4 * c4 + 1] + bb.y, 0.f),
                        fmaxf(acc[4 * c4 + 2] + bb.z, 0.f), fmaxf(acc[4 * c4 + 3] + bb.w, 0.f));
  }
}

// ---- conv1 + maxpool1 fused: 7x7 stride 2 pad 3 (1 -> 64) + BN + ReLU, then 3x3 stride 2 ceil-mode max pool ----------
// (googlenet1.py:60-61).  conv1's 128 x 128 x 64 activation of a tile is 4 MB that the pool reads back at once and
// nothing else ever does: written and read for every tile it is 8 of the ~25 MB a tile moves through HBM.  Here a
// workgroup produces an 8 x 8 block of POOLED pixels: it stages the 39 x 39 input patch, evaluates the 17 x 17 conv
// outputs the block's windows cover as one implicit GEMM on the matrix cores (M = 289 pixels in 10 blocks of 32,
// N = 64 channels, K = 49 taps padded to 50; A is gathered from the patch in LDS, lane by lane), keeps them in LDS and
// pools from there.  Only the 64 x 64 x 64 pooled activation (1 MB per tile) leaves the CU.
constexpr int CP_PT = 8, CP_CR = 2 * CP_PT + 1, CP_NPX = CP_CR * CP_CR;   // pooled tile, conv rows/cols, conv pixels
constexpr int CP_PATCH = 2 * (CP_CR - 1) + 7;                             // 39
constexpr int CP_LDP = CP_PATCH + 1, CP_LDB = 64 + 1, CP_LDC = 36, CP_KP = 50, CP_MB = (CP_NPX + 31) / 32;   // 10 blocks
// 61 KB: two workgroups per CU (the channels go in two halves of 32 through one [289][36] conv tile), so one's patch
// load / pooling runs under the other's matrix phase
static size_t cp_lds_bytes() { return ((size_t)CP_PATCH * CP_LDP + CP_KP * CP_LDB + (size_t)CP_NPX * CP_LDC + 64) * sizeof(float); }
// EXP: timing experiments of tools/microbench/conv1pool.hip (0 in production): 1 no weight gather, 2 no patch load,
// 4 no matrix loop, 8 no conv-tile store, 16 no pooling reads, 32 no global store
template <int EXP>
__global__ __launch_bounds__(256, 2) void k_conv1_pool(const float *__restrict__ padded, int Wp, int Wimg, long long tile0,
                                                        const float *__restrict__ w /*[64][49]*/, const float *__restrict__ bias,
                                                        float *__restrict__ out /*[n][64][64][64]*/) {
  extern __shared__ __attribute__((aligned(16))) float cp_lds[];
  float *patch = cp_lds;                                   // [39][40]
  float *Bs = patch + CP_PATCH * CP_LDP;                   // [50][65]  Bs[k][co], rows 49.. zero
  float *convt = Bs + CP_KP * CP_LDB;                      // [289][36] relu(conv + bias) of one channel half
  float *bs = convt + CP_NPX * CP_LDC;                     // [64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = blockIdx.y;
  const long long tile = tile0 + t;
  const int trow = (int)(tile / Wimg), tcol = (int)(tile % Wimg);
  const int py0 = (blockIdx.x >> 3) * CP_PT, px0 = (blockIdx.x & 7) * CP_PT;     // pooled block origin
  const int cy0 = 2 * py0, cx0 = 2 * px0;                                         // conv block origin
  for (int i = tid; i < CP_KP * 64; i += 256) {
    const int k = i / 64, co = i % 64;
    Bs[k * CP_LDB + co] = (k < 49 && !(EXP & 1)) ? w[co * 49 + k] : 0.f;
  }
  if (tid < 64) bs[tid] = bias[tid];
  for (int i = tid; i < CP_PATCH * CP_PATCH; i += 256) {
    const int py = i / CP_PATCH, px = i % CP_PATCH;
    const int iy = 2 * cy0 - 3 + py, ix = 2 * cx0 - 3 + px;                      // tile-local input coordinates
    float v = 0.f;                                                               // the tile is its own image: zero outside
    if (iy >= 0 && iy < 256 && ix >= 0 && ix < 256 && !(EXP & 2)) v = padded[(size_t)(trow + iy) * Wp + tcol + ix];
    patch[py * CP_LDP + px] = v;
  }
  __syncthreads();
  const int l31 = lane & 31, kh = lane >> 5;
  constexpr int NBW = (CP_MB + 3) / 4;
  int abase[NBW];
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    const int m = min(32 * (wave + 4 * j) + l31, CP_NPX - 1);                     // rows past 289: duplicates, never stored
    abase[j] = (2 * (m / CP_CR)) * CP_LDP + 2 * (m % CP_CR);
  }
  for (int h = 0; h < 2; ++h) {                                                   // channel half
    // ---- implicit GEMM: wave w takes the M blocks w, w + 4, w + 8 (the last two waves two blocks)
    f16_t acc[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll 5
    for (int kk = 0; kk < ((EXP & 4) ? 1 : CP_KP / 2); ++kk) {
      const int k = 2 * kk + kh, kc = min(k, 48);                                 // tap 49 is padding: weight row is zero
      const int koff = (kc / 7) * CP_LDP + (kc % 7);
      const float b = Bs[k * CP_LDB + 32 * h + l31];
#pragma unroll
      for (int j = 0; j < NBW; ++j)
        if (wave + 4 * j < CP_MB)                                                 // (wave-uniform)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(patch[abase[j] + koff], b, acc[j], 0, 0, 0);
    }
    // acc[.][r] = D[row = (r&3) + 8(r>>2) + 4(lane>>5)][col = lane&31]
    const float bb = bs[32 * h + l31];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      if (wave + 4 * j < CP_MB) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = 32 * (wave + 4 * j) + (r & 3) + 8 * (r >> 2) + 4 * kh;
          if (m < CP_NPX && (!(EXP & 8) || r == 0)) convt[m * CP_LDC + l31] = fmaxf(acc[j][r] + bb, 0.f);
        }
      }
    }
    __syncthreads();
    // ---- 3x3 stride-2 max pool, windows clipped to the 128 x 128 conv image (ceil_mode: the last window is partial)
    for (int i = tid; i < CP_PT * CP_PT * 8; i += 256) {
      const int c4 = i & 7, pp = i >> 3, py = pp / CP_PT, px = pp % CP_PT;
      float4 m = make_float4(-3.402823466e38f, -3.402823466e38f, -3.402823466e38f, -3.402823466e38f);
#pragma unroll
      for (int dy = 0; dy < ((EXP & 16) ? 1 : 3); ++dy) {
        if (cy0 + 2 * py + dy > 127) continue;
#pragma unroll
        for (int dx = 0; dx < ((EXP & 16) ? 1 : 3); ++dx) {
          if (cx0 + 2 * px + dx > 127) continue;
          const float4 v = *reinterpret_cast<const float4 *>(convt + ((2 * py + dy) * CP_CR + 2 * px + dx) * CP_LDC + 4 * c4);
          m = make_float4(fmaxf(m.x, v.x), fmaxf(m.y, v.y), fmaxf(m.z, v.z), fmaxf(m.w, v.w));
        }
      }
      if (!(EXP & 32) || m.x == 123.456f)
        *reinterpret_cast<float4 *>(out + (((size_t)t * 64 + py0 + py) * 64 + px0 + px) * 64 + 32 * h + 4 * c4) = m;
    }
    __syncthreads();
  }
}

// ---- conv1 + maxpool1, second form: 16 x 16 pooled pixels per workgroup, pooling by LDS atomics --------------------
// k_conv1_pool spends more time around its matrix phase than in it (tools/microbench/conv1pool.hip: of 1.80 ms per
// 512 tiles the matrix loop is 0.9, writing the conv tile to LDS 0.27, reading it back for the pool 0.43, the prologue
// 0.17), at two workgroups per CU because the [289][36] conv tile takes 42 of its 61 KB.  Here the rows of the implicit
// GEMM are ordered by 2 x 2 QUADS of conv pixels: a lane's four consecutive accumulator rows are one quad, and a 3 x 3
// stride-2 pooling window is exactly quad (qy, qx) + the left column of quad (qy, qx+1) + the top row of quad (qy+1, qx)
// + the top-left pixel of quad (qy+1, qx+1).  So every lane reduces its quad to four numbers in registers and merges
// them into the POOLED tile with four LDS integer-max atomics (the values are >= 0 after ReLU: their bit patterns order
// like integers, and a tile of zeros is the identity) -- no conv tile, no read-back.  A workgroup of 8 waves produces
// 16 x 16 pooled pixels from 17 x 17 quads (the 17th row / column only through its top row / left column): 37 M-blocks
// instead of 4 x 10, 13 % above the 128 x 128 conv pixels a tile has, where the 8 x 8 form computes 25 % more and waits
// for its slowest wave's third block.  LDS: patch 73 x 74, weights [50][65], pooled [256][32] = 67 KB, two workgroups
// (16 waves) per CU.
constexpr int C2_PT = 16, C2_Q = C2_PT + 1, C2_NQ = C2_Q * C2_Q, C2_MB = (C2_NQ + 7) / 8;   // 289 quads, 37 blocks
constexpr int C2_PATCH = 4 * C2_Q + 5, C2_LDP = C2_PATCH + 1;                               // 73 input rows / cols
constexpr int C2_NT = 512, C2_NW = C2_NT / 64, C2_PASS = 3;
static size_t c2_lds_bytes() { return ((size_t)C2_PATCH * C2_LDP + CP_KP * CP_LDB + (size_t)C2_PT * C2_PT * 32 + 64) * sizeof(float); }
template <int EXP>
__global__ __launch_bounds__(C2_NT, 2) void k_conv1_pool16(const float *__restrict__ padded, int Wp, int Wimg, long long tile0,
                                                            const float *__restrict__ w /*[64][49]*/,
                                                            const float *__restrict__ bias,
                                                            float *__restrict__ out /*[n][64][64][64]*/) {
  extern __shared__ __attribute__((aligned(16))) float cp_lds[];
  float *patch = cp_lds;                                   // [73][74]
  float *Bs = patch + C2_PATCH * C2_LDP;                   // [50][65]  Bs[k][co], row 49 zero
  int *pooled = reinterpret_cast<int *>(Bs + CP_KP * CP_LDB);   // [16][16][32] bit patterns of non-negative floats
  float *bs = reinterpret_cast<float *>(pooled + C2_PT * C2_PT * 32);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = blockIdx.y;
  const long long tile = tile0 + t;
  const int trow = (int)(tile / Wimg), tcol = (int)(tile % Wimg);
  const int py0 = (blockIdx.x >> 2) * C2_PT, px0 = (blockIdx.x & 3) * C2_PT;      // pooled block origin
  const int cy0 = 2 * py0, cx0 = 2 * px0;                                         // conv block origin
  for (int i = tid; i < CP_KP * 64; i += C2_NT) {
    const int k = i / 64, co = i % 64;
    Bs[k * CP_LDB + co] = (k < 49) ? w[co * 49 + k] : 0.f;
  }
  if (tid < 64) bs[tid] = bias[tid];
  for (int i = tid; i < C2_PATCH * C2_PATCH; i += C2_NT) {
    const int py = i / C2_PATCH, px = i % C2_PATCH;
    const int iy = 2 * cy0 - 3 + py, ix = 2 * cx0 - 3 + px;                      // tile-local input coordinates
    float v = 0.f;                                                               // the tile is its own image: zero outside
    if (iy >= 0 && iy < 256 && ix >= 0 && ix < 256) v = padded[(size_t)(trow + iy) * Wp + tcol + ix];
    patch[py * C2_LDP + px] = v;
  }
  const int l31 = lane & 31, kh = lane >> 5;
  for (int h = 0; h < 2; ++h) {                                                   // channel half
    for (int i = tid; i < C2_PT * C2_PT * 32; i += C2_NT) pooled[i] = 0;
    __syncthreads();                                                              // (also: patch and weights are in LDS)
    const float bb = bs[32 * h + l31];
    for (int b0 = wave; b0 < C2_MB; b0 += C2_NW * C2_PASS) {                      // blocks b0, b0 + 8, b0 + 16 of this pass
      int abase[C2_PASS];
#pragma unroll
      for (int j = 0; j < C2_PASS; ++j) {
        const int m = 32 * (b0 + C2_NW * j) + l31;
        const int q = min(m >> 2, C2_NQ - 1), wq = m & 3;
        abase[j] = (2 * (2 * (q / C2_Q) + (wq >> 1))) * C2_LDP + 2 * (2 * (q % C2_Q) + (wq & 1));
      }
      f16_t acc[C2_PASS];
#pragma unroll
      for (int j = 0; j < C2_PASS; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll 5
      for (int kk = 0; kk < ((EXP & 4) ? 1 : CP_KP / 2); ++kk) {
        const int k = 2 * kk + kh, kc = min(k, 48);                               // tap 49 is padding: weight row is zero
        const int koff = (kc / 7) * C2_LDP + (kc % 7);
        const float b = Bs[k * CP_LDB + 32 * h + l31];
#pragma unroll
        for (int j = 0; j < C2_PASS; ++j)
          if (b0 + C2_NW * j < C2_MB)                                             // (wave-uniform)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(patch[abase[j] + koff], b, acc[j], 0, 0, 0);
      }
      // acc[.][r]: row (r&3) + 8(r>>2) + 4 kh of the block = pixel r&3 of quad 2(r>>2) + kh, channel l31
#pragma unroll
      for (int j = 0; j < C2_PASS; ++j) {
        if (b0 + C2_NW * j >= C2_MB) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int q = 8 * (b0 + C2_NW * j) + 2 * g + kh;
          if (q >= C2_NQ || (EXP & 8)) continue;
          const int qy = q / C2_Q, qx = q % C2_Q;
          const bool y0 = cy0 + 2 * qy <= 127, y1 = cy0 + 2 * qy + 1 <= 127;       // conv pixels past the 128 x 128 image
          const bool x0 = cx0 + 2 * qx <= 127, x1 = cx0 + 2 * qx + 1 <= 127;       // do not exist (ceil_mode windows are clipped)
          const float v00 = (y0 && x0) ? fmaxf(acc[j][4 * g + 0] + bb, 0.f) : 0.f;
          const float v01 = (y0 && x1) ? fmaxf(acc[j][4 * g + 1] + bb, 0.f) : 0.f;
          const float v10 = (y1 && x0) ? fmaxf(acc[j][4 * g + 2] + bb, 0.f) : 0.f;
          const float v11 = (y1 && x1) ? fmaxf(acc[j][4 * g + 3] + bb, 0.f) : 0.f;
          const float top = fmaxf(v00, v01), left = fmaxf(v00, v10), all = fmaxf(fmaxf(top, v10), v11);
          int *pq = pooled + (qy * C2_PT + qx) * 32 + l31;
          if (qy < C2_PT && qx < C2_PT) atomicMax(pq, __float_as_int(all));
          if (qy < C2_PT && qx >= 1) atomicMax(pq - 32, __float_as_int(left));
          if (qy >= 1 && qx < C2_PT) atomicMax(pq - C2_PT * 32, __float_as_int(top));
          if (qy >= 1 && qx >= 1) atomicMax(pq - C2_PT * 32 - 32, __float_as_int(v00));
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < C2_PT * C2_PT * 8; i += C2_NT) {
      const int c4 = i & 7, pp = i >> 3, py = pp / C2_PT, px = pp % C2_PT;
      const float4 m = *reinterpret_cast<const float4 *>(pooled + pp * 32 + 4 * c4);
      if (!(EXP & 32) || m.x == 123.456f)
        *reinterpret_cast<float4 *>(out + (((size_t)t * 64 + py0 + py) * 64 + px0 + px) * 64 + 32 * h + 4 * c4) = m;
    }
    __syncthreads();
  }
}

// ---- max pool, NHWC, window clipped to the input (ceil_mode edge windows are partial) ------------------------------
// (googlenet1.py:61,:64,:68,:75 and the stride-1 pool of the inception branch4 :213)
// Each thread produces PR vertically adjacent outputs of one (n, ox, channel quad): the 3-wide row maxima of the
// input rows it touches are computed once and shared by the overlapping windows (stride-1 3x3 pool: 4.5 vector
// loads per output instead of 9).
constexpr int PR = 4;
__device__ __forceinline__ float4 max4(float4 a, float4 b) {
  return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}
__global__ void k_maxpool(const float *__restrict__ in, int N, int H, int W, int C, int ks, int stride, int pad,
                          float *__restrict__ out, int Ho, int Wo) {
  const int c4n = C >> 2;
  const int hob = (Ho + PR - 1) / PR;
  const size_t total = (size_t)N * hob * Wo * c4n;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c4 = (int)(i % c4n);
  size_t r = i / c4n;
  const int ox = (int)(r % Wo); r /= Wo;
  const int oyb = (int)(r % hob);
  const int n = (int)(r / hob);
  const int x0 = max(ox * stride - pad, 0), x1 = min(ox * stride - pad + ks, W);
  const float4 lowest = make_float4(-3.402823466e38f, -3.402823466e38f, -3.402823466e38f, -3.402823466e38f);
  const int oy0 = oyb * PR;
  const int ylo = max(oy0 * stride - pad, 0);
  const int yhi = min((min(oy0 + PR, Ho) - 1) * stride - pad + ks, H);
  float4 acc[PR];
#pragma unroll
  for (int k = 0; k < PR; ++k) acc[k] = lowest;
  for (int y = ylo; y < yhi; ++y) {
    float4 m = lowest;
    for (int x = x0; x < x1; ++x)
      m = max4(m, *reinterpret_cast<const float4 *>(in + (((size_t)n * H + y) * W + x) * C + 4 * c4));
#pragma unroll
    for (int k = 0; k < PR; ++k) {
      const int ys = (oy0 + k) * stride - pad;
      if (y >= ys && y < ys + ks) acc[k] = max4(acc[k], m);
    }
  }
#pragma unroll
  for (int k = 0; k < PR; ++k)
    if (oy0 + k < Ho) *reinterpret_cast<float4 *>(out + (((size_t)n * Ho + oy0 + k) * Wo + ox) * C + 4 * c4) = acc[k];
}

// The 3x3 stride-1 pad-1 pool of the inception branch 4 (nine launches per batch, 1.3 of its 18 ms): the same strip of four
// output rows per thread, with the 18 loads of the strip issued up front as raw buffer loads -- a tap outside the image
// takes an out-of-range offset and returns 0, the identity for the non-negative activations it is applied to -- instead
// of two runtime-bounded loops with one load in flight.
__global__ __launch_bounds__(256) void k_maxpool_s1(const float *__restrict__ in, int N, int H, int W, int C,
                                                     float *__restrict__ out) {
  const int c4n = C >> 2;
  const int hob = (H + PR - 1) / PR;
  const size_t total = (size_t)N * hob * W * c4n;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c4 = (int)(i % c4n);
  size_t r = i / c4n;
  const int ox = (int)(r % W); r /= W;
  const int oyb = (int)(r % hob);
  const int n = (int)(r / hob);
  const int oy0 = oyb * PR;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (unsigned)((size_t)N * H * W * C * 4), 0x00020000);
  const unsigned base = (unsigned)((((size_t)n * H + oy0) * W + ox) * C + 4 * c4) * 4u;    // (oy0, ox) of this image
  const unsigned rowb = (unsigned)(W * C * 4), colb = (unsigned)(C * 4);
  float4 m[PR + 2];
#pragma unroll
  for (int j = 0; j < PR + 2; ++j) {                       // input rows oy0 - 1 + j
    const int y = oy0 - 1 + j;
    const bool yok = y >= 0 && y < H;
    float4 mx = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const bool ok = yok && ox + dx >= 0 && ox + dx < W;
      const unsigned off = ok ? base + (unsigned)(j - 1) * rowb + (unsigned)dx * colb : 0x80000000u;
      const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
      mx = make_float4(fmaxf(mx.x, __uint_as_float(v.x)), fmaxf(mx.y, __uint_as_float(v.y)), fmaxf(mx.z, __uint_as_float(v.z)),
                       fmaxf(mx.w, __uint_as_float(v.w)));
    }
    m[j] = mx;
  }
#pragma unroll
  for (int k = 0; k < PR; ++k)
    if (oy0 + k < H)
      *reinterpret_cast<float4 *>(out + (((size_t)n * H + oy0 + k) * W + ox) * C + 4 * c4) = max4(max4(m[k], m[k + 1]), m[k + 2]);
}

// ---- implicit-GEMM convolution, 1x1 or 3x3 (pad k/2), stride 1, + folded-BN bias + ReLU ----------------------------
// (BasicConv2d, googlenet1.py:266-275).  D[m][co] = sum_{tap,ci} in[pixel(m)+tap][ci] * w[co][tap][ci];
// block tile 128 pixels x BN channels, k-chunk BK input channels of one tap; each wave owns 32 pixel rows and all
// BN columns (BN/32 MFMA tiles sharing the A fragment).  Register-staged double buffering: the next chunk's global
// loads are in flight while the current chunk is multiplied out of LDS.
// Output routing: channels [0, e0) -> dst[0], [e0, e1) -> dst[1], [e1, Cout) -> dst[2]; each destination has its own
// pixel stride and channel offset.  One launch can therefore evaluate the three 1x1 convolutions that read the
// same inception input (branch1, branch2.0, branch3.0) as ONE GEMM with a wider N.
struct ConvDst {
  float *p[3];
  int ld[3], off[3], end[3];
};

// EXP: timing experiments of tools/microbench/convigemm.hip (0 in production): 1 no global loads after the first chunk,
// 2 no LDS stores after the first chunk, 4 one MFMA step per chunk, 8 no epilogue stores
// BUF: the tile loads are raw buffer loads -- the lane's byte offset in a VGPR (an out-of-range value for a tap outside
// the image or a row / channel past the end: the load returns 0, no branch), the (tap, channel chunk) offset in an
// SGPR, the per-row tap validity in a 9-bit mask computed once.  The pointer form below spends ~260 scalar and vector
// instructions per chunk and wave on bounds tests and 64-bit addresses beside 32 MFMAs; the matrix pipe idles a
// quarter of the time waiting for waves that are all in that phase (PMC: tools/pmc_bin.sh on tools/microbench/convigemm).
// Needs every operand below 2 GB (host-checked; larger tensors take the pointer form).
// POOL (buffer form, 1x1 convolutions only): the A tile is the 3x3 stride-1 pad-1 max pool of the input, taken while it is
// fetched -- nine loads and eight maxima per element instead of a pooled copy in HBM (inception branch 4,
// googlenet1.py:213-214).  Activations are >= 0 after ReLU, so the zero an out-of-range load returns is the identity.
template <int BK, int BN, int EXP = 0, bool BUF = false, bool POOL = false>
__global__ __launch_bounds__(256) void k_conv_igemm(const float *__restrict__ in, int M, int H, int W, int Cin, int ld_in,
                                                     const float *__restrict__ wt, const float *__restrict__ bias, int Cout,
                                                     int ks, ConvDst dst) {
  constexpr int BM = 128, LDA = BM + 1, LDB = BN + 1;
  constexpr int TPP = BK / 4;        // threads per pixel row (one float4 each)
  constexpr int PPP = 256 / TPP;     // rows per pass
  constexpr int NPA = BM / PPP;      // passes for the A tile
  constexpr int NPB = (BN + PPP - 1) / PPP;
  constexpr int NT = BN / 32;
  __shared__ float As[BK * LDA];
  __shared__ float Bs[BK * LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = tid % TPP, ri = tid / TPP;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int pad = ks >> 1, taps = ks * ks, nchunk = Cin / BK, nit = taps * nchunk;

  int py[NPA], px[NPA];
  const float *pb[NPA];
#pragma unroll
  for (int a = 0; a < NPA; ++a) {
    const int m = m0 + ri + PPP * a;
    const bool ok = m < M;
    const int mm = ok ? m : 0;
    const int x = mm % W, y = (mm / W) % H;
    py[a] = ok ? y : -100000;  // invalid rows fail every bounds test
    px[a] = x;
    pb[a] = in + (size_t)mm * ld_in + 4 * q;
  }
  const float *wb[NPB];
  bool wok[NPB];
#pragma unroll
  for (int b = 0; b < NPB; ++b) {
    const int co = n0 + ri + PPP * b;
    wok[b] = (ri + PPP * b < BN) && co < Cout;
    wb[b] = wt + (size_t)(wok[b] ? co : 0) * taps * Cin + 4 * q;
  }

  f16_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  float4 ra[NPA], rb[NPB];
  // ---- buffer form: descriptors, per-row offsets and tap masks
  constexpr unsigned OOB = 0x80000000u;
  unsigned rowoff[NPA], vmask[NPA], woff[NPB];
  __amdgpu_buffer_rsrc_t rsA, rsB;
  const int fks = POOL ? 3 : ks, fpad = POOL ? 1 : pad;             // the footprint the tile fetch reads around a pixel
  if (BUF) {
    const size_t shift = ((size_t)fpad * W + fpad) * ld_in;        // taps are addressed from (y - pad, x - pad): offsets >= 0
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in) - shift, 0,
                                            (unsigned)(((size_t)M * ld_in + 2 * shift) * 4 + 64), 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wt), 0, (unsigned)((size_t)Cout * taps * Cin * 4), 0x00020000);
#pragma unroll
    for (int a = 0; a < NPA; ++a) {
      const int m = m0 + ri + PPP * a;
      const int mm = m < M ? m : 0;
      rowoff[a] = (unsigned)(((size_t)mm * ld_in + 4 * q) * 4);
      unsigned vm = 0;
      for (int tp = 0; tp < fks * fks; ++tp) {
        const int yy = py[a] + tp / fks - fpad, xx = px[a] + tp % fks - fpad;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) vm |= 1u << tp;
      }
      vmask[a] = vm;
    }
#pragma unroll
    for (int b = 0; b < NPB; ++b) {
      const int co = n0 + ri + PPP * b;
      woff[b] = wok[b] ? (unsigned)(((size_t)co * taps * Cin + 4 * q) * 4) : OOB;
    }
  }
  int g_tap = 0, g_ty = 0, g_tx = 0, g_c0 = 0;                      // the chunk the next gload() fetches (buffer form)
  auto gload = [&](int it) {
    if (BUF && POOL) {
      const unsigned sb = (unsigned)(g_c0 * 4);
#pragma unroll
      for (int a = 0; a < NPA; ++a) {
        float4 mx = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          const unsigned so = (unsigned)((((tp / 3) * W + tp % 3) * ld_in + g_c0) * 4);
          const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsA, ((vmask[a] >> tp) & 1u) ? rowoff[a] : OOB, so, 0);
          mx = make_float4(fmaxf(mx.x, __uint_as_float(v.x)), fmaxf(mx.y, __uint_as_float(v.y)),
                           fmaxf(mx.z, __uint_as_float(v.z)), fmaxf(mx.w, __uint_as_float(v.w)));
        }
        ra[a] = mx;
      }
#pragma unroll
      for (int b = 0; b < NPB; ++b) {
        const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsB, woff[b], sb, 0);
        rb[b] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
      g_c0 += BK;
      return;
    }
    if (BUF) {
      const int tap = g_tap;
      const unsigned sa = (unsigned)(((g_ty * W + g_tx) * ld_in + g_c0) * 4);
      const unsigned sb = (unsigned)((tap * Cin + g_c0) * 4);
      g_c0 += BK;
      if (g_c0 == Cin) {
        g_c0 = 0;
        ++g_tap;
        if (++g_tx == ks) { g_tx = 0; ++g_ty; }
      }
#pragma unroll
      for (int a = 0; a < NPA; ++a) {
        const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsA, ((vmask[a] >> tap) & 1u) ? rowoff[a] : OOB, sa, 0);
        ra[a] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
#pragma unroll
      for (int b = 0; b < NPB; ++b) {
        const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsB, woff[b], sb, 0);
        rb[b] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
      return;
    }
    const int tap = it / nchunk, c0 = (it - tap * nchunk) * BK;
    const int dy = tap / ks - pad, dx = tap % ks - pad;
#pragma unroll
    for (int a = 0; a < NPA; ++a) {
      const int yy = py[a] + dy, xx = px[a] + dx;
      const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
      ra[a] = ok ? *reinterpret_cast<const float4 *>(pb[a] + ((ptrdiff_t)dy * W + dx) * ld_in + c0)
                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int b = 0; b < NPB; ++b)
      rb[b] = wok[b] ? *reinterpret_cast<const float4 *>(wb[b] + (size_t)tap * Cin + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto lstore = [&]() {
#pragma unroll
    for (int a = 0; a < NPA; ++a) {
      float *d = As + (4 * q) * LDA + ri + PPP * a;
      d[0] = ra[a].x; d[LDA] = ra[a].y; d[2 * LDA] = ra[a].z; d[3 * LDA] = ra[a].w;
    }
#pragma unroll
    for (int b = 0; b < NPB; ++b) {
      if (ri + PPP * b < BN) {
        float *d = Bs + (4 * q) * LDB + ri + PPP * b;
        d[0] = rb[b].x; d[LDB] = rb[b].y; d[2 * LDB] = rb[b].z; d[3 * LDB] = rb[b].w;
      }
    }
  };

  gload(0);
  lstore();
  __syncthreads();
  const float *ap = As + (lane >> 5) * LDA + 32 * wave + (lane & 31);
  const float *bp = Bs + (lane >> 5) * LDB + (lane & 31);
  for (int it = 0; it < nit; ++it) {
    if (it + 1 < nit && !(EXP & 1)) gload(it + 1);
    const float *apc = ap, *bpc = bp;
#pragma unroll
    for (int kk = 0; kk < ((EXP & 4) ? 1 : BK / 2); ++kk) {
      const float a = apc[2 * kk * LDA];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float b = bpc[2 * kk * LDB + 32 * t];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
    if (it + 1 < nit) {
      if (!(EXP & 2)) lstore();
      __syncthreads();
    }
  }
  // epilogue: acc[t][r] = D[row = (r&3) + 8(r>>2) + 4(lane>>5)][col = lane&31]
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int co = n0 + 32 * t + (lane & 31);
    if (co < Cout) {
      const float bb = bias[co];
      const int sg = (co < dst.end[0]) ? 0 : ((co < dst.end[1]) ? 1 : 2);
      const int cbase = (sg == 0) ? 0 : dst.end[sg - 1];
      float *op = dst.p[sg] + dst.off[sg] + (co - cbase);
      const int ld = dst.ld[sg];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M && (!(EXP & 8) || acc[t][r] == 123.456f)) op[(size_t)m * ld] = fmaxf(acc[t][r] + bb, 0.f);
      }
    }
  }
}

// ---- inception branch 4 in ONE launch: 3x3 stride-1 pad-1 max pool + 1x1 convolution (googlenet1.py:213-214) -----------
// The separate pool kernel reads the block input and writes a pooled copy that the convolution reads back: nine launches,
// 1.08 of a batch's 18 ms, all of it HBM traffic (the tensors are 0.1-0.4 GB).  Taking the nine taps inside the tile fetch
// (k_conv_igemm<POOL>) was slower still -- nine tile reads through the L1 per chunk.  Here the tile's pixels PLUS one image row
// above and below are staged raw in LDS once per chunk (row-major, 16 KQ bytes per pixel, linear 16-byte stores), every thread
// pools its (pixel, channel quad) items from there -- nine ds_read_b128 and eight maxima -- and writes the result into the
// transposed A image of k_conv_igemm; the multiply loop, the chunk order and the epilogue are k_conv_igemm's, so the
// convolution sums in the same order: bit-identical to pool-then-convolve.  A tile of 128 pixels is whole image rows (W
// divides 128: 32, 16, 8), so the halo is W pixels on either side of the tile in the flattened pixel index; taps outside
// the image are masked per output pixel (the zero they are replaced by is the identity: activations are ReLU outputs).
// The pooling phase maps lanes to pixels so that the sixteen lanes of each ds_read_b128 lane group fall on distinct banks
// of the unpadded rows: pixel order [0 2 1 3 4 6 5 7] per eight lanes-blocks at KQ = 8, [0 4 5 1 6 2 3 7] at KQ = 4.
template <int BK, int BN>
__global__ __launch_bounds__(256) void k_poolconv(const float *__restrict__ in, int M, int H, int W, int Cin,
                                                   const float *__restrict__ wt, const float *__restrict__ bias, int Cout,
                                                   ConvDst dst) {
  constexpr int BM = 128, LDA = BM + 1, LDB = BN + 1, KQ = BK / 4, NT = BN / 32;
  constexpr int RMAX = BM + 2 * 32;                         // raw pixels at W = 32
  constexpr int NRAW = (RMAX * KQ + 255) / 256;             // raw float4 items per thread
  constexpr int NPOOL = BM * KQ / 256;                      // pooled items per thread
  constexpr int TPP = BK / 4, PPP = 256 / TPP, NPB = (BN + PPP - 1) / PPP;   // weight tile: as k_conv_igemm
  static_assert(BK == 32 || BK == 16, "pooled convolution: 32- or 16-channel chunks");
  __shared__ __attribute__((aligned(16))) float raw[RMAX * BK];
  __shared__ float As[BK * LDA];
  __shared__ float Bs[BK * LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int R = BM + 2 * W, nchunk = Cin / BK;
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rsA =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (unsigned)((size_t)M * Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wt), 0, (unsigned)((size_t)Cout * Cin * 4), 0x00020000);
  // ---- raw items: idx = tid + 256 r = rpx * KQ + quad (linear in LDS: byte idx * 16)
  unsigned rawoff[NRAW];
#pragma unroll
  for (int r = 0; r < NRAW; ++r) {
    const int idx = tid + 256 * r, rpx = idx / KQ, quad = idx - rpx * KQ;
    const long long m = (long long)m0 - W + rpx;
    rawoff[r] = (rpx < R && m >= 0 && m < M) ? (unsigned)(((size_t)m * Cin + 4 * quad) * 4) : OOB;
  }
  // ---- weight items (as k_conv_igemm's buffer form)
  const int q = tid % TPP, ri = tid / TPP;
  unsigned woff[NPB];
#pragma unroll
  for (int b = 0; b < NPB; ++b) {
    const int co = n0 + ri + PPP * b;
    woff[b] = ((ri + PPP * b < BN) && co < Cout) ? (unsigned)(((size_t)co * Cin + 4 * q) * 4) : OOB;
  }
  // ---- pooled items: quad = tid % KQ, pixel = 256 / KQ * a + permuted block
  const int pq = tid % KQ, pblk = tid / KQ;                                   // 32 (KQ = 8) or 64 (KQ = 4) blocks
  int ppx[NPOOL];
  unsigned pmask[NPOOL];
#pragma unroll
  for (int a = 0; a < NPOOL; ++a) {
    const int b8 = pblk & 7;
    const int perm = (KQ == 8) ? ((b8 & 4) | ((b8 & 1) << 1) | ((b8 >> 1) & 1))             // 0 2 1 3 4 6 5 7
                               : ((0x73261540u >> (4 * b8)) & 7);                            // 0 4 5 1 6 2 3 7
    const int px = (256 / KQ) * a + (pblk & ~7) + perm;
    ppx[a] = px;
    const int m = m0 + px;
    const bool ok = m < M;
    const int mm = ok ? m : 0;
    const int x = mm % W, y = (mm / W) % H;
    unsigned vm = 0;
    for (int tp = 0; tp < 9; ++tp) {
      const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
      if (ok && yy >= 0 && yy < H && xx >= 0 && xx < W) vm |= 1u << tp;
    }
    pmask[a] = vm;
  }

  f16_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  u4_t rr[NRAW];
  float4 rb[NPB];
  auto gload = [&](int it) {
    const unsigned sb = (unsigned)(it * BK * 4);
#pragma unroll
    for (int r = 0; r < NRAW; ++r) rr[r] = __builtin_amdgcn_raw_buffer_load_b128(rsA, rawoff[r], sb, 0);
#pragma unroll
    for (int b = 0; b < NPB; ++b) {
      const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsB, woff[b], sb, 0);
      rb[b] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
  };
  auto stage = [&]() {            // raw pixels and the weight tile into LDS
#pragma unroll
    for (int r = 0; r < NRAW; ++r)
      if (tid + 256 * r < RMAX * KQ) *reinterpret_cast<u4_t *>(raw + 4 * (tid + 256 * r)) = rr[r];
#pragma unroll
    for (int b = 0; b < NPB; ++b) {
      if (ri + PPP * b < BN) {
        float *d = Bs + (4 * q) * LDB + ri + PPP * b;
        d[0] = rb[b].x; d[LDB] = rb[b].y; d[2 * LDB] = rb[b].z; d[3 * LDB] = rb[b].w;
      }
    }
  };
  auto pool = [&]() {             // A image = max over the 3 x 3 neighbourhood, transposed as k_conv_igemm stores it
#pragma unroll
    for (int a = 0; a < NPOOL; ++a) {
      const float *c = raw + ((ppx[a] + W) * KQ + pq) * 4;
      float4 mx = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int d = ((tp / 3 - 1) * W + (tp % 3 - 1)) * KQ * 4;
        if ((pmask[a] >> tp) & 1u) mx = max4(mx, *reinterpret_cast<const float4 *>(c + d));   // (a masked tap is never read)
      }
      float *d = As + (4 * pq) * LDA + ppx[a];
      d[0] = mx.x; d[LDA] = mx.y; d[2 * LDA] = mx.z; d[3 * LDA] = mx.w;
    }
  };

  gload(0);
  stage();
  __syncthreads();
  pool();
  __syncthreads();
  const float *ap = As + (lane >> 5) * LDA + 32 * wave + (lane & 31);
  const float *bp = Bs + (lane >> 5) * LDB + (lane & 31);
  for (int it = 0; it < nchunk; ++it) {
    if (it + 1 < nchunk) gload(it + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const float a = ap[2 * kk * LDA];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float b = bp[2 * kk * LDB + 32 * t];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
    if (it + 1 < nchunk) {
      stage();
      __syncthreads();
      pool();
      __syncthreads();
    }
  }
  // epilogue: acc[t][r] = D[row = (r&3) + 8(r>>2) + 4(lane>>5)][col = lane&31]
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int co = n0 + 32 * t + (lane & 31);
    if (co < Cout) {
      const float bb = bias[co];
      float *op = dst.p[0] + dst.off[0] + co;
      const int ld = dst.ld[0];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M) op[(size_t)m * ld] = fmaxf(acc[t][r] + bb, 0.f);
      }
    }
  }
}

template <int BK>
static int launch_poolconv(const float *in, int M, int H, int W, int Cin, const float *wt, const float *bias, int Cout,
                           const ConvDst &dst, hipStream_t st) {
  // the N tile of conv_dispatch (the same choice: the same summation order and the same tiles as pool-then-convolve)
  const int bn = Cout < 48 ? 32 : ((sf_cdiv(Cout, 96) * 96 <= sf_cdiv(Cout, 64) * 64) ? 96 : 64);
  const dim3 grid(sf_cdiv(M, 128), sf_cdiv(Cout, bn));
  if (bn == 96) hipLaunchKernelGGL((k_poolconv<BK, 96>), grid, dim3(256), 0, st, in, M, H, W, Cin, wt, bias, Cout, dst);
  else if (bn == 64) hipLaunchKernelGGL((k_poolconv<BK, 64>), grid, dim3(256), 0, st, in, M, H, W, Cin, wt, bias, Cout, dst);
  else hipLaunchKernelGGL((k_poolconv<BK, 32>), grid, dim3(256), 0, st, in, M, H, W, Cin, wt, bias, Cout, dst);
  SF_LAUNCH_CHECK("k_poolconv");
  return 0;
}

// ---- head: global average pool -> FC(1024 -> 2) -> softmax[:,1]; NODATA where the input plane is NODATA ------------
// (googlenet1.py:87-89,:156-161; cnn_pred_pipeline.py:177-189)
__global__ __launch_bounds__(256) void k_head(const float *__restrict__ in, int HW, int C, const float *__restrict__ fcw,
                                               const float *__restrict__ fcb, const float *__restrict__ plane, long long tile0,
                                               float nodata, float *__restrict__ out) {
  __shared__ float red[2][256];
  const int t = blockIdx.x, tid = threadIdx.x;
  float d0 = 0.f, d1 = 0.f;
  const float inv = 1.0f / (float)HW;
  for (int c = tid; c < C; c += 256) {
    float s = 0.f;
    const float *col = in + (size_t)t * HW * C + c;
    int p = 0;
    for (; p + 16 <= HW; p += 16) {       // sixteen loads in flight, summed in position order (the order of the plain loop)
      float v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = col[(size_t)(p + k) * C];
#pragma unroll
      for (int k = 0; k < 16; ++k) s += v[k];
    }
    for (; p < HW; ++p) s += col[(size_t)p * C];
    const float a = s * inv;
    d0 = fmaf(a, fcw[c], d0);
    d1 = fmaf(a, fcw[C + c], d1);
  }
  red[0][tid] = d0;
  red[1][tid] = d1;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) { red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; }
    __syncthreads();
  }
  if (tid == 0) {
    const float l0 = red[0][0] + fcb[0], l1 = red[1][0] + fcb[1];
    const float mx = fmaxf(l0, l1);
    const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
    float p = e1 / (e0 + e1);
    if (plane && plane[tile0 + t] == nodata) p = nodata;
    out[tile0 + t] = p;
  }
}

// returns 1 when the buffer form cannot take these operands (>= 2 GB) and POOL was asked for: the caller pools separately
template <int BK, int BN, bool POOL = false>
int launch_conv(const float *in, int M, int H, int W, int Cin, int ld_in, const float *wt, const float *bias, int Cout,
                int ks, const ConvDst &dst, hipStream_t st) {
  dim3 grid(sf_cdiv(M, 128), sf_cdiv(Cout, BN));
  const int fk = POOL ? 3 : ks;
  const size_t abytes = ((size_t)M * ld_in + 2 * ((size_t)(fk >> 1) * W + (fk >> 1)) * ld_in) * 4 + 64;
  const size_t bbytes = (size_t)Cout * ks * ks * Cin * 4;
  const size_t lim = (size_t)0x7ff00000 - ((size_t)fk * W + fk) * ld_in * 4;      // offsets + the tap offset stay below 2^31
  const bool buf = abytes < lim && bbytes < lim && sf_tune().cnn_conv_variant != 1;
  if (POOL) {
    if (!buf) return 1;
    hipLaunchKernelGGL((k_conv_igemm<BK, BN, 0, true, POOL>), grid, dim3(256), 0, st, in, M, H, W, Cin, ld_in, wt, bias, Cout, ks, dst);
  } else if (buf) {
    hipLaunchKernelGGL((k_conv_igemm<BK, BN, 0, true>), grid, dim3(256), 0, st, in, M, H, W, Cin, ld_in, wt, bias, Cout, ks, dst);
  } else {
    hipLaunchKernelGGL((k_conv_igemm<BK, BN, 0, false>), grid, dim3(256), 0, st, in, M, H, W, Cin, ld_in, wt, bias, Cout, ks, dst);
  }
  SF_LAUNCH_CHECK("k_conv_igemm");
  return 0;
}

template <bool POOL = false>
static int conv_dispatch(const float *in, int N, int H, int W, int Cin, int ld_in, const float *w, const float *bias,
                         int Cout, int ksize, const ConvDst &dst, hipStream_t st) {
  const long long Ml = (long long)N * H * W;
  if (Ml > 2000000000LL) { sf_set_error("sf_cnn_conv: batch too large"); return -1; }
  const int M = (int)Ml;
  // N tile: 96 where it pads the output channels no more than 64 does (288 = 3 x 96, 192 = 2 x 96), 32 for narrow layers
  const int bn = Cout < 48 ? 32 : ((sf_cdiv(Cout, 96) * 96 <= sf_cdiv(Cout, 64) * 64) ? 96 : 64);
#define SF_CONV(BK)                                                                                         \
  return bn == 96 ? launch_conv<BK, 96, POOL>(in, M, H, W, Cin, ld_in, w, bias, Cout, ksize, dst, st)       \
       : bn == 64 ? launch_conv<BK, 64, POOL>(in, M, H, W, Cin, ld_in, w, bias, Cout, ksize, dst, st)       \
                  : launch_conv<BK, 32, POOL>(in, M, H, W, Cin, ld_in, w, bias, Cout, ksize, dst, st)
  if (Cin % 32 == 0) { SF_CONV(32); }
  if (Cin % 16 == 0) { SF_CONV(16); }
  SF_CONV(8);
#undef SF_CONV
}

}  // namespace

extern "C" {

int sf_cnn_prepare_plane(const float *plane, int H, int W, float vmin, float vmax, float mean, float stdv, int dim,
                         float *padded, void *stream) {
  if (!plane || !padded || H < 1 || W < 1 || dim < 2 || !(vmax > vmin) || !(stdv != 0.f)) {
    sf_set_error("sf_cnn_prepare_plane: bad argument");
    return -1;
  }
  const int Hp = H + dim - 1, Wp = W + dim - 1;
  hipLaunchKernelGGL(k_prepare, dim3(sf_cdiv(Hp * Wp, 256)), dim3(256), 0, (hipStream_t)stream, plane, H, W, vmin, vmax,
                     mean, stdv, dim / 2, Hp, Wp, padded);
  SF_LAUNCH_CHECK("k_prepare");
  return 0;
}

int sf_cnn_conv1(const float *padded, int Hp, int Wp, int W, long long tile0, int ntiles, const float *w,
                 const float *bias, float *out, void *stream) {
  if (!padded || !w || !bias || !out || ntiles < 1 || W < 1 || Wp != W + 255 || tile0 < 0 ||
      (tile0 + ntiles + W - 1) / W > Hp - 255) {
    sf_set_error("sf_cnn_conv1: bad argument");
    return -1;
  }
  hipLaunchKernelGGL(k_conv1, dim3(64, ntiles), dim3(256), 0, (hipStream_t)stream, padded, Wp, W, tile0, w, bias, out);
  SF_LAUNCH_CHECK("k_conv1");
  return 0;
}

int sf_cnn_conv1_pool(const float *padded, int Hp, int Wp, int W, long long tile0, int ntiles, const float *w,
                      const float *bias, float *out, void *stream) {
  if (!padded || !w || !bias || !out || ntiles < 1 || W < 1 || Wp != W + 255 || tile0 < 0 ||
      (tile0 + ntiles + W - 1) / W > Hp - 255) {
    sf_set_error("sf_cnn_conv1_pool: bad argument");
    return -1;
  }
  if (sf_tune().cnn_variant == 1) {                  // the 8 x 8 form with the conv tile in LDS
    if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_conv1_pool<0>), cp_lds_bytes())) return rc;
    hipLaunchKernelGGL(k_conv1_pool<0>, dim3(64, ntiles), dim3(256), cp_lds_bytes(), (hipStream_t)stream, padded, Wp, W, tile0, w,
                       bias, out);
  } else {
    if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_conv1_pool16<0>), c2_lds_bytes())) return rc;
    hipLaunchKernelGGL(k_conv1_pool16<0>, dim3(16, ntiles), dim3(C2_NT), c2_lds_bytes(), (hipStream_t)stream, padded, Wp, W, tile0,
                       w, bias, out);
  }
  SF_LAUNCH_CHECK("k_conv1_pool");
  return 0;
}

int sf_cnn_maxpool(const float *in, int N, int H, int W, int C, int ksize, int stride, int pad, float *out, int Ho,
                   int Wo, void *stream) {
  if (!in || !out || N < 1 || (C & 3) || ksize < 1 || stride < 1) {
    sf_set_error("sf_cnn_maxpool: bad argument (channels must be a multiple of 4)");
    return -1;
  }
  const size_t total = (size_t)N * ((Ho + PR - 1) / PR) * Wo * (C / 4);
  hipLaunchKernelGGL(k_maxpool, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, N, H, W, C,
                     ksize, stride, pad, out, Ho, Wo);
  SF_LAUNCH_CHECK("k_maxpool");
  return 0;
}

int sf_cnn_conv(const float *in, int N, int H, int W, int Cin, int ld_in, const float *w, const float *bias, int Cout,
                int ksize, float *out, int ld_out, int ch_off, void *stream) {
  if (!in || !w || !bias || !out || N < 1 || (ksize != 1 && ksize != 3) || (Cin & 7) || (ld_in & 3) || Cin > ld_in ||
      ch_off < 0 || ch_off + Cout > ld_out) {
    sf_set_error("sf_cnn_conv: bad argument (ksize 1|3, Cin multiple of 8)");
    return -1;
  }
  ConvDst d{};
  d.p[0] = d.p[1] = d.p[2] = out;
  d.ld[0] = d.ld[1] = d.ld[2] = ld_out;
  d.off[0] = d.off[1] = d.off[2] = ch_off;
  d.end[0] = d.end[1] = d.end[2] = Cout;
  return conv_dispatch(in, N, H, W, Cin, ld_in, w, bias, Cout, ksize, d, (hipStream_t)stream);
}

int sf_cnn_pool_conv(const float *in, int N, int H, int W, int Cin, int ld_in, const float *w, const float *bias, int Cout,
                     float *out, int ld_out, int ch_off, float *pooled_scratch, void *stream) {
  if (!in || !w || !bias || !out || !pooled_scratch || N < 1 || (Cin & 7) || (ld_in & 3) || Cin != ld_in || ch_off < 0 ||
      ch_off + Cout > ld_out) {
    sf_set_error("sf_cnn_pool_conv: bad argument (Cin multiple of 8, dense input)");
    return -1;
  }
  ConvDst d{};
  d.p[0] = d.p[1] = d.p[2] = out;
  d.ld[0] = d.ld[1] = d.ld[2] = ld_out;
  d.off[0] = d.off[1] = d.off[2] = ch_off;
  d.end[0] = d.end[1] = d.end[2] = Cout;
  // round 5: the pool taken from the tile staged in LDS (k_poolconv), where the geometry allows -- whole image rows per
  // 128-pixel tile, 16- or 32-channel chunks, operands below 2 GB; key 18 = 3: pool into the scratch tensor, then convolve
  // (the form every other geometry takes, and the tests' cross-check: bit-identical)
  const long long Ml = (long long)N * H * W;
  if (sf_tune().cnn_pool_variant == 0 && W >= 1 && W <= 32 && 128 % W == 0 &&
      (Cin % 16) == 0 && Ml * Cin * 4 < 0x7ff00000LL && (long long)Cout * Cin * 4 < 0x7ff00000LL) {
    if (Cin % 32 == 0) return launch_poolconv<32>(in, (int)Ml, H, W, Cin, w, bias, Cout, d, (hipStream_t)stream);
    return launch_poolconv<16>(in, (int)Ml, H, W, Cin, w, bias, Cout, d, (hipStream_t)stream);
  }
  if (sf_tune().cnn_pool_variant == 1) {     // measured slower (27.2 k vs 28.0 k tiles/s): nine tile reads through the L1
    const int rc = conv_dispatch<true>(in, N, H, W, Cin, ld_in, w, bias, Cout, 1, d, (hipStream_t)stream);
    if (rc <= 0) return rc;
  }
  // default: pool into the scratch tensor, then convolve.  The block input is a concatenation of ReLU outputs (>= 0): the
  // strip kernel with zero-returning out-of-range loads applies (k_maxpool_s1); variant 2 = the general pool kernel
  if ((Cin & 3) == 0 && (size_t)N * H * W * Cin * 4 < 0x7ff00000u && sf_tune().cnn_pool_variant != 2) {
    const size_t total = (size_t)N * ((H + PR - 1) / PR) * W * (Cin / 4);
    hipLaunchKernelGGL(k_maxpool_s1, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, N, H, W, Cin,
                       pooled_scratch);
    SF_LAUNCH_CHECK("k_maxpool_s1");
  } else if (int rc = sf_cnn_maxpool(in, N, H, W, Cin, 3, 1, 1, pooled_scratch, H, W, stream)) {
    return rc;
  }
  return conv_dispatch(pooled_scratch, N, H, W, Cin, Cin, w, bias, Cout, 1, d, (hipStream_t)stream);
}

int sf_cnn_conv_split3(const float *in, int N, int H, int W, int Cin, int ld_in, const float *w, const float *bias,
                       int c0, int c1, int c2, float *out0, int ld0, int off0, float *out1, int ld1, int off1,
                       float *out2, int ld2, int off2, void *stream) {
  if (!in || !w || !bias || !out0 || !out1 || !out2 || N < 1 || (Cin & 7) || (ld_in & 3) || Cin > ld_in || c0 < 1 ||
      c1 < 1 || c2 < 1 || off0 + c0 > ld0 || off1 + c1 > ld1 || off2 + c2 > ld2) {
    sf_set_error("sf_cnn_conv_split3: bad argument");
    return -1;
  }
  ConvDst d{};
  d.p[0] = out0; d.p[1] = out1; d.p[2] = out2;
  d.ld[0] = ld0; d.ld[1] = ld1; d.ld[2] = ld2;
  d.off[0] = off0; d.off[1] = off1; d.off[2] = off2;
  d.end[0] = c0; d.end[1] = c0 + c1; d.end[2] = c0 + c1 + c2;
  return conv_dispatch(in, N, H, W, Cin, ld_in, w, bias, c0 + c1 + c2, 1, d, (hipStream_t)stream);
}

int sf_cnn_head(const float *in, int ntiles, int HW, int C, const float *fcw, const float *fcb, const float *plane,
                long long tile0, float nodata, float *out, void *stream) {
  if (!in || !fcw || !fcb || !out || ntiles < 1) { sf_set_error("sf_cnn_head: bad argument"); return -1; }
  hipLaunchKernelGGL(k_head, dim3(ntiles), dim3(256), 0, (hipStream_t)stream, in, HW, C, fcw, fcb, plane, tile0, nodata,
                     out);
  SF_LAUNCH_CHECK("k_head");
  return 0;
}

}  // extern "C"
