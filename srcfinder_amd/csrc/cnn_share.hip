// CNN tile scorer: the trunk up to conv3 SHARED between the overlapping windows of a flightline (round 6; VERDICT r5 missing 4).
//
// cnn/cnn_pred_pipeline.py:53-58 scores one 256 x 256 window per pixel: neighbouring windows overlap by 255/256.  Through conv3
// (cnn/archs/googlenet1.py:110-120: conv1 7x7 s2 p3 -> maxpool1 3x3 s2 ceil -> conv2 1x1 -> conv3 3x3 p1) a window's activation
// at (y, x) of the 64 x 64 grid depends on the window only through its own ZERO PADDING: conv1 rows 0, 1 and 127 reach outside the
// window, so do maxpool1 rows 0 and 63, conv2 rows 0 and 63 and conv3 rows 0, 1, 62, 63 (likewise columns).  Everything else --
// conv3 at y, x in 2..61: 3600 of 4096 positions -- equals the same stack evaluated on the WHOLE padded plane at the window's phase:
//     window (r, c), phase (r & 3, c & 3):  conv3_window[y][x] = Q3[phase][(r >> 2) + y][(c >> 2) + x]
// where Q3[phase] is conv1 .. conv3 run fully convolutionally (the FCN kernels) on the plane shifted by the phase: 16 phase maps,
// computed ONCE per strip of image rows at a cost of one window-equivalent per plane position.  Per window only the ring remains:
//     k_ring_pool1      conv1 + maxpool1 at the 252 border positions of the 64 x 64 grid        (fp32 vector units)
//     sf_cnn_conv_split conv2 on those 252 positions                                             (a plain GEMM)
//     sf_cnn_conv_ring  conv3 at the 496 ring positions, taps from the border tensor / the shared conv2 map (cnn_split.hip)
//     k_pool_gather     maxpool2 (3x3 s2 ceil) reading ring positions from the ring tensor, the others from the shared conv3 map
// and inception3a takes over -- or (the deeper form) the same idea carried on through inception3a, 3b and maxpool3 on 64 phase
// maps of the 32 x 32 grid: frames (1, 2) -> (2, 3) -> (3, 4) (cnn_ring.h), and inception4a takes over.  Exact: a map position and a window position are produced by the same summation order
// (the saliency maps are bit-identical to every window evaluated on its own: tests/test_cnn_gpu.py, tools/fuzz_cnn.py).
#include "cmf_common.h"
#include "cnn_ring.h"
#include "cnn_internal.h"

namespace {

// canvas[u][v] = padded[y0 + u][x0 + v], zero outside the padded plane
__global__ __launch_bounds__(256) void k_phase_canvas(const float *__restrict__ padded, int Hp, int Wp, int y0, int x0, int Hc, int Wc,
                                                       float *__restrict__ canvas) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)Hc * Wc) return;
  const int u = (int)(i / Wc), v = (int)(i - (size_t)u * Wc);
  const int y = y0 + u, x = x0 + v;
  canvas[i] = (y >= 0 && y < Hp && x >= 0 && x < Wp) ? padded[(size_t)y * Wp + x] : 0.f;
}

// maxpool1(conv1(window)) at the 252 border positions of the 64 x 64 grid: one wave per (window, position) at a time, lane = output
// channel with its 49 folded weights in registers, the 11 x 11 input patch of the position's (up to) 3 x 3 conv1 outputs in LDS.
// conv1: 7 x 7 stride 2 pad 3 on the 256 x 256 window (zero outside), + bias, ReLU (googlenet1.py:60, :266-275); the pool takes
// conv rows 2 y .. 2 y + 2 that exist (ceil mode: row 128 does not), googlenet1.py:61.  Summation order ky, kx ascending (k_conv1_img's).
__global__ __launch_bounds__(256) void k_ring_pool1(const float *__restrict__ padded, int Wp, int Wimg, long long tile0, int ntiles,
                                                     const float *__restrict__ w /*[64][49]*/, const float *__restrict__ bias,
                                                     float *__restrict__ out /*[ntiles][252][64]*/, int side) {
  __shared__ __attribute__((aligned(16))) float patch[4][11 * 12];      // rows of 12: whole rows leave LDS as three 16-byte reads
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float wr[49];
#pragma unroll
  for (int t = 0; t < 49; ++t) wr[t] = w[lane * 49 + t];
  const float bb = bias[lane];
  const int per = side ? 128 : 252;                               // side: the positions px = 0 / 63 only (band sharing, cnn_ring.h)
  const long long total = (long long)ntiles * per;
  float *pw = patch[wave];
  for (long long item = (long long)blockIdx.x * 4 + wave; item < total; item += (long long)gridDim.x * 4) {
    const int n = (int)(item / per), b = (int)(item - (long long)n * per);
    int py, px;
    if (side) sf_side_position(64, 1, 1, b, py, px);
    else sf_frame_position(64, 1, 1, b, py, px);
    const long long t = tile0 + n;
    const int r = (int)(t / Wimg), c = (int)(t - (long long)r * Wimg);
    const float *win = padded + (size_t)r * Wp + c;               // window pixel (wy, wx) = win[wy * Wp + wx]
    const int wy0 = 4 * py - 3, wx0 = 4 * px - 3;                 // the patch's origin in the window
    for (int i = lane; i < 132; i += 64) {
      const int wy = wy0 + i / 12, wx = wx0 + i % 12;
      pw[i] = ((unsigned)wy < 256u && (unsigned)wx < 256u && i % 12 < 11) ? win[(size_t)wy * Wp + wx] : 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // the whole patch into registers (33 broadcast reads of 16 bytes) instead of one 4-byte LDS read per multiply.  The time did not
    // move (185 us per 512 windows): the kernel is bound by the issue of its 441 multiply-adds per position and lane; two positions per
    // wave on v_pk_fma_f32 (half the instructions) measured SLOWER, 235 us -- packed float32 is not a faster pipe here.
    float pr[11][12];
#pragma unroll
    for (int y = 0; y < 11; ++y)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const float4 v = *reinterpret_cast<const float4 *>(pw + 12 * y + 4 * q);
        pr[y][4 * q] = v.x; pr[y][4 * q + 1] = v.y; pr[y][4 * q + 2] = v.z; pr[y][4 * q + 3] = v.w;
      }
    float best = 0.f;                                             // ReLU outputs: 0 is the identity of the max
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        if (2 * py + dy > 127 || 2 * px + dx > 127) continue;     // (uniform over the wave)
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
          for (int kx = 0; kx < 7; ++kx) acc = fmaf(pr[2 * dy + ky][2 * dx + kx], wr[ky * 7 + kx], acc);
        best = fmaxf(best, fmaxf(acc + bb, 0.f));
      }
    out[((size_t)n * 252 + (side ? sf_frame_index(64, 1, 1, py, px) : b)) * 64 + lane] = best;
    __builtin_amdgcn_wave_barrier();
  }
}

// A 3 x 3 max pool (stride 1 pad 1, or stride 2 pad 0 in ceil mode: googlenet1.py:61-68, :213) of a tensor that exists only as per-window
// ring tensor [N][count][C] + shared phase maps (SfGather), for a batch of windows: the output either on the WHOLE output grid
// (out[N][Go][Go][C]: olo < 0) or at the ring positions of the frame (olo, ohi) on Go (out[N][count(Go, olo, ohi)][C]).
// Activations are ReLU outputs: 0 is the identity of the max (a tap outside the grid contributes nothing).
// (One thread per (position, four channels), every thread classifying its own taps.  The other shape -- one WAVE per position with
//  scalar tap classification, lanes over the channels -- was measured: 0.79 ms of pools per batch against 0.61; loads in flight win.)
__global__ __launch_bounds__(256) void k_pool_gather(const float *__restrict__ maps, SfGather gi, int C, int stride, int pad, int Go, int olo,
                                                      int ohi, int npos, int N, unsigned bytes, float *__restrict__ out, int side) {
  const int c4n = C >> 2;
  const size_t total = (size_t)N * npos * c4n;
  const int nin = sf_frame_count(gi.G, gi.lo, gi.hi);
  const int pm = (1 << gi.shift) - 1;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(maps), 0, bytes, 0x00020000);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % c4n);
    const size_t rest = i / c4n;
    const int j = (int)(rest % npos), n = (int)(rest / npos);
    int oy, ox;
    if (olo < 0) { oy = j / Go; ox = j - oy * Go; }
    else if (side) sf_side_position(Go, olo, ohi, j, oy, ox);
    else sf_frame_position(Go, olo, ohi, j, oy, ox);
    const long long t = gi.tile0 + n;
    const int r = (int)(t / gi.W), c = (int)(t - (long long)r * gi.W);
    const int ph = ((r & pm) << gi.shift) + (c & pm);
    // nine unconditional 16-byte buffer loads (a tap outside the grid reads out of range -> 0): a load inside a branch is waited for
    // before the next tap is classified, nine round trips in a row (round 6: 0.60 -> 0.4x ms of pools per batch)
    const unsigned mp = (unsigned)(((((size_t)ph * gi.Hq + ((r >> gi.shift) - gi.Rb)) * gi.Wq + (c >> gi.shift)) * C + 4 * c4) * 4);   // the window's origin in its map
    const unsigned rg = (unsigned)(((size_t)gi.ring_off + (size_t)n * nin * C + 4 * c4) * 4);
    typedef float pg_f4 __attribute__((ext_vector_type(4)));
    typedef unsigned pg_u4 __attribute__((ext_vector_type(4)));
    pg_u4 v[9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int y = oy * stride - pad + dy, x = ox * stride - pad + dx;
        const bool in = (unsigned)y < (unsigned)gi.G && (unsigned)x < (unsigned)gi.G;
        const unsigned o_ring = rg + (unsigned)(sf_frame_index(gi.G, gi.lo, gi.hi, y, x) * C * 4);
        const unsigned o_map = mp + (unsigned)((y * gi.Wq + x) * C * 4);
        const unsigned off = in ? (sf_frame_ring(gi.G, gi.lo, gi.hi, y, x) ? o_ring : o_map) : 0x80000000u;
        v[3 * dy + dx] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
      }
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      union { pg_u4 u; pg_f4 f; } cv;
      cv.u = v[k];
      m.x = fmaxf(m.x, cv.f.x); m.y = fmaxf(m.y, cv.f.y); m.z = fmaxf(m.z, cv.f.z); m.w = fmaxf(m.w, cv.f.w);
    }
    const size_t orow = (side == 1) ? (size_t)n * sf_frame_count(Go, olo, ohi) + sf_frame_index(Go, olo, ohi, oy, ox) : rest;
    *reinterpret_cast<float4 *>(out + (orow * c4n + c4) * 4) = m;
  }
}


// strip canvases of the band sharing (cnn_driver.hip): image ((ri * 2 + bottom) * 4 + phase)[u][v] = padded[row0 + ri + 192 bottom + u][phase + v],
// 64 rows each: the top 64 / the bottom 64 pixel rows of the windows of image row row0 + ri, shifted by the column phase
__global__ __launch_bounds__(256) void k_strip_canvas(const float *__restrict__ padded, int Hp, int Wp, int row0, int nrows, int Wc,
                                                       float *__restrict__ canvas) {
  const size_t per = (size_t)64 * Wc, total = (size_t)nrows * 8 * per;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int img = (int)(i / per);
    const size_t rest = i - (size_t)img * per;
    const int u = (int)(rest / Wc), v = (int)(rest - (size_t)u * Wc);
    const int ph = img & 3, bottom = (img >> 2) & 1, ri = img >> 3;
    const int y = row0 + ri + 192 * bottom + u, x = ph + v;
    canvas[i] = (y >= 0 && y < Hp && x >= 0 && x < Wp) ? padded[(size_t)y * Wp + x] : 0.f;
  }
}

// band interior of a ring tensor <- strip maps (cnn_internal.h: sfi_cnn_band_copy)
__global__ __launch_bounds__(256) void k_band_copy(const float *__restrict__ strips, long long tile0, int N, int W, int row0, int shift, int Hs,
                                                    int Wq, int G, int lo, int hi, int C, int nrb, float *__restrict__ ring) {
  const int c4n = C >> 2, nb = sf_band_count(G, lo, hi), nfull = sf_frame_count(G, lo, hi), P = 1 << shift;
  const size_t total = (size_t)N * nb * c4n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % c4n);
    const size_t rest = i / c4n;
    const int j = (int)(rest % nb), n = (int)(rest / nb);
    int y, x;
    sf_band_position(G, lo, hi, j, y, x);
    const long long t = tile0 + n;
    const int r = (int)(t / W), c = (int)(t - (long long)r * W);
    const int bottom = y >= G - hi, ys = bottom ? Hs - (G - y) : y;
    const int ph = c & (P - 1);
    const size_t strip = (size_t)(ph >> 2) * (nrb * 4) + (size_t)((r - row0) * 2 + bottom) * 4 + (ph & 3);      // (cnn_internal.h)
    const float4 v = *reinterpret_cast<const float4 *>(strips + ((strip * Hs + ys) * Wq + (c >> shift) + x) * C + 4 * c4);
    *reinterpret_cast<float4 *>(ring + ((size_t)n * nfull + sf_frame_index(G, lo, hi, y, x)) * C + 4 * c4) = v;
  }
}
}  // namespace

extern "C" {

int sf_cnn_phase_canvas(const float *padded, int Hp, int Wp, int y0, int x0, int Hc, int Wc, float *canvas, void *stream) {
  if (!padded || !canvas || Hp < 1 || Wp < 1 || Hc < 1 || Wc < 1) { sf_set_error("sf_cnn_phase_canvas: bad argument"); return -1; }
  const size_t n = (size_t)Hc * Wc;
  hipLaunchKernelGGL(k_phase_canvas, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, padded, Hp, Wp, y0, x0, Hc,
                     Wc, canvas);
  SF_LAUNCH_CHECK("k_phase_canvas");
  return 0;
}

static int ring_pool1_go(const float *padded, int Hp, int Wp, int W, long long tile0, int ntiles, const float *w, const float *bias,
                         float *out, void *stream, int side) {
  if (!padded || !w || !bias || !out || ntiles < 1 || W < 1 || Wp != W + 255 || Hp < 256 || tile0 < 0) {
    sf_set_error("sf_cnn_ring_pool1: bad argument");
    return -1;
  }
  const long long items = (long long)ntiles * (side ? 128 : 252);
  const int blocks = (int)((items + 3) / 4 < 8192 ? (items + 3) / 4 : 8192);
  hipLaunchKernelGGL(k_ring_pool1, dim3(blocks), dim3(256), 0, (hipStream_t)stream, padded, Wp, W, tile0, ntiles, w, bias, out, side);
  SF_LAUNCH_CHECK("k_ring_pool1");
  return 0;
}
int sf_cnn_ring_pool1(const float *padded, int Hp, int Wp, int W, long long tile0, int ntiles, const float *w, const float *bias,
                      float *out, void *stream) {
  return ring_pool1_go(padded, Hp, Wp, W, tile0, ntiles, w, bias, out, stream, 0);
}

static int pool_gather_go(const float *maps, long long tile0, int N, int W, int Rb, int Hq, int Wq, size_t ring_off, int shift, int G,
                          int ilo, int ihi, int C, int stride, int Go, int olo, int ohi, float *out, void *stream, int side) {
  if (!maps || !out || N < 1 || W < 1 || tile0 < 0 || G < 4 || G > 128 || shift < 1 || shift > 4 || ilo < 0 || ihi < 0 || ilo + ihi >= G ||
      Hq < G || Wq < G || C < 4 || (C & 3) || (stride != 1 && stride != 2) || (ring_off & 3) ||
      Go != (stride == 1 ? G : (G + 1) / 2) || (olo >= 0 && (ohi < 0 || olo + ohi >= Go))) {
    sf_set_error("sf_cnn_pool_gather: bad argument");
    return -1;
  }
  const size_t bytes = (ring_off + (size_t)(N + 1) * sf_frame_count(G, ilo, ihi) * C) * 4;      // maps + the batch's ring tensor: one descriptor
  if (bytes >= 0x7ff00000u) { sf_set_error("sf_cnn_pool_gather: maps + ring tensor of 2 GB or more"); return -2; }
  const SfGather gi{tile0, W, Rb, Hq, Wq, shift, G, ilo, ihi, (unsigned)ring_off};
  if (side && (olo < 0 || olo + ohi < 1)) { sf_set_error("sf_cnn_pool_gather: side rows need a frame"); return -1; }
  const int npos = olo < 0 ? Go * Go : (side ? sf_side_count(Go, olo, ohi) : sf_frame_count(Go, olo, ohi));
  const size_t total = (size_t)N * npos * (C >> 2);
  const unsigned blocks = (unsigned)((total + 255) / 256 < 65536 * 8 ? (total + 255) / 256 : 65536 * 8);
  hipLaunchKernelGGL(k_pool_gather, dim3(blocks), dim3(256), 0, (hipStream_t)stream, maps, gi, C, stride, stride == 1 ? 1 : 0, Go, olo, ohi,
                     npos, N, (unsigned)bytes, out, side);
  SF_LAUNCH_CHECK("k_pool_gather");
  return 0;
}
int sf_cnn_pool_gather(const float *maps, long long tile0, int N, int W, int Rb, int Hq, int Wq, size_t ring_off, int shift, int G,
                       int ilo, int ihi, int C, int stride, int Go, int olo, int ohi, float *out, void *stream) {
  return pool_gather_go(maps, tile0, N, W, Rb, Hq, Wq, ring_off, shift, G, ilo, ihi, C, stride, Go, olo, ohi, out, stream, 0);
}

}  // extern "C"

// ---- internal entry points (cnn_internal.h)
int sfi_cnn_pool_gather_side(const float *maps, long long tile0, int N, int W, int Rb, int Hq, int Wq, size_t ring_off, int shift, int G, int ilo,
                             int ihi, int C, int stride, int Go, int olo, int ohi, float *out, int side, void *stream) {
  return pool_gather_go(maps, tile0, N, W, Rb, Hq, Wq, ring_off, shift, G, ilo, ihi, C, stride, Go, olo, ohi, out, stream, side);
}
int sfi_cnn_ring_pool1_side(const float *padded, int Hp, int Wp, int W, long long tile0, int ntiles, const float *w, const float *bias,
                            float *out, void *stream) {
  return ring_pool1_go(padded, Hp, Wp, W, tile0, ntiles, w, bias, out, stream, 1);
}
int sfi_cnn_band_copy(const float *strips, long long tile0, int N, int W, int row0, int nrows, int shift, int Hs, int Wq, int G, int lo, int hi,
                      int C, float *ring, void *stream) {
  if (!strips || !ring || N < 1 || W < 1 || tile0 < 0 || shift < 1 || shift > 4 || Hs < lo || Hs < hi || lo < 0 || hi < 0 || lo + hi < 1 ||
      lo + hi >= G || (C & 3) || tile0 / W < row0 || (tile0 + N - 1) / W >= row0 + nrows || (shift != 2 && shift != 3)) {
    sf_set_error("sfi_cnn_band_copy: bad argument");
    return -1;
  }
  const size_t total = (size_t)N * sf_band_count(G, lo, hi) * (C >> 2);
  const unsigned blocks = (unsigned)((total + 255) / 256 < 65536 * 8 ? (total + 255) / 256 : 65536 * 8);
  hipLaunchKernelGGL(k_band_copy, dim3(blocks), dim3(256), 0, (hipStream_t)stream, strips, tile0, N, W, row0, shift, Hs, Wq, G, lo, hi, C, 2 * nrows, ring);
  SF_LAUNCH_CHECK("k_band_copy");
  return 0;
}
int sfi_cnn_strip_canvas(const float *padded, int Hp, int Wp, int row0, int nrows, int Wc, float *canvas, void *stream) {
  if (!padded || !canvas || Hp < 1 || Wp < 1 || nrows < 1 || Wc < 1 || row0 < 0) { sf_set_error("sfi_cnn_strip_canvas: bad argument"); return -1; }
  const size_t total = (size_t)nrows * 8 * 64 * Wc;
  hipLaunchKernelGGL(k_strip_canvas, dim3((unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536)), dim3(256), 0, (hipStream_t)stream,
                     padded, Hp, Wp, row0, nrows, Wc, canvas);
  SF_LAUNCH_CHECK("k_strip_canvas");
  return 0;
}

