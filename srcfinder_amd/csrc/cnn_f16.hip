// Reduced-precision option of the CNN tile scorer: float16 activations and weights, float32 accumulation on the
// matrix cores (v_mfma_f32_32x32x16_f16, 16x the fp32 MFMA rate).  This is the precision class the reference
// itself runs at on any Ampere-or-newer GPU (cuDNN's default TF32 convolutions: 10-bit mantissa operands, fp32
// accumulate); it is NOT the parity path -- the fp32 kernels of cnn_kernels.hip are -- and carries its own
// tolerance (tests: 5e-3 on the saliency).  Same graph, same NHWC layout, same fusions.
//
// Implicit GEMM: block tile 128 pixels x BN channels (BN = 128: 2x2 waves of 64x64; BN = 64: 4x1 waves of 32x64),
// k-chunk = 32 input channels of one tap.  With NHWC the 8 consecutive k a lane feeds to one MFMA are 16
// contiguous bytes of one pixel / one output channel's weight row, so both operand tiles are stored
// [row][k] with 80-byte rows (16-byte reads of 16 lanes at stride 80 B hit 16 distinct 16-byte slots) and are
// filled with plain 16-byte copies: no transposes anywhere.
#include "cmf_common.h"

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f16x_t __attribute__((ext_vector_type(16)));

namespace {

constexpr int HB_LD = 40;  // halves per LDS row (32 + 8 pad = 80 bytes)

struct ConvDstH {
  _Float16 *p[3];
  int ld[3], off[3], end[3];
};

// BUF: raw buffer loads for the tile fetch (cnn_kernels.hip: the lane's byte offset in a VGPR, an out-of-range offset for a
// tap outside the image / a row, channel chunk or output channel past the end, the (tap, chunk) offset in an SGPR) -- the
// pointer form's bounds tests and 64-bit addresses are ~150 instructions per chunk and wave beside EIGHT matrix
// instructions here.  Operands below 2 GB (host-checked).
typedef unsigned hu4_t __attribute__((ext_vector_type(4)));
template <int BN, bool BUF = false>
__global__ __launch_bounds__(256) void k_conv_igemm_f16(const _Float16 *__restrict__ in, int M, int H, int W, int Cin,
                                                         int ld_in, const _Float16 *__restrict__ wt,
                                                         const float *__restrict__ bias, int Cout, int ks, ConvDstH dst) {
  constexpr int BM = 128, BK = 32;
  constexpr int WN = (BN == 128) ? 2 : 1;          // waves along N
  constexpr int WM = 4 / WN;                        // waves along M
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);  // MFMA tiles per wave: 2x2 or 1x2
  constexpr int NPB = BN / 64;                      // B-tile 16-byte chunks per thread
  __shared__ __attribute__((aligned(16))) _Float16 As[BM * HB_LD];
  __shared__ __attribute__((aligned(16))) _Float16 Bs[BN * HB_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int q = tid & 3, ri = tid >> 2;             // 16-byte chunk q of row ri (64 rows per pass)
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int pad = ks >> 1, taps = ks * ks, nchunk = (Cin + BK - 1) / BK, nit = taps * nchunk;

  int py[2], px[2];
  const _Float16 *pb[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int m = m0 + ri + 64 * a;
    const bool ok = m < M;
    const int mm = ok ? m : 0;
    py[a] = ok ? (mm / W) % H : -100000;
    px[a] = mm % W;
    pb[a] = in + (size_t)mm * ld_in + 8 * q;
  }
  const _Float16 *wb[NPB];
  bool wok[NPB];
#pragma unroll
  for (int b = 0; b < NPB; ++b) {
    const int co = n0 + ri + 64 * b;
    wok[b] = co < Cout;
    wb[b] = wt + (size_t)(wok[b] ? co : 0) * taps * Cin + 8 * q;
  }
  f16x_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const h8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
  h8_t ra[2], rb[NPB];
  constexpr unsigned OOB = 0x80000000u;
  unsigned rowoff[2], vmask[2], woff[NPB];
  __amdgpu_buffer_rsrc_t rsA, rsB;
  int g_tap = 0, g_ty = 0, g_tx = 0, g_c0 = 0;
  if (BUF) {
    const size_t shift = ((size_t)pad * W + pad) * ld_in;          // taps are addressed from (y - pad, x - pad): offsets >= 0
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(in) - shift, 0,
                                            (unsigned)(((size_t)M * ld_in + 2 * shift) * 2 + 64), 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(wt), 0, (unsigned)((size_t)Cout * taps * Cin * 2), 0x00020000);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int m = m0 + ri + 64 * a;
      const int mm = m < M ? m : 0;
      rowoff[a] = (unsigned)(((size_t)mm * ld_in + 8 * q) * 2);
      unsigned vm = 0;
      for (int tp = 0; tp < taps; ++tp) {
        const int yy = py[a] + tp / ks - pad, xx = px[a] + tp % ks - pad;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) vm |= 1u << tp;
      }
      vmask[a] = vm;
    }
#pragma unroll
    for (int b = 0; b < NPB; ++b) {
      const int co = n0 + ri + 64 * b;
      woff[b] = wok[b] ? (unsigned)(((size_t)co * taps * Cin + 8 * q) * 2) : OOB;
    }
  }
  auto as_h8 = [](hu4_t v) {
    union { hu4_t u; h8_t h; } c;
    c.u = v;
    return c.h;
  };
  auto gload = [&](int it) {
    if (BUF) {
      const bool kin = g_c0 + 8 * q < Cin;          // Cin is a multiple of 8: whole chunk in or out
      const unsigned sa = (unsigned)(((g_ty * W + g_tx) * ld_in + g_c0) * 2);
      const unsigned sb = (unsigned)((g_tap * Cin + g_c0) * 2);
#pragma unroll
      for (int a = 0; a < 2; ++a)
        ra[a] = as_h8(__builtin_amdgcn_raw_buffer_load_b128(rsA, (kin && ((vmask[a] >> g_tap) & 1u)) ? rowoff[a] : OOB, sa, 0));
#pragma unroll
      for (int b = 0; b < NPB; ++b) rb[b] = as_h8(__builtin_amdgcn_raw_buffer_load_b128(rsB, kin ? woff[b] : OOB, sb, 0));
      g_c0 += BK;
      if (g_c0 >= Cin) {
        g_c0 = 0;
        ++g_tap;
        if (++g_tx == ks) { g_tx = 0; ++g_ty; }
      }
      return;
    }
    const int tap = it / nchunk, c0 = (it - tap * nchunk) * BK;
    const int dy = tap / ks - pad, dx = tap % ks - pad;
    const bool kin = c0 + 8 * q < Cin;              // Cin is a multiple of 8: whole chunk in or out
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int yy = py[a] + dy, xx = px[a] + dx;
      const bool ok = kin && yy >= 0 && yy < H && xx >= 0 && xx < W;
      ra[a] = ok ? *reinterpret_cast<const h8_t *>(pb[a] + ((ptrdiff_t)dy * W + dx) * ld_in + c0) : zero;
    }
#pragma unroll
    for (int b = 0; b < NPB; ++b)
      rb[b] = (wok[b] && kin) ? *reinterpret_cast<const h8_t *>(wb[b] + (size_t)tap * Cin + c0) : zero;
  };
  auto lstore = [&]() {
#pragma unroll
    for (int a = 0; a < 2; ++a) *reinterpret_cast<h8_t *>(As + (ri + 64 * a) * HB_LD + 8 * q) = ra[a];
#pragma unroll
    for (int b = 0; b < NPB; ++b) *reinterpret_cast<h8_t *>(Bs + (ri + 64 * b) * HB_LD + 8 * q) = rb[b];
  };

  gload(0);
  lstore();
  __syncthreads();
  // MFMA 32x32x16: lane l holds A[row l&31][k = 8*(l>>5) + j], B[k = 8*(l>>5) + j][col l&31], j = 0..7
  const _Float16 *ap = As + (32 * TM * wm + (lane & 31)) * HB_LD + 8 * (lane >> 5);
  const _Float16 *bp = Bs + (32 * TN * wn + (lane & 31)) * HB_LD + 8 * (lane >> 5);
  for (int it = 0; it < nit; ++it) {
    if (it + 1 < nit) gload(it + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      h8_t a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const h8_t *>(ap + i * 32 * HB_LD + 16 * kk);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const h8_t *>(bp + j * 32 * HB_LD + 16 * kk);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (it + 1 < nit) {
      lstore();
      __syncthreads();
    }
  }
  // epilogue: acc[r] = D[row = (r&3) + 8(r>>2) + 4(lane>>5)][col = lane&31]
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = n0 + 32 * (TN * wn + j) + (lane & 31);
    if (co < Cout) {
      const float bb = bias[co];
      const int sg = (co < dst.end[0]) ? 0 : ((co < dst.end[1]) ? 1 : 2);
      const int cbase = (sg == 0) ? 0 : dst.end[sg - 1];
      _Float16 *op = dst.p[sg] + dst.off[sg] + (co - cbase);
      const int ld = dst.ld[sg];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + 32 * (TM * wm + i) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (m < M) op[(size_t)m * ld] = (_Float16)fmaxf(acc[i][j][r] + bb, 0.f);
        }
    }
  }
}

// conv1 (7x7 s2, fp32 arithmetic on the VALU as in the parity path) with float16 output
constexpr int C1_PATCH = 37;
__global__ __launch_bounds__(256) void k_conv1_h(const float *__restrict__ padded, int Wp, int Wimg, long long tile0,
                                                  const float *__restrict__ w, const float *__restrict__ bias,
                                                  _Float16 *__restrict__ out) {
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
    const int iy = 2 * oy0 - 3 + py, ix = 2 * ox0 - 3 + px;
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
  h8_t *o = reinterpret_cast<h8_t *>(out + (((size_t)t * 128 + oy0 + ty) * 128 + ox0 + tx) * 64);
#pragma unroll
  for (int c8 = 0; c8 < 8; ++c8) {
    h8_t v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (_Float16)fmaxf(acc[8 * c8 + j] + bias[8 * c8 + j], 0.f);
    o[c8] = v;
  }
}

constexpr int PRH = 4;  // vertically adjacent outputs per thread (row maxima shared between overlapping windows)
__global__ void k_maxpool_h(const _Float16 *__restrict__ in, int N, int H, int W, int C, int ks, int stride, int pad,
                            _Float16 *__restrict__ out, int Ho, int Wo) {
  const int c8n = C >> 3;
  const int hob = (Ho + PRH - 1) / PRH;
  const size_t total = (size_t)N * hob * Wo * c8n;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c8 = (int)(i % c8n);
  size_t r = i / c8n;
  const int ox = (int)(r % Wo); r /= Wo;
  const int oyb = (int)(r % hob);
  const int n = (int)(r / hob);
  const int x0 = max(ox * stride - pad, 0), x1 = min(ox * stride - pad + ks, W);
  h8_t lowest;
#pragma unroll
  for (int j = 0; j < 8; ++j) lowest[j] = (_Float16)(-65504.0f);
  const int oy0 = oyb * PRH;
  const int ylo = max(oy0 * stride - pad, 0);
  const int yhi = min((min(oy0 + PRH, Ho) - 1) * stride - pad + ks, H);
  h8_t acc[PRH];
#pragma unroll
  for (int k = 0; k < PRH; ++k) acc[k] = lowest;
  for (int y = ylo; y < yhi; ++y) {
    h8_t m = lowest;
    for (int x = x0; x < x1; ++x)
      m = __builtin_elementwise_max(m, *reinterpret_cast<const h8_t *>(in + (((size_t)n * H + y) * W + x) * C + 8 * c8));
#pragma unroll
    for (int k = 0; k < PRH; ++k) {
      const int ys = (oy0 + k) * stride - pad;
      if (y >= ys && y < ys + ks) acc[k] = __builtin_elementwise_max(acc[k], m);
    }
  }
#pragma unroll
  for (int k = 0; k < PRH; ++k)
    if (oy0 + k < Ho) *reinterpret_cast<h8_t *>(out + (((size_t)n * Ho + oy0 + k) * Wo + ox) * C + 8 * c8) = acc[k];
}

__global__ __launch_bounds__(256) void k_head_h(const _Float16 *__restrict__ in, int HW, int C,
                                                 const float *__restrict__ fcw, const float *__restrict__ fcb,
                                                 const float *__restrict__ plane, long long tile0, float nodata,
                                                 float *__restrict__ out) {
  __shared__ float red[2][256];
  const int t = blockIdx.x, tid = threadIdx.x;
  float d0 = 0.f, d1 = 0.f;
  const float inv = 1.0f / (float)HW;
  for (int c = tid; c < C; c += 256) {
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += (float)in[((size_t)t * HW + p) * C + c];
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

template <int BN>
int launch_conv_h(const _Float16 *in, int M, int H, int W, int Cin, int ld_in, const _Float16 *wt, const float *bias,
                  int Cout, int ks, const ConvDstH &dst, hipStream_t st) {
  dim3 grid(sf_cdiv(M, 128), sf_cdiv(Cout, BN));
  const size_t abytes = ((size_t)M * ld_in + 2 * ((size_t)(ks >> 1) * W + (ks >> 1)) * ld_in) * 2 + 64;
  const size_t bbytes = (size_t)Cout * ks * ks * Cin * 2;
  const size_t lim = (size_t)0x7ff00000 - ((size_t)ks * W + ks) * ld_in * 2;
  if (abytes < lim && bbytes < lim && sf_tune().cnn_conv_variant == 0)
    hipLaunchKernelGGL((k_conv_igemm_f16<BN, true>), grid, dim3(256), 0, st, in, M, H, W, Cin, ld_in, wt, bias, Cout, ks, dst);
  else
    hipLaunchKernelGGL((k_conv_igemm_f16<BN>), grid, dim3(256), 0, st, in, M, H, W, Cin, ld_in, wt, bias, Cout, ks, dst);
  SF_LAUNCH_CHECK("k_conv_igemm_f16");
  return 0;
}

int conv_dispatch_h(const void *in, int N, int H, int W, int Cin, int ld_in, const void *w, const float *bias, int Cout,
                    int ksize, const ConvDstH &dst, hipStream_t st) {
  const long long Ml = (long long)N * H * W;
  if (Ml > 2000000000LL) { sf_set_error("sf_cnn_conv_f16: batch too large"); return -1; }
  const int M = (int)Ml;
  // BN = 128 only where it wastes no more columns than BN = 64 would
  const int waste64 = sf_cdiv(Cout, 64) * 64, waste128 = sf_cdiv(Cout, 128) * 128;
  if (Cout >= 128 && waste128 <= waste64)
    return launch_conv_h<128>((const _Float16 *)in, M, H, W, Cin, ld_in, (const _Float16 *)w, bias, Cout, ksize, dst, st);
  return launch_conv_h<64>((const _Float16 *)in, M, H, W, Cin, ld_in, (const _Float16 *)w, bias, Cout, ksize, dst, st);
}

}  // namespace

extern "C" {

int sf_cnn_conv1_f16(const float *padded, int Hp, int Wp, int W, long long tile0, int ntiles, const float *w,
                     const float *bias, void *out, void *stream) {
  if (!padded || !w || !bias || !out || ntiles < 1 || W < 1 || Wp != W + 255 || tile0 < 0 ||
      (tile0 + ntiles + W - 1) / W > Hp - 255) {
    sf_set_error("sf_cnn_conv1_f16: bad argument");
    return -1;
  }
  hipLaunchKernelGGL(k_conv1_h, dim3(64, ntiles), dim3(256), 0, (hipStream_t)stream, padded, Wp, W, tile0, w, bias,
                     (_Float16 *)out);
  SF_LAUNCH_CHECK("k_conv1_h");
  return 0;
}

int sf_cnn_maxpool_f16(const void *in, int N, int H, int W, int C, int ksize, int stride, int pad, void *out, int Ho,
                       int Wo, void *stream) {
  if (!in || !out || N < 1 || (C & 7) || ksize < 1 || stride < 1) {
    sf_set_error("sf_cnn_maxpool_f16: bad argument (channels must be a multiple of 8)");
    return -1;
  }
  const size_t total = (size_t)N * ((Ho + PRH - 1) / PRH) * Wo * (C / 8);
  hipLaunchKernelGGL(k_maxpool_h, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16 *)in, N, H, W, C, ksize, stride, pad, (_Float16 *)out, Ho, Wo);
  SF_LAUNCH_CHECK("k_maxpool_h");
  return 0;
}

int sf_cnn_conv_f16(const void *in, int N, int H, int W, int Cin, int ld_in, const void *w, const float *bias, int Cout,
                    int ksize, void *out, int ld_out, int ch_off, void *stream) {
  if (!in || !w || !bias || !out || N < 1 || (ksize != 1 && ksize != 3) || (Cin & 7) || (ld_in & 7) || Cin > ld_in ||
      ch_off < 0 || ch_off + Cout > ld_out) {
    sf_set_error("sf_cnn_conv_f16: bad argument (ksize 1|3, Cin and ld_in multiples of 8)");
    return -1;
  }
  ConvDstH d{};
  d.p[0] = d.p[1] = d.p[2] = (_Float16 *)out;
  d.ld[0] = d.ld[1] = d.ld[2] = ld_out;
  d.off[0] = d.off[1] = d.off[2] = ch_off;
  d.end[0] = d.end[1] = d.end[2] = Cout;
  return conv_dispatch_h(in, N, H, W, Cin, ld_in, w, bias, Cout, ksize, d, (hipStream_t)stream);
}

int sf_cnn_conv_split3_f16(const void *in, int N, int H, int W, int Cin, int ld_in, const void *w, const float *bias,
                           int c0, int c1, int c2, void *out0, int ld0, int off0, void *out1, int ld1, int off1,
                           void *out2, int ld2, int off2, void *stream) {
  if (!in || !w || !bias || !out0 || !out1 || !out2 || N < 1 || (Cin & 7) || (ld_in & 7) || Cin > ld_in || c0 < 1 ||
      c1 < 1 || c2 < 1 || off0 + c0 > ld0 || off1 + c1 > ld1 || off2 + c2 > ld2) {
    sf_set_error("sf_cnn_conv_split3_f16: bad argument");
    return -1;
  }
  ConvDstH d{};
  d.p[0] = (_Float16 *)out0; d.p[1] = (_Float16 *)out1; d.p[2] = (_Float16 *)out2;
  d.ld[0] = ld0; d.ld[1] = ld1; d.ld[2] = ld2;
  d.off[0] = off0; d.off[1] = off1; d.off[2] = off2;
  d.end[0] = c0; d.end[1] = c0 + c1; d.end[2] = c0 + c1 + c2;
  return conv_dispatch_h(in, N, H, W, Cin, ld_in, w, bias, c0 + c1 + c2, 1, d, (hipStream_t)stream);
}

int sf_cnn_head_f16(const void *in, int ntiles, int HW, int C, const float *fcw, const float *fcb, const float *plane,
                    long long tile0, float nodata, float *out, void *stream) {
  if (!in || !fcw || !fcb || !out || ntiles < 1) { sf_set_error("sf_cnn_head_f16: bad argument"); return -1; }
  hipLaunchKernelGGL(k_head_h, dim3(ntiles), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)in, HW, C, fcw, fcb,
                     plane, tile0, nodata, out);
  SF_LAUNCH_CHECK("k_head_h");
  return 0;
}

}  // extern "C"
