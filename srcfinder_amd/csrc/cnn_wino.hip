// CNN tile scorer: the 3 x 3 stride-1 pad-1 convolutions (BasicConv2d, cnn/archs/googlenet1.py:266-275; conv3 and the
// branch2 / branch3 convolutions of the nine inception blocks, :62-78, :184-228) by Winograd's minimal filtering F(2 x 2, 3 x 3).
//
// Two thirds of the scorer's 3.7 GFLOP per window are 3 x 3 convolutions, and the implicit-GEMM kernel runs them at
// 0.80-0.84 of the fp32 matrix peak: what is left to win is the arithmetic itself.  For a 2 x 2 block of outputs,
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A,     d = the 4 x 4 input patch, g = the 3 x 3 filter,
//     B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1],
// summed over the input channels BEFORE the output transform: 16 multiplications per channel pair and output block instead
// of 36, i.e. sixteen independent GEMMs  M_xi [tiles x Cout] = V_xi [tiles x Cin] U_xi [Cin x Cout]  (xi = the 16 positions of
// the transformed 4 x 4 patch) on the same v_mfma_f32_32x32x2_f32 as the direct kernel -- float32 in, float32 accumulate,
// nothing at reduced precision.  The transforms are additions (B^T, A^T) and a once-per-weight-upload product (G): the result
// differs from the direct convolution by rounding only (a few 1e-7 relative, tools/check_conv.py); the tile scorer's
// goldens hold at their 1e-4 as before.  sf_debug_set(17, 2) keeps the direct kernel for every 3 x 3 convolution.
//
// One workgroup = 64 output blocks (8 x 8 blocks = 16 x 16 pixels of one image; on the 8 x 8 stage 4 x 4 blocks of four
// images) x BN output channels, eight waves, Cin in chunks of 16:
//   1. the (2 T + 2)^2 input pixels of the region, 16 channels each, arrive in LDS raw (buffer loads issued a chunk ahead;
//      pixels outside the image read as zero), together with the chunk of U: [16 xi][BN][16 ci];
//   2. every thread transforms half a patch of one (block, channel quad): 12 float4 reads, 64 additions, 8 float4 writes into
//      V [16 xi][64 blocks][16 ci] (quads XOR-swizzled by the block index so that the MFMA lanes' 16-byte reads are conflict-free);
//   3. wave w multiplies xi = 2 w and 2 w + 1: 64 MFMAs per chunk, fragments by ds_read_b128 (four k-steps per read);
// after the last chunk the sixteen M_xi meet in LDS (32 channels at a time: 128 KB over the operand buffers), every thread
// applies A^T . A to its (block, channel) items, adds the folded-BatchNorm bias, ReLU, and stores 32 consecutive channels.
#include "cmf_common.h"
#include <type_traits>

typedef float wn_f16 __attribute__((ext_vector_type(16)));
typedef unsigned wn_u4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int WN_NT = 512, WN_BK = 16, WN_T = 64;   // threads, channels per chunk, output blocks per workgroup
constexpr int WN_RPX = 400;                          // raw pixels: 18 x 18 (one image) or 4 x 10 x 10 (four images of the 8 x 8 stage)
constexpr int WN_NRAW = (WN_RPX * 4 + WN_NT - 1) / WN_NT;
constexpr unsigned WN_OOB = 0x80000000u;

// U[xi = 4 r + c][co][ci] = sum_{ky, kx} G[r][ky] g[co][ky][kx][ci] G[c][kx]   (float64 arithmetic, rounded once)
__global__ void k_wino_weights(const float *__restrict__ w /*[Cout][9][Cin]*/, int Cout, int Cin, float *__restrict__ U) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)Cout * Cin) return;
  const int co = (int)(i / Cin), ci = (int)(i - (size_t)co * Cin);
  const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  double g[3][3];
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) g[ky][kx] = (double)w[((size_t)co * 9 + ky * 3 + kx) * Cin + ci];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      double s = 0.0;
      for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) s += G[r][ky] * g[ky][kx] * G[c][kx];
      U[((size_t)(4 * r + c) * Cout + co) * Cin + ci] = (float)s;
    }
}

__device__ __forceinline__ float4 wn_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 wn_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

template <int BN>
__global__ __launch_bounds__(WN_NT) void k_wino(const float *__restrict__ in, int N, int H, int W, int Cin, int ld_in,
                                                 const float *__restrict__ U, const float *__restrict__ bias, int Cout,
                                                 float *__restrict__ out, int ld_out, int ch_off, int TYX) {
  constexpr int NTN = BN / 32;                                    // 32-channel MFMA tiles along N
  constexpr int NU = (16 * BN * 4 + WN_NT - 1) / WN_NT;           // float4 items of a U chunk per thread
  extern __shared__ __attribute__((aligned(16))) float wn_lds[];
  float *raw = wn_lds;                                            // [WN_RPX][16]
  float *V = raw + WN_RPX * WN_BK;                                // [16][64][16]
  float *Ub = V + 16 * WN_T * WN_BK;                              // [16][BN][16]
  float *Mx = wn_lds;                                             // epilogue: [16][64][32] over everything above
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
  const int n0 = blockIdx.y * BN;
  // ---- region of this workgroup
  const int P = 2 * TYX + 2, PP = P * P, NI = WN_T / (TYX * TYX);
  int img0, oy0 = 0, ox0 = 0;
  if (NI == 1) {
    const int RX = W / (2 * TYX), RY = H / (2 * TYX);
    const int g = blockIdx.x;
    img0 = g / (RX * RY);
    const int r = g - img0 * (RX * RY);
    oy0 = (r / RX) * 2 * TYX;
    ox0 = (r % RX) * 2 * TYX;
  } else {
    img0 = blockIdx.x * NI;
  }
  const __amdgpu_buffer_rsrc_t rsA =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (unsigned)((size_t)N * H * W * ld_in * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsU =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(U), 0, (unsigned)((size_t)16 * Cout * Cin * 4), 0x00020000);
  // ---- raw items: idx = tid + 512 u = (raw pixel, channel quad)
  unsigned rawoff[WN_NRAW];
#pragma unroll
  for (int u = 0; u < WN_NRAW; ++u) {
    const int idx = tid + WN_NT * u, rpx = idx >> 2, quad = idx & 3;
    const int il = rpx / PP, rem = rpx - il * PP, py = rem / P, px = rem - py * P;
    const int n = img0 + il, iy = oy0 - 1 + py, ix = ox0 - 1 + px;
    const bool ok = rpx < NI * PP && n < N && iy >= 0 && iy < H && ix >= 0 && ix < W;
    rawoff[u] = ok ? (unsigned)((((size_t)n * H + iy) * W + ix) * ld_in + 4 * quad) * 4u : WN_OOB;
  }
  // ---- U items: idx = tid + 512 u = (xi, co, quad); 4 BN divides 512, so (co, quad) do not depend on u and xi advances by
  //      512 / (4 BN) per item: one offset and one stride instead of a table (registers)
  static_assert(WN_NT % (4 * BN) == 0 && NU * (WN_NT / (4 * BN)) == 16, "U items");
  constexpr int XSTEP = WN_NT / (4 * BN);
  const int uquad = tid & 3, uco = (tid >> 2) % BN, uxi0 = tid / (4 * BN);
  const bool uok = n0 + uco < Cout;
  const unsigned uoff0 = (unsigned)((((size_t)uxi0 * Cout + n0 + uco) * Cin + 4 * uquad) * 4);
  const unsigned ustride = (unsigned)((size_t)XSTEP * Cout * Cin * 4);
  const int udst0 = (uxi0 * BN + uco) * WN_BK + 4 * (uquad ^ ((uco >> 2) & 3));
  // ---- transform item: block t, channel quad q, half hx (rows 2 hx, 2 hx + 1 of the transformed patch)
  const int hx = tid & 1, tq = (tid >> 1) & 3, tt = tid >> 3;
  int tbase;   // raw pixel index of the patch's top-left corner
  {
    const int tpi = TYX * TYX, il = tt / tpi, r = tt - il * tpi, ty = r / TYX, tx = r - ty * TYX;
    tbase = il * PP + (2 * ty) * P + 2 * tx;
  }
  const int vdst = (tt * WN_BK + 4 * (tq ^ ((tt >> 2) & 3)));     // + xi * 64 * 16

  wn_f16 acc[2][2][NTN];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < NTN; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][c][r] = 0.f;

  wn_u4 rr[WN_NRAW], ru[NU];
  // (the two halves of a chunk's prefetch are issued apart: the raw pixels before the input transform, the weights behind it --
  //  requested together they are 48 registers held across the transform's 80, beside 128 accumulators)
  auto gload_raw = [&](int c0) {
    const unsigned sb = (unsigned)(c0 * 4);
#pragma unroll
    for (int u = 0; u < WN_NRAW; ++u) rr[u] = __builtin_amdgcn_raw_buffer_load_b128(rsA, rawoff[u], sb, 0);
  };
  auto gload_u = [&](int c0) {
    const unsigned sb = (unsigned)(c0 * 4);
#pragma unroll
    for (int u = 0; u < NU; ++u) ru[u] = __builtin_amdgcn_raw_buffer_load_b128(rsU, uok ? uoff0 + u * ustride : WN_OOB, sb, 0);
  };
  const int swz = (l31 >> 2) & 3;
  gload_raw(0);
  gload_u(0);
  for (int c0 = 0; c0 < Cin; c0 += WN_BK) {
    // 1. stage the chunk (the previous chunk's MFMAs are done: the barrier that ended the last iteration)
#pragma unroll
    for (int u = 0; u < WN_NRAW; ++u)
      if (tid + WN_NT * u < WN_RPX * 4) *reinterpret_cast<wn_u4 *>(raw + 4 * (tid + WN_NT * u)) = rr[u];
#pragma unroll
    for (int u = 0; u < NU; ++u) *reinterpret_cast<wn_u4 *>(Ub + udst0 + u * (XSTEP * BN * WN_BK)) = ru[u];
    __syncthreads();
    if (c0 + WN_BK < Cin) gload_raw(c0 + WN_BK);
    // 2. input transform: T = B^T d (rows 2 hx, 2 hx + 1), V = T B
    {
      float4 T[2][4];
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        float4 d[3];   // patch rows hx .. hx + 2 of column px
#pragma unroll
        for (int py = 0; py < 3; ++py) d[py] = *reinterpret_cast<const float4 *>(raw + (tbase + (hx + py) * P + px) * WN_BK + 4 * tq);
        // hx = 0: rows 0, 1 of B^T d = d0 - d2, d1 + d2 (d = patch rows 0..2);  hx = 1: rows 2, 3 = d2 - d1, d1 - d3 (patch rows 1..3)
        T[0][px] = hx ? wn_sub(d[1], d[0]) : wn_sub(d[0], d[2]);
        T[1][px] = hx ? wn_sub(d[0], d[2]) : wn_add(d[1], d[2]);
      }
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const float4 v0 = wn_sub(T[r][0], T[r][2]), v1 = wn_add(T[r][1], T[r][2]), v2 = wn_sub(T[r][2], T[r][1]),
                     v3 = wn_sub(T[r][1], T[r][3]);
        float *vp = V + (size_t)(4 * (2 * hx + r)) * (WN_T * WN_BK) + vdst;
        *reinterpret_cast<float4 *>(vp) = v0;
        *reinterpret_cast<float4 *>(vp + WN_T * WN_BK) = v1;
        *reinterpret_cast<float4 *>(vp + 2 * WN_T * WN_BK) = v2;
        *reinterpret_cast<float4 *>(vp + 3 * WN_T * WN_BK) = v3;
      }
    }
    __syncthreads();
    if (c0 + WN_BK < Cin) gload_u(c0 + WN_BK);
    // 3. the wave's two GEMMs: M_xi += V_xi U_xi
#pragma unroll
    for (int xl = 0; xl < 2; ++xl) {
      const int xi = 2 * wave + xl;
      const float *vx = V + (size_t)xi * (WN_T * WN_BK) + l31 * WN_BK, *ux = Ub + (size_t)xi * (BN * WN_BK) + l31 * WN_BK;
#pragma unroll
      for (int i = 0; i < WN_BK / 8; ++i) {
        const int qo = 4 * ((2 * i + kh) ^ swz);
        float4 a4[2], b4[NTN];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) a4[mb] = *reinterpret_cast<const float4 *>(vx + mb * 32 * WN_BK + qo);
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) b4[nt] = *reinterpret_cast<const float4 *>(ux + nt * 32 * WN_BK + qo);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) {
              const float av = j == 0 ? a4[mb].x : (j == 1 ? a4[mb].y : (j == 2 ? a4[mb].z : a4[mb].w));
              const float bv = j == 0 ? b4[nt].x : (j == 1 ? b4[nt].y : (j == 2 ? b4[nt].z : b4[nt].w));
              acc[xl][mb][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[xl][mb][nt], 0, 0, 0);
            }
      }
    }
    __syncthreads();   // raw, V and U are rewritten by the next chunk
  }
  // ---- epilogue: the sixteen M_xi of 32 channels meet in LDS, output transform, bias, ReLU
  const int och = tid & 31, otq = tid >> 5;
#pragma unroll
  for (int nt = 0; nt < NTN; ++nt) {
#pragma unroll
    for (int xl = 0; xl < 2; ++xl)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * kh;
          Mx[((size_t)(2 * wave + xl) * WN_T + row) * 32 + l31] = acc[xl][mb][nt][r];
        }
    __syncthreads();
    const int co = n0 + 32 * nt + och;
    const float bb = (co < Cout) ? bias[co] : 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int t = otq + 16 * it;
      float m[16];
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) m[xi] = Mx[((size_t)xi * WN_T + t) * 32 + och];
      float tm[2][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        tm[0][c] = (m[c] + m[4 + c]) + m[8 + c];
        tm[1][c] = (m[4 + c] - m[8 + c]) - m[12 + c];
      }
      const int tpi = TYX * TYX, il = t / tpi, rr2 = t - il * tpi, ty = rr2 / TYX, tx = rr2 - ty * TYX;
      const int n = img0 + il;
      if (co < Cout && n < N) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float y0 = (tm[i][0] + tm[i][1]) + tm[i][2], y1 = (tm[i][1] - tm[i][2]) - tm[i][3];
          const int oy = oy0 + 2 * ty + i, ox = ox0 + 2 * tx;
          float *op = out + (((size_t)n * H + oy) * W + ox) * ld_out + ch_off + co;
          op[0] = fmaxf(y0 + bb, 0.f);
          op[ld_out] = fmaxf(y1 + bb, 0.f);
        }
      }
    }
    __syncthreads();
  }
}

template <int BN>
size_t wino_lds() { return ((size_t)WN_RPX * WN_BK + 16 * WN_T * WN_BK + 16 * BN * WN_BK) * sizeof(float); }

}  // namespace

extern "C" {

int sf_cnn_wino_ok(int H, int W, int Cin) { return (H == W) && (W == 8 || (W >= 16 && W % 16 == 0)) && Cin >= 16 && Cin % 16 == 0; }

size_t sf_cnn_wino_weight_floats(int Cout, int Cin) { return (size_t)16 * Cout * Cin; }

int sf_cnn_wino_weights(const float *w, int Cout, int Cin, float *U, void *stream) {
  if (!w || !U || Cout < 1 || Cin < 1) { sf_set_error("sf_cnn_wino_weights: bad argument"); return -1; }
  const size_t n = (size_t)Cout * Cin;
  hipLaunchKernelGGL(k_wino_weights, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, U);
  SF_LAUNCH_CHECK("k_wino_weights");
  return 0;
}

int sf_cnn_conv3x3_wino(const float *in, int N, int H, int W, int Cin, int ld_in, const float *U, const float *bias, int Cout,
                        float *out, int ld_out, int ch_off, void *stream) {
  if (!in || !U || !bias || !out || N < 1 || Cout < 1 || (ld_in & 3) || Cin > ld_in || ch_off < 0 || ch_off + Cout > ld_out ||
      !sf_cnn_wino_ok(H, W, Cin)) {
    sf_set_error("sf_cnn_conv3x3_wino: bad argument (square images of 8 or a multiple of 16 pixels, Cin a multiple of 16)");
    return -1;
  }
  if ((size_t)N * H * W * ld_in * 4 >= 0x7ff00000u || (size_t)16 * Cout * Cin * 4 >= 0x7ff00000u) {
    sf_set_error("sf_cnn_conv3x3_wino: operand of 2 GB or more (use sf_cnn_conv)");
    return -2;
  }
  const int TYX = (W >= 16) ? 8 : 4, NI = 64 / (TYX * TYX);
  const int groups = (NI == 1) ? N * (H / 16) * (W / 16) : sf_cdiv(N, NI);
  hipStream_t st = (hipStream_t)stream;
  if (Cout > 32) {
    const size_t lds = wino_lds<64>();
    if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_wino<64>), lds)) return rc;
    hipLaunchKernelGGL(k_wino<64>, dim3(groups, sf_cdiv(Cout, 64)), dim3(WN_NT), lds, st, in, N, H, W, Cin, ld_in, U, bias, Cout, out,
                       ld_out, ch_off, TYX);
  } else {
    const size_t lds = wino_lds<32>() > (size_t)16 * WN_T * 32 * 4 ? wino_lds<32>() : (size_t)16 * WN_T * 32 * 4;
    if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_wino<32>), lds)) return rc;
    hipLaunchKernelGGL(k_wino<32>, dim3(groups, sf_cdiv(Cout, 32)), dim3(WN_NT), lds, st, in, N, H, W, Cin, ld_in, U, bias, Cout, out,
                       ld_out, ch_off, TYX);
  }
  SF_LAUNCH_CHECK("k_wino");
  return 0;
}

}  // extern "C"
