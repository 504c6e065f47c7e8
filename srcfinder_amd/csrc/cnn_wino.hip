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
// images) x BN output channels, eight waves, Cin in chunks of 8 (Cin % 8 == 0: every 3 x 3 convolution of the net but conv1):
//   1. the (2 T + 2)^2 input pixels of the region, 8 channels each, arrive in LDS raw (buffer loads issued two chunks ahead;
//      pixels outside the image read as zero), together with the chunk of U: [16 xi][BN][8 ci];
//   2. every thread transforms one row of the patch of one (block, channel quad): 8 float4 reads, 32 additions, 4 float4 writes
//      into V [16 xi][64 blocks][8 ci] (quads XOR-swizzled by the block index: the MFMA lanes' 16-byte reads are conflict-free);
//   3. wave w multiplies xi = 2 w and 2 w + 1: 32 MFMAs per chunk, fragments by ds_read_b128 (two k-steps per read);
// after the last chunk the sixteen M_xi meet in LDS (32 channels at a time: 128 KB over the operand buffers), every thread
// applies A^T . A to its (block, channel) items, adds the folded-BatchNorm bias, ReLU, and stores 32 consecutive channels.
// Measured (profiles/r05_cnn_layers.txt, tools/check_conv.py --time): 1.3-1.6 x the direct kernel on the layers with >= 64
// input channels, i.e. 0.47-0.55 of the matrix peak on the Winograd flop count -- the instruction stream AROUND the MFMAs
// (staging 0.34 ms, transform 0.62 ms, stores 0.39 ms, fragment reads and barriers 1.4 ms of conv3's 2.7 ms, timed with the
// EXP arms below) is the bound, not the matrix pipe (1.31 ms).
#include "cmf_common.h"
#include <type_traits>

typedef float wn_f16 __attribute__((ext_vector_type(16)));
typedef unsigned wn_u4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int WN_NT = 512, WN_T = 64;   // threads, output blocks per workgroup
constexpr int WN_RPX = 400;                          // raw pixels: 18 x 18 (one image) or 4 x 10 x 10 (four images of the 8 x 8 stage)
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

// Chunk pipeline (second form of the round; the first staged, transformed and multiplied a 16-channel chunk in three phases
// between three barriers and measured the same within 1 %).  Chunks are 8 channels, every operand buffer exists twice (2 x 77 KB), and ONE barrier separates the chunks:
//   iteration c:  barrier | stage U(c + 1) and the raw pixels of chunk c + 2 from registers, request chunk c + 3's pixels and
//                 U(c + 2) | transform chunk c + 1 (raw -> V) BETWEEN the MFMAs of chunk c (the matrix instructions are
//                 asynchronous: the transform's 8 reads, 32 additions and 4 writes per thread ride in their shadow).
// EXP (timing experiments, wrong results, -DSF_CONV_EXPERIMENTS + sf_debug_set(17, 10 + bits)): 1 no staging / loads after the
// prologue, 2 no input transform, 4 no MFMAs, 8 no epilogue stores
template <int BN, int EXP = 0>
__global__ __launch_bounds__(WN_NT) void k_wino(const float *__restrict__ in, int N, int H, int W, int Cin, int ld_in,
                                                 const float *__restrict__ U, const float *__restrict__ bias, int Cout,
                                                 float *__restrict__ out, int ld_out, int ch_off, int TYX) {
  constexpr int NTN = BN / 32;                                    // 32-channel MFMA tiles along N
  constexpr int BK = 8;                                           // channels per chunk: two quads
  constexpr int NRAW = (WN_RPX * 2 + WN_NT - 1) / WN_NT;          // float4 items of a raw chunk per thread (2)
  constexpr int NU = 16 * BN * 2 / WN_NT;                         // float4 items of a U chunk per thread (4 / 2)
  constexpr int RAWF = WN_RPX * BK, VF = 16 * WN_T * BK, UF = 16 * BN * BK, BUFF = RAWF + VF + UF;   // floats per buffer set
  static_assert(16 * BN * 2 % WN_NT == 0, "U items");
  extern __shared__ __attribute__((aligned(16))) float wn_lds[];  // [2][raw | V | U]; the epilogue's M [16][64][32] over it
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  // ---- raw items: idx = tid + 512 u = (raw pixel, channel quad of the chunk); LDS [pixel][8 floats], linear
  unsigned rawoff[NRAW];
#pragma unroll
  for (int u = 0; u < NRAW; ++u) {
    const int idx = tid + WN_NT * u, rpx = idx >> 1, quad = idx & 1;
    const int il = rpx / PP, rem = rpx - il * PP, py = rem / P, px = rem - py * P;
    const int n = img0 + il, iy = oy0 - 1 + py, ix = ox0 - 1 + px;
    const bool ok = rpx < NI * PP && n < N && iy >= 0 && iy < H && ix >= 0 && ix < W;
    rawoff[u] = ok ? (unsigned)((((size_t)n * H + iy) * W + ix) * ld_in + 4 * quad) * 4u : WN_OOB;
  }
  // ---- U items: idx = tid + 512 u = (xi, co, quad); (co, quad) do not depend on u, xi advances by 512 / (2 BN) per item
  constexpr int XSTEP = WN_NT / (2 * BN);
  const int uquad = tid & 1, uco = (tid >> 1) % BN, uxi0 = tid / (2 * BN);
  const bool uok = n0 + uco < Cout;
  const unsigned uoff0 = (unsigned)((((size_t)uxi0 * Cout + n0 + uco) * Cin + 4 * uquad) * 4);
  const unsigned ustride = (unsigned)((size_t)XSTEP * Cout * Cin * 4);
  // rows of 8 floats: the quad of row r sits in slot quad ^ ((r >> 3) & 1) (conflict-free 16-byte reads by the MFMA lanes)
  const int udst0 = (uxi0 * BN + uco) * BK + 4 * (uquad ^ ((uco >> 3) & 1));
  // ---- transform item: block tt, channel quad tq, row tr of the transformed patch
  const int tr = tid & 3, tq = (tid >> 2) & 1, tt = tid >> 3;
  int tbase;   // raw pixel index of the patch's top-left corner
  {
    const int tpi = TYX * TYX, il = tt / tpi, r = tt - il * tpi, ty = r / TYX, tx = r - ty * TYX;
    tbase = il * PP + (2 * ty) * P + 2 * tx;
  }
  // row tr of B^T d = (tr == 0: d0 - d2; 1: d1 + d2; 2: d2 - d1; 3: d1 - d3): patch rows pa, pb and the sign of the second
  const int pa = (tr == 0) ? 0 : ((tr == 2) ? 2 : 1), pb = (tr == 3) ? 3 : ((tr == 2) ? 1 : 2);
  const float sgn = (tr == 1) ? 1.f : -1.f;
  const int rsrc = (tbase + pa * P) * BK + 4 * tq, rsrd = (tbase + pb * P) * BK + 4 * tq;    // + px * BK
  const int vdst = (4 * tr * WN_T + tt) * BK + 4 * (tq ^ ((tt >> 3) & 1));                     // + c * 64 * BK

  // wave w multiplies xi = 2 w and 2 w + 1 over the whole 64-block x BN-channel tile (v_mfma_f32_32x32x2_f32).  (A form with all
  // sixteen xi of a 16-block x 32-channel piece in one wave -- v_mfma_f32_16x16x4_f32, output transform in registers, no LDS
  // exchange behind the last chunk -- was built and measured 6 % slower: twice the fragment reads per MFMA cycle.)
  wn_f16 acc[2][2][NTN];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < NTN; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][c][r] = 0.f;
  const int l31 = lane & 31, kh = lane >> 5;

  const int nc = Cin / BK;
  wn_u4 rr[NRAW], ru[NU];
  auto gload_raw = [&](int c) {     // chunk c's pixels -> registers (a chunk past the end: nothing)
    const unsigned sb = (unsigned)(c * BK * 4);
#pragma unroll
    for (int u = 0; u < NRAW; ++u) rr[u] = __builtin_amdgcn_raw_buffer_load_b128(rsA, c < nc ? rawoff[u] : WN_OOB, sb, 0);
  };
  auto gload_u = [&](int c) {
    const unsigned sb = (unsigned)(c * BK * 4);
#pragma unroll
    for (int u = 0; u < NU; ++u) ru[u] = __builtin_amdgcn_raw_buffer_load_b128(rsU, (uok && c < nc) ? uoff0 + u * ustride : WN_OOB, sb, 0);
  };
  auto stage_raw = [&](int buf) {
    float *raw = wn_lds + buf * BUFF;
#pragma unroll
    for (int u = 0; u < NRAW; ++u)
      if (tid + WN_NT * u < WN_RPX * 2) *reinterpret_cast<wn_u4 *>(raw + 4 * (tid + WN_NT * u)) = rr[u];
  };
  auto stage_u = [&](int buf) {
    float *Ub = wn_lds + buf * BUFF + RAWF + VF;
#pragma unroll
    for (int u = 0; u < NU; ++u) *reinterpret_cast<wn_u4 *>(Ub + udst0 + u * (XSTEP * BN * BK)) = ru[u];
  };
  // input transform of one chunk, in three pieces the multiply loop spreads between its MFMAs
  float4 td[2][4], tv[4];
  auto tf_read = [&](int buf) {
    const float *raw = wn_lds + buf * BUFF;
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      td[0][px] = *reinterpret_cast<const float4 *>(raw + rsrc + px * BK);
      td[1][px] = *reinterpret_cast<const float4 *>(raw + rsrd + px * BK);
    }
  };
  auto tf_math = [&]() {
    float4 T[4];
#pragma unroll
    for (int px = 0; px < 4; ++px)
      T[px] = make_float4(__builtin_fmaf(sgn, td[1][px].x, td[0][px].x), __builtin_fmaf(sgn, td[1][px].y, td[0][px].y),
                          __builtin_fmaf(sgn, td[1][px].z, td[0][px].z), __builtin_fmaf(sgn, td[1][px].w, td[0][px].w));
    tv[0] = wn_sub(T[0], T[2]);
    tv[1] = wn_add(T[1], T[2]);
    tv[2] = wn_sub(T[2], T[1]);
    tv[3] = wn_sub(T[1], T[3]);
  };
  auto tf_write = [&](int buf) {
    float *V = wn_lds + buf * BUFF + RAWF;
#pragma unroll
    for (int c = 0; c < 4; ++c) *reinterpret_cast<float4 *>(V + vdst + c * (WN_T * BK)) = tv[c];
  };

  // ---- prologue: chunk 0 staged and transformed, chunk 1's pixels staged, chunk 2's pixels and U(1) in registers
  // (requesting chunk 1's loads before waiting for chunk 0's -- one memory round trip instead of three -- costs 16 more live
  //  registers and measured 7 % SLOWER)
  gload_raw(0);
  gload_u(0);
  stage_raw(0);
  stage_u(0);
  gload_raw(1);
  __syncthreads();
  tf_read(0);
  tf_math();
  tf_write(0);
  stage_raw(1);
  gload_raw(2);
  gload_u(1);
  const int swz = (l31 >> 3) & 1;
  const int aro = l31 * BK + 4 * (kh ^ swz);      // this lane's quad (channels 4 kh .. 4 kh + 3) of row l31 (+ 32 rows per MFMA block)
  for (int c = 0; c < nc; ++c) {
    const int cur = c & 1, nxt = cur ^ 1;
    __syncthreads();   // V(c), U(c) and the raw pixels of chunk c + 1 are complete; everybody is past chunk c - 1
    // stage what the registers hold: U(c + 1) (its buffer was read by the MFMAs of chunk c - 1), the pixels of chunk c + 2 (their
    // buffer was read by the transform of chunk c in the last iteration); then request the next ones
    if (!(EXP & 1)) {
      stage_u(nxt);
      stage_raw(cur);
      gload_raw(c + 3);
      gload_u(c + 2);
    }
    const bool more = (EXP & 2) ? false : c + 1 < nc;
    const float *V = wn_lds + cur * BUFF + RAWF, *Ub = V + VF;
    float4 a4[2][2], b4[2][NTN];
#pragma unroll
    for (int xl = 0; xl < 2; ++xl) {
      const int xi = 2 * wave + xl;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) a4[xl][mb] = *reinterpret_cast<const float4 *>(V + (xi * WN_T + 32 * mb) * BK + aro);
#pragma unroll
      for (int nt = 0; nt < NTN; ++nt) b4[xl][nt] = *reinterpret_cast<const float4 *>(Ub + (xi * BN + 32 * nt) * BK + aro);
    }
    if (more) tf_read(nxt);
#pragma unroll
    for (int xl = 0; xl < 2; ++xl) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (!(EXP & 4)) {
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) {
              const float av = j == 0 ? a4[xl][mb].x : (j == 1 ? a4[xl][mb].y : (j == 2 ? a4[xl][mb].z : a4[xl][mb].w));
              const float bv = j == 0 ? b4[xl][nt].x : (j == 1 ? b4[xl][nt].y : (j == 2 ? b4[xl][nt].z : b4[xl][nt].w));
              acc[xl][mb][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[xl][mb][nt], 0, 0, 0);
            }
        } else if (j == 0) {
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) acc[xl][mb][nt][0] += a4[xl][mb].x * b4[xl][nt].x + a4[xl][mb].w * b4[xl][nt].w;
        }
        if (xl == 0 && j == 1) {                 // the transform of the next chunk rides between the MFMA groups
          __builtin_amdgcn_sched_barrier(0);
          if (more) tf_math();
          __builtin_amdgcn_sched_barrier(0);
        }
        if (xl == 1 && j == 0) {
          __builtin_amdgcn_sched_barrier(0);
          if (more) tf_write(nxt);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  __syncthreads();   // the last chunk's fragments are read: M may overwrite the operand buffers
  // ---- epilogue: the sixteen M_xi of 32 channels meet in LDS, output transform, bias, ReLU
  float *Mx = wn_lds;
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
          if (!(EXP & 8) || y0 == 1.2345f) {
            op[0] = fmaxf(y0 + bb, 0.f);
            op[ld_out] = fmaxf(y1 + bb, 0.f);
          }
        }
      }
    }
    __syncthreads();
  }
}

template <int BN>
size_t wino_lds() {   // two sets of (raw | V | U) at 8 channels a chunk; at least the epilogue's [16][64][32] floats
  const size_t a = (size_t)2 * (WN_RPX * 8 + 16 * WN_T * 8 + 16 * BN * 8) * sizeof(float), b = (size_t)16 * WN_T * 32 * sizeof(float);
  return a > b ? a : b;
}

}  // namespace

extern "C" {

int sf_cnn_wino_ok(int H, int W, int Cin) { return (H == W) && (W == 8 || (W >= 16 && W % 16 == 0)) && Cin >= 16 && Cin % 8 == 0; }

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
    sf_set_error("sf_cnn_conv3x3_wino: bad argument (square images of 8 or a multiple of 16 pixels, Cin a multiple of 8, at least 16)");
    return -1;
  }
  if ((size_t)16 * Cout * Cin * 4 >= 0x7ff00000u) { sf_set_error("sf_cnn_conv3x3_wino: weights of 2 GB or more (use sf_cnn_conv)"); return -2; }
  const size_t img = (size_t)H * W * ld_in * 4;
  if ((size_t)N * img >= 0x7ff00000u) {   // the buffer descriptors address < 2 GB: images are independent, run them in pieces
    int per = (int)((0x7ff00000u - 1) / img) & ~3;
    if (per < 4) { sf_set_error("sf_cnn_conv3x3_wino: one image of 512 MB or more (use sf_cnn_conv)"); return -2; }
    for (int n0 = 0; n0 < N; n0 += per) {
      const int nn = (N - n0 < per) ? N - n0 : per;
      if (int rc = sf_cnn_conv3x3_wino(in + (size_t)n0 * H * W * ld_in, nn, H, W, Cin, ld_in, U, bias, Cout,
                                       out + (size_t)n0 * H * W * ld_out, ld_out, ch_off, stream))
        return rc;
    }
    return 0;
  }
  const int TYX = (W >= 16) ? 8 : 4, NI = 64 / (TYX * TYX);
  const int groups = (NI == 1) ? N * (H / 16) * (W / 16) : sf_cdiv(N, NI);
  hipStream_t st = (hipStream_t)stream;
#ifdef SF_CONV_EXPERIMENTS
#define WN_EXP(E)                                                                                                                \
  if (sf_tune().cnn_conv_variant == 10 + E) {                                                                                    \
    const size_t lds = wino_lds<64>();                                                                                           \
    if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_wino<64, E>), lds)) return rc;                                     \
    hipLaunchKernelGGL((k_wino<64, E>), dim3(groups, sf_cdiv(Cout, 64)), dim3(WN_NT), lds, st, in, N, H, W, Cin, ld_in, U, bias, \
                       Cout, out, ld_out, ch_off, TYX);                                                                          \
    SF_LAUNCH_CHECK("k_wino");                                                                                                   \
    return 0;                                                                                                                    \
  }
  WN_EXP(1) WN_EXP(2) WN_EXP(3) WN_EXP(4) WN_EXP(7) WN_EXP(8) WN_EXP(15)
#endif
  if (Cout > 32) {
    const size_t lds = wino_lds<64>();
    if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_wino<64>), lds)) return rc;
    hipLaunchKernelGGL(k_wino<64>, dim3(groups, sf_cdiv(Cout, 64)), dim3(WN_NT), lds, st, in, N, H, W, Cin, ld_in, U, bias, Cout, out,
                       ld_out, ch_off, TYX);
  } else {
    const size_t lds = wino_lds<32>();
    if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_wino<32>), lds)) return rc;
    hipLaunchKernelGGL(k_wino<32>, dim3(groups, sf_cdiv(Cout, 32)), dim3(WN_NT), lds, st, in, N, H, W, Cin, ld_in, U, bias, Cout, out,
                       ld_out, ch_off, TYX);
  }
  SF_LAUNCH_CHECK("k_wino");
  return 0;
}

}  // extern "C"
