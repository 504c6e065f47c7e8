// C-side driver of the tile scorer: the whole eval graph of cnn/archs/googlenet1.py for a range of image rows in ONE call
// (sf_cnn_score_rows).  It sequences the public kernels of cnn_kernels.hip -- the same launches, in the
// same order, as srcfinder_amd/cnn.py::GoogLeNetHIP -- so a C caller needs no Python to score a flightline:
//   cnn_pred_pipeline.py:159-189   for each batch of tiles: model(tile) -> softmax[:, 1]; NODATA where the plane is NODATA
// Weights: ONE float32 blob of BatchNorm-folded convolutions in the order of sf_cnn_blob_layout() (conv weights
// [Cout][k*k][Cin], then the bias [Cout]); GoogLeNetHIP.packed_blob() builds it from a state_dict.
#include "cmf_common.h"
#include "cnn_ring.h"
#include "cnn_internal.h"

namespace {

struct Incep { int cin, c1, c3r, c3, c5r, c5, pp; };
// googlenet1.py:66-78
constexpr Incep INC[9] = {{192, 64, 96, 128, 16, 32, 32},   {256, 128, 128, 192, 32, 96, 64}, {480, 192, 96, 208, 16, 48, 64},
                          {512, 160, 112, 224, 24, 64, 64}, {512, 128, 128, 256, 24, 64, 64}, {512, 112, 144, 288, 32, 64, 64},
                          {528, 256, 160, 320, 32, 128, 128}, {832, 256, 160, 320, 32, 128, 128}, {832, 384, 192, 384, 48, 128, 128}};

struct Layer { size_t w, b; };           // float offsets into the blob
struct Blob {
  Layer conv1, conv2, conv3, head3[9], b2[9], b3[9], b4[9], fc;
  size_t total;
};
constexpr size_t conv_floats(int cout, int taps, int cin) { return (size_t)cout * taps * cin; }
Blob blob_layout() {
  Blob L{};
  size_t o = 0;
  auto take = [&](Layer &l, int cout, int taps, int cin) { l.w = o; o += conv_floats(cout, taps, cin); l.b = o; o += cout; };
  take(L.conv1, 64, 49, 1);
  take(L.conv2, 64, 1, 64);
  take(L.conv3, 192, 9, 64);
  for (int i = 0; i < 9; ++i) {
    const Incep &s = INC[i];
    take(L.head3[i], s.c1 + s.c3r + s.c5r, 1, s.cin);   // branch1 | branch2.0 | branch3.0 stacked (one GEMM)
    take(L.b2[i], s.c3, 9, s.c3r);
    take(L.b3[i], s.c5, 9, s.c5r);                       // 3x3 (googlenet1.py:207-209)
    take(L.b4[i], s.pp, 1, s.cin);
  }
  take(L.fc, 2, 1, 1024);
  L.total = o;
  return L;
}
int pool_out(int n, int k, int s, int p) {   // ceil_mode with PyTorch's last-window rule
  int o = (n + 2 * p - k + s - 1) / s + 1;
  if ((o - 1) * s >= n + p) --o;
  return o;
}
// Winograd-domain weights of the 3 x 3 layers (cnn_wino.hip: U = G g G^T, 16 / 9 of the folded weights), kept in the caller's
// workspace behind the activations and recomputed by every sf_cnn_score_rows call (19 tiny launches)
struct Wino { size_t conv3, b2[9], b3[9], total; };
Wino wino_layout() {
  Wino w{};
  size_t o = 0;
  auto take = [&](size_t &slot, int cout, int cin) { slot = o; o += (cin % 8 == 0) ? (size_t)16 * cout * cin : 0; };
  take(w.conv3, 192, 64);
  for (int i = 0; i < 9; ++i) { take(w.b2[i], INC[i].c3, INC[i].c3r); take(w.b3[i], INC[i].c5, INC[i].c5r); }
  w.total = o;
  return w;
}
// Split-operand weights (cnn_split.hip: fp16 hi | lo halves of the folded weights scaled per output channel, + the scales) of every
// convolution but conv1, kept behind the Winograd weights and recomputed by every call (38 tiny launches)
struct SplitL { size_t h, s; };          // offsets: halves (hi; lo follows at + cout * taps * cin), scale floats
struct Splits { SplitL conv2, conv3, head3[9], b2[9], b3[9], b4[9]; size_t halves, scales; };
Splits split_layout() {
  Splits S{};
  size_t oh = 0, os = 0;
  auto take = [&](SplitL &l, int cout, int taps, int cin) { l.h = oh; oh += 2 * conv_floats(cout, taps, cin); l.s = os; os += cout; };
  take(S.conv2, 64, 1, 64);
  take(S.conv3, 192, 9, 64);
  for (int i = 0; i < 9; ++i) {
    const Incep &s = INC[i];
    take(S.head3[i], s.c1 + s.c3r + s.c5r, 1, s.cin);
    take(S.b2[i], s.c3, 9, s.c3r);
    take(S.b3[i], s.c5, 9, s.c5r);
    take(S.b4[i], s.pp, 1, s.cin);
  }
  S.halves = (oh + 7) / 8 * 8;
  S.scales = (os + 3) / 4 * 4;      // (the halves behind the scales start on a 16-byte boundary)
  return S;
}
// activation buffers of a batch of n tiles (floats): the largest of each role over the graph
struct Acts { size_t pool1, conv2, conv3, x, y, t2, t3, pooled, total; };
Acts acts(size_t n) {
  Acts a{};
  a.pool1 = n * 64 * 64 * 64;
  a.conv2 = n * 64 * 64 * 64;
  a.conv3 = n * 64 * 64 * 192;
  a.x = n * 32 * 32 * 480;        // block input / output ping-pong: the largest concat (3b: 32 x 32 x 480)
  a.y = n * 32 * 32 * 480;
  a.t2 = n * 32 * 32 * 128;       // 3x3 reduce outputs (3b: 128 channels at 32 x 32)
  a.t3 = n * 32 * 32 * 32;
  a.pooled = n * 32 * 32 * 256;   // branch-4 pool of the block input (3b: 256 channels at 32 x 32)
  a.total = a.pool1 + a.conv2 + a.conv3 + a.x + a.y + a.t2 + a.t3 + a.pooled;
  return a;
}

// per-layer activation scales of the split-operand route (cnn_split.hip): one power of two per tensor a split convolution reads
//   [0] maxpool1's output (conv2's input)   [1] conv2's output (conv3's)   [2 + 3 i + {0, 1, 2}] inception block i: its input (the
//   three stacked 1 x 1 and the pool-projection), the 3 x 3 reducer's output, the "5 x 5" reducer's output
constexpr int NSCALE = 2 + 3 * 9;
constexpr int NFLAG = 1024;              // overflow slots: one per batch, read back every NFLAG batches
// behind the weights' split forms: the overflow slots of the batches in flight and the calibration maxima
size_t tail_bytes() { return (size_t)NFLAG * sizeof(int) + 64 * sizeof(float); }
size_t base_bytes(int batch);            // the workspace without the sharing buffers (below)

// Trunk sharing (cnn_share.hip): the phase maps over a strip of image rows and a batch's ring tensors.  depth 1: conv1 .. conv3 on
// 16 phase maps of the 64 x 64 grid; depth 2: also maxpool2, inception3a, inception3b on 64 phase maps of the 32 x 32 grid (frames
// (1, 2) -> (2, 3) -> (3, 4), cnn_ring.h) with maxpool3 assembling inception4a's input.  Every tensor a gather convolution reads has its
// ring tensor RIGHT BEHIND its maps (one < 2 GB buffer descriptor).
constexpr int SHARE_ROWS1 = 2048, SHARE_ROWS2 = 512;   // image rows a set of maps serves (a batch may reach `rows_batch` rows further)
constexpr int STRIP_ROWS = 16;                         // image rows a set of strip maps serves (band sharing): one build per ~19 batches of 512 at 598 columns
constexpr int F32_PL = 1, F32_PH = 2, F3A_L = 2, F3A_H = 3, F3B_L = 3, F3B_H = 4;      // frames on the 32 x 32 grid
struct MapT { size_t map, ring; int C; };    // float offsets of a tensor's maps and of its ring tensor (ring - map = the maps' size)
struct Share {
  int depth, rows, Hq, Wq, Hc, Wc, H8, W8;   // rows covered from the strip's first row; map / canvas geometry (H8 x W8: the 64-phase maps)
  size_t canvas, c1, p1, p1ring, pooled, pring;
  MapT q2, q3, x3a, t2a, t3a, y3a, t2b, t3b, y3b;
  // band sharing (depth 2; cnn_ring.h): the strip maps of the image rows a batch touches -- every layer run on the top and the bottom
  // 64 pixel rows of those rows' windows (per column phase: 8 nrows images of 16 rows at the 64 x 64 grid, 16 nrows of 8 rows at 32 x 32)
  int band, nrows;
  size_t s_canvas, s_c1, s_p1, s_q2, s_q3, s_x3, s_t2[2], s_t3[2], s_y[2], s_pool;
  size_t total;
};
Share share_layout(int batch, int H, int W, int depth, int rows_call = 0) {
  Share S{};
  if (H < 1 || W < 1 || depth < 1) return S;
  const int rows_batch = (batch + W - 1) / W + 1;
  const int Wq = (((W - 1) >> 2) + 64 + 4 + 1) & ~1;
  const int n3b = sf_frame_count(32, F3B_L, F3B_H), n3a = sf_frame_count(32, F3A_L, F3A_H), npl = sf_frame_count(32, F32_PL, F32_PH);
  long long strip = depth >= 2 ? SHARE_ROWS2 : SHARE_ROWS1;
  if (strip > H) strip = H;
  for (;; strip /= 2) {                      // the largest map + ring pair must stay below 2 GB
    if (strip < 1) return S;                 // (an image too wide to share: rows = 0)
    const long long rows = strip + rows_batch, Hq = ((((rows + 3) >> 2) + 1 + 64 + 4) + 1) & ~1ll;
    const long long H8 = Hq / 2 - 1, W8 = Wq / 2 - 1;
    const long long pair1 = ((long long)16 * Hq * Wq * 192 + (long long)(batch + 1) * 496 * 192) * 4;
    const long long pair2 = ((long long)64 * H8 * W8 * 480 + (long long)(batch + 1) * n3b * 480) * 4;
    if (pair1 < 0x7f000000ll && (depth < 2 || pair2 < 0x7f000000ll)) break;
  }
  S.depth = depth;
  S.rows = (int)strip + rows_batch;
  S.Hq = ((((S.rows + 3) >> 2) + 1 + 64 + 4) + 1) & ~1;
  S.Wq = Wq;
  S.Hc = 4 * S.Hq;
  S.Wc = 4 * S.Wq;
  S.H8 = S.Hq / 2 - 1;
  S.W8 = S.Wq / 2 - 1;
  size_t o = 0;
  auto take = [&](size_t &slot, size_t n) { slot = o; o += (n + 63) / 64 * 64; };
  auto pair = [&](MapT &t, size_t positions, int C, int ringpos) {
    t.C = C;
    t.map = o;
    o += positions * C;                      // (the ring tensor exactly behind the maps)
    t.ring = o;
    o += ((size_t)(batch + 1) * ringpos * C + 63) / 64 * 64;
  };
  take(S.canvas, (size_t)S.Hc * S.Wc + 64);
  take(S.c1, (size_t)(S.Hc / 2) * (S.Wc / 2) * 64);
  take(S.p1, (size_t)S.Hq * S.Wq * 64);
  take(S.p1ring, (size_t)batch * 252 * 64);
  const size_t P16 = (size_t)16 * S.Hq * S.Wq, P64 = (size_t)64 * S.H8 * S.W8;
  pair(S.q2, P16, 64, 252);
  pair(S.q3, P16 + (depth >= 2 ? (size_t)S.Wq + 2 : 0), 192, 496);     // (+ a row: the offset views of the maxpool2 maps)
  if (depth >= 2) {
    pair(S.x3a, P64, 192, npl);
    pair(S.t2a, P64, 96, n3a);
    pair(S.t3a, P64, 16, n3a);
    pair(S.y3a, P64, 256, n3a);
    pair(S.t2b, P64, 128, n3b);
    pair(S.t3b, P64, 32, n3b);
    pair(S.y3b, P64, 480, n3b);
    take(S.pooled, P64 * 256);               // branch 4's pooled input while the maps are built
    take(S.pring, (size_t)batch * n3b * 256);   // ... and at the ring positions of a batch
    S.band = 1;
    S.nrows = rows_batch > STRIP_ROWS ? rows_batch : STRIP_ROWS;
    const size_t n4 = (size_t)S.nrows * 8, n8 = (size_t)S.nrows * 16, p4 = n4 * 16 * S.Wq, p8 = n8 * 8 * S.W8;
    take(S.s_canvas, n4 * 64 * S.Wc);
    take(S.s_c1, n4 * 32 * (S.Wc / 2) * 64);
    take(S.s_p1, p4 * 64);
    take(S.s_q2, p4 * 64);
    take(S.s_q3, (p4 + 2) * 192);            // (+ the column a shifted pool view may name)
    take(S.s_x3, p8 * 192);
    for (int i = 0; i < 2; ++i) {
      take(S.s_t2[i], p8 * INC[i].c3r);
      take(S.s_t3[i], p8 * INC[i].c5r);
      take(S.s_y[i], p8 * (INC[i].c1 + INC[i].c3 + INC[i].c5 + INC[i].pp));
    }
    take(S.s_pool, p8 * 256);
  }
  S.total = o;
  // a call over fewer image rows than a strip serves builds its maps over those rows only (the buffers keep the full strip's offsets:
  // sf_cnn_score_workspace_bytes does not know the rows)
  if (rows_call > 0 && rows_call + 8 < strip) {
    S.rows = rows_call + rows_batch + 8;      // (+ 8: the maps start on an even canvas row, up to 7 image rows above the first)
    S.Hq = ((((S.rows + 3) >> 2) + 1 + 64 + 4) + 1) & ~1;
    S.Hc = 4 * S.Hq;
    S.H8 = S.Hq / 2 - 1;
  }
  return S;
}

struct Net {
  const float *blob;
  Share SH;
  float *share = nullptr;                // the sharing buffers (behind everything else); null: no sharing
  int map_r0 = -1, map_r1 = -1, map_Rb = 0;   // image rows the maps in memory serve, their first canvas row
  bool map_ok = false;
  int strip_r0 = -1, strip_nr = 0;              // image rows the strip maps in memory serve
  int row_end = 0;                              // the call's last image row + 1 (maps and strips are not built past it)
  bool strip_ok = false;
  Blob L; Wino WL; Splits SL; Acts A;
  float *pool1, *conv2, *conv3, *xa, *xb, *t2, *t3, *pooled, *wino, *sscale;
  _Float16 *shalf;
  int *flags;
  float *amax;
  bool wino_ready = false, split_ready = false;
  hipStream_t st;
};

Net make_net(const float *blob, int batch, void *workspace, void *stream, int H = 0, int W = 0, int depth = 0, int r0 = 0, int r1 = 0) {
  Net N{};
  N.blob = blob;
  N.SH = share_layout(batch, H, W, depth, r1 - r0);
  N.row_end = r1;
  N.L = blob_layout(); N.WL = wino_layout(); N.SL = split_layout(); N.A = acts((size_t)batch);
  float *ws = reinterpret_cast<float *>(workspace);
  N.pool1 = ws; N.conv2 = N.pool1 + N.A.pool1; N.conv3 = N.conv2 + N.A.conv2; N.xa = N.conv3 + N.A.conv3; N.xb = N.xa + N.A.x;
  N.t2 = N.xb + N.A.y; N.t3 = N.t2 + N.A.t2; N.pooled = N.t3 + N.A.t3;
  N.wino = N.pooled + N.A.pooled;
  N.sscale = N.wino + N.WL.total;
  N.shalf = reinterpret_cast<_Float16 *>(N.sscale + N.SL.scales);
  N.flags = reinterpret_cast<int *>(N.shalf + N.SL.halves);
  N.amax = reinterpret_cast<float *>(N.flags + NFLAG);
  if (N.SH.rows > 0) N.share = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + base_bytes(batch));
  N.st = (hipStream_t)stream;
  return N;
}
#define W_(l) (N.blob + (l).w)
#define B_(l) (N.blob + (l).b)
_Float16 *half_lo(Net &N, const SplitL &sl, int cout, int taps, int cin) { return N.shalf + sl.h + conv_floats(cout, taps, cin); }

int prepare_split(Net &N) {
  if (N.split_ready) return 0;
  int rc = 0;
  auto prep = [&](const Layer &l, const SplitL &sl, int cout, int taps, int cin) {
    return sf_cnn_split_weights(W_(l), cout, taps * cin, N.shalf + sl.h, half_lo(N, sl, cout, taps, cin), N.sscale + sl.s, N.st);
  };
  if ((rc = prep(N.L.conv2, N.SL.conv2, 64, 1, 64))) return rc;
  if ((rc = prep(N.L.conv3, N.SL.conv3, 192, 9, 64))) return rc;
  for (int i = 0; i < 9; ++i) {
    const Incep &s = INC[i];
    if ((rc = prep(N.L.head3[i], N.SL.head3[i], s.c1 + s.c3r + s.c5r, 1, s.cin))) return rc;
    if ((rc = prep(N.L.b2[i], N.SL.b2[i], s.c3, 9, s.c3r))) return rc;
    if ((rc = prep(N.L.b3[i], N.SL.b3[i], s.c5, 9, s.c5r))) return rc;
    if ((rc = prep(N.L.b4[i], N.SL.b4[i], s.pp, 1, s.cin))) return rc;
  }
  N.split_ready = true;
  return 0;
}
int prepare_wino(Net &N) {
  if (N.wino_ready) return 0;
  int rc = 0;
  if ((rc = sf_cnn_wino_weights(W_(N.L.conv3), 192, 64, N.wino + N.WL.conv3, N.st))) return rc;
  for (int i = 0; i < 9; ++i) {
    if (INC[i].c3r % 8 == 0 && (rc = sf_cnn_wino_weights(W_(N.L.b2[i]), INC[i].c3, INC[i].c3r, N.wino + N.WL.b2[i], N.st))) return rc;
    if (INC[i].c5r % 8 == 0 && (rc = sf_cnn_wino_weights(W_(N.L.b3[i]), INC[i].c5, INC[i].c5r, N.wino + N.WL.b3[i], N.st))) return rc;
  }
  N.wino_ready = true;
  return 0;
}

int build_strips(Net &N, const float *padded, int H, int W, int ra, const float *as);

// One batch of windows through the eval graph (googlenet1.py:110-163) on the given route:
//   0 operand splitting on the fp16 matrix cores (cnn_split.hip; `as`: the NSCALE activation scales, `flag`: this batch's overflow slot)
//   4 Winograd F(2 x 2, 3 x 3) for the 3 x 3 layers where the geometry allows + the fp32 implicit GEMM (cnn_wino.hip / cnn_kernels.hip)
//   2 the direct fp32 implicit GEMM for everything (1: its pointer-form tile loads, chosen inside sf_cnn_conv by sf_debug_set(17, 1))
// amax != nullptr (routes 4 / 2): the largest magnitude of every tensor a split convolution would read is folded into amax[NSCALE]
int run_batch(Net &N, const float *padded, const float *plane, int H, int W, long long tile0, int n, int route, const float *as, int *flag,
              float *amax, float *out) {
  const int Hp = H + 255, Wp = W + 255;
  const bool share = route == 0;             // 0: operand splitting with the trunk up to conv3 shared; 3: every window on its own
  const bool use_split = route == 0 || route == 3, use_wino = route == 4;
  void *stream = (void *)N.st;
  int rc = 0;
  auto conv3x3 = [&](const float *in, int hw, int cin, const Layer &l, size_t uoff, const SplitL &sl, int cout, float a_in, float *o, int ldo,
                     int off) -> int {
    if (use_split)      // (its input -- conv2's output, a 3 x 3 reducer's -- arrives in the split format, scaled by a_in)
      return sf_cnn_conv_split(in, 1, n, hw, hw, cin, cin, N.shalf + sl.h, half_lo(N, sl, cout, 9, cin), N.sscale + sl.s, B_(l), cout, 3, a_in,
                               o, 0, 1.0f, ldo, off, flag, stream);
    if (use_wino && sf_cnn_wino_ok(hw, hw, cin))
      return sf_cnn_conv3x3_wino(in, n, hw, hw, cin, cin, N.wino + uoff, B_(l), cout, o, ldo, off, stream);
    return sf_cnn_conv(in, n, hw, hw, cin, cin, W_(l), B_(l), cout, 3, o, ldo, off, stream);
  };
  auto peak = [&](const float *x, size_t count, int slot) -> int { return amax ? sf_cnn_absmax(x, count, amax + slot, stream) : 0; };
  int hw = pool_out(64, 3, 2, 0);
  int first_block = 0;
  if (share) {
    // conv1 .. maxpool2 (depth 2: .. maxpool3) with everything but each window's padding-dependent ring taken from the phase maps
    const Share &S = N.SH;
    float *B = N.share;
    const int Rb = N.map_Rb, Rb8 = Rb >> 1;
    const int ra = (int)(tile0 / W), rb = (int)((tile0 + n - 1) / W);
    // BAND SHARING (round 6; cnn_ring.h): with strip maps over the batch's image rows only the SIDE positions of every ring are computed
    // per window; the band interiors -- the rows that see the window's top / bottom padding only -- are copied from the strips
    const bool band = S.band && S.depth >= 2 && N.strip_ok && N.strip_r0 >= 0 && N.strip_r0 <= ra && rb < N.strip_r0 + N.strip_nr;
    auto copy = [&](size_t strips, int shift, int Hs, int Wm, int G, int lo, int hi, int C, float *ring) -> int {
      return band ? sfi_cnn_band_copy(B + strips, tile0, n, W, N.strip_r0, N.strip_nr, shift, Hs, Wm, G, lo, hi, C, ring, stream) : 0;
    };
    // a convolution at the ring (or its side) of the frame (olo, ohi), input gathered from `in`'s maps + ring tensor
    auto ring_conv = [&](const MapT &in, int in_split, int rb_, int Hm, int Wm, int shift, int G, int ilo, int ihi, int olo, int ohi, int cin,
                         const Layer &l, const SplitL &sl, int c0, int c1, int c2, int ks, float a_in, float *o0, int ld0, int off0, float *o1,
                         float *o2, int o12_split, float s1, float s2) -> int {
      const int cout = c0 + c1 + c2;
      auto fn = band ? sfi_cnn_conv_side : sf_cnn_conv_ring;
      return fn(B + in.map, in_split, tile0, n, W, rb_, Hm, Wm, in.ring - in.map, shift, G, ilo, ihi, olo, ohi, cin, N.shalf + sl.h,
                half_lo(N, sl, cout, ks * ks, cin), N.sscale + sl.s, B_(l), c0, c1, c2, ks, a_in, o0, ld0, off0, o1, c1, 0, o2, c2, 0, o12_split, s1,
                s2, flag, stream);
    };
    // a 3 x 3 pool at the ring (side) positions of (olo, ohi) on Go into the ring tensor `out`
    auto ring_pool = [&](const MapT &in, int rb_, int Hm, int Wm, int shift, int G, int ilo, int ihi, int C, int stride, int Go, int olo, int ohi,
                         float *out) -> int {
      return band ? sfi_cnn_pool_gather_side(B + in.map, tile0, n, W, rb_, Hm, Wm, in.ring - in.map, shift, G, ilo, ihi, C, stride, Go, olo, ohi, out, 1,
                                             stream)
                  : sf_cnn_pool_gather(B + in.map, tile0, n, W, rb_, Hm, Wm, in.ring - in.map, shift, G, ilo, ihi, C, stride, Go, olo, ohi, out, stream);
    };
    // conv1 + maxpool1 at the border of the 64 x 64 grid, conv2 on it (a plain GEMM over the 252 rows), conv3 at its ring
    if ((rc = band ? sfi_cnn_ring_pool1_side(padded, Hp, Wp, W, tile0, n, W_(N.L.conv1), B_(N.L.conv1), B + S.p1ring, stream)
                   : sf_cnn_ring_pool1(padded, Hp, Wp, W, tile0, n, W_(N.L.conv1), B_(N.L.conv1), B + S.p1ring, stream)))
      return rc;
    if ((rc = copy(S.s_p1, 2, 16, S.Wq, 64, 1, 1, 64, B + S.p1ring))) return rc;
    if ((rc = sf_cnn_conv_split(B + S.p1ring, 0, 1, 1, n * 252, 64, 64, N.shalf + N.SL.conv2.h, half_lo(N, N.SL.conv2, 64, 1, 64),
                                N.sscale + N.SL.conv2.s, B_(N.L.conv2), 64, 1, as[0], B + S.q2.ring, 1, as[1], 64, 0, flag, stream)))
      return rc;
    if ((rc = ring_conv(S.q2, 1, Rb, S.Hq, S.Wq, 2, 64, 1, 1, 2, 2, 64, N.L.conv3, N.SL.conv3, 192, 0, 0, 3, as[1], B + S.q3.ring, 192, 0, nullptr,
                        nullptr, 0, 1.0f, 1.0f)))
      return rc;
    if ((rc = copy(S.s_q3, 2, 16, S.Wq, 64, 2, 2, 192, B + S.q3.ring))) return rc;
    if (S.depth < 2) {
      if ((rc = sf_cnn_pool_gather(B + S.q3.map, tile0, n, W, Rb, S.Hq, S.Wq, S.q3.ring - S.q3.map, 2, 64, 2, 2, 192, 2, 32, -1, 0, N.xa,
                                   stream)))
        return rc;
    } else {
      // maxpool2 at its ring, inception3a and 3b at theirs, maxpool3 assembling inception4a's input (googlenet1.py:64-68, :184-228)
      if ((rc = ring_pool(S.q3, Rb, S.Hq, S.Wq, 2, 64, 2, 2, 192, 2, 32, F32_PL, F32_PH, B + S.x3a.ring))) return rc;
      if ((rc = copy(S.s_x3, 3, 8, S.W8, 32, F32_PL, F32_PH, 192, B + S.x3a.ring))) return rc;
      const MapT *xin = &S.x3a;
      int ilo = F32_PL, ihi = F32_PH;
      for (int i = 0; i < 2; ++i) {
        const Incep &s = INC[i];
        const int cout = s.c1 + s.c3 + s.c5 + s.pp, olo = ilo + 1, ohi = ihi + 1, nout = sf_frame_count(32, olo, ohi);
        const MapT &t2 = i == 0 ? S.t2a : S.t2b, &t3 = i == 0 ? S.t3a : S.t3b, &yo = i == 0 ? S.y3a : S.y3b;
        const float ax = as[2 + 3 * i], a2 = as[3 + 3 * i], a3 = as[4 + 3 * i];
        float *yr = B + yo.ring;
        // branch1 | 3x3 reduce | "5x5" reduce at the OUTPUT frame's ring positions, the input gathered (ring tensor / maps)
        if ((rc = ring_conv(*xin, 0, Rb8, S.H8, S.W8, 3, 32, ilo, ihi, olo, ohi, s.cin, N.L.head3[i], N.SL.head3[i], s.c1, s.c3r, s.c5r, 1, ax, yr,
                            cout, 0, B + t2.ring, B + t3.ring, 1, a2, a3)))
          return rc;
        if ((rc = copy(S.s_t2[i], 3, 8, S.W8, 32, olo, ohi, s.c3r, B + t2.ring))) return rc;
        if ((rc = copy(S.s_t3[i], 3, 8, S.W8, 32, olo, ohi, s.c5r, B + t3.ring))) return rc;
        if ((rc = ring_conv(t2, 1, Rb8, S.H8, S.W8, 3, 32, olo, ohi, olo, ohi, s.c3r, N.L.b2[i], N.SL.b2[i], s.c3, 0, 0, 3, a2, yr, cout, s.c1,
                            nullptr, nullptr, 0, 1.0f, 1.0f)))
          return rc;
        if ((rc = ring_conv(t3, 1, Rb8, S.H8, S.W8, 3, 32, olo, ohi, olo, ohi, s.c5r, N.L.b3[i], N.SL.b3[i], s.c5, 0, 0, 3, a3, yr, cout,
                            s.c1 + s.c3, nullptr, nullptr, 0, 1.0f, 1.0f)))
          return rc;
        // branch 4: the 3 x 3 / 1 pool at the ring (side) positions, then its 1 x 1 convolution over those rows
        if (band) {
          if ((rc = sfi_cnn_pool_gather_side(B + xin->map, tile0, n, W, Rb8, S.H8, S.W8, xin->ring - xin->map, 3, 32, ilo, ihi, s.cin, 1, 32, olo,
                                             ohi, B + S.pring, 2, stream)))
            return rc;
          if ((rc = sfi_cnn_conv_rows_side(B + S.pring, n, 32, olo, ohi, s.cin, N.shalf + N.SL.b4[i].h, half_lo(N, N.SL.b4[i], s.pp, 1, s.cin),
                                           N.sscale + N.SL.b4[i].s, B_(N.L.b4[i]), s.pp, ax, yr, cout, s.c1 + s.c3 + s.c5, flag, stream)))
            return rc;
        } else {
          if ((rc = sf_cnn_pool_gather(B + xin->map, tile0, n, W, Rb8, S.H8, S.W8, xin->ring - xin->map, 3, 32, ilo, ihi, s.cin, 1, 32, olo, ohi,
                                       B + S.pring, stream)))
            return rc;
          if ((rc = sf_cnn_conv_split(B + S.pring, 0, 1, 1, n * nout, s.cin, s.cin, N.shalf + N.SL.b4[i].h, half_lo(N, N.SL.b4[i], s.pp, 1, s.cin),
                                      N.sscale + N.SL.b4[i].s, B_(N.L.b4[i]), s.pp, 1, ax, yr, 0, 1.0f, cout, s.c1 + s.c3 + s.c5, flag, stream)))
            return rc;
        }
        if ((rc = copy(S.s_y[i], 3, 8, S.W8, 32, olo, ohi, cout, yr))) return rc;
        xin = &yo;
        ilo = olo;
        ihi = ohi;
      }
      if ((rc = sf_cnn_pool_gather(B + S.y3b.map, tile0, n, W, Rb8, S.H8, S.W8, S.y3b.ring - S.y3b.map, 3, 32, F3B_L, F3B_H, 480, 2, 16, -1,
                                   0, N.xa, stream)))
        return rc;
      first_block = 2;
      hw = 16;
    }
  } else {
  // conv1 + maxpool1 (googlenet1.py:60-61), conv2, conv3, maxpool2 (:62-64)
  if ((rc = sf_cnn_conv1_pool(padded, Hp, Wp, W, tile0, n, W_(N.L.conv1), B_(N.L.conv1), N.pool1, stream))) return rc;
  if ((rc = peak(N.pool1, (size_t)n * 64 * 64 * 64, 0))) return rc;
  if (use_split)
    rc = sf_cnn_conv_split(N.pool1, 0, n, 64, 64, 64, 64, N.shalf + N.SL.conv2.h, half_lo(N, N.SL.conv2, 64, 1, 64), N.sscale + N.SL.conv2.s,
                           B_(N.L.conv2), 64, 1, as[0], N.conv2, 1, as[1], 64, 0, flag, stream);
  else
    rc = sf_cnn_conv(N.pool1, n, 64, 64, 64, 64, W_(N.L.conv2), B_(N.L.conv2), 64, 1, N.conv2, 64, 0, stream);
  if (rc) return rc;
  if ((rc = peak(N.conv2, (size_t)n * 64 * 64 * 64, 1))) return rc;
  if ((rc = conv3x3(N.conv2, 64, 64, N.L.conv3, N.WL.conv3, N.SL.conv3, 192, use_split ? as[1] : 1.0f, N.conv3, 192, 0))) return rc;
  if ((rc = sf_cnn_maxpool(N.conv3, n, 64, 64, 192, 3, 2, 0, N.xa, hw, hw, stream))) return rc;
  }
  float *x = N.xa, *y = N.xb;
  int cin = first_block == 2 ? 480 : 192;
  for (int i = first_block; i < 9; ++i) {
    const Incep &s = INC[i];
    const int cout = s.c1 + s.c3 + s.c5 + s.pp;
    const float ax = use_split ? as[2 + 3 * i] : 1.0f, a2 = use_split ? as[3 + 3 * i] : 1.0f, a3 = use_split ? as[4 + 3 * i] : 1.0f;
    if ((rc = peak(x, (size_t)n * hw * hw * cin, 2 + 3 * i))) return rc;
    // branch1 | 3x3 reduce | "5x5" reduce in one GEMM, then the two 3x3 convolutions, the pool branch (:184-228)
    // inception4e on the split route: maxpool4 (2 x 2 / 2 on 16 x 16, :75) is taken in the four producers' epilogues (round 6: the pool
    // kernel read 436 MB to write 109) -- the block's output lands as [n][8][8][cout]; sf_debug_set(16, 4) keeps the pool kernel
    const bool pool4 = use_split && i == 6 && hw == 16 && sf_cnn_pool_conv_split_ok(n, hw, hw, cin, s.pp) && sf_tune().cnn_variant != 4;
    if (pool4)
      rc = sfi_cnn_conv_split3_pool2(x, n, cin, cin, N.shalf + N.SL.head3[i].h, half_lo(N, N.SL.head3[i], s.c1 + s.c3r + s.c5r, 1, cin),
                                     N.sscale + N.SL.head3[i].s, B_(N.L.head3[i]), s.c1, s.c3r, s.c5r, ax, y, cout, 0, N.t2, N.t3, a2, a3, flag, stream);
    else if (use_split)
      rc = sf_cnn_conv_split3_split(x, n, hw, hw, cin, cin, N.shalf + N.SL.head3[i].h, half_lo(N, N.SL.head3[i], s.c1 + s.c3r + s.c5r, 1, cin),
                                    N.sscale + N.SL.head3[i].s, B_(N.L.head3[i]), s.c1, s.c3r, s.c5r, ax, y, cout, 0, N.t2, s.c3r, 0, N.t3,
                                    s.c5r, 0, 1, a2, a3, flag, stream);
    else
      rc = sf_cnn_conv_split3(x, n, hw, hw, cin, cin, W_(N.L.head3[i]), B_(N.L.head3[i]), s.c1, s.c3r, s.c5r, y, cout, 0, N.t2, s.c3r, 0,
                              N.t3, s.c5r, 0, stream);
    if (rc) return rc;
    if ((rc = peak(N.t2, (size_t)n * hw * hw * s.c3r, 3 + 3 * i))) return rc;
    if ((rc = peak(N.t3, (size_t)n * hw * hw * s.c5r, 4 + 3 * i))) return rc;
    if (pool4) {
      if ((rc = sfi_cnn_conv_split_pool2(N.t2, 1, n, s.c3r, s.c3r, N.shalf + N.SL.b2[i].h, half_lo(N, N.SL.b2[i], s.c3, 9, s.c3r),
                                         N.sscale + N.SL.b2[i].s, B_(N.L.b2[i]), s.c3, 3, a2, y, cout, s.c1, flag, stream)))
        return rc;
      if ((rc = sfi_cnn_conv_split_pool2(N.t3, 1, n, s.c5r, s.c5r, N.shalf + N.SL.b3[i].h, half_lo(N, N.SL.b3[i], s.c5, 9, s.c5r),
                                         N.sscale + N.SL.b3[i].s, B_(N.L.b3[i]), s.c5, 3, a3, y, cout, s.c1 + s.c3, flag, stream)))
        return rc;
      if ((rc = sfi_cnn_pool_conv_split_pool2(x, n, cin, N.shalf + N.SL.b4[i].h, half_lo(N, N.SL.b4[i], s.pp, 1, cin), N.sscale + N.SL.b4[i].s,
                                              B_(N.L.b4[i]), s.pp, ax, y, cout, s.c1 + s.c3 + s.c5, flag, stream)))
        return rc;
      float *t = x; x = y; y = t;
      cin = cout;
      hw = pool_out(hw, 2, 2, 0);
      continue;
    }
    if ((rc = conv3x3(N.t2, hw, s.c3r, N.L.b2[i], N.WL.b2[i], N.SL.b2[i], s.c3, a2, y, cout, s.c1))) return rc;
    if ((rc = conv3x3(N.t3, hw, s.c5r, N.L.b3[i], N.WL.b3[i], N.SL.b3[i], s.c5, a3, y, cout, s.c1 + s.c3))) return rc;
    if (use_split && sf_cnn_pool_conv_split_ok(n, hw, hw, cin, s.pp))
      rc = sf_cnn_pool_conv_split(x, n, hw, hw, cin, N.shalf + N.SL.b4[i].h, half_lo(N, N.SL.b4[i], s.pp, 1, cin), N.sscale + N.SL.b4[i].s,
                                  B_(N.L.b4[i]), s.pp, ax, y, cout, s.c1 + s.c3 + s.c5, flag, stream);
    else
      rc = sf_cnn_pool_conv(x, n, hw, hw, cin, cin, W_(N.L.b4[i]), B_(N.L.b4[i]), s.pp, y, cout, s.c1 + s.c3 + s.c5, N.pooled, stream);
    if (rc) return rc;
    float *t = x; x = y; y = t;
    cin = cout;
    if (i == 1 || i == 6) {      // maxpool3 after 3b (3x3 s2), maxpool4 after 4e (2x2 s2), both ceil_mode (:68, :75)
      const int k = (i == 1) ? 3 : 2, ho = pool_out(hw, k, 2, 0);
      if ((rc = sf_cnn_maxpool(x, n, hw, hw, cin, k, 2, 0, y, ho, ho, stream))) return rc;
      t = x; x = y; y = t;
      hw = ho;
    }
  }
  // global average pool, FC, softmax[:, 1], NODATA rule (:87-89; cnn_pred_pipeline.py:177-189)
  if (out) return sf_cnn_head(x, n, hw * hw, cin, W_(N.L.fc), B_(N.L.fc), plane, tile0, -9999.0f, out, stream);
  return 0;
}

// The phase maps for the image rows [r0, r0 + SH.rows) (clipped to the plane): per phase (r & 3, c & 3) the plane shifted by the
// phase through conv1 (fully convolutional kernel) -> maxpool1 -> conv2 -> conv3, the last two by operand splitting with the
// batch kernels' scales.  Synchronises the stream (the maps' own overflow slot is read): once per SHARE_ROWS image rows.
// map_ok = false when an activation left float16's range: the strip's batches then run unshared (route 3).
int build_maps(Net &N, const float *padded, int H, int W, int r0, const float *as) {
  const Share &S = N.SH;
  const int Hp = H + 255, Wp = W + 255, Rb = (r0 >> 3) << 1;       // (an even canvas row: the 64-phase maps start on a multiple of 8)
  void *stream = (void *)N.st;
  float *B = N.share;
  float *canvas = B + S.canvas, *c1 = B + S.c1, *p1 = B + S.p1, *q2 = B + S.q2.map, *q3 = B + S.q3.map;
  int *mflag = reinterpret_cast<int *>(N.amax + 48);
  SF_HIP(hipMemsetAsync(mflag, 0, sizeof(int), N.st));
  int rc = 0;
  const size_t plane = (size_t)S.Hq * S.Wq;
  for (int ph = 0; ph < 16; ++ph) {
    if ((rc = sf_cnn_phase_canvas(padded, Hp, Wp, 4 * Rb + (ph >> 2), ph & 3, S.Hc, S.Wc, canvas, stream))) return rc;
    if ((rc = sf_cnn_conv1_image(canvas, 1, S.Hc, S.Wc, W_(N.L.conv1), B_(N.L.conv1), c1, 0, stream))) return rc;
    if ((rc = sf_cnn_maxpool(c1, 1, S.Hc / 2, S.Wc / 2, 64, 3, 2, 0, p1, S.Hq, S.Wq, stream))) return rc;
    if ((rc = sf_cnn_conv_split(p1, 0, 1, S.Hq, S.Wq, 64, 64, N.shalf + N.SL.conv2.h, half_lo(N, N.SL.conv2, 64, 1, 64),
                                N.sscale + N.SL.conv2.s, B_(N.L.conv2), 64, 1, as[0], q2 + ph * plane * 64, 1, as[1], 64, 0, mflag, stream)))
      return rc;
    if ((rc = sf_cnn_conv_split(q2 + ph * plane * 64, 1, 1, S.Hq, S.Wq, 64, 64, N.shalf + N.SL.conv3.h, half_lo(N, N.SL.conv3, 192, 9, 64),
                                N.sscale + N.SL.conv3.s, B_(N.L.conv3), 192, 3, as[1], q3 + ph * plane * 192, 0, 1.0f, 192, 0, mflag,
                                stream)))
      return rc;
  }
  if (S.depth >= 2) {
    // The 64 phase maps of the 32 x 32 grid.  Window (r, c): its conv3 rows 2 p .. 2 p + 2 sit at rows 2 ((r >> 3) - Rb / 2 + p) +
    // ((r >> 2) & 1) + {0, 1, 2} of the conv3 map of phase (r & 3, c & 3): maxpool2's map of phase (r & 7, c & 7) is the 3 x 3 / 2
    // pool of that conv3 map read from row (r >> 2) & 1, column (c >> 2) & 1 on.
    const size_t p8 = (size_t)S.H8 * S.W8;
    for (int a = 0; a < 8; ++a)
      for (int b = 0; b < 8; ++b) {
        const int ph4 = (a & 3) * 4 + (b & 3), oa = a >> 2, ob = b >> 2;
        const float *src = q3 + (ph4 * plane + (size_t)oa * S.Wq + ob) * 192;
        if ((rc = sf_cnn_maxpool(src, 1, S.Hq - oa, S.Wq, 192, 3, 2, 0, B + S.x3a.map + (size_t)(a * 8 + b) * p8 * 192, S.H8, S.W8, stream)))
          return rc;
      }
    // inception3a, 3b fully convolutionally on the 64 maps (googlenet1.py:184-228): reducers' maps in the split format
    const MapT *xin = &S.x3a;
    for (int i = 0; i < 2; ++i) {
      const Incep &s = INC[i];
      const int cout = s.c1 + s.c3 + s.c5 + s.pp;
      const MapT &t2 = i == 0 ? S.t2a : S.t2b, &t3 = i == 0 ? S.t3a : S.t3b, &yo = i == 0 ? S.y3a : S.y3b;
      const float ax = as[2 + 3 * i], a2 = as[3 + 3 * i], a3 = as[4 + 3 * i];
      if ((rc = sf_cnn_conv_split3_split(B + xin->map, 64, S.H8, S.W8, s.cin, s.cin, N.shalf + N.SL.head3[i].h,
                                         half_lo(N, N.SL.head3[i], s.c1 + s.c3r + s.c5r, 1, s.cin), N.sscale + N.SL.head3[i].s,
                                         B_(N.L.head3[i]), s.c1, s.c3r, s.c5r, ax, B + yo.map, cout, 0, B + t2.map, s.c3r, 0, B + t3.map,
                                         s.c5r, 0, 1, a2, a3, mflag, stream)))
        return rc;
      if ((rc = sf_cnn_conv_split(B + t2.map, 1, 64, S.H8, S.W8, s.c3r, s.c3r, N.shalf + N.SL.b2[i].h, half_lo(N, N.SL.b2[i], s.c3, 9, s.c3r),
                                  N.sscale + N.SL.b2[i].s, B_(N.L.b2[i]), s.c3, 3, a2, B + yo.map, 0, 1.0f, cout, s.c1, mflag, stream)))
        return rc;
      if ((rc = sf_cnn_conv_split(B + t3.map, 1, 64, S.H8, S.W8, s.c5r, s.c5r, N.shalf + N.SL.b3[i].h, half_lo(N, N.SL.b3[i], s.c5, 9, s.c5r),
                                  N.sscale + N.SL.b3[i].s, B_(N.L.b3[i]), s.c5, 3, a3, B + yo.map, 0, 1.0f, cout, s.c1 + s.c3, mflag, stream)))
        return rc;
      if ((rc = sf_cnn_maxpool(B + xin->map, 64, S.H8, S.W8, s.cin, 3, 1, 1, B + S.pooled, S.H8, S.W8, stream))) return rc;
      if ((rc = sf_cnn_conv_split(B + S.pooled, 0, 64, S.H8, S.W8, s.cin, s.cin, N.shalf + N.SL.b4[i].h, half_lo(N, N.SL.b4[i], s.pp, 1, s.cin),
                                  N.sscale + N.SL.b4[i].s, B_(N.L.b4[i]), s.pp, 1, ax, B + yo.map, 0, 1.0f, cout, s.c1 + s.c3 + s.c5, mflag,
                                  stream)))
        return rc;
      xin = &yo;
    }
  }
  int raised = 0;
  SF_HIP(hipMemcpyAsync(&raised, mflag, sizeof(int), hipMemcpyDeviceToHost, N.st));
  SF_HIP(hipStreamSynchronize(N.st));
  N.map_r0 = r0;
  N.map_r1 = (r0 + S.rows - 8 < H) ? r0 + S.rows - 8 : H;          // (the even canvas row may sit up to 7 image rows above r0)
  N.map_Rb = Rb;
  N.map_ok = raised == 0;
  return 0;
}

// Band sharing: the strip maps of the image rows [ra, ra + nr) -- conv1 .. inception3b as in build_maps, on the top and the bottom 64
// pixel rows of those rows' windows instead of the whole plane: a strip's first (last) row IS the windows' top (bottom) edge, so the
// dense kernels' own zero padding reproduces what a window sees there; what the strip's far edge contaminates (<= 3 rows of 8 at
// the 32 x 32 grid) lies outside every band.  Images in (row, top | bottom, column phase) order; at the 32 x 32 grid the eight column
// phases come from two pool launches over the four-phase conv3 strips (column offset (phase >> 2): build_maps' views), phases 4 .. 7
// behind 0 .. 3.  One build serves STRIP_ROWS image rows (the launches are small: built per batch they cost as much as they saved) and
// synchronises the stream to read the strips' own overflow slot; strip_ok = false: those rows' batches compute their whole rings.
int build_strips(Net &N, const float *padded, int H, int W, int ra, const float *as) {
  const Share &S = N.SH;
  const int last = (N.row_end > 0 && N.row_end < H) ? N.row_end : H;
  const int nr = (last - ra < S.nrows) ? last - ra : S.nrows;
  const int Hp = H + 255, Wp = W + 255, n4 = nr * 8, n8 = nr * 16;
  void *stream = (void *)N.st;
  float *B = N.share;
  int *flag = reinterpret_cast<int *>(N.amax + 49);
  SF_HIP(hipMemsetAsync(flag, 0, sizeof(int), N.st));
  int rc = 0;
  if ((rc = sfi_cnn_strip_canvas(padded, Hp, Wp, ra, nr, S.Wc, B + S.s_canvas, stream))) return rc;
  if ((rc = sf_cnn_conv1_image(B + S.s_canvas, n4, 64, S.Wc, W_(N.L.conv1), B_(N.L.conv1), B + S.s_c1, 0, stream))) return rc;
  if ((rc = sf_cnn_maxpool(B + S.s_c1, n4, 32, S.Wc / 2, 64, 3, 2, 0, B + S.s_p1, 16, S.Wq, stream))) return rc;
  if ((rc = sf_cnn_conv_split(B + S.s_p1, 0, n4, 16, S.Wq, 64, 64, N.shalf + N.SL.conv2.h, half_lo(N, N.SL.conv2, 64, 1, 64),
                              N.sscale + N.SL.conv2.s, B_(N.L.conv2), 64, 1, as[0], B + S.s_q2, 1, as[1], 64, 0, flag, stream)))
    return rc;
  if ((rc = sf_cnn_conv_split(B + S.s_q2, 1, n4, 16, S.Wq, 64, 64, N.shalf + N.SL.conv3.h, half_lo(N, N.SL.conv3, 192, 9, 64),
                              N.sscale + N.SL.conv3.s, B_(N.L.conv3), 192, 3, as[1], B + S.s_q3, 0, 1.0f, 192, 0, flag, stream)))
    return rc;
  const size_t p8 = (size_t)8 * S.W8;
  for (int ob = 0; ob < 2; ++ob)
    if ((rc = sf_cnn_maxpool(B + S.s_q3 + (size_t)ob * 192, n4, 16, S.Wq, 192, 3, 2, 0, B + S.s_x3 + (size_t)ob * n4 * p8 * 192, 8, S.W8, stream)))
      return rc;
  size_t xin = S.s_x3;
  for (int i = 0; i < 2; ++i) {
    const Incep &s = INC[i];
    const int cout = s.c1 + s.c3 + s.c5 + s.pp;
    const float ax = as[2 + 3 * i], a2 = as[3 + 3 * i], a3 = as[4 + 3 * i];
    float *y = B + S.s_y[i], *t2 = B + S.s_t2[i], *t3 = B + S.s_t3[i];
    if ((rc = sf_cnn_conv_split3_split(B + xin, n8, 8, S.W8, s.cin, s.cin, N.shalf + N.SL.head3[i].h,
                                       half_lo(N, N.SL.head3[i], s.c1 + s.c3r + s.c5r, 1, s.cin), N.sscale + N.SL.head3[i].s,
                                       B_(N.L.head3[i]), s.c1, s.c3r, s.c5r, ax, y, cout, 0, t2, s.c3r, 0, t3, s.c5r, 0, 1, a2, a3, flag, stream)))
      return rc;
    if ((rc = sf_cnn_conv_split(t2, 1, n8, 8, S.W8, s.c3r, s.c3r, N.shalf + N.SL.b2[i].h, half_lo(N, N.SL.b2[i], s.c3, 9, s.c3r),
                                N.sscale + N.SL.b2[i].s, B_(N.L.b2[i]), s.c3, 3, a2, y, 0, 1.0f, cout, s.c1, flag, stream)))
      return rc;
    if ((rc = sf_cnn_conv_split(t3, 1, n8, 8, S.W8, s.c5r, s.c5r, N.shalf + N.SL.b3[i].h, half_lo(N, N.SL.b3[i], s.c5, 9, s.c5r),
                                N.sscale + N.SL.b3[i].s, B_(N.L.b3[i]), s.c5, 3, a3, y, 0, 1.0f, cout, s.c1 + s.c3, flag, stream)))
      return rc;
    if ((rc = sf_cnn_maxpool(B + xin, n8, 8, S.W8, s.cin, 3, 1, 1, B + S.s_pool, 8, S.W8, stream))) return rc;
    if ((rc = sf_cnn_conv_split(B + S.s_pool, 0, n8, 8, S.W8, s.cin, s.cin, N.shalf + N.SL.b4[i].h, half_lo(N, N.SL.b4[i], s.pp, 1, s.cin),
                                N.sscale + N.SL.b4[i].s, B_(N.L.b4[i]), s.pp, 1, ax, y, 0, 1.0f, cout, s.c1 + s.c3 + s.c5, flag, stream)))
      return rc;
    xin = S.s_y[i];
  }
  int raised = 0;
  SF_HIP(hipMemcpyAsync(&raised, flag, sizeof(int), hipMemcpyDeviceToHost, N.st));
  SF_HIP(hipStreamSynchronize(N.st));
  N.strip_r0 = ra;
  N.strip_nr = nr;
  N.strip_ok = raised == 0;
  return 0;
}

// The activation scales from a FIXED sample of the plane's windows (eight groups of up to eight consecutive windows, evenly spaced
// over all H W of them -- a function of the plane alone, not of `batch`, so every row shard and batch size of a flightline works with the same
// scales and produces the same bits): one pass on the fp32 matrix cores, the largest magnitude of every tensor a split convolution
// reads, then the power of two that puts it into [2^9, 2^10)
int calibrate(Net &N, const float *padded, int H, int W, int batch, float *scales) {
  int rc = 0;
  if ((rc = prepare_wino(N))) return rc;
  SF_HIP(hipMemsetAsync(N.amax, 0, NSCALE * sizeof(float), N.st));
  const long long T = (long long)H * W;
  const int per = (int)(T < 8 ? T : 8);               // windows per group: independent of `batch` (run in pieces of <= batch)
  const int groups = (T <= per) ? 1 : 8;
  long long last = -1;
  for (int g = 0; g < groups; ++g) {
    const long long t0 = (groups == 1) ? 0 : (T - per) * g / (groups - 1);
    if (t0 <= last) continue;                         // (tiny planes: groups that coincide)
    last = t0;
    for (int k = 0; k < per; k += batch) {
      const int n = (per - k < batch) ? per - k : batch;
      if ((rc = run_batch(N, padded, nullptr, H, W, t0 + k, n, 4, nullptr, nullptr, N.amax, nullptr))) return rc;
    }
  }
  float mx[NSCALE];
  SF_HIP(hipMemcpyAsync(mx, N.amax, sizeof(mx), hipMemcpyDeviceToHost, N.st));
  SF_HIP(hipStreamSynchronize(N.st));
  for (int i = 0; i < NSCALE; ++i) {
    float s = 1.0f;
    if (mx[i] > 0.f && mx[i] < 3.0e38f) {
      int ex;
      frexpf(mx[i], &ex);                             // mx = f 2^ex, f in [0.5, 1)
      int e = 10 - ex;
      e = e > 40 ? 40 : (e < -40 ? -40 : e);
      s = ldexpf(1.0f, e);
    }
    scales[i] = s;
  }
  return 0;
}

size_t base_bytes(int batch) {
  const Splits S = split_layout();
  return sf_align((acts((size_t)batch).total + wino_layout().total + S.scales) * sizeof(float) + S.halves * 2 + tail_bytes());
}

}  // namespace

extern "C" {

size_t sf_cnn_blob_floats(void) { return blob_layout().total; }
int sf_cnn_num_scales(void) { return NSCALE; }
size_t sf_cnn_score_workspace_bytes(int batch, int H, int W) {
  if (batch < 1) return 0;
  return base_bytes(batch) + sf_align(share_layout(batch, H, W, 2).total * sizeof(float));    // (the deeper form's: it is the larger)
}

int sf_cnn_calibrate(const float *padded, int H, int W, const float *blob, int batch, void *workspace, size_t workspace_bytes,
                     float *scales, void *stream) {
  if (!padded || !blob || !workspace || !scales || H < 1 || W < 1 || batch < 1) { sf_set_error("sf_cnn_calibrate: bad argument"); return -1; }
  if (workspace_bytes < sf_cnn_score_workspace_bytes(batch, 0, 0)) {
    sf_set_error("sf_cnn_calibrate: workspace too small: need %zu bytes, got %zu", sf_cnn_score_workspace_bytes(batch, 0, 0), workspace_bytes);
    return -4;
  }
  Net N = make_net(blob, batch, workspace, stream);
  return calibrate(N, padded, H, W, batch, scales);
}

int sf_cnn_score_rows(const float *padded, const float *plane, int H, int W, int r0, int r1, const float *blob, float *out,
                      int batch, int route, const float *scales, int *info, void *workspace, size_t workspace_bytes, void *stream) {
  if (!padded || !blob || !out || !workspace || H < 1 || W < 1 || r0 < 0 || r1 > H || r0 > r1 || batch < 1 ||
      (route != 0 && route != 5 && route != 3 && route != 4 && route != 2 && route != 1)) {
    sf_set_error("sf_cnn_score_rows: bad argument");
    return -1;
  }
  const bool sharing = route == 0 || route == 5;       // 0: through inception3b (depth 2); 5: through conv3 (depth 1, round 6's first form)
  const size_t need = sf_cnn_score_workspace_bytes(batch, sharing ? H : 0, sharing ? W : 0);
  if (workspace_bytes < need) {
    sf_set_error("sf_cnn_score_rows: workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    return -4;
  }
  if (info) info[0] = info[1] = 0;          // [0] batches scored again on the fp32 matrix cores, [1] batches that ran on the shared trunk
  Net N = make_net(blob, batch, workspace, stream, sharing ? H : 0, sharing ? W : 0, route == 0 ? 2 : 1, r0, r1);
  const long long i0 = (long long)r0 * W, i1 = (long long)r1 * W;
  if (i0 >= i1) return 0;
  int rc = 0;
  if (!sharing && route != 3) {
    if (route == 4 && (rc = prepare_wino(N))) return rc;
    for (long long tile0 = i0; tile0 < i1; tile0 += batch) {
      const int n = (int)((i1 - tile0 < batch) ? (i1 - tile0) : batch);
      if ((rc = run_batch(N, padded, plane, H, W, tile0, n, route, nullptr, nullptr, nullptr, out))) return rc;
    }
    return 0;
  }
  // route 0: operand splitting with its two range contracts.  Underflow: the per-layer activation scales (the caller's, or a
  // calibration pass over the plane).  Overflow: every batch owns a device flag; the flags are read back every NFLAG batches and
  // the batches that raised theirs are scored again on the fp32 matrix cores -- so this route synchronises the stream before it
  // returns, and what it returns is never silently wrong
  float as[NSCALE];
  if (scales) {
    for (int i = 0; i < NSCALE; ++i) {
      int e;
      if (!(scales[i] > 0.f && scales[i] < 3.0e38f && frexpf(scales[i], &e) == 0.5f)) {
        sf_set_error("sf_cnn_score_rows: scales[%d] is not a power of two", i);
        return -1;
      }
      as[i] = scales[i];
    }
  } else if ((rc = calibrate(N, padded, H, W, batch, as))) return rc;
  if ((rc = prepare_split(N))) return rc;
  int host_flags[NFLAG];
  long long group0 = i0;
  while (group0 < i1) {
    SF_HIP(hipMemsetAsync(N.flags, 0, NFLAG * sizeof(int), N.st));
    int nb = 0;
    long long tile0 = group0;
    for (; tile0 < i1 && nb < NFLAG; tile0 += batch, ++nb) {
      const int n = (int)((i1 - tile0 < batch) ? (i1 - tile0) : batch);
      int rt = 3;
      if (sharing && N.share) {                // the phase maps must cover the batch's image rows
        const int rf = (int)(tile0 / W), rl = (int)((tile0 + n - 1) / W);
        if (!(N.map_r0 >= 0 && N.map_r0 <= rf && rl < N.map_r1))
          if ((rc = build_maps(N, padded, H, W, rf, as))) return rc;
        if (N.map_ok) rt = 0;
        if (N.map_ok && info) ++info[1];
        if (N.map_ok && N.SH.band && route == 0 && sf_tune().cnn_variant != 3 && !(N.strip_r0 >= 0 && N.strip_r0 <= rf && rl < N.strip_r0 + N.strip_nr))
          if ((rc = build_strips(N, padded, H, W, rf, as))) return rc;
      }
      if ((rc = run_batch(N, padded, plane, H, W, tile0, n, rt, as, N.flags + nb, nullptr, out))) return rc;
    }
    SF_HIP(hipMemcpyAsync(host_flags, N.flags, nb * sizeof(int), hipMemcpyDeviceToHost, N.st));
    SF_HIP(hipStreamSynchronize(N.st));
    for (int b = 0; b < nb; ++b) {
      if (!host_flags[b]) continue;
      const long long t0 = group0 + (long long)b * batch;
      const int n = (int)((i1 - t0 < batch) ? (i1 - t0) : batch);
      if ((rc = prepare_wino(N))) return rc;
      if ((rc = run_batch(N, padded, plane, H, W, t0, n, 4, nullptr, nullptr, nullptr, out))) return rc;
      if (info) ++info[0];
    }
    group0 = tile0;
  }
  SF_HIP(hipStreamSynchronize(N.st));
  return 0;
}
#undef W_
#undef B_

}  // extern "C"
