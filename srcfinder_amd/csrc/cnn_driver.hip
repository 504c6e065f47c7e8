// C-side driver of the tile scorer: the whole eval graph of cnn/archs/googlenet1.py for a range of image rows in ONE call
// (sf_cnn_score_rows).  It sequences the public kernels of cnn_kernels.hip -- the same launches, in the
// same order, as srcfinder_amd/cnn.py::GoogLeNetHIP -- so a C caller needs no Python to score a flightline:
//   cnn_pred_pipeline.py:159-189   for each batch of tiles: model(tile) -> softmax[:, 1]; NODATA where the plane is NODATA
// Weights: ONE float32 blob of BatchNorm-folded convolutions in the order of sf_cnn_blob_layout() (conv weights
// [Cout][k*k][Cin], then the bias [Cout]); GoogLeNetHIP.packed_blob() builds it from a state_dict.
#include "cmf_common.h"

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

}  // namespace

extern "C" {

size_t sf_cnn_blob_floats(void) { return blob_layout().total; }
size_t sf_cnn_score_workspace_bytes(int batch) {
  const Splits S = split_layout();
  return batch < 1 ? 0 : sf_align((acts((size_t)batch).total + wino_layout().total + S.scales) * sizeof(float) + S.halves * 2);
}

int sf_cnn_score_rows(const float *padded, const float *plane, int H, int W, int r0, int r1, const float *blob, float *out,
                      int batch, void *workspace, size_t workspace_bytes, void *stream) {
  if (!padded || !blob || !out || !workspace || H < 1 || W < 1 || r0 < 0 || r1 > H || r0 > r1 || batch < 1) {
    sf_set_error("sf_cnn_score_rows: bad argument");
    return -1;
  }
  if (workspace_bytes < sf_cnn_score_workspace_bytes(batch)) {
    sf_set_error("sf_cnn_score_rows: workspace too small: need %zu bytes, got %zu", sf_cnn_score_workspace_bytes(batch), workspace_bytes);
    return -4;
  }
  const Blob L = blob_layout();
  const Acts A = acts((size_t)batch);
  float *ws = reinterpret_cast<float *>(workspace);
  float *pool1 = ws, *conv2 = pool1 + A.pool1, *conv3 = conv2 + A.conv2, *xa = conv3 + A.conv3, *xb = xa + A.x,
        *t2 = xb + A.y, *t3 = t2 + A.t2, *pooled = t3 + A.t3;
  const int Hp = H + 255, Wp = W + 255;
  const long long i0 = (long long)r0 * W, i1 = (long long)r1 * W;
  int rc = 0;
#define W_(l) (blob + (l).w)
#define B_(l) (blob + (l).b)
  // sf_debug_set(17, .): 0 every convolution but conv1 and the pool-projections by operand splitting on the fp16 matrix cores
  // (cnn_split.hip; the default), 4 round 5's first form -- the 3 x 3 convolutions by Winograd F(2 x 2, 3 x 3) where the geometry
  // allows, the rest on the fp32 matrix cores --, 2 the direct fp32 kernel for everything
  const Wino WL = wino_layout();
  const Splits SL = split_layout();
  float *wino = pooled + A.pooled;
  float *sscale = wino + WL.total;
  _Float16 *shalf = reinterpret_cast<_Float16 *>(sscale + SL.scales);
  const int mode = sf_tune().cnn_conv_variant;
  const bool use_split = mode == 0, use_wino = mode == 4;
  auto half_lo = [&](const SplitL &sl, int cout, int taps, int cin) { return shalf + sl.h + conv_floats(cout, taps, cin); };
  auto conv3x3 = [&](const float *in, int n, int hw, int cin, const Layer &l, size_t uoff, const SplitL &sl, int cout, float *o,
                     int ldo, int off) -> int {
    if (use_split)      // (its input -- conv2's output, a 3 x 3 reducer's -- arrives in the split format)
      return sf_cnn_conv_split(in, 1, n, hw, hw, cin, cin, shalf + sl.h, half_lo(sl, cout, 9, cin), sscale + sl.s, B_(l), cout, 3, 1.0f,
                               o, 0, ldo, off, stream);
    if (use_wino && sf_cnn_wino_ok(hw, hw, cin))
      return sf_cnn_conv3x3_wino(in, n, hw, hw, cin, cin, wino + uoff, B_(l), cout, o, ldo, off, stream);
    return sf_cnn_conv(in, n, hw, hw, cin, cin, W_(l), B_(l), cout, 3, o, ldo, off, stream);
  };
  if (use_split && i0 < i1) {
    auto prep = [&](const Layer &l, const SplitL &sl, int cout, int taps, int cin) {
      return sf_cnn_split_weights(W_(l), cout, taps * cin, shalf + sl.h, half_lo(sl, cout, taps, cin), sscale + sl.s, stream);
    };
    if ((rc = prep(L.conv2, SL.conv2, 64, 1, 64))) return rc;
    if ((rc = prep(L.conv3, SL.conv3, 192, 9, 64))) return rc;
    for (int i = 0; i < 9; ++i) {
      const Incep &s = INC[i];
      if ((rc = prep(L.head3[i], SL.head3[i], s.c1 + s.c3r + s.c5r, 1, s.cin))) return rc;
      if ((rc = prep(L.b2[i], SL.b2[i], s.c3, 9, s.c3r))) return rc;
      if ((rc = prep(L.b3[i], SL.b3[i], s.c5, 9, s.c5r))) return rc;
      if ((rc = prep(L.b4[i], SL.b4[i], s.pp, 1, s.cin))) return rc;
    }
  }
  if (use_wino && i0 < i1) {
    if ((rc = sf_cnn_wino_weights(W_(L.conv3), 192, 64, wino + WL.conv3, stream))) return rc;
    for (int i = 0; i < 9; ++i) {
      if (INC[i].c3r % 8 == 0 && (rc = sf_cnn_wino_weights(W_(L.b2[i]), INC[i].c3, INC[i].c3r, wino + WL.b2[i], stream))) return rc;
      if (INC[i].c5r % 8 == 0 && (rc = sf_cnn_wino_weights(W_(L.b3[i]), INC[i].c5, INC[i].c5r, wino + WL.b3[i], stream))) return rc;
    }
  }
  for (long long tile0 = i0; tile0 < i1; tile0 += batch) {
    const int n = (int)((i1 - tile0 < batch) ? (i1 - tile0) : batch);
    // conv1 + maxpool1 (googlenet1.py:60-61), conv2, conv3, maxpool2 (:62-64)
    if ((rc = sf_cnn_conv1_pool(padded, Hp, Wp, W, tile0, n, W_(L.conv1), B_(L.conv1), pool1, stream))) return rc;
    if (use_split)
      rc = sf_cnn_conv_split(pool1, 0, n, 64, 64, 64, 64, shalf + SL.conv2.h, half_lo(SL.conv2, 64, 1, 64), sscale + SL.conv2.s,
                             B_(L.conv2), 64, 1, 1.0f, conv2, 1, 64, 0, stream);
    else
      rc = sf_cnn_conv(pool1, n, 64, 64, 64, 64, W_(L.conv2), B_(L.conv2), 64, 1, conv2, 64, 0, stream);
    if (rc) return rc;
    if ((rc = conv3x3(conv2, n, 64, 64, L.conv3, WL.conv3, SL.conv3, 192, conv3, 192, 0))) return rc;
    int hw = pool_out(64, 3, 2, 0);
    if ((rc = sf_cnn_maxpool(conv3, n, 64, 64, 192, 3, 2, 0, xa, hw, hw, stream))) return rc;
    float *x = xa, *y = xb;
    int cin = 192;
    for (int i = 0; i < 9; ++i) {
      const Incep &s = INC[i];
      const int cout = s.c1 + s.c3 + s.c5 + s.pp;
      // branch1 | 3x3 reduce | "5x5" reduce in one GEMM, then the two 3x3 convolutions, the pool branch (:184-228)
      if (use_split)
        rc = sf_cnn_conv_split3_split(x, n, hw, hw, cin, cin, shalf + SL.head3[i].h, half_lo(SL.head3[i], s.c1 + s.c3r + s.c5r, 1, cin),
                                      sscale + SL.head3[i].s, B_(L.head3[i]), s.c1, s.c3r, s.c5r, 1.0f, y, cout, 0, t2, s.c3r, 0, t3,
                                      s.c5r, 0, 1, stream);
      else
        rc = sf_cnn_conv_split3(x, n, hw, hw, cin, cin, W_(L.head3[i]), B_(L.head3[i]), s.c1, s.c3r, s.c5r, y, cout, 0, t2, s.c3r, 0,
                                t3, s.c5r, 0, stream);
      if (rc) return rc;
      if ((rc = conv3x3(t2, n, hw, s.c3r, L.b2[i], WL.b2[i], SL.b2[i], s.c3, y, cout, s.c1))) return rc;
      if ((rc = conv3x3(t3, n, hw, s.c5r, L.b3[i], WL.b3[i], SL.b3[i], s.c5, y, cout, s.c1 + s.c3))) return rc;
      if (use_split && sf_cnn_pool_conv_split_ok(n, hw, hw, cin, s.pp))
        rc = sf_cnn_pool_conv_split(x, n, hw, hw, cin, shalf + SL.b4[i].h, half_lo(SL.b4[i], s.pp, 1, cin), sscale + SL.b4[i].s, B_(L.b4[i]),
                                    s.pp, y, cout, s.c1 + s.c3 + s.c5, stream);
      else
        rc = sf_cnn_pool_conv(x, n, hw, hw, cin, cin, W_(L.b4[i]), B_(L.b4[i]), s.pp, y, cout, s.c1 + s.c3 + s.c5, pooled, stream);
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
    if ((rc = sf_cnn_head(x, n, hw * hw, cin, W_(L.fc), B_(L.fc), plane, tile0, -9999.0f, out, stream))) return rc;
  }
#undef W_
#undef B_
  return 0;
}

}  // extern "C"
