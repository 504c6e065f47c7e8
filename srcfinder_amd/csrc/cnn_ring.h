// Trunk sharing of the CNN tile scorer (cnn_share.hip, cnn_split.hip): the geometry shared by its kernels.
//
// A layer's activation for one window lives on a G x G grid.  Its FRAME (lo, hi) says how far the window's zero padding reaches
// into that grid: rows / columns 0 .. lo - 1 and G - hi .. G - 1 -- the RING -- hold values only this window has; the interior
// [lo, G - 1 - hi]^2 equals the layer evaluated fully convolutionally on the whole padded plane at the window's phase (the phase
// MAPS).  Ring positions are enumerated: the lo top rows, the hi bottom rows (G positions each), then for the rows between them
// the lo left and the hi right columns.
//   window (r, c), 2^shift phases per axis:  phase = (r & (P - 1)) P + (c & (P - 1)),  origin in its map ((r >> shift) - Rb, c >> shift)
//   grid 64 (conv2 / conv3: shift 2): maxpool1 and conv2 (1, 1), conv3 (2, 2)
//   grid 32 (shift 3): maxpool2 (1, 2), inception3a (2, 3), inception3b (3, 4);  grid 16 (after maxpool3): (2, 3)
// (cnn/archs/googlenet1.py:60-68, :110-123: each 3 x 3 / stride-1 layer widens both sides by one, a 3 x 3 stride-2 ceil-mode pool
//  maps (lo, hi) on G to (ceil(lo / 2), floor((hi + 2) / 2)) on G / 2.)
#pragma once

struct SfFrame { int G, lo, hi; };

__host__ __device__ inline int sf_frame_count(int G, int lo, int hi) { return (lo + hi) * G + (G - lo - hi) * (lo + hi); }
__host__ __device__ inline bool sf_frame_ring(int G, int lo, int hi, int y, int x) {
  return y < lo || y >= G - hi || x < lo || x >= G - hi;
}
__host__ __device__ inline int sf_frame_index(int G, int lo, int hi, int y, int x) {      // ring position (y, x) -> 0 .. count - 1
  if (y < lo) return y * G + x;
  if (y >= G - hi) return (lo + y - (G - hi)) * G + x;
  return (lo + hi) * G + (y - lo) * (lo + hi) + (x < lo ? x : lo + x - (G - hi));
}
__host__ __device__ inline void sf_frame_position(int G, int lo, int hi, int j, int &y, int &x) {
  const int band = (lo + hi) * G;
  if (j < band) {
    const int row = j / G;
    x = j - row * G;
    y = row < lo ? row : G - hi + (row - lo);
  } else {
    const int k = j - band, w = lo + hi, row = k / w, t = k - row * w;
    y = lo + row;
    x = t < lo ? t : G - hi + (t - lo);
  }
}

// Round 6, band sharing: the ring splits into its SIDE part -- the columns x < lo and x >= G - hi of EVERY row (G (lo + hi) positions:
// they see the window's left / right padding) -- and the BAND interior -- rows y < lo and y >= G - hi at lo <= x < G - hi: those see
// only the window's top / bottom padding, i.e. they are the same for every window of an image row at the same column phase, and come
// from STRIP maps (the layer run on 64 canvas rows starting at the window's top edge / ending at its bottom edge: cnn_driver.hip).
// Side positions are enumerated row by row: j = y (lo + hi) + (x < lo ? x : lo + x - (G - hi)).
__host__ __device__ inline int sf_side_count(int G, int lo, int hi) { return G * (lo + hi); }
__host__ __device__ inline void sf_side_position(int G, int lo, int hi, int j, int &y, int &x) {
  const int w = lo + hi;
  y = j / w;
  const int t = j - y * w;
  x = t < lo ? t : G - hi + (t - lo);
}
__host__ __device__ inline int sf_band_count(int G, int lo, int hi) { return (lo + hi) * (G - lo - hi); }
__host__ __device__ inline void sf_band_position(int G, int lo, int hi, int j, int &y, int &x) {     // band interior, row by row
  const int w = G - lo - hi, row = j / w;
  x = lo + (j - row * w);
  y = row < lo ? row : G - hi + (row - lo);
}

// What a gather kernel needs to find (window, position) of a layer's INPUT: the per-window ring tensor [N][count][C] or the phase maps
// [P * P][Hq][Wq][C].  Both live in ONE allocation (maps first, the ring tensor `ring_off` floats behind their start) so that a single
// < 2 GB buffer descriptor serves a tap wherever it lands.
struct SfGather {
  long long tile0;     // first window of the batch: window t = image pixel (t / W, t % W)
  int W;               // image width
  int Rb, Hq, Wq;      // the maps' first row (units of P image rows) and their geometry
  int shift;           // log2 P
  int G, lo, hi;       // the input tensor's grid and frame
  unsigned ring_off;   // float offset of the ring tensor behind the maps
};
