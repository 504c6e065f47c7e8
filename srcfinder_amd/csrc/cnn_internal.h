// Entry points between the tile scorer's translation units that are NOT part of the C ABI (hidden visibility): the side-row forms of
// the ring kernels that the band-sharing driver (cnn_driver.hip, round 6) sequences.  Geometry: cnn_ring.h.
#pragma once
#include <cstddef>

// sf_cnn_conv_ring with the rows restricted to the SIDE positions of the output frame (G (olo + ohi) per window); row (window, j) is
// written to row window * count(G, olo, ohi) + its ring index of out0 / out1 / out2 -- the full ring tensors, whose band interior
// sfi_cnn_band_copy fills
int sfi_cnn_conv_side(const float *maps, int in_split, long long tile0, int N, int W, int Rb, int Hq, int Wq, size_t ring_off, int shift, int G,
                      int ilo, int ihi, int olo, int ohi, int Cin, const void *whi, const void *wlo, const float *wscale, const float *bias,
                      int c0, int c1, int c2, int ksize, float ascale, float *out0, int ld0, int off0, float *out1, int ld1, int off1,
                      float *out2, int ld2, int off2, int out12_split, float oscale1, float oscale2, int *overflow, void *stream);
// a 1 x 1 convolution of the compact rows in[N * G (olo + ohi)][Cin] (float32), written like sfi_cnn_conv_side's
int sfi_cnn_conv_rows_side(const float *in, int N, int G, int olo, int ohi, int Cin, const void *whi, const void *wlo, const float *wscale,
                           const float *bias, int Cout, float ascale, float *out, int ld_out, int ch_off, int *overflow, void *stream);
// sf_cnn_pool_gather at the SIDE positions of the output frame (olo, ohi) on Go: side = 1 -> out[N][count(Go, olo, ohi)][C] (the full ring
// tensor, rows scattered), side = 2 -> compact out[N][Go (olo + ohi)][C]
int sfi_cnn_pool_gather_side(const float *maps, long long tile0, int N, int W, int Rb, int Hq, int Wq, size_t ring_off, int shift, int G, int ilo,
                             int ihi, int C, int stride, int Go, int olo, int ohi, float *out, int side, void *stream);
// sf_cnn_ring_pool1 at the side positions only (px = 0 or 63) of the 64 x 64 grid's border: out[ntiles][252][64], the other rows untouched
int sfi_cnn_ring_pool1_side(const float *padded, int Hp, int Wp, int W, long long tile0, int ntiles, const float *w, const float *bias,
                            float *out, void *stream);
// The band interior of a ring tensor from the strip maps of the batch's image rows row0 .. row0 + nrows - 1: strips[image][Hs][Wq][C],
//   image = (ph >> 2) (8 nrows) + ((row(n) - row0) * 2 + bottom) * 4 + (ph & 3),  ph = c & (2^shift - 1)
//   ring[n][index(y, x)][:] = strips[image][ys][(c >> shift) + x][:]
// for y < lo (top strip, ys = y) and y >= G - hi (bottom strip, ys = Hs - (G - y)), lo <= x < G - hi; window n = pixel tile0 + n = (row, c).
// (shift 2: four column phases, images in (row, bottom, phase) order; shift 3: eight -- the images of phases 4 .. 7 behind those of 0 .. 3,
//  the order in which two pool launches produce them from the four-phase strips.)
int sfi_cnn_band_copy(const float *strips, long long tile0, int N, int W, int row0, int nrows, int shift, int Hs, int Wq, int G, int lo, int hi,
                      int C, float *ring, void *stream);
// canvas[(ri * 2 + bottom) * 4 + phase][64][Wc] = the top / bottom 64 pixel rows of the windows of image row row0 + ri, from column `phase` on
int sfi_cnn_strip_canvas(const float *padded, int Hp, int Wp, int row0, int nrows, int Wc, float *canvas, void *stream);

// inception4e with maxpool4 (googlenet1.py:75: 2 x 2 stride 2 on the 16 x 16 grid) taken in the convolutions' epilogues: the float32
// outputs land pooled, [window][8][8][ld] -- sf_cnn_conv_split / sf_cnn_conv_split3_split (its first segment; the reducers' split-format
// outputs stay on the 16 x 16 grid) / sf_cnn_pool_conv_split with H = W = 16.  Same bits as pooling afterwards (monotone epilogue).
int sfi_cnn_conv_split_pool2(const float *in, int in_split, int N, int Cin, int ld_in, const void *whi, const void *wlo, const float *wscale,
                             const float *bias, int Cout, int ksize, float ascale, float *out, int ld_out, int ch_off, int *overflow, void *stream);
int sfi_cnn_conv_split3_pool2(const float *in, int N, int Cin, int ld_in, const void *whi, const void *wlo, const float *wscale, const float *bias,
                              int c0, int c1, int c2, float ascale, float *out0, int ld0, int off0, float *out1, float *out2, float oscale1,
                              float oscale2, int *overflow, void *stream);
int sfi_cnn_pool_conv_split_pool2(const float *in, int N, int Cin, const void *whi, const void *wlo, const float *wscale, const float *bias, int Cout,
                                  float ascale, float *out, int ld_out, int ch_off, int *overflow, void *stream);
