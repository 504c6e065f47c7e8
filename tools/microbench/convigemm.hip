// Where does k_conv_igemm<32,64> (the implicit-GEMM convolution of the tile scorer) spend its time?  conv3's shape
// (3x3, 64 -> 192 channels on 64x64 pixels, 256 tiles) with one phase removed at a time (EXP bits, see cnn_kernels.hip).
// Build (from the repo root): hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Isrcfinder_amd/csrc \
//   tools/microbench/convigemm.hip -Lsrcfinder_amd -lsrcfinder_amd -Wl,-rpath,'$ORIGIN/../../srcfinder_amd' -o tools/microbench/convigemm
#include "../../srcfinder_amd/csrc/cnn_kernels.hip"
#include <cstdio>
#include <vector>

template <int EXP, int BN = 64, bool BUF = false>
void run(const float *in, int N, int H, int W, int Cin, const float *wt, const float *bias, int Cout, int ks, float *out) {
  ConvDst d;
  for (int i = 0; i < 3; ++i) { d.p[i] = out; d.ld[i] = Cout; d.off[i] = 0; d.end[i] = Cout; }
  const int M = N * H * W;
  dim3 grid((M + 127) / 128, (Cout + BN - 1) / BN);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_conv_igemm<32, BN, EXP, BUF>), grid, dim3(256), 0, 0, in, M, H, W, Cin, Cin, wt, bias, Cout, ks, d);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 3; ++r)
    hipLaunchKernelGGL((k_conv_igemm<32, BN, EXP, BUF>), grid, dim3(256), 0, 0, in, M, H, W, Cin, Cin, wt, bias, Cout, ks, d);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 3;
  const double fl = 2.0 * M * Cout * Cin * ks * ks;
  printf("%s BN %3d ks %d Cin %3d Cout %3d %dx%d EXP %2d: %8.1f us  %6.1f TFLOP/s (nominal)\n", BUF ? "buf" : "ptr", BN, ks, Cin, Cout, H, W, EXP, ms * 1e3f, fl / ms / 1e9);
}

int main(int argc, char **) {
  const int N = 256;
  float *in, *wt, *bias, *out;
  (void)hipMalloc(&in, (size_t)N * 64 * 64 * 256 * 4);
  (void)hipMalloc(&wt, (size_t)512 * 9 * 512 * 4);
  (void)hipMalloc(&bias, 4096);
  (void)hipMalloc(&out, (size_t)N * 64 * 64 * 288 * 4);
  (void)hipMemset(in, 0, (size_t)N * 64 * 64 * 256 * 4);
  (void)hipMemset(wt, 0, (size_t)512 * 9 * 512 * 4);
  (void)hipMemset(bias, 0, 4096);
  if (argc > 1) {   // counter runs: the production kernel and the one without global loads, conv3's shape
    run<0>(in, N, 64, 64, 64, wt, bias, 192, 3, out);
    run<1>(in, N, 64, 64, 64, wt, bias, 192, 3, out);
    run<3>(in, N, 64, 64, 64, wt, bias, 192, 3, out);
    run<0, 64, true>(in, N, 64, 64, 64, wt, bias, 192, 3, out);
    return 0;
  }
#define ALL(...)                                                                                           \
  run<0>(__VA_ARGS__); run<1>(__VA_ARGS__); run<2>(__VA_ARGS__); run<3>(__VA_ARGS__); run<4>(__VA_ARGS__); \
  run<8>(__VA_ARGS__); run<11>(__VA_ARGS__); run<7>(__VA_ARGS__);
  ALL(in, N, 64, 64, 64, wt, bias, 192, 3, out);          // conv3
  ALL(in, N * 4, 32, 32, 256, wt, bias, 288, 1, out);     // inception 3b's 1x1 triple (N*4 images of 32x32 = the same pixel count)
  run<0, 64, true>(in, N, 64, 64, 64, wt, bias, 192, 3, out);
  run<0, 192, true>(in, N, 64, 64, 64, wt, bias, 192, 3, out);
  run<0, 64, true>(in, N * 4, 32, 32, 256, wt, bias, 288, 1, out);
  run<0, 96, true>(in, N * 4, 32, 32, 256, wt, bias, 288, 1, out);
  run<0, 64, true>(in, N * 4, 32, 32, 128, wt, bias, 192, 3, out);
#define BNS(...) run<0, 96>(__VA_ARGS__); run<0, 128>(__VA_ARGS__); run<0, 160>(__VA_ARGS__); run<0, 192>(__VA_ARGS__);
  BNS(in, N, 64, 64, 64, wt, bias, 192, 3, out);
  BNS(in, N * 4, 32, 32, 256, wt, bias, 288, 1, out);
  BNS(in, N * 4, 32, 32, 128, wt, bias, 192, 3, out);     // inception 3b branch 2
  run<0, 64>(in, N * 4, 32, 32, 128, wt, bias, 192, 3, out);
  return 0;
}
