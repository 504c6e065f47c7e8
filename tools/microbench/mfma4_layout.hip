// Register layout probe for v_mfma_f64_4x4x4f64 (4 blocks of 4x4x4): which A lane / B lane feeds which D lane.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CB, int AB>
__global__ void k(double *out) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la) {
    double a = lane == la ? 1.0 : 0.0, b = 1.0 + lane;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CB, AB, 0);
    out[la * 64 + lane] = d;
  }
}
#include <cstdlib>
int main(int argc, char **argv) {
  double *o; hipMalloc(&o, 64 * 64 * 8);
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  if (mode == 0) hipLaunchKernelGGL((k<0, 0>), dim3(1), dim3(64), 0, 0, o);
  if (mode == 1) hipLaunchKernelGGL((k<2, 0>), dim3(1), dim3(64), 0, 0, o);
  if (mode == 2) hipLaunchKernelGGL((k<2, 1>), dim3(1), dim3(64), 0, 0, o);
  if (mode == 3) hipLaunchKernelGGL((k<2, 3>), dim3(1), dim3(64), 0, 0, o);
  static double h[64 * 64]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  printf("mode %d\n", mode);
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d ->", la);
    for (int l = 0; l < 64; ++l) if (h[la * 64 + l] != 0) printf("  D[%2d]=B[%2d]", l, (int)h[la * 64 + l] - 1);
    printf("\n");
  }
  return 0;
}
