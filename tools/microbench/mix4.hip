// Does VALU work issued between 4x4x4 fp64 MFMAs cost matrix throughput?  (1 wave/SIMD, 16 chains)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NI, int ND>
__global__ __launch_bounds__(256) void k(double *out, int iters, long long *cyc) {
  double acc[16];
  for (int c = 0; c < 16; ++c) acc[c] = 0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  int iv[8]; double dv[8];
  for (int c = 0; c < 8; ++c) { iv[c] = threadIdx.x + c; dv[c] = threadIdx.x * 1e-3 + c; }
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < NI; ++f) iv[(c * NI + f) & 7] = iv[(c * NI + f) & 7] * 3 + 1 ^ i;
#pragma unroll
      for (int f = 0; f < ND; ++f) dv[(c * ND + f) & 7] = __builtin_fma(dv[(c * ND + f) & 7], 1.0000001, 1e-9);
    }
  }
  const long long t1 = clock64();
  double s = 0;
  for (int c = 0; c < 16; ++c) s += acc[c];
  for (int c = 0; c < 8; ++c) s += iv[c] + dv[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NI, int ND> void run(double *out, long long *cyc, int ncu) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<NI, ND>), dim3(ncu), dim3(256), 0, 0, out, iters, cyc);
  hipLaunchKernelGGL((k<NI, ND>), dim3(ncu), dim3(256), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("per MFMA: %d int VALU + %d fp64 FMA -> %.1f cycles per MFMA slot\n", NI, ND, (double)h / (iters * 16.0));
}
int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  double *out; long long *cyc; hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 16);
  const int n = prop.multiProcessorCount;
  run<0, 0>(out, cyc, n); run<1, 0>(out, cyc, n); run<2, 0>(out, cyc, n); run<3, 0>(out, cyc, n); run<4, 0>(out, cyc, n); run<6, 0>(out, cyc, n);
  run<0, 1>(out, cyc, n); run<0, 2>(out, cyc, n); run<0, 3>(out, cyc, n); run<2, 1>(out, cyc, n);
  return 0;
}
