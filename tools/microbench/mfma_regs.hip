// 4x4x4 fp64 MFMA issue rate vs where the operands live (VGPR/AGPR) and how many distinct registers are used.
#include <hip/hip_runtime.h>
#include <cstdio>
#define MF(ACC, A, B, CA, CB, CC) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : CC(ACC) : CA(A), CB(B))
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(double *out, int iters, long long *cyc) {
  double acc[32], a[8], b[8];
  for (int t = 0; t < 32; ++t) acc[t] = 0;
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3 + i; b[i] = 1.0 + threadIdx.x * 1e-6 * i; }
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      if (MODE == 0) MF(acc[c], a[0], b[0], "v", "v", "+a");          // same A,B regs, acc AGPR
      if (MODE == 1) MF(acc[c], a[c & 7], b[(c >> 2) & 7], "v", "v", "+a");  // distinct VGPR A,B; acc AGPR
      if (MODE == 2) MF(acc[c], a[c & 7], b[(c >> 2) & 7], "v", "a", "+a");  // B in AGPR
      if (MODE == 3) MF(acc[c], a[c & 7], b[(c >> 2) & 7], "v", "v", "+v");  // all VGPR
      if (MODE == 4) MF(acc[c], a[c & 7], b[(c >> 2) & 7], "v", "a", "+v");  // B AGPR, acc VGPR
      if (MODE == 5) MF(acc[c], a[c & 7], a[(c >> 2) & 7], "v", "v", "+a");  // A and B from the same VGPR set (syrk)
      if (MODE == 6) MF(acc[c], a[c & 7], b[(c >> 2) & 7], "a", "a", "+v");  // A,B AGPR, acc VGPR
    }
  }
  const long long t1 = clock64();
  double s = 0;
  for (int t = 0; t < 32; ++t) s += acc[t];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char *what, double *out, long long *cyc, int ncu) {
  const int iters = 2000;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<MODE>, dim3(ncu), dim3(256), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-52s %.1f cycles per MFMA\n", what, (double)h / (iters * 32.0));
}
int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  double *out; long long *cyc; hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 16);
  const int n = prop.multiProcessorCount;
  run<0>("same A,B VGPR; acc AGPR", out, cyc, n);
  run<1>("distinct A,B VGPR; acc AGPR", out, cyc, n);
  run<2>("A VGPR, B AGPR; acc AGPR", out, cyc, n);
  run<3>("A,B VGPR; acc VGPR", out, cyc, n);
  run<4>("A VGPR, B AGPR; acc VGPR", out, cyc, n);
  run<5>("A,B from one VGPR set; acc AGPR", out, cyc, n);
  run<6>("A,B AGPR; acc VGPR", out, cyc, n);
  return 0;
}
