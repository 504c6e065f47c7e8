// fp64 MFMA issue-rate probe: chains x waves/SIMD x instruction shape (is 41-46 TFLOP/s the real ceiling?).
// Build: hipcc --offload-arch=gfx950 -O3 mfma64.hip -o mfma64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int CHAINS, int SHAPE>
__global__ __launch_bounds__(256) void k(double *out, int iters, long long *cyc) {
  d4 acc[CHAINS];
  double acc1[CHAINS];
  for (int c = 0; c < CHAINS; ++c) { acc[c] = d4{0, 0, 0, 0}; acc1[c] = 0; }
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      if (SHAPE == 16) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
      else acc1[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1[c], 0, 0, 0);
    }
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3] + acc1[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

template <int CHAINS, int SHAPE>
void run(int wps, double *out, long long *cyc, int ncu) {
  const int iters = 4000, blocks = ncu * wps;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<CHAINS, SHAPE>), dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<CHAINS, SHAPE>), dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
  const double fl = (SHAPE == 16 ? 2048.0 : 512.0) * blocks * 4.0 * iters * CHAINS;
  printf("mfma_f64_%s chains %2d waves/SIMD %d: %6.1f TFLOP/s  %.1f shader cycles/instr/wave  clock %.2f GHz\n",
         SHAPE == 16 ? "16x16x4" : "4x4x4  ", CHAINS, wps, fl / ms / 1e9, (double)h[0] / ((double)iters * CHAINS),
         (double)h[0] / ((double)h[1] * 10.0));
}

int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  double *out; long long *cyc; hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 16);
  const int n = prop.multiProcessorCount;
  run<1, 16>(1, out, cyc, n); run<2, 16>(1, out, cyc, n); run<4, 16>(1, out, cyc, n); run<8, 16>(1, out, cyc, n); run<16, 16>(1, out, cyc, n);
  run<4, 16>(2, out, cyc, n); run<8, 16>(2, out, cyc, n); run<4, 16>(4, out, cyc, n); run<8, 16>(4, out, cyc, n);
  run<1, 4>(1, out, cyc, n); run<4, 4>(1, out, cyc, n); run<16, 4>(1, out, cyc, n); run<8, 4>(2, out, cyc, n); run<8, 4>(4, out, cyc, n);
  return 0;
}
