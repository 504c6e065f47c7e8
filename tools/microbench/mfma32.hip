// fp32 / fp16 MFMA issue-rate probe (what is the real ceiling of the CNN's implicit-GEMM kernels?)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
template <int CHAINS, int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  f16v acc[CHAINS];
  f4v acc4[CHAINS];
  for (int c = 0; c < CHAINS; ++c) { for (int i = 0; i < 16; ++i) acc[c][i] = 0; for (int i = 0; i < 4; ++i) acc4[c][i] = 0; }
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-6f;
  h8v ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(threadIdx.x * 1e-3f); hb[i] = (_Float16)1.0f; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      if (KIND == 0) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
      if (KIND == 1) acc4[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[c], 0, 0, 0);
      if (KIND == 2) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[c], 0, 0, 0);
    }
  }
  float s = 0;
  for (int c = 0; c < CHAINS; ++c) { for (int i = 0; i < 16; ++i) s += acc[c][i]; for (int i = 0; i < 4; ++i) s += acc4[c][i]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS, int KIND> void run(const char *name, double flop, int wps, float *out, int ncu) {
  const int iters = 4000, blocks = ncu * wps;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<CHAINS, KIND>), dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<CHAINS, KIND>), dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  printf("%s chains %d waves/SIMD %d: %7.1f TFLOP/s\n", name, CHAINS, wps, flop * blocks * 4.0 * iters * CHAINS / ms / 1e9);
}
int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  float *out; hipMalloc(&out, 1 << 24);
  const int n = prop.multiProcessorCount;
  run<4, 0>("mfma_f32_32x32x2_f32 ", 4096.0, 1, out, n); run<4, 0>("mfma_f32_32x32x2_f32 ", 4096.0, 2, out, n); run<4, 0>("mfma_f32_32x32x2_f32 ", 4096.0, 4, out, n);
  run<8, 1>("mfma_f32_16x16x4_f32 ", 2048.0, 1, out, n); run<8, 1>("mfma_f32_16x16x4_f32 ", 2048.0, 2, out, n);
  run<4, 2>("mfma_f32_32x32x16_f16", 32768.0, 1, out, n); run<4, 2>("mfma_f32_32x32x16_f16", 32768.0, 2, out, n); run<4, 2>("mfma_f32_32x32x16_f16", 32768.0, 4, out, n);
  return 0;
}
