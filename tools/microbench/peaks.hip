// Measured peaks on the box (reported beside the vendor figures in DESIGN.md):
//   fp64 MFMA issue rate, fp64 VALU FMA rate, HBM streaming read bandwidth by access width.
// Build: hipcc --offload-arch=gfx950 -O3 peaks.hip -o peaks ; run: ./peaks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)

template <int CHAINS>
__global__ __launch_bounds__(256) void k_mfma64(double *out, int iters) {
  d4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// MFMA + independent fp64 VALU FMAs in the same wave: do the two pipes overlap?
template <int NFMA>
__global__ __launch_bounds__(256) void k_mix(double *out, int iters, long long *cyc) {
  d4 acc[4];
  for (int c = 0; c < 4; ++c) acc[c] = d4{0, 0, 0, 0};
  double x[8];
  for (int c = 0; c < 8; ++c) x[c] = threadIdx.x * 1e-3 + c;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  const double fa = 1.0000001, fb = 1e-9;
  const long long t0 = clock64();
  const long long w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < NFMA; ++f) x[(c * NFMA + f) & 7] = __builtin_fma(x[(c * NFMA + f) & 7], fa, fb);
    }
  }
  const long long t1 = clock64();
  const long long w1 = wall_clock64();
  double s = 0;
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  for (int c = 0; c < 8; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

__global__ __launch_bounds__(256) void k_fma64(double *out, int iters) {
  double x[8];
  for (int c = 0; c < 8; ++c) x[c] = threadIdx.x * 1e-3 + c;
  const double a = 1.0000001, b = 1e-9;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < 8; ++c) x[c] = __builtin_fma(x[c], a, b);
  }
  double s = 0;
  for (int c = 0; c < 8; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename T>
__global__ __launch_bounds__(256) void k_read(const T *__restrict__ in, size_t n, float *out) {
  float s = 0;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    T a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
    const float *pa = (const float *)&a, *pb = (const float *)&b, *pc = (const float *)&c, *pd = (const float *)&d;
    s += pa[0] + pb[0] + pc[0] + pd[0];
  }
  for (; i < n; i += stride) { T a = in[i]; s += ((const float *)&a)[0]; }
  if (s == 123.456f) out[0] = s;
}

__global__ __launch_bounds__(256) void k_copy16(const float4 *__restrict__ in, float4 *__restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) o[i] = in[i];
}

template <typename F>
float time_ms(F f, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < reps; ++r) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs %d clock %d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
  double *out;
  CK(hipMalloc(&out, 1 << 24));
  const int iters = 4000;
  {
    const int blocks = prop.multiProcessorCount * 2;  // 8 waves per CU = 2 per SIMD
    float ms = time_ms([&] { hipLaunchKernelGGL(k_mfma64<4>, dim3(blocks), dim3(256), 0, 0, out, iters); }, 3);
    double flops = (double)blocks * 4 * iters * 4 * 2048.0;
    printf("mfma_f64_16x16x4 (2 waves/SIMD x 4 chains): %.1f TFLOP/s, %.1f cycles/instr/SIMD at %d MHz\n",
           flops / ms / 1e9, (double)ms * 1e-3 * prop.clockRate * 1e3 / (2.0 * iters * 4), prop.clockRate / 1000);
    const int blocks1 = prop.multiProcessorCount;  // 1 wave per SIMD
    ms = time_ms([&] { hipLaunchKernelGGL(k_mfma64<4>, dim3(blocks1), dim3(256), 0, 0, out, iters); }, 3);
    flops = (double)blocks1 * 4 * iters * 4 * 2048.0;
    printf("mfma_f64_16x16x4 (1 wave/SIMD x 4 chains): %.1f TFLOP/s\n", flops / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_mfma64<1>, dim3(blocks1), dim3(256), 0, 0, out, iters); }, 3);
    flops = (double)blocks1 * 4 * iters * 1 * 2048.0;
    printf("mfma_f64_16x16x4 (1 wave/SIMD x 1 dependent chain): %.1f TFLOP/s, %.1f cycles/instr\n", flops / ms / 1e9,
           (double)ms * 1e-3 * prop.clockRate * 1e3 / iters);
  }
  {
    long long *cyc;
    CK(hipMalloc(&cyc, 16));
    const int blocks1 = prop.multiProcessorCount;  // 1 wave per SIMD, like the sweep kernel
    auto mix = [&](auto kern, int nf) {
      float ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(blocks1), dim3(256), 0, 0, out, iters, cyc); }, 3);
      long long h[2];
      hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
      const double mf = (double)blocks1 * 4 * iters * 4 * 2048.0, vf = (double)blocks1 * 256 * iters * 4.0 * nf * 2.0;
      printf("mix 1 MFMA + %2d fp64 FMA (1 wave/SIMD): MFMA %.1f TF + VALU %.1f TF; %.1f shader cycles per MFMA slot; clock %.2f GHz\n",
             nf, mf / ms / 1e9, vf / ms / 1e9, (double)h[0] / (iters * 4.0), (double)h[0] / ((double)h[1] * 10.0) );
    };
    mix(k_mix<0>, 0); mix(k_mix<2>, 2); mix(k_mix<4>, 4); mix(k_mix<6>, 6); mix(k_mix<8>, 8); mix(k_mix<12>, 12); mix(k_mix<16>, 16);
  }
  {
    const int blocks = prop.multiProcessorCount * 8;
    float ms = time_ms([&] { hipLaunchKernelGGL(k_fma64, dim3(blocks), dim3(256), 0, 0, out, iters); }, 3);
    double flops = (double)blocks * 256 * iters * 8 * 2.0;
    printf("v_fma_f64 (8 waves/SIMD): %.1f TFLOP/s\n", flops / ms / 1e9);
  }
  {
    const size_t bytes = (size_t)4 << 30;
    float *buf, *buf2;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&buf2, bytes));
    CK(hipMemset(buf, 0, bytes));
    CK(hipMemset(buf2, 0, bytes));
    const int blocks = prop.multiProcessorCount * 16;
    float ms = time_ms([&] { hipLaunchKernelGGL(k_read<float>, dim3(blocks), dim3(256), 0, 0, buf, bytes / 4, (float *)out); }, 5);
    printf("HBM read, 4 B/lane : %.0f GB/s\n", bytes / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_read<float2>, dim3(blocks), dim3(256), 0, 0, (float2 *)buf, bytes / 8, (float *)out); }, 5);
    printf("HBM read, 8 B/lane : %.0f GB/s\n", bytes / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_read<float4>, dim3(blocks), dim3(256), 0, 0, (float4 *)buf, bytes / 16, (float *)out); }, 5);
    printf("HBM read, 16 B/lane: %.0f GB/s\n", bytes / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_copy16, dim3(blocks), dim3(256), 0, 0, (float4 *)buf, (float4 *)buf2, bytes / 16); }, 5);
    printf("HBM copy, 16 B/lane: %.0f GB/s (read+write)\n", 2.0 * bytes / ms / 1e6);
  }
  return 0;
}
