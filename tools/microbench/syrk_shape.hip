// The register shape of k_syrk4 without memory: 18 operand registers, 171 accumulators, same MFMA order.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
constexpr int NG = 18;
constexpr int tri_index(int I, int J) { return I * NG - I * (I - 1) / 2 + (J - I); }
__global__ __launch_bounds__(256, 1) void k(double *out, int iters, long long *cyc) {
  double acc[171], f[NG];
  for (int t = 0; t < 171; ++t) acc[t] = 0;
  for (int i = 0; i < NG; ++i) f[i] = threadIdx.x * 1e-3 + i;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    static_for<0, NG>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      static_for<I, NG>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        acc[tri_index(I, J)] = __builtin_amdgcn_mfma_f64_4x4x4f64(f[I], f[J], acc[tri_index(I, J)], 0, 0, 0);
      });
    });
#pragma unroll
    for (int i = 0; i < NG; ++i) asm volatile("" : "+v"(f[i]));
  }
  const long long t1 = clock64();
  double s = 0;
  for (int t = 0; t < 171; ++t) s += acc[t];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  double *out; long long *cyc; hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 16);
  const int iters = 500;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(prop.multiProcessorCount), dim3(256), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("syrk shape: %.1f cycles per MFMA\n", (double)h / (iters * 171.0));
  return 0;
}
