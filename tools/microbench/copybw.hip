// What do plain copies and pure write streams reach on this box?  (The extract kernel moves 6.9 GB per flightline at
// 4.4 TB/s; profiles/r01_measured_peaks.txt has a 16 B/lane copy at 4.75 TB/s; the guide quotes 6.29 TB/s for a float4 copy.)
// Variants: plain / non-temporal loads and stores, elements per thread in flight, grid = many workgroups or persistent.
// Build: hipcc --offload-arch=gfx950 -O3 copybw.hip -o copybw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)

template <int U, bool NTL, bool NTS, bool WRITE_ONLY>
__global__ __launch_bounds__(256) void k_copy(const f4_t *__restrict__ src, f4_t *__restrict__ dst, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base < n4; base += stride) {
    f4_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + (size_t)u * 256;
      if (WRITE_ONLY) v[u] = f4_t{1.f, 2.f, 3.f, (float)u};
      else if (i < n4) v[u] = NTL ? __builtin_nontemporal_load(src + i) : src[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + (size_t)u * 256;
      if (i < n4) { if (NTS) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u]; }
    }
  }
}

template <typename F>
float time_ms(F f, int reps) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  f();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  for (int r = 0; r < reps; ++r) f();
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  const size_t bytes = (size_t)3440 << 20;   // the size of one xt / one active window
  f4_t *src, *dst;
  CK(hipMalloc(&src, bytes));
  CK(hipMalloc(&dst, bytes));
  CK(hipMemset(src, 1, bytes));
  CK(hipMemset(dst, 0, bytes));
  const size_t n4 = bytes / 16;
#define RUN(U, NTL, NTS, WO, GRID)                                                                                   \
  do {                                                                                                               \
    const int g = (GRID) > 0 ? (GRID) : (int)((n4 + 256 * U - 1) / (256 * U));                                       \
    float ms = time_ms([&] { hipLaunchKernelGGL((k_copy<U, NTL, NTS, WO>), dim3(g), dim3(256), 0, 0, src, dst, n4); }, 3); \
    printf("%s U %d loads %s stores %s grid %7d : %7.1f GB/s %s (%.3f ms)\n", WO ? "write" : "copy ", U, NTL ? "nt   " : "plain", \
           NTS ? "nt   " : "plain", g, (WO ? 1.0 : 2.0) * bytes / ms / 1e6, WO ? "written" : "read+written", ms);    \
  } while (0)
  RUN(1, false, false, false, 0);
  RUN(4, false, false, false, 0);
  RUN(4, true, false, false, 0);
  RUN(4, false, true, false, 0);
  RUN(4, true, true, false, 0);
  RUN(8, true, true, false, 0);
  RUN(4, true, true, false, 2048);
  RUN(4, true, true, false, 4096);
  RUN(8, true, true, false, 1024);
  RUN(4, false, false, true, 0);
  RUN(4, false, true, true, 0);
  RUN(8, false, true, true, 0);
  RUN(4, false, true, true, 2048);
  RUN(4, false, false, true, 2048);
  return 0;
}
