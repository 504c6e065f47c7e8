// What does each class of non-MFMA instruction cost the 4x4x4 fp64 MFMA stream at ONE and at TWO waves per SIMD?
// (round 3: the sweep kernel's mix is 814 MFMA + ~290 ds_read_b128 + ~170 s_waitcnt + ~150 s_nop + ~100 int VALU + ~220 fp64 VALU
// per 16-row tile.)  One workgroup per CU (100 KB of LDS), NW waves; per 8 MFMAs (8 independent chains) PER fillers of one kind.
// Build: hipcc --offload-arch=gfx950 -O3 mix4w.hip -o mix4w
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2_t __attribute__((ext_vector_type(2)));

template <int KIND, int PER, int NW>
__global__ __launch_bounds__(64 * NW, 1) void k(double *out, int iters, long long *cyc) {
  extern __shared__ double sm[];
  double acc[8];
  for (int c = 0; c < 8; ++c) acc[c] = 0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  int iv[8]; double dv[8]; d2_t lv[8];
  for (int c = 0; c < 8; ++c) { iv[c] = threadIdx.x + c; dv[c] = 1.0 + threadIdx.x * 1e-9 + c; lv[c] = d2_t{0, 0}; }
  for (int i = threadIdx.x; i < 12800; i += 64 * NW) sm[i] = i;
  __syncthreads();
  const unsigned adr = (unsigned)(size_t)sm + (threadIdx.x & 63) * 16;
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
      if (c < PER) {
        if (KIND == 1) asm volatile("v_or_b32 %0, %0, %1" : "+v"(iv[c]) : "v"(i));
        if (KIND == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(dv[c]) : "v"(b));
        if (KIND == 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(lv[c]) : "v"(adr), "n"(1024 * (PER > 0 ? 1 : 0)));
        if (KIND == 4) asm volatile("s_nop 0");
        if (KIND == 5) asm volatile("s_waitcnt lgkmcnt(15)");
        if (KIND == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(iv[c]) : "v"(i) : );
        if (KIND == 7) asm volatile("v_add_f64 %0, %0, %1" : "+v"(dv[c]) : "v"(b));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lv[0]), "+v"(lv[1]), "+v"(lv[2]), "+v"(lv[3]), "+v"(lv[4]), "+v"(lv[5]), "+v"(lv[6]), "+v"(lv[7]));
  }
  const long long t1 = clock64();
  double s = 0;
  for (int c = 0; c < 8; ++c) s += acc[c] + iv[c] + dv[c] + lv[c].x + lv[c].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// the sweep's own mix per 8 MFMAs: 3 ds_read_b128, 2 s_waitcnt (counted), 1 s_nop, 2 fp64 VALU, 1 int VALU
template <int NW, int SCALE>
__global__ __launch_bounds__(64 * NW, 1) void kmix(double *out, int iters, long long *cyc) {
  extern __shared__ double sm[];
  double acc[8];
  for (int c = 0; c < 8; ++c) acc[c] = 0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  int iv[8]; double dv[8]; d2_t lv[8];
  for (int c = 0; c < 8; ++c) { iv[c] = threadIdx.x + c; dv[c] = 1.0 + threadIdx.x * 1e-9 + c; lv[c] = d2_t{0, 0}; }
  for (int i = threadIdx.x; i < 12800; i += 64 * NW) sm[i] = i;
  __syncthreads();
  const unsigned adr = (unsigned)(size_t)sm + (threadIdx.x & 63) * 16;
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
      if (SCALE >= 1) {
        if (c == 0 || c == 3 || c == 6) asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(lv[c]) : "v"(adr));
        if (c == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(lv[0]));
        if (c == 5) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(lv[3]));
        if (c == 2) asm volatile("s_nop 0");
      }
      if (SCALE >= 2) {
        if (c == 4) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(dv[c]) : "v"(b));
        if (c == 7) asm volatile("v_add_f64 %0, %0, %1" : "+v"(dv[c]) : "v"(b));
        if (c == 2) asm volatile("v_or_b32 %0, %0, %1" : "+v"(iv[c]) : "v"(i));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = clock64();
  double s = 0;
  for (int c = 0; c < 8; ++c) s += acc[c] + iv[c] + dv[c] + lv[c].x + lv[c].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int PER, int NW> void run(double *out, long long *cyc, int ncu, const char *name) {
  const int iters = 20000;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<KIND, PER, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 102400);
  hipLaunchKernelGGL((k<KIND, PER, NW>), dim3(ncu), dim3(64 * NW), 102400, 0, out, iters, cyc);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, PER, NW>), dim3(ncu), dim3(64 * NW), 102400, 0, out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%d waves/SIMD  %d x %-22s per 8 MFMA -> %5.2f s_memtime ticks per MFMA on the SIMD, %6.1f TFLOP/s by wall clock (%.1f ns per MFMA and SIMD)\n", NW / 4, PER, name,
         (double)h / (iters * 8.0 * (NW / 4)), 512.0 * ncu * NW * iters * 8.0 / ms / 1e9, ms * 1e6 / (iters * 8.0 * (NW / 4)));
}
template <int NW, int SCALE> void runmix(double *out, long long *cyc, int ncu) {
  const int iters = 20000;
  hipFuncSetAttribute(reinterpret_cast<const void *>(kmix<NW, SCALE>), hipFuncAttributeMaxDynamicSharedMemorySize, 102400);
  hipLaunchKernelGGL((kmix<NW, SCALE>), dim3(ncu), dim3(64 * NW), 102400, 0, out, iters, cyc);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((kmix<NW, SCALE>), dim3(ncu), dim3(64 * NW), 102400, 0, out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%d waves/SIMD  sweep mix level %d (1: LDS reads + waits + nop, 2: + 2 fp64 + 1 int VALU) -> %5.2f ticks per MFMA on the SIMD, %6.1f TFLOP/s by wall clock\n",
         NW / 4, SCALE, (double)h / (iters * 8.0 * (NW / 4)), 512.0 * ncu * NW * iters * 8.0 / ms / 1e9);
}
#define BOTH(KIND, PER, name) run<KIND, PER, 4>(out, cyc, n, name); run<KIND, PER, 8>(out, cyc, n, name);
int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  double *out; long long *cyc; hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 16);
  const int n = prop.multiProcessorCount;
  BOTH(0, 0, "nothing")
  BOTH(1, 1, "v_or_b32") BOTH(1, 2, "v_or_b32") BOTH(1, 4, "v_or_b32") BOTH(1, 8, "v_or_b32")
  BOTH(2, 1, "v_mul_f64") BOTH(2, 2, "v_mul_f64") BOTH(2, 4, "v_mul_f64") BOTH(2, 8, "v_mul_f64")
  BOTH(7, 2, "v_add_f64") BOTH(7, 4, "v_add_f64")
  BOTH(3, 2, "ds_read_b128") BOTH(3, 4, "ds_read_b128") BOTH(3, 8, "ds_read_b128")
  BOTH(4, 2, "s_nop 0") BOTH(4, 4, "s_nop 0") BOTH(4, 8, "s_nop 0")
  BOTH(5, 2, "s_waitcnt") BOTH(5, 4, "s_waitcnt") BOTH(5, 8, "s_waitcnt")
  BOTH(6, 2, "v_cndmask_b32") BOTH(6, 4, "v_cndmask_b32")
  runmix<4, 0>(out, cyc, n); runmix<8, 0>(out, cyc, n);
  runmix<4, 1>(out, cyc, n); runmix<8, 1>(out, cyc, n);
  runmix<4, 2>(out, cyc, n); runmix<8, 2>(out, cyc, n);
  return 0;
}
