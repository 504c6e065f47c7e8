// Streaming-READ bandwidth of the box by load form: what is the ceiling the score kernel (cmf_score.hip) can reach?
//   forms: plain / non-temporal global_load_dwordx{1,2,4}, LDS-DMA (global_load_lds_dwordx4, plain / nt),
//   grid-stride vs workgroup-contiguous chunks, loads in flight per lane, workgroups per CU,
//   and the cube's own access shape: rows of 2392 B (598 samples) read as 64-sample (256 B) or whole-row pieces.
// Build: hipcc --offload-arch=gfx950 -O3 readbw.hip -o readbw ; run: ./readbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)

template <typename T, bool NT>
__device__ __forceinline__ T ld(const T *p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}
__device__ __forceinline__ float first(float v) { return v; }
__device__ __forceinline__ float first(f2_t v) { return v.x; }
__device__ __forceinline__ float first(f4_t v) { return v.x + v.w; }

// workgroup-contiguous: workgroup b streams [b*chunk, (b+1)*chunk) with U loads in flight per lane
template <typename T, int U, bool NT>
__global__ __launch_bounds__(256) void k_chunk(const T *__restrict__ in, size_t chunk_elems, float *out) {
  const T *p = in + (size_t)blockIdx.x * chunk_elems;
  float s = 0;
  for (size_t i = threadIdx.x; i < chunk_elems; i += 256 * U) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<T, NT>(p + i + (size_t)u * 256);
#pragma unroll
    for (int u = 0; u < U; ++u) s += first(v[u]);
  }
  if (s == 123.456f) out[0] = s;
}

// grid-stride with U loads in flight
template <typename T, int U, bool NT>
__global__ __launch_bounds__(256) void k_stride(const T *__restrict__ in, size_t n, float *out) {
  float s = 0;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<T, NT>(in + i + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) s += first(v[u]);
  }
  if (s == 123.456f) out[0] = s;
}

// LDS-DMA: every wave streams its share of the workgroup's chunk into a private LDS ring, never reads it
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_ldsdma(const f4_t *__restrict__ in, size_t chunk_elems, float *out) {
  extern __shared__ f4_t ring[];                       // [4 waves][U][64]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const f4_t *p = in + (size_t)blockIdx.x * chunk_elems;
  f4_t *my = ring + (size_t)wave * U * 64;
  for (size_t i = threadIdx.x; i < chunk_elems; i += 256 * U) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + i + (size_t)u * 256),
                                       (__attribute__((address_space(3))) void *)(my + u * 64), 16, 0, NT ? 2 : 0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U / 2) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 999) out[0] = my[0].x;
}

// the cube's shape: [lines][425][598] float32, active window 72 bands from band 350.
// (a) lane = sample, 64-sample column blocks (the round-1 score kernel's loads): wave instruction = 256 B at an arbitrary
//     4-byte offset;  (b) whole rows: a 320-thread workgroup reads rows of 598 floats as f2_t (299 lanes, 8-byte aligned).
template <int LPI, int UB, bool NT>
__global__ __launch_bounds__(256) void k_cube64(const float *__restrict__ cube, int L, int B, int C, int b0, int p,
                                                int lines_per_wg, int ncb, float *out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cbi = blockIdx.x % ncb, chunk = blockIdx.x / ncb;
  const int col = min(cbi * 64 + lane, C - 1);
  const int lbeg = chunk * lines_per_wg, lend = min(L, lbeg + lines_per_wg);
  float s = 0;
  for (int l = lbeg + wave * LPI; l < lend; l += 4 * LPI) {
    for (int bc = 0; bc < p; bc += UB) {
      float v[LPI][UB];
#pragma unroll
      for (int bb = 0; bb < UB; ++bb)
#pragma unroll
        for (int j = 0; j < LPI; ++j)
          v[j][bb] = ld<float, NT>(cube + ((size_t)min(l + j, lend - 1) * B + b0 + min(bc + bb, p - 1)) * C + col);
#pragma unroll
      for (int bb = 0; bb < UB; ++bb)
#pragma unroll
        for (int j = 0; j < LPI; ++j) s += v[j][bb];
    }
  }
  if (s == 123.456f) out[0] = s;
}

template <int LPI, int UB, bool NT, int NTHR>
__global__ __launch_bounds__(NTHR) void k_cuberow(const float *__restrict__ cube, int L, int B, int C, int b0, int p,
                                                  int lines_per_wg, float *out) {
  // C even; lane t < C/2 owns samples 2t, 2t+1; the workgroup's NTHR/320 line groups each take LPI lines at a time
  const int half = C / 2;
  const int grp = threadIdx.x / 320, t = min((int)threadIdx.x % 320, half - 1);
  constexpr int NG = NTHR / 320;
  const int lbeg = blockIdx.x * lines_per_wg, lend = min(L, lbeg + lines_per_wg);
  float s = 0;
  for (int l = lbeg + grp * LPI; l < lend; l += NG * LPI) {
    for (int bc = 0; bc < p; bc += UB) {
      f2_t v[LPI][UB];
#pragma unroll
      for (int bb = 0; bb < UB; ++bb)
#pragma unroll
        for (int j = 0; j < LPI; ++j)
          v[j][bb] = ld<f2_t, NT>(reinterpret_cast<const f2_t *>(
                                        cube + ((size_t)min(l + j, lend - 1) * B + b0 + min(bc + bb, p - 1)) * C) + t);
#pragma unroll
      for (int bb = 0; bb < UB; ++bb)
#pragma unroll
        for (int j = 0; j < LPI; ++j) s += v[j][bb].x + v[j][bb].y;
    }
  }
  if (s == 123.456f) out[0] = s;
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

int main(int argc, char **argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs %d clock %d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
  const size_t bytes = (size_t)4 << 30;
  float *buf, *out;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&out, 4096));
  CK(hipMemset(buf, 1, bytes));
  const int R = 5;
#define RUN(name, ...)                                                          \
  do {                                                                          \
    float ms = time_ms([&] { __VA_ARGS__; }, R);                                \
    printf("%-64s %7.1f GB/s  (%.3f ms)\n", name, bytes / ms / 1e6, ms);        \
  } while (0)

  // ---- grid-stride, by width / loads in flight / NT -------------------------------------------------
  for (int bpc : {8, 16, 32}) {
    const int blocks = 256 * bpc;
    char nm[128];
    snprintf(nm, sizeof nm, "stride f4 U4  plain  %2d WG/CU", bpc);
    RUN(nm, hipLaunchKernelGGL((k_stride<f4_t, 4, false>), dim3(blocks), dim3(256), 0, 0, (const f4_t *)buf, bytes / 16, out));
    snprintf(nm, sizeof nm, "stride f4 U8  plain  %2d WG/CU", bpc);
    RUN(nm, hipLaunchKernelGGL((k_stride<f4_t, 8, false>), dim3(blocks), dim3(256), 0, 0, (const f4_t *)buf, bytes / 16, out));
    snprintf(nm, sizeof nm, "stride f4 U4  nt     %2d WG/CU", bpc);
    RUN(nm, hipLaunchKernelGGL((k_stride<f4_t, 4, true>), dim3(blocks), dim3(256), 0, 0, (const f4_t *)buf, bytes / 16, out));
    snprintf(nm, sizeof nm, "stride f4 U8  nt     %2d WG/CU", bpc);
    RUN(nm, hipLaunchKernelGGL((k_stride<f4_t, 8, true>), dim3(blocks), dim3(256), 0, 0, (const f4_t *)buf, bytes / 16, out));
    snprintf(nm, sizeof nm, "stride f1 U16 plain  %2d WG/CU", bpc);
    RUN(nm, hipLaunchKernelGGL((k_stride<float, 16, false>), dim3(blocks), dim3(256), 0, 0, (const float *)buf, bytes / 4, out));
    snprintf(nm, sizeof nm, "stride f1 U16 nt     %2d WG/CU", bpc);
    RUN(nm, hipLaunchKernelGGL((k_stride<float, 16, true>), dim3(blocks), dim3(256), 0, 0, (const float *)buf, bytes / 4, out));
    snprintf(nm, sizeof nm, "stride f2 U8  nt     %2d WG/CU", bpc);
    RUN(nm, hipLaunchKernelGGL((k_stride<f2_t, 8, true>), dim3(blocks), dim3(256), 0, 0, (const f2_t *)buf, bytes / 8, out));
  }
  // ---- workgroup-contiguous chunks ------------------------------------------------------------------
  for (size_t chunk_kb : {256, 1024, 4096}) {
    const size_t ce = chunk_kb * 1024 / 16;
    const int blocks = (int)(bytes / (chunk_kb * 1024));
    char nm[128];
    snprintf(nm, sizeof nm, "chunk %4zu KB f4 U8 plain", chunk_kb);
    RUN(nm, hipLaunchKernelGGL((k_chunk<f4_t, 8, false>), dim3(blocks), dim3(256), 0, 0, (const f4_t *)buf, ce, out));
    snprintf(nm, sizeof nm, "chunk %4zu KB f4 U8 nt", chunk_kb);
    RUN(nm, hipLaunchKernelGGL((k_chunk<f4_t, 8, true>), dim3(blocks), dim3(256), 0, 0, (const f4_t *)buf, ce, out));
    snprintf(nm, sizeof nm, "chunk %4zu KB f4 U16 nt", chunk_kb);
    RUN(nm, hipLaunchKernelGGL((k_chunk<f4_t, 16, true>), dim3(blocks), dim3(256), 0, 0, (const f4_t *)buf, ce, out));
    snprintf(nm, sizeof nm, "chunk %4zu KB LDS-DMA U8 plain", chunk_kb);
    RUN(nm, hipLaunchKernelGGL((k_ldsdma<8, false>), dim3(blocks), dim3(256), 4 * 8 * 64 * 16, 0, (const f4_t *)buf, ce, out));
    snprintf(nm, sizeof nm, "chunk %4zu KB LDS-DMA U8 nt", chunk_kb);
    RUN(nm, hipLaunchKernelGGL((k_ldsdma<8, true>), dim3(blocks), dim3(256), 4 * 8 * 64 * 16, 0, (const f4_t *)buf, ce, out));
    snprintf(nm, sizeof nm, "chunk %4zu KB LDS-DMA U16 nt", chunk_kb);
    RUN(nm, hipLaunchKernelGGL((k_ldsdma<16, true>), dim3(blocks), dim3(256), 4 * 16 * 64 * 16, 0, (const f4_t *)buf, ce, out));
  }
#undef RUN
  // ---- the cube's access shape ----------------------------------------------------------------------
  {
    const int L = 4000, B = 425, C = 598, b0 = 350, p = 72;      // 4.07 GB cube; the active window is 0.689 GB
    const size_t cb = (size_t)L * B * C * 4;
    if (cb > bytes) { printf("cube does not fit\n"); return 1; }
    const double act = (double)L * p * C * 4;
#define RUNC(name, ...)                                                         \
  do {                                                                          \
    float ms = time_ms([&] { __VA_ARGS__; }, R);                                \
    printf("%-64s %7.1f GB/s  (%.3f ms)\n", name, act / ms / 1e6, ms);          \
  } while (0)
    const int ncb = (C + 63) / 64;
    for (int lpw : {32, 64}) {
      const int nchunk = (L + lpw - 1) / lpw;
      char nm[128];
      snprintf(nm, sizeof nm, "cube 64-col blocks LPI8 UB4 plain, %d lines/WG", lpw);
      RUNC(nm, hipLaunchKernelGGL((k_cube64<8, 4, false>), dim3(ncb * nchunk), dim3(256), 0, 0, buf, L, B, C, b0, p, lpw, ncb, out));
      snprintf(nm, sizeof nm, "cube 64-col blocks LPI8 UB4 nt,    %d lines/WG", lpw);
      RUNC(nm, hipLaunchKernelGGL((k_cube64<8, 4, true>), dim3(ncb * nchunk), dim3(256), 0, 0, buf, L, B, C, b0, p, lpw, ncb, out));
      snprintf(nm, sizeof nm, "cube 64-col blocks LPI8 UB8 plain, %d lines/WG", lpw);
      RUNC(nm, hipLaunchKernelGGL((k_cube64<8, 8, false>), dim3(ncb * nchunk), dim3(256), 0, 0, buf, L, B, C, b0, p, lpw, ncb, out));
    }
    for (int lpw : {8, 16, 32}) {
      const int nchunk = (L + lpw - 1) / lpw;
      char nm[128];
      snprintf(nm, sizeof nm, "cube rows f2 320thr LPI4 UB4 plain, %d lines/WG", lpw);
      RUNC(nm, hipLaunchKernelGGL((k_cuberow<4, 4, false, 320>), dim3(nchunk), dim3(320), 0, 0, buf, L, B, C, b0, p, lpw, out));
      snprintf(nm, sizeof nm, "cube rows f2 320thr LPI4 UB4 nt,    %d lines/WG", lpw);
      RUNC(nm, hipLaunchKernelGGL((k_cuberow<4, 4, true, 320>), dim3(nchunk), dim3(320), 0, 0, buf, L, B, C, b0, p, lpw, out));
      snprintf(nm, sizeof nm, "cube rows f2 320thr LPI4 UB8 nt,    %d lines/WG", lpw);
      RUNC(nm, hipLaunchKernelGGL((k_cuberow<4, 8, true, 320>), dim3(nchunk), dim3(320), 0, 0, buf, L, B, C, b0, p, lpw, out));
      snprintf(nm, sizeof nm, "cube rows f2 640thr LPI4 UB4 nt,    %d lines/WG", lpw);
      RUNC(nm, hipLaunchKernelGGL((k_cuberow<4, 4, true, 640>), dim3(nchunk), dim3(640), 0, 0, buf, L, B, C, b0, p, lpw, out));
      snprintf(nm, sizeof nm, "cube rows f2 640thr LPI8 UB4 plain, %d lines/WG", lpw);
      RUNC(nm, hipLaunchKernelGGL((k_cuberow<8, 4, false, 640>), dim3(nchunk), dim3(640), 0, 0, buf, L, B, C, b0, p, lpw, out));
    }
#undef RUNC
  }
  return 0;
}
