// Which access geometry streams the active window of the BIL cube fastest?  Loads only (the sum keeps them alive), on
// the benchmark's own shape: [20000][425][598] float32, bands 350..421.  A workgroup of 4 waves takes a column block of
// 64*VW samples (lane = VW adjacent samples: 4 / 8 / 16-byte loads) and a chunk of lines; a wave keeps LPI lines x UB
// bands x 2 batches of loads in flight.  Block order natural or XCD-aware (all column blocks of a line chunk on one XCD).
// Build: hipcc --offload-arch=gfx950 -O3 readbw2.hip -o readbw2 ; run: ./readbw2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef float f2u_t __attribute__((ext_vector_type(2), aligned(4)));
typedef float f4u_t __attribute__((ext_vector_type(4), aligned(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)

template <int VW> struct V;
template <> struct V<1> { typedef float T; typedef float U; };
template <> struct V<2> { typedef f2_t T; typedef f2u_t U; };
template <> struct V<4> { typedef f4_t T; typedef f4u_t U; };
__device__ __forceinline__ float first(float v) { return v; }
__device__ __forceinline__ float first(f2_t v) { return v.x + v.y; }
__device__ __forceinline__ float first(f4_t v) { return v.x + v.w; }

template <int VW, int LPI, int UB, bool NT>
__global__ __launch_bounds__(256) void k_blk(const float *__restrict__ cube, int L, int B, int C, int b0, int p,
                                             int lines_per_wg, int ncb, int nchunk, int xcdmap, float *out) {
  typedef typename V<VW>::T T;
  typedef typename V<VW>::U U;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int cbi, chunk;
  if (xcdmap) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    chunk = (slot / ncb) * 8 + xcd;
    cbi = slot % ncb;
    if (chunk >= nchunk) return;
  } else {
    cbi = blockIdx.x % ncb;
    chunk = blockIdx.x / ncb;
  }
  const int col = min((cbi * 64 + lane) * VW, C - VW);
  const int lbeg = chunk * lines_per_wg, lend = min(L, lbeg + lines_per_wg);
  float s = 0;
  for (int l = lbeg + wave * LPI; l < lend; l += 4 * LPI) {
    T va[LPI][UB], vb[LPI][UB];
    auto load = [&](T (&v)[LPI][UB], int bc) {
#pragma unroll
      for (int bb = 0; bb < UB; ++bb)
#pragma unroll
        for (int j = 0; j < LPI; ++j) {
          const U *q = reinterpret_cast<const U *>(cube + ((size_t)min(l + j, lend - 1) * B + b0 + min(bc + bb, p - 1)) * C + col);
          v[j][bb] = NT ? __builtin_nontemporal_load(q) : *q;
        }
    };
    auto use = [&](T (&v)[LPI][UB]) {
#pragma unroll
      for (int bb = 0; bb < UB; ++bb)
#pragma unroll
        for (int j = 0; j < LPI; ++j) s += first(v[j][bb]);
    };
    load(va, 0);
    for (int bc = 0; bc < p; bc += 2 * UB) {
      load(vb, bc + UB);
      use(va);
      if (bc + 2 * UB < p) load(va, bc + 2 * UB);
      use(vb);
    }
  }
  if (s == 123.456f) out[0] = s;
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
  const int L = 20000, B = 425, C = 598, b0 = 350, p = 72;
  const size_t bytes = (size_t)L * B * C * 4;
  float *cube, *out;
  CK(hipMalloc(&cube, bytes));
  CK(hipMalloc(&out, 4096));
  CK(hipMemset(cube, 1, bytes));
  const double act = (double)L * p * C * 4;
  const int R = 3;
#define RUN(VW, LPI, UB, NT, LPW, XCD)                                                                              \
  do {                                                                                                              \
    const int ncb = (C + 64 * VW - 1) / (64 * VW), nchunk = (L + LPW - 1) / LPW;                                    \
    const int nblk = XCD ? (nchunk + 7) / 8 * 8 * ncb : ncb * nchunk;                                               \
    float ms = time_ms([&] { hipLaunchKernelGGL((k_blk<VW, LPI, UB, NT>), dim3(nblk), dim3(256), 0, 0, cube, L, B, \
                                                C, b0, p, LPW, ncb, nchunk, XCD, out); }, R);                      \
    printf("VW %d (%3d-col blocks) LPI %d UB %d %s lines/WG %3d xcd %d : %7.1f GB/s (%.3f ms)\n", VW, 64 * VW, LPI, \
           UB, NT ? "nt   " : "plain", LPW, XCD, act / ms / 1e6, ms);                                               \
  } while (0)
  RUN(1, 8, 4, false, 32, 1);
  RUN(1, 8, 4, false, 32, 0);
  RUN(1, 8, 4, true, 32, 1);
  RUN(1, 8, 8, false, 32, 1);
  RUN(1, 8, 4, false, 64, 1);
  RUN(2, 8, 4, false, 32, 1);
  RUN(2, 8, 4, true, 32, 1);
  RUN(2, 8, 4, false, 32, 0);
  RUN(2, 8, 4, true, 32, 0);
  RUN(2, 4, 4, true, 16, 1);
  RUN(2, 4, 8, true, 16, 1);
  RUN(2, 4, 8, true, 32, 1);
  RUN(2, 8, 4, true, 64, 1);
  RUN(2, 2, 8, true, 8, 1);
  RUN(4, 8, 4, false, 32, 1);
  RUN(4, 8, 4, true, 32, 1);
  RUN(4, 4, 4, true, 16, 1);
  RUN(4, 4, 4, true, 32, 1);
  RUN(4, 4, 8, true, 32, 1);
  RUN(4, 2, 8, true, 8, 1);
  RUN(4, 2, 8, true, 16, 1);
  RUN(4, 8, 4, true, 32, 0);
  RUN(4, 4, 4, false, 16, 1);
  return 0;
}
