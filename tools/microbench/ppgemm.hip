// Prototype of the tile scorer's next convolution main loop (round 6): the split-operand product C = A B^T of cnn_split.hip
// (three v_mfma_f32_32x32x16_f16 per fragment pair) as a PING-PONG pipeline -- 256 x 128 tile, eight waves in two groups that
// alternate between the matrix pipe and the memory pipes, operands staged global -> LDS by LDS-DMA (buffer_load ... lds) into a
// three-stage ring, counted vmcnt, raw barriers (cdna_hip_programming.md, "The 256^2 8-phase template").
//   A: [M][K / 8][hi 8 | lo 8] halves (the split format of an activation tensor), B: [N][K / 8][hi 8 | lo 8], C: [M][N] float32
// Build: hipcc --offload-arch=gfx950 -O3 ppgemm.hip -o ppgemm;  run: ./ppgemm [M] [N] [K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int BM = 256, BN = 128, BU = 16, NST = 6, AHEAD = 4;   // a UNIT = 16 k (two 8-channel groups); ring of NST units
constexpr int ROWB = 64;                                    // LDS bytes per tile row and unit: 2 groups of (8 hi | 8 lo) halves
constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
constexpr int BK = BU;
typedef __attribute__((address_space(3))) void *lds_ptr;

__device__ __forceinline__ u4 lds_read(unsigned addr) {
  u4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
__device__ __forceinline__ h8 as_h8(u4 v) { union { u4 u; h8 h; } c; c.u = v; return c.h; }

__global__ __launch_bounds__(512, 2) void k_ppgemm(const _Float16 *__restrict__ A, const _Float16 *__restrict__ B, float *__restrict__ C,
                                                   int M, int N, int K, int mode) {
  extern __shared__ __attribute__((aligned(1024))) char sm[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int grp = w >> 2, wm = (w >> 1) & 1, wn = w & 1;
  const int MT = (M + BM - 1) / BM, NT = (N + BN - 1) / BN, per = (MT + 7) / 8;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int mt = xcd * per + slot / NT, nt_ = slot % NT;
  if (mt >= MT || slot / NT >= per) return;
  const int m0 = mt * BM, n0 = nt_ * BN;
  const int nunit = K / BU;
  const unsigned rowbytes = (unsigned)K * 4;                // hi + lo halves of one row

  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(A), 0, (unsigned)((size_t)M * rowbytes), 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(B), 0, (unsigned)((size_t)N * rowbytes), 0x00020000);
  // staging: an instruction fills sixteen tile rows (lane -> row lane >> 2, 16-byte slot lane & 3 holding piece slot ^ ((row >> 2) & 3))
  unsigned ga[2], gb;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 16 * (w + 8 * i) + (lane >> 2), p = (lane & 3) ^ ((r >> 2) & 3);
    ga[i] = (m0 + r < M) ? (unsigned)(m0 + r) * rowbytes + 16 * p : 0x80000000u;
  }
  {
    const int r = 16 * w + (lane >> 2), p = (lane & 3) ^ ((r >> 2) & 3);
    gb = (n0 + r < N) ? (unsigned)(n0 + r) * rowbytes + 16 * p : 0x80000000u;
  }
  const unsigned smbase = (unsigned)(size_t)(__attribute__((address_space(3))) char *)sm;             // (LDS addresses are 32-bit offsets)
  auto stage = [&](int u) {                                 // a unit's three staging instructions of this wave
    char *st = sm + (u % NST) * STAGE;
    const unsigned so = (unsigned)u * (BU * 4);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(st + 1024 * w), 16, ga[0], so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(st + 1024 * (w + 8)), 16, ga[1], so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(st + A_BYTES + 1024 * w), 16, gb, so, 0, 0);
  };
  // fragment reads: row (lane & 31) of a 32-row block, pieces 2 (lane >> 5) + {0: hi, 1: lo}
  const int x = ((lane & 31) >> 2) & 3;
  unsigned fo[2];
#pragma unroll
  for (int lo = 0; lo < 2; ++lo) fo[lo] = (unsigned)((lane & 31) * ROWB + 16 * ((2 * (lane >> 5) + lo) ^ x));
  const unsigned arow = (unsigned)((128 * grp + 64 * wm) * ROWB), brow = (unsigned)(A_BYTES + 64 * wn * ROWB);

  f16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // prologue: units 0 .. AHEAD - 1 in flight, unit 0 landed; group 1 then falls one barrier behind group 0
#pragma unroll
  for (int u = 0; u < AHEAD; ++u)
    if (u < nunit) stage(u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (prototype: the prologue waits for everything)
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();

  u4 ah[2], al[2], bh[2], bl[2];
  for (int u = 0; u < nunit; ++u) {
    const unsigned sb = smbase + (unsigned)((u % NST) * STAGE);
    if (!(mode & 4) || u == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = lds_read(sb + arow + i * 32 * ROWB + fo[0]);
        al[i] = lds_read(sb + arow + i * 32 * ROWB + fo[1]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh[j] = lds_read(sb + brow + j * 32 * ROWB + fo[0]);
        bl[j] = lds_read(sb + brow + j * 32 * ROWB + fo[1]);
      }
    }
    // unit u + AHEAD goes into the slot unit u + AHEAD - NST left NST - AHEAD phases ago; unit u + 1 must have landed
    if (u + AHEAD < nunit && !(mode & 1)) {
      stage(u + AHEAD);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (AHEAD - 1)) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(al[i]), as_h8(bh[j]), acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(ah[i]), as_h8(bl[j]), acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(ah[i]), as_h8(bh[j]), acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();

#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + 64 * wn + 32 * j + (lane & 31);
    if (n < N)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + 128 * grp + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (m < M) C[(size_t)m * N + n] = acc[i][j][r];
        }
  }
}

// v3: no wave groups -- every wave software-pipelines itself: the fragment reads of unit u + 1 and the staging of unit u + AHEAD are
// issued BETWEEN the matrix instructions of unit u (two fragment register sets), one barrier per unit
template <int X>
__device__ __forceinline__ void pin() { __builtin_amdgcn_sched_barrier(0); }

__global__ __launch_bounds__(512, 2) void k_ppgemm3(const _Float16 *__restrict__ A, const _Float16 *__restrict__ B, float *__restrict__ C,
                                                    int M, int N, int K, int mode) {
  extern __shared__ __attribute__((aligned(1024))) char sm[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int grp = w >> 2, wm = (w >> 1) & 1, wn = w & 1;
  const int MT = (M + BM - 1) / BM, NT = (N + BN - 1) / BN, per = (MT + 7) / 8;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int mt = xcd * per + slot / NT, nt_ = slot % NT;
  if (mt >= MT || slot / NT >= per) return;
  const int m0 = (mode & 32) ? 0 : mt * BM, n0 = nt_ * BN;
  const int nunit = K / BU;
  const unsigned rowbytes = (unsigned)K * 4;
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(A), 0, (unsigned)((size_t)M * rowbytes), 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(B), 0, (unsigned)((size_t)N * rowbytes), 0x00020000);
  unsigned ga[2], gb;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 16 * (w + 8 * i) + (lane >> 2), p = (lane & 3) ^ ((r >> 2) & 3);
    ga[i] = (m0 + r < M) ? (unsigned)(m0 + r) * rowbytes + 16 * p : 0x80000000u;
  }
  {
    const int r = 16 * w + (lane >> 2), p = (lane & 3) ^ ((r >> 2) & 3);
    gb = (n0 + r < N) ? (unsigned)(n0 + r) * rowbytes + 16 * p : 0x80000000u;
  }
  const unsigned smbase = (unsigned)(size_t)(__attribute__((address_space(3))) char *)sm;
  const int x = ((lane & 31) >> 2) & 3;
  unsigned fo[2];
#pragma unroll
  for (int lo = 0; lo < 2; ++lo) fo[lo] = (unsigned)((lane & 31) * ROWB + 16 * ((2 * (lane >> 5) + lo) ^ x));
  const unsigned arow = (unsigned)((128 * grp + 64 * wm) * ROWB), brow = (unsigned)(A_BYTES + 64 * wn * ROWB);
  f16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto stage1 = [&](int u, int which) {
    char *st = sm + (u % NST) * STAGE;
    const unsigned so = (unsigned)u * (BU * 4);
    if (which == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(st + 1024 * w), 16, ga[0], so, 0, 0);
    if (which == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(st + 1024 * (w + 8)), 16, ga[1], so, 0, 0);
    if (which == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(st + A_BYTES + 1024 * w), 16, gb, so, 0, 0);
  };
#pragma unroll
  for (int u = 0; u < AHEAD; ++u)
    if (u < nunit) { stage1(u, 0); stage1(u, 1); stage1(u, 2); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  u4 F[2][8];                                               // [set][ah0 al0 ah1 al1 bh0 bl0 bh1 bl1]
  {
#pragma unroll
    for (int q = 0; q < 8; ++q)
      F[0][q] = lds_read(smbase + (q < 4 ? arow : brow) + ((q >> 1) & 1) * 32 * ROWB + fo[q & 1]);
  }
  // one unit: the 12 matrix instructions on set S with the next unit's 8 fragment reads (into set S ^ 1) and 3 staging instructions between them
#define PP_UNIT(S, u)                                                                                                              \
  {                                                                                                                                \
    const bool more = (u) + 1 < nunit, st_ = (u) + AHEAD < nunit && !(mode & 1);                                                   \
    const unsigned sbn = smbase + (unsigned)((((u) + 1) % NST) * STAGE);                                                           \
    if (st_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (AHEAD - 2)) : "memory");                                                \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                             \
    __builtin_amdgcn_s_barrier();                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                             \
    _Pragma("unroll") for (int t = 0; t < 12; ++t) {                                                                               \
      const int term = t >> 2, i = (t >> 1) & 1, j = t & 1;                                                                        \
      const u4 a_ = (term == 0) ? F[S][2 * i + 1] : F[S][2 * i];                                                                   \
      const u4 b_ = (term == 1) ? F[S][4 + 2 * j + 1] : F[S][4 + 2 * j];                                                           \
      if (!(mode & 16)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(a_), as_h8(b_), acc[i][j], 0, 0, 0);              \
      __builtin_amdgcn_sched_barrier(0);                                                                                           \
      if (t < 8) {                                                                                                                 \
        if (more && (!(mode & 4)))                                                                                                 \
          F[S ^ 1][t] = lds_read(sbn + (t < 4 ? arow : brow) + ((t >> 1) & 1) * 32 * ROWB + fo[t & 1]);                            \
      } else if (t < 11) {                                                                                                         \
        if (st_) stage1((u) + AHEAD, t - 8);                                                                                       \
      }                                                                                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                                           \
    }                                                                                                                              \
  }
  int u = 0;
  for (; u + 1 < nunit; u += 2) {
    PP_UNIT(0, u)
    PP_UNIT(1, u + 1)
  }
  if (u < nunit) PP_UNIT(0, u)
#undef PP_UNIT
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + 64 * wn + 32 * j + (lane & 31);
    if (n < N)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + 128 * grp + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (m < M) C[(size_t)m * N + n] = acc[i][j][r];
        }
  }
}

// reference: the same three products in double, a few rows
static double ref_dot(const std::vector<_Float16> &A, const std::vector<_Float16> &B, int K, int m, int n) {
  double s = 0;
  for (int g = 0; g < K / 8; ++g)
    for (int k = 0; k < 8; ++k) {
      const double ah = (double)(float)A[((size_t)m * (K / 8) + g) * 16 + k], al = (double)(float)A[((size_t)m * (K / 8) + g) * 16 + 8 + k];
      const double bh = (double)(float)B[((size_t)n * (K / 8) + g) * 16 + k], bl = (double)(float)B[((size_t)n * (K / 8) + g) * 16 + 8 + k];
      s += al * bh + ah * bl + ah * bh;
    }
  return s;
}

int main(int argc, char **argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 131072, N = argc > 2 ? atoi(argv[2]) : 256, K = argc > 3 ? atoi(argv[3]) : 1024;
  const int mode = argc > 4 ? atoi(argv[4]) : 0;
  if (K % BK) { printf("K must be a multiple of %d\n", BK); return 1; }
  std::vector<_Float16> hA((size_t)M * K * 2), hB((size_t)N * K * 2);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (size_t i = 0; i < hA.size() / 16; ++i)
    for (int k = 0; k < 8; ++k) {
      const float v = rnd() * 8.f;
      const _Float16 h = (_Float16)v;
      hA[i * 16 + k] = h;
      hA[i * 16 + 8 + k] = (_Float16)(v - (float)h);
    }
  for (size_t i = 0; i < hB.size() / 16; ++i)
    for (int k = 0; k < 8; ++k) {
      const float v = rnd();
      const _Float16 h = (_Float16)v;
      hB[i * 16 + k] = h;
      hB[i * 16 + 8 + k] = (_Float16)(v - (float)h);
    }
  _Float16 *dA, *dB;
  float *dC;
  CK(hipMalloc(&dA, hA.size() * 2));
  CK(hipMalloc(&dB, hB.size() * 2));
  CK(hipMalloc(&dC, (size_t)M * N * 4));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
  const int MT = (M + BM - 1) / BM, NT = (N + BN - 1) / BN;
  dim3 grid(8 * ((MT + 7) / 8) * NT);
  const int lds = NST * STAGE;
  CK(hipFuncSetAttribute((const void *)k_ppgemm, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute((const void *)k_ppgemm3, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  auto kern = (mode & 8) ? k_ppgemm3 : k_ppgemm;
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, 0, dA, dB, dC, M, N, K, mode);
  CK(hipDeviceSynchronize());
  std::vector<float> hC((size_t)M * N);
  CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  int bad = 0;
  for (int t = 0; t < 4000; ++t) {
    s = s * 1664525u + 1013904223u;
    const int m = (t < 600) ? (t % 300) + (t < 300 ? 0 : M - 300) : (int)((s >> 4) % (unsigned)M);
    s = s * 1664525u + 1013904223u;
    const int n = (int)((s >> 4) % (unsigned)N);
    const double r = ref_dot(hA, hB, K, m, n), g = hC[(size_t)m * N + n];
    const double e = fabs(g - r) / (fabs(r) + 1.0);
    if (!(e < 4e-5)) { if (bad < 5) printf("mismatch m=%d n=%d got %g want %g\n", m, n, g, r); ++bad; }
    if (e > worst) worst = e;
  }
  printf("check: %d bad of 4000, worst rel %.2e\n", bad, worst);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    const int it = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, grid, dim3(512), lds, 0, dA, dB, dC, M, N, K, mode);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / it, tf = 3.0 * 2.0 * M * (double)N * K / (us * 1e-6) / 1e12;
    printf("mode %d M %d N %d K %d: %.1f us, %.0f TFLOP/s of fp16 products (%.3f of 2500)\n", mode, M, N, K, us, tf, tf / 2500.0);
  }
  return bad ? 2 : 0;
}
