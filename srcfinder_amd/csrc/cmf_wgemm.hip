// Wide-window statistics, round 4: the two contractions on v_mfma_f64_4x4x4_f64, fused with what surrounds them.
//
// Round 3 ran the wide windows (more than 96 bands: the reference's reflectance window 5..420, cmf/robust_mf.py:186-187;
// the full-band stress shape) as separate kernels around a generic 16x16x4 GEMM (cmf_wide.hip: k_center -> X~ in float64,
// k_dgemm covariance, k_dgemm Y^2, k_dgemm r, k_nllrows): 166 MB of float64 scratch per column (X~, Z, r), so 36 columns
// per batch, every operand round-tripping HBM, and an instruction (16x16x4) that peaks at 47 TFLOP/s on MI355X where the
// 4x4x4 form reaches 71 (tools/microbench/mfma64.hip).  Here:
//
//   k_wsyrk   S = X~^T X~ / (n - 1)  (numpy.cov as called by looshrinkage, robust_mf.py:52-70, :98, :130) straight from
//             the extracted float32 rows: centring, validity mask and promotion happen on the way into LDS (interleaved by
//             hand with the MFMA groups), 96 x 96 band tiles (upper triangle of tiles), four waves of 48 x 48, K = all
//             rows of the column.
//   k_wsweep  the LOO sweep of robust_mf.py:105-117 in its eigen form: per 64-row tile
//                 Y = X~ W   (W = D^-1 V)      -> registers (each wave a quarter of the columns of Y)
//                 Z = Y.^2                      -> the same registers, already in the A-operand layout of the next product
//                 r = Z C    (C[j][a] = 1 / (n beta_a lam_j + alpha_a))   four alphas at a time, summed over the waves
//                 sum_k log(1 - beta r), sum_k r / (1 - beta r)            -> per (column, row split) partials for k_nll
//             W and C stream through LDS in 16-band / 8-alpha chunks (L2-resident: 1.5 + 0.7 MB per column); nothing
//             but the 2 x 208 partial sums per (column, split) is written.  No float64 scratch per column any more.
//   k_wsweep8 the same sweep with eight waves (two per SIMD), wave-private operand slices and no workgroup barrier in the
//             first product: the form that runs for float32 rows and 256 < p <= 432 (the reflectance window, the full-band
//             shape); 483 k -> 367-380 k cycles per 64-row tile.
//
// MFMA operand roles (validated by cmf_cov4.hip / cmf_wjac.hip): D_m[i][j] += sum_k A_m[i][k] B_m[k][j] for the four
// blocks m, lane = 16 q + 4 m + n, A[i][k] at (q = k, n = i), B[k][j] at (q = k, n = j), D[i][j] at (q = i, n = j).
#include "cmf_common.h"

namespace {

__device__ unsigned long long g_ws_stamps[8];   // phase clocks (sf_debug_wsweep_stamps)

// --------------------------------------------------------------------------------------------------------------------
// k_wsyrk
// --------------------------------------------------------------------------------------------------------------------
constexpr int SY_KC = 16;              // rows per LDS chunk
#ifndef SY_STAMPS
#define SY_STAMPS 0
#endif

__device__ __forceinline__ void sy_load4(const float *p, float (&v)[4]) {
  const float4 f = *reinterpret_cast<const float4 *>(p);
  v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
}
__device__ __forceinline__ void sy_load4(const double *p, double (&v)[4]) {
  const double2 a = *reinterpret_cast<const double2 *>(p), b = *reinterpret_cast<const double2 *>(p + 2);
  v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}

// TI: 16-band groups per wave along I (the wave tile is 16 TI x 16 TI bands, the workgroup tile T = 32 TI: 2 x 2 waves).
// TI = 4: 128-band tiles, 64 accumulators per lane, two workgroups per CU.  TI = 3: 96-band tiles -- 15 tile pairs of 96^2
// instead of 10 of 128^2 at p = 425 (16 % fewer MFMAs: less of the upper triangle's padding), 36 accumulators, three per CU.
template <typename XT, int TI, int OCC>
__global__ __launch_bounds__(256, OCC) void k_wsyrk(const XT *__restrict__ xt, const uint8_t *__restrict__ mask_t,
                                                     const int32_t *__restrict__ nuse, const double *__restrict__ mu, int L, int p,
                                                     int ps, int ntile, double *__restrict__ cov) {
  constexpr int TJ = 4 * TI;           // 4-band groups per wave along J
  constexpr bool SY_IL = true;         // the loader's store half interleaved with the MFMA groups
  constexpr int T = 32 * TI;           // band tile
  constexpr int LD = 2 * T + 16;       // doubles per chunk row: [I bands | J bands] + pad (= 16 mod 32)
  constexpr int NQ = 2 * T / 4;        // 4-band quads per chunk row
  constexpr int NIT = (SY_KC * NQ + 255) / 256;   // loader items per thread
  extern __shared__ __attribute__((aligned(16))) double sm[];   // [2][SY_KC][LD], then mu [2 T], zeros [2 T]
  double *mus = sm + 2 * SY_KC * LD;
  const int c = blockIdx.y;
  int ti = 0, tj = 0;
  {   // tile pair blockIdx.x of the upper triangle, row-major
    int rem = blockIdx.x, rowlen = ntile;
    while (rem >= rowlen) { rem -= rowlen; ++ti; --rowlen; }
    tj = ti + rem;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, m = (lane >> 2) & 3, n = lane & 3;
  const int wi = wave >> 1, wj = wave & 1;
  for (int i = tid; i < 2 * T; i += 256) {
    const int b = (i < T ? ti * T : tj * T - T) + i;
    mus[i] = (b < p) ? mu[(size_t)c * p + b] : 0.0;
  }
  const XT *xc = xt + (size_t)c * L * ps;
  const uint8_t *mp = mask_t + (size_t)c * L;
  // loader: item = (row of the chunk, 4-band quad of [I half | J half]); consecutive lanes = consecutive quads of a row.
  // Everything that does not change from chunk to chunk is worked out once: the item's source offset, its LDS slot, and the
  // number of its bands inside the window -- a chunk then costs a mask, one conversion and one subtraction per
  // element (the SIMD's vector instructions take matrix time: profiles/r04_pmc_wide_window.txt).  An invalid row reads its
  // means from a zero table and has its bits masked off: exactly 0 whatever the row held.
  double *zeros = mus + 2 * T;           // [2 T] zeros
  for (int i = tid; i < 2 * T; i += 256) zeros[i] = 0.0;
  int irow[NIT], ioff[NIT], lslot[NIT], nv[NIT];   // nv: bands of the quad inside the window (4 except at the window's end)
#pragma unroll
  for (int u = 0; u < NIT; ++u) {
    const int item = tid + 256 * u;
    const int r = item / NQ, qd = item - r * NQ;
    const int gb = ((qd < NQ / 2) ? ti * T : tj * T - T) + 4 * qd;
    irow[u] = (item < SY_KC * NQ) ? r : -1;
    ioff[u] = min(gb, ps - 4);                       // (a quad past the padded row: any in-bounds address, masked off below)
    lslot[u] = r * LD + 4 * qd;
    nv[u] = min(max(p - gb, 0), 4);
  }
  unsigned voff[NIT];   // the item's byte offset from the chunk's first row (scalar base + this: no address arithmetic per chunk)
#pragma unroll
  for (int u = 0; u < NIT; ++u) voff[u] = (unsigned)((max(irow[u], 0) * ps + ioff[u]) * (int)sizeof(XT));
  XT pre[NIT][4];   // raw values of the next chunk (promoted when they are stored)
  unsigned char pmk[NIT];   // the rows' validity bytes, as loaded: NOTHING in a fetch depends on loaded data (a compare here made
  bool pin[NIT];            // every item wait for all the loads in flight -- vmcnt(0) four times per chunk)
  auto gload = [&](int r0) {
    if (r0 + SY_KC <= L) {   // (uniform) a whole chunk: scalar base of the chunk + the item's constant offset
      const char *xb = reinterpret_cast<const char *>(xc + (size_t)r0 * ps);
      const uint8_t *mb = mp + r0;
#pragma unroll
      for (int u = 0; u < NIT; ++u) {
        pin[u] = irow[u] >= 0;
        pmk[u] = mb[max(irow[u], 0)];
        sy_load4(reinterpret_cast<const XT *>(xb + voff[u]), pre[u]);
      }
    } else {
#pragma unroll
      for (int u = 0; u < NIT; ++u) {
        const int row = r0 + max(irow[u], 0);
        const int rr = row < L ? row : L - 1;
        pin[u] = (irow[u] >= 0) && (row < L);
        pmk[u] = mp[rr];
        sy_load4(xc + (size_t)rr * ps + ioff[u], pre[u]);
      }
    }
  };
  auto lstore_item = [&](int buf, int u) {
    {
      if (irow[u] < 0) return;
      const bool ok = pin[u] && pmk[u] != 0;
      double *dst = sm + (size_t)buf * SY_KC * LD + lslot[u];
      const double *mq = (ok ? mus : zeros) + (lslot[u] - irow[u] * LD);
      const int nvu = ok ? nv[u] : 0;
      double o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned mk = (e < nvu) ? 0xffffffffu : 0u;
        if constexpr (sizeof(XT) == 4) {
          o[e] = (double)__uint_as_float(__float_as_uint(pre[u][e]) & mk) - mq[e];
        } else {
          o[e] = (mk ? pre[u][e] : 0.0) - mq[e];
        }
      }
      *reinterpret_cast<double2 *>(dst) = make_double2(o[0], o[1]);
      *reinterpret_cast<double2 *>(dst + 2) = make_double2(o[2], o[3]);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int u = 0; u < NIT; ++u) lstore_item(buf, u);
  };
  double acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = 0.0;
  __syncthreads();   // mus
  gload(0);
  lstore(0);
  const int nchunk = (L + SY_KC - 1) / SY_KC;
  const bool stw = SY_STAMPS && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
  unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, d0 = 0, d1 = 0, d2 = 0, d3 = 0;
  for (int ch = 0; ch < nchunk; ++ch) {
    const int buf = ch & 1;
    if (stw) s0 = __builtin_readcyclecounter();
    if (ch + 1 < nchunk) gload((ch + 1) * SY_KC);
    if (stw) s1 = __builtin_readcyclecounter();
    __syncthreads();   // chunk ch is in sm[buf]; everybody is done with sm[buf ^ 1] (chunk ch - 1)
    if (stw) s2 = __builtin_readcyclecounter();
    const double *xs = sm + (size_t)buf * SY_KC * LD;
    // operands of MFMA step k4 + 1 are fetched before the MFMAs of step k4 (the pinned order below would otherwise expose the LDS
    // latency four times a chunk)
    double a[2][TI], b[2][TJ];
    auto opload = [&](int k4, int slot) {
      const double *row = xs + (size_t)(4 * k4 + q) * LD;
#pragma unroll
      for (int I = 0; I < TI; ++I) a[slot][I] = row[16 * TI * wi + 16 * I + 4 * m + n];
#pragma unroll
      for (int J = 0; J < TJ; ++J) b[slot][J] = row[T + 16 * TI * wj + 4 * J + n];
    };
    opload(0, 0);
#pragma unroll
    for (int k4 = 0; k4 < SY_KC / 4; ++k4) {
      if (k4 + 1 < SY_KC / 4) opload(k4 + 1, (k4 + 1) & 1);
      if (SY_IL) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int I = 0; I < TI; ++I)
#pragma unroll
        for (int J = 0; J < TJ; ++J) acc[I][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[k4 & 1][I], b[k4 & 1][J], acc[I][J], 0, 0, 0);
      // The next chunk's rows are promoted and stored (into the other buffer: nobody reads it before the next barrier) BETWEEN
      // the MFMA groups of this chunk, an item a group: a vector instruction of a wave that is NOT streaming MFMAs waits for a gap
      // in the other waves' MFMA streams (~40 cycles each; the store phase alone was 4.3 k cycles of a 13.8 k-cycle chunk, the
      // load phase's address arithmetic 4.7 k: sf_debug_wsweep_stamps with -DSY_STAMPS=1), inside the stream it costs ~6.
      if (SY_IL && ch + 1 < nchunk) {
        __builtin_amdgcn_sched_barrier(0);
        if (k4 >= 1 && k4 - 1 < NIT) lstore_item(buf ^ 1, k4 - 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (stw) { asm volatile("" : "+v"(acc[0][0])); s3 = __builtin_readcyclecounter(); }
    if (ch + 1 < nchunk) {
      if (!SY_IL) lstore(buf ^ 1);
      else {
#pragma unroll
        for (int u = SY_KC / 4 - 1; u < NIT; ++u) lstore_item(buf ^ 1, u);
      }
    }
    if (stw) { const unsigned long long s4 = __builtin_readcyclecounter(); d0 += s1 - s0; d1 += s2 - s1; d2 += s3 - s2; d3 += s4 - s3; }
  }
  if (stw) { g_ws_stamps[3] = d0; g_ws_stamps[4] = d1; g_ws_stamps[5] = d2; g_ws_stamps[6] = d3; g_ws_stamps[7] = nchunk; }
  const double inv = 1.0 / ((double)nuse[c] - 1.0);
  double *co = cov + (size_t)c * p * p;
#pragma unroll
  for (int I = 0; I < TI; ++I)
#pragma unroll
    for (int J = 0; J < TJ; ++J) {
      const int bi = ti * T + 16 * TI * wi + 16 * I + 4 * m + q, bj = tj * T + 16 * TI * wj + 4 * J + n;
      if (bi < p && bj < p) {
        const double v = acc[I][J] * inv;
        co[(size_t)bi * p + bj] = v;
        if (ti != tj) co[(size_t)bj * p + bi] = v;
      }
    }
}

// --------------------------------------------------------------------------------------------------------------------
// operands of the sweep: W zero-padded to [P16][LDW] (row-major) and C in chunks of 8 alphas, [NCC][LDW][8]: the byte
// images the sweep's LDS buffers hold (global_load_lds copies are lane-linear: LDS image = global image)
// --------------------------------------------------------------------------------------------------------------------
__global__ void k_wmat_p(const double *__restrict__ evec, const double *__restrict__ d, int p, int P16, int LDW,
                         double *__restrict__ W) {
  const int c = blockIdx.y;
  const double *ev = evec + (size_t)c * p * p, *dd = d + (size_t)c * p;
  double *o = W + (size_t)c * P16 * LDW;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P16 * LDW; i += gridDim.x * blockDim.x) {
    const int b = i / LDW, j = i - b * LDW;
    o[i] = (b < p && j < p) ? ev[(size_t)j * p + b] / dd[b] : 0.0;
  }
}
__global__ void k_cmat_t(const double *__restrict__ lam, const int32_t *__restrict__ nloo, const int32_t *__restrict__ status,
                         const double *__restrict__ alphas, int nalpha, int NCC, int p, int LDW, double *__restrict__ Ct) {
  const int c = blockIdx.y;
  const double nn = (double)nloo[c];
  const bool ok = status[c] == 0;
  const double *lc = lam + (size_t)c * p;
  double *o = Ct + (size_t)c * NCC * LDW * 8;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < NCC * LDW * 8; i += gridDim.x * blockDim.x) {
    const int ch = i / (LDW * 8), r = i - ch * (LDW * 8), j = r >> 3, a = 8 * ch + (r & 7);
    double v = 0.0;
    if (ok && a < nalpha && j < p) {
      const double al = alphas[a];
      const double beta = (1.0 - al) / (nn - 1.0);
      v = 1.0 / ((nn * beta) * lc[j] + al);
    }
    o[i] = v;
  }
}

// --------------------------------------------------------------------------------------------------------------------
// k_wsweep
// --------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double ws_cvt(float x) { return (double)x; }
__device__ __forceinline__ double ws_cvt(double x) { return x; }
typedef __attribute__((address_space(3))) void ws_lds_void;
typedef const __attribute__((address_space(1))) void ws_glb_void;

template <int CTRL>
__device__ __forceinline__ double ws_dpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

// phase clocks of the sweep's tiles (sf_debug_set(22, 1); sf_debug_wsweep_stamps): [0] tiles, [1] Y = X~ W, [2] r = Z C + rows
// (declared near the top of the file) g_ws_stamps: k_wsweep8 also: [3] the r phase's MFMAs, [4] row reductions, [5] first barrier, [6] exchange + second barrier

constexpr int WS_CA = 8;    // alphas per C chunk (two groups of four: their row reductions are interleaved)

// XT: float32 (the extracted cube) or float64 (the function-level looshrinkage()).  NI: 16-row groups per tile.
// NJW: 4-column groups of Y per wave (LDW = 16 NJW >= P16).  One chunk = 16 rows of W (16 x LDW doubles, 55 KB at LDW = 432)
// or 8 alphas of C ([LDW][8], half that), copied global -> LDS by global_load_lds (no registers), two buffers: the copy of chunk
// s + 1 is issued right after the barrier that publishes chunk s and has that chunk's 432 MFMAs per wave to land.  X is
// streamed beside W (16 bands x the tile's rows), centred and promoted when it is read as an operand.
// CH: bands per W chunk (16: 55 KB chunks, one workgroup per CU; 8: half that, two workgroups per CU -- or one beside a
// workgroup of another kernel, e.g. the eigensolver of another column group).  OCC: workgroups per CU the registers allow.
// NW: waves per workgroup.  4: one wave per SIMD, a quarter of Y's columns each (NJW 4-column groups).  8: two waves per SIMD
// (<= 256 registers each), an eighth each -- LDW / 4 groups dealt NJW to waves 0-3 and NJL = LDW / 16 - NJW to waves 4-7 (14 + 13
// of the 108 groups at LDW = 432), so that one wave's operand reads, barriers and row reductions run under the other's MFMAs.
template <typename XT, int NI, int NJW, int CH, int OCC, int NW = 4, int LDWT = 16 * NJW>
__global__ __launch_bounds__(64 * NW, OCC) void k_wsweep(const XT *__restrict__ xt, const uint8_t *__restrict__ mask_t,
                                                 const int32_t *__restrict__ nloo, const double *__restrict__ mu,
                                                 const double *__restrict__ Wp, const double *__restrict__ Ct,
                                                 const int32_t *__restrict__ status, const double *__restrict__ alphas, int nalpha,
                                                 int NA, int L, int p, int ps, int P16, int rows_per_wg, int nsplit, int ncols,
                                                 double *__restrict__ part, int stamp) {
  constexpr int RT = 16 * NI;        // rows per tile
  constexpr int LDW = LDWT;
  constexpr int NJL = NW == 4 ? NJW : LDW / 16 - NJW;   // groups of waves 4-7
  constexpr int NT = 64 * NW;
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  static_assert((NW == 4 ? 4 * NJW : 4 * NJW + 4 * NJL) == LDW / 4 && NJL <= NJW && NJL > 0, "column groups");
  constexpr int WS_CH = CH;
  constexpr int CHD = (CH > WS_CA ? CH : WS_CA) * LDW;   // doubles per chunk buffer
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  // ---- LDS carve (ws_lds_bytes below)
  double *Bs = reinterpret_cast<double *>(smraw);                 // [2][CHD]
  double *mus = Bs + 2 * CHD;                                      // [P16]
  double *red = mus + P16;                                         // [2][NW][NI][64]
  double *Pacc = red + 2 * NW * NI * 64;                           // [NA]
  double *Sacc = Pacc + NA;                                        // [NA]
  double *betas = Sacc + NA;                                       // [NA]
  XT *Xs = reinterpret_cast<XT *>(betas + NA);                     // [2][WS_CH][RT]
  int *Eacc = reinterpret_cast<int *>(Xs + 2 * WS_CH * RT);        // [NA]
  int *Nacc = Eacc + NA;                                           // [NA]
  int *rowok = Nacc + NA;                                          // [RT]

  // XCD-aware order: the splits of a column are adjacent on ONE XCD (its L2 holds the column's W and C)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int c = 8 * (slot / nsplit) + xcd, split = slot % nsplit;
  if (c >= ncols) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, m = (lane >> 2) & 3, n = lane & 3;
  double *po = part + ((size_t)c * nsplit + split) * 2 * NA;
  if (status[c] != 0) {
    for (int i = tid; i < 2 * NA; i += NT) po[i] = 0.0;
    return;
  }
  const double nn = (double)nloo[c];
  for (int i = tid; i < NA; i += NT) {
    Pacc[i] = 1.0;
    Sacc[i] = 0.0;
    Eacc[i] = 0;
    Nacc[i] = 0;
    betas[i] = (i < nalpha) ? (1.0 - alphas[i]) / (nn - 1.0) : 0.0;
  }
  for (int i = tid; i < P16; i += NT) mus[i] = (i < p) ? mu[(size_t)c * p + i] : 0.0;
  const int rbeg = split * rows_per_wg, rend = min(L, rbeg + rows_per_wg);
  const XT *xc = xt + (size_t)c * L * ps;
  const uint8_t *mp = mask_t + (size_t)c * L;
  const char *Wc = reinterpret_cast<const char *>(Wp + (size_t)c * P16 * LDW);
  const int NKC = P16 / WS_CH, NCC = NA / WS_CA, NS = NKC + NCC;
  const char *Cc = reinterpret_cast<const char *>(Ct + (size_t)c * NCC * LDW * WS_CA);
  const int j0w = 4 * (wave < 4 ? NJW * wave : 4 * NJW + NJL * (wave - 4));   // first Y column of this wave
  const int njw = wave < 4 ? NJW : NJL;                                        // its 4-column groups (wave-uniform)
  constexpr int WCD = CH * LDW;                // doubles per W chunk
  constexpr int CCD = WS_CA * LDW;             // doubles per C chunk
  constexpr int NPIECE = WCD * 8 / 1024, NPIECE_C = CCD * 8 / 1024;   // 1 KB wave-pieces (LDW is a multiple of 16)
  // chunk s of the stream -> Bs[s & 1]: s < NKC: rows 16 s .. of W; else alpha chunk s - NKC of C.  Lane-linear copies.
  // (tried: the pieces of chunk s + 1 issued one at a time between the MFMAs of chunk s instead of in a row after the barrier --
  //  the Y phase went from 292 k to 427 k cycles per tile: an LDS-DMA between ds_reads and MFMAs stalls the stream)
  auto glds = [&](int s) {
    const char *src = (s < NKC ? Wc + (size_t)s * WCD * 8 : Cc + (size_t)(s - NKC) * CCD * 8) + lane * 16;
    char *dst = reinterpret_cast<char *>(Bs + (size_t)(s & 1) * CHD);
    const int npc = s < NKC ? NPIECE : NPIECE_C;
    for (int pc = wave; pc < npc; pc += NW)
      __builtin_amdgcn_global_load_lds((ws_glb_void *)(src + (size_t)pc * 1024), (ws_lds_void *)(dst + (size_t)pc * 1024), 16, 0, 0);
  };
  // the X chunk of W chunk s: lane item = (row, band quad) of [RT rows][4 quads]
  const int xrow = tid % RT, xq = tid / RT;
  const bool xact = tid < (CH / 4) * RT;
  XT xv[4];
  auto xload = [&](int s, int r0) {
    const int b = WS_CH * s + 4 * xq;
    if (xact && b + 3 < ps) {
      const XT *src = xc + (size_t)min(r0 + xrow, L - 1) * ps + b;
      if constexpr (sizeof(XT) == 4) {
        const float4 f = *reinterpret_cast<const float4 *>(src);
        xv[0] = f.x; xv[1] = f.y; xv[2] = f.z; xv[3] = f.w;
      } else {
        const double2 a = *reinterpret_cast<const double2 *>(src), b2 = *reinterpret_cast<const double2 *>(src + 2);
        xv[0] = a.x; xv[1] = a.y; xv[2] = b2.x; xv[3] = b2.y;
      }
    } else {
      xv[0] = xv[1] = xv[2] = xv[3] = (XT)0;
    }
  };
  auto xstore = [&](int s) {
    if (xact) {
      XT *dst = Xs + (size_t)(s & 1) * WS_CH * RT;
      const int b = WS_CH * s + 4 * xq;
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[(4 * xq + e) * RT + xrow] = (b + e < p) ? xv[e] : (XT)0;
    }
  };
  // rows of the tile x the two alphas ai0, ai1 of this wave (one per alpha group of the chunk), their chains interleaved: the
  // four waves' partial r summed, q = 1 - beta r, product and sum over the valid rows (fixed reduction tree: butterfly inside
  // the 16-lane rows, then (row0 row1)(row2 row3)), folded into the split's running (mantissa, exponent, sum, flag) by lane 0
  auto rdl = [](double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
  };
  auto nll_rows2 = [&](int ai0, int ai1) {
    if (ai0 >= nalpha) return;   // wave-uniform (ai1 > ai0)
    const bool two = ai1 < nalpha;
    const double beta0 = betas[ai0], beta1 = betas[two ? ai1 : ai0];
    double prod0 = 1.0, ssum0 = 0.0, prod1 = 1.0, ssum1 = 0.0;
    int neg = 0;
#pragma unroll
    for (int rr = 0; rr < RT; rr += 64) {
      const int row = rr + lane;
      if (row < RT) {
        const int I = row >> 4, src = 16 * (row & 3) + 4 * ((row >> 2) & 3) + wave;
        const double *rp0 = red, *rp1 = red + NW * NI * 64;
        const double r0v = (rp0[(0 * NI + I) * 64 + src] + rp0[(1 * NI + I) * 64 + src]) +
                           (rp0[(2 * NI + I) * 64 + src] + rp0[(3 * NI + I) * 64 + src]);
        const double r1v = (rp1[(0 * NI + I) * 64 + src] + rp1[(1 * NI + I) * 64 + src]) +
                           (rp1[(2 * NI + I) * 64 + src] + rp1[(3 * NI + I) * 64 + src]);
        const bool ok = rowok[row] != 0;
        const double q0 = __builtin_fma(-beta0, r0v, 1.0), q1 = __builtin_fma(-beta1, r1v, 1.0);
        // r / q by reciprocal + two Newton steps (1-2 ulp; the IEEE division sequence is 4x the instructions)
        double y0 = __builtin_amdgcn_rcp(q0), y1 = __builtin_amdgcn_rcp(q1);
        y0 = __builtin_fma(y0, __builtin_fma(-q0, y0, 1.0), y0);
        y1 = __builtin_fma(y1, __builtin_fma(-q1, y1, 1.0), y1);
        y0 = __builtin_fma(y0, __builtin_fma(-q0, y0, 1.0), y0);
        y1 = __builtin_fma(y1, __builtin_fma(-q1, y1, 1.0), y1);
        neg |= (ok && q0 < 0.0) ? 1 : 0;
        neg |= (ok && two && q1 < 0.0) ? 2 : 0;
        prod0 *= ok ? q0 : 1.0;
        prod1 *= ok ? q1 : 1.0;
        ssum0 += ok ? r0v * y0 : 0.0;
        ssum1 += ok ? r1v * y1 : 0.0;
      }
    }
    prod0 *= ws_dpp<0xB1>(prod0);  prod1 *= ws_dpp<0xB1>(prod1);  ssum0 += ws_dpp<0xB1>(ssum0);  ssum1 += ws_dpp<0xB1>(ssum1);
    prod0 *= ws_dpp<0x4E>(prod0);  prod1 *= ws_dpp<0x4E>(prod1);  ssum0 += ws_dpp<0x4E>(ssum0);  ssum1 += ws_dpp<0x4E>(ssum1);
    prod0 *= ws_dpp<0x141>(prod0); prod1 *= ws_dpp<0x141>(prod1); ssum0 += ws_dpp<0x141>(ssum0); ssum1 += ws_dpp<0x141>(ssum1);
    prod0 *= ws_dpp<0x140>(prod0); prod1 *= ws_dpp<0x140>(prod1); ssum0 += ws_dpp<0x140>(ssum0); ssum1 += ws_dpp<0x140>(ssum1);
    const double pw0 = (rdl(prod0, 0) * rdl(prod0, 16)) * (rdl(prod0, 32) * rdl(prod0, 48));
    const double pw1 = (rdl(prod1, 0) * rdl(prod1, 16)) * (rdl(prod1, 32) * rdl(prod1, 48));
    const double sw0 = (rdl(ssum0, 0) + rdl(ssum0, 16)) + (rdl(ssum0, 32) + rdl(ssum0, 48));
    const double sw1 = (rdl(ssum1, 0) + rdl(ssum1, 16)) + (rdl(ssum1, 32) + rdl(ssum1, 48));
    const int n0 = __any(neg & 1) ? 1 : 0, n1 = __any(neg & 2) ? 1 : 0;
    if (lane == 0) {
      const double pm0 = Pacc[ai0] * pw0;
      Eacc[ai0] += __builtin_amdgcn_frexp_exp(pm0);
      Pacc[ai0] = __builtin_amdgcn_frexp_mant(pm0);
      Sacc[ai0] += sw0;
      Nacc[ai0] |= n0;
      if (two) {
        const double pm1 = Pacc[ai1] * pw1;
        Eacc[ai1] += __builtin_amdgcn_frexp_exp(pm1);
        Pacc[ai1] = __builtin_amdgcn_frexp_mant(pm1);
        Sacc[ai1] += sw1;
        Nacc[ai1] |= n1;
      }
    }
  };

  // eight waves: one alpha per wave (group t = wave / 4, slot n = wave % 4 of the chunk), eight partials per row
  auto nll_rows1 = [&](int ai) {
    if (ai >= nalpha) return;   // wave-uniform
    const double beta0 = betas[ai];
    const double *rp = red + (size_t)(wave >> 2) * NW * NI * 64;
    double prod0 = 1.0, ssum0 = 0.0;
    int neg = 0;
#pragma unroll
    for (int rr = 0; rr < RT; rr += 64) {
      const int row = rr + lane;
      if (row < RT) {
        const int I = row >> 4, src = 16 * (row & 3) + 4 * ((row >> 2) & 3) + (wave & 3);
        const double r0v = ((rp[(0 * NI + I) * 64 + src] + rp[(1 * NI + I) * 64 + src]) +
                            (rp[(2 * NI + I) * 64 + src] + rp[(3 * NI + I) * 64 + src])) +
                           ((rp[(4 * NI + I) * 64 + src] + rp[(5 * NI + I) * 64 + src]) +
                            (rp[(6 * NI + I) * 64 + src] + rp[(7 * NI + I) * 64 + src]));
        const bool ok = rowok[row] != 0;
        const double q0 = __builtin_fma(-beta0, r0v, 1.0);
        double y0 = __builtin_amdgcn_rcp(q0);
        y0 = __builtin_fma(y0, __builtin_fma(-q0, y0, 1.0), y0);
        y0 = __builtin_fma(y0, __builtin_fma(-q0, y0, 1.0), y0);
        neg |= (ok && q0 < 0.0) ? 1 : 0;
        prod0 *= ok ? q0 : 1.0;
        ssum0 += ok ? r0v * y0 : 0.0;
      }
    }
    prod0 *= ws_dpp<0xB1>(prod0);  ssum0 += ws_dpp<0xB1>(ssum0);
    prod0 *= ws_dpp<0x4E>(prod0);  ssum0 += ws_dpp<0x4E>(ssum0);
    prod0 *= ws_dpp<0x141>(prod0); ssum0 += ws_dpp<0x141>(ssum0);
    prod0 *= ws_dpp<0x140>(prod0); ssum0 += ws_dpp<0x140>(ssum0);
    const double pw0 = (rdl(prod0, 0) * rdl(prod0, 16)) * (rdl(prod0, 32) * rdl(prod0, 48));
    const double sw0 = (rdl(ssum0, 0) + rdl(ssum0, 16)) + (rdl(ssum0, 32) + rdl(ssum0, 48));
    const int n0 = __any(neg & 1) ? 1 : 0;
    if (lane == 0) {
      const double pm0 = Pacc[ai] * pw0;
      Eacc[ai] += __builtin_amdgcn_frexp_exp(pm0);
      Pacc[ai] = __builtin_amdgcn_frexp_mant(pm0);
      Sacc[ai] += sw0;
      Nacc[ai] |= n0;
    }
  };

  for (int r0 = rbeg; r0 < rend; r0 += RT) {
    __syncthreads();   // the previous tile's readers of Bs / Xs / rowok / red are done
    unsigned long long tk0 = 0, tk1 = 0;
    if (stamp) tk0 = __builtin_readcyclecounter();
    if (tid < RT) rowok[tid] = (r0 + tid < rend) && (mp[min(r0 + tid, L - 1)] != 0);
    glds(0);
    xload(0, r0);
    xstore(0);
    double acc[NI][NJW];
#pragma unroll
    for (int I = 0; I < NI; ++I)
#pragma unroll
      for (int J = 0; J < NJW; ++J) acc[I][J] = 0.0;
    for (int s = 0; s < NS; ++s) {
      __syncthreads();   // chunk s has landed in Bs[s & 1] (the barrier's vmcnt(0)), its X chunk is stored; chunk s - 1 is consumed
      if (s + 1 < NS) glds(s + 1);
      const double *bs = Bs + (size_t)(s & 1) * CHD;
      if (s < NKC) {
        if (s + 1 < NKC) xload(s + 1, r0);
        // ---- Y^T tile += W^T X~^T over 16 bands: A operand W[16 s + 4 k4 + q][j0w + 4 J + n], B operand X~[row 16 I + 4 m + n][band]
        const XT *xs = Xs + (size_t)(s & 1) * WS_CH * RT;
#pragma unroll
        for (int k4 = 0; k4 < WS_CH / 4; ++k4) {
          double xr[NI];
          const double mub = mus[WS_CH * s + 4 * k4 + q];
#pragma unroll
          for (int I = 0; I < NI; ++I) xr[I] = ws_cvt(xs[(4 * k4 + q) * RT + 16 * I + 4 * m + n]) - mub;
          const double *brow = bs + (size_t)(4 * k4 + q) * LDW + j0w + n;
#pragma unroll
          for (int J = 0; J < NJW; ++J) {
            if (NJL == NJW || J < njw) {
              const double wr = brow[4 * J];
#pragma unroll
              for (int I = 0; I < NI; ++I) acc[I][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(wr, xr[I], acc[I][J], 0, 0, 0);
            }
          }
        }
        if (s + 1 < NKC) xstore(s + 1);
        if (s == NKC - 1) {   // Z = Y.^2: lane (q, m, n) holds Z[row 16 I + 4 m + n][col j0w + 4 J + q]
          if (stamp) tk1 = __builtin_readcyclecounter();
#pragma unroll
          for (int I = 0; I < NI; ++I)
#pragma unroll
            for (int J = 0; J < NJW; ++J) acc[I][J] = acc[I][J] * acc[I][J];
        }
      } else {
        // ---- the chunk's two alpha groups: r[row][alpha] over this wave's columns of Z, A operand Z, B operand
        //      C[col j0w + 4 J + q][alpha 8 ch + 4 t + n]; the waves' partials meet in red, one wave per alpha finishes them
        const int ch = s - NKC;
        double ra[NI], rb[NI];
#pragma unroll
        for (int I = 0; I < NI; ++I) { ra[I] = 0.0; rb[I] = 0.0; }
        const double *bcol = bs + (size_t)(j0w + q) * WS_CA + n;
#pragma unroll
        for (int J = 0; J < NJW; ++J) {
          if (NJL == NJW || J < njw) {
            const double ca = bcol[(size_t)4 * WS_CA * J], cb = bcol[(size_t)4 * WS_CA * J + 4];
#pragma unroll
            for (int I = 0; I < NI; ++I) {
              ra[I] = __builtin_amdgcn_mfma_f64_4x4x4f64(acc[I][J], ca, ra[I], 0, 0, 0);
              rb[I] = __builtin_amdgcn_mfma_f64_4x4x4f64(acc[I][J], cb, rb[I], 0, 0, 0);
            }
          }
        }
        // lane (q, m, n): r[row 16 I + 4 m + q][alpha 8 ch + 4 t + n] -> red[t][wave][I][lane]
#pragma unroll
        for (int I = 0; I < NI; ++I) {
          red[(wave * NI + I) * 64 + lane] = ra[I];
          red[NW * NI * 64 + (wave * NI + I) * 64 + lane] = rb[I];
        }
        __syncthreads();   // the waves' partials of both alpha groups are in red (rewritten after the next chunk barrier)
        if constexpr (NW == 4) nll_rows2(WS_CA * ch + wave, WS_CA * ch + 4 + wave);
        else nll_rows1(WS_CA * ch + wave);
      }
    }
    if (stamp && tid == 0) {
      const unsigned long long tk2 = __builtin_readcyclecounter();
      atomicAdd(&g_ws_stamps[0], 1ull);
      atomicAdd(&g_ws_stamps[1], tk1 - tk0);
      atomicAdd(&g_ws_stamps[2], tk2 - tk1);
    }
  }
  __syncthreads();
  for (int i = tid; i < NA; i += NT) {
    if (i < nalpha) {
      po[i] = log(Pacc[i]) + (double)Eacc[i] * 0.6931471805599453094;
      po[NA + i] = Nacc[i] ? __builtin_nan("") : Sacc[i];
    } else {
      po[i] = 0.0;
      po[NA + i] = 0.0;
    }
  }
}

template <typename XT, int NI, int CH, int NW, int LDW>
size_t ws_lds_bytes(int P16, int NA) {
  const int RT = 16 * NI;
  size_t b = (size_t)(2 * (CH > WS_CA ? CH : WS_CA) * LDW + P16 + 2 * NW * NI * 64 + 3 * NA) * sizeof(double);
  b += (size_t)2 * CH * RT * sizeof(XT);
  b += (size_t)(2 * NA + RT) * sizeof(int);
  return b;
}

template <typename XT, int NI, int NJW, int CH, int OCC, int NW = 4, int LDWT = 16 * NJW>
int ws_launch(const void *xt, const uint8_t *mask_t, const int32_t *nloo, const double *mu, const double *Wp, const double *Ct,
              const int32_t *status, const double *alphas, const SfGeom &g, int P16, int NA, int rows, int nsplit, double *part,
              hipStream_t st) {
  const size_t lds = ws_lds_bytes<XT, NI, CH, NW, LDWT>(P16, NA);
  if (lds > 160 * 1024) {
    sf_set_error("wide sweep: %d bands need %zu bytes of LDS", g.p, lds);
    return -2;
  }
  if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_wsweep<XT, NI, NJW, CH, OCC, NW, LDWT>), lds)) return rc;
  const int grid = 8 * sf_cdiv(g.ncols, 8) * nsplit;
  hipLaunchKernelGGL((k_wsweep<XT, NI, NJW, CH, OCC, NW, LDWT>), dim3(grid), dim3(64 * NW), lds, st, reinterpret_cast<const XT *>(xt), mask_t, nloo, mu, Wp,
                     Ct, status, alphas, g.nalpha, NA, g.lines, g.p, g.ps, P16, rows, nsplit, g.ncols, part, sf_tune().wjac_stamps);
  SF_LAUNCH_CHECK("k_wsweep");
  return 0;
}


// --------------------------------------------------------------------------------------------------------------------
// k_wsweep8: the sweep with two waves per SIMD and NO workgroup barrier in the Y phase.  Each of the eight waves owns a
// slice of Y's columns (NJW 4-column groups for waves 0-3, NJL for waves 4-7: 14 + 13 of the 108 groups at p <= 432) and
// streams ITS slice of W -- and later of C -- through its own two LDS buffers by global_load_lds; the operand images are laid
// out per (chunk, wave) by k_wmat_p8 / k_cmat_t8, so nothing a wave reads was written by another wave and its only
// synchronisation is its own s_waitcnt vmcnt.  The rows of X~ go from global memory straight to the B-operand registers:
// MFMA step k4 of a 16-band chunk takes band 16 s + 4 q + k4 from lane quarter q (any bijection of the chunk's bands onto
// (k4, q) is a valid K order as long as the W operand uses the same one), so one float4 per 16-row group is everything the
// lane needs of the chunk.  The waves drift apart; one's waits and address work run under the other's MFMAs.  The r phase
// keeps one exchange per 8-alpha chunk (the eight K-slices' partial r meet in `red`), with the row reductions of chunk
// ch - 1 placed beside the MFMAs of chunk ch.
// --------------------------------------------------------------------------------------------------------------------
// FACT (round 5): the r phase through the rank factorisation of its coefficient matrix (cmf_wlr.hip): beta_a r(a) = Z C diag(beta)
// = (Z Uc) T with Uc [p x 32] and T [32 x 208] -- four chunks of eight factor columns take the place of the 26 chunks of eight
// alphas (G = Z Uc: 448 instead of 2912 MFMAs per wave and tile, 8 exchanges instead of 26), then every wave multiplies the
// complete G [64 rows x 32] by ITS 28 alphas of T (224 MFMAs, no exchange) and reduces the rows of its alphas in registers.
// Factor column 31 is the all-ones column: G[., 31] = sum_j Z_j = r(alpha = 1), the grid point whose beta = 0 scales everything
// else of its column to nothing (T[31][a] = 1 there, 0 elsewhere).  Columns the factorisation refused (wlr == 0) are swept by the
// plain instantiation; each returns at once on the other's columns.
constexpr int W8_NW = 8;
constexpr int W8_NCF = 4;      // chunks of eight factor columns
constexpr int W8_GST = 36;     // row stride of G in LDS (doubles): the A-operand reads of 16 rows x 4 factors spread over the banks
constexpr int W8_NAG = 7;      // groups of four alphas per wave in the second product (8 x 28 = 224 >= 208 slots)
template <int NJW, int NJL, bool FACT = false>
__global__ __launch_bounds__(64 * W8_NW, 1) void k_wsweep8(const float *__restrict__ xt, const uint8_t *__restrict__ mask_t,
                                                       const int32_t *__restrict__ nloo, const double *__restrict__ mu,
                                                       const double *__restrict__ W8, const double *__restrict__ C8,
                                                       const int32_t *__restrict__ status, const double *__restrict__ alphas,
                                                       int nalpha, int NA, int L, int p, int ps, int P16, int rows_per_wg,
                                                       int nsplit, int ncols, double *__restrict__ part, int stamp,
                                                       const double *__restrict__ T8, const int32_t *__restrict__ wlr) {
  constexpr int NI = 4, RT = 64, NW = W8_NW, NT = 64 * NW;
  constexpr int SW = 4 * NJW;            // columns per wave slot
  constexpr int WSL = 16 * SW;           // doubles per W slice [16 band rows][SW]
  constexpr int CSL = WS_CA * SW;        // doubles per C slice [SW][8 alphas]
  constexpr int NPW = WSL * 8 / 1024, NPC = (CSL * 8 + 1023) / 1024;
  static_assert(WSL * 8 % 1024 == 0 && NJL <= NJW && NJL > 0 && NJW >= 13, "slice geometry (the row reductions are spread over 13 MFMA groups)");
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  double *Bs = reinterpret_cast<double *>(smraw);                 // [2][NW][WSL]
  double *mus = Bs + 2 * NW * WSL;                                 // [P16]
  constexpr int RS = 65, RW = NI * RS;   // red: [2 groups][NW][NI][RS]: the 16-row groups one double apart in the banks (the rows'
                                         // reads are a stride-4 gather inside a group: 8-way conflicts at stride 64, 2-way at 65)
  double *red = mus + P16;
  double *betas = red + 2 * NW * RW;                               // [NA]

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int c = 8 * (slot / nsplit) + xcd, split = slot % nsplit;
  if (c >= ncols) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, m = (lane >> 2) & 3, n = lane & 3;
  double *po = part + ((size_t)c * nsplit + split) * 2 * NA;
  if (status[c] != 0) {
    if (!FACT)
      for (int i = tid; i < 2 * NA; i += NT) po[i] = 0.0;
    return;
  }
  if (wlr != nullptr && (wlr[c] != 0) != FACT) return;   // the other instantiation's column
  const double nn = (double)nloo[c];
  for (int i = tid; i < NA; i += NT) betas[i] = (i < nalpha) ? (1.0 - alphas[i]) / (nn - 1.0) : 0.0;
  for (int i = tid; i < P16; i += NT) mus[i] = (i < p) ? mu[(size_t)c * p + i] : 0.0;
  __syncthreads();
  const int rbeg = split * rows_per_wg, rend = min(L, rbeg + rows_per_wg);
  const float *xc = xt + (size_t)c * L * ps;
  const uint8_t *mp = mask_t + (size_t)c * L;
  const int NKC = P16 / 16, NCC = FACT ? W8_NCF : NA / WS_CA;
  // the split's running (mantissa, exponent, sum, flag) of alpha 8 ch + wave live in lane ch of this wave's registers
  double Pv = 1.0, Sv = 0.0;
  int Ev = 0, Nv = 0;
  const char *Wc = reinterpret_cast<const char *>(W8 + ((size_t)c * NKC * NW + wave) * WSL);   // (wave-uniform)
  const char *Cc = reinterpret_cast<const char *>(C8 + ((size_t)c * NCC * NW + wave) * CSL);
  const unsigned loff = lane * 16;
  const int njw = wave < 4 ? NJW : NJL;
  double *bw0 = Bs + wave * WSL;   // this wave's slice of buffer 0 (buffer 1: + NW * WSL; plain arithmetic keeps the pointers in the LDS address space)
  // chunk s of a tile's stream (running count gs over the tiles: NS is odd, the buffer parity alternates) -> this wave's
  // buffer gs & 1
  auto glds = [&](int s, int gs) {
    char *dst = reinterpret_cast<char *>(bw0 + (gs & 1) * (NW * WSL));
    // (the instruction's immediate offset moves the global AND the LDS address: one address pair per 4 KB)
    if (s < NKC) {
      const char *src = Wc + (size_t)s * NW * WSL * 8;
#pragma unroll
      for (int pb = 0; pb < NPW; pb += 4) {
        ws_glb_void *gp = (ws_glb_void *)(src + pb * 1024 + loff);
        ws_lds_void *lp = (ws_lds_void *)(dst + pb * 1024);
        __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
        if (pb + 1 < NPW) __builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, 0);
        if (pb + 2 < NPW) __builtin_amdgcn_global_load_lds(gp, lp, 16, 2048, 0);
        if (pb + 3 < NPW) __builtin_amdgcn_global_load_lds(gp, lp, 16, 3072, 0);
      }
    } else if (FACT && s == NKC + W8_NCF) {   // this wave's 28 alphas of T: [8 factor groups][7 alpha groups][16], 7 KB like a W slice
      const char *src = reinterpret_cast<const char *>(T8 + ((size_t)c * NW + wave) * WSL);
#pragma unroll
      for (int pb = 0; pb < NPW; pb += 4) {
        ws_glb_void *gp = (ws_glb_void *)(src + pb * 1024 + loff);
        ws_lds_void *lp = (ws_lds_void *)(dst + pb * 1024);
        __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
        if (pb + 1 < NPW) __builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, 0);
        if (pb + 2 < NPW) __builtin_amdgcn_global_load_lds(gp, lp, 16, 2048, 0);
        if (pb + 3 < NPW) __builtin_amdgcn_global_load_lds(gp, lp, 16, 3072, 0);
      }
    } else {
      const char *src = Cc + (size_t)(s - NKC) * NW * CSL * 8;
      static_assert(NPC <= 4, "C slice pieces");
      ws_glb_void *gp = (ws_glb_void *)(src + loff);   // (the last piece runs 512 B past the slice: into the wave's own buffer / the next slice)
      ws_lds_void *lp = (ws_lds_void *)dst;
      __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
      if (1 < NPC) __builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, 0);
      if (2 < NPC) __builtin_amdgcn_global_load_lds(gp, lp, 16, 2048, 0);
      if (3 < NPC) __builtin_amdgcn_global_load_lds(gp, lp, 16, 3072, 0);
    }
  };
  // this lane's rows of X~, bands 16 s + 4 q .. + 3 (clamped inside the row; bands past the window are zeroed at use)
  auto xload = [&](float4 (&xv)[NI], int s, int r0) {
    const int b = min(16 * s + 4 * q, ps - 4);
#pragma unroll
    for (int I = 0; I < NI; ++I)
      xv[I] = *reinterpret_cast<const float4 *>(xc + (size_t)min(r0 + 16 * I + 4 * m + n, L - 1) * ps + b);
  };
  auto rdl = [](double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
  };
  // one alpha per wave (group t = wave / 4, slot wave % 4 of the chunk), lane = row, eight partials per (row, alpha); fixed
  // reduction tree
  constexpr int R2OFF = 512;   // red2: doubles into a wave's operand slice (above the 4 KB a C chunk's copy covers)
  static_assert(R2OFF * 8 >= NPC * 1024 && R2OFF + RW <= WSL, "red2 fits the free part of the slices");
  auto nll_rows1 = [&](int chi, bool ok, bool live = true) {
    const int ai = WS_CA * chi + wave;   // (alpha slots past the grid: beta = 0, never stored)
    const double beta0 = betas[ai];
    const int t = wave >> 2, odd = chi & 1;
    const double *rp = odd ? Bs + (size_t)t * NW * WSL + R2OFF : red + (size_t)t * NW * RW;   // partial of wave w: + w * rs
    const int rs = odd ? WSL : RW;
    const int I = lane >> 4, src = 16 * (lane & 3) + 4 * ((lane >> 2) & 3) + (wave & 3);
    rp += I * RS + src;
    const double r0v = ((rp[0 * rs] + rp[1 * rs]) + (rp[2 * rs] + rp[3 * rs])) + ((rp[4 * rs] + rp[5 * rs]) + (rp[6 * rs] + rp[7 * rs]));
    const double q0 = __builtin_fma(-beta0, r0v, 1.0);
    double y0 = __builtin_amdgcn_rcp(q0);
    y0 = __builtin_fma(y0, __builtin_fma(-q0, y0, 1.0), y0);
    y0 = __builtin_fma(y0, __builtin_fma(-q0, y0, 1.0), y0);
    const int neg = (ok && q0 < 0.0) ? 1 : 0;
    double prod0 = ok ? q0 : 1.0, ssum0 = ok ? r0v * y0 : 0.0;
    prod0 *= ws_dpp<0xB1>(prod0);  ssum0 += ws_dpp<0xB1>(ssum0);
    prod0 *= ws_dpp<0x4E>(prod0);  ssum0 += ws_dpp<0x4E>(ssum0);
    prod0 *= ws_dpp<0x141>(prod0); ssum0 += ws_dpp<0x141>(ssum0);
    prod0 *= ws_dpp<0x140>(prod0); ssum0 += ws_dpp<0x140>(ssum0);
    const double pw0 = (rdl(prod0, 0) * rdl(prod0, 16)) * (rdl(prod0, 32) * rdl(prod0, 48));
    const double sw0 = (rdl(ssum0, 0) + rdl(ssum0, 16)) + (rdl(ssum0, 32) + rdl(ssum0, 48));
    const int n0 = __any(neg) ? 1 : 0;
    const bool mine = live && lane == chi && ai < nalpha;
    const double pm0 = Pv * pw0;
    Ev += mine ? __builtin_amdgcn_frexp_exp(pm0) : 0;
    Pv = mine ? __builtin_amdgcn_frexp_mant(pm0) : Pv;
    Sv += mine ? sw0 : 0.0;
    Nv |= mine ? n0 : 0;
  };

  int gs = 0;
  float4 xa[NI], xb[NI];
  if (rbeg < rend) {
    glds(0, 0);
    xload(xa, 0, rbeg);
  }
  for (int r0 = rbeg; r0 < rend; r0 += RT) {
    unsigned long long tk0 = 0, tk1 = 0;
    if (stamp) tk0 = __builtin_readcyclecounter();
    const unsigned long long pt0 = tk0;
    const uint8_t mrow = mp[min(r0 + lane, L - 1)];   // (compared where it is used: nothing waits for it here)
    const bool rin = r0 + lane < rend;
    double acc[NI][NJW];
#pragma unroll
    for (int I = 0; I < NI; ++I)
#pragma unroll
      for (int J = 0; J < NJW; ++J) acc[I][J] = 0.0;
    // ---- Y^T slice += W^T X~^T, 16 bands a chunk: A operand W[band 16 s + 4 q + k4][col 4 J + n] at slice row 4 k4 + q,
    //      B operand X~[row 16 I + 4 m + n][the same band]
    for (int s = 0; s < NKC; ++s, ++gs) {
      __builtin_amdgcn_s_waitcnt(0x0070 | 0x0F00);   // vmcnt(0) (expcnt / lgkmcnt untouched): chunk s and its rows have landed
      asm volatile("" ::: "memory");
      // (the SIMD's arbiter favours one of its two waves for as long as both are ready: the priority alternates chunk by chunk,
      //  so both reach the end of the phase together)
      if (((s ^ (wave >> 2)) & 1) != 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
      glds(s + 1, gs + 1);                // (s + 1 == NKC: the first C chunk)
      if (s + 1 < NKC) xload(xb, s + 1, r0);
      const double *bs = bw0 + (gs & 1) * (NW * WSL);
      const double *mq = mus + 16 * s + 4 * q;
      const double mu4[4] = {mq[0], mq[1], mq[2], mq[3]};
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const bool bok = 16 * s + 4 * q + k4 < p;
        double xr[NI];
#pragma unroll
        for (int I = 0; I < NI; ++I) {
          const float xe = k4 == 0 ? xa[I].x : k4 == 1 ? xa[I].y : k4 == 2 ? xa[I].z : xa[I].w;
          xr[I] = bok ? (double)xe - mu4[k4] : 0.0;
        }
        const double *brow = bs + (size_t)(4 * k4 + q) * SW + n;
#pragma unroll
        for (int J = 0; J < NJW; ++J) {
          if (NJL == NJW || J < njw) {
            const double wr = brow[4 * J];
#pragma unroll
            for (int I = 0; I < NI; ++I) acc[I][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(wr, xr[I], acc[I][J], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);   // (keeps the next step's operand reads from being hoisted over this step: registers)
      }
#pragma unroll
      for (int I = 0; I < NI; ++I) xa[I] = xb[I];
    }
    if (stamp) tk1 = __builtin_readcyclecounter();
    // ---- Z = Y.^2: lane (q, m, n) holds Z[row 16 I + 4 m + n][col 4 J + q] of the slice
#pragma unroll
    for (int I = 0; I < NI; ++I)
#pragma unroll
      for (int J = 0; J < NJW; ++J) acc[I][J] = acc[I][J] * acc[I][J];
    // ---- r = Z C, eight alphas a chunk.  One exchange per chunk: the waves' partial r go to red (even chunks) or red2 (odd
    //      chunks: the upper 3 KB of the waves' own operand buffers, free while the 3.5 KB C slices stream through the lower
    //      4 KB), one barrier, and the row reductions of chunk ch - 1 are spread between the MFMAs of chunk ch (below).
    const bool more = r0 + RT < rend;
    const bool rowok = rin && mrow != 0;
    // (stamp = 16 + w: wave w's clocks of the chunk loop, summed in registers, one set of atomics per tile:
    //  [3] wait for the chunk's copy, [5] MFMAs + row reductions, [6] exchange + barrier, [7] the wave's whole tile)
    const bool probe = stamp >= 16 && wave == stamp - 16;
    unsigned long long pd0 = 0, pd1 = 0, pd2 = 0, pd3 = 0;
    if constexpr (FACT) {
      static_assert(W8_GST * 64 <= 2 * NW * RW && WSL == 8 * W8_NAG * 16, "G fits red; a T slice is a W slice");
      const unsigned long long vrows = __ballot(rowok);   // bit r: row r of the tile counts
      double gch[W8_NCF];
      // ---- G = Z Uc, eight factor columns a chunk: the waves' K-slice partials meet in red, wave w sums factor 8 ch + w (lane = row)
#pragma unroll
      for (int ch = 0; ch < W8_NCF; ++ch, ++gs) {
        __builtin_amdgcn_s_waitcnt(0x0070 | 0x0F00);
        asm volatile("" ::: "memory");
        glds(NKC + ch + 1, gs + 1);        // the next chunk; behind the last one the T slice
        const double *bcol = bw0 + (gs & 1) * (NW * WSL) + q * WS_CA + n;
        double ra[NI], rb[NI];
#pragma unroll
        for (int I = 0; I < NI; ++I) { ra[I] = 0.0; rb[I] = 0.0; }
#pragma unroll
        for (int J = 0; J < NJW; ++J) {
          const double ca = bcol[(size_t)4 * WS_CA * J], cb = bcol[(size_t)4 * WS_CA * J + 4];
#pragma unroll
          for (int I = 0; I < NI; ++I) {
            ra[I] = __builtin_amdgcn_mfma_f64_4x4x4f64(acc[I][J], ca, ra[I], 0, 0, 0);
            rb[I] = __builtin_amdgcn_mfma_f64_4x4x4f64(acc[I][J], cb, rb[I], 0, 0, 0);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // red is free: the last chunk's partials (the last tile's G) are read
        double *wa = red + (size_t)wave * RW, *wb = red + (size_t)(NW + wave) * RW;
#pragma unroll
        for (int I = 0; I < NI; ++I) {
          wa[I * RS + lane] = ra[I];
          wb[I * RS + lane] = rb[I];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const double *rp = red + (size_t)(wave >> 2) * NW * RW + (lane >> 4) * RS + 16 * (lane & 3) + 4 * ((lane >> 2) & 3) + (wave & 3);
        gch[ch] = ((rp[0 * RW] + rp[1 * RW]) + (rp[2 * RW] + rp[3 * RW])) + ((rp[4 * RW] + rp[5 * RW]) + (rp[6 * RW] + rp[7 * RW]));
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // every wave has its factors: red becomes G [64][GST]
#pragma unroll
      for (int ch = 0; ch < W8_NCF; ++ch) red[lane * W8_GST + 8 * ch + wave] = gch[ch];
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      double ga[NI][2 * W8_NCF];           // A operands: G[row 16 I + 4 m + n][factor 4 kk + q]
#pragma unroll
      for (int I = 0; I < NI; ++I)
#pragma unroll
        for (int kk = 0; kk < 2 * W8_NCF; ++kk) ga[I][kk] = red[(16 * I + 4 * m + n) * W8_GST + 4 * kk + q];
      __builtin_amdgcn_s_waitcnt(0x0070 | 0x0F00);   // the T slice has landed
      asm volatile("" ::: "memory");
      const double *tb = bw0 + (gs & 1) * (NW * WSL) + 4 * q + n;
      ++gs;
      if (more) {   // the next tile's first chunk goes to the other buffer (its last reader was factor chunk 3)
        glds(0, gs);
        xload(xa, 0, r0 + RT);
      }
      // ---- beta r = G T for this wave's 28 alphas; lane (q, m, n) gets rows 16 I + 4 m + q of alpha 28 w + 4 ag + n
#pragma unroll
      for (int ag = 0; ag < W8_NAG; ++ag) {
        double o[NI];
#pragma unroll
        for (int I = 0; I < NI; ++I) o[I] = 0.0;
#pragma unroll
        for (int kk = 0; kk < 2 * W8_NCF; ++kk) {
          const double tv = tb[(kk * W8_NAG + ag) * 16];
#pragma unroll
          for (int I = 0; I < NI; ++I) o[I] = __builtin_amdgcn_mfma_f64_4x4x4f64(ga[I][kk], tv, o[I], 0, 0, 0);
        }
        const int ai = 28 * wave + 4 * ag + n;
        const double beta0 = betas[ai < NA ? ai : NA - 1];
        const bool bz = !(beta0 > 0.0);                    // alpha = 1 (and the padding slots): o is r itself, q = 1
        double prod0 = 1.0, ssum0 = 0.0;
        int neg = 0;
#pragma unroll
        for (int I = 0; I < NI; ++I) {
          const bool ok = (vrows >> (16 * I + 4 * m + q)) & 1ull;
          const double q0 = bz ? 1.0 : 1.0 - o[I];
          double y0 = __builtin_amdgcn_rcp(q0);
          y0 = __builtin_fma(y0, __builtin_fma(-q0, y0, 1.0), y0);
          y0 = __builtin_fma(y0, __builtin_fma(-q0, y0, 1.0), y0);
          prod0 *= ok ? q0 : 1.0;
          ssum0 += ok ? o[I] * y0 : 0.0;
          neg |= (ok && q0 < 0.0) ? 1 : 0;
        }
        // the 16 lanes (q, m) of an alpha: two rotations inside the 16-lane rows (m), two exchanges across them (q)
        prod0 *= ws_dpp<0x124>(prod0);  ssum0 += ws_dpp<0x124>(ssum0);   // row_ror:4
        prod0 *= ws_dpp<0x128>(prod0);  ssum0 += ws_dpp<0x128>(ssum0);   // row_ror:8
        neg |= __builtin_amdgcn_update_dpp(0, neg, 0x124, 0xF, 0xF, false);
        neg |= __builtin_amdgcn_update_dpp(0, neg, 0x128, 0xF, 0xF, false);
#pragma unroll
        for (int msk = 16; msk <= 32; msk <<= 1) {
          const double po2 = __hiloint2double(__shfl_xor(__double2hiint(prod0), msk, 64), __shfl_xor(__double2loint(prod0), msk, 64));
          const double so2 = __hiloint2double(__shfl_xor(__double2hiint(ssum0), msk, 64), __shfl_xor(__double2loint(ssum0), msk, 64));
          prod0 *= po2;
          ssum0 += so2;
          neg |= __shfl_xor(neg, msk, 64);
        }
        const bool mine = (lane >> 2) == ag && lane < 4 * W8_NAG;   // lane l keeps alpha 28 w + l (its n is l & 3)
        const double pm0 = Pv * prod0;
        Ev += mine ? __builtin_amdgcn_frexp_exp(pm0) : 0;
        Pv = mine ? __builtin_amdgcn_frexp_mant(pm0) : Pv;
        Sv += mine ? ssum0 : 0.0;
        Nv |= mine ? neg : 0;
      }
    } else {
    for (int ch = 0; ch < NCC; ++ch, ++gs) {
      unsigned long long pk0 = 0, pk1 = 0, pk2 = 0, pk3 = 0, pk4 = 0;
      if (probe) pk0 = __builtin_readcyclecounter();
      __builtin_amdgcn_s_waitcnt(0x0070 | 0x0F00);
      asm volatile("" ::: "memory");
      if (probe) pk1 = __builtin_readcyclecounter();
      if (wave >= 4) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);   // (first half of the chunk; see J == JH below)
      if (ch + 1 < NCC) glds(NKC + ch + 1, gs + 1);
      double ra[NI], rb[NI];
      const double *bcol = bw0 + (gs & 1) * (NW * WSL) + q * WS_CA + n;
      // the slice's operands: the first half up front, the second half while the first is being consumed
      constexpr int JH = NJW / 2;
      double ca[NJW], cb[NJW];
#pragma unroll
      for (int J = 0; J < JH; ++J) { ca[J] = bcol[(size_t)4 * WS_CA * J]; cb[J] = bcol[(size_t)4 * WS_CA * J + 4]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int I = 0; I < NI; ++I) { ra[I] = 0.0; rb[I] = 0.0; }
      // The row reductions of chunk ch - 1 (alpha 8 (ch - 1) + wave, lane = row; chunk 0 reduces stale partials into nothing)
      // are spread BY HAND between the 14 groups of eight MFMAs, one dependent step per group, the order pinned by
      // sched_barrier: float64 vector instructions of ANOTHER wave are not issued while a wave streams float64 MFMAs (measured:
      // the reductions of a SIMD's second wave took 3.5 k cycles beside the first wave's MFMAs, 0.9 k alone), and the
      // scheduler leaves the two parts one after the other even with sched_group_barrier pipelines -- so the overlap is
      // written into the instruction stream: each step's latency is covered by the next group's 136 cycles of MFMA.
      const int chi = max(ch - 1, 0), ai = WS_CA * chi + wave;
      const bool live = ch > 0 && ai < nalpha;
      const double *rp;
      int rs;
      double pr[8], r0v = 0.0, q0 = 1.0, y0 = 0.0, prod0 = 1.0, ssum0 = 0.0, beta0 = 0.0, pw0 = 1.0, sw0 = 0.0;
      int neg = 0;
#pragma unroll
      for (int J = 0; J < NJW; ++J) {   // (a wave of NJL groups runs its last group on zeros: Z = 0 there, C zero-padded)
#pragma unroll
        for (int I = 0; I < NI; ++I) {
          ra[I] = __builtin_amdgcn_mfma_f64_4x4x4f64(acc[I][J], ca[J], ra[I], 0, 0, 0);
          rb[I] = __builtin_amdgcn_mfma_f64_4x4x4f64(acc[I][J], cb[J], rb[I], 0, 0, 0);
        }
        if (J == JH) __builtin_amdgcn_s_setprio(0);   // every exchange waits for the SIMD's slower wave: the younger wave leads the
                                                      // first half of a chunk, the older one the second, and they finish together
        if (J == 0) {
          const int t = wave >> 2, odd = chi & 1;
          rp = odd ? Bs + (size_t)t * NW * WSL + R2OFF : red + (size_t)t * NW * RW;   // partial of wave w: + w * rs
          rs = odd ? WSL : RW;
          rp += (lane >> 4) * RS + 16 * (lane & 3) + 4 * ((lane >> 2) & 3) + (wave & 3);
#pragma unroll
          for (int w = 0; w < 4; ++w) pr[w] = rp[w * rs];
          beta0 = betas[ai];
        } else if (J == 1) {
#pragma unroll
          for (int JJ = JH; JJ < NJW; ++JJ) { ca[JJ] = bcol[(size_t)4 * WS_CA * JJ]; cb[JJ] = bcol[(size_t)4 * WS_CA * JJ + 4]; }
        } else if (J == 2) {
          pr[0] += pr[1]; pr[2] += pr[3];
#pragma unroll
          for (int w = 4; w < 8; ++w) pr[w] = rp[w * rs];
        } else if (J == 3) {
          pr[0] += pr[2]; pr[4] += pr[5]; pr[6] += pr[7];
          r0v = pr[0] + (pr[4] + pr[6]);
          q0 = __builtin_fma(-beta0, r0v, 1.0);
          y0 = __builtin_amdgcn_rcp(q0);
        } else if (J == 4) {
          y0 = __builtin_fma(y0, __builtin_fma(-q0, y0, 1.0), y0);
        } else if (J == 5) {
          y0 = __builtin_fma(y0, __builtin_fma(-q0, y0, 1.0), y0);
          neg = (rowok && q0 < 0.0) ? 1 : 0;
          prod0 = rowok ? q0 : 1.0;
        } else if (J == 6) {
          ssum0 = rowok ? r0v * y0 : 0.0;
          prod0 *= ws_dpp<0xB1>(prod0);
        } else if (J == 7) {
          ssum0 += ws_dpp<0xB1>(ssum0);
          prod0 *= ws_dpp<0x4E>(prod0);
        } else if (J == 8) {
          ssum0 += ws_dpp<0x4E>(ssum0);
          prod0 *= ws_dpp<0x141>(prod0);
        } else if (J == 9) {
          ssum0 += ws_dpp<0x141>(ssum0);
          prod0 *= ws_dpp<0x140>(prod0);
        } else if (J == 10) {
          ssum0 += ws_dpp<0x140>(ssum0);
          pw0 = (rdl(prod0, 0) * rdl(prod0, 16)) * (rdl(prod0, 32) * rdl(prod0, 48));
        } else if (J == 11) {
          sw0 = (rdl(ssum0, 0) + rdl(ssum0, 16)) + (rdl(ssum0, 32) + rdl(ssum0, 48));
        } else if (J == 12) {
          const int n0 = __any(neg) ? 1 : 0;
          const bool mine = live && lane == chi;
          const double pm0 = Pv * pw0;
          Ev += mine ? __builtin_amdgcn_frexp_exp(pm0) : 0;
          Pv = mine ? __builtin_amdgcn_frexp_mant(pm0) : Pv;
          Sv += mine ? sw0 : 0.0;
          Nv |= mine ? n0 : 0;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      pk2 = pk1;
      if (probe) { asm volatile("" : "+v"(ra[0]), "+v"(rb[3]), "+v"(Pv), "+v"(Sv)); pk3 = __builtin_readcyclecounter(); }
      double *wa = (ch & 1) ? Bs + (size_t)wave * WSL + R2OFF : red + (size_t)wave * RW;
      double *wb = (ch & 1) ? Bs + (size_t)(NW + wave) * WSL + R2OFF : red + (size_t)(NW + wave) * RW;
#pragma unroll
      for (int I = 0; I < NI; ++I) {
        wa[I * RS + lane] = ra[I];
        wb[I * RS + lane] = rb[I];
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (no vmcnt: the next chunk's copy stays in flight)
      if (probe) {
        pk4 = __builtin_readcyclecounter();
        pd0 += pk1 - pk0; pd1 += pk2 - pk1; pd2 += pk3 - pk2; pd3 += pk4 - pk3;
      }
    }
    if (probe && lane == 0) {
      atomicAdd(&g_ws_stamps[7], __builtin_readcyclecounter() - pt0);   // this wave's whole tile
      atomicAdd(&g_ws_stamps[3], pd0);
      atomicAdd(&g_ws_stamps[4], pd1);
      atomicAdd(&g_ws_stamps[5], pd2);
      atomicAdd(&g_ws_stamps[6], pd3);
    }
    nll_rows1(NCC - 1, rowok);
    if (more) {   // the next tile's first chunk (not earlier: W slices cover the buffers' upper parts, which held red2)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave has read the last chunk's partials
      glds(0, gs);
      xload(xa, 0, r0 + RT);
    }
    }
    if (stamp && tid == 0) {
      const unsigned long long tk2 = __builtin_readcyclecounter();
      atomicAdd(&g_ws_stamps[0], 1ull);
      atomicAdd(&g_ws_stamps[1], tk1 - tk0);
      atomicAdd(&g_ws_stamps[2], tk2 - tk1);
    }
  }
  if constexpr (FACT) {
    if (lane < 4 * W8_NAG) {
      const int ai = 28 * wave + lane;
      if (ai < NA) {
        const bool in = ai < nalpha;
        const double beta0 = betas[ai];
        po[ai] = in ? log(Pv) + (double)Ev * 0.6931471805599453094 : 0.0;
        po[NA + ai] = in ? (Nv ? __builtin_nan("") : (beta0 > 0.0 ? Sv / beta0 : Sv)) : 0.0;   // the sums were of beta r / q
      }
    }
  } else if (lane < NCC) {
    const int ai = WS_CA * lane + wave;
    const bool in = ai < nalpha;
    po[ai] = in ? log(Pv) + (double)Ev * 0.6931471805599453094 : 0.0;
    po[NA + ai] = in ? (Nv ? __builtin_nan("") : Sv) : 0.0;
  }
}

// the operand images of k_wsweep8: W8 [NKC][8 waves][16 slice rows][SW], slice row 4 k4 + qq = band 16 s + 4 qq + k4;
// C8 [NCC][8 waves][SW][8 alphas]; a wave's columns: j0(wave) + jl, jl < 4 (NJW or NJL), zero elsewhere
template <int NJW, int NJL>
__global__ void k_wmat_p8(const double *__restrict__ evec, const double *__restrict__ d, int p, int NKC, double *__restrict__ W) {
  constexpr int SW = 4 * NJW, WSL = 16 * SW;
  const int c = blockIdx.y;
  const double *ev = evec + (size_t)c * p * p, *dd = d + (size_t)c * p;
  double *o = W + (size_t)c * NKC * 8 * WSL;
  const int total = NKC * 8 * WSL;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int s = i / (8 * WSL), r1 = i - s * 8 * WSL, w = r1 / WSL, r2 = r1 - w * WSL, row = r2 / SW, jl = r2 - row * SW;
    const int b = 16 * s + 4 * (row & 3) + (row >> 2);
    const int j = 4 * (w < 4 ? NJW * w : 4 * NJW + NJL * (w - 4)) + jl;
    const bool in = jl < 4 * (w < 4 ? NJW : NJL) && b < p && j < p;
    o[i] = in ? ev[(size_t)j * p + b] / dd[b] : 0.0;
  }
}
template <int NJW, int NJL>
__global__ void k_cmat_t8(const double *__restrict__ lam, const int32_t *__restrict__ nloo, const int32_t *__restrict__ status,
                          const double *__restrict__ alphas, int nalpha, int NCC, int p, double *__restrict__ Ct) {
  constexpr int SW = 4 * NJW, CSL = WS_CA * SW;
  const int c = blockIdx.y;
  const double nn = (double)nloo[c];
  const bool ok = status[c] == 0;
  const double *lc = lam + (size_t)c * p;
  double *o = Ct + (size_t)c * NCC * 8 * CSL;
  const int total = NCC * 8 * CSL;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int ch = i / (8 * CSL), r1 = i - ch * 8 * CSL, w = r1 / CSL, r2 = r1 - w * CSL, jl = r2 >> 3, a = 8 * ch + (r2 & 7);
    const int j = 4 * (w < 4 ? NJW * w : 4 * NJW + NJL * (w - 4)) + jl;
    double v = 0.0;
    if (ok && a < nalpha && jl < 4 * (w < 4 ? NJW : NJL) && j < p) {
      const double al = alphas[a];
      const double beta = (1.0 - al) / (nn - 1.0);
      v = 1.0 / ((nn * beta) * lc[j] + al);
    }
    o[i] = v;
  }
}
template <int NJW>
size_t ws8_lds_bytes(int P16, int NA) {
  return (size_t)(2 * W8_NW * 64 * NJW + P16 + 2 * W8_NW * 4 * 65 + NA) * sizeof(double);
}
}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------------
int sf_wgemm_splits(const SfGeom &g) {   // a function of the number of lines only (bit-identical shards, cmf_common.h)
  // 32 splits of 640 rows at 20000 lines: the 32 CUs of an XCD then work on ONE column at a time (sf_launch_wsweep's block
  // order), whose W and C (2.2 MB) stay in that XCD's 4 MB L2 while its ten 64-row tiles per split stream them
  int ns = sf_cdiv(g.lines, 640);
  if (ns > 32) ns = 32;
  return ns < 1 ? 1 : ns;
}
static int wg_p16(const SfGeom &g) { return (g.p + 15) / 16 * 16; }
static int wg_ldw(const SfGeom &g) { return g.p <= 256 ? 256 : (g.p <= 432 ? 432 : 512); }   // 16 NJW of the instantiation used
static int wg_na(const SfGeom &g) { return g.nu * 16; }   // (a multiple of the 8 alphas of a C chunk)   // alpha slots = the stride k_nll reads the partials with
// scratch of the fused route: W [ncols][P16][LDW], Ct [ncols][NA / 16][LDW][16], the sweep partials [ncols][nsplit][2][NA]
static size_t wg_ldw_alloc(const SfGeom &g) { const size_t l = wg_ldw(g); return l == 432 ? 448 : l; }   // (k_wsweep8's 8 x 56 slots)
static bool wg_factored(const SfGeom &g, int xt_f64) {   // the rank-factored r phase: k_wsweep8's windows, 201-point grid
  return !xt_f64 && wg_ldw(g) == 432 && wg_na(g) == 208 && sf_tune().wsweep_variant == 0;
}
size_t sf_wgemm_operand_bytes(const SfGeom &g) {
  const size_t P16 = wg_p16(g), LDW = wg_ldw_alloc(g), NA = wg_na(g);
  size_t b = sf_align((size_t)g.ncols * P16 * LDW * sizeof(double)) + sf_align((size_t)g.ncols * NA * LDW * sizeof(double) + 1024);
  if (wg_ldw(g) == 432 && wg_na(g) == 208) b += sf_wlr_image_bytes(g.ncols) + sf_wlr_scratch_bytes(g.ncols);   // (whatever the knob says)
  return b;
}
size_t sf_wgemm_part_bytes(const SfGeom &g) {
  return sf_align((size_t)g.ncols * sf_wgemm_splits(g) * 2 * wg_na(g) * sizeof(double));
}

template <typename XT, int TI, int OCC>
static int wsyrk_go(const void *xt, const uint8_t *mask_t, const int32_t *nuse, const double *mu, const SfGeom &g, int c0, int nb,
                    double *cov, hipStream_t st) {
  constexpr int T = 32 * TI;
  const int ntile = sf_cdiv(g.p, T);
  const int npair = ntile * (ntile + 1) / 2;
  const size_t lds = ((size_t)2 * SY_KC * (2 * T + 16) + 4 * T) * sizeof(double);
  const size_t colx = (size_t)g.lines * g.ps;
  if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_wsyrk<XT, TI, OCC>), lds)) return rc;
  hipLaunchKernelGGL((k_wsyrk<XT, TI, OCC>), dim3(npair, nb), dim3(256), lds, st, reinterpret_cast<const XT *>(xt) + (size_t)c0 * colx,
                     mask_t + (size_t)c0 * g.lines, nuse + c0, mu + (size_t)c0 * g.p, g.lines, g.p, g.ps, ntile,
                     cov + (size_t)c0 * g.p * g.p);
  SF_LAUNCH_CHECK("k_wsyrk");
  return 0;
}
int sf_launch_wsyrk(const void *xt, int xt_f64, const uint8_t *mask_t, const int32_t *nuse, const double *mu, const SfGeom &g,
                    int c0, int nb, double *cov, hipStream_t st) {
  // 96-band tiles (16 % fewer MFMAs at p = 425 than the 128-band tiles of the round's first half), two workgroups per CU: with the
  // store half of the loader inside the MFMA stream and the operands of the next MFMA step prefetched the kernel wants 216
  // registers (three workgroups per CU at 168 registers spilled: 435 ms a flightline against 332)
  if (xt_f64) return wsyrk_go<double, 3, 2>(xt, mask_t, nuse, mu, g, c0, nb, cov, st);
  // Round 6 ("the tile size has to follow p"): the band tile that pads the upper triangle of tile pairs least.
  // 64-band tiles are 28 pairs at p = 425 / 416 (114.7 k band pairs) against the 15 pairs of 96 (138.2 k): 17 % fewer MFMAs for a
  // third more operand reads per MFMA -- measured 285 -> 275 ms a flightline at p = 425, the covariance bit-identical (every
  // element is the same sum over the rows in the same order).  sf_debug_set(5, 3): 96-band tiles whatever p.
  auto pairs = [&](int T) { const long long n = sf_cdiv(g.p, T); return n * (n + 1) / 2 * T * T; };
  if (sf_tune().cov_variant != 3 && pairs(64) * 103 < pairs(96) * 100) return wsyrk_go<float, 2, 3>(xt, mask_t, nuse, mu, g, c0, nb, cov, st);
  return wsyrk_go<float, 3, 2>(xt, mask_t, nuse, mu, g, c0, nb, cov, st);
}

// the sweep of columns c0 .. c0 + nb - 1: operands into `opnd` (sf_wgemm_operand_bytes of the nb-column geometry), partials
// into part[(c0 + c) * nsplit + split] of the whole-flightline array
int sf_launch_wsweep(const void *xt, int xt_f64, const uint8_t *mask_t, const int32_t *nloo, const double *mu, const double *d,
                     const double *lam, const double *evec, const int32_t *status, const double *alphas, const SfGeom &g, int c0,
                     int nb, void *opnd, double *part, hipStream_t st) {
  const int P16 = wg_p16(g), LDW = wg_ldw(g), NA = wg_na(g), nsplit = sf_wgemm_splits(g);
  const int rows = sf_cdiv(sf_cdiv(g.lines, nsplit), 64) * 64;   // whole 64-row tiles (also a multiple of the 32- / 16-row tiles)
  SfGeom gb = g;
  gb.ncols = nb;
  double *Wp = reinterpret_cast<double *>(opnd);
  double *Ct = reinterpret_cast<double *>(reinterpret_cast<char *>(opnd) + sf_align((size_t)nb * P16 * wg_ldw_alloc(g) * sizeof(double)));
  const size_t colx = (size_t)g.lines * g.ps * (xt_f64 ? sizeof(double) : sizeof(float));
  const void *xb = reinterpret_cast<const char *>(xt) + (size_t)c0 * colx;
  // sf_debug_set(24, v): 0 = k_wsweep8 where it applies (float32 rows, 256 < p <= 432); 1: 32-row tiles, 8-band chunks, two
  // workgroups per CU; 2: eight waves on shared chunks; 4: round 4's first form (four waves, shared chunks)
  const int lite = sf_tune().wsweep_variant;
  if (!xt_f64 && LDW == 432 && (lite == 0 || lite == 5)) {   // (5: k_wsweep8 with the r phase unfactored for every column)
    // eight waves, wave-private operand slices, no barrier in the Y phase (k_wsweep8)
    constexpr int NJW = 14, NJL = 13;
    const size_t lds = ws8_lds_bytes<NJW>(P16, NA);
    if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_wsweep8<NJW, NJL>), lds)) return rc;
    hipLaunchKernelGGL((k_wmat_p8<NJW, NJL>), dim3(128, nb), dim3(256), 0, st, evec + (size_t)c0 * g.p * g.p, d + (size_t)c0 * g.p, g.p,
                       P16 / 16, Wp);
    SF_LAUNCH_CHECK("k_wmat_p8");
    hipLaunchKernelGGL((k_cmat_t8<NJW, NJL>), dim3(64, nb), dim3(256), 0, st, lam + (size_t)c0 * g.p, nloo + c0, status + c0, alphas,
                       g.nalpha, NA / 8, g.p, Ct);
    SF_LAUNCH_CHECK("k_cmat_t8");
    const int grid = 8 * sf_cdiv(nb, 8) * nsplit;
    // sf_debug_set(24, 5): every column through the unfactored r phase (the A/B of the round-5 factorisation)
    const double *U8 = nullptr, *T8 = nullptr;
    const int32_t *wlr = nullptr;
    if (wg_factored(g, xt_f64)) {
      char *img = reinterpret_cast<char *>(Ct) + sf_align((size_t)nb * NA * wg_ldw_alloc(g) * sizeof(double) + 1024);
      char *scr = img + sf_wlr_image_bytes(nb);
      if (int rc = sf_launch_wlr(lam + (size_t)c0 * g.p, nloo + c0, status + c0, alphas, g.nalpha, g.p, nb, NJW, NJL, scr, img, &U8, &T8,
                                 &wlr, st))
        return rc;
      if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_wsweep8<NJW, NJL, true>), lds)) return rc;
      hipLaunchKernelGGL((k_wsweep8<NJW, NJL, true>), dim3(grid), dim3(64 * W8_NW), lds, st, reinterpret_cast<const float *>(xb),
                         mask_t + (size_t)c0 * g.lines, nloo + c0, mu + (size_t)c0 * g.p, Wp, U8, status + c0, alphas, g.nalpha, NA,
                         g.lines, g.p, g.ps, P16, rows, nsplit, nb, part + (size_t)c0 * nsplit * 2 * NA, 0, T8, wlr);
      SF_LAUNCH_CHECK("k_wsweep8(factored)");
    }
    hipLaunchKernelGGL((k_wsweep8<NJW, NJL>), dim3(grid), dim3(64 * W8_NW), lds, st, reinterpret_cast<const float *>(xb),
                       mask_t + (size_t)c0 * g.lines, nloo + c0, mu + (size_t)c0 * g.p, Wp, Ct, status + c0, alphas, g.nalpha, NA,
                       g.lines, g.p, g.ps, P16, rows, nsplit, nb, part + (size_t)c0 * nsplit * 2 * NA, sf_tune().wjac_stamps, nullptr, wlr);
    SF_LAUNCH_CHECK("k_wsweep8");
    return 0;
  }
  hipLaunchKernelGGL(k_wmat_p, dim3(128, nb), dim3(256), 0, st, evec + (size_t)c0 * g.p * g.p, d + (size_t)c0 * g.p, g.p, P16, LDW, Wp);
  SF_LAUNCH_CHECK("k_wmat_p");
  hipLaunchKernelGGL(k_cmat_t, dim3(64, nb), dim3(256), 0, st, lam + (size_t)c0 * g.p, nloo + c0, status + c0, alphas, g.nalpha,
                     NA / 8, g.p, LDW, Ct);
  SF_LAUNCH_CHECK("k_cmat_t");
  double *pb = part + (size_t)c0 * nsplit * 2 * NA;
#define WS_GO(XT, NI, NJW, CH, OCC, ...)                                                                                          \
  return ws_launch<XT, NI, NJW, CH, OCC, ##__VA_ARGS__>(xb, mask_t + (size_t)c0 * g.lines, nloo + c0, mu + (size_t)c0 * g.p, Wp, Ct, status + c0, alphas, gb, \
                                P16, NA, rows, nsplit, pb, st)
  if (!xt_f64) {
    if (LDW == 256) WS_GO(float, 4, 16, 16, 1);
    if (LDW == 432 && lite == 1) WS_GO(float, 2, 27, 8, 2);
    if (LDW == 432 && lite == 2) WS_GO(float, 4, 14, 16, 1, 8, 432);
    if (LDW == 432) WS_GO(float, 4, 27, 16, 1);
    WS_GO(float, 2, 32, 16, 1);
  } else {
    if (LDW == 256) WS_GO(double, 2, 16, 16, 1);
    if (LDW == 432) WS_GO(double, 2, 27, 16, 1);
    WS_GO(double, 2, 32, 16, 1);
  }
#undef WS_GO
}

extern "C" int sf_debug_wsweep_stamps(unsigned long long *out4 /* [8] */, int reset) {
  if (out4) SF_HIP(hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_ws_stamps), 8 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    SF_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_ws_stamps), z, sizeof(z)));
  }
  return 0;
}
