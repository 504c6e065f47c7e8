// Stage 7: per-pixel matched-filter score + output assembly -- the HBM-roofline kernel of the path.
//
// Replaces cmf/robust_mf.py:377-397 and :266: mf = (x - mu) C^-1 t / (t C^-1 t) is one p-long dot product
// per pixel with the per-column vector from stage 6 (score = x . filt - bias, float64 accumulate -- the mean
// term is ~600x the spread of the scores), rows that fail the validity test keep NODATA,
// and the three RGB bands are copied next to the score so each pixel's 32-byte BIP record
// [R, G, B, CMF] (float64) is written once, whole.
//
// k_score (production): lane = sample, 64-sample column blocks (256 contiguous bytes per wave instruction), the
// block's 64 filter vectors resident in LDS as [band][64] for the workgroup's lifetime, four workgroups = 16 waves per
// CU with 64 loads each in flight (256 KB per CU), XCD-aware block order.  Round 2 added the staged stores: a wave's
// 64 records of a line (2 KB of contiguous output) pass through 1 KB of LDS and leave as 1 KB contiguous store
// instructions instead of 16-byte pieces at a 32-byte stride (-0 .. -3 % per launch, never slower).
// k_score_blk2 (sf_debug_set(1, 10); measured alternative): 128-sample blocks, a lane owns two adjacent samples.
//
// What round 2 measured about this kernel (profiles/r02_score_kernel_experiments.md; tools/microbench/readbw*.hip,
// tools/tune_score.py with -DSF_SCORE_EXPERIMENTS): the launch is NOT limited by its load geometry.  With the stores
// switched off every form -- 64-sample blocks, 128-sample blocks with 8-byte loads, whole rows per workgroup with LDS
// filter tiles or with the filter streamed from L2 -- reads the window at 5.4-6.0 TB/s (0.60-0.70 ms); the 0.38 GB of
// product stores then add 0.17-0.19 ms whatever their form (16-byte pieces or 1 KB runs, the product's layout or one
// contiguous run per wave, plain / nt / sc1 / sc0 sc1, issued before or after the next batch's loads): 0.45 ms per GB
// written beside a saturated read stream, twice what a plain copy pays.  The fused RGB copy is 3/4 of those bytes.
// All forms accumulate a pixel's dot product in band order with one float64 FMA per band: bit-identical results.
// Algorithmic bytes per pixel: 4p (cube) + 8 (score) [+ 12 read + 24 written when RGB is fused].  2p flops per
// pixel -> HBM-bound by a wide margin.
#include "cmf_common.h"
#include <map>
#include <mutex>
#include <tuple>

namespace {

// SC_LPI lines per wave per iteration, SC_UB bands per load batch: SC_LPI*SC_UB loads are issued back to back,
// two batches in flight (template parameters; the default is picked in sf_launch_score).

// One batch of loads: SC_UB bands x SC_LPI lines of this lane's column.  Row pointers are wave-uniform
// (scalar base + per-lane 32-bit offset addressing); bands past the window are clamped to the last band
// (in-bounds duplicate, weighted by 0 below).
template <int SC_LPI, int SC_UB>
__device__ __forceinline__ void score_load(float (&x)[SC_LPI][SC_UB], const float *const (&lp)[SC_LPI], int bc, int p,
                                           int C, int lanec) {
#pragma unroll
  for (int bb = 0; bb < SC_UB; ++bb) {
    const int b = min(bc + bb, p - 1);
#pragma unroll
    for (int j = 0; j < SC_LPI; ++j) x[j][bb] = (lp[j] + (size_t)b * C)[lanec];
  }
}

template <int SC_LPI, int SC_UB>
__device__ __forceinline__ void score_fma(const float (&x)[SC_LPI][SC_UB], const double *__restrict__ ws, int wld, int bc,
                                          int p, int lane, double (&acc)[SC_LPI], bool (&ok)[SC_LPI]) {
#pragma unroll
  for (int bb = 0; bb < SC_UB; ++bb) {
    const int b = bc + bb;
    const double wv = (b < p) ? ws[(size_t)min(b, p - 1) * wld + lane] : 0.0;
#pragma unroll
    for (int j = 0; j < SC_LPI; ++j) {
      ok[j] = ok[j] & sf_valid(x[j][bb]);
      acc[j] = __builtin_fma((double)x[j][bb], wv, acc[j]);
    }
  }
}

template <int SC_LPI, int SC_UB>
__device__ __forceinline__ void score_sum(const float (&x)[SC_LPI][SC_UB], double (&acc)[SC_LPI]) {
#pragma unroll
  for (int bb = 0; bb < SC_UB; ++bb)
#pragma unroll
    for (int j = 0; j < SC_LPI; ++j) acc[j] += (double)x[j][bb];
}

// WGL: the 64 filter vectors are read from a transposed global copy wT[band][column] (L2-resident) instead of
// LDS -- for windows too wide for a [p][64] float64 LDS tile.
// CW: 64-column blocks per workgroup.  With CW = 2 the two halves of a 128-column block read ADJACENT 256-byte row
// segments at the same time from the same CU, so the 128-byte lines straddling their boundary are fetched once
// (the row stride, 2392 B, is not a multiple of the line size: every segment starts mid-line).
// TRAF (sf_debug_set(1, 200), bench.py `in_step_traffic_ms`): the launch's traffic with none of its work -- the same loads
// in the same order, the same staged record stores, but the "score" is the plain sum of the loaded values: no filter table
// (no LDS fill, no weight reads), no validity test, no statistics, no metadata image.  Wrong results by design; it exists so
// that the traffic bound can be timed in the kernel's own position inside the step.
template <bool RGB, int SC_LPI, int SC_UB, bool WGL, int CW = 1, bool STG = false, bool TRAF = false>
__global__ __launch_bounds__(256 * CW) void k_score(const float *__restrict__ cube, int L, int B, int C, int s0, int Cs,
                                                int b0, int p, const double *__restrict__ filt,
                                                const double *__restrict__ bias, const int32_t *__restrict__ status,
                                                const int32_t *__restrict__ alphaidx, int rgb0, int rgb1, int rgb2,
                                                double nodata, double *__restrict__ out, int oS, int os0,
                                                int16_t *__restrict__ bgmeta, double *__restrict__ stat_part,
                                                int lines_per_wg, int ncb, int nchunk, int xcdmap, const double *__restrict__ wT, int ldw) {
  extern __shared__ __attribute__((aligned(16))) double ws_all[];  // [CW][p][64]
  __shared__ double sred[4 * CW][64][2];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int half = wv >> 2, wave = wv & 3;
  int cbi, chunk;
  if (xcdmap) {
    if (!sf_xcd_map(blockIdx.x, ncb, nchunk, cbi, chunk)) return;
  } else {
    cbi = blockIdx.x % ncb;
    chunk = blockIdx.x / ncb;
    if (chunk >= nchunk) return;
  }
  const int colbase = min(cbi * 64 * CW + 64 * half, Cs - 1);   // (a second half past the last column idles on it)
  const int ncol = (cbi * 64 * CW + 64 * half < Cs) ? min(64, Cs - colbase) : 0;
  const bool colok = lane < ncol;
  const int lanec = colok ? lane : max(ncol - 1, 0);  // idle lanes re-read the last column (in bounds), never write
  const int col = colbase + lanec;
  double *ws = ws_all + (size_t)half * p * 64;

  if (!WGL && !TRAF) {
    for (int idx = tid; idx < CW * 64 * p; idx += 256 * CW) {
      const int hh = idx / (64 * p), r = idx - hh * (64 * p);
      const int cl = r / p, b = r - cl * p;
      const int cc = cbi * 64 * CW + 64 * hh + cl;
      ws_all[(size_t)hh * p * 64 + b * 64 + cl] = (cc < Cs) ? filt[(size_t)cc * p + b] : 0.0;
    }
  }
  const double *wsrc = WGL ? (wT + colbase + lanec - lane) : ws;   // wsrc[b*wld + lane] is this lane's weight
  const int wld = WGL ? ldw : 64;
  const double mybias = bias[col];
  const int st = status[col];
  const int ai = alphaidx[col];
  __syncthreads();

  const int lbeg = chunk * lines_per_wg, lend = min(L, lbeg + lines_per_wg);
  const size_t lstride = (size_t)B * C;
  const float *cb = cube + (size_t)(s0 + colbase);  // wave-uniform
  double s1 = 0.0, s2 = 0.0;

  for (int l = lbeg + wave * SC_LPI; l < lend; l += 4 * SC_LPI) {
    const int nl = min(SC_LPI, lend - l);
    const float *lp[SC_LPI];
#pragma unroll
    for (int j = 0; j < SC_LPI; ++j)  // tail lines alias the last line of the chunk (loaded, never written)
      lp[j] = cb + (size_t)min(l + j, lend - 1) * lstride + (size_t)b0 * C;
    double acc[SC_LPI];
    bool ok[SC_LPI];
#pragma unroll
    for (int j = 0; j < SC_LPI; ++j) { acc[j] = 0.0; ok[j] = true; }
    float xa[SC_LPI][SC_UB], xb[SC_LPI][SC_UB];
    score_load<SC_LPI, SC_UB>(xa, lp, 0, p, C, lanec);
    for (int bc = 0; bc < p; bc += 2 * SC_UB) {
      if (bc + SC_UB < p) score_load<SC_LPI, SC_UB>(xb, lp, bc + SC_UB, p, C, lanec);
      if (TRAF) score_sum<SC_LPI, SC_UB>(xa, acc); else score_fma<SC_LPI, SC_UB>(xa, wsrc, wld, bc, p, lane, acc, ok);
      if (bc + 2 * SC_UB < p) score_load<SC_LPI, SC_UB>(xa, lp, bc + 2 * SC_UB, p, C, lanec);
      if (bc + SC_UB < p) {
        if (TRAF) score_sum<SC_LPI, SC_UB>(xb, acc); else score_fma<SC_LPI, SC_UB>(xb, wsrc, wld, bc + SC_UB, p, lane, acc, ok);
      }
    }
    float rgbv[SC_LPI][3];
    if (RGB) {
#pragma unroll
      for (int j = 0; j < SC_LPI; ++j) {
        const float *pl = cb + (size_t)min(l + j, lend - 1) * lstride;
        rgbv[j][0] = (pl + (size_t)rgb0 * C)[lanec];
        rgbv[j][1] = (pl + (size_t)rgb1 * C)[lanec];
        rgbv[j][2] = (pl + (size_t)rgb2 * C)[lanec];
      }
    }
    if (RGB && STG) {
      // staged stores: the wave's 64 records [R, G, B, CMF] of a line are 2 KB of contiguous output; they pass through a
      // 1 KB LDS block of the wave (sred, free until the statistics at the end) in two halves and leave as 1 KB
      // contiguous store instructions instead of 16-byte pieces at a 32-byte stride
      typedef double d2v_t __attribute__((ext_vector_type(2)));
      d2v_t *stg = reinterpret_cast<d2v_t *>(&sred[wv][0][0]);
#pragma unroll
      for (int j = 0; j < SC_LPI; ++j) {
        if (j >= nl) break;
        const bool v = ok[j];
        const double sc = v ? ((st == 2) ? 0.0 : (acc[j] - mybias)) : nodata;
        if (!TRAF && v && colok) { s1 += sc; s2 += sc * sc; }
        const bool cp = st != 1;
        const d2v_t ra = {cp ? (double)rgbv[j][0] : 0.0, cp ? (double)rgbv[j][1] : 0.0};
        const d2v_t rb = {cp ? (double)rgbv[j][2] : 0.0, sc};
        d2v_t *orow = reinterpret_cast<d2v_t *>(out + ((size_t)(l + j) * oS + os0 + colbase) * 4);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if ((lane >> 5) == h) {
            const int s = lane & 31, sw = (s >> 3) & 1;
            stg[2 * s + (0 ^ sw)] = ra;
            stg[2 * s + (1 ^ sw)] = rb;
          }
          // lanes exchange data through the wave's LDS block: make the order explicit (no instruction is emitted: a
          // wave's LDS operations issue in order, but the memory model does not promise it to the compiler)
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const int s = lane >> 1;
          const d2v_t val = stg[2 * s + ((lane & 1) ^ ((s >> 3) & 1))];
          if (32 * h + s < ncol) orow[(size_t)(32 * h + s) * 2 + (lane & 1)] = val;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the reads above before the next half's writes
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (!TRAF && bgmeta && colok) {
          const uint32_t m = (v && st == 0) ? ((uint32_t)(uint16_t)(int16_t)ai << 16) : 0u;
          reinterpret_cast<uint32_t *>(bgmeta)[(size_t)(l + j) * oS + os0 + col] = m;
        }
      }
    } else
#pragma unroll
    for (int j = 0; j < SC_LPI; ++j) {
      if (j < nl && colok) {
        const bool v = ok[j];
        // status 2 (singular C): filt = bias = 0 -> the valid rows get exactly 0 (robust_mf.py:373)
        const double sc = v ? ((st == 2) ? 0.0 : (acc[j] - mybias)) : nodata;
        if (v) { s1 += sc; s2 += sc * sc; }
        const size_t pix = (size_t)(l + j) * oS + os0 + col;
        if (RGB) {
          // columns without a valid row are skipped before the RGB copy (:303-304)
          const double r = (st != 1) ? (double)rgbv[j][0] : 0.0;
          const double gg = (st != 1) ? (double)rgbv[j][1] : 0.0;
          const double bb = (st != 1) ? (double)rgbv[j][2] : 0.0;
          double2 *o = reinterpret_cast<double2 *>(out + pix * 4);
          o[0] = make_double2(r, gg);
          o[1] = make_double2(bb, sc);
        } else {
          out[pix] = sc;
        }
        if (bgmeta) {
          // int16 pair (cluster id = 0, alpha index); written only on valid rows of solved columns (:365)
          const uint32_t m = (v && st == 0) ? ((uint32_t)(uint16_t)(int16_t)ai << 16) : 0u;
          reinterpret_cast<uint32_t *>(bgmeta)[pix] = m;
        }
      }
    }
  }
  if (!TRAF && stat_part) {
    sred[wv][lane][0] = s1;
    sred[wv][lane][1] = s2;
    __syncthreads();
    if (wave == 0 && colok) {
      double a = 0.0, b = 0.0;
      for (int w = 0; w < 4; ++w) { a += sred[4 * half + w][lane][0]; b += sred[4 * half + w][lane][1]; }
      double *o = stat_part + ((size_t)chunk * Cs + col) * 2;
      o[0] = a;
      o[1] = b;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// helpers of k_score_blk2
// ---------------------------------------------------------------------------------------------------------------
typedef float f2a_t __attribute__((ext_vector_type(2)));
typedef unsigned u2_t __attribute__((ext_vector_type(2)));
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef double sc_d2_t __attribute__((ext_vector_type(2)));

// Cube loads are raw buffer loads: one descriptor per LPI-line batch (base = the batch's first line at the shard's
// first sample; the range ends with the batch's last line, reads past it return 0), the lane's byte offset in a
// VGPR and the (line, band) offset in an SGPR -- no 64-bit VALU address arithmetic and no per-line pointer registers.
template <bool NT>
__device__ __forceinline__ f2a_t sc_ld2(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  const u2_t v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, NT ? 2 : 0);   // aux 2 = nt
  f2a_t r;
  r.x = __uint_as_float(v.x);
  r.y = __uint_as_float(v.y);
  return r;
}
// Row validity ((~(x<0)) & isfinite(x), robust_mf.py:282) as ONE running unsigned maximum per pixel: x + 0.0f turns
// -0.0 (valid: not < 0) into +0.0 and leaves every other value alone, and then the valid values are exactly the bit
// patterns <= 0x7F7FFFFF (+0 .. FLT_MAX); negatives, infinities and NaNs of either sign are larger as unsigned integers.
__device__ __forceinline__ uint32_t sc_vkey(float x) { return __float_as_uint(x + 0.0f); }
constexpr uint32_t SC_VKEY_MAX = 0x7F7FFFFFu;

// ---------------------------------------------------------------------------------------------------------------
// k_score_blk2: 128-sample column blocks, two samples per lane (a measured alternative, not the default)
// ---------------------------------------------------------------------------------------------------------------
// Measured on the benchmark cube, loads only (tools/microbench/readbw2.hip, 598 x 20000 x 425, bands 350..421, every
// form with >= 256 KB in flight per CU): 64-sample blocks with 4-byte loads 5.3 TB/s, 128-sample blocks with 8-byte
// loads 6.0, 256-sample blocks with 16-byte loads 6.0, whole rows 6.0 -- and 7.0 for one contiguous run, which a
// strided window of a BIL cube is not.  What a form needs is (a) wide enough pieces and (b) bytes in flight, i.e.
// registers: whole rows per workgroup (git history: e35b16f) put the filter of ALL columns on the path (344 KB: LDS
// tiles with barriers, or L2 reads through registers) and lose (b); column blocks keep the block's filter resident in LDS for
// the workgroup's lifetime, with no barrier after the prologue.  So: 128-sample blocks, a lane owns two adjacent
// samples (8-byte loads, 512 contiguous bytes per wave instruction), [p][128] float64 filter tile (73.7 KB at p = 72:
// two workgroups = 8 waves per CU), 8 lines x 4 bands x 2 batches = 64 loads = 32 KB in flight per wave.
// A workgroup is persistent: one column block, a balanced contiguous range of 8-line batches dealt to its 4 waves;
// the grid is sized to be resident at once and all column blocks of a line range sit on one XCD (sf_xcd_map), so
// the 128-byte lines that straddle block boundaries are fetched by neighbours on the same L2 at about the same time.
// Validity is one running unsigned maximum per pixel (sc_vkey), the wave's records leave through a 2 KB LDS staging
// block as 1 KB contiguous stores, the statistics partials are per 8-line batch (independent of the launch shape).
constexpr int SB_LPI = 8, SB_UB = 4, SB_CB = 128;
static_assert(SB_UB >= 3, "the RGB rows reuse a load buffer");
// one 16-byte piece of the product.  mode (timing experiments only): 1 = nt, 2 = sc1 (write-through), 3 = sc0 sc1
__device__ __forceinline__ void sc_store16(sc_d2_t *p, sc_d2_t v, int mode) {
#ifdef SF_SCORE_EXPERIMENTS
  if (mode == 1) { __builtin_nontemporal_store(v, p); return; }
  if (mode == 2) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); return; }
  if (mode == 3) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); return; }
#endif
  (void)mode;
  *p = v;
}
static size_t sb_lds_bytes(int p) { return ((size_t)(p + 2 * SB_UB - 1) / (2 * SB_UB) * (2 * SB_UB) * SB_CB + 4 * 256) * sizeof(double); }
template <bool RGB, bool NT>
__global__ __launch_bounds__(256, 2) void k_score_blk2(const float *__restrict__ cube, int L, int B, int C, int s0, int Cs,
                                                        int b0, int p, const double *__restrict__ filt,
                                                        const double *__restrict__ bias,
                                                        const int32_t *__restrict__ status,
                                                        const int32_t *__restrict__ alphaidx, int rgb0, int rgb1,
                                                        int rgb2, double nodata, double *__restrict__ out, int oS,
                                                        int os0, int16_t *__restrict__ bgmeta,
                                                        double *__restrict__ stat_part, int ncb, int nk, int nbatch,
                                                        int expf) {
#ifndef SF_SCORE_EXPERIMENTS
  expf = 0;
#endif
  constexpr int LPI = SB_LPI, UB = SB_UB;
  extern __shared__ __attribute__((aligned(16))) double tile[];      // [pr][128] filter tile, then 4 x 2 KB staging
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int cbi, kk;
  if (!sf_xcd_map(blockIdx.x, ncb, nk, cbi, kk)) return;
  const int colbase = cbi * SB_CB, ncol = min(SB_CB, Cs - colbase);
  // the lane's pair: local columns 2 lane, 2 lane + 1.  A pair (half) past the block's last column is never written;
  // its loads stay inside the batch's buffer range (or return 0 past it) and its tile weights are 0.
  const int lc0 = 2 * lane;
  const bool own0 = lc0 < ncol, own1 = lc0 + 1 < ncol;
  const int c0 = colbase + lc0;
  const int pr = (p + 2 * UB - 1) / (2 * UB) * (2 * UB);      // tile rows: the window padded to whole load batches
  for (int idx = tid; idx < pr * SB_CB; idx += 256) {
    const int cl = idx / pr, b = idx - cl * pr;
    tile[b * SB_CB + cl] = (cl < ncol && b < p) ? filt[(size_t)(colbase + cl) * p + b] : 0.0;
  }
  const int c0c = min(c0, Cs - 1), c1c = min(c0 + 1, Cs - 1);
  const double bias0 = bias[c0c], bias1 = bias[c1c];
  const int st0 = status[c0c], st1 = status[c1c];
  const int ai0 = alphaidx[c0c], ai1 = alphaidx[c1c];
  const size_t lstride = (size_t)B * C;
  const unsigned lstride4 = (unsigned)lstride * 4u, C4 = (unsigned)C * 4u;
  const unsigned coff = (unsigned)lc0 * 4u;
  const double *wp = tile + lc0;
  sc_d2_t *stg = reinterpret_cast<sc_d2_t *>(tile + (size_t)pr * SB_CB) + 128 * wave;   // 2 KB per wave
  __syncthreads();

  const int bat_beg = (int)((long)kk * nbatch / nk), bat_end = (int)((long)(kk + 1) * nbatch / nk);
  auto batch_rsrc = [&](int sbx) {   // from the batch's first line at the block's first sample to the end of its last line
    const int lx = sbx * LPI, nlx = min(LPI, L - lx);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(cube + (size_t)lx * lstride + (size_t)s0 + colbase), 0,
                                             (unsigned)nlx * lstride4 - (unsigned)(s0 + colbase) * 4u, 0x00020000);
  };
  auto load = [&](f2a_t (&x)[LPI][UB], __amdgpu_buffer_rsrc_t rs, int bs) {
#pragma unroll
    for (int bb = 0; bb < UB; ++bb) {
      const unsigned boff = (unsigned)(b0 + min(bs + bb, p - 1)) * C4;   // bands past the window: duplicates, weight 0
#pragma unroll
      for (int j = 0; j < LPI; ++j) x[j][bb] = sc_ld2<NT>(rs, coff, boff + (unsigned)j * lstride4);
    }
  };
  int sb = bat_beg + wave;
  if (sb >= bat_end) return;
  __amdgpu_buffer_rsrc_t rsrc = batch_rsrc(sb);
  f2a_t xa[LPI][UB], xb[LPI][UB];
  load(xa, rsrc, 0);

  // One line batch.  On entry P holds the batch's first band batch (in flight); on exit Q holds the NEXT line batch's
  // first band batch: it is issued BEFORE this batch's epilogue, because vmcnt retires in order -- loads issued after
  // the epilogue's 32 stores could not be consumed until every one of those stores had been acknowledged (~10 us
  // behind the CU's queue), and the wave's stream would stand still that long once per batch.
  auto batch = [&](f2a_t (&P)[LPI][UB], f2a_t (&Q)[LPI][UB]) -> bool {
    const int l0 = sb * LPI;
    const int nl = min(LPI, L - l0);
    double acc[LPI][2];
    uint32_t vk[LPI][2];
#pragma unroll
    for (int j = 0; j < LPI; ++j) { acc[j][0] = acc[j][1] = 0.0; vk[j][0] = vk[j][1] = 0u; }
    auto fma = [&](const f2a_t (&x)[LPI][UB], int bs) {
      if (expf & 4) { acc[0][0] += (double)x[0][0].x + (double)x[LPI - 1][UB - 1].y; return; }
#pragma unroll
      for (int bb = 0; bb < UB; ++bb) {
        const double w0 = wp[(size_t)(bs + bb) * SB_CB], w1 = wp[(size_t)(bs + bb) * SB_CB + 1];   // rows >= p are zero
#pragma unroll
        for (int j = 0; j < LPI; ++j) {
          vk[j][0] = max(vk[j][0], sc_vkey(x[j][bb].x));
          vk[j][1] = max(vk[j][1], sc_vkey(x[j][bb].y));
          acc[j][0] = __builtin_fma((double)x[j][bb].x, w0, acc[j][0]);
          acc[j][1] = __builtin_fma((double)x[j][bb].y, w1, acc[j][1]);
        }
      }
    };
    for (int bc = 0; bc < pr; bc += 2 * UB) {
      load(Q, rsrc, bc + UB);
      fma(P, bc);
      if (bc + 2 * UB < pr) {
        load(P, rsrc, bc + 2 * UB);
      } else if (RGB) {
        // the last band batch is in flight and P is free: the three RGB rows of the batch's lines (whole-row pieces
        // again) queue up right behind it, in P's registers
#pragma unroll
        for (int j = 0; j < LPI; ++j) {
          P[j][0] = sc_ld2<NT>(rsrc, coff, (unsigned)rgb0 * C4 + (unsigned)j * lstride4);
          P[j][1] = sc_ld2<NT>(rsrc, coff, (unsigned)rgb1 * C4 + (unsigned)j * lstride4);
          P[j][2] = sc_ld2<NT>(rsrc, coff, (unsigned)rgb2 * C4 + (unsigned)j * lstride4);
        }
      }
      fma(Q, bc + UB);
    }
    const int sbn = sb + 4;
    const bool more = sbn < bat_end;
    if (more) {
      rsrc = batch_rsrc(sbn);
      load(Q, rsrc, 0);
    }
    if (!((expf & 1) && acc[0][0] != 1.2345e300)) {
      double s1[2] = {0.0, 0.0}, s2[2] = {0.0, 0.0};
      {
        const f2a_t (&rgbv)[LPI][UB] = P;
#pragma unroll
        for (int jj = 0; jj < LPI; ++jj) {
          const int j = jj;
          if (j >= nl) break;                                   // (wave-uniform)
          // status 2 (singular C): filt = bias = 0 -> the valid rows get exactly 0 (robust_mf.py:373)
          const bool ok0 = vk[j][0] <= SC_VKEY_MAX, ok1 = vk[j][1] <= SC_VKEY_MAX;
          const double sc0 = ok0 ? ((st0 == 2) ? 0.0 : (acc[j][0] - bias0)) : nodata;
          const double sc1 = ok1 ? ((st1 == 2) ? 0.0 : (acc[j][1] - bias1)) : nodata;
          if (ok0) { s1[0] += sc0; s2[0] += sc0 * sc0; }
          if (ok1) { s1[1] += sc1; s2[1] += sc1 * sc1; }
          const size_t pix = (size_t)(l0 + j) * oS + os0 + c0;
          if (expf & 8) continue;
          if (RGB) {
            // the wave's 128 records [R, G, B, CMF] of this line are 4 KB of contiguous output; they pass through the wave's
            // 2 KB staging block in two halves (32 lanes each) and leave as 1 KB contiguous store instructions.
            // Layout: 16-byte piece m of lane s' at 4 s' + (m ^ ((s' >> 2) & 3)) -- conflict-free both ways.
            // columns without a valid row are skipped before the RGB copy (:303-304)
            const bool cp0 = st0 != 1, cp1 = st1 != 1;
            const sc_d2_t r0a = {cp0 ? (double)rgbv[jj][0].x : 0.0, cp0 ? (double)rgbv[jj][1].x : 0.0};
            const sc_d2_t r0b = {cp0 ? (double)rgbv[jj][2].x : 0.0, sc0};
            const sc_d2_t r1a = {cp1 ? (double)rgbv[jj][0].y : 0.0, cp1 ? (double)rgbv[jj][1].y : 0.0};
            const sc_d2_t r1b = {cp1 ? (double)rgbv[jj][2].y : 0.0, sc1};
            sc_d2_t *orow = reinterpret_cast<sc_d2_t *>(out + ((size_t)(l0 + j) * oS + os0 + colbase) * 4);
            if (expf & 32)   // (experiment: the same bytes, but each wave's batch as ONE contiguous 32 KB run)
              orow = reinterpret_cast<sc_d2_t *>(out) + (((size_t)sb * ncb + cbi) * LPI + j) * 256;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              if ((lane >> 5) == h) {
                const int s = lane & 31, sw = (s >> 2) & 3;
                stg[4 * s + (0 ^ sw)] = r0a;
                stg[4 * s + (1 ^ sw)] = r0b;
                stg[4 * s + (2 ^ sw)] = r1a;
                stg[4 * s + (3 ^ sw)] = r1b;
              }
#pragma unroll
              for (int k = 0; k < 2; ++k) {
                const int P = 64 * k + lane;                    // 16-byte piece of this half's 2 KB
                const int s = P >> 2;
                const sc_d2_t v = stg[4 * s + ((P & 3) ^ ((s >> 2) & 3))];
                const int lcol = 64 * h + (P >> 1);
                if (lcol < ncol && !((expf & 64) && v.x != 1.2345e300)) sc_store16(orow + (size_t)lcol * 2 + (P & 1), v, expf >> 8);
              }
            }
          } else {
            if (own0) out[pix] = sc0;
            if (own1) out[pix + 1] = sc1;
          }
          if (bgmeta) {
            // int16 pair (cluster id = 0, alpha index); written only on valid rows of solved columns (:365)
            uint32_t *bm = reinterpret_cast<uint32_t *>(bgmeta) + pix;
            if (own0) bm[0] = (ok0 && st0 == 0) ? ((uint32_t)(uint16_t)(int16_t)ai0 << 16) : 0u;
            if (own1) bm[1] = (ok1 && st1 == 0) ? ((uint32_t)(uint16_t)(int16_t)ai1 << 16) : 0u;
          }
        }
      }
      if (stat_part) {
        double2 *o = reinterpret_cast<double2 *>(stat_part) + (size_t)sb * Cs + c0;
        if (own0) o[0] = make_double2(s1[0], s2[0]);
        if (own1) o[1] = make_double2(s1[1], s2[1]);
      }
    }
    sb = sbn;
    return more;
  };
  for (;;) {
    if (!batch(xa, xb)) break;
    if (!batch(xb, xa)) break;
  }
}

// npix / mean / std (ddof 0) of the written scores per column (robust_mf.py:388-392).  One 1024-thread
// workgroup per 64 columns: lane = column, the 16 waves split the chunk list (a shard has few columns but
// hundreds of chunks: the kernel is a chain of dependent-latency loads), fixed combination order.
__global__ __launch_bounds__(1024) void k_colstats(const double *__restrict__ stat_part, int nchunk, int Cs,
                                                    const int32_t *__restrict__ nuse, const int32_t *__restrict__ status,
                                                    double nodata, double *__restrict__ colstats) {
  __shared__ double red[16][64][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  double a = 0.0, b = 0.0;
  if (c < Cs) {
    const double2 *sp = reinterpret_cast<const double2 *>(stat_part);
#pragma unroll 4
    for (int k = wave; k < nchunk; k += 16) {
      const double2 v = sp[(size_t)k * Cs + c];
      a += v.x;
      b += v.y;
    }
  }
  red[wave][lane][0] = a;
  red[wave][lane][1] = b;
  __syncthreads();
  if (wave != 0 || c >= Cs) return;
  if (status[c] == 1) {  // column skipped: stats keep their initial value (:293-295)
    colstats[c] = nodata; colstats[Cs + c] = nodata; colstats[2 * Cs + c] = nodata;
    return;
  }
  a = 0.0; b = 0.0;
  for (int w = 0; w < 16; ++w) { a += red[w][lane][0]; b += red[w][lane][1]; }
  const double n = (double)nuse[c];
  const double mean = a / n;
  double var = b / n - mean * mean;
  if (var < 0.0) var = 0.0;
  colstats[c] = n;
  colstats[Cs + c] = mean;
  colstats[2 * Cs + c] = sqrt(var);
}

// ---- column profile of a finished CMF product (triage/cmf_profile.py:110-140, non-robust statistics) ----------
// Over the pixels that are valid (not NODATA, not NaN) AND positive: npix, mean, std (ddof 0), min, max per column.
// lane = column, workgroups split the lines; partials [chunk][col][5] combined in a fixed order.
__global__ __launch_bounds__(256) void k_profile(const double *__restrict__ img, int L, int S, int nb, int band,
                                                  double nodata, int lines_per_wg, double *__restrict__ part) {
  __shared__ double red[4][64][5];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int lbeg = blockIdx.y * lines_per_wg, lend = min(L, lbeg + lines_per_wg);
  double n = 0, s1 = 0, s2 = 0, mn = __builtin_inf(), mx = -__builtin_inf();
  if (c < S)
    for (int l = lbeg + wave; l < lend; l += 4) {
      const double v = img[((size_t)l * S + c) * nb + band];
      const double vf = (double)(float)v;                 // the reference profiles the float32 cast of the product
      if (v == v && v != nodata && vf > 0.0) { n += 1; s1 += vf; s2 += vf * vf; mn = fmin(mn, vf); mx = fmax(mx, vf); }
    }
  red[wave][lane][0] = n; red[wave][lane][1] = s1; red[wave][lane][2] = s2; red[wave][lane][3] = mn; red[wave][lane][4] = mx;
  __syncthreads();
  if (wave == 0 && c < S) {
    for (int w = 1; w < 4; ++w) {
      n += red[w][lane][0]; s1 += red[w][lane][1]; s2 += red[w][lane][2];
      mn = fmin(mn, red[w][lane][3]); mx = fmax(mx, red[w][lane][4]);
    }
    double *o = part + ((size_t)blockIdx.y * S + c) * 5;
    o[0] = n; o[1] = s1; o[2] = s2; o[3] = mn; o[4] = mx;
  }
}
__global__ void k_profile_finish(const double *__restrict__ part, int nchunk, int S, double *__restrict__ prof) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= S) return;
  double n = 0, s1 = 0, s2 = 0, mn = __builtin_inf(), mx = -__builtin_inf();
  for (int k = 0; k < nchunk; ++k) {
    const double *o = part + ((size_t)k * S + c) * 5;
    n += o[0]; s1 += o[1]; s2 += o[2]; mn = fmin(mn, o[3]); mx = fmax(mx, o[4]);
  }
  const double nanv = __builtin_nan("");
  const double mean = n > 0 ? s1 / n : nanv;
  double var = n > 0 ? s2 / n - mean * mean : nanv;
  if (var < 0) var = 0;
  prof[c] = n;
  prof[S + c] = mean;
  prof[2 * S + c] = sqrt(var);
  prof[3 * S + c] = n > 0 ? mn : nanv;
  prof[4 * S + c] = n > 0 ? mx : nanv;
}

}  // namespace

static size_t score_stat_bytes(int lines, int ncols) {
  const int lpw = 8;  // upper bound on the number of line chunks whatever the kernel / tuning
  return sf_align((size_t)sf_cdiv(lines, lpw) * ncols * 2 * sizeof(double));
}
constexpr int SC_WT_ROWS = 512 + 64;   // widest supported window + one band group of padding
size_t sf_score_scratch_bytes(int lines, int ncols) {
  // statistics partials + the transposed, zero-padded filter
  return score_stat_bytes(lines, ncols) + sf_align((size_t)SC_WT_ROWS * ((ncols + 63) / 64 * 64) * sizeof(double));
}

// filt[c][p] -> wT[b][ldw], rows p..nrows-1 and columns Cs..ldw-1 zero
__global__ void k_filt_transpose(const double *__restrict__ filt, int Cs, int p, int ldw, int nrows, double *__restrict__ wT) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrows * ldw) return;
  const int b = i / ldw, c = i - b * ldw;
  wT[i] = (c < Cs && b < p) ? filt[(size_t)c * p + b] : 0.0;
}

template <bool RGB, int LPI, int UB, bool WGL = false, int CW = 1, bool STG = false, bool TRAF = false>
int launch_score_t(const float *cube, int lines, int bands, int samples, int s0, int ncols, int b0, int p,
                   const double *filt, const double *bias, const int32_t *status, const int32_t *alphaidx, int rgb0,
                   int rgb1, int rgb2, double nodata, double *out, int out_samples, int out_s0, int16_t *bgmeta,
                   double *stat_part, int lpw, hipStream_t st, hipEvent_t ea, hipEvent_t eb, const double *wT = nullptr,
                   int ldw = 0) {
  const size_t lds = WGL ? 0 : (size_t)CW * p * 64 * sizeof(double);
  if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_score<RGB, LPI, UB, WGL, CW, STG, TRAF>), lds)) return rc;
  const int nchunk = sf_cdiv(lines, lpw);
  const int ncb = sf_cdiv(ncols, 64 * CW);
  const int xcd = sf_tune().score_xcd;
  const int nblk = xcd ? sf_xcd_grid(ncb, nchunk) : ncb * nchunk;
  if (ea) SF_HIP(hipEventRecord(ea, st));
  hipLaunchKernelGGL((k_score<RGB, LPI, UB, WGL, CW, STG, TRAF>), dim3(nblk), dim3(256 * CW), lds, st, cube, lines, bands, samples, s0, ncols, b0,
                     p, filt, bias, status, alphaidx, rgb0, rgb1, rgb2, nodata, out, out_samples, out_s0, bgmeta, stat_part,
                     lpw, ncb, nchunk, xcd, wT, ldw);
  SF_LAUNCH_CHECK("k_score");
  if (eb) SF_HIP(hipEventRecord(eb, st));
  return 0;
}

// ---- launch plan: which kernel scores a shard, and with which line granularity of the statistics partials --------
// A function of the geometry and the calling thread's tuning knobs only (sf_launch_colstats must agree with it).
namespace {
struct ScorePlan {
  int kernel;                        // 0: k_score_blk2, 2: k_score (production)
  int lpw;                           // lines per statistics partial
};
ScorePlan score_plan(int lines, int ncols, int p) {
  const int v = sf_tune().score_variant;
  ScorePlan pl;
  pl.kernel = ((v == 10 || v == 11) && sb_lds_bytes(p) <= 160 * 1024) ? 0 : 2;
  pl.lpw = pl.kernel == 0 ? SB_LPI : (sf_tune().score_lpw > 0 ? sf_tune().score_lpw : sf_score_lines_per_wg(lines, ncols));
  return pl;
}

// workgroups of `fn` (nthr threads, lds bytes of dynamic LDS) that are resident at once on the current device
int resident_wgs(const void *fn, int nthr, size_t lds, int *out) {
  static std::mutex mu;
  static std::map<std::tuple<int, const void *, int, size_t>, int> have;
  int dev = 0;
  SF_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  auto key = std::make_tuple(dev, fn, nthr, lds);
  auto it = have.find(key);
  if (it == have.end()) {
    int per_cu = 0, cus = 0;
    SF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, nthr, lds));
    SF_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (per_cu < 1) per_cu = 1;
    it = have.emplace(key, per_cu * cus).first;
  }
  *out = it->second;
  return 0;
}

}  // namespace

int sf_score_lpw(int lines, int ncols, int p) { return score_plan(lines, ncols, p).lpw; }

#define SC_ARGS cube, lines, bands, samples, s0, ncols, b0, p, filt, bias, status, alphaidx, rgb0, rgb1, rgb2, nodata, out, \
                out_samples, out_s0, bgmeta, stat_part, lpw, st, ea, eb
int sf_launch_score(const float *cube, int lines, int bands, int samples, int s0, int ncols, int b0, int p,
                    const double *filt, const double *bias, const int32_t *status, const int32_t *alphaidx,
                    int rgb0, int rgb1, int rgb2, double nodata, double *out, int out_samples, int out_s0,
                    int out_bands, int16_t *bgmeta, void *scratch, int want_stats, hipStream_t st, hipEvent_t ea,
                    hipEvent_t eb) {
  const ScorePlan pl = score_plan(lines, ncols, p);
  const int lpw = pl.lpw;
  double *stat_part = (scratch && want_stats) ? reinterpret_cast<double *>(scratch) : nullptr;
  double *wT = scratch ? reinterpret_cast<double *>(reinterpret_cast<char *>(scratch) + score_stat_bytes(lines, ncols)) : nullptr;
  const bool rgb = out_bands == 4;
  if (pl.kernel == 0 && (size_t)bands * samples * 4 * SB_LPI >= ((size_t)1 << 32)) {
    sf_set_error("score kernel: %d lines of %d x %d values exceed a buffer descriptor", SB_LPI, bands, samples);
    return -2;
  }
  if (pl.kernel == 0) {
    const size_t lds = sb_lds_bytes(p);
    const bool nt = sf_tune().score_variant == 11;            // (experiment: non-temporal loads)
    const void *fn = rgb ? (nt ? (const void *)k_score_blk2<true, true> : (const void *)k_score_blk2<true, false>)
                         : (nt ? (const void *)k_score_blk2<false, true> : (const void *)k_score_blk2<false, false>);
    if (int rc = sf_lds_attr(fn, lds)) return rc;
    int slots = 0;
    if (int rc = resident_wgs(fn, 256, lds, &slots)) return rc;
    if (sf_tune().score_wgs > 0) slots = sf_tune().score_wgs * 256;
    const int ncb = sf_cdiv(ncols, SB_CB), nbatch = sf_cdiv(lines, SB_LPI);
    int nk = slots / (8 * ncb) * 8;                           // every workgroup resident, whole XCD rounds
    if (nk < 8) nk = 8;
    if (nk > sf_cdiv(nbatch, 4)) nk = sf_cdiv(nbatch, 4);     // at least one batch per wave
    if (nk < 1) nk = 1;
    const int grid = sf_xcd_grid(ncb, nk);
    if (ea) SF_HIP(hipEventRecord(ea, st));
#define SB_LAUNCH(RGBV, NTV)                                                                                          \
    hipLaunchKernelGGL((k_score_blk2<RGBV, NTV>), dim3(grid), dim3(256), lds, st, cube, lines, bands, samples, s0, ncols,  \
                       b0, p, filt, bias, status, alphaidx, rgb0, rgb1, rgb2, nodata, out, out_samples, out_s0, bgmeta,     \
                       stat_part, ncb, nk, nbatch, sf_tune().score_exp)
    if (rgb) { if (nt) SB_LAUNCH(true, true); else SB_LAUNCH(true, false); }
    else { if (nt) SB_LAUNCH(false, true); else SB_LAUNCH(false, false); }
#undef SB_LAUNCH
    SF_LAUNCH_CHECK("k_score_blk2");
    if (eb) SF_HIP(hipEventRecord(eb, st));
    return 0;
  }
  // ---- column-block kernel (round 1)
  if ((size_t)p * 64 * sizeof(double) > 100 * 1024) {  // wide window: filter from a transposed global copy
    if (!scratch) { sf_set_error("score kernel: a window of %d bands needs scratch", p); return -1; }
    const int ldw = (ncols + 63) / 64 * 64;
    hipLaunchKernelGGL(k_filt_transpose, dim3(sf_cdiv(p * ldw, 256)), dim3(256), 0, st, filt, ncols, p, ldw, p, wT);
    SF_LAUNCH_CHECK("k_filt_transpose");
    if (!rgb) return launch_score_t<false, 4, 8, true>(SC_ARGS, wT, ldw);
    return launch_score_t<true, 4, 8, true>(SC_ARGS, wT, ldw);
  }
  if (!rgb) return launch_score_t<false, 8, 4>(SC_ARGS);
  switch (sf_tune().score_variant) {
    case 1: return launch_score_t<true, 2, 16>(SC_ARGS);
    case 3: return launch_score_t<true, 4, 4>(SC_ARGS);
    case 4: return launch_score_t<true, 2, 8>(SC_ARGS);
    case 5: return launch_score_t<true, 8, 8>(SC_ARGS);
    case 6: return launch_score_t<true, 4, 8>(SC_ARGS);
    case 7: return launch_score_t<true, 8, 4, false, 2>(SC_ARGS);
    case 8: return launch_score_t<true, 4, 8, false, 2>(SC_ARGS);
    case 9: return launch_score_t<true, 4, 4, false, 2>(SC_ARGS);
    case 100: return launch_score_t<true, 8, 4>(SC_ARGS);                    // round 1: 16-byte pieces stored by the lanes
    case 200: return launch_score_t<true, 8, 4, false, 1, true, true>(SC_ARGS);   // the production launch's traffic only (timing)
    default: return launch_score_t<true, 8, 4, false, 1, true>(SC_ARGS);     // 8 lines x 4 bands per batch, staged stores
  }
}
#undef SC_ARGS

int sf_launch_colstats(const void *stat_scratch, int lines, int samples, int s0, int ncols, int p, const int32_t *nuse,
                       const int32_t *status, double nodata, double *colstats, hipStream_t st) {
  (void)samples; (void)s0;
  const int lpw = sf_score_lpw(lines, ncols, p);
  const int nchunk = sf_cdiv(lines, lpw);
  hipLaunchKernelGGL(k_colstats, dim3(sf_cdiv(ncols, 64)), dim3(1024), 0, st,
                     reinterpret_cast<const double *>(stat_scratch), nchunk, ncols, nuse, status, nodata, colstats);
  SF_LAUNCH_CHECK("k_colstats");
  return 0;
}

// ---- robust column profile (triage/cmf_profile.py:124-127, use_robust_stats): median, MAD, 5th / 95th percentile
// ('nearest') of a column's valid positive pixels.  One 1024-thread workgroup per column: the values (float32,
// as the reference casts them) are gathered into LDS, bitonic-sorted, and the order statistics read off; the
// MAD is a second sort of |x - median|.  Holds up to 32768 lines.
namespace {
constexpr int PR_NT = 1024, PR_CAP = 32768;

__device__ __forceinline__ void lds_bitonic_sort(float *a, int npow2, int tid) {
  for (int k = 2; k <= npow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npow2; i += PR_NT) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const float x = a[i], y = a[ixj];
          const bool up = (i & k) == 0;
          if ((x > y) == up) { a[i] = y; a[ixj] = x; }
        }
      }
      __syncthreads();
    }
  }
}
// numpy's method='nearest' on a float32 array (numpy >= 2: NEP 50 keeps the percentile arithmetic in the array's dtype):
// quantile = float32(q) / float32(100), virtual index = float32(n - 1) * quantile rounded to float32, then rint
// (half to even).  In float64 the 5th percentile of 2451 values is index 123 (122.50000000000011), numpy takes 122.
__device__ __forceinline__ int nearest_index(int n, float qpercent) {
  const float quant = qpercent / 100.0f;
  const float v = (float)(n - 1) * quant;
  int idx = (int)rintf(v);
  return idx < 0 ? 0 : (idx >= n ? n - 1 : idx);
}

__global__ __launch_bounds__(PR_NT) void k_profile_robust(const double *__restrict__ img, int L, int S, int nb, int band,
                                                           double nodata, float plo, float phi,
                                                           double *__restrict__ prof) {
  extern __shared__ float vals[];   // [npow2]
  __shared__ int cnt;
  __shared__ float smed;
  const int c = blockIdx.x, tid = threadIdx.x;
  int npow2 = 1;
  while (npow2 < L) npow2 <<= 1;
  if (tid == 0) cnt = 0;
  for (int i = tid; i < npow2; i += PR_NT) vals[i] = __builtin_inff();   // +inf pads sort to the end
  __syncthreads();
  for (int l = tid; l < L; l += PR_NT) {
    const double v = img[((size_t)l * S + c) * nb + band];
    const float vf = (float)v;
    if (v == v && v != nodata && vf > 0.f) vals[atomicAdd(&cnt, 1)] = vf;   // order is irrelevant: sorted next
  }
  __syncthreads();
  const int n = cnt;
  const double nanv = __builtin_nan("");
  if (n == 0) {
    if (tid == 0) { prof[c] = 0; prof[S + c] = nanv; prof[2 * S + c] = nanv; prof[3 * S + c] = nanv; prof[4 * S + c] = nanv; }
    return;
  }
  lds_bitonic_sort(vals, npow2, tid);
  if (tid == 0) {
    const float med = (n & 1) ? vals[n / 2] : (vals[n / 2 - 1] + vals[n / 2]) * 0.5f;   // float32 mean of the two
    smed = med;
    prof[c] = (double)n;
    prof[S + c] = (double)med;
    prof[3 * S + c] = (double)vals[nearest_index(n, plo)];
    prof[4 * S + c] = (double)vals[nearest_index(n, phi)];
  }
  __syncthreads();
  const float med = smed;
  for (int i = tid; i < n; i += PR_NT) vals[i] = fabsf(vals[i] - med);
  __syncthreads();
  lds_bitonic_sort(vals, npow2, tid);
  if (tid == 0) prof[2 * S + c] = (double)((n & 1) ? vals[n / 2] : (vals[n / 2 - 1] + vals[n / 2]) * 0.5f);
}
}  // namespace

extern "C" int sf_cmf_column_profile_robust(const double *img, int lines, int samples, int nbands, int band, double nodata,
                                            double p, double *profile, void *stream) {
  if (!img || !profile || lines < 1 || samples < 1 || band < 0 || band >= nbands || !(p > 0.0 && p < 1.0)) {
    sf_set_error("sf_cmf_column_profile_robust: bad argument");
    return -1;
  }
  if (lines > PR_CAP) {
    sf_set_error("sf_cmf_column_profile_robust: %d lines exceed the LDS-resident sort (max %d)", lines, PR_CAP);
    return -2;
  }
  int npow2 = 1;
  while (npow2 < lines) npow2 <<= 1;
  const size_t lds = (size_t)npow2 * sizeof(float);
  if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_profile_robust), lds)) return rc;
  // the reference's percentile arguments are computed in float64, (1-p)*100 and p*100 (srcfinder_util.py:652-653);
  // numpy divides them by 100 in the array's float32
  const float plo = (float)((1.0 - p) * 100.0), phi = (float)(p * 100.0);
  hipLaunchKernelGGL(k_profile_robust, dim3(samples), dim3(PR_NT), lds, (hipStream_t)stream, img, lines, samples, nbands, band,
                     nodata, plo, phi, profile);
  SF_LAUNCH_CHECK("k_profile_robust");
  return 0;
}

extern "C" int sf_cmf_column_profile(const double *img, int lines, int samples, int nbands, int band, double nodata,
                                     double *profile, void *scratch, void *stream) {
  if (!img || !profile || !scratch || lines < 1 || samples < 1 || band < 0 || band >= nbands) {
    sf_set_error("sf_cmf_column_profile: bad argument");
    return -1;
  }
  const int lpw = 256, nchunk = sf_cdiv(lines, lpw);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_profile, dim3(sf_cdiv(samples, 64), nchunk), dim3(256), 0, st, img, lines, samples, nbands, band,
                     nodata, lpw, reinterpret_cast<double *>(scratch));
  SF_LAUNCH_CHECK("k_profile");
  hipLaunchKernelGGL(k_profile_finish, dim3(sf_cdiv(samples, 128)), dim3(128), 0, st,
                     reinterpret_cast<const double *>(scratch), nchunk, samples, profile);
  SF_LAUNCH_CHECK("k_profile_finish");
  return 0;
}
