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
// (k_score_blk2, round 2's measured alternative -- 128-sample blocks, a lane owns two adjacent samples -- was removed in round 5:
//  tools/experiments/score_blk2_r02.hip.txt.)
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
  int lpw;                           // lines per statistics partial
};
ScorePlan score_plan(int lines, int ncols, int p) {
  (void)p;
  ScorePlan pl;
  pl.lpw = sf_tune().score_lpw > 0 ? sf_tune().score_lpw : sf_score_lines_per_wg(lines, ncols);
  return pl;
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
// MAD is a second sort of |x - median|.  The LDS sort holds up to 32768 lines; longer columns (the reference has no cap:
// triage/cmf_profile.py:124-127) take k_profile_robust_big: the same order statistics by radix selection on the float32 bit
// patterns straight from the product in global memory (positive floats order like their bits) -- exact, so both kernels
// give the same numbers.
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
// ---- any number of lines: the k-th smallest of a column's valid values, mode 0 = the float32 values themselves, mode 1 =
// |x - med|, by four 8-bit passes from the most significant byte down (a 256-bin histogram of the values that match the
// prefix found so far, in LDS; thread 0 walks the bins).  Nothing is stored: every pass re-reads the column from the product.
__device__ __forceinline__ bool pr_key(const double *__restrict__ img, size_t idx, double nodata, int mode, float med, unsigned &bits) {
  const double v = img[idx];
  const float vf = (float)v;
  const bool ok = (v == v) && v != nodata && vf > 0.f;
  bits = __float_as_uint(mode ? fabsf(vf - med) : vf);
  return ok;
}
__device__ float pr_select(const double *__restrict__ img, int L, int S, int nb, int band, int c, double nodata, int mode, float med,
                           unsigned k, unsigned *hist, unsigned *sh) {
  const int tid = threadIdx.x;
  unsigned prefix = 0, mask = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = tid; i < 256; i += PR_NT) hist[i] = 0;
    __syncthreads();
    for (int l = tid; l < L; l += PR_NT) {
      unsigned b;
      if (pr_key(img, ((size_t)l * S + c) * nb + band, nodata, mode, med, b) && (b & mask) == prefix) atomicAdd(&hist[(b >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned cum = 0, bsel = 255;
      for (unsigned b = 0; b < 256; ++b) {
        if (k < cum + hist[b]) { bsel = b; break; }
        cum += hist[b];
      }
      sh[0] = prefix | (bsel << shift);
      sh[1] = k - cum;
    }
    __syncthreads();
    prefix = sh[0];
    k = sh[1];
    mask |= 255u << shift;
    __syncthreads();
  }
  return __uint_as_float(prefix);
}

__global__ __launch_bounds__(PR_NT) void k_profile_robust_big(const double *__restrict__ img, int L, int S, int nb, int band,
                                                               double nodata, float plo, float phi, double *__restrict__ prof) {
  __shared__ unsigned hist[256];
  __shared__ unsigned sh[2];
  __shared__ int cnt;
  const int c = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) cnt = 0;
  __syncthreads();
  int mine = 0;
  for (int l = tid; l < L; l += PR_NT) {
    unsigned b;
    mine += pr_key(img, ((size_t)l * S + c) * nb + band, nodata, 0, 0.f, b) ? 1 : 0;
  }
  atomicAdd(&cnt, mine);
  __syncthreads();
  const int n = cnt;
  const double nanv = __builtin_nan("");
  if (n == 0) {
    if (tid == 0) { prof[c] = 0; prof[S + c] = nanv; prof[2 * S + c] = nanv; prof[3 * S + c] = nanv; prof[4 * S + c] = nanv; }
    return;
  }
  auto sel = [&](int mode, float med, int k) { return pr_select(img, L, S, nb, band, c, nodata, mode, med, (unsigned)k, hist, sh); };
  const float hi = sel(0, 0.f, n / 2);
  const float med = (n & 1) ? hi : (sel(0, 0.f, n / 2 - 1) + hi) * 0.5f;   // float32 mean of the two middle values
  const float vlo = sel(0, 0.f, nearest_index(n, plo)), vhi = sel(0, 0.f, nearest_index(n, phi));
  const float mhi = sel(1, med, n / 2);
  const float mad = (n & 1) ? mhi : (sel(1, med, n / 2 - 1) + mhi) * 0.5f;
  if (tid == 0) {
    prof[c] = (double)n;
    prof[S + c] = (double)med;
    prof[2 * S + c] = (double)mad;
    prof[3 * S + c] = (double)vlo;
    prof[4 * S + c] = (double)vhi;
  }
}
}  // namespace

extern "C" int sf_cmf_column_profile_robust(const double *img, int lines, int samples, int nbands, int band, double nodata,
                                            double p, double *profile, void *stream) {
  if (!img || !profile || lines < 1 || samples < 1 || band < 0 || band >= nbands || !(p > 0.0 && p < 1.0)) {
    sf_set_error("sf_cmf_column_profile_robust: bad argument");
    return -1;
  }
  const float plo_ = (float)((1.0 - p) * 100.0), phi_ = (float)(p * 100.0);
  if (lines > PR_CAP) {   // longer than the LDS sort holds: radix selection from global memory (the same numbers)
    hipLaunchKernelGGL(k_profile_robust_big, dim3(samples), dim3(PR_NT), 0, (hipStream_t)stream, img, lines, samples, nbands, band,
                       nodata, plo_, phi_, profile);
    SF_LAUNCH_CHECK("k_profile_robust_big");
    return 0;
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
