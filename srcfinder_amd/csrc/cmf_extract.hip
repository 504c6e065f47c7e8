// Stage 1+2 of the column matched filter: column extract + valid-row mask, masked column mean.
//
// Replaces cmf/robust_mf.py:298-302 (strided column view, useidx, float64 promotion) and :347 (mean).
// The BIL cube keeps adjacent SAMPLES adjacent in memory, so one detector column's spectra are strided
// by `samples` floats.  The statistics kernels want one column's (lines x p) matrix in front of one
// workgroup, so this pass transposes the active window once into column-major xt[col][line][ps]
// through LDS: global reads are 256-byte rows (64 samples x 4 B) of one (line, band); global writes are
// 256-byte runs of one column's consecutive (line, band) values.  HBM-bound: 4p B read + 4p B written
// per pixel.
#include "cmf_common.h"
#include <type_traits>

namespace {

constexpr int XT_PBMAX = 84;  // band chunk held in LDS at once (84: the CO2 window, p = 83, is one chunk and its column sums are fused)
constexpr int XT_NB = 18;     // loads in flight per wave

// LDS: tile[64 columns][cs] floats, cs odd -> both the column-strided writes (lane = column) and the
// row-contiguous reads (lane = element) are bank-conflict free.  TL = lines per tile.
// FUSE_SUM (single band chunk only): the masked column sums of stage 2 are accumulated here, while the
// tile is still in LDS, and written as per-chunk partials -- saves re-reading xt (4p B per pixel).
template <int TL, bool FUSE_SUM>
__global__ __launch_bounds__(256) void k_extract(const float *__restrict__ cube, int L, int B, int C, int s0,
                                                  int Cs, int b0, int p, int PS, float *__restrict__ xt,
                                                  uint8_t *__restrict__ mask_t, int lines_per_wg, int pbmax, int cs, int ncb,
                                                  int nchunk, double *__restrict__ sum_part, int *__restrict__ cnt_part) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  __shared__ uint8_t vf[64][4];
  constexpr int NSUM = FUSE_SUM ? (XT_PBMAX + 3) / 4 : 1;  // bands per wave: wave, wave+4, ...
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int cbi, chunk;
  if (!sf_xcd_map(blockIdx.x, ncb, nchunk, cbi, chunk)) return;
  const int colbase = cbi * 64;
  const int ncol = min(64, Cs - colbase);
  const bool colok = lane < ncol;
  const int lbeg = chunk * lines_per_wg;
  const int lend = min(L, lbeg + lines_per_wg);
  const int lanec = colok ? lane : ncol - 1;
  const float *cbase = cube + (size_t)(s0 + colbase);  // wave-uniform base, per-lane 32-bit offset
  double sums[NSUM];
  int nvalid = 0;
#pragma unroll
  for (int i = 0; i < NSUM; ++i) sums[i] = 0.0;

  for (int l0 = lbeg; l0 < lend; l0 += TL) {
    const int nl = min(TL, lend - l0);
    bool ok = true;  // thread (line = wave, column = lane)
    for (int bc0 = 0; bc0 < p; bc0 += pbmax) {
      const int pb = min(pbmax, p - bc0);
      const int nrows = nl * pb;  // (line, band) rows of 64 samples in this chunk
      // ---- global -> LDS, XT_NB rows (256 B each) in flight per wave
      for (int base = wave; base < nrows; base += 4 * XT_NB) {
        float v[XT_NB];
#pragma unroll
        for (int u = 0; u < XT_NB; ++u) {
          const int rr = min(base + 4 * u, nrows - 1);      // clamped duplicates are loaded but not stored
          const int l = rr / pb, b = rr - l * pb;
          v[u] = (cbase + ((size_t)(l0 + l) * B + (b0 + bc0 + b)) * C)[lanec];
        }
#pragma unroll
        for (int u = 0; u < XT_NB; ++u) {
          const int rr = base + 4 * u;
          if (rr < nrows) tile[lane * cs + rr] = v[u];
        }
      }
      __syncthreads();
      // ---- validity of (line = wave, column = lane) over this band chunk
      if (wave < nl) {
        const float *tp = tile + lane * cs + wave * pb;
        for (int b = 0; b < pb; ++b) ok = ok & sf_valid(tp[b]);
        if (bc0 + pb >= p) vf[lane][wave] = ok ? 1 : 0;
      }
      if (FUSE_SUM) __syncthreads();  // vf of this tile is complete
      // ---- LDS -> xt; wave handles columns wave, wave+4, ...
      if (pb == PS) {  // single chunk, no padding: nl*PS contiguous floats per column
        const int nel = nl * PS;
        for (int c = wave; c < ncol; c += 4) {
          float *dst = xt + ((size_t)(colbase + c) * L + l0) * PS;
          const float *src = tile + c * cs;
          for (int k = lane; k < nel; k += 64) dst[k] = src[k];
        }
      } else {
        const bool last = (bc0 + pb >= p);
        const int wid = last ? (PS - bc0) : pb;  // last chunk also writes the zero padding p..PS-1
        for (int c = wave; c < ncol; c += 4) {
          for (int l = 0; l < nl; ++l) {
            float *dst = xt + ((size_t)(colbase + c) * L + l0 + l) * PS + bc0;
            const float *src = tile + c * cs + l * pb;
            for (int b = lane; b < wid; b += 64) dst[b] = (b < pb) ? src[b] : 0.f;
          }
        }
      }
      if (FUSE_SUM) {
        // ---- masked sums: lane = column, this wave's bands wave, wave+4, ...; lines in fixed order
        bool vl[TL];
#pragma unroll
        for (int l = 0; l < TL; ++l) vl[l] = (l < nl) && vf[lane][l];
        if (wave == 0) {
#pragma unroll
          for (int l = 0; l < TL; ++l) nvalid += vl[l] ? 1 : 0;
        }
        const float *tp = tile + lane * cs;
#pragma unroll
        for (int i = 0; i < NSUM; ++i) {
          const int b = wave + 4 * i;
          if (b < pb) {
#pragma unroll
            for (int l = 0; l < TL; ++l)
              if (vl[l]) sums[i] += (double)tp[l * pb + b];
          }
        }
      }
      __syncthreads();
    }
    // ---- mask: the tile's line flags of a column, one 32-bit store where possible
    if (!FUSE_SUM) __syncthreads();
    if (wave == 0 && colok) {
      uint8_t *mp = mask_t + (size_t)(colbase + lane) * L + l0;
      if (TL == 4 && nl == 4 && (((size_t)(colbase + lane) * L + l0) & 3) == 0) {
        *reinterpret_cast<uint32_t *>(mp) = *reinterpret_cast<const uint32_t *>(&vf[lane][0]);
      } else if (TL == 2 && nl == 2 && (((size_t)(colbase + lane) * L + l0) & 1) == 0) {
        *reinterpret_cast<uint16_t *>(mp) = *reinterpret_cast<const uint16_t *>(&vf[lane][0]);
      } else {
        for (int l = 0; l < nl; ++l) mp[l] = vf[lane][l];
      }
    }
    // vf is rewritten only after the next tile's barriers
  }
  if (FUSE_SUM && colok) {
    double *o = sum_part + ((size_t)chunk * Cs + colbase + lane) * PS;
#pragma unroll
    for (int i = 0; i < NSUM; ++i) {
      const int b = wave + 4 * i;
      if (b < p) o[b] = sums[i];
    }
    if (wave == 0) {
      cnt_part[chunk * Cs + colbase + lane] = nvalid;
      for (int b = p; b < PS; ++b) o[b] = 0.0;
    }
  }
}

// Software-pipelined form of k_extract<TL, true> for a single band chunk (p <= XT_PBMAX, the production window): the
// (line, band) rows of tile i+1 are requested into registers BEFORE tile i is taken out of LDS, and the barriers
// order LDS only (s_waitcnt lgkmcnt(0) + s_barrier: __syncthreads() would drain vmcnt and with it the prefetch), so the
// global loads of one tile stay in flight under the validity test, the xt stores and the column sums of the previous
// one.  Measured on the full flightline: loads alone 1.59 ms, stores alone 1.13 ms, unpipelined 2.01 ms, this 1.43 ms.
// The band count is a template parameter: with runtime trip counts every load and LDS store sits in its own
// exec-masked block with a vmcnt(0) wait (174 VGPRs, two workgroups per CU, 2.49 ms).
// Same arithmetic, same order of the masked sums: bit-identical outputs.
__device__ __forceinline__ void xt_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// (152 VGPRs: three workgroups per CU; forcing four -- 128 VGPRs -- spills and is slower: 1.96 against 1.43 ms)
// (at least two waves per SIMD for the kernel below: left alone the CO2 instantiation took 270 registers and one workgroup per CU)
// NW: waves per workgroup (rows of the tile are dealt wave, wave + NW, ...).  Eight waves pay on the wide windows' one-line tile (one
// workgroup per CU: 14.6 -> 9.5 ms); on the CH4 window's four-line tile (two workgroups of four waves per CU already) two workgroups
// of eight measured the same step (9.00-9.11 against 9.03-9.05 ms) and were not kept.
template <int TL, int P, bool NTS, int NW = 4>
__device__ __forceinline__ void extract_pipe_body(const float *__restrict__ cube, int L, int B, int C, int s0,
                                                  int Cs, int b0, int PS, float *__restrict__ xt,
                                                  uint8_t *__restrict__ mask_t, int lines_per_wg, int cs, int ncb,
                                                  int nchunk, double *__restrict__ sum_part, int *__restrict__ cnt_part) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  __shared__ uint8_t vf[64][NW > TL ? NW : TL];
  constexpr int NSUM = (P + NW - 1) / NW;
  constexpr int NLD = (TL * P + NW - 1) / NW;              // rows per wave per tile: row = wave + NW u (the last u may be short)
  constexpr bool EVEN = (TL * P) % NW == 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int cbi, chunk;
  if (!sf_xcd_map(blockIdx.x, ncb, nchunk, cbi, chunk)) return;
  const int colbase = cbi * 64;
  const int ncol = min(64, Cs - colbase);
  const bool colok = lane < ncol;
  const int lbeg = chunk * lines_per_wg;
  const int lend = min(L, lbeg + lines_per_wg);
  const int lanec = colok ? lane : ncol - 1;
  const float *cbase = cube + (size_t)(s0 + colbase) + (size_t)b0 * C;
  double sums[NSUM];
  int nvalid = 0;
#pragma unroll
  for (int i = 0; i < NSUM; ++i) sums[i] = 0.0;
  float v[NLD];
  // rows of the tile starting at line l0 -> registers; lines past the chunk alias its last line (loaded, not stored)
  auto request = [&](int l0) {
    if constexpr (TL == 1) {
      // one line, rows wave, wave + 4, ...: ONE running pointer per lane (left to itself the compiler keeps an address per
      // load: 107 loads x 64 bits do not fit beside 107 float64 sums)
      const float *pp = cbase + ((size_t)min(l0, lend - 1) * B + wave) * C + lanec;
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        if (!EVEN && u == NLD - 1) pp -= (size_t)max(wave + NW * u - (P - 1), 0) * C;   // (a short last round re-reads the last row)
        v[u] = *pp;
        pp += NW * (size_t)C;
        asm volatile("" : "+v"(pp));
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int rr = EVEN ? wave + NW * u : min(wave + NW * u, TL * P - 1);   // (a short last round re-reads the last row)
      const int l = rr / P, b = rr - l * P;                // wave-uniform
      const int line = min(l0 + l, lend - 1);
      v[u] = (cbase + ((size_t)line * B + b) * C)[lanec];
      if constexpr (P > 72) __builtin_amdgcn_sched_barrier(0);   // one address at a time (63 row addresses do not fit the scalar file)
    }
  };
  auto body = [&](int l0, auto fullc) {
    constexpr bool FULL = decltype(fullc)::value;
    const int nl = FULL ? TL : min(TL, lend - l0);
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int rr = wave + NW * u;
      if ((FULL && (EVEN || rr < TL * P)) || rr < nl * P) tile[lane * cs + rr] = v[u];
    }
    xt_lds_barrier();
    if (l0 + TL < lend) request(l0 + TL);                 // in flight until the top of the next iteration
    if constexpr (TL == 1) {
      // one line per tile (the wide windows): every wave tests the bands it sums (wave, wave + 4, ...); the four partial
      // verdicts meet in vf[lane][0..3] and are combined after the barrier
      bool ok = true;
      const float *tp = tile + lane * cs;
#pragma unroll 8
      for (int i = 0; i < NSUM; ++i) {
        const int b = wave + NW * i;
        if (b < P) ok = ok & sf_valid(tp[b]);
      }
      vf[lane][wave] = ok ? 1 : 0;
      xt_lds_barrier();
      bool all = true;
#pragma unroll
      for (int w = 0; w < NW; ++w) all = all & (vf[lane][w] != 0);
      xt_lds_barrier();                                   // everybody has read the partial verdicts
      if (wave == 0) vf[lane][0] = all ? 1 : 0;
    } else if (FULL || wave < nl) {
      if (wave < TL) {
        bool ok = true;
        const float *tp = tile + lane * cs + wave * P;
#pragma unroll 8
        for (int b = 0; b < P; ++b) ok = ok & sf_valid(tp[b]);
        vf[lane][wave] = ok ? 1 : 0;
      }
    }
    xt_lds_barrier();                                     // vf of this tile is complete
    if (P == PS) {
      const int nel = nl * P;
      constexpr int NST = (TL * P + 63) / 64;
      for (int c = wave; c < ncol; c += NW) {
        float *dst = xt + ((size_t)(colbase + c) * L + l0) * PS;
        const float *src = tile + c * cs;
        float r[NST];
#pragma unroll
        for (int j = 0; j < NST; ++j) r[j] = src[min(lane + 64 * j, nel - 1)];   // the LDS reads first, then the stores
#pragma unroll
        for (int j = 0; j < NST; ++j)
          if (lane + 64 * j < nel) {
            if (NTS) __builtin_nontemporal_store(r[j], dst + lane + 64 * j);
            else dst[lane + 64 * j] = r[j];
          }
      }
    } else {
      // rows padded to PS = P rounded up to four: the nl lines of a column are still one contiguous run of nl * PS floats
      // (element e = band e % PS of line e / PS; the padding bands are written as zeros)
      constexpr int PSC = (P + 3) / 4 * 4;
      const int nel = nl * PSC;
      constexpr int NST = (TL * PSC + 63) / 64;
      for (int c = wave; c < ncol; c += NW) {
        float *dst = xt + ((size_t)(colbase + c) * L + l0) * PSC;
        const float *src = tile + c * cs;
        float r[NST];
#pragma unroll
        for (int j = 0; j < NST; ++j) {
          const int e = min(lane + 64 * j, nel - 1), l = e / PSC, b = e - l * PSC;
          r[j] = (b < P) ? src[l * P + b] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NST; ++j)
          if (lane + 64 * j < nel) {
            if (NTS) __builtin_nontemporal_store(r[j], dst + lane + 64 * j);
            else dst[lane + 64 * j] = r[j];
          }
      }
    }
    {
      bool vl[TL];
#pragma unroll
      for (int l = 0; l < TL; ++l) vl[l] = (FULL || l < nl) && vf[lane][l];
      if (wave == 0) {
#pragma unroll
        for (int l = 0; l < TL; ++l) nvalid += vl[l] ? 1 : 0;
      }
      const float *tp = tile + lane * cs;
#pragma unroll
      for (int i = 0; i < NSUM; ++i) {
        const int b = wave + NW * i;
        if (b < P) {
#pragma unroll
          for (int l = 0; l < TL; ++l)
            if (vl[l]) sums[i] += (double)tp[l * P + b];
        }
      }
    }
    if (wave == 0 && colok) {
      uint8_t *mp = mask_t + (size_t)(colbase + lane) * L + l0;
      if (TL == 2 && nl == 2 && (((size_t)(colbase + lane) * L + l0) & 1) == 0) {
        *reinterpret_cast<uint16_t *>(mp) = *reinterpret_cast<const uint16_t *>(&vf[lane][0]);
      } else {
        for (int l = 0; l < nl; ++l) mp[l] = vf[lane][l];
      }
    }
    xt_lds_barrier();                                     // tile and vf are rewritten by the next iteration
  };
  if (lbeg < lend) request(lbeg);
  int l0 = lbeg;
  for (; l0 + TL <= lend; l0 += TL) body(l0, std::true_type{});
  if (l0 < lend) body(l0, std::false_type{});
  if (colok) {
    double *o = sum_part + ((size_t)chunk * Cs + colbase + lane) * PS;
#pragma unroll
    for (int i = 0; i < NSUM; ++i) {
      const int b = wave + NW * i;
      if (b < P) o[b] = sums[i];
    }
    if (wave == 0) {
      cnt_part[chunk * Cs + colbase + lane] = nvalid;
      for (int b = P; b < PS; ++b) o[b] = 0.0;
    }
  }
}

template <int TL, int P, bool NTS = false>
__global__ __launch_bounds__(256, 2) void k_extract_pipe(const float *__restrict__ cube, int L, int B, int C, int s0, int Cs, int b0,
                                                          int PS, float *__restrict__ xt, uint8_t *__restrict__ mask_t,
                                                          int lines_per_wg, int cs, int ncb, int nchunk,
                                                          double *__restrict__ sum_part, int *__restrict__ cnt_part) {
  extract_pipe_body<TL, P, NTS>(cube, L, B, C, s0, Cs, b0, PS, xt, mask_t, lines_per_wg, cs, ncb, nchunk, sum_part, cnt_part);
}
// the wide windows: ONE line per tile, one wave per SIMD (the whole register file: 107 row loads in flight and 107 float64
// column sums per lane)
template <int P>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_extract_wide(const float *__restrict__ cube, int L, int B, int C, int s0, int Cs, int b0, int PS, float *__restrict__ xt,
                    uint8_t *__restrict__ mask_t, int lines_per_wg, int cs, int ncb, int nchunk, double *__restrict__ sum_part,
                    int *__restrict__ cnt_part) {
  extract_pipe_body<1, P, true>(cube, L, B, C, s0, Cs, b0, PS, xt, mask_t, lines_per_wg, cs, ncb, nchunk, sum_part, cnt_part);
}

// the same with EIGHT waves (two per SIMD, half the rows and half the sums each: round 6; the default)
template <int P>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_extract_wide8(const float *__restrict__ cube, int L, int B, int C, int s0, int Cs, int b0, int PS, float *__restrict__ xt,
                     uint8_t *__restrict__ mask_t, int lines_per_wg, int cs, int ncb, int nchunk, double *__restrict__ sum_part,
                     int *__restrict__ cnt_part) {
  extract_pipe_body<1, P, true, 8>(cube, L, B, C, s0, Cs, b0, PS, xt, mask_t, lines_per_wg, cs, ncb, nchunk, sum_part, cnt_part);
}

// Masked column sums over a chunk of lines.  One workgroup = (column, line chunk); thread (sub, q4)
// accumulates the float4 at band 4*q4 of rows sub, sub+rpi, ...; partials are combined in a fixed order
// so the mean is bit-reproducible.
template <typename XT>
__global__ __launch_bounds__(256) void k_colsum(const XT *__restrict__ xt, const uint8_t *__restrict__ mask_t,
                                                 int L, int PS, int lines_per_wg, double *__restrict__ sum_part,
                                                 int *__restrict__ cnt_part) {
  __shared__ double red[256 * 4];
  __shared__ int cred[256];
  const int tid = threadIdx.x;
  const int c = blockIdx.x, ch = blockIdx.y, Cs = gridDim.x;
  const int tpr = PS >> 2;
  const int rpi = 256 / tpr;
  const int sub = tid / tpr, q4 = tid - sub * tpr;
  const bool active = sub < rpi;
  const int lbeg = ch * lines_per_wg, lend = min(L, lbeg + lines_per_wg);
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  int cnt = 0;
  if (active) {
    const uint8_t *mp = mask_t + (size_t)c * L;
    const XT *xp = xt + (size_t)c * L * PS + 4 * q4;
    for (int r = lbeg + sub; r < lend; r += rpi) {
      if (mp[r]) {
        double v0, v1, v2, v3;
        sf_load4(xp + (size_t)r * PS, v0, v1, v2, v3);
        a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        ++cnt;
      }
    }
  }
  red[tid * 4 + 0] = a0; red[tid * 4 + 1] = a1; red[tid * 4 + 2] = a2; red[tid * 4 + 3] = a3;
  cred[tid] = cnt;
  __syncthreads();
  if (tid < tpr) {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int s = 0; s < rpi; ++s) {
      const int t = s * tpr + tid;
      s0 += red[t * 4 + 0]; s1 += red[t * 4 + 1]; s2 += red[t * 4 + 2]; s3 += red[t * 4 + 3];
    }
    double *o = sum_part + ((size_t)ch * Cs + c) * PS + 4 * tid;
    o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3;
  }
  if (tid == 0) {
    int n = 0;
    for (int s = 0; s < rpi; ++s) n += cred[s * tpr];  // q4 == 0 threads count rows
    cnt_part[ch * Cs + c] = n;
  }
}

// Combine the per-chunk partial sums: one 512-thread workgroup per column; lane = band, the 8 waves split the
// chunk range and are combined in a fixed order (bit-reproducible), so a long chunk list costs nchunk/8 steps.
__global__ __launch_bounds__(512) void k_mean(const double *__restrict__ sum_part, const int *__restrict__ cnt_part,
                                               int nch, int Cs, int p, int PS, int32_t *__restrict__ nuse,
                                               double *__restrict__ mu) {
  __shared__ double red[8][64];
  __shared__ int cred[512];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int cnt = 0;
  for (int ch = tid; ch < nch; ch += 512) cnt += cnt_part[ch * Cs + c];
  cred[tid] = cnt;
  __syncthreads();
  if (tid == 0) {
    int n = 0;
    for (int i = 0; i < 512; ++i) n += cred[i];
    cred[0] = n;
    nuse[c] = n;
  }
  __syncthreads();
  const int n = cred[0];
  for (int b0 = 0; b0 < p; b0 += 64) {
    const int b = b0 + lane;
    double s = 0;
    if (b < p) {
#pragma unroll 8
      for (int ch = wave; ch < nch; ch += 8) s += sum_part[((size_t)ch * Cs + c) * PS + b];
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && b < p) {
      double t = 0;
      for (int w = 0; w < 8; ++w) t += red[w][lane];
      mu[(size_t)c * p + b] = n > 0 ? t / (double)n : 0.0;
    }
    __syncthreads();
  }
}

// ---- narrow cubes (a rank's column shard held compactly: C == the number of columns wanted) ---------------
// Here the whole active window of a line is ONE contiguous run of p*C floats, so the 64-column blocking above
// (256-byte rows, a second block that is mostly idle lanes) is the wrong shape: k_extract_flat streams the run
// with fully coalesced loads, one workgroup per line chunk, the next tile's loads in flight (registers) while
// the current one is transposed out of LDS.  Same chunking and the same line order of the fused masked sums
// as k_extract, so a column's mean is bit-identical whichever kernel (shard layout) produced it.
// XF_NT = 1024: one 16-wave workgroup per CU (tiles of up to 12 * 1024 floats = 48 KB, 6 (band, column) pairs per thread for the fused
// sums: p C <= 6144).  (Round 6 measured the other shape -- XF_NT = 512, two 8-wave workgroups per CU on one-line tiles, 11 pairs per
// thread: 128 registers with 6 spilled, the 75-column shard step 1.53-1.54 ms against 1.48 -- and kept this one.)
template <int XF_NT, int XF_MAXLD, int XF_MAXSUM, bool V4 = false>      // V4: the lines fetched as 16-byte pieces (rows of a multiple of four floats)
__global__ __launch_bounds__(XF_NT) void k_extract_flat(const float *__restrict__ cube, int L, int B, int C, int b0,
                                                       int p, int PS, int TL, float *__restrict__ xt,
                                                       uint8_t *__restrict__ mask_t, int lines_per_wg,
                                                       double *__restrict__ sum_part, int *__restrict__ cnt_part) {
  extern __shared__ __attribute__((aligned(16))) float tile[];   // [TL][p][C] rounded up to whole XF_NT rows, then int vf[TL][C]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = blockIdx.x;
  const int rowlen = p * C;
  constexpr int XF_ROW = V4 ? 4 * XF_NT : XF_NT;              // floats a whole row of stores covers
  int *vf = reinterpret_cast<int *>(tile + (size_t)XF_ROW * ((TL * rowlen + XF_ROW - 1) / XF_ROW));
  const int lbeg = chunk * lines_per_wg, lend = min(L, lbeg + lines_per_wg);
  const bool fuse = sum_part != nullptr;
  double sums[XF_MAXSUM];
  int ccol[XF_MAXSUM];
#pragma unroll
  for (int k = 0; k < XF_MAXSUM; ++k) { sums[k] = 0.0; ccol[k] = (tid + XF_NT * k) % C; }
  int nvalid = 0;

  // V4 (round 6): XF_MAXLD pieces of 16 bytes per thread instead of XF_MAXLD floats -- four times the bytes in flight per load
  // instruction and four-line tiles (86 KB in flight per CU against 43).  A window row starts on any float, so the pieces are only
  // 4-byte aligned in global memory (the unaligned-capable global path); in the LDS tile they sit on 16-byte boundaries.
  typedef float xf_f4u __attribute__((ext_vector_type(4), aligned(4)));
  typedef float xf_f4 __attribute__((ext_vector_type(4)));
  constexpr int NV = V4 ? 4 : 1;
  float v[XF_MAXLD * NV];
  auto fetch = [&](int l0) {
    const int nl = min(TL, lend - l0);
    if constexpr (V4) {
      const int r4 = rowlen >> 2, n4 = nl * r4;
#pragma unroll
      for (int k = 0; k < XF_MAXLD; ++k) {
        const int idx = min(tid + XF_NT * k, n4 - 1);
        const int l = (idx >= r4) + (idx >= 2 * r4) + (idx >= 3 * r4);
        const xf_f4 t = *reinterpret_cast<const xf_f4u *>(cube + ((size_t)(l0 + l) * B + b0) * C + 4 * (idx - l * r4));
        v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
      }
    } else {
    const int n = nl * rowlen;
#pragma unroll
    for (int k = 0; k < XF_MAXLD; ++k) {
      const int idx = min(tid + XF_NT * k, n - 1);            // clamped duplicates are loaded but not stored
      const int l = (idx >= rowlen) + (idx >= 2 * rowlen) + (idx >= 3 * rowlen);
      v[k] = cube[((size_t)(l0 + l) * B + b0) * C + (idx - l * rowlen)];
    }
    }
  };
  if (lbeg < lend) fetch(lbeg);
  for (int l0 = lbeg; l0 < lend; l0 += TL) {
    const int nl = min(TL, lend - l0);
    const int n = nl * rowlen;
    // whole rows of XF_NT floats (a uniform trip count: per-lane predicates would put every store in its own
    // exec-masked block with a vmcnt(0) wait); the clamped duplicates of the last row land in the slack before vf
    const int kmax = ((V4 ? n >> 2 : n) + XF_NT - 1) / XF_NT;
#pragma unroll
    for (int k = 0; k < XF_MAXLD; ++k)
      if (k < kmax) {
        if constexpr (V4) reinterpret_cast<xf_f4 *>(tile)[tid + XF_NT * k] = xf_f4{v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
        else tile[tid + XF_NT * k] = v[k];
      }
    for (int i = tid; i < nl * C; i += XF_NT) vf[i] = 1;
    // LDS-only barriers from here on: __syncthreads() would wait for the tile requested next (vmcnt(0))
    xt_lds_barrier();
    if (l0 + TL < lend) fetch(l0 + TL);
    // ---- validity of (line, column): four threads share the bands of one pixel
    for (int i = tid; i < nl * C * 4; i += XF_NT) {
      const int qd = i & 3, lc = i >> 2;
      const int l = lc / C, c = lc - l * C;
      const float *tp = tile + (size_t)l * rowlen + c;
      bool ok = true;
      for (int b = qd; b < p; b += 4) ok = ok & sf_valid(tp[b * C]);
      if (!ok) vf[lc] = 0;   // benign race: every writer stores 0
    }
    xt_lds_barrier();
    // ---- LDS -> xt: a wave writes the nl*PS contiguous floats of a column
    const int nel = nl * PS;
    constexpr int XF_NST = (4 * XT_PBMAX + 63) / 64;        // <= 4 lines x 80 floats per column: 5 elements per lane
    for (int c = wave; c < C; c += XF_NT / 64) {
      float *dst = xt + ((size_t)c * L + l0) * PS;
      float r[XF_NST];
#pragma unroll
      for (int j = 0; j < XF_NST; ++j) {                     // all LDS reads first (clamped), then the stores
        const int k = min(lane + 64 * j, nel - 1);
        const int l = (k >= PS) + (k >= 2 * PS) + (k >= 3 * PS);
        const int b = k - l * PS;
        const float t = tile[(size_t)l * rowlen + min(b, p - 1) * C + c];
        r[j] = (b < p) ? t : 0.f;
      }
#pragma unroll
      for (int j = 0; j < XF_NST; ++j)
        if (lane + 64 * j < nel) dst[lane + 64 * j] = r[j];
    }
    for (int i = tid; i < nl * C; i += XF_NT) {
      const int l = i / C, c = i - l * C;
      mask_t[(size_t)c * L + l0 + l] = (uint8_t)vf[i];
    }
    if (fuse) {
      // masked sums, lines in order (the order k_extract uses): thread owns pairs idx = tid + 256 k = b*C + c
#pragma unroll
      for (int k = 0; k < XF_MAXSUM; ++k) {
        const int idx = tid + XF_NT * k;
        if (idx < rowlen) {
          for (int l = 0; l < nl; ++l) {   // branch-free: the two LDS reads of every line are independent
            const double xv = (double)tile[(size_t)l * rowlen + idx];
            sums[k] += vf[l * C + ccol[k]] ? xv : 0.0;
          }
        }
      }
      if (tid < C)
        for (int l = 0; l < nl; ++l) nvalid += vf[l * C + tid];
    }
    xt_lds_barrier();
  }
  if (fuse) {
#pragma unroll
    for (int k = 0; k < XF_MAXSUM; ++k) {
      const int idx = tid + XF_NT * k;
      if (idx < rowlen) {
        const int b = idx / C;
        sum_part[((size_t)chunk * C + ccol[k]) * PS + b] = sums[k];
      }
    }
    if (tid < C) {
      cnt_part[chunk * C + tid] = nvalid;
      for (int b = p; b < PS; ++b) sum_part[((size_t)chunk * C + tid) * PS + b] = 0.0;
    }
  }
}

}  // namespace


static int extract_chunks(int lines, int ncols, int *lpw_out) {
  const int lpw = sf_extract_lines_per_wg(lines, ncols);
  *lpw_out = lpw;
  return sf_cdiv(lines, lpw);
}

int sf_launch_extract(const float *cube, int lines, int bands, int samples, int s0, int ncols, int b0, int p,
                      float *xt, uint8_t *mask_t, double *sum_part, int *cnt_part, hipStream_t st) {
  const int PS = (p + 3) / 4 * 4;
  const int pbmax = (p <= XT_PBMAX) ? p : XT_PBMAX;
  constexpr int TL = 2;
  const int cs = (TL * pbmax) | 1;
  const size_t lds = (size_t)64 * cs * sizeof(float);
  const bool fuse = sum_part != nullptr && sf_extract_fuses_sum(p);
  const size_t maxlds = (size_t)64 * ((TL * XT_PBMAX) | 1) * sizeof(float);
  if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_extract<TL, true>), maxlds)) return rc;
  if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_extract<TL, false>), maxlds)) return rc;
  int lpw;
  const int nchunk = extract_chunks(lines, ncols, &lpw);
  // a compact narrow cube (a rank's shard): the flat kernel, if a tile of at least one line fits
  if (s0 == 0 && ncols == samples && p <= XT_PBMAX && (size_t)p * ncols <= 1024 * (size_t)6 && ncols <= 256 &&
      sf_tune().extract_variant != 1) {
    auto flat = [&](auto kern, int NT, int MAXLD, int threads) -> int {     // (NT: floats a row of stores covers, MAXLD rows per thread)
      int tl = (NT * MAXLD) / (p * ncols);
      if (tl > 4) tl = 4;
      const size_t ldsf = ((size_t)NT * (((size_t)tl * p * ncols + NT - 1) / NT) + (size_t)tl * ncols) * sizeof(float);
      if (int rc = sf_lds_attr(reinterpret_cast<const void *>(kern), ldsf > 64 * 1024 ? ldsf : (size_t)64 * 1024)) return rc;
      hipLaunchKernelGGL(kern, dim3(nchunk), dim3(threads), ldsf, st, cube, lines, bands, samples, b0, p, PS, tl, xt, mask_t, lpw,
                         fuse ? sum_part : nullptr, fuse ? cnt_part : nullptr);
      return 0;
    };
    // 16-byte pieces where a window row is a whole number of them (p = 72: 5400 floats at 75 columns); knob 6 = 7 keeps the 4-byte form
    const bool v4 = ((p * ncols) & 3) == 0 && sf_tune().extract_variant != 7;
    const int rc = v4 ? flat(k_extract_flat<1024, 6, 6, true>, 4096, 6, 1024) : flat(k_extract_flat<1024, 12, 6>, 1024, 12, 1024);
    if (rc) return rc;
    SF_LAUNCH_CHECK("k_extract_flat");
    return 0;
  }
  const int ncb = sf_cdiv(ncols, 64);
  if (fuse && (sf_tune().extract_variant == 0 || sf_tune().extract_variant == 3) && p == 72) {
    // the production window: software-pipelined kernel, FOUR lines per tile (72 row loads in flight per wave, 74 KB of
    // LDS, two workgroups per CU: 147 KB of loads in flight per CU against 110 KB with two-line tiles at three
    // workgroups -- 10.45 / 10.40 against 10.64 / 10.48 ms per flightline, same box, same bits; tools/tune_extract.py)
    const int tl = sf_tune().extract_variant == 3 ? 3 : 4, csx = (tl * 72) | 1;
    const size_t ldsx = (size_t)64 * csx * sizeof(float);
    if (tl == 3) {
      if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_extract_pipe<3, 72>), ldsx)) return rc;
      hipLaunchKernelGGL((k_extract_pipe<3, 72>), dim3(sf_xcd_grid(ncb, nchunk)), dim3(256), ldsx, st, cube, lines, bands, samples,
                         s0, ncols, b0, PS, xt, mask_t, lpw, csx, ncb, nchunk, sum_part, cnt_part);
    } else if (!sf_tune().extract_nt) {     // non-temporal xt stores (3.4 GB, re-read only after the whole pass): -0.08 ms alone, same bits
      if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_extract_pipe<4, 72, true>), ldsx)) return rc;
      hipLaunchKernelGGL((k_extract_pipe<4, 72, true>), dim3(sf_xcd_grid(ncb, nchunk)), dim3(256), ldsx, st, cube, lines, bands, samples,
                         s0, ncols, b0, PS, xt, mask_t, lpw, csx, ncb, nchunk, sum_part, cnt_part);
    } else {
      if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_extract_pipe<4, 72>), ldsx)) return rc;
      hipLaunchKernelGGL((k_extract_pipe<4, 72>), dim3(sf_xcd_grid(ncb, nchunk)), dim3(256), ldsx, st, cube, lines, bands, samples,
                         s0, ncols, b0, PS, xt, mask_t, lpw, csx, ncb, nchunk, sum_part, cnt_part);
    }
  } else if (fuse && (sf_tune().extract_variant == 0 || sf_tune().extract_variant == 9) && (p == 425 || p == 416)) {
    // the full-band window of the benchmark (1..425) and the reference's -R window (5..420: robust_mf.py:186-187): ONE line per
    // tile (64 columns x p bands = 109 KB of LDS, one workgroup per CU with the whole register file: 107 row loads in flight
    // and 107 float64 column sums per lane), the column sums fused as on the narrow windows -- round 4 ran the blocked kernel
    // (2.7 TB/s) and re-read the 20.5 GB of xt for the sums (k_colsum)
    const int csx = p | 1;
    const size_t ldsx = (size_t)64 * csx * sizeof(float);
    // round 6: EIGHT waves (two per SIMD, half the rows and half the float64 sums each: 196 registers) -- 14.6 -> ~10 ms a
    // flightline at p = 425 (the full-band step 274.4 -> 270.1 ms), the same bits; sf_debug_set(6, 9): the four-wave form
    const bool w4 = sf_tune().extract_variant == 9;
#define SF_WIDE_GO(K)                                                                                                             \
  {                                                                                                                               \
    if (int rc = sf_lds_attr(reinterpret_cast<const void *>(K), ldsx)) return rc;                                                 \
    hipLaunchKernelGGL((K), dim3(sf_xcd_grid(ncb, nchunk)), dim3(w4 ? 256 : 512), ldsx, st, cube, lines, bands, samples, s0, ncols, \
                       b0, PS, xt, mask_t, lpw, csx, ncb, nchunk, sum_part, cnt_part);                                            \
  }
    if (p == 425) {
      if (w4) SF_WIDE_GO(k_extract_wide<425>) else SF_WIDE_GO(k_extract_wide8<425>)
    } else {
      if (w4) SF_WIDE_GO(k_extract_wide<416>) else SF_WIDE_GO(k_extract_wide8<416>)
    }
#undef SF_WIDE_GO
  } else if (fuse && sf_tune().extract_variant == 0 && p == 83) {
    // the CO2 window (robust_mf.py:190-191): the same kernel, three lines per tile (64 KB of LDS: two workgroups per CU;
    // four lines would be 85 KB and one), rows padded to 84 floats, non-temporal stores
    const int csx = (3 * 83) | 1;
    const size_t ldsx = (size_t)64 * csx * sizeof(float);
    if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_extract_pipe<3, 83, true>), ldsx)) return rc;
    hipLaunchKernelGGL((k_extract_pipe<3, 83, true>), dim3(sf_xcd_grid(ncb, nchunk)), dim3(256), ldsx, st, cube, lines, bands, samples,
                       s0, ncols, b0, PS, xt, mask_t, lpw, csx, ncb, nchunk, sum_part, cnt_part);
  } else if (fuse && sf_tune().extract_variant != 2 && p == 72) {     // variant 5 (any other value): two lines per tile, round 1's form
    if (int rc = sf_lds_attr(reinterpret_cast<const void *>(k_extract_pipe<TL, 72>), maxlds)) return rc;
    hipLaunchKernelGGL((k_extract_pipe<TL, 72>), dim3(sf_xcd_grid(ncb, nchunk)), dim3(256), lds, st, cube, lines, bands, samples,
                       s0, ncols, b0, PS, xt, mask_t, lpw, cs, ncb, nchunk, sum_part, cnt_part);
  } else if (fuse && p <= XT_PBMAX)
    hipLaunchKernelGGL((k_extract<TL, true>), dim3(sf_xcd_grid(ncb, nchunk)), dim3(256), lds, st, cube, lines, bands,
                       samples, s0, ncols, b0, p, PS, xt, mask_t, lpw, pbmax, cs, ncb, nchunk, sum_part, cnt_part);
  else
    hipLaunchKernelGGL((k_extract<TL, false>), dim3(sf_xcd_grid(ncb, nchunk)), dim3(256), lds, st, cube, lines, bands,
                       samples, s0, ncols, b0, p, PS, xt, mask_t, lpw, pbmax, cs, ncb, nchunk, nullptr, nullptr);
  SF_LAUNCH_CHECK("k_extract");
  return 0;
}

// scratch of the fused extract+sum path: [nchunk][ncols][ps] doubles + [nchunk][ncols] ints
size_t sf_extract_sum_bytes(const SfGeom &g) {
  int lpw;
  const int nchunk = extract_chunks(g.lines, g.ncols, &lpw);
  return sf_align((size_t)nchunk * g.ncols * g.ps * sizeof(double)) + sf_align((size_t)nchunk * g.ncols * sizeof(int));
}
bool sf_extract_fuses_sum(int p) {
  return p <= XT_PBMAX || ((p == 425 || p == 416) && (sf_tune().extract_variant == 0 || sf_tune().extract_variant == 9));
}

int sf_launch_extract_fused(const float *cube, int lines, int bands, int samples, int s0, int b0, const SfGeom &g,
                            float *xt, uint8_t *mask_t, void *scratch, hipStream_t st) {
  int lpw;
  const int nchunk = extract_chunks(g.lines, g.ncols, &lpw);
  double *sum_part = reinterpret_cast<double *>(scratch);
  int *cnt_part = reinterpret_cast<int *>(reinterpret_cast<char *>(scratch) +
                                          sf_align((size_t)nchunk * g.ncols * g.ps * sizeof(double)));
  return sf_launch_extract(cube, lines, bands, samples, s0, g.ncols, b0, g.p, xt, mask_t, sum_part, cnt_part, st);
}

size_t sf_mean_scratch_bytes(const SfGeom &g) {
  const int nch = sf_colsum_chunks(g.lines, g.ncols);
  return sf_align((size_t)nch * g.ncols * g.ps * sizeof(double)) + sf_align((size_t)nch * g.ncols * sizeof(int));
}

int sf_launch_mean_from_partials(const SfGeom &g, int32_t *nuse, double *mu, void *scratch, hipStream_t st) {
  int lpw;
  const int nchunk = extract_chunks(g.lines, g.ncols, &lpw);
  double *sum_part = reinterpret_cast<double *>(scratch);
  int *cnt_part = reinterpret_cast<int *>(reinterpret_cast<char *>(scratch) +
                                          sf_align((size_t)nchunk * g.ncols * g.ps * sizeof(double)));
  hipLaunchKernelGGL(k_mean, dim3(g.ncols), dim3(512), 0, st, sum_part, cnt_part, nchunk, g.ncols, g.p, g.ps, nuse, mu);
  SF_LAUNCH_CHECK("k_mean");
  return 0;
}

int sf_launch_mean(const void *xt, int xt_f64, const uint8_t *mask_t, const SfGeom &g, int32_t *nuse, double *mu,
                   void *scratch, hipStream_t st) {
  if (g.ps > 1024) {
    sf_set_error("active window of %d bands is too wide for the column-sum kernel", g.p);
    return -2;
  }
  const int nch = sf_colsum_chunks(g.lines, g.ncols);
  const int lpw = sf_cdiv(g.lines, nch);
  double *sum_part = reinterpret_cast<double *>(scratch);
  int *cnt_part = reinterpret_cast<int *>(reinterpret_cast<char *>(scratch) +
                                          sf_align((size_t)nch * g.ncols * g.ps * sizeof(double)));
  if (xt_f64)
    hipLaunchKernelGGL(k_colsum<double>, dim3(g.ncols, nch), dim3(256), 0, st, (const double *)xt, mask_t, g.lines,
                       g.ps, lpw, sum_part, cnt_part);
  else
    hipLaunchKernelGGL(k_colsum<float>, dim3(g.ncols, nch), dim3(256), 0, st, (const float *)xt, mask_t, g.lines,
                       g.ps, lpw, sum_part, cnt_part);
  SF_LAUNCH_CHECK("k_colsum");
  hipLaunchKernelGGL(k_mean, dim3(g.ncols), dim3(512), 0, st, sum_part, cnt_part, nch, g.ncols, g.p, g.ps, nuse, mu);
  SF_LAUNCH_CHECK("k_mean");
  return 0;
}
