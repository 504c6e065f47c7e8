// Stage 3: per-column sample covariance of the centred valid rows in float64 on the matrix cores.
//
// Replaces numpy.cov as called by looshrinkage (cmf/robust_mf.py:52-70, :98, :130): S = X^T X / (n-1) with
// X = valid rows minus the column mean.  (numpy.cov subtracts the mean of the already-centred data once
// more; that mean is O(1e-17) and is not re-subtracted here.)  The 100x "stability scaling" of :94-99 is a
// pure factor 1e4 on S and is applied analytically downstream.
//
// One workgroup = (column, row split).  Rows are staged 32 at a time into LDS as float64 (centred, invalid
// rows and the band padding zeroed -> they contribute exactly 0), then every wave accumulates its share of
// the upper-triangular 16x16 tiles with v_mfma_f64_16x16x4_f64: A = X[:, tile_i]^T, B = X[:, tile_j].
// LDS row stride is 16 (mod 32) doubles so the 4 rows x 16 bands an operand fetch touches hit 64 distinct
// banks.  MFMA-bound: 2*16*16*4 flop per instruction, p=72 -> 15 tiles per 4 rows.
#include "cmf_common.h"

namespace {

constexpr int SY_TLS = 32;  // rows staged per step

template <int NT>
struct SyrkCfg {
  static constexpr int PP = 16 * NT;
  static constexpr int LDX = PP + ((PP % 32 == 0) ? 16 : 0);
  static constexpr int NTRI = NT * (NT + 1) / 2;
  static constexpr int TPW = (NTRI + 3) / 4;  // tiles per wave
};

// upper-triangular tile list (ti <= tj), row-major, as a compile-time table
template <int NT>
struct TriTab {
  int ti[NT * (NT + 1) / 2], tj[NT * (NT + 1) / 2];
  constexpr TriTab() : ti{}, tj{} {
    int q = 0;
    for (int i = 0; i < NT; ++i)
      for (int j = i; j < NT; ++j) { ti[q] = i; tj[q] = j; ++q; }
  }
};
__device__ __forceinline__ void tri_decode(int q, int nt, int &ti, int &tj) {
  int i = 0, rowlen = nt;
  while (q >= rowlen) { q -= rowlen; ++i; --rowlen; }
  ti = i;
  tj = i + q;
}

// MFMA work of wave W on one staged tile: its tile list is a compile-time constant, so the NT operand
// fragments of a k-step live in named registers (no dynamic indexing) and are read from LDS once.
template <int NT, int W>
__device__ __forceinline__ void syrk_wave(const double *__restrict__ xs, d4_t (&acc)[SyrkCfg<NT>::TPW]) {
  using Cfg = SyrkCfg<NT>;
  constexpr TriTab<NT> tab{};
#pragma unroll 2
  for (int k0 = 0; k0 < SY_TLS; k0 += 4) {
    double f[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) f[t] = xs[k0 * Cfg::LDX + 16 * t];
#pragma unroll
    for (int u = 0; u < Cfg::TPW; ++u) {
      constexpr int q0 = W * Cfg::TPW;
      if (q0 + u < Cfg::NTRI) {
        const int ti = tab.ti[q0 + u < Cfg::NTRI ? q0 + u : 0], tj = tab.tj[q0 + u < Cfg::NTRI ? q0 + u : 0];
        acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[ti], f[tj], acc[u], 0, 0, 0);
      }
    }
  }
}

template <int NT, typename XT>
__global__ __launch_bounds__(256, 4) void k_syrk(const XT *__restrict__ xt, const uint8_t *__restrict__ mask_t,
                                               const double *__restrict__ mu, int L, int p, int PS, int rows_per_wg,
                                               double *__restrict__ part) {
  using Cfg = SyrkCfg<NT>;
  __shared__ __attribute__((aligned(16))) double Xs[SY_TLS * Cfg::LDX];
  __shared__ double mus[Cfg::PP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int c = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
  const int rbeg = split * rows_per_wg, rend = min(L, rbeg + rows_per_wg);

  for (int i = tid; i < Cfg::PP; i += 256) mus[i] = (i < p) ? mu[(size_t)c * p + i] : 0.0;
  for (int i = tid; i < SY_TLS * Cfg::LDX; i += 256) Xs[i] = 0.0;  // band padding stays zero forever

  d4_t acc[Cfg::TPW];
#pragma unroll
  for (int u = 0; u < Cfg::TPW; ++u) acc[u] = d4_t{0.0, 0.0, 0.0, 0.0};

  const int tpr = PS >> 2;
  const uint8_t *mp = mask_t + (size_t)c * L;
  const XT *xc = xt + (size_t)c * L * PS;
  const double *xs = Xs + g * Cfg::LDX + li;  // this lane's operand element of k-step 0, tile 0
  for (int r0 = rbeg; r0 < rend; r0 += SY_TLS) {
    __syncthreads();  // previous tile fully consumed (also orders the initial zero fill / mus)
    for (int it = tid; it < SY_TLS * tpr; it += 256) {
      const int row = it / tpr, q4 = it - row * tpr;
      const int r = r0 + row;
      double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
      if (r < rend && mp[r]) {
        double f0, f1, f2, f3;
        sf_load4(xc + (size_t)r * PS + 4 * q4, f0, f1, f2, f3);
        const int b = 4 * q4;
        v0 = (b + 0 < p) ? f0 - mus[b + 0] : 0.0;
        v1 = (b + 1 < p) ? f1 - mus[b + 1] : 0.0;
        v2 = (b + 2 < p) ? f2 - mus[b + 2] : 0.0;
        v3 = (b + 3 < p) ? f3 - mus[b + 3] : 0.0;
      }
      double *dst = Xs + row * Cfg::LDX + 4 * q4;
      dst[0] = v0; dst[1] = v1; dst[2] = v2; dst[3] = v3;
    }
    __syncthreads();
    switch (wave) {  // wave-uniform
      case 0: syrk_wave<NT, 0>(xs, acc); break;
      case 1: syrk_wave<NT, 1>(xs, acc); break;
      case 2: syrk_wave<NT, 2>(xs, acc); break;
      default: syrk_wave<NT, 3>(xs, acc); break;
    }
  }
#pragma unroll
  for (int u = 0; u < Cfg::TPW; ++u) {
    const int q = wave * Cfg::TPW + u;
    if (q < Cfg::NTRI) {
      double *o = part + (((size_t)c * nsplit + split) * Cfg::NTRI + q) * 256 + lane;
      o[0] = acc[u][0]; o[64] = acc[u][1]; o[128] = acc[u][2]; o[192] = acc[u][3];
    }
  }
}

// Sum the split partials in a fixed order, divide by n-1, expand the triangle into the full matrix.
// Accumulator element (reg, lane) of tile (ti,tj) is D[16ti + (lane>>4) + 4reg][16tj + (lane&15)].
template <int NT>
__global__ __launch_bounds__(256) void k_syrk_reduce(const double *__restrict__ part, int nsplit,
                                                      const int32_t *__restrict__ nuse, int p,
                                                      double *__restrict__ cov) {
  using Cfg = SyrkCfg<NT>;
  const int c = blockIdx.x, q = blockIdx.y, tid = threadIdx.x, lane = tid & 63, reg = tid >> 6;
  const double denom = (double)nuse[c] - 1.0;
  int ti, tj;
  tri_decode(q, NT, ti, tj);
  double s = 0;
  for (int sp = 0; sp < nsplit; ++sp) s += part[(((size_t)c * nsplit + sp) * Cfg::NTRI + q) * 256 + tid];
  s /= denom;
  const int i = 16 * ti + (lane >> 4) + 4 * reg, j = 16 * tj + (lane & 15);
  if (i < p && j < p) {
    cov[((size_t)c * p + i) * p + j] = s;
    if (ti != tj) cov[((size_t)c * p + j) * p + i] = s;
  }
}

template <int NT>
int launch_cov_nt(const void *xt, int xt_f64, const uint8_t *mask_t, const int32_t *nuse, const double *mu,
                  const SfGeom &g, double *cov, void *scratch, hipStream_t st) {
  const int nsplit = sf_syrk_splits(g.lines, g.ncols);
  int rows = sf_cdiv(g.lines, nsplit);
  rows = (rows + SY_TLS - 1) / SY_TLS * SY_TLS;
  double *part = reinterpret_cast<double *>(scratch);
  if (xt_f64)
    hipLaunchKernelGGL((k_syrk<NT, double>), dim3(g.ncols, nsplit), dim3(256), 0, st, (const double *)xt, mask_t, mu,
                       g.lines, g.p, g.ps, rows, part);
  else
    hipLaunchKernelGGL((k_syrk<NT, float>), dim3(g.ncols, nsplit), dim3(256), 0, st, (const float *)xt, mask_t, mu,
                       g.lines, g.p, g.ps, rows, part);
  SF_LAUNCH_CHECK("k_syrk");
  hipLaunchKernelGGL(k_syrk_reduce<NT>, dim3(g.ncols, SyrkCfg<NT>::NTRI), dim3(256), 0, st, part, nsplit, nuse, g.p, cov);
  SF_LAUNCH_CHECK("k_syrk_reduce");
  return 0;
}

}  // namespace


size_t sf_cov_scratch_bytes(const SfGeom &g) {
  const int nsplit = sf_syrk_splits(g.lines, g.ncols);
  const size_t ntri = (size_t)g.nt * (g.nt + 1) / 2;
  const size_t a = sf_align((size_t)g.ncols * nsplit * ntri * 256 * sizeof(double)), b = sf_cov4_scratch_bytes(g);
  return a > b ? a : b;
}

int sf_launch_cov(const void *xt, int xt_f64, const uint8_t *mask_t, const int32_t *nuse, const double *mu,
                  const SfGeom &g, double *cov, void *scratch, hipStream_t st) {
  if (!xt_f64 && sf_tune().cov_variant != 1 && sf_sw4_groups(g.p))   // p in 69..72 (CH4), 81..84 (CO2), 93..96
    return sf_launch_cov4((const float *)xt, mask_t, nuse, mu, g, cov, scratch, st);
  switch (g.nt) {
    case 1: return launch_cov_nt<1>(xt, xt_f64, mask_t, nuse, mu, g, cov, scratch, st);
    case 2: return launch_cov_nt<2>(xt, xt_f64, mask_t, nuse, mu, g, cov, scratch, st);
    case 3: return launch_cov_nt<3>(xt, xt_f64, mask_t, nuse, mu, g, cov, scratch, st);
    case 4: return launch_cov_nt<4>(xt, xt_f64, mask_t, nuse, mu, g, cov, scratch, st);
    case 5: return launch_cov_nt<5>(xt, xt_f64, mask_t, nuse, mu, g, cov, scratch, st);
    case 6: return launch_cov_nt<6>(xt, xt_f64, mask_t, nuse, mu, g, cov, scratch, st);
    default:
      sf_set_error("active window of %d bands exceeds the fused statistics path (max %d)", g.p, SF_MAX_ACTIVE_FUSED);
      return -2;
  }
}
