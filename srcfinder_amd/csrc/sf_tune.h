// Tuning / experiment knobs of the library (sf_debug_set) -- kept apart from cmf_common.h so that adding a knob does not
// change the hash a PMC record of the score kernel is tied to (bench.py: kernel_source_sha).
#pragma once
// Tuning / experiment knobs (sf_debug_set): a per-THREAD context, zero-initialised = the built-in choices.  The library
// holds no process-wide mutable state: a thread that flips a knob (tools/tune_*.py, the kernel-variant tests) changes
// only the launches it issues itself.
struct SfTune {
  int score_variant = 0;      // key 1: score kernel form (cmf_score.hip): 100 = round 1 (16-byte pieces stored by the lanes), 200 = the launch's traffic only (timing), 1..9 = batch shapes
  int score_lpw = 0;          // key 2: lines per workgroup of the column-block score kernel
  int score_xcd = 1;          // key 3: XCD-aware block map of the column-block score kernel
  int sweep_variant = 0;      // key 4: 1 = force the 16x16x4 sweep, 2 = full-rank 4x4x4 sweep only
  int cov_variant = 0;        // key 5: 1 = force the 16x16x4 covariance, 3 = two waves per SIMD
  int extract_variant = 0;    // key 6: 1 = never the flat (narrow-cube) kernel, 2 = the unpipelined blocked kernel, 3 / 5 = 3- / 2-line tiles in the pipelined kernel (default 4), 9 = the four-wave form of the wide-window kernel (default: eight waves), 7 = the flat kernel fetches 4-byte pieces (default: 16-byte pieces where the window row allows)
  int sweep_grid = 0;         // key 21: k_sweep4s workgroup order: 0 = the splits of a column adjacent (default), 1 = columns fastest (round 2's order)
  int extract_nt = 0;         // key 19: 1 = plain (not non-temporal) xt stores in the pipelined extract kernel
  int eigh_lpp = 0;           // key 7: lanes per column pair of the Jacobi eigensolver (4 / 8 / 16); 2 = the sweeps behind the tridiagonal preconditioner of cmf_eigh_pre.h (8 lanes)
  int sweep4r_waves = 8;      // key 8: waves per workgroup of the rank-factored sweep (k_sweep4r, form 1)
  int sweep4_form = 0;        // key 20: rank-28 sweep kernel: 0 = k_sweep4s (round 3: one streamed ring), 1 = k_sweep4r (round 2, both ranks), 3 = k_sweep4r for the rank-36 columns only, 4 = k_sweep4s renormalising after every tile; 100 + bits = timing experiments (-DSF_SWEEP_EXPERIMENTS)
  int wide_eigh_variant = 0;  // key 10: 0 = blocked Jacobi behind the tridiagonal preconditioner (cmf_wtri.hip) for calls of 32 columns or more; 7 = the preconditioner for any number of columns; 6 = the sweeps from the Cholesky factor (round 4's first form: 11-12 sweeps); 1 = the single-workgroup eigensolver for every wide matrix, 8 = as 7 with every preconditioner refused afterwards (the fallback's test)
  int wjac_stamps = 0;        // key 22: 1 = the fused wide sweep accumulates its phase clocks (sf_debug_wsweep_stamps)
  int wsweep_variant = 0;     // key 24: the fused wide sweep: 0 = k_wsweep8 (eight waves, wave-private operand slices) where it applies; 1 = 32-row tiles, two workgroups per CU; 2 = eight waves on shared chunks; 4 = four waves on shared chunks (round 4's first form)
  int cnn_pool_variant = 0;   // key 18: inception branch 4: 0 = pool taken from the tile staged in LDS (k_poolconv), 3 = pool kernel then convolution, 2 = the same with the general pool kernel, 1 = pool inside the 1x1 convolution's tile fetch (nine reads; slower)
  int cnn_conv_variant = 0;   // key 17: the DEFAULT route of srcfinder_amd.cnn when a call passes none (tools' A/B runs; the product passes its route as an argument -- sf_cnn_score_rows(route), forward_tiles(route=) -- and the overflow rescue never touches this knob): 0 = operand splitting on the fp16 matrix cores (cnn_split.hip; default), 4 = Winograd F(2x2, 3x3) for the 3 x 3 layers + the fp32 implicit GEMM (cnn_wino.hip / cnn_kernels.hip), 2 = the direct fp32 implicit GEMM for everything, 1 = its pointer-form tile loads (the form operands of 2 GB or more take)
  int cnn_variant = 0;        // key 16: 1 = the 8 x 8 conv1+pool kernel (cnn_kernels.hip); 4 = maxpool4 as its own kernel (default on the split route: in inception4e's epilogues); 2 = k_conv_split never takes 160-channel tiles (round 5's choice); 3 = no band sharing in sf_cnn_score_rows (the rings computed whole per window: round 6's first form)
  int det_variant = 0;        // key 15: 1 = the plain window rule of the exact-determinant pass in one round, 2 = no pass in sf_cmf_run's narrow branch
  int det_slots = 0;          // key 26: workgroups (work matrices) of a launch of the exact-determinant pass; 0 = as many as fit (512)
  int lu_variant = 0;         // key 14: 1 = the unblocked LU in the determinant passes (linalg.hip)
};
SfTune &sf_tune();   // c_api.hip (thread_local)

// exact-determinant pass on windows of up to 96 bands: grid points per range crossing (two rounds of four; linalg.hip)
constexpr int SF_NARROW_DET_WINDOW = 8;
