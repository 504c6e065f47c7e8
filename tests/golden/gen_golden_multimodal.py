#!/usr/bin/env python3
"""Golden vectors for the multimodal background (-k 2) by EXECUTING THE REAL REFERENCE (development container only).

Run:  OMP_NUM_THREADS=1 python tests/golden/gen_golden_multimodal.py      (needs /root/reference and scikit-learn)

The reference's MiniBatchKMeans is unseeded (cmf/robust_mf.py:312); numpy's global RandomState, which sklearn
falls back to, is seeded here so that the run can be repeated, and the labels the reference chose are part of
the golden anyway: the bgmeta image holds (cluster id, alpha index) per valid pixel (:327, :365).  The parity
tests inject those labels and compare everything downstream of the clustering.
"""
import os
import sys

os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402


def main():
    G.install_spectral_stub()
    lib = np.float64(np.loadtxt(G.LIB_TXT))
    lines, samples, seed = 1600, 5, 555
    cube = G.make_cube_numpy(lines, samples, seed=seed, abscf_full=lib[:, 2], nodata_column=3)
    # a second background mode: brighten the scene over part of the flightline (different mean and covariance)
    cube[600:1150] *= np.float32(1.35)
    np.random.seed(7)
    r = G.run_reference_main(cube, extra_args=("-k", "2"), tag="K2")
    lab = r["bgmeta"][:, :, 0]
    print("cluster sizes per column:", [(int((lab[:, c] == 0).sum()), int((lab[:, c] == 1).sum())) for c in range(samples)])
    print("alpha idx:", np.unique(r["bgmeta"][:, :, 1]))
    np.savez_compressed(os.path.join(HERE, "cmf_K2_multimodal.npz"), seed=seed, lines=lines, samples=samples,
                        nodata_column=3, bright=np.array([600, 1150, 1.35]), out=r["out"], bgmeta=r["bgmeta"],
                        colstats=r["colstats"], modelparms=np.array(r["modelparms"]), versions=G.versions())


def main_reject_full():
    """-k 3 -r (cluster rejection) and -k 2 -f (full-column regulariser), same recipe."""
    G.install_spectral_stub()
    lib = np.float64(np.loadtxt(G.LIB_TXT))
    lines, samples, seed = 1200, 5, 556
    cube = G.make_cube_numpy(lines, samples, seed=seed, abscf_full=lib[:, 2], nodata_column=1)
    cube[500:900] *= np.float32(1.35)
    # a handful of very bright rows: k-means gives them a cluster of their own, smaller than int(71*1.2) = 85 rows
    cube[1000:1040] *= np.float32(2.6)
    np.random.seed(11)
    r = G.run_reference_main(cube, extra_args=("-k", "3", "-r"), tag="K3R")
    lab = r["bgmeta"][:, :, 0]
    print("-k 3 -r cluster ids per column:", [sorted(set(np.unique(lab[:, c]).tolist())) for c in range(samples)])
    print("   sizes:", [[int((lab[:, c] == v).sum()) for v in np.unique(lab[:, c])] for c in range(samples)])
    np.savez_compressed(os.path.join(HERE, "cmf_K3_reject.npz"), seed=seed, lines=lines, samples=samples,
                        nodata_column=1, bright=np.array([[500, 900, 1.35], [1000, 1040, 2.6]]), out=r["out"],
                        bgmeta=r["bgmeta"], colstats=r["colstats"], modelparms=np.array(r["modelparms"]),
                        versions=G.versions())
    np.random.seed(12)
    r = G.run_reference_main(cube, extra_args=("-k", "2", "-f"), tag="K2F")
    print("-k 2 -f alpha idx:", np.unique(r["bgmeta"][:, :, 1]))
    np.savez_compressed(os.path.join(HERE, "cmf_K2_full.npz"), seed=seed, lines=lines, samples=samples,
                        nodata_column=1, bright=np.array([[500, 900, 1.35], [1000, 1040, 2.6]]), out=r["out"],
                        bgmeta=r["bgmeta"], colstats=r["colstats"], modelparms=np.array(r["modelparms"]),
                        versions=G.versions())


def main_empirical():
    """-M empirical (sample covariance, no shrinkage; cmf/robust_mf.py:350-351, :366-367) without -m: with -m the reference
    dies on `alphas`."""
    G.install_spectral_stub()
    lib = np.float64(np.loadtxt(G.LIB_TXT))
    lines, samples, seed = 600, 5, 557
    cube = G.make_cube_numpy(lines, samples, seed=seed, abscf_full=lib[:, 2], nodata_column=2)
    r = G.run_reference_main(cube, extra_args=("-M", "empirical"), tag="EMP", metadata=False)
    print("empirical: score range", r["out"][..., 3][r["out"][..., 3] != -9999].min(), r["out"][..., 3].max(), r["modelparms"])
    np.savez_compressed(os.path.join(HERE, "cmf_empirical.npz"), seed=seed, lines=lines, samples=samples, nodata_column=2,
                        out=r["out"], colstats=r["colstats"], modelparms=np.array(r["modelparms"]), versions=G.versions())


def main_empirical_more():
    """-M empirical on the other two branches: multimodal (-k 2) and the wide reflectance window (-R, p = 416).
    The reference cannot save its labels with -M empirical (-m dies on `alphas`), but the clustering happens before the
    model is fitted and consumes numpy's global RNG identically: the SAME seed with -k 2 -m (looshrinkage) gives the
    labels of the -M empirical -k 2 run (checked here: the two runs' NODATA patterns and cluster-dependent scores are
    reproduced by the oracle with those labels)."""
    G.install_spectral_stub()
    lib = np.float64(np.loadtxt(G.LIB_TXT))
    lines, samples, seed = 1400, 4, 558
    cube = G.make_cube_numpy(lines, samples, seed=seed, abscf_full=lib[:, 2], nodata_column=1)
    cube[500:1000] *= np.float32(1.35)
    np.random.seed(21)
    ra = G.run_reference_main(cube, extra_args=("-k", "2"), tag="EK2A")                 # labels (bgmeta band 0)
    np.random.seed(21)
    rb = G.run_reference_main(cube, extra_args=("-k", "2", "-M", "empirical"), tag="EK2B", metadata=False)
    np.savez_compressed(os.path.join(HERE, "cmf_empirical_K2.npz"), seed=seed, lines=lines, samples=samples, nodata_column=1,
                        bright=np.array([500, 1000, 1.35]), labels=ra["bgmeta"][:, :, 0], out=rb["out"], colstats=rb["colstats"],
                        modelparms=np.array(rb["modelparms"]), versions=G.versions())
    print("empirical -k 2:", rb["modelparms"], "cluster sizes", [(int((ra["bgmeta"][:, c, 0] == 0).sum()), int((ra["bgmeta"][:, c, 0] == 1).sum())) for c in range(samples)])
    lines, samples, seed = 900, 3, 559
    cube = G.make_cube_numpy(lines, samples, seed=seed, abscf_full=lib[:, 2], nodata_column=2)
    cube = np.float32(np.clip(cube, -1e9, None) * (cube > 0) * 0.08 + cube * (cube <= 0))   # reflectance-like magnitudes
    rc = G.run_reference_main(cube, extra_args=("-R", "-M", "empirical"), tag="ER", metadata=False)
    np.savez_compressed(os.path.join(HERE, "cmf_empirical_R.npz"), seed=seed, lines=lines, samples=samples, nodata_column=2,
                        scale=0.08, out=rc["out"], colstats=rc["colstats"], modelparms=np.array(rc["modelparms"]),
                        versions=G.versions())
    print("empirical -R:", rc["modelparms"])


def main_full_wide():
    """-R -k 2 -f: the full-column regulariser on the reflectance window (p = 416) -- looshrinkage(..., I_reg) with a
    416 x 416 target (cmf/robust_mf.py:99, :131, :354)."""
    G.install_spectral_stub()
    lib = np.float64(np.loadtxt(G.LIB_TXT))
    lines, samples, seed = 1000, 3, 560
    cube = G.make_cube_numpy(lines, samples, seed=seed, abscf_full=lib[:, 2], active=(5, 420), nodata_column=1)
    cube[400:750] *= np.float32(1.3)
    np.random.seed(31)
    r = G.run_reference_main(cube, extra_args=("-R", "-k", "2", "-f"), tag="RK2F")
    lab = r["bgmeta"][:, :, 0]
    print("-R -k 2 -f cluster sizes:", [(int((lab[:, c] == 0).sum()), int((lab[:, c] == 1).sum())) for c in range(samples)],
          "alpha idx:", np.unique(r["bgmeta"][:, :, 1]), r["modelparms"])
    np.savez_compressed(os.path.join(HERE, "cmf_R_K2_full.npz"), seed=seed, lines=lines, samples=samples, nodata_column=1,
                        bright=np.array([[400, 750, 1.3]]), out=r["out"], bgmeta=r["bgmeta"], colstats=r["colstats"],
                        modelparms=np.array(r["modelparms"]), versions=G.versions())


if __name__ == "__main__":
    if "--full-wide" in sys.argv:
        main_full_wide()
        sys.exit(0)
    if "--empirical-more" in sys.argv:
        main_empirical_more()
        sys.exit(0)
    if "--empirical" in sys.argv:
        main_empirical()
        sys.exit(0)
    if "--reject-full" in sys.argv:
        main_reject_full()
        sys.exit(0)
    main()
