"""The CPU oracle (oracle/cmf_oracle.py) against golden vectors produced by the REAL reference
(tests/golden/gen_golden.py).  Pins the oracle; runs anywhere (no GPU, no /root/reference)."""
import os

import numpy as np
import pytest

from oracle import cmf_oracle as O
from srcfinder_amd.synth import make_cube_numpy, synth_columns


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_alpha_grid():
    a = O.alpha_grid()
    assert a.shape == (201,)
    assert a[0] == 1e-10
    assert a[200] == 1.0000000000003273


def test_S_config_bit_exact(golden_dir, library):
    g = _load(golden_dir, "cmf_S_radiance.npz")
    cube = make_cube_numpy(int(g["lines"]), int(g["samples"]), seed=int(g["seed"]), abscf_full=library[:, 2])
    r = O.robust_mf_oracle(cube, library)
    assert np.array_equal(r["bgmeta"], g["bgmeta"])                 # alpha index image, int16
    assert np.array_equal(r["out"][..., 3] == -9999.0, g["out"][..., 3] == -9999.0)   # NODATA placement
    assert np.array_equal(r["out"][..., :3], g["out"][..., :3])     # RGB copy (all-nodata column stays 0)
    # same numpy/scipy as the generator -> bit-identical; otherwise LAPACK may differ in the last bits
    same_build = str(g["versions"]) == "numpy %s scipy %s" % (np.__version__, __import__("scipy").__version__)
    if same_build:
        assert np.array_equal(r["out"], g["out"])
        assert np.array_equal(r["colstats"], g["colstats"])
    else:
        np.testing.assert_allclose(r["out"], g["out"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(r["colstats"], g["colstats"], rtol=1e-9)
    # the eigen restatement (the GPU algorithm) selects the same alpha index everywhere
    r2 = O.robust_mf_oracle(cube, library, shrinkage=O.looshrinkage_eig)
    assert np.array_equal(r2["bgmeta"], g["bgmeta"])
    v = g["out"][..., 3] != -9999.0
    np.testing.assert_allclose(r2["out"][..., 3][v], g["out"][..., 3][v], rtol=1e-8, atol=1e-8)


def test_reflectance_mode(golden_dir, library):
    g = _load(golden_dir, "cmf_R_reflectance.npz")
    cube = make_cube_numpy(int(g["lines"]), int(g["samples"]), seed=int(g["seed"]), abscf_full=library[:, 2],
                           active=(5, 420), nodata_column=int(g["nodata_column"]))
    r = O.robust_mf_oracle(cube, library, reflectance=True)
    assert np.array_equal(r["bgmeta"], g["bgmeta"])
    np.testing.assert_allclose(r["out"], g["out"], rtol=1e-9, atol=1e-12)


def test_singular_column(golden_dir, library):
    g = _load(golden_dir, "cmf_singular_column.npz")
    cube = make_cube_numpy(int(g["lines"]), int(g["samples"]), seed=int(g["seed"]), abscf_full=library[:, 2],
                           nodata_column=-1)
    cube[:, int(g["const_band"]), int(g["const_col"])] = g["const_value"]
    r = O.robust_mf_oracle(cube, library)
    assert r["status"][int(g["const_col"])] == 2
    assert np.array_equal(r["bgmeta"], g["bgmeta"])
    np.testing.assert_allclose(r["out"], g["out"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(r["colstats"], g["colstats"], rtol=1e-9, atol=1e-12)


CASES = ["n100_p8", "n512_p72", "n2000_p72", "n2000_p425", "n300_p425", "n100_p8_big"]


@pytest.mark.parametrize("name", CASES)
def test_looshrinkage_cases(golden_dir, name):
    g = _load(golden_dir, "cmf_looshrinkage_cases.npz")
    n, p, seed, scale = g[name + "_spec"]
    x = synth_columns(int(n), int(p), int(seed), float(scale))
    izm = x - x.mean(axis=0)
    nll = np.zeros(201)
    C, mindex = O.looshrinkage(izm, g["alphas"], nll, int(n))
    assert mindex == int(g[name + "_mindex"])
    ref = g[name + "_nll"]
    assert np.array_equal(np.isinf(nll), np.isinf(ref))             # det over/underflow exclusions
    f = np.isfinite(ref)
    np.testing.assert_allclose(nll[f], ref[f], rtol=1e-10)
    if int(p) <= 72:
        np.testing.assert_allclose(C, g[name + "_C"], rtol=1e-12, atol=0)
    else:
        np.testing.assert_allclose(np.diag(C), g[name + "_Cdiag"], rtol=1e-12)
        np.testing.assert_allclose(C[::17, ::13], g[name + "_Csub"], rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("name", ["n100_p8", "n512_p72", "n2000_p72", "n100_p8_big"])
def test_eig_restatement_matches_reference(golden_dir, name):
    g = _load(golden_dir, "cmf_looshrinkage_cases.npz")
    n, p, seed, scale = g[name + "_spec"]
    x = synth_columns(int(n), int(p), int(seed), float(scale))
    nll = np.zeros(201)
    C, mindex = O.looshrinkage_eig(x - x.mean(axis=0), g["alphas"], nll, int(n))
    assert mindex == int(g[name + "_mindex"])
    np.testing.assert_allclose(nll, g[name + "_nll"], rtol=1e-10)
    np.testing.assert_allclose(C, g[name + "_C"], rtol=1e-12)


def test_const_band_all_inf(golden_dir):
    g = _load(golden_dir, "cmf_looshrinkage_cases.npz")
    n, p, seed, scale = g["const_band_spec"]
    x = synth_columns(int(n), int(p), int(seed), float(scale))
    x[:, 3] = np.float64(np.float32(1.25))
    nll = np.zeros(201)
    C, mindex = O.looshrinkage(x - x.mean(axis=0), g["alphas"], nll, int(n))
    assert mindex == -1 == int(g["const_band_mindex"])
    assert np.all(np.isinf(nll))
    np.testing.assert_allclose(C, g["const_band_C"], rtol=1e-12, atol=0)
    assert bool(g["const_band_inv_raises"])
    with pytest.raises(Exception):
        O.inv(C)
    nll2 = np.zeros(201)
    _, m2 = O.looshrinkage_eig(x - x.mean(axis=0), g["alphas"], nll2, int(n))
    assert m2 == -1


def test_wrappers(golden_dir):
    g = _load(golden_dir, "cmf_looshrinkage_cases.npz")
    n, p, seed, scale = g["wrap_spec"]
    a = synth_columns(int(n), int(p), int(seed), float(scale))
    np.testing.assert_allclose(O.cov(a), g["wrap_cov"], rtol=1e-13)
    np.testing.assert_allclose(O.inv(O.cov(a)), g["wrap_inv"], rtol=1e-9)
    np.testing.assert_allclose(O.det(O.cov(a)), g["wrap_det"], rtol=1e-10)
    ev, evec = O.eig(O.cov(a))
    np.testing.assert_allclose(np.sort(ev.real), np.sort(g["wrap_eigvals"].real), rtol=1e-9)


def test_multimodal_oracle_reproduces_reference_with_injected_labels(golden_dir, library):
    """-k 2 golden from the real reference (tests/golden/gen_golden_multimodal.py): with the reference's own cluster
    labels injected, the restatement reproduces scores, alpha indices and column statistics bit for bit."""
    g = np.load(os.path.join(golden_dir, "cmf_K2_multimodal.npz"))
    lines, samples = int(g["lines"]), int(g["samples"])
    cube = make_cube_numpy(lines, samples, seed=int(g["seed"]), abscf_full=library[:, 2], nodata_column=int(g["nodata_column"]))
    b0, b1, f = g["bright"]
    cube[int(b0):int(b1)] *= np.float32(f)
    o = O.robust_mf_multimodal_oracle(cube, library, g["bgmeta"][:, :, 0].astype(np.int64))
    assert np.array_equal(o["out"], g["out"])
    assert np.array_equal(o["bgmeta"], g["bgmeta"])
    assert np.array_equal(o["colstats"], g["colstats"])


def _reject_full_cube(g, library):
    cube = make_cube_numpy(int(g["lines"]), int(g["samples"]), seed=int(g["seed"]), abscf_full=library[:, 2],
                           nodata_column=int(g["nodata_column"]))
    for b0, b1, f in g["bright"]:
        cube[int(b0):int(b1)] *= np.float32(f)
    return cube


def test_multimodal_oracle_reproduces_reference_with_cluster_rejection(golden_dir, library):
    """-k 3 -r golden from the real reference: a 38-40 row cluster is relabelled -1 in four columns, never scored, and
    the model of the remaining rows overwrites cluster 0's scores (robust_mf.py:317-341).  Bit for bit."""
    g = np.load(os.path.join(golden_dir, "cmf_K3_reject.npz"))
    assert (g["bgmeta"][:, :, 0] < 0).any()
    o = O.robust_mf_multimodal_oracle(_reject_full_cube(g, library), library, np.abs(g["bgmeta"][:, :, 0].astype(np.int64)),
                                      reject=True)
    assert np.array_equal(o["out"], g["out"])
    assert np.array_equal(o["bgmeta"], g["bgmeta"])
    assert np.array_equal(o["colstats"], g["colstats"])


def test_multimodal_oracle_reproduces_reference_with_full_regulariser(golden_dir, library):
    """-k 2 -f golden from the real reference: shrinkage target = covariance of the whole column (:354, :99)."""
    g = np.load(os.path.join(golden_dir, "cmf_K2_full.npz"))
    o = O.robust_mf_multimodal_oracle(_reject_full_cube(g, library), library, g["bgmeta"][:, :, 0].astype(np.int64), full=True)
    assert np.array_equal(o["out"], g["out"])
    assert np.array_equal(o["bgmeta"], g["bgmeta"])
    assert np.array_equal(o["colstats"], g["colstats"])


def test_multimodal_oracle_reproduces_reference_with_full_regulariser_on_the_reflectance_window(golden_dir, library):
    """-R -k 2 -f golden from the real reference (p = 416, a 332-row cluster: S singular, the 416 x 416 target not)."""
    g = np.load(os.path.join(golden_dir, "cmf_R_K2_full.npz"))
    cube = make_cube_numpy(int(g["lines"]), int(g["samples"]), seed=int(g["seed"]), abscf_full=library[:, 2],
                           active=(5, 420), nodata_column=int(g["nodata_column"]))
    for b0, b1, f in g["bright"]:
        cube[int(b0):int(b1)] *= np.float32(f)
    with np.errstate(all="ignore"):
        o = O.robust_mf_multimodal_oracle(cube, library, g["bgmeta"][:, :, 0].astype(np.int64), reflectance=True, full=True)
    assert np.array_equal(o["out"], g["out"])
    assert np.array_equal(o["bgmeta"], g["bgmeta"])
    assert np.array_equal(o["colstats"], g["colstats"])


def test_empirical_model_oracle_reproduces_reference(golden_dir, library):
    """-M empirical golden from the real reference (no -m: with it the reference dies on `alphas`)."""
    g = np.load(os.path.join(golden_dir, "cmf_empirical.npz"))
    cube = make_cube_numpy(int(g["lines"]), int(g["samples"]), seed=int(g["seed"]), abscf_full=library[:, 2],
                           nodata_column=int(g["nodata_column"]))
    o = O.robust_mf_oracle(cube, library, model="empirical")
    assert np.array_equal(o["out"], g["out"])
    assert np.array_equal(o["colstats"], g["colstats"])


def test_triage_profile_golden(golden_dir):
    """N2: the column-profile restatement against the CSV tables written by the real triage/cmf_profile.py
    (tests/golden/gen_golden_triage.py) -- bit for bit, NaN placement included."""
    import importlib.util
    from oracle import triage_oracle as TO
    spec = importlib.util.spec_from_file_location("gen_golden_triage", os.path.join(golden_dir, "gen_golden_triage.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    g = np.load(os.path.join(golden_dir, "triage_profile.npz"))
    img = gen.product(int(g["lines"]), int(g["samples"]), int(g["seed"]))
    plain = TO.column_profile(img[..., 3]).T
    rob = TO.column_profile_robust(img[..., 3]).T
    # the statistics are float32 values; the CSV carries them with pandas' 16 significant digits -> compare as float32
    assert np.array_equal(np.float32(plain), np.float32(g["plain"]), equal_nan=True)
    assert np.array_equal(np.float32(rob), np.float32(g["robust"]), equal_nan=True)
    assert np.isnan(g["plain"][7, 1]) and g["plain"][11, 0] == 1 and g["plain"][13, 0] == 0
    # the 40000-line product of round 5 (longer than the GPU's LDS-resident sort): the same restatement, the same bar
    gb = np.load(os.path.join(golden_dir, "triage_profile_big.npz"))
    big = gen.product(int(gb["lines"]), int(gb["samples"]), int(gb["seed"]))
    assert np.array_equal(np.float32(TO.column_profile(big[..., 3]).T), np.float32(gb["plain"]), equal_nan=True)
    assert np.array_equal(np.float32(TO.column_profile_robust(big[..., 3]).T), np.float32(gb["robust"]), equal_nan=True)


def test_empirical_multimodal_and_wide_goldens(golden_dir, library):
    """-M empirical on the multimodal branch (labels of the same-seed -k 2 -m run, see gen_golden_multimodal.py) and on
    the wide reflectance window: the oracle reproduces the real reference's products bit for bit."""
    from srcfinder_amd.synth import make_cube_numpy
    g = np.load(os.path.join(golden_dir, "cmf_empirical_K2.npz"))
    cube = make_cube_numpy(int(g["lines"]), int(g["samples"]), seed=int(g["seed"]), abscf_full=library[:, 2],
                           nodata_column=int(g["nodata_column"]))
    b0, b1, f = g["bright"]
    cube[int(b0):int(b1)] *= np.float32(f)
    o = O.robust_mf_multimodal_oracle(cube, library, g["labels"], model="empirical")
    assert np.array_equal(o["out"], g["out"])
    g = np.load(os.path.join(golden_dir, "cmf_empirical_R.npz"))
    cube = make_cube_numpy(int(g["lines"]), int(g["samples"]), seed=int(g["seed"]), abscf_full=library[:, 2],
                           nodata_column=int(g["nodata_column"]))
    cube = np.float32(np.clip(cube, -1e9, None) * (cube > 0) * 0.08 + cube * (cube <= 0))
    o = O.robust_mf_oracle(cube, library, reflectance=True, model="empirical")
    assert np.array_equal(o["out"], g["out"])
