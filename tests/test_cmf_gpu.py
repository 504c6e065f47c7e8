"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the CPU
oracle and the golden vectors made by the real reference.

Bar: integer outputs (valid rows / NODATA placement, alpha index, nuse, status) bit-exact; float64 scores within
1e-4 relative (metric of SURVEY.md §7.3: |d| <= 1e-4 |ref| + 1e-9 max|ref|) -- in practice ~1e-9.
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from conftest import score_close  # noqa: E402
from oracle import cmf_oracle as O  # noqa: E402
from srcfinder_amd import _ffi, cmf  # noqa: E402
from srcfinder_amd.synth import make_cube_numpy, synth_columns  # noqa: E402


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    assert os.path.isfile(_ffi.LIB_PATH), "HIP library not built -- the product has no fallback"
    return torch


def run_stages(torch, cube_np, a0, a1, abscf, reflectance=False):
    """Drive stages 1..6 one by one through the C ABI and return every intermediate as numpy."""
    L = _ffi.lib()
    dev = torch.device("cuda:0")
    cube = torch.as_tensor(cube_np).to(dev)
    lines, bands, samples = cube_np.shape
    p = a1 - a0 + 1
    ps = (p + 3) // 4 * 4
    alphas_np = cmf.alpha_grid()
    nalpha = len(alphas_np)
    ws = torch.empty(L.sf_cmf_workspace_bytes(lines, p, samples, nalpha), dtype=torch.uint8, device=dev)
    f64 = dict(dtype=torch.float64, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    xt = torch.full((samples, lines, ps), -777.0, dtype=torch.float32, device=dev)
    mask = torch.full((samples, lines), 7, dtype=torch.uint8, device=dev)
    nuse = torch.empty(samples, **i32)
    mu = torch.empty((samples, p), **f64)
    S = torch.empty((samples, p, p), **f64)
    d = torch.empty((samples, p), **f64)
    lam = torch.empty((samples, p), **f64)
    evec = torch.empty((samples, p, p), **f64)
    status = torch.empty(samples, **i32)
    nll = torch.empty((samples, nalpha), **f64)
    aidx = torch.empty(samples, **i32)
    filt = torch.empty((samples, p), **f64)
    bias = torch.empty(samples, **f64)
    al = torch.as_tensor(alphas_np, device=dev)
    ab = torch.as_tensor(np.ascontiguousarray(abscf), device=dev)
    st = _ffi.stream_ptr()
    P = _ffi.ptr
    _ffi.check(L.sf_cmf_extract_columns(P(cube), lines, bands, samples, 0, samples, a0 - 1, p, P(xt), P(mask), st), "extract")
    _ffi.check(L.sf_cmf_column_mean(P(xt), 0, P(mask), lines, p, samples, P(nuse), P(mu), P(ws), st), "mean")
    _ffi.check(L.sf_cmf_covariance(P(xt), 0, P(mask), P(nuse), P(mu), lines, p, samples, P(S), P(ws), st), "cov")
    _ffi.check(L.sf_cmf_eigh(P(S), P(nuse), p, samples, P(d), P(lam), P(evec), P(status), P(ws), st), "eigh")
    _ffi.check(L.sf_cmf_loocv(P(xt), 0, P(mask), P(nuse), P(mu), P(d), P(lam), P(evec), P(status), P(al), nalpha,
                              lines, p, samples, P(nll), P(aidx), P(ws), st), "loocv")
    _ffi.check(L.sf_cmf_filter(P(mu), P(d), P(lam), P(evec), P(al), P(aidx), P(ab), int(reflectance), p, samples,
                               P(status), P(filt), P(bias), st), "filter")
    torch.cuda.synchronize()
    g = lambda t: t.cpu().numpy()
    return dict(xt=g(xt), mask=g(mask), nuse=g(nuse), mu=g(mu), S=g(S), d=g(d), lam=g(lam), evec=g(evec),
                status=g(status), nll=g(nll), aidx=g(aidx), filt=g(filt), bias=g(bias))


@pytest.fixture(scope="module")
def small_case(torch_cuda, library):
    """97 lines x 70 samples: ragged against every tile size (4-line, 16-row, 32-row, 64-column)."""
    cube = make_cube_numpy(97, 70, seed=7, abscf_full=library[:, 2], nodata_lines=3)
    a0, a1 = 351, 422
    st = run_stages(torch_cuda, cube, a0, a1, library[a0 - 1:a1, 2])
    return cube, a0, a1, st


def test_stage1_extract_and_mask(small_case):
    cube, a0, a1, st = small_case
    sub = cube[:, a0 - 1:a1, :]                                # [lines, p, samples]
    want = np.ascontiguousarray(sub.transpose(2, 0, 1))         # [samples, lines, p]
    assert np.array_equal(st["xt"].view(np.uint32), want.view(np.uint32))      # bit-exact copy (NaN included)
    valid = ((~(sub < 0)) & np.isfinite(sub)).all(axis=1).T     # [samples, lines]
    assert np.array_equal(st["mask"], valid.astype(np.uint8))


def test_stage2_mean(small_case):
    cube, a0, a1, st = small_case
    for c in range(cube.shape[2]):
        x = np.float64(cube[:, a0 - 1:a1, c])
        use = O.useidx(x)
        assert st["nuse"][c] == len(use)
        if len(use):
            np.testing.assert_allclose(st["mu"][c], x[use].mean(axis=0), rtol=1e-14)


def test_stage3_covariance(small_case):
    cube, a0, a1, st = small_case
    for c in range(0, cube.shape[2], 3):
        x = np.float64(cube[:, a0 - 1:a1, c])
        use = O.useidx(x)
        if len(use) < 2:
            continue
        want = O.cov(x[use] - x[use].mean(axis=0))
        np.testing.assert_allclose(st["S"][c], want, rtol=1e-11, atol=1e-13 * np.abs(want).max())
        assert np.array_equal(st["S"][c], st["S"][c].T)


def test_stage4_eigh(small_case):
    cube, a0, a1, st = small_case
    p = a1 - a0 + 1
    for c in range(0, cube.shape[2], 5):
        if st["status"][c] != 0:
            continue
        S, d, lam, V = st["S"][c], st["d"][c], st["lam"][c], st["evec"][c]     # V[j] = eigenvector j
        np.testing.assert_allclose(d, np.sqrt(np.diag(S)), rtol=1e-15)
        R = S / np.outer(d, d)
        np.testing.assert_allclose(V @ V.T, np.eye(p), atol=1e-13)
        np.testing.assert_allclose((V.T * lam) @ V, R, atol=5e-14)
        np.testing.assert_allclose(np.sort(lam), np.sort(np.linalg.eigvalsh(R)).clip(0), atol=1e-13)


def test_stage5_nll_and_alpha(small_case):
    cube, a0, a1, st = small_case
    alphas = cmf.alpha_grid()
    for c in range(0, cube.shape[2], 7):
        x = np.float64(cube[:, a0 - 1:a1, c])
        use = O.useidx(x)
        if len(use) < 2:
            continue
        nll = np.zeros(201)
        _, mindex = O.looshrinkage(x[use] - x[use].mean(axis=0), alphas, nll, len(use))
        assert st["aidx"][c] == mindex
        assert np.array_equal(np.isinf(st["nll"][c]), np.isinf(nll))
        f = np.isfinite(nll)
        np.testing.assert_allclose(st["nll"][c][f], nll[f], rtol=1e-10)


def _compare_run(res, want, lines, samples):
    out, ref = res.out, want["out"]
    assert out.shape == ref.shape
    nod = ref[..., -1] == -9999.0
    assert np.array_equal(out[..., -1] == -9999.0, nod)                       # NODATA placement, bit-exact
    assert np.array_equal(res.bgmeta, want["bgmeta"])                          # alpha index image, bit-exact
    if ref.shape[-1] == 4:
        assert np.array_equal(out[..., :3], ref[..., :3])                      # RGB copy, bit-exact
    ok = score_close(out[..., -1][~nod], ref[..., -1][~nod])
    assert ok.all(), "%d of %d scores outside 1e-4 relative" % ((~ok).sum(), ok.size)
    scale = np.abs(ref[..., -1][~nod]).max() if (~nod).any() else 1.0
    assert np.abs(out[..., -1][~nod] - ref[..., -1][~nod]).max() <= 1e-7 * scale
    cs, cr = res.colstats, want["colstats"]
    assert np.array_equal(cs[0], cr[0])                                        # npix
    std = np.where(cr[2] > 0, cr[2], 1.0)
    assert np.all(np.abs(cs[1] - cr[1]) <= 1e-7 * std + 1e-12)                 # mean (is ~0: absolute vs spread)
    np.testing.assert_allclose(cs[2], cr[2], rtol=1e-6, atol=1e-12)


def test_golden_S_config(torch_cuda, golden_dir, library):
    """BASELINE config 0: 64 cols x 512 lines x 425 bands through the REAL reference (golden)."""
    g = np.load(os.path.join(golden_dir, "cmf_S_radiance.npz"))
    cube = make_cube_numpy(int(g["lines"]), int(g["samples"]), seed=int(g["seed"]), abscf_full=library[:, 2])
    res = cmf.robust_mf(torch_cuda.as_tensor(cube).cuda(), library, metadata=True, to_numpy=True, return_nll=True)
    _compare_run(res, g, int(g["lines"]), int(g["samples"]))
    # against the oracle too: same alpha index per column, same nuse, status
    o = O.robust_mf_oracle(cube, library)
    assert np.array_equal(res.nuse, o["nuse"])
    assert np.array_equal(res.status, o["status"])
    solved = o["status"] == 0
    assert np.array_equal(res.alphaidx[solved], o["alphaidx"][solved])


def test_golden_singular_column(torch_cuda, golden_dir, library):
    g = np.load(os.path.join(golden_dir, "cmf_singular_column.npz"))
    cube = make_cube_numpy(int(g["lines"]), int(g["samples"]), seed=int(g["seed"]), abscf_full=library[:, 2],
                           nodata_column=-1)
    cube[:, int(g["const_band"]), int(g["const_col"])] = g["const_value"]
    res = cmf.robust_mf(cube, library, metadata=True, to_numpy=True)
    assert res.status[int(g["const_col"])] == 2
    _compare_run(res, g, int(g["lines"]), int(g["samples"]))


def test_ragged_cube_against_oracle(torch_cuda, library, small_case):
    cube, a0, a1, _ = small_case
    res = cmf.robust_mf(cube, library, metadata=True, to_numpy=True)
    o = O.robust_mf_oracle(cube, library)
    _compare_run(res, o, *cube.shape[::2])
    assert np.array_equal(res.nuse, o["nuse"]) and np.array_equal(res.status, o["status"])


def test_score_only_output_and_co2_window(torch_cuda, library):
    """rgb_bands=() -> one-band product; CO2 window (p = 83) exercises the NT = 6 kernels."""
    cube = make_cube_numpy(150, 20, seed=11, abscf_full=library[:, 2], active=(309, 391), nodata_column=7)
    lib = library.copy()
    lib[:, 2] = -np.abs(np.sin(np.arange(425) / 9.0)) * 0.05          # a synthetic "co2" absorption column
    res = cmf.robust_mf(cube, lib, gas="co2", rgb_bands=(), metadata=True, to_numpy=True)
    o = O.robust_mf_oracle(cube, lib, gas="co2", rgb_bands=())
    assert res.out.shape == (150, 20, 1)
    _compare_run(res, o, 150, 20)


def test_reflectance_flag_against_oracle(torch_cuda, library):
    """-R semantics (target = abscf - mu, no ppm scaling; robust_mf.py:378-386) on a window the LDS-resident
    statistics path supports; the reference's own p = 416 window is pinned for the oracle only."""
    cube = make_cube_numpy(180, 33, seed=21, abscf_full=library[:, 2], nodata_column=9)
    res = cmf.robust_mf(cube, library, reflectance=True, active=(351, 422), metadata=True, to_numpy=True)
    o = O.robust_mf_oracle(cube, library, reflectance=True, active=(351, 422))
    _compare_run(res, o, 180, 33)


def test_golden_reflectance_wide_window(torch_cuda, golden_dir, library):
    """The reference's own -R run (active 5..420, p = 416): wide-window statistics path (batched fp64 GEMMs +
    global-memory Jacobi) against the golden produced by the real reference."""
    g = np.load(os.path.join(golden_dir, "cmf_R_reflectance.npz"))
    cube = make_cube_numpy(int(g["lines"]), int(g["samples"]), seed=int(g["seed"]), abscf_full=library[:, 2],
                           active=(5, 420), nodata_column=int(g["nodata_column"]))
    res = cmf.robust_mf(cube, library, reflectance=True, metadata=True, to_numpy=True)
    assert res.modelparms == str(g["modelparms"])
    _compare_run(res, g, int(g["lines"]), int(g["samples"]))


def test_fewer_valid_rows_than_bands(torch_cuda, library):
    """n < p: the sample correlation matrix is singular, Cholesky fails and the eigensolver takes its
    fallback (Jacobi on R with replayed rotations); the shrinkage still makes every G_alpha invertible."""
    cube = make_cube_numpy(58, 9, seed=41, abscf_full=library[:, 2], nodata_lines=2, nodata_column=4)
    res = cmf.robust_mf(cube, library, metadata=True, to_numpy=True, return_nll=True)
    o = O.robust_mf_oracle(cube, library, return_nll=True)
    assert np.array_equal(res.nuse, o["nuse"]) and res.nuse.max() < 72
    solved = o["status"] == 0
    assert np.array_equal(res.alphaidx[solved], o["alphaidx"][solved])
    f = np.isfinite(o["nll"][solved])
    assert np.array_equal(np.isfinite(res.nll[solved]), f)
    # for the smallest alphas G_alpha has condition ~1e10: the reference's own LU det/inverse carry ~1e-6 there
    np.testing.assert_allclose(res.nll[solved][f], o["nll"][solved][f], rtol=5e-5)
    near = o["nll"][solved] < o["nll"][solved].min(axis=1, keepdims=True) + 5.0
    np.testing.assert_allclose(res.nll[solved][near], o["nll"][solved][near], rtol=1e-9)
    nod = o["out"][..., 3] == -9999.0
    assert np.array_equal(res.out[..., 3] == -9999.0, nod)
    # C is numerically singular-ish here (alpha ~ 1e-3 on a rank-deficient S): scores agree to the
    # conditioning of the reference's own LU inverse, far inside the 1e-4 bar
    assert score_close(res.out[..., 3][~nod], o["out"][..., 3][~nod], rel=1e-4).all()


@pytest.mark.parametrize("active,decades", [((309, 391), (1.5, 3.0, 3.6, 5.5)), ((320, 415), (1.5, 3.6, 6.0)), ((331, 411), (2.0, 5.5)),
                                            ((300, 392), (2.0, 3.6))])
def test_sweep_routes_of_the_co2_and_96_band_windows(torch_cuda, library, active, decades):
    """windows of 81..84 (CO2: 309..391, p = 83) and 93..96 bands run the 4x4x4 kernels (k_syrk4h, k_lowrank<NJ>,
    k_sweep4s<NK, NJ>).  One cube holds benchmark-like columns (rank 28) and columns whose correlation spectra span 3 to 6
    decades (rank 36 where its tables fit the LDS, else refused -> the 16x16x4 kernel): every route of the launch against
    the faithful oracle -- alpha index exact, NODATA placement exact, scores 1e-4."""
    torch = torch_cuda
    a0, a1 = active
    p = a1 - a0 + 1
    lines, samples = 1800, 4 + 3 * len(decades)
    cube = make_cube_numpy(lines, samples, seed=61 + p, abscf_full=library[:, 2], active=active, nodata_column=2, nodata_lines=3)
    rng = np.random.default_rng(p)
    for k, dec in enumerate(decades):
        for c in range(4 + 3 * k, 7 + 3 * k):
            qmat, _ = np.linalg.qr(rng.standard_normal((p, p)))
            sd = np.sqrt(np.exp(np.linspace(0.0, -dec * np.log(10.0), p)))
            x = 10.0 + 0.5 * (rng.standard_normal((lines, p)) * sd) @ qmat.T
            cube[3:, a0 - 1:a1, c] = x[3:].astype(np.float32)
    routes = cmf.sweep_routes(torch.as_tensor(cube).cuda(), library, active=active)
    assert routes["rank24"] + routes["rank28"] >= 3 and routes["rank36"] + routes["full"] >= 1, routes
    res = cmf.robust_mf(cube, library, active=active, metadata=True, to_numpy=True)
    o = O.robust_mf_oracle(cube, library, active=active)
    _compare_run(res, o, lines, samples)
    assert np.array_equal(res.nuse, o["nuse"]) and np.array_equal(res.status, o["status"])
    # a 3-column shard of the same cube: bit-identical columns (every float64 sum in an order that depends on the lines only)
    sh = cmf.robust_mf(cube, library, active=active, metadata=True, to_numpy=True, columns=(3, 6))
    assert np.array_equal(sh.out, res.out[:, 3:6]) and np.array_equal(sh.alphaidx, res.alphaidx[3:6])
    print("p = %d routes %s" % (p, routes))


@pytest.mark.parametrize("active", [(351, 421), (360, 366), (340, 422)])
def test_odd_and_unusual_windows(torch_cuda, library, active):
    """p = 71 (odd: the eigensolver pads a dummy column), p = 7 (single MFMA tile), p = 83 with another offset."""
    cube = make_cube_numpy(131, 19, seed=51, abscf_full=library[:, 2], active=active, nodata_column=3)
    res = cmf.robust_mf(cube, library, active=active, metadata=True, to_numpy=True)
    o = O.robust_mf_oracle(cube, library, active=active)
    _compare_run(res, o, 131, 19)


def test_degenerate_shapes(torch_cuda, library):
    """One column; fewer lines than a tile; a column with a single valid row (NaN score, alpha index 0, as the reference)
    next to normal ones; +inf inside the window invalidates the row."""
    cube = make_cube_numpy(90, 1, seed=61, abscf_full=library[:, 2], nodata_lines=0, nodata_column=-1)
    cube[5, 360, 0] = np.inf
    res = cmf.robust_mf(cube, library, to_numpy=True)
    o = O.robust_mf_oracle(cube, library)
    assert res.out[5, 0, 3] == -9999.0 and o["out"][5, 0, 3] == -9999.0
    nod = o["out"][..., 3] == -9999.0
    assert np.array_equal(res.out[..., 3] == -9999.0, nod)
    assert score_close(res.out[..., 3][~nod], o["out"][..., 3][~nod]).all()
    cube = make_cube_numpy(3, 5, seed=62, abscf_full=library[:, 2], nodata_lines=0, nodata_column=2)
    res = cmf.robust_mf(cube, library, to_numpy=True)          # 3 rows << 72 bands: runs, no crash
    assert res.out.shape == (3, 5, 4) and res.status[2] == 1 and np.all(res.out[:, 2, 3] == -9999.0)
    assert np.array_equal(res.nuse, np.array([3, 3, 0, 3, 3]))
    cube = make_cube_numpy(40, 4, seed=63, abscf_full=library[:, 2], nodata_lines=0, nodata_column=-1)
    cube[1:, 351:360, 1] = -9999.0                              # column 1 keeps exactly one valid row
    res = cmf.robust_mf(cube, library, to_numpy=True)
    assert res.nuse[1] == 1 and res.status[1] == 0 and res.alphaidx[1] == 0 and np.isnan(res.out[0, 1, 3])
    o = O.robust_mf_oracle(cube[:, :, [0, 2, 3]], library)
    assert score_close(res.out[:, [0, 2, 3], 3], o["out"][..., 3]).all()


def test_column_shards_are_bit_identical(torch_cuda, library):
    """Sharding columns over ranks must not change any column's arithmetic."""
    torch = torch_cuda
    cube = torch.as_tensor(make_cube_numpy(200, 150, seed=5, abscf_full=library[:, 2])).cuda()
    full = cmf.robust_mf(cube, library, metadata=True)
    out = torch.full_like(full.out, -1.0)
    pieces = []
    for s0, s1 in [(0, 37), (37, 101), (101, 150)]:
        r = cmf.robust_mf(cube, library, columns=(s0, s1), out=out, out_column0=s0)
        pieces.append(r)
    assert torch.equal(out, full.out)
    assert torch.equal(torch.cat([r.alphaidx for r in pieces]), full.alphaidx)
    assert torch.equal(torch.cat([r.colstats for r in pieces], dim=1), full.colstats)
    # a rank that holds only its own column slice of the cube gets the same numbers
    sl = cube[:, :, 37:101].contiguous()
    r = cmf.robust_mf(sl, library)
    assert torch.equal(r.out, full.out[:, 37:101])


LOO_CASES = ["n100_p8", "n512_p72", "n2000_p72", "n100_p8_big"]


@pytest.mark.parametrize("name", LOO_CASES)
def test_looshrinkage_function_against_reference_golden(torch_cuda, golden_dir, name):
    g = np.load(os.path.join(golden_dir, "cmf_looshrinkage_cases.npz"))
    n, p, seed, scale = g[name + "_spec"]
    x = synth_columns(int(n), int(p), int(seed), float(scale))
    nll = np.zeros(201)
    Cm, mindex = cmf.looshrinkage(x - x.mean(axis=0), g["alphas"], nll, int(n))
    assert mindex == int(g[name + "_mindex"])
    np.testing.assert_allclose(nll, g[name + "_nll"], rtol=1e-10)
    np.testing.assert_allclose(Cm, g[name + "_C"], rtol=1e-11, atol=1e-14 * np.abs(g[name + "_C"]).max())


@pytest.mark.parametrize("name,rtol", [("n2000_p425", 1e-10), ("n300_p425", 1e-5)])
def test_looshrinkage_function_full_band_golden(torch_cuda, golden_dir, name, rtol):
    """The p = 425 goldens of the real reference through the function-level entry (wide path, float64 input).
    scipy's det is the running product of the LU pivots and over/underflows when a PREFIX does (cmf/robust_mf.py:111-113):
    the finite / inf pattern of nll[] must match the reference EXACTLY -- the wide path factorises the grid points for
    real (linalg.hip) instead of trusting the total log-determinant (round 1: one resp. two grid points differed).
    (n300: n < p, G is near-singular at the small alphas and the reference's own LU inverse is only good to ~1e-6.)"""
    g = np.load(os.path.join(golden_dir, "cmf_looshrinkage_cases.npz"))
    n, p, seed, scale = g[name + "_spec"]
    x = synth_columns(int(n), int(p), int(seed), float(scale))
    nll = np.zeros(201)
    Cm, mindex = cmf.looshrinkage(x - x.mean(axis=0), g["alphas"], nll, int(n))
    ref = g[name + "_nll"]
    assert mindex == int(g[name + "_mindex"])
    assert np.array_equal(np.isfinite(nll), np.isfinite(ref)) and np.array_equal(np.isnan(nll), np.isnan(ref))
    assert np.array_equal(np.isposinf(nll), np.isposinf(ref))
    both = np.isfinite(ref)
    np.testing.assert_allclose(nll[both], ref[both], rtol=rtol)
    np.testing.assert_allclose(np.diag(Cm), g[name + "_Cdiag"], rtol=1e-11)
    np.testing.assert_allclose(Cm[::17, ::13], g[name + "_Csub"], rtol=1e-10, atol=1e-13 * np.abs(g[name + "_Csub"]).max())


def test_looshrinkage_constant_band(torch_cuda, golden_dir):
    g = np.load(os.path.join(golden_dir, "cmf_looshrinkage_cases.npz"))
    n, p, seed, scale = g["const_band_spec"]
    x = synth_columns(int(n), int(p), int(seed), float(scale))
    x[:, 3] = np.float64(np.float32(1.25))
    nll = np.zeros(201)
    Cm, mindex = cmf.looshrinkage(x - x.mean(axis=0), g["alphas"], nll, int(n))
    assert mindex == -1 and np.all(np.isinf(nll))
    np.testing.assert_allclose(Cm, g["const_band_C"], rtol=1e-11, atol=1e-18)


def test_cov_wrapper(torch_cuda, golden_dir):
    g = np.load(os.path.join(golden_dir, "cmf_looshrinkage_cases.npz"))
    n, p, seed, scale = g["wrap_spec"]
    a = synth_columns(int(n), int(p), int(seed), float(scale))
    np.testing.assert_allclose(cmf.cov(a), g["wrap_cov"], rtol=1e-11)


def test_nodata_positive_is_refused(torch_cuda, library):
    cube = make_cube_numpy(16, 8, seed=1, abscf_full=library[:, 2])
    with pytest.raises(Exception, match="nodata"):
        cmf.robust_mf(cube, library, nodata=5.0)


def test_full_size_properties(torch_cuda, library):
    """BASELINE config 1 shape at reduced lines (memory of the test box permitting the full 598 columns):
    size-independent properties of the product -- NODATA exactly on invalid rows, per-column mean of the
    scores ~ 0 (sum_k (x_k - mu).w = 0), idempotent re-run bit-identical, sampled columns match the oracle."""
    torch = torch_cuda
    from srcfinder_amd.synth import make_cube_torch
    lines, samples = 4000, 598
    cube = make_cube_torch(lines, samples, seed=1234, abscf_full=library[:, 2], device="cuda")
    r1 = cmf.robust_mf(cube, library, metadata=True)
    r2 = cmf.robust_mf(cube, library, metadata=True)
    assert torch.equal(r1.out, r2.out) and torch.equal(r1.alphaidx, r2.alphaidx)
    sub = cube[:, 350:422, :]
    valid = ((sub >= 0) & torch.isfinite(sub)).all(dim=1)                      # [lines, samples]
    assert torch.equal(r1.out[..., 3] != -9999.0, valid)
    assert torch.equal(r1.nuse.long(), valid.sum(dim=0))
    cs = r1.colstats.cpu().numpy()
    okc = r1.status.cpu().numpy() == 0
    assert np.all(np.abs(cs[1][okc]) <= 1e-7 * cs[2][okc])
    cols = [0, 13, 199, 300, 597]
    host = cube[:, :, cols].cpu().numpy()
    o = O.robust_mf_oracle(host, library)
    got = r1.out[:, cols, :].cpu().numpy()
    nod = o["out"][..., 3] == -9999.0
    assert np.array_equal(got[..., 3] == -9999.0, nod)
    assert score_close(got[..., 3][~nod], o["out"][..., 3][~nod]).all()
    so = o["status"] == 0
    assert np.array_equal(r1.alphaidx.cpu().numpy()[cols][so], o["alphaidx"][so])


@pytest.fixture(scope="module")
def full_flightline(torch_cuda, library):
    """BASELINE.json config 2 at FULL size: 598 samples x 20000 lines x 425 bands float32 BIL (20.3 GB, generated on
    the device exactly as bench.py does) and its product.  ~26 GB of HBM with the workspace."""
    torch = torch_cuda
    from srcfinder_amd.synth import make_cube_torch
    lines, samples = 20000, 598
    cube = make_cube_torch(lines, samples, seed=1234, abscf_full=library[:, 2], device="cuda", nodata_column=samples // 3)
    res = cmf.robust_mf(cube, library, metadata=True)
    torch.cuda.synchronize()
    yield cube, res
    del cube, res
    torch.cuda.empty_cache()


def test_full_flightline_598x20000x425(torch_cuda, library, full_flightline):
    """The headline configuration itself: validity == NODATA placement exactly, idempotent re-run bit-identical,
    sum of a column's scores ~ 0, and 24 evenly spaced columns plus the all-NODATA column against the faithful oracle
    (alpha index and valid-row sets exact, scores 1e-4 relative)."""
    torch = torch_cuda
    cube, r1 = full_flightline
    lines, _, samples = cube.shape
    r2 = cmf.robust_mf(cube, library, metadata=True)
    assert torch.equal(r1.out, r2.out) and torch.equal(r1.alphaidx, r2.alphaidx) and torch.equal(r1.bgmeta, r2.bgmeta)
    assert torch.equal(r1.colstats, r2.colstats)
    del r2
    valid = torch.ones((lines, samples), dtype=torch.bool, device=cube.device)
    for b in range(350, 422):                                   # band by band: no 3.4 GB temporary
        x = cube[:, b, :]
        valid &= (x >= 0) & torch.isfinite(x)
    assert torch.equal(r1.out[..., 3] != -9999.0, valid)
    assert torch.equal(r1.nuse.long(), valid.sum(dim=0))
    assert torch.equal(r1.bgmeta[..., 1] != 0, valid & (r1.status == 0)[None, :] & (r1.alphaidx != 0)[None, :])
    st = r1.status.cpu().numpy()
    assert st[samples // 3] == 1 and (st == 0).sum() == samples - 1
    cs = r1.colstats.cpu().numpy()
    okc = st == 0
    assert np.all(np.abs(cs[1][okc]) <= 1e-7 * cs[2][okc])
    # RGB copy: bands 60, 42, 24 as float64 for every line of a processed column; 0 for the skipped column
    for k, b in enumerate((60, 42, 24)):
        want = cube[:, b, :].double()
        want[:, samples // 3] = 0.0
        assert torch.equal(r1.out[..., k], want)
    # oracle columns on every usable host core (spawned workers, OMP_NUM_THREADS=1; oracle/pool.py): 8 per core, at
    # least 120 of the 598 -- ~2 s each on one core, so the wall time is that of the 25 columns the serial check took
    from oracle import pool as OP
    ncheck = min(samples - 1, max(120, 8 * OP.usable_cores()))
    cols = sorted(set([int(round(i * (samples - 1) / (ncheck - 1))) for i in range(ncheck)] + [samples // 3]))
    a0, a1 = cmf.active_window("ch4", False)
    host = cube[:, a0 - 1:a1, :].index_select(2, torch.as_tensor(cols, device=cube.device)).cpu().numpy()
    o = OP.oracle_columns(host, library[a0 - 1:a1, 2], per_job=2)
    got = r1.out[:, cols, 3].cpu().numpy()
    nod = o["score"] == -9999.0
    assert np.array_equal(got == -9999.0, nod)
    assert score_close(got[~nod], o["score"][~nod]).all()
    so = o["status"] == 0
    assert np.array_equal(st[cols], o["status"])
    aidx = r1.alphaidx.cpu().numpy()[cols]
    assert np.array_equal(aidx[so], o["alphaidx"][so])
    assert np.array_equal(r1.nuse.cpu().numpy()[cols], o["nuse"])
    bg = r1.bgmeta[:, cols, :].cpu().numpy()
    assert not bg[..., 0].any()                                            # cluster band: 0 when k = 1 (:327)
    want_bg = np.where(nod | ~so[None, :], 0, o["alphaidx"][None, :]).astype(np.int16)
    assert np.array_equal(bg[..., 1], want_bg)
    print("full-size parity: %d oracle columns on %d workers in %.1f s" % (len(cols), o["workers"], o["seconds"]))


def test_full_flightline_co2_window_against_oracle(torch_cuda, library, full_flightline):
    """The CO2 window (309..391, p = 83: robust_mf.py:190-191) at the benchmark's size, on the 4x4x4 kernels: validity == NODATA
    placement over all 11.96 Mpixel, idempotent, and evenly spaced columns against the faithful oracle (alpha index exact,
    scores 1e-4) -- at least 48 columns, 4 per usable core."""
    torch = torch_cuda
    cube, _ = full_flightline
    lines, _, samples = cube.shape
    a0, a1 = cmf.active_window("co2", False)
    r1 = cmf.robust_mf(cube, library, gas="co2", metadata=True)
    r2 = cmf.robust_mf(cube, library, gas="co2", metadata=True)
    assert torch.equal(r1.out, r2.out) and torch.equal(r1.alphaidx, r2.alphaidx)
    del r2
    valid = torch.ones((lines, samples), dtype=torch.bool, device=cube.device)
    for b in range(a0 - 1, a1):
        x = cube[:, b, :]
        valid &= (x >= 0) & torch.isfinite(x)
    assert torch.equal(r1.out[..., 3] != -9999.0, valid)
    assert torch.equal(r1.nuse.long(), valid.sum(dim=0))
    from oracle import pool as OP
    ncheck = min(samples - 1, max(48, 4 * OP.usable_cores()))
    cols = sorted(set([int(round(i * (samples - 1) / (ncheck - 1))) for i in range(ncheck)] + [samples // 3]))
    host = cube[:, a0 - 1:a1, :].index_select(2, torch.as_tensor(cols, device=cube.device)).cpu().numpy()
    o = OP.oracle_columns(host, library[a0 - 1:a1, 2], per_job=2)
    got = r1.out[:, cols, 3].cpu().numpy()
    nod = o["score"] == -9999.0
    assert np.array_equal(got == -9999.0, nod)
    assert score_close(got[~nod], o["score"][~nod]).all()
    so = o["status"] == 0
    assert np.array_equal(r1.status.cpu().numpy()[cols], o["status"])
    assert np.array_equal(r1.alphaidx.cpu().numpy()[cols][so], o["alphaidx"][so])
    assert np.array_equal(r1.nuse.cpu().numpy()[cols], o["nuse"])
    print("CO2 window, full size: %d oracle columns on %d workers in %.1f s; routes %s"
          % (len(cols), o["workers"], o["seconds"], cmf.sweep_routes(cube, library, gas="co2")))
    del r1
    cmf._Workspace._bufs.clear()
    torch.cuda.empty_cache()


def test_full_flightline_wide_window_against_oracle(torch_cuda, library, full_flightline):
    """The full-band window (the route of the reference's -R window 5..420, robust_mf.py:186-187) at
    benchmark size: validity placement over the whole flightline, and TWO columns against the faithful oracle at
    20000 x 425 (~45 s each on a core, side by side in the spawned pool): alpha index exact, scores 1e-4 relative."""
    torch = torch_cuda
    from oracle import pool as OP
    cube, _ = full_flightline
    lines, bands, samples = cube.shape
    res = cmf.robust_mf(cube, library, active=(1, bands), metadata=True)
    torch.cuda.synchronize()
    valid = torch.ones((lines, samples), dtype=torch.bool, device=cube.device)
    for b in range(bands):
        x = cube[:, b, :]
        valid &= (x >= 0) & torch.isfinite(x)
    assert torch.equal(res.out[..., 3] != -9999.0, valid)
    assert torch.equal(res.nuse.long(), valid.sum(dim=0))
    cols = [97, 431]
    host = cube.index_select(2, torch.as_tensor(cols, device=cube.device)).cpu().numpy()
    o = OP.oracle_columns(host, library[:, 2], per_job=1)
    got = res.out[:, cols, 3].cpu().numpy()
    nod = o["score"] == -9999.0
    assert np.array_equal(got == -9999.0, nod)
    assert np.array_equal(res.status.cpu().numpy()[cols], o["status"])
    assert np.array_equal(res.alphaidx.cpu().numpy()[cols], o["alphaidx"])
    assert score_close(got[~nod], o["score"][~nod]).all()
    del res
    cmf._Workspace._bufs.clear()
    torch.cuda.empty_cache()


def test_host_cube_is_staged_compact_and_bit_identical(torch_cuda, library):
    """N3 ingest: a HOST cube (ndarray, any ENVI interleave) reaches the GPU as the active window + the RGB bands only
    (pinned chunks, asynchronous copies; ingest.py) and gives the product of the resident full cube bit for bit --
    unimodal, metadata, reflectance window, score-only output, column shards, multimodal with injected labels."""
    torch = torch_cuda
    from srcfinder_amd import ingest
    cube = make_cube_numpy(300, 70, seed=31, abscf_full=library[:, 2])
    dev_cube = torch.as_tensor(cube).cuda()
    for kw in (dict(metadata=True), dict(reflectance=True), dict(rgb_bands=()), dict(columns=(5, 41)),
               dict(gas="co2", rgb_bands=(10, 400, 424))):
        want = cmf.robust_mf(dev_cube, library, **kw)
        got = cmf.robust_mf(cube, library, **kw)
        assert torch.equal(got.out, want.out) and torch.equal(got.alphaidx, want.alphaidx)
        assert torch.equal(got.colstats, want.colstats) and torch.equal(got.nuse, want.nuse)
        assert got.modelparms == want.modelparms
        if kw.get("metadata"):
            assert torch.equal(got.bgmeta, want.bgmeta)
    # the staging itself: shape, band order, statistics, small chunks (several trips round the two buffers), pageable mode
    a0, a1 = cmf.active_window("ch4", False)
    for il, src in (("bil", cube), ("bip", np.ascontiguousarray(cube.transpose(0, 2, 1))),
                    ("bsq", np.ascontiguousarray(cube.transpose(1, 0, 2)))):
        for pinned in (True, False):
            cc = ingest.stage_cube(src, (a0, a1), (60, 42, 24), interleave=il, chunk_bytes=1 << 20, pinned=pinned, threads=3)
            assert cc.shape == (300, 75, 70) and cc.compact_rgb == (72, 73, 74) and cc.stats["bands_moved"] == 75
            bits = lambda t: t.contiguous().view(torch.int32)          # (the cube holds a NaN: compare bit patterns)
            assert torch.equal(bits(cc.tensor[:, :72]), bits(dev_cube[:, a0 - 1:a1]))
            assert torch.equal(bits(cc.tensor[:, 72:]), bits(dev_cube[:, [60, 42, 24]]))
    want = cmf.robust_mf(dev_cube, library, metadata=True)
    got = cmf.robust_mf(cc, library, metadata=True)
    assert torch.equal(got.out, want.out) and torch.equal(got.bgmeta, want.bgmeta)
    labels = np.random.default_rng(5).integers(0, 2, size=(300, 70))
    want = cmf.robust_mf(dev_cube, library, kmeans=2, labels=labels, metadata=True)
    got = cmf.robust_mf(cube, library, kmeans=2, labels=labels, metadata=True)
    assert torch.equal(got.out, want.out) and torch.equal(got.bgmeta, want.bgmeta)
    # and the way back
    host = ingest.fetch_product(want.out, chunk_bytes=1 << 18)
    assert np.array_equal(host, want.out.cpu().numpy())
    with pytest.raises(IndexError):
        ingest.stage_cube(cube, (a0, a1), (60, 42, 999))
    # a rank's column shard: only its samples are read and moved
    sh = ingest.stage_cube(cube, (a0, a1), (60, 42, 24), columns=(13, 40), chunk_bytes=1 << 19)
    assert sh.shape == (300, 75, 27) and sh.stats["columns"] == (13, 40)
    assert torch.equal(bits(sh.tensor[:, :72]), bits(dev_cube[:, a0 - 1:a1, 13:40]))
    want = cmf.robust_mf(dev_cube, library, metadata=True, columns=(13, 40))
    for got in (cmf.robust_mf(sh, library, metadata=True), cmf.robust_mf(cube, library, metadata=True, columns=(13, 40))):
        assert torch.equal(got.out, want.out) and torch.equal(got.bgmeta, want.bgmeta) and torch.equal(got.alphaidx, want.alphaidx)


def test_host_cubes_back_to_back_without_a_sync(torch_cuda, library):
    """robust_mf(host) does not synchronise on return, and the next call's compact cube may reuse the block the
    previous one just freed -- its staging copies (a private stream) must wait for the allocating stream, or the second
    flightline's H2D copies overwrite a cube whose score kernel is still reading it.  Two DIFFERENT cubes of the same shape
    back to back, large enough that the first call's kernels are still in flight when the second stages."""
    torch = torch_cuda
    a = make_cube_numpy(3000, 96, seed=41, abscf_full=library[:, 2])
    b = make_cube_numpy(3000, 96, seed=42, abscf_full=library[:, 2])
    want_a = cmf.robust_mf(torch.as_tensor(a).cuda(), library).out.clone()
    want_b = cmf.robust_mf(torch.as_tensor(b).cuda(), library).out.clone()
    torch.cuda.synchronize()
    for _ in range(3):
        ra = cmf.robust_mf(a, library)
        rb = cmf.robust_mf(b, library)               # no synchronisation in between
        oa, ob = ra.out, rb.out
        del ra, rb
        assert torch.equal(oa, want_a) and torch.equal(ob, want_b)


def test_full_flightline_shard_is_bit_identical(torch_cuda, library, full_flightline):
    """What one rank of 8 does with the same flightline: a COMPACT 75-column cube (its own allocation, flat extract
    kernel, three flightlines in flight on three streams) must reproduce columns 224..298 of the single-GPU product
    bit for bit -- every float64 sum is accumulated in an order that depends on the number of lines only
    (cmf_common.h, split counts) -- and so must the column range of the full cube processed in place."""
    torch = torch_cuda
    from srcfinder_amd.dist import shard_columns
    from srcfinder_amd.inflight import FlightlinePipeline
    cube, full = full_flightline
    samples = cube.shape[2]
    s0, s1 = shard_columns(samples, 8, 3)
    assert (s0, s1) == (224, 299)
    shard = cube[:, :, s0:s1].contiguous()
    with FlightlinePipeline(3, cube.device) as pipe:
        tickets = [pipe.submit(shard, library, metadata=True) for _ in range(4)]
        results = [t.synchronize() for t in tickets]
    inplace = cmf.robust_mf(cube, library, metadata=True, columns=(s0, s1))
    for r in results + [inplace]:
        assert torch.equal(r.out, full.out[:, s0:s1])
        assert torch.equal(r.bgmeta, full.bgmeta[:, s0:s1])
        assert torch.equal(r.alphaidx, full.alphaidx[s0:s1]) and torch.equal(r.nuse, full.nuse[s0:s1])
        assert torch.equal(r.status, full.status[s0:s1])
        assert torch.equal(r.colstats, full.colstats[:, s0:s1])


def test_shard_of_a_full_width_run_is_bit_identical(torch_cuda, library):
    """the same claim on a mid-size cube whose geometry exercises more than one row split of the covariance
    and the sweep (4500 lines -> 3 splits), for an odd-width shard at an odd offset, the CO2 window (16x16x4 kernels)
    and with the NLL curves compared too."""
    torch = torch_cuda
    from srcfinder_amd.synth import make_cube_torch
    cube = make_cube_torch(4500, 300, seed=99, abscf_full=library[:, 2], device="cuda", nodata_column=40)
    for kw in ({}, {"gas": "co2"}):
        full = cmf.robust_mf(cube, library, metadata=True, return_nll=True, **kw)
        for s0, s1 in ((37, 112), (0, 1), (150, 300)):
            a = cmf.robust_mf(cube[:, :, s0:s1].contiguous(), library, metadata=True, return_nll=True, **kw)
            b = cmf.robust_mf(cube, library, metadata=True, return_nll=True, columns=(s0, s1), **kw)
            for r in (a, b):
                assert torch.equal(r.out, full.out[:, s0:s1]) and torch.equal(r.bgmeta, full.bgmeta[:, s0:s1])
                assert torch.equal(r.alphaidx, full.alphaidx[s0:s1]) and torch.equal(r.nuse, full.nuse[s0:s1])
                assert torch.equal(r.colstats, full.colstats[:, s0:s1])
                assert torch.equal(r.nll.view(torch.int64), full.nll[s0:s1].view(torch.int64))


def test_score_kernels_agree_bit_for_bit(torch_cuda, library):
    """Every form of the score kernel (round 1's lane-stored records = 100, the production kernel with staged stores = 0,
    the 128-sample-block kernel with plain / non-temporal buffer loads = 10 / 11) accumulates a pixel's dot product in the
    same order: identical products on ragged geometries (odd widths and offsets, 1- and 2-column shards, line counts
    that are not multiples of the 8-line batch, -0.0 / +-inf values), with and without RGB."""
    torch = torch_cuda
    L = _ffi.lib()
    rng = np.random.default_rng(11)
    for lines, samples, cols in ((97, 70, None), (203, 151, (37, 112)), (164, 9, (2, 9)), (133, 1000, None), (141, 5, (1, 3))):
        cube = make_cube_numpy(lines, samples, seed=int(rng.integers(1 << 30)), abscf_full=library[:, 2], nodata_lines=2)
        cube[5, 360, min(3, samples - 1)] = -0.0            # -0.0 is a VALID value for the reference (not < 0)
        cube[6, 361, min(2, samples - 1)] = np.inf
        cube[7, 362, samples - 1] = -np.inf
        t = torch.as_tensor(cube).cuda()
        for rgb in ((60, 42, 24), ()):
            outs = []
            for variant in (100, 0, 5, 7):
                try:
                    L.sf_debug_set(1, variant)
                    outs.append(cmf.robust_mf(t, library, metadata=True, columns=cols, rgb_bands=rgb))
                finally:
                    L.sf_debug_set(1, 0)
            for r in outs[1:]:
                assert torch.equal(r.out, outs[0].out) and torch.equal(r.bgmeta, outs[0].bgmeta)
                np.testing.assert_allclose(r.colstats.cpu().numpy(), outs[0].colstats.cpu().numpy(), rtol=1e-12, equal_nan=True)
        c0, c1 = (0, samples) if cols is None else cols
        pick = list(range(c0, c1, max(1, (c1 - c0) // 12)))          # a dozen columns through the oracle
        ref = O.robust_mf_oracle(cube, library, columns=pick)
        got = outs[1].out.cpu().numpy()[:, [c - c0 for c in pick], -1]
        want = ref["out"][:, pick, -1]
        nod = want == -9999.0
        assert np.array_equal(got == -9999.0, nod)
        assert score_close(got[~nod], want[~nod]).all()


def test_column_profile_and_systematics(torch_cuda, library):
    """N2: triage column profile of a real product of the pipeline on the GPU against the numpy restatement (itself
    pinned by test_triage_profile_golden), and the rolling-median flag rule on a planted column."""
    from oracle import triage_oracle as TO
    from srcfinder_amd import triage
    cube = make_cube_numpy(300, 70, seed=71, abscf_full=library[:, 2], nodata_column=11)
    res = cmf.robust_mf(torch_cuda.as_tensor(cube).cuda(), library)
    prof = triage.column_profile(res.out)
    want = TO.column_profile(res.out[..., 3].cpu().numpy())
    assert np.array_equal(prof[0], want[0])
    assert np.array_equal(np.isnan(prof[1]), np.isnan(want[1])) and np.isnan(prof[1][11])
    f = ~np.isnan(want[1])
    np.testing.assert_allclose(prof[1:, f], want[1:, f], rtol=2e-5)
    rob = triage.column_profile(res.out, robust=True)
    wantr = TO.column_profile_robust(res.out[..., 3].cpu().numpy())
    assert np.array_equal(rob[0], wantr[0]) and np.isnan(rob[1][11])
    assert np.array_equal(rob[1:, f], wantr[1:, f])         # order statistics of float32 values: exact
    avg = prof[1].copy()
    avg[40] += 50 * np.nanstd(avg)                       # a planted systematic column
    coldiff, sigma, counts = triage.systematics_flags(avg)
    assert sigma > 0 and counts[2] >= 1 and np.nanargmax(coldiff) == 40


def test_column_profile_against_reference_golden(torch_cuda, golden_dir):
    """N2 pinned: sf_cmf_column_profile / _robust on the GPU against the CSV tables the real triage/cmf_profile.py wrote
    (tests/golden/triage_profile.npz).  Counts and order statistics (median of float32 values, MAD, 'nearest'
    percentiles, min, max) exact; mean / std to the rounding of numpy's float32 reductions."""
    import importlib.util
    from srcfinder_amd import triage
    spec = importlib.util.spec_from_file_location("gen_golden_triage", os.path.join(golden_dir, "gen_golden_triage.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    g = np.load(os.path.join(golden_dir, "triage_profile.npz"))
    img = gen.product(int(g["lines"]), int(g["samples"]), int(g["seed"]))
    prof = triage.column_profile(img).T
    want = g["plain"]
    assert np.array_equal(prof[:, 0], want[:, 0])
    assert np.array_equal(np.isnan(prof), np.isnan(want))
    f = ~np.isnan(want[:, 1])
    assert np.array_equal(np.float32(prof[f][:, 3:]), np.float32(want[f][:, 3:]))   # min, max (the CSV has 16 digits)
    np.testing.assert_allclose(prof[f][:, 1:3], want[f][:, 1:3], rtol=2e-5, atol=1e-4)
    rob = triage.column_profile(img, robust=True).T
    assert np.array_equal(np.float32(rob), np.float32(g["robust"]), equal_nan=True)
    # more lines than the LDS-resident sort holds (40000 > 32768; the reference has no cap, cmf_profile.py:124-127): the radix
    # selection from global memory, against the table the real script wrote for that product
    gb = np.load(os.path.join(golden_dir, "triage_profile_big.npz"))
    big = gen.product(int(gb["lines"]), int(gb["samples"]), int(gb["seed"]))
    robb = triage.column_profile(big, robust=True).T
    assert np.array_equal(np.float32(robb), np.float32(gb["robust"]), equal_nan=True)
    pb = triage.column_profile(big).T
    assert np.array_equal(pb[:, 0], gb["plain"][:, 0]) and np.array_equal(np.isnan(pb), np.isnan(gb["plain"]))
    # and the two kernels against each other on a product both can take: the same numbers
    from srcfinder_amd import _ffi
    short = torch_cuda.as_tensor(np.ascontiguousarray(big[:32768])).cuda()
    a = triage.column_profile(short, robust=True)
    longer = torch_cuda.as_tensor(np.ascontiguousarray(np.concatenate([big[:32768], np.full((1, big.shape[1], 4), -9999.0)]))).cuda()
    b = triage.column_profile(longer, robust=True)          # 32769 lines -> the radix-select kernel; the extra line is NODATA
    assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def test_narrow_cube_extract_is_bit_identical(torch_cuda, library):
    """A compact narrow cube (one rank's shard) takes the flat extract kernel; forcing the blocked kernel on the
    same cube, and cutting the same columns out of a wider cube, must give bit-identical products."""
    L = _ffi.lib()
    wide = make_cube_numpy(203, 150, seed=77, abscf_full=library[:, 2], nodata_column=33)
    narrow = np.ascontiguousarray(wide[:, :, 20:95])
    try:
        L.sf_debug_set(6, 0)
        a = cmf.robust_mf(narrow, library, metadata=True, to_numpy=True)
        L.sf_debug_set(6, 1)
        b = cmf.robust_mf(narrow, library, metadata=True, to_numpy=True)
    finally:
        L.sf_debug_set(6, 0)
    c = cmf.robust_mf(wide, library, metadata=True, to_numpy=True, columns=(20, 95))
    for x, y in ((a, b), (a, c)):
        assert np.array_equal(x.out, y.out) and np.array_equal(x.bgmeta, y.bgmeta)
        assert np.array_equal(x.alphaidx, y.alphaidx) and np.array_equal(x.nuse, y.nuse)
        assert np.array_equal(x.colstats, y.colstats, equal_nan=True)


def _k2_cube(g, library):
    lines, samples = int(g["lines"]), int(g["samples"])
    cube = make_cube_numpy(lines, samples, seed=int(g["seed"]), abscf_full=library[:, 2], nodata_column=int(g["nodata_column"]))
    b0, b1, f = g["bright"]
    cube[int(b0):int(b1)] *= np.float32(f)
    return cube


def test_multimodal_golden_with_injected_labels(torch_cuda, golden_dir, library):
    """N1: -k 2 golden of the real reference.  With the reference's own cluster labels injected, everything
    downstream of the clustering must match: NODATA placement, labels and alpha indices exactly, scores to 1e-4."""
    g = np.load(os.path.join(golden_dir, "cmf_K2_multimodal.npz"))
    cube = _k2_cube(g, library)
    lab = g["bgmeta"][:, :, 0].astype(np.int64)
    res = cmf.robust_mf(cube, library, kmeans=2, labels=lab, metadata=True, to_numpy=True)
    ref = g["out"]
    assert np.array_equal(res.out[..., 3] == -9999.0, ref[..., 3] == -9999.0)
    assert np.array_equal(res.out[..., :3], ref[..., :3])
    assert np.array_equal(res.bgmeta, g["bgmeta"])
    assert score_close(res.out[..., 3], ref[..., 3]).all()
    ok = g["colstats"][0] > 0
    assert np.array_equal(res.colstats[0][ok], g["colstats"][0][ok])
    np.testing.assert_allclose(res.colstats[1:, ok], g["colstats"][1:, ok], rtol=1e-6, atol=1e-9 * np.abs(ref[..., 3]).max())


def _bright_cube(g, library):
    cube = make_cube_numpy(int(g["lines"]), int(g["samples"]), seed=int(g["seed"]), abscf_full=library[:, 2],
                           nodata_column=int(g["nodata_column"]))
    for b0, b1, f in g["bright"]:
        cube[int(b0):int(b1)] *= np.float32(f)
    return cube


def _check_multimodal(res, ref_out, ref_bgmeta, ref_colstats):
    assert np.array_equal(res.out[..., 3] == -9999.0, ref_out[..., 3] == -9999.0)
    assert np.array_equal(res.out[..., :3], ref_out[..., :3])
    assert np.array_equal(res.bgmeta, ref_bgmeta)
    assert score_close(res.out[..., 3], ref_out[..., 3]).all()
    ok = ref_colstats[0] > 0
    assert np.array_equal(res.colstats[0][ok], ref_colstats[0][ok])
    np.testing.assert_allclose(res.colstats[1:, ok], ref_colstats[1:, ok], rtol=1e-6, atol=1e-9 * np.abs(ref_out[..., 3]).max())


def test_multimodal_cluster_rejection_golden(torch_cuda, golden_dir, library):
    """N1 -r: -k 3 -r golden of the real reference (a 38-40 row cluster rejected in four columns).  Rejected rows stay
    NODATA with cluster id -1, cluster 0's rows carry the model of all kept rows, statistics skip rejected rows."""
    g = np.load(os.path.join(golden_dir, "cmf_K3_reject.npz"))
    cube = _bright_cube(g, library)
    lab = np.abs(g["bgmeta"][:, :, 0].astype(np.int64))
    res = cmf.robust_mf(cube, library, kmeans=3, reject=True, labels=lab, metadata=True, to_numpy=True)
    assert (res.bgmeta[..., 0] < 0).any()
    _check_multimodal(res, g["out"], g["bgmeta"], g["colstats"])
    assert res.modelparms == str(g["modelparms"])


def test_multimodal_cluster_rejection_label_zero_and_all_rejected(torch_cuda, golden_dir, library):
    """The -r corner cases against the oracle: a small cluster with label 0 cannot be flagged (-0 == 0, :323), and a
    column whose clusters are ALL flagged proceeds without rejection but keeps the negative ids in bgmeta (:327-332)."""
    g = np.load(os.path.join(golden_dir, "cmf_K3_reject.npz"))
    cube = _bright_cube(g, library)
    lab = np.abs(g["bgmeta"][:, :, 0].astype(np.int64))
    swapped = np.where(lab == 0, 1, np.where(lab == 1, 0, lab))          # the 40-row cluster becomes label 0
    small = np.ones_like(lab)                                               # labels {1, 2}, both below 85 rows
    small[:, 0] = 2 - (np.arange(lab.shape[0]) % 2)
    cube_small = cube.copy()
    cube_small[150:] = -9999.0                                              # 143 valid rows in every column
    for cb, lb in ((cube, swapped), (cube_small, small)):
        res = cmf.robust_mf(cb, library, kmeans=3, reject=True, labels=lb, metadata=True, to_numpy=True)
        o = O.robust_mf_multimodal_oracle(cb, library, lb, reject=True)
        _check_multimodal(res, o["out"], o["bgmeta"], o["colstats"])
    assert (o["bgmeta"][..., 0] < 0).any()


def test_multimodal_full_regulariser_golden(torch_cuda, golden_dir, library):
    """N1 -f: -k 2 -f golden of the real reference (shrinkage target = covariance of the whole column; the selected
    alpha indices sit at the ends of the grid: 0, 2 and 200).  Generalised whitening path, sf_cmf_eigh_general."""
    g = np.load(os.path.join(golden_dir, "cmf_K2_full.npz"))
    cube = _bright_cube(g, library)
    lab = g["bgmeta"][:, :, 0].astype(np.int64)
    res = cmf.robust_mf(cube, library, kmeans=2, full=True, labels=lab, metadata=True, to_numpy=True)
    _check_multimodal(res, g["out"], g["bgmeta"], g["colstats"])
    assert res.modelparms == str(g["modelparms"])
    # and with the device's own labels against the oracle fed the same labels
    res = cmf.robust_mf(cube, library, kmeans=2, full=True, metadata=True, to_numpy=True, kmeans_seed=5)
    lab = np.where(res.labels != 255, res.labels, 0).astype(np.int64)
    o = O.robust_mf_multimodal_oracle(cube, library, lab, full=True)
    _check_multimodal(res, o["out"], o["bgmeta"], o["colstats"])


def test_looshrinkage_function_with_full_target(torch_cuda):
    """looshrinkage(I_zm, alphas, nll, n, I_reg) with a non-empty I_reg (robust_mf.py:99, :131): same index, NLL curve and
    final covariance as the faithful oracle."""
    from srcfinder_amd.synth import synth_columns
    x = synth_columns(900, 72, 321)
    sub = x[100:400] - x[100:400].mean(0)
    reg = x - x[100:400].mean(0)
    al = cmf.alpha_grid()
    nll_o, nll_g = np.zeros(len(al)), np.zeros(len(al))
    c_o, i_o = O.looshrinkage(sub, al, nll_o, 900, reg)
    c_g, i_g = cmf.looshrinkage(sub, al, nll_g, 900, reg)
    assert i_g == i_o
    fin = np.isfinite(nll_o)
    assert np.array_equal(np.isfinite(nll_g), fin)
    np.testing.assert_allclose(nll_g[fin], nll_o[fin], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(c_g, c_o, rtol=1e-10, atol=1e-12 * np.abs(c_o).max())


def test_full_regulariser_on_a_wide_window(torch_cuda, golden_dir, library):
    """-R -k 2 -f (p = 416): looshrinkage with a 416 x 416 target -- blocked Cholesky of the target, substitution
    whitening, unit-mode block Jacobi (sf_cmf_wide_stats_target) -- against the golden of the real reference, then
    against the oracle with a cluster of fewer rows than bands (S singular, T not) and with -r."""
    g = np.load(os.path.join(golden_dir, "cmf_R_K2_full.npz"))
    cube = make_cube_numpy(int(g["lines"]), int(g["samples"]), seed=int(g["seed"]), abscf_full=library[:, 2],
                           active=(5, 420), nodata_column=int(g["nodata_column"]))
    for b0, b1, f in g["bright"]:
        cube[int(b0):int(b1)] *= np.float32(f)
    lab = g["bgmeta"][:, :, 0].astype(np.int64)
    res = cmf.robust_mf(cube, library, reflectance=True, kmeans=2, full=True, labels=lab, metadata=True, to_numpy=True)
    _check_multimodal(res, g["out"], g["bgmeta"], g["colstats"])
    assert res.modelparms == str(g["modelparms"])
    cube = make_cube_numpy(700, 2, seed=78, abscf_full=library[:, 2], active=(5, 420), nodata_lines=1)
    cube[300:500] *= np.float32(1.25)
    lab = np.zeros((700, 2), np.int64)
    lab[300:500] = 1                                             # 200 rows < 416 bands
    lab[650:, 1] = 2
    for kw in (dict(kmeans=2, labels=np.minimum(lab, 1)), dict(kmeans=3, labels=lab, reject=True)):
        res = cmf.robust_mf(cube, library, reflectance=True, full=True, metadata=True, to_numpy=True, **kw)
        with np.errstate(all="ignore"):
            o = O.robust_mf_multimodal_oracle(cube, library, kw["labels"], reflectance=True, full=True,
                                              reject=kw.get("reject", False))
        _check_multimodal(res, o["out"], o["bgmeta"], o["colstats"])


def test_looshrinkage_function_with_full_target_wide(torch_cuda):
    """looshrinkage(I_zm, alphas, nll, n, I_reg) with 425 bands and a non-empty I_reg: index, NLL curve (finite pattern
    included: the determinants leave the float64 range at both ends of the grid) and final covariance as the oracle."""
    from srcfinder_amd.synth import synth_columns
    for scale in (1.0, 30.0):
        x = synth_columns(1200, 425, 322) * scale
        sub = x[100:700] - x[100:700].mean(0)
        reg = x - x[100:700].mean(0)
        al = cmf.alpha_grid()
        nll_o, nll_g = np.zeros(len(al)), np.zeros(len(al))
        with np.errstate(all="ignore"):
            c_o, i_o = O.looshrinkage(sub, al, nll_o, 1200, reg)
        c_g, i_g = cmf.looshrinkage(sub, al, nll_g, 1200, reg)
        assert i_g == i_o
        fin = np.isfinite(nll_o)
        assert np.array_equal(np.isfinite(nll_g), fin)
        np.testing.assert_allclose(nll_g[fin], nll_o[fin], rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(c_g, c_o, rtol=1e-10, atol=1e-12 * np.abs(c_o).max())


def test_multimodal_device_kmeans(torch_cuda, golden_dir, library):
    """Device k-means: deterministic (same seed -> same labels), finds the planted bright region the reference's
    MiniBatchKMeans found (>= 90 % agreement up to a permutation), and the pipeline downstream of ITS labels matches
    the oracle run with the same labels."""
    g = np.load(os.path.join(golden_dir, "cmf_K2_multimodal.npz"))
    cube = _k2_cube(g, library)
    a = cmf.robust_mf(cube, library, kmeans=2, metadata=True, to_numpy=True, kmeans_seed=3)
    b = cmf.robust_mf(cube, library, kmeans=2, metadata=True, to_numpy=True, kmeans_seed=3)
    assert np.array_equal(a.labels, b.labels) and np.array_equal(a.out, b.out)
    ref_lab = g["bgmeta"][:, :, 0]
    valid = a.labels != 255
    for c in range(cube.shape[2]):
        v = valid[:, c]
        if not v.any():
            continue
        agree = np.mean(a.labels[v, c] == ref_lab[v, c])
        assert max(agree, 1.0 - agree) >= 0.9, (c, agree)
    lab = np.where(valid, a.labels, 0).astype(np.int64)
    o = O.robust_mf_multimodal_oracle(cube, library, lab)
    assert np.array_equal(a.out[..., 3] == -9999.0, o["out"][..., 3] == -9999.0)
    assert np.array_equal(a.bgmeta, o["bgmeta"])
    assert score_close(a.out[..., 3], o["out"][..., 3]).all()


def test_lowrank_factorisation_of_sweep_coefficients(torch_cuda):
    """cmf_lowrank.hip: B_ji = beta_i/(n beta_i lam_j + alpha_i) = U W with K = 28 (or 36 for a wider eigenvalue range),
    W rows orthonormal, error at the rounding level of B -- on the eigenvalue spectra of flightline-like columns; a
    singular spectrum (n < p) is flagged for the full-rank sweep instead.  The factored matrix is the row-scaled
    B' = diag(lam) B (the sweep multiplies it by the whitened squares z_j / lam_j)."""
    torch = torch_cuda
    from srcfinder_amd.synth import synth_columns
    L = _ffi.lib()
    lams, ns = [], []
    for seed, rows in ((100, 20000), (101, 20000), (104, 3000), (105, 3000), (106, 90), (107, 60)):
        x = synth_columns(rows, 72, seed)
        x -= x.mean(0)
        S = np.cov(x.T)
        dd = np.sqrt(np.diag(S))
        lams.append(np.linalg.eigvalsh(S / np.outer(dd, dd)))
        ns.append(rows)
    wide = np.sort(np.r_[30.0, 8.0, 2.0, 0.5, np.exp(np.linspace(np.log(0.1), np.log(3e-4), 68))])   # a densely populated 5-decade spectrum: rank 28 is not enough (numpy QRCP: 32)
    lams.append(wide * 72.0 / wide.sum())
    ns.append(20000)
    lam = np.ascontiguousarray(np.stack(lams))
    nuse = np.asarray(ns, dtype=np.int32)
    nc, p = lam.shape
    al = cmf.alpha_grid()
    d = lambda a: torch.as_tensor(a).cuda()
    uf = torch.zeros((nc, 18 * 9 * 16), dtype=torch.float64, device="cuda")
    wf = torch.zeros((nc, 13 * 9 * 64), dtype=torch.float64, device="cuda")
    ok = torch.zeros(nc, dtype=torch.int32, device="cuda")
    status = torch.zeros(nc, dtype=torch.int32, device="cuda")
    lam_d, nuse_d, al_d = d(lam), d(nuse), d(al)          # keep the device copies alive across the launch
    _ffi.check(L.sf_debug_lowrank(_ffi.ptr(lam_d), _ffi.ptr(nuse_d), _ffi.ptr(status), _ffi.ptr(al_d), len(al), p, nc,
                                  _ffi.ptr(uf), _ffi.ptr(wf), _ffi.ptr(ok), _ffi.stream_ptr()), "sf_debug_lowrank")
    torch.cuda.synchronize()
    uf, wf, ok = uf.cpu().numpy(), wf.cpu().numpy(), ok.cpu().numpy()
    # rank 24 (code 3) or 28 (code 1) / n ~ p (condition 1e6): still factored / n < p (singular): full-rank sweep / 5.5 decades: rank 36
    assert all(v in (1, 3) for v in ok[:4]) and ok[4] in (1, 2, 3) and ok[5] == 0 and ok[6] == 2, ok
    for c in (0, 1, 2, 3, 4, 6):
        K = {3: 24, 1: 28, 2: 36}[int(ok[c])]
        n = float(nuse[c])
        beta = (1.0 - al) / (n - 1.0)
        B = lam[c][:, None] * beta[None, :] / (n * beta[None, :] * lam[c][:, None] + al[None, :])   # row-scaled B' [72, 201]
        U = -uf[c].reshape(18, 9, 4, 4).transpose(0, 2, 1, 3).reshape(72, 36)[:, :K]      # [jg, mg, q, n] -> [j, m]
        W = wf[c].reshape(13, 9, 4, 16).transpose(1, 2, 0, 3).reshape(36, 208)[:K]        # [M, mg, q, a16] -> [m, alpha]
        np.testing.assert_allclose(W @ W.T, np.eye(K), atol=1e-13)
        assert np.abs(U @ W[:, :201] - B).max() <= 2e-14 * np.abs(B).max()
        assert np.abs(W[:, 201:]).max() <= 1e-15


def test_rank36_sweep_on_wide_spectrum_columns(torch_cuda, library):
    """Columns whose correlation spectrum is densely spread over ~3.5 decades (1500 rows) need the rank-36 factorisation: the
    k_sweep4r<.., 9> path must be taken (lrok == 2) and agree with the faithful oracle (alpha index exact, scores 1e-4)."""
    torch = torch_cuda
    lines, samples, a0, a1 = 1500, 6, 351, 422
    cube = make_cube_numpy(lines, samples, seed=91, abscf_full=library[:, 2], nodata_column=-1, nodata_lines=0, inject=False)
    rng = np.random.default_rng(17)
    p = a1 - a0 + 1
    for c in range(samples):
        qmat, _ = np.linalg.qr(rng.standard_normal((p, p)))
        sd = np.sqrt(np.exp(np.linspace(np.log(1.0), np.log(3e-4), p)))          # variances over ~3.5 decades (numpy QRCP of the row-scaled coefficients: rank 32-33)
        x = 10.0 + 0.5 * (rng.standard_normal((lines, p)) * sd) @ qmat.T
        cube[:, a0 - 1:a1, c] = x.astype(np.float32)
    st = run_stages(torch, cube, a0, a1, library[a0 - 1:a1, 2])
    L = _ffi.lib()
    al = cmf.alpha_grid()
    d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    lam_d, nuse_d, al_d, st_d = d(st["lam"]), d(st["nuse"].astype(np.int32)), d(al), d(st["status"].astype(np.int32))
    uf = torch.zeros((samples, 18 * 9 * 16), dtype=torch.float64, device="cuda")
    wf = torch.zeros((samples, 13 * 9 * 64), dtype=torch.float64, device="cuda")
    ok = torch.zeros(samples, dtype=torch.int32, device="cuda")
    _ffi.check(L.sf_debug_lowrank(_ffi.ptr(lam_d), _ffi.ptr(nuse_d), _ffi.ptr(st_d), _ffi.ptr(al_d), len(al), p, samples,
                                  _ffi.ptr(uf), _ffi.ptr(wf), _ffi.ptr(ok), _ffi.stream_ptr()), "sf_debug_lowrank")
    torch.cuda.synchronize()
    assert (ok.cpu().numpy() == 2).any(), ok
    res = cmf.robust_mf(cube, library, to_numpy=True)
    o = O.robust_mf_oracle(cube, library)
    assert np.array_equal(res.alphaidx, o["alphaidx"])
    nod = o["out"][..., 3] == -9999.0
    assert np.array_equal(res.out[..., 3] == -9999.0, nod)
    assert score_close(res.out[..., 3][~nod], o["out"][..., 3][~nod]).all()


def test_flightline_pipeline_matches_sequential_calls(torch_cuda, library):
    """srcfinder_amd.inflight: three flightlines in flight on three streams (own scratch each) give the bits of three
    sequential robust_mf calls, whatever the interleaving."""
    torch = torch_cuda
    from srcfinder_amd.inflight import FlightlinePipeline
    cubes = [torch.as_tensor(make_cube_numpy(700 + 40 * i, 70, seed=300 + i, abscf_full=library[:, 2], nodata_column=5 * i)).cuda()
             for i in range(5)]
    seq = [cmf.robust_mf(c, library, metadata=True) for c in cubes]
    torch.cuda.synchronize()
    pipe = FlightlinePipeline(depth=3)
    tickets = [pipe.submit(c, library, metadata=True) for c in cubes]
    for t, s in zip(tickets, seq):
        r = t.synchronize()
        assert torch.equal(r.out, s.out) and torch.equal(r.bgmeta, s.bgmeta) and torch.equal(r.alphaidx, s.alphaidx)
        assert torch.equal(r.colstats, s.colstats)
    with pytest.raises(ValueError):
        pipe.submit(cubes[0], library, to_numpy=True)
    keys = [(str(pipe.device), int(st.cuda_stream)) for st in pipe.streams]
    assert all(k in cmf._Workspace._bufs for k in keys)
    pipe.close()
    assert not any(k in cmf._Workspace._bufs for k in keys)           # the slots' scratch is released


def test_wide_function_level_entries_separate_n_and_cov(torch_cuda):
    """looshrinkage() / cov() beyond 96 bands go through sf_cmf_wide_stats on float64 input; the n of beta may differ
    from the row count (the multimodal caller's convention) -- against the faithful oracle."""
    x = synth_columns(700, 130, 4242)
    sub = x[:400] - x[:400].mean(0)
    al = cmf.alpha_grid()
    nll_o, nll_g = np.zeros(len(al)), np.zeros(len(al))
    c_o, i_o = O.looshrinkage(sub, al, nll_o, 700)
    c_g, i_g = cmf.looshrinkage(sub, al, nll_g, 700)
    assert i_g == i_o
    fin = np.isfinite(nll_o)
    assert np.array_equal(np.isfinite(nll_g), fin)
    np.testing.assert_allclose(nll_g[fin], nll_o[fin], rtol=1e-9)
    np.testing.assert_allclose(c_g, c_o, rtol=1e-10, atol=1e-13 * np.abs(c_o).max())
    np.testing.assert_allclose(cmf.cov(x), np.cov(x.T), rtol=1e-10, atol=1e-13 * np.abs(np.cov(x.T)).max())


@pytest.mark.parametrize("rows,p", [(700, 100), (900, 128), (1300, 512)])
def test_wide_path_block_geometries(torch_cuda, rows, p):
    """Blocked Cholesky / blocked Jacobi at the corners of the wide range: 7 blocks with a short last one (p = 100), an
    even block count without a dummy slot (128), the largest supported window (512: 135 KB of LDS per block pair)."""
    x = synth_columns(rows, p, 9000 + p)
    x = x - x.mean(0)
    al = cmf.alpha_grid()
    nll_o, nll_g = np.zeros(len(al)), np.zeros(len(al))
    c_o, i_o = O.looshrinkage(x, al, nll_o, rows)
    c_g, i_g = cmf.looshrinkage(x, al, nll_g, rows)
    assert i_g == i_o
    # det over/underflow (see the p = 425 goldens): the same finite / inf pattern as the LU-prefix product of the oracle
    assert np.array_equal(np.isfinite(nll_o), np.isfinite(nll_g)), np.nonzero(np.isfinite(nll_o) != np.isfinite(nll_g))[0]
    both = np.isfinite(nll_o)
    np.testing.assert_allclose(nll_g[both], nll_o[both], rtol=1e-8)
    np.testing.assert_allclose(c_g, c_o, rtol=1e-9, atol=1e-12 * np.abs(c_o).max())


def test_linalg_wrappers_follow_scipy(torch_cuda):
    """inv / det / eig of the Python surface (cmf/robust_mf.py:72-90): LU semantics of scipy.linalg -- determinants as
    the running pivot product (overflow / underflow of a PREFIX sticks), LinAlgError for an exactly singular matrix."""
    import scipy.linalg as sla
    rng = np.random.default_rng(21)
    for n in (1, 2, 7, 72, 130):
        a = rng.normal(size=(n, n)) + n * np.eye(n) * 0.1
        np.testing.assert_allclose(cmf.det(a), O.det(a), rtol=1e-12)
        np.testing.assert_allclose(cmf.inv(a), O.inv(a), rtol=1e-9, atol=1e-12 * np.abs(O.inv(a)).max())
    spd = np.cov(rng.normal(size=(300, 40)).T)
    w, v = cmf.eig(spd)
    wo = np.sort(np.linalg.eigvalsh(spd))
    assert np.iscomplexobj(w) and np.allclose(np.sort(w.real), wo, rtol=1e-11) and np.all(w.imag == 0)
    np.testing.assert_allclose(spd @ v, v * w.real, rtol=0, atol=1e-11 * wo[-1])
    # above 96 rows the wide eigensolver (sf_cmf_eigh_wide): the reference's PCA calls eig(cov(...)) at p = 416 with -R -k 2
    # (cmf/robust_mf.py:310); a flightline-like spectrum (a cluster at the noise floor + a few signal directions) and a plain one
    for n, spec in ((130, None), (416, "cluster")):
        x = rng.normal(size=(3 * n, n))
        if spec == "cluster":
            x = 0.01 * x + rng.normal(size=(3 * n, 5)) @ rng.normal(size=(5, n))
        big = np.cov(x.T)
        w, v = cmf.eig(big)
        wo = np.sort(np.linalg.eigvalsh(big))
        # (absolute accuracy eps |A|, as LAPACK's geev has it: eig() shifts a matrix that is not diagonally dominant)
        assert np.allclose(np.sort(w.real), wo, rtol=1e-10, atol=2e-12 * wo[-1]) and np.all(w.imag == 0)
        np.testing.assert_allclose(big @ v, v * w.real, rtol=0, atol=1e-10 * wo[-1])
        np.testing.assert_allclose(v.T @ v, np.eye(n), rtol=0, atol=1e-10)
    with pytest.raises(NotImplementedError):
        cmf.eig(np.eye(513))
    # prefix overflow: diag(1e200, 1e200, 1e-300): the total determinant is 1e100, the running product reaches inf first
    d = np.diag([1e200, 1e200, 1e-300])
    assert O.det(d) == np.inf and cmf.det(d) == np.inf
    d = np.diag([1e-200, 1e-200, 1e300])
    assert O.det(d) == 0.0 and cmf.det(d) == 0.0
    sing = np.ones((5, 5))
    assert cmf.det(sing) == O.det(sing) == 0.0
    with pytest.raises(np.linalg.LinAlgError):
        O.inv(np.zeros((4, 4)))
    with pytest.raises(np.linalg.LinAlgError):
        cmf.inv(np.zeros((4, 4)))


def test_starved_clusters_take_the_exact_determinant_pass(torch_cuda, library):
    """Clusters of 15 .. 70 rows in a 72-band window (S singular, the shrinkage makes G_alpha regular): for the small alphas
    det(G_alpha) underflows, the first finite grid point is the minimum, and WHICH point is the first finite one is decided by
    the running pivot product of the reference's LU (robust_mf.py:111-113) -- tools/fuzz_multimodal.py seed 321 found the
    total log-determinant off by one grid point there.  sf_cmf_exact_det after stage 5: alpha indices equal to the oracle's,
    with the diagonal and with the full-column target; the function-level looshrinkage: the same finite / inf pattern."""
    cube = make_cube_numpy(500, 4, seed=4711, abscf_full=library[:, 2], nodata_lines=2, nodata_column=-1)
    lab = np.zeros((500, 4), np.int64)
    lab[60:100, 0] = 1
    lab[60:85, 1] = 1
    lab[60:130, 2] = 1
    lab[200:215, 3] = 1
    for full in (True, False):
        res = cmf.robust_mf(cube, library, kmeans=2, labels=lab, full=full, metadata=True, to_numpy=True)
        with np.errstate(all="ignore"):
            o = O.robust_mf_multimodal_oracle(cube, library, lab, full=full)
        assert np.array_equal(res.alphaidx, o["alphaidx"]), (full, res.alphaidx.tolist(), o["alphaidx"].tolist())
        assert np.array_equal(res.bgmeta, o["bgmeta"])
        assert np.array_equal(res.out[..., 3] == -9999.0, o["out"][..., 3] == -9999.0)
    from srcfinder_amd.synth import synth_columns
    al = cmf.alpha_grid()
    for seed, nsub, scale in ((6, 20, 1.0), (9, 10, 1.0), (6, 20, 0.2), (11, 30, 0.05)):
        x = synth_columns(500, 72, seed) * scale
        sub = x[100:100 + nsub] - x[100:100 + nsub].mean(0)
        reg = x - x[100:100 + nsub].mean(0)
        for target in (reg, []):
            nll_o, nll_g = np.zeros(201), np.zeros(201)
            with np.errstate(all="ignore"):
                _, i_o = O.looshrinkage(sub, al, nll_o, 500, target)
            _, i_g = cmf.looshrinkage(sub, al, nll_g, 500, target)
            assert np.array_equal(np.isfinite(nll_g), np.isfinite(nll_o)), (seed, nsub, scale, len(target))
            assert i_g == i_o, (seed, nsub, scale, len(target), i_g, i_o)


def test_blocked_lu_determinant(torch_cuda):
    """The blocked determinant kernel (panel of 16 columns in LDS, one column of the trailing part per thread) against
    scipy and against the unblocked kernel: sizes around the panel and thread-count boundaries, rows that need pivoting
    in every column, prefix over/underflow in the middle of a panel, a zero pivot column."""
    from srcfinder_amd import _ffi
    rng = np.random.default_rng(77)
    mats = [rng.normal(size=(n, n)) for n in (3, 15, 16, 17, 31, 33, 100, 257, 425, 512, 600)]
    mats.append(np.cov(rng.normal(size=(900, 425)).T) * 50.0)                  # SPD, the production shape
    m = rng.normal(size=(40, 40)); m[:, 7] = 0.0; mats.append(m)               # zero column: info > 0 -> 0.0
    mats.append(np.diag(np.r_[np.full(20, 1e30), np.full(20, 1e-30)]) + 1e-40 * rng.normal(size=(40, 40)))   # prefix -> inf
    mats.append(np.diag(np.r_[np.full(20, 1e-30), np.full(20, 1e30)]) + 1e-40 * rng.normal(size=(40, 40)))   # prefix -> 0
    for a in mats:
        ref = O.det(a)
        got = cmf.det(a)
        _ffi.lib().sf_debug_set(14, 1)
        try:
            old = cmf.det(a)
        finally:
            _ffi.lib().sf_debug_set(14, 0)
        if np.isfinite(ref) and ref != 0.0:
            np.testing.assert_allclose(got, ref, rtol=1e-9)
            np.testing.assert_allclose(old, ref, rtol=1e-9)
        else:
            assert got == ref and old == ref, (a.shape, got, old, ref)


def test_exact_determinant_window_on_the_flightline_path(torch_cuda, library):
    """The column loop at p = 425 (full-band window) takes the windowed exact-determinant pass (the grid points within
    24 of a lost one): the finite / inf pattern of every column's NLL curve equals the faithful oracle's."""
    cube = make_cube_numpy(520, 12, seed=31, abscf_full=library[:, 2], nodata_lines=2, nodata_column=5)
    res = cmf.robust_mf(cube, library, active=(1, 425), return_nll=True, to_numpy=True)
    o = O.robust_mf_oracle(cube, library, active=(1, 425), return_nll=True)
    ok = o["status"] == 0
    assert np.array_equal(res.alphaidx[ok], o["alphaidx"][ok])
    assert np.array_equal(np.isfinite(res.nll[ok]), np.isfinite(o["nll"][ok]))
    assert (~np.isfinite(o["nll"][ok])).sum() > 100              # the case does overflow
    f = np.isfinite(o["nll"]) & ok[:, None]
    np.testing.assert_allclose(res.nll[f], o["nll"][f], rtol=1e-8)
    # the rounds of four points per crossing (default) and the plain 24-point window in one round: the same curves
    from srcfinder_amd import _ffi
    _ffi.lib().sf_debug_set(15, 1)
    try:
        plain = cmf.robust_mf(cube, library, active=(1, 425), return_nll=True, to_numpy=True)
    finally:
        _ffi.lib().sf_debug_set(15, 0)
    assert np.array_equal(plain.alphaidx, res.alphaidx)
    assert np.array_equal(np.isfinite(plain.nll), np.isfinite(res.nll))
    np.testing.assert_allclose(plain.nll[f], res.nll[f], rtol=1e-12)


def test_cli_multimodal_flags_reach_the_device_path(torch_cuda, tmp_path, library):
    """cli_robust_mf -k 3 -r -f -m on an ENVI file == robust_mf(kmeans=3, reject=True, full=True) on the same cube
    (device k-means, default seed), header string with the reference's multimodal fields (robust_mf.py:246-259)."""
    from srcfinder_amd import cli_robust_mf, envi
    cube = make_cube_numpy(700, 6, seed=808, abscf_full=library[:, 2], nodata_lines=3, nodata_column=4)
    cube[200:420] *= np.float32(1.4)
    cube[600:630] *= np.float32(2.5)
    inp = str(tmp_path / "ang_mm_rdn")
    mm = envi.create_image(inp, {"lines": 700, "samples": 6, "bands": 425, "data ignore value": -9999}, np.float32, "bil")
    mm[...] = cube
    mm.flush()
    libpath = str(tmp_path / "ang_ch4_unit_3col_425chan.txt")
    np.savetxt(libpath, library, fmt="%.12f")
    outp = str(tmp_path / "ang_mm_ch4mf")
    assert cli_robust_mf.main(["-k", "3", "-r", "-f", "-m", inp, libpath, outp]) == 0
    prod, meta = envi.open_memmap(outp)
    bg, _ = envi.open_memmap(outp + "_bgmeta")
    ref = cmf.robust_mf(cube, np.loadtxt(libpath), kmeans=3, reject=True, full=True, metadata=True, to_numpy=True)
    assert np.array_equal(np.asarray(prod), ref.out) and np.array_equal(np.asarray(bg), ref.bgmeta)
    assert meta["model parameters"] == ("{ modelname=looshrinkage, bgmodel=multimodal, bgmodes=3, pcadim=6, reject=True, "
                                        "regfull=True, aminexp=-10.0, amaxexp=0.0, astep=0.05, reflectance=False, "
                                        "active_bands=[351, 422] }")
    assert len(np.unique(np.asarray(bg)[..., 0])) >= 3          # clusters were found (a rejected one shows as -id)


def test_wide_eigensolver_variants_agree(torch_cuda, golden_dir, library):
    """The blocked eigensolver of the wide path behind its tridiagonal preconditioner (default), the same sweeps from the plain
    Cholesky factor (debug key 10 = 6), the single-workgroup kernel it replaced (1) and the preconditioned route with every
    preconditioner refused (8) give the same product on the reference's reflectance configuration (p = 416): alpha indices exact,
    scores 1e-9."""
    L = _ffi.lib()
    cube = make_cube_numpy(300, 5, seed=4, abscf_full=library[:, 2], nodata_column=2)
    a = cmf.robust_mf(cube, library, reflectance=True, to_numpy=True)
    for variant in (1, 6, 7, 8):          # (6: never / 7: always the tridiagonal preconditioner -- the default since round 5 --, 8: always
                                           #  and every one refused afterwards: the single-workgroup fallback)
        L.sf_debug_set(10, variant)
        try:
            b = cmf.robust_mf(cube, library, reflectance=True, to_numpy=True)
        finally:
            L.sf_debug_set(10, 0)
        assert np.array_equal(a.alphaidx, b.alphaidx) and np.array_equal(a.status, b.status)
        nod = a.out[..., 3] == -9999.0
        assert np.array_equal(nod, b.out[..., 3] == -9999.0)
        np.testing.assert_allclose(a.out[..., 3][~nod], b.out[..., 3][~nod], rtol=1e-9,
                                   atol=1e-12 * np.abs(b.out[..., 3][~nod]).max())


@pytest.mark.parametrize("p", [260, 333, 425, 432])
def test_wide_sweep_forms_agree(torch_cuda, p):
    """The sweep kernels of the wide path on float32 rows -- k_wsweep8 (default for 256 < p <= 432: eight waves, wave-private
    operand slices, a wave's 14 or 13 column groups) against round 4's first form (sf_debug_set(24, 4): four waves, shared
    chunks) and the 32-row form (1) -- through sf_cmf_wide_stats on odd geometry: a row count that is no multiple of the
    64-row tile or the 640-row split, invalid rows holding NaN, NaN in the rows' padding, a column without valid rows.
    NLL curves to 1e-13, alpha indices and the inf pattern exact, a re-run bit-identical."""
    import torch
    L = _ffi.lib()
    dev = torch.device("cuda:0")
    ncols, rows = 11, 1500 + 37
    ps = (p + 3) // 4 * 4
    g = torch.Generator(device=dev); g.manual_seed(100 + p)
    base = 5.0 * torch.exp(-3.0 * torch.arange(p, device=dev) / (p - 1)) + 0.2
    xt = torch.full((ncols, rows, ps), float("nan"), dtype=torch.float32, device=dev)
    for c in range(ncols):
        lm = torch.randn((5, p), generator=g, device=dev) * 0.1 * base
        xt[c, :, :p] = base + torch.randn((rows, 5), generator=g, device=dev) @ lm + torch.randn((rows, p), generator=g, device=dev) * 0.01 * base
    mask = torch.ones((ncols, rows), dtype=torch.uint8, device=dev)
    mask[:, :5] = 0
    mask[:, 700] = 0
    xt[:, 700, :] = float("nan")
    mask[3, :] = 0                                   # status 1: no valid row
    mask[5, 64:128] = 0                              # a whole tile of invalid rows
    al = torch.as_tensor(cmf.alpha_grid(), device=dev); nalpha = al.numel()
    f64k = dict(dtype=torch.float64, device=dev)
    nuse = torch.empty(ncols, dtype=torch.int32, device=dev); mu = torch.empty((ncols, p), **f64k)
    ws = torch.empty(L.sf_cmf_workspace_bytes(rows, p, ncols, nalpha), dtype=torch.uint8, device=dev)
    P, st = _ffi.ptr, _ffi.stream_ptr()
    _ffi.check(L.sf_cmf_column_mean(P(xt), 0, P(mask), rows, p, ncols, P(nuse), P(mu), P(ws), st), "mean")
    out = {}
    try:
        for v in (0, 4, 1, 0):
            L.sf_debug_set(24, v)
            S = torch.empty((ncols, p, p), **f64k); d = torch.empty((ncols, p), **f64k); lam = torch.empty((ncols, p), **f64k)
            evec = torch.empty((ncols, p, p), **f64k); status = torch.empty(ncols, dtype=torch.int32, device=dev)
            nll = torch.empty((ncols, nalpha), **f64k); aidx = torch.empty(ncols, dtype=torch.int32, device=dev)
            _ffi.check(L.sf_cmf_wide_stats(P(xt), 0, P(mask), P(nuse), P(nuse), P(mu), P(al), nalpha, rows, p, ncols, P(S), P(d),
                                           P(lam), P(evec), P(status), P(nll), P(aidx), P(ws), st), "wide_stats")
            cur = dict(nll=nll.cpu().numpy(), aidx=aidx.cpu().numpy(), status=status.cpu().numpy())
            if v in out:
                assert np.array_equal(out[v]["nll"], cur["nll"], equal_nan=True)      # re-run bit-identical
            elif out:
                ref = out[0]
                assert np.array_equal(ref["status"], cur["status"]) and np.array_equal(ref["aidx"], cur["aidx"])
                assert np.array_equal(np.isfinite(ref["nll"]), np.isfinite(cur["nll"]))
                fin = np.isfinite(ref["nll"])
                np.testing.assert_allclose(cur["nll"][fin], ref["nll"][fin], rtol=1e-13)
            out.setdefault(v, cur)
    finally:
        L.sf_debug_set(24, 0)
    assert out[0]["status"][3] == 1 and (np.delete(out[0]["status"], 3) == 0).all()


def test_wide_preconditioner_over_column_groups(torch_cuda, library):
    """170 columns of a full-band window (p = 425): two column groups of 85, the second phase of the preconditioner's first half
    running over both groups in one set of launches (cmf_wide.hip: `phased`).  Against the same sweeps without the
    preconditioner (sf_debug_set(10, 6)): statuses and alpha indices exact, scores 1e-9; and a re-run is bit-identical."""
    L = _ffi.lib()
    cube = make_cube_numpy(700, 170, seed=11, abscf_full=library[:, 2], nodata_column=40)
    a = cmf.robust_mf(cube, library, active=(1, 425), to_numpy=True)
    a2 = cmf.robust_mf(cube, library, active=(1, 425), to_numpy=True)
    assert np.array_equal(a.out, a2.out, equal_nan=True) and np.array_equal(a.alphaidx, a2.alphaidx)
    L.sf_debug_set(10, 6)
    try:
        b = cmf.robust_mf(cube, library, active=(1, 425), to_numpy=True)
    finally:
        L.sf_debug_set(10, 0)
    assert np.array_equal(a.alphaidx, b.alphaidx) and np.array_equal(a.status, b.status)
    assert (a.status[np.arange(170) != 40] == 0).all()
    nod = a.out[..., 3] == -9999.0
    assert np.array_equal(nod, b.out[..., 3] == -9999.0)
    # (700 rows for 425 bands: an ill-conditioned covariance -- the two eigensolver routes agree to 1e-11 of the largest score)
    np.testing.assert_allclose(a.out[..., 3][~nod], b.out[..., 3][~nod], rtol=1e-9, atol=1e-10 * np.abs(b.out[..., 3][~nod]).max())


@pytest.mark.parametrize("p,decades", [(130, 2.0), (425, 2.0), (300, 6.0), (512, 3.0)])
def test_wide_tridiagonal_preconditioner(torch_cuda, p, decades):
    """csrc/cmf_wtri.hip through its test entry: for correlation matrices R = L L^T (benchmark-like and with a spectrum spread over
    `decades` more orders of magnitude) the preconditioned factor F satisfies F F^T = R to rounding (what keeps the eigensolver's
    accuracy the Jacobi's: F = L W' with W' orthogonal to rounding) and has nearly orthogonal columns (what saves the sweeps);
    the tridiagonal route's eigenvalues agree with numpy's to ~1e-9 (their own accuracy is not part of the product)."""
    import torch
    L = _ffi.lib()
    rng = np.random.default_rng(1000 + p)
    nb, n = 3, 3000
    Rs, Ls = [], []
    for m in range(nb):
        b = (5.0 * np.exp(-3.0 * np.arange(p) / (p - 1)) + 0.2) * 10.0 ** (-0.5 * decades * np.arange(p) / (p - 1))
        lm = rng.standard_normal((5, p)) * 0.1 * b
        x = b + rng.standard_normal((n, 5)) @ lm + rng.standard_normal((n, p)) * 0.01 * b
        x = np.float64(np.float32(x)); x -= x.mean(0)
        S = x.T @ x / (n - 1); d = np.sqrt(np.diag(S)); R = S / np.outer(d, d); R = 0.5 * (R + R.T)
        Rs.append(R); Ls.append(np.linalg.cholesky(R))
    R = np.stack(Rs); Lc = np.stack([l.T.copy() for l in Ls])
    dev = torch.device("cuda:0")
    Rt = torch.as_tensor(R, device=dev); Lt = torch.as_tensor(Lc, device=dev)
    F = torch.empty((nb, p, p), dtype=torch.float64, device=dev); tl = torch.empty((nb, p), dtype=torch.float64, device=dev)
    pf = torch.empty(nb, dtype=torch.int32, device=dev)
    ws = torch.empty(L.sf_debug_wtri_scratch_bytes(p, nb), dtype=torch.uint8, device=dev)
    _ffi.check(L.sf_debug_wtri(_ffi.ptr(Rt), _ffi.ptr(Lt), p, nb, _ffi.ptr(F), _ffi.ptr(tl), _ffi.ptr(pf), _ffi.ptr(ws),
                               _ffi.stream_ptr()), "wtri")
    torch.cuda.synchronize()
    assert (pf.cpu().numpy() == 0).all()
    Fh, tlh = F.cpu().numpy(), tl.cpu().numpy()
    for m in range(nb):
        Fm = Fh[m].T
        assert np.abs(Fm @ Fm.T - R[m]).max() <= 5e-14
        M = Fm.T @ Fm
        nn = np.sqrt(np.diag(M))
        C = M / np.outer(nn, nn) - np.eye(p)
        assert np.abs(C).max() <= 1e-7, np.abs(C).max()          # (measured: 1e-11 .. 1e-9)
        ref = np.linalg.eigvalsh(R[m])
        np.testing.assert_allclose(tlh[m], ref, rtol=1e-7, atol=1e-13 * ref[-1])


def test_multimodal_return_nll(torch_cuda, golden_dir, library):
    """return_nll with kmeans > 1: one NLL curve per (column, cluster); its argmin is the cluster's alpha index and it
    equals the curve looshrinkage() computes for the cluster's rows with n = the column's valid-row count."""
    g = np.load(os.path.join(golden_dir, "cmf_K2_multimodal.npz"))
    cube = _k2_cube(g, library)
    lab = g["bgmeta"][:, :, 0].astype(np.int64)
    res = cmf.robust_mf(cube, library, kmeans=2, labels=lab, return_nll=True, to_numpy=True)
    assert res.nll.shape == (cube.shape[2], 2, 201)
    for c in range(cube.shape[2]):
        for k in range(2):
            if res.status[c, k] == 0:
                assert int(np.argmin(res.nll[c, k])) == int(res.alphaidx[c, k])
            else:
                assert np.all(np.isinf(res.nll[c, k]))
    c, k = 0, 1
    x = np.float64(cube[:, 350:422, c])
    valid = ((~(x < 0)) & np.isfinite(x)).all(axis=1)
    rows = x[valid & (lab[:, c] == k)]
    nll = np.zeros(201)
    O.looshrinkage(rows - rows.mean(0), cmf.alpha_grid(), nll, int(valid.sum()))
    fin = np.isfinite(nll)
    np.testing.assert_allclose(res.nll[c, k][fin], nll[fin], rtol=1e-9)


def test_integration_md_ctypes_stub_runs(torch_cuda, library):
    """The ctypes stub printed in INTEGRATION.md section 2 is executed as written (its variables are the reference script's
    locals) and must reproduce robust_mf(): the documentation a maintainer would paste is tested code."""
    import re
    torch = torch_cuda
    md = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
    stub = next(b for b in blocks if "L.sf_cmf_run(" in b and "C.CDLL" in b)
    stub = stub.replace('C.CDLL("libsrcfinder_amd.so")', "C.CDLL(_ffi.LIB_PATH)")
    cube = make_cube_numpy(300, 9, seed=21, abscf_full=library[:, 2], nodata_column=4)
    nrows, nbands, ncols = cube.shape
    active = cmf.active_window("ch4", False)
    env = dict(np=np, _ffi=_ffi, img_mm=cube, nrows=nrows, nbands=nbands, ncols=ncols, active=list(active),
               abscf=np.ascontiguousarray(library[active[0] - 1:active[1], 2]), alphas=cmf.alpha_grid(), reflectance=False,
               rgb_bands=[60, 42, 24], nodata=-9999.0, savebgmeta=True,
               outimg_mm=np.zeros((nrows, ncols, 4)), bgimg_mm=np.zeros((nrows, ncols, 2), np.int16))
    exec(compile(stub, "INTEGRATION.md#stub", "exec"), env)
    ref = cmf.robust_mf(cube, library, metadata=True, to_numpy=True)
    assert np.array_equal(env["outimg_mm"], ref.out) and np.array_equal(env["bgimg_mm"], ref.bgmeta)
    assert np.array_equal(env["colnum"], ref.colstats[0]) and np.array_equal(env["colavg"], ref.colstats[1])


def test_column_with_a_single_valid_row(torch_cuda, library):
    """Found by tools/fuzz_parity.py: with ONE valid row numpy.cov (ddof 1) is NaN, every NLL is NaN, argmin returns 0,
    inv() of the NaN matrix does not raise, and the reference writes a NaN score with alpha index 0 (not 0.0, not NODATA)."""
    cube = make_cube_numpy(120, 4, seed=5, abscf_full=library[:, 2], nodata_column=-1, nodata_lines=0)
    cube[:, 380, 2] = -9999.0
    cube[57, 380, 2] = 1.0                        # the only valid row of column 2
    res = cmf.robust_mf(cube, library, metadata=True, to_numpy=True)
    with np.errstate(all="ignore"):
        o = O.robust_mf_oracle(cube, library)
    assert res.nuse[2] == 1 and o["nuse"][2] == 1
    assert np.array_equal(res.status, o["status"]) and res.status[2] == 0
    assert np.array_equal(res.alphaidx, o["alphaidx"]) and res.alphaidx[2] == 0
    assert np.array_equal(res.bgmeta, o["bgmeta"])
    assert np.isnan(res.out[57, 2, 3]) and np.isnan(o["out"][57, 2, 3])
    assert np.array_equal(np.isnan(res.out[..., 3]), np.isnan(o["out"][..., 3]))
    assert np.array_equal(res.out[..., 3] == -9999.0, o["out"][..., 3] == -9999.0)
    assert np.isnan(res.colstats[1, 2]) and np.isnan(o["colstats"][1, 2]) and res.colstats[0, 2] == 1
    ok = [0, 1, 3]
    assert score_close(res.out[:, ok, 3], o["out"][:, ok, 3]).all()


def test_sweep_kernels_agree_bit_for_bit(torch_cuda, library):
    """Round 3's streamed sweep kernel (k_sweep4s: one operand ring per tile, validity applied on t, 16-byte operand pairs)
    against round 2's (k_sweep4r, sf_debug_set(20, 1)): same accumulation order in every chain, so the NLL curves are equal
    bit for bit -- on windows of 72 and 70 bands (out-of-window bands switched off), with a ragged line count (a second,
    short row split; tiles that end inside a 16-row group), NaN / negative rows and an all-NODATA column.  Form 4 is the
    streamed kernel renormalising its running products after every tile instead of every fourth (scaling by powers of two:
    the same bits); form 5 the default without its rank-24 tier.  The default itself (rank 24 where the factorisation
    allows) gives the same alpha indices and product, its NLL curves agree to 1e-12."""
    from srcfinder_amd import _ffi
    L = _ffi.lib()
    cube = make_cube_numpy(2500, 21, seed=31, abscf_full=library[:, 2], nodata_column=4, nodata_lines=5)
    rng = np.random.default_rng(3)
    for _ in range(40):
        cube[rng.integers(5, 2500), 351 + rng.integers(0, 70), rng.integers(0, 21)] = rng.choice([np.nan, -1.0, np.inf])
    dev = torch_cuda.as_tensor(cube).cuda()
    for active in ((351, 422), (352, 421)):
        runs = []
        for form in (1, 4, 5, 0):
            L.sf_debug_set(20, form)
            try:
                runs.append(cmf.robust_mf(dev, library, active=active, metadata=True, to_numpy=True, return_nll=True))
            finally:
                L.sf_debug_set(20, 0)
        new = runs[-2]                                            # form 5: the streamed kernel at ranks 28 / 36
        for old in runs[:-2]:
            assert np.array_equal(old.status, new.status) and np.array_equal(old.alphaidx, new.alphaidx)
            assert np.array_equal(old.nll, new.nll, equal_nan=True), active
            assert np.array_equal(old.out, new.out, equal_nan=True) and np.array_equal(old.bgmeta, new.bgmeta)
        assert (new.status == 0).sum() == 20 and np.isfinite(new.nll[new.status == 0]).any()
        # the default adds the rank-24 tier (round 5): the same factorisation stopped four steps earlier where the trailing
        # block is already at the rounding floor -- same alpha, same product, NLL to rounding
        dflt = runs[-1]
        assert cmf.sweep_routes(dev, library, active=active)["rank24"] >= 10
        assert np.array_equal(dflt.status, new.status) and np.array_equal(dflt.alphaidx, new.alphaidx)
        assert np.array_equal(dflt.out, new.out, equal_nan=True) and np.array_equal(dflt.bgmeta, new.bgmeta)
        fin = np.isfinite(new.nll)
        assert np.array_equal(np.isfinite(dflt.nll), fin) and np.allclose(dflt.nll[fin], new.nll[fin], rtol=1e-12, atol=0)


def test_preconditioned_eigensolver_against_the_plain_sweeps(torch_cuda, library):
    """Round 6 (csrc/cmf_eigh_pre.h; sf_debug_set(7, 2) -- measured no faster than the plain sweeps, so not the default:
    profiles/r06_eigh_precond.md): on the CH4 / CO2 window sizes the one-sided Jacobi starts from a factor rotated by a
    tridiagonal preconditioner (Householder, bisection, twisted vectors, Newton-Schulz) instead of from the Cholesky factor:
    one sweep instead of 8-9.  The result is still the Jacobi's on a factor of R that is exact to rounding: against the plain
    sweeps the eigenvalues agree to 1e-12 relative, statuses, alpha indices, metadata and NODATA placement
    are identical, NLL curves and scores agree to 1e-9 -- on healthy columns, a starved column (fewer valid rows than bands: R is
    singular, the preconditioner refuses it and the plain route's fallback serves it), a column with a constant band (status 2)
    and an all-NODATA one; and the eigenpairs hold against numpy.linalg.eigh with orthonormal vectors."""
    L = _ffi.lib()
    cube = make_cube_numpy(1500, 14, seed=91, abscf_full=library[:, 2], nodata_column=4, nodata_lines=5)
    cube[60:, :, 7] = -9999.0                                   # 55 valid rows in column 7: fewer than 72 / 83 bands
    cube[:, 360, 9] = 3.25                                      # a constant band inside both windows' ... (CH4 only: 351..422)
    dev = torch_cuda.as_tensor(cube).cuda()
    for gas in ("ch4", "co2"):
        runs = []
        for knob in (2, 0):
            L.sf_debug_set(7, knob)
            try:
                runs.append(cmf.robust_mf(dev, library, gas=gas, metadata=True, to_numpy=True, return_nll=True))
            finally:
                L.sf_debug_set(7, 0)
        new, old = runs
        assert np.array_equal(new.status, old.status) and np.array_equal(new.alphaidx, old.alphaidx), gas
        assert np.array_equal(new.bgmeta, old.bgmeta) and np.array_equal(new.nuse, old.nuse)
        nod = old.out[..., 3] == -9999.0
        assert np.array_equal(new.out[..., 3] == -9999.0, nod)
        fin = np.isfinite(old.nll)
        assert np.array_equal(np.isfinite(new.nll), fin) and np.allclose(new.nll[fin], old.nll[fin], rtol=1e-9, atol=0)
        a, b = new.out[..., 3][~nod], old.out[..., 3][~nod]
        okv = np.isfinite(b)
        assert np.array_equal(np.isfinite(a), okv) and np.allclose(a[okv], b[okv], rtol=1e-9, atol=1e-12 * np.abs(b[okv]).max())
        assert (new.status == 0).sum() >= 10
    # the stage entry alone: eigenpairs of both routes against numpy, and the sweeps each needed (the rotation log's first slot)
    a0, a1 = cmf.active_window("ch4", False)
    res = {}
    for knob in (2, 0):
        L.sf_debug_set(7, knob)
        try:
            res[knob != 2] = run_stages(torch_cuda, cube, a0, a1, library[a0 - 1:a1, 2])
        finally:
            L.sf_debug_set(7, 0)
    for c in (0, 1, 2, 3, 5, 6):
        S = res[0]["S"][c]
        dd = np.sqrt(np.diag(S))
        R = S / np.outer(dd, dd)
        w = np.linalg.eigvalsh(R)
        for knob in (0, 1):
            lam, ev = res[knob]["lam"][c], res[knob]["evec"][c]
            o = np.argsort(lam)
            assert np.max(np.abs(lam[o] - w) / w) < 1e-12, (c, knob)
            assert np.abs(R @ ev.T - ev.T * lam).max() < 1e-13 * w[-1] * 10
            assert np.abs(ev @ ev.T - np.eye(len(w))).max() < 1e-12
    assert np.array_equal(res[0]["status"], res[1]["status"]) and np.array_equal(res[0]["aidx"], res[1]["aidx"])


def test_zero_target_scores_nan_like_the_reference(torch_cuda, library):
    """A library without absorption in the window makes the target t = abscf * mu the zero vector: the reference divides by
    t^T C^-1 t = 0 (robust_mf.py:380-381) and every valid row scores NaN (0 / 0); status stays 0, the alpha index is kept,
    NODATA rows stay NODATA.  (Rounds 1-2 returned status 2 and zeros here.)"""
    lib0 = library.copy()
    lib0[:, 2] = 0.0
    cube = make_cube_numpy(160, 5, seed=11, abscf_full=library[:, 2], nodata_column=1, nodata_lines=3)
    res = cmf.robust_mf(cube, lib0, metadata=True, to_numpy=True)
    with np.errstate(all="ignore"):
        o = O.robust_mf_oracle(cube, lib0)
    assert np.array_equal(res.status, o["status"]) and list(res.status) == [0, 1, 0, 0, 0]
    solved = res.status == 0
    assert np.array_equal(res.alphaidx[solved], o["alphaidx"][solved]) and np.array_equal(res.bgmeta, o["bgmeta"])
    assert np.array_equal(np.isnan(res.out[..., 3]), np.isnan(o["out"][..., 3])) and np.isnan(res.out[10, 0, 3])
    assert np.array_equal(res.out[..., 3] == -9999.0, o["out"][..., 3] == -9999.0)
    assert np.array_equal(res.out[..., :3], o["out"][..., :3])
    assert np.array_equal(res.colstats[0], o["colstats"][0]) and np.isnan(res.colstats[1, 0]) and np.isnan(o["colstats"][1, 0])


def test_multimodal_and_wide_column_shards(torch_cuda, golden_dir, library):
    """columns=(s0, s1) on the multimodal branch (injected labels, -r) and on a wide window: the shard's product equals
    the same columns of the full run bit for bit (per-column arithmetic independent of the sharding)."""
    g = np.load(os.path.join(golden_dir, "cmf_K3_reject.npz"))
    cube = _bright_cube(g, library)
    lab = np.abs(g["bgmeta"][:, :, 0].astype(np.int64))
    full = cmf.robust_mf(cube, library, kmeans=3, reject=True, labels=lab, metadata=True, to_numpy=True)
    part = cmf.robust_mf(cube, library, kmeans=3, reject=True, labels=lab, metadata=True, to_numpy=True, columns=(1, 4))
    assert np.array_equal(part.out, full.out[:, 1:4], equal_nan=True) and np.array_equal(part.bgmeta, full.bgmeta[:, 1:4])
    assert np.array_equal(part.alphaidx, full.alphaidx[1:4]) and np.array_equal(part.colstats, full.colstats[:, 1:4], equal_nan=True)
    cube = make_cube_numpy(260, 7, seed=12, abscf_full=library[:, 2], nodata_column=3)
    full = cmf.robust_mf(cube, library, reflectance=True, metadata=True, to_numpy=True)
    part = cmf.robust_mf(cube, library, reflectance=True, metadata=True, to_numpy=True, columns=(2, 6))
    assert np.array_equal(part.out, full.out[:, 2:6], equal_nan=True) and np.array_equal(part.bgmeta, full.bgmeta[:, 2:6])
    assert np.array_equal(part.alphaidx, full.alphaidx[2:6])


def test_wide_sweep_rank_factored_against_the_unfactored_kernel(torch_cuda, library):
    """Round 5: on k_wsweep8's windows (257..432 bands) the sweep's second product runs through the rank factorisation of its
    coefficient matrix (cmf_wlr.hip: proxy eigenvalues -> k_lowrank -> checked and completed against the real spectrum) where
    the column's spectrum allows it -- a noise-floor cluster plus a few signal directions, as a flightline column has -- and
    unfactored otherwise (sf_debug_set(24, 5): every column unfactored).  One cube holds both kinds: same status, same alpha
    indices, the same product bit for bit (the scores depend on alpha only), NLL curves to 1e-11; and both routes against the
    faithful oracle."""
    from srcfinder_amd import _ffi
    torch = torch_cuda
    L = _ffi.lib()
    lines, samples, active = 5000, 6, (5, 420)
    p = active[1] - active[0] + 1
    cube = make_cube_numpy(lines, samples, seed=58, abscf_full=library[:, 2], active=active, nodata_column=2, nodata_lines=3)
    rng = np.random.default_rng(8)
    for c in (4, 5):                                             # spectra spread densely over 4.5 decades: not factored
        qmat, _ = np.linalg.qr(rng.standard_normal((p, p)))
        sd = np.sqrt(np.exp(np.linspace(0.0, -4.5 * np.log(10.0), p)))
        x = 10.0 + 0.5 * (rng.standard_normal((lines, p)) * sd) @ qmat.T
        cube[3:, active[0] - 1:active[1], c] = x[3:].astype(np.float32)
    dev = torch.as_tensor(cube).cuda()
    routes = cmf.sweep_routes(dev, library, active=active)
    assert routes["factored"] >= 3 and routes["unfactored"] >= 1 and routes["skipped"] == 1, routes
    assert all(24 <= k <= 31 for k in routes["ranks"]), routes
    runs = []
    for knob in (0, 5):
        L.sf_debug_set(24, knob)
        try:
            runs.append(cmf.robust_mf(dev, library, active=active, metadata=True, to_numpy=True, return_nll=True))
        finally:
            L.sf_debug_set(24, 0)
    a, b = runs
    assert np.array_equal(a.status, b.status) and np.array_equal(a.alphaidx, b.alphaidx) and np.array_equal(a.bgmeta, b.bgmeta)
    assert np.array_equal(a.out, b.out, equal_nan=True)
    fin = np.isfinite(b.nll)
    assert np.array_equal(np.isfinite(a.nll), fin) and not np.array_equal(a.nll, b.nll)     # two different evaluations ...
    assert np.allclose(a.nll[fin], b.nll[fin], rtol=1e-11, atol=0)                         # ... of the same curves
    with np.errstate(all="ignore"):
        o = O.robust_mf_oracle(cube, library, active=active)
    _compare_run(a, o, lines, samples)


def test_multimodal_on_a_wide_window(torch_cuda, library):
    """-R -k 2 (reflectance window 5..420, p = 416) with injected labels and with -r: the per-cluster statistics go through
    sf_cmf_wide_stats (separate row count and n), the cluster score kernel reads its filter from global memory."""
    cube = make_cube_numpy(1300, 3, seed=77, abscf_full=library[:, 2], nodata_column=-1, nodata_lines=2)
    cube[500:900] *= np.float32(1.3)
    lab = np.zeros((1300, 3), np.int64)
    lab[500:900] = 1
    lab[1250:, 1] = 2                                            # a 50-row cluster in column 1: rejected with -r
    for kw in (dict(kmeans=2, labels=np.minimum(lab, 1)), dict(kmeans=3, labels=lab, reject=True)):
        res = cmf.robust_mf(cube, library, reflectance=True, metadata=True, to_numpy=True, **kw)
        with np.errstate(all="ignore"):
            o = O.robust_mf_multimodal_oracle(cube, library, kw["labels"], reflectance=True, reject=kw.get("reject", False))
        assert np.array_equal(res.out[..., 3] == -9999.0, o["out"][..., 3] == -9999.0)
        assert np.array_equal(res.bgmeta, o["bgmeta"])
        assert score_close(res.out[..., 3], o["out"][..., 3]).all()
    a = cmf.robust_mf(cube, library, reflectance=True, kmeans=2, metadata=True, to_numpy=True, kmeans_seed=1)    # device k-means
    b = cmf.robust_mf(cube, library, reflectance=True, kmeans=2, metadata=True, to_numpy=True, kmeans_seed=1)
    assert np.array_equal(a.labels, b.labels) and np.array_equal(a.out, b.out)
    found = a.labels[:, 0]
    agree = np.mean(found[2:] == np.minimum(lab, 1)[2:, 0])
    assert max(agree, 1 - agree) > 0.9                           # the brightened block is what it separates


def test_empirical_model_golden(torch_cuda, golden_dir, library):
    """-M empirical (C = the sample covariance, no shrinkage) against the golden of the real reference; with metadata the
    reference dies on an undefined name and so does the mirror."""
    g = np.load(os.path.join(golden_dir, "cmf_empirical.npz"))
    cube = make_cube_numpy(int(g["lines"]), int(g["samples"]), seed=int(g["seed"]), abscf_full=library[:, 2],
                           nodata_column=int(g["nodata_column"]))
    res = cmf.robust_mf(cube, library, model="empirical", to_numpy=True)
    ref = g["out"]
    assert np.array_equal(res.out[..., 3] == -9999.0, ref[..., 3] == -9999.0)
    assert np.array_equal(res.out[..., :3], ref[..., :3])
    assert score_close(res.out[..., 3], ref[..., 3]).all()
    ok = g["colstats"][0] > 0
    np.testing.assert_allclose(res.colstats[:, ok], g["colstats"][:, ok], rtol=1e-6, atol=1e-9 * np.abs(ref[..., 3]).max())
    assert res.modelparms == str(g["modelparms"]) and np.all(res.alphaidx == -1)
    with pytest.raises(NameError):
        cmf.robust_mf(cube, library, model="empirical", metadata=True)


def test_empirical_model_on_the_multimodal_and_wide_branches(torch_cuda, golden_dir, library):
    """-M empirical -k 2 (labels injected) and -M empirical -R (p = 416) against goldens of the real reference."""
    g = np.load(os.path.join(golden_dir, "cmf_empirical_K2.npz"))
    cube = make_cube_numpy(int(g["lines"]), int(g["samples"]), seed=int(g["seed"]), abscf_full=library[:, 2],
                           nodata_column=int(g["nodata_column"]))
    b0, b1, f = g["bright"]
    cube[int(b0):int(b1)] *= np.float32(f)
    res = cmf.robust_mf(cube, library, model="empirical", kmeans=2, labels=g["labels"], to_numpy=True)
    ref = g["out"]
    assert np.array_equal(res.out[..., 3] == -9999.0, ref[..., 3] == -9999.0) and np.array_equal(res.out[..., :3], ref[..., :3])
    assert score_close(res.out[..., 3], ref[..., 3]).all()
    assert res.modelparms == str(g["modelparms"]) and np.all(res.alphaidx[res.status == 0] == -1)
    ok = g["colstats"][0] > 0
    np.testing.assert_allclose(res.colstats[:, ok], g["colstats"][:, ok], rtol=1e-6, atol=1e-9 * np.abs(ref[..., 3]).max())
    with pytest.raises(NameError):
        cmf.robust_mf(cube, library, model="empirical", kmeans=2, labels=g["labels"], metadata=True)
    g = np.load(os.path.join(golden_dir, "cmf_empirical_R.npz"))
    cube = make_cube_numpy(int(g["lines"]), int(g["samples"]), seed=int(g["seed"]), abscf_full=library[:, 2],
                           nodata_column=int(g["nodata_column"]))
    cube = np.float32(np.clip(cube, -1e9, None) * (cube > 0) * 0.08 + cube * (cube <= 0))
    res = cmf.robust_mf(cube, library, model="empirical", reflectance=True, to_numpy=True)
    ref = g["out"]
    assert np.array_equal(res.out[..., 3] == -9999.0, ref[..., 3] == -9999.0)
    assert score_close(res.out[..., 3], ref[..., 3]).all()
    assert res.modelparms == str(g["modelparms"])
