#!/usr/bin/env python3
"""Generate the CMF golden vectors by EXECUTING THE REAL REFERENCE (development container only).

Run:  OMP_NUM_THREADS=1 python tests/golden/gen_golden.py
Needs /root/reference (read-only).  Writes small .npz files next to this script; they are the
committed parity pin for ``oracle/cmf_oracle.py`` and, through it, for the HIP path.

How the reference is driven:
  * ``spectral`` is absent from this image -> a stand-in module object is put in ``sys.modules``
    whose ``envi.open`` / ``envi.create_image`` hand numpy arrays to the script in the layouts
    the real library would (input (lines, bands, samples); outputs (lines, samples, bands)).
  * ``cmf/robust_mf.py`` is then executed UNMODIFIED with ``runpy`` as ``__main__`` with the
    names it forgets to import (os, sys, np) and a list-returning ``map`` injected; the broken
    column-stats DataFrame at :401 raises ValueError after every pixel output is complete --
    the per-column stats are recovered from the script's frame.
  * ``looshrinkage`` is imported from the same file and called directly for the function-level
    cases.
Nothing of the reference's text is stored: only inputs (as seeds / small arrays) and outputs.
"""
import builtins
import os
import runpy
import sys
import types

os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
LIB_TXT = os.path.join(REF, "cmf", "ang_ch4_unit_3col_425chan.txt")

from srcfinder_amd.synth import make_cube_numpy, synth_columns  # noqa: E402

arrays = {}
store = {}


class _Img:
    def __init__(self, arr, meta):
        self.arr, self.metadata = arr, meta

    def open_memmap(self, **kw):
        return self.arr


def _open(hdr, image=None):
    return _Img(arrays[image], dict(store[image]))


def _create(hdr, meta, force=False, ext=""):
    path = hdr[:-4]
    dt = {4: np.float32, 5: np.float64, 2: np.int16}[int(meta["data type"])]
    arrays[path] = np.zeros((int(meta["lines"]), int(meta["samples"]), int(meta["bands"])), dt)
    store[path] = dict(meta)
    return _Img(arrays[path], dict(meta))


def install_spectral_stub():
    envi = types.ModuleType("spectral.io.envi")
    envi.open, envi.create_image = _open, _create
    envi.dtype_to_envi = {"f": 4, "d": 5, "h": 2}
    sp, spio = types.ModuleType("spectral"), types.ModuleType("spectral.io")
    sp.io, spio.envi = spio, envi
    sys.modules.update({"spectral": sp, "spectral.io": spio, "spectral.io.envi": envi})


def run_reference_main(cube, extra_args=(), libpath=LIB_TXT, tag="in", metadata=True):
    """Execute the reference script on `cube` ([lines,bands,samples] f32). Returns dict of outputs."""
    lines, bands, samples = cube.shape
    IN, OUT = "/virtual/%s" % tag, "/virtual/%s_out" % tag
    arrays[IN] = cube
    store[IN] = {"lines": lines, "samples": samples, "bands": bands, "interleave": "bil", "data type": 4,
                 "data ignore value": "-9999", "wavelength": ["0"] * bands}
    old_argv = sys.argv
    sys.argv = ["robust_mf.py", *(["-m"] if metadata else []), *extra_args, IN, libpath, OUT]
    _map = builtins.map
    frame_globals = {}
    try:
        runpy.run_path(os.path.join(REF, "cmf", "robust_mf.py"), run_name="__main__",
                       init_globals={"os": os, "sys": sys, "np": np,
                                     "map": lambda f, a: list(_map(f, a))})
        raise RuntimeError("reference unexpectedly ran to completion")
    except ValueError as e:          # the column-stats DataFrame; outputs are complete
        tb = e.__traceback__
        while tb is not None:
            g = tb.tb_frame.f_globals
            if "colavg" in g and "colnum" in g:
                frame_globals = g
            tb = tb.tb_next
    finally:
        sys.argv = old_argv
    assert frame_globals, "could not recover the column statistics from the reference frame"
    out = arrays[OUT]
    bg = arrays[OUT + "_bgmeta"] if metadata else np.zeros((lines, samples, 2), np.int16)
    return dict(out=out.copy(), bgmeta=bg.copy(),
                colstats=np.stack([frame_globals["colnum"], frame_globals["colavg"], frame_globals["colstd"]]),
                modelparms=str(store[OUT]["model parameters"]))


def versions():
    return np.array("numpy %s scipy %s" % (np.__version__, scipy.__version__))


def main():
    install_spectral_stub()
    lib = np.float64(np.loadtxt(LIB_TXT))
    np.savez_compressed(os.path.join(HERE, "ch4_library.npz"), library=lib, source=np.array(
        "cmf/ang_ch4_unit_3col_425chan.txt (channel, wavelength nm, unit CH4 absorption)"))

    # ---- (1) S config: 64 samples x 512 lines x 425 bands, radiance mode, -m --------------------------
    cube = make_cube_numpy(512, 64, seed=1234, abscf_full=lib[:, 2])
    r = run_reference_main(cube, tag="S")
    aidx = np.array([np.unique(r["bgmeta"][:, c, 1]) for c in range(64)], dtype=object)
    np.savez_compressed(os.path.join(HERE, "cmf_S_radiance.npz"),
                        seed=1234, lines=512, samples=64, out=r["out"], bgmeta=r["bgmeta"],
                        colstats=r["colstats"], modelparms=np.array(r["modelparms"]), versions=versions())
    print("S radiance: alpha idx per column:", sorted(set(int(v) for a in aidx for v in a)))

    # ---- (2) reflectance mode (-R): active 5..420 (p = 416); small cube ------------------------------
    cube_r = make_cube_numpy(600, 6, seed=4321, abscf_full=lib[:, 2], active=(5, 420), nodata_column=4)
    r = run_reference_main(cube_r, extra_args=("-R",), tag="R")
    np.savez_compressed(os.path.join(HERE, "cmf_R_reflectance.npz"),
                        seed=4321, lines=600, samples=6, nodata_column=4, out=r["out"], bgmeta=r["bgmeta"],
                        colstats=r["colstats"], modelparms=np.array(r["modelparms"]), versions=versions())
    print("R reflectance: alpha idx:", np.unique(r["bgmeta"][:, :, 1]))

    # ---- (3) direct looshrinkage cases ---------------------------------------------------------------
    sys.path.insert(0, os.path.join(REF, "cmf"))
    import robust_mf as R
    alphas = 10.0 ** np.arange(-10.0, 0.0 + 0.05, 0.05)
    cases = {}

    synth_cols = synth_columns

    specs = [("n100_p8", 100, 8, 11), ("n512_p72", 512, 72, 12), ("n2000_p72", 2000, 72, 13),
             ("n2000_p425", 2000, 425, 14),       # det overflow excludes the large alphas
             ("n300_p425", 300, 425, 15),         # n < p: underflow AND overflow exclusions
             ("n100_p8_big", 100, 8, 16)]
    for name, n, p, seed in specs:
        x = synth_cols(n, p, seed, scale=(1e3 if name.endswith("big") else 1.0))
        izm = x - x.mean(axis=0)
        nll = np.zeros(len(alphas))
        C, mindex = R.looshrinkage(izm, alphas, nll, n)
        cases[name + "_spec"] = np.array([n, p, seed, 1e3 if name.endswith("big") else 1.0])
        if p <= 72:
            cases[name + "_C"] = C
        else:                      # keep the fixture small: diagonal + a strided sub-block of C
            cases[name + "_Cdiag"] = np.diag(C).copy()
            cases[name + "_Csub"] = C[::17, ::13].copy()
        cases[name + "_mindex"] = mindex
        cases[name + "_nll"] = nll.copy()
        print(name, "mindex", mindex, "ninf", int(np.isinf(nll).sum()), "nnan", int(np.isnan(nll).sum()))
    # constant band -> exactly singular S, det == 0 for every alpha -> all NLL inf -> mindex -1
    x = synth_cols(200, 8, 17)
    x[:, 3] = np.float64(np.float32(1.25))
    izm = x - x.mean(axis=0)
    nll = np.zeros(len(alphas))
    C, mindex = R.looshrinkage(izm, alphas, nll, 200)
    cases["const_band_spec"] = np.array([200, 8, 17, 1.0])
    cases["const_band_C"] = C
    cases["const_band_mindex"] = mindex
    cases["const_band_nll"] = nll.copy()
    print("const_band mindex", mindex, "ninf", int(np.isinf(nll).sum()))
    try:
        R.inv(C)
        cases["const_band_inv_raises"] = False
    except Exception as e:      # scipy LinAlgError
        cases["const_band_inv_raises"] = True
        print("const_band inv raises:", type(e).__name__)
    # cov/inv/det/eig wrapper defaults
    a = synth_cols(50, 6, 18)
    cases["wrap_spec"] = np.array([50, 6, 18, 1.0])
    cases["wrap_cov"] = R.cov(a)
    cases["wrap_inv"] = R.inv(R.cov(a))
    cases["wrap_det"] = R.det(R.cov(a))
    ev, evec = R.eig(R.cov(a))
    cases["wrap_eigvals"] = ev
    cases["wrap_eigvecs"] = evec
    np.savez_compressed(os.path.join(HERE, "cmf_looshrinkage_cases.npz"), alphas=alphas, versions=versions(), **cases)

    # ---- (4) constant band inside a full column loop: singular C -> mode pixels := 0 -----------------
    cube_s = make_cube_numpy(300, 4, seed=99, abscf_full=lib[:, 2], nodata_column=-1)
    cube_s[:, 351 - 1 + 20, 1] = np.float32(0.75)
    r = run_reference_main(cube_s, tag="SING")
    np.savez_compressed(os.path.join(HERE, "cmf_singular_column.npz"),
                        seed=99, lines=300, samples=4, const_band=351 - 1 + 20, const_col=1,
                        const_value=np.float32(0.75), out=r["out"], bgmeta=r["bgmeta"],
                        colstats=r["colstats"], versions=versions())
    print("singular column: out[:,1,3] unique:", np.unique(r["out"][:, 1, 3]))


if __name__ == "__main__":
    main()
