#!/usr/bin/env python3
"""Golden vectors for the triage column profile, produced by EXECUTING the reference.

Only runs in the development container (needs /root/reference).  ``triage/cmf_profile.py`` is run unmodified as
``__main__`` (its ``summarize()`` is a closure of the script body); ``srcfinder_util`` is the real module, imported
with stand-ins for the third-party packages this image lacks (gdal, rasterio, spectral, geopandas, skimage,
LatLongUTMconversion -- none of them is touched by the column statistics), and its ``openimgmm`` is replaced by a
function that hands the script an in-memory product (the file reader is not on the path).  Everything numerical --
the float32 casts, the validity rule, ``np.nanmean/nanstd/nanmin/nanmax``, ``np.nanmedian`` and
``srcfinder_util.extrema`` (``np.nanpercentile(..., interpolation='nearest')``) -- is the reference's own code.

Stored: the seed and shape of the synthetic product (tests regenerate it with ``product()`` below) and the two CSV
tables the script writes (plain and ``--robust``).

    python tests/golden/gen_golden_triage.py
"""
import os
import runpy
import sys
import tempfile
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
LINES, SAMPLES, SEED = 700, 40, 4242
BIG_LINES, BIG_SAMPLES, BIG_SEED = 40000, 24, 777      # longer than the 32768 lines an LDS-resident sort holds


def product(lines=LINES, samples=SAMPLES, seed=SEED):
    """A CMF product [lines, samples, 4] float64 (R, G, B, CMF) with everything the rule looks at: NODATA pixels, NaN,
    negative and zero scores, an all-NODATA column, a column with a single positive value, heavy ties (quantised values)
    so that the 'nearest' percentile index arithmetic matters."""
    rng = np.random.default_rng(seed)
    img = np.empty((lines, samples, 4))
    img[..., :3] = rng.uniform(0, 20, (lines, samples, 3))
    cmf = rng.normal(150.0, 400.0, (lines, samples))
    cmf[:, ::3] = np.round(cmf[:, ::3] / 25.0) * 25.0            # ties
    cmf[rng.random((lines, samples)) < 0.05] = -9999.0
    cmf[rng.random((lines, samples)) < 0.01] = np.nan
    cmf[rng.random((lines, samples)) < 0.01] = 0.0
    cmf[:9] = -9999.0
    cmf[:, 7] = -9999.0                                          # no valid pixel at all
    cmf[:, 11] = -5.0
    cmf[100, 11] = 42.5                                          # exactly one positive pixel
    cmf[:, 13] = -1.0                                            # valid but never positive
    ragged = rng.integers(1, lines, samples)                     # different counts per column (odd and even)
    for c in range(20, samples):
        cmf[ragged[c]:, c] = -9999.0
    img[..., 3] = cmf
    return img


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def run_reference(img, robust):
    for name in ("gdal", "rasterio", "geopandas", "spectral", "spectral.io", "skimage", "dask", "dask.distributed"):
        _stub(name)
    _stub("osgeo", gdal=sys.modules["gdal"])
    _stub("osgeo.gdal", gdalconst=None, ogr=None, osr=None)
    sys.modules["gdal"].gdalconst = sys.modules["gdal"].ogr = sys.modules["gdal"].osr = None
    _stub("spectral.io.envi", open=lambda *a, **k: None)
    _stub("LatLongUTMconversion", UTMtoLL=None, LLtoUTM=None)
    _stub("skimage.measure", label=None)
    sys.modules["skimage"].__path__ = []                         # a package: srcfinder_util imports its submodules lazily
    _stub("skimage.morphology", disk=lambda r, **k: np.ones((2 * r + 1, 2 * r + 1), bool))   # default arguments only
    sys.path.insert(0, REF)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import srcfinder_util as U                               # the real module
    meta = {"band names": ["Red (x)", "Green (x)", "Blue (x)", "CH4 (ppmm)"], "data ignore value": "-9999"}
    U.openimgmm = lambda f, **k: (types.SimpleNamespace(metadata=dict(meta)), img)
    out = tempfile.mkdtemp()
    name = "ang20200101t000000_cmf_v1_img"
    argv = ["cmf_profile.py", "--outdir", out] + (["--robust"] if robust else []) + [name]
    old = sys.argv
    sys.argv = argv
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                      # all-NaN columns: "Mean of empty slice" etc.
            runpy.run_path(os.path.join(REF, "triage", "cmf_profile.py"), run_name="__main__")
    except SystemExit:
        pass
    finally:
        sys.argv = old
    import pandas as pd
    df = pd.read_csv(os.path.join(out, name + "_column_stats.csv"))
    return list(df.columns), df.to_numpy(dtype=np.float64)


def main():
    img = product()
    cols_p, plain = run_reference(img, robust=False)
    cols_r, rob = run_reference(img, robust=True)
    assert cols_p == ["npix", "avg", "std", "min", "max"] and cols_r == ["npix", "med", "mad", "p05", "p95"]
    import scipy
    np.savez_compressed(os.path.join(HERE, "triage_profile.npz"), lines=LINES, samples=SAMPLES, seed=SEED,
                        plain=plain, robust=rob, columns_plain=np.array(cols_p), columns_robust=np.array(cols_r),
                        versions=np.array(["numpy " + np.__version__, "scipy " + scipy.__version__]))
    print("plain\n", plain[:14], "\nrobust\n", rob[:14])
    # a product with more lines than the GPU's LDS sort holds: the reference has no cap (cmf_profile.py:124-127)
    big = product(BIG_LINES, BIG_SAMPLES, BIG_SEED)
    _c, bplain = run_reference(big, robust=False)
    _c, brob = run_reference(big, robust=True)
    np.savez_compressed(os.path.join(HERE, "triage_profile_big.npz"), lines=BIG_LINES, samples=BIG_SAMPLES, seed=BIG_SEED,
                        plain=bplain, robust=brob, versions=np.array(["numpy " + np.__version__, "scipy " + scipy.__version__]))
    print("big robust\n", brob)


if __name__ == "__main__":
    main()
