#!/usr/bin/env python3
"""Golden vectors for saliency map -> detections, produced by EXECUTING the reference.

Only runs in the development container (needs /root/reference).  ``salience_predictions.py`` is run unmodified as
``__main__`` (its ``salience2detections`` reads a global of the script body) with the real ``srcfinder_util``
(``extrema``, ``findobj``, ``sl2latlon`` / ``sl2xy`` / ``rotxy`` / ``utm2latlon``, ``mapinfo``; the map info carries the
rotation of the reference's own sample product, ``rotation=17``; the UTM -> lat/lon conversion itself is a third-party
module, ``LatLongUTMconversion``, that is absent: it is served by ``oracle/utm_oracle.py``, a statement of that module's
published series pinned by ``tests/test_geo_cpu.py``); stand-ins: the file reader (``openimgmm`` hands over
in-memory arrays), and the third-party functions this image lacks, by their published definitions --
``skimage.measure.label`` (connectivity 2 -> scipy.ndimage.label with the 3 x 3 structure, same raster numbering) and
``statsmodels.robust.scale.mad`` (``median(|a - center|) / c``).  The table the script builds (one row per region) is
captured where it is assembled (``DataFrame.from_records``); the figures it draws per region go to a scratch directory.

    python tests/golden/gen_golden_detections.py
"""
import os
import runpy
import sys
import tempfile
import types
import warnings

import numpy as np
import scipy.ndimage as ndi

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
LINES, SAMPLES, SEED = 180, 90, 77
# ulx / uly / pixel size / zone / rotation of /root/reference/cnn/samples/ang20200924t211102_ch4mf_v2y1_img.hdr
MAPINFO = ["UTM", "1", "1", "272247.152557", "3992010.65018", "3.1", "3.1", "11", "North", "WGS-84", "units=Meters",
           "rotation=17.0000000"]


def scene(lines=LINES, samples=SAMPLES, seed=SEED):
    """(saliency [lines, samples, 1] float32, product [lines, samples, 4] float64): a dozen blobs of saliency above 0.5
    (touching diagonally, plateaus with several maxima, one on the NODATA border), CMF enhancements under most of them."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:lines, 0:samples]
    sal = rng.uniform(0.0, 0.3, (lines, samples))
    cmf = rng.normal(40.0, 120.0, (lines, samples))
    for k in range(12):
        cy, cx = rng.integers(6, lines - 6), rng.integers(6, samples - 6)
        ry, rx = rng.uniform(1.5, 7), rng.uniform(1.5, 6)
        blob = np.exp(-(((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2))
        sal = np.maximum(sal, 0.98 * blob)
        cmf += (300 + 900 * rng.random()) * blob * (rng.random((lines, samples)) * 0.5 + 0.75)
    sal[40:44, 20:24] = 0.875                       # a plateau: four-by-four pixels share the maximum
    sal[44, 24] = sal[45, 25] = 0.7                 # joined to the plateau only diagonally
    cmf[40:44, 20:24] = 777.0
    sal = np.round(sal.astype(np.float32), 3)       # ties
    img = np.empty((lines, samples, 4))
    img[..., :3] = rng.uniform(0.5, 12, (lines, samples, 3))
    img[..., 3] = cmf
    img[:5, :, :3] = -9999.0                        # NODATA border (RGB band 0 == -9999)
    img[:5, :, 3] = -9999.0
    sal[3:8, 50:56] = 0.9                           # a region that straddles the border
    cmf2 = img[..., 3]
    cmf2[5:8, 50:56] = 650.0
    return sal[..., None], img


def scene_big(lines=420, samples=330, seed=99):
    """The scene() of that size plus ONE smooth giant plume: ~38 000 saliency pixels above 0.5 and ~30 000 CMF pixels above
    250 ppm m in a single region -- more than the GPU's LDS-resident sort takes (32768 / 16384).  The reference has no cap
    (salience_predictions.py:81-103).  VERDICT r4 item 8."""
    sal, img = scene(lines, samples, seed)
    rng = np.random.default_rng(seed + 1)
    yy, xx = np.mgrid[0:lines, 0:samples]
    blob = np.exp(-(((yy - 230) / 150.0) ** 2 + ((xx - 170) / 120.0) ** 2))
    s2 = np.maximum(sal[..., 0], np.round((0.98 * blob).astype(np.float32), 3))
    img[5:, :, 3] += 900.0 * blob[5:] * (rng.random(blob[5:].shape) * 0.6 + 0.7)
    return np.float32(s2)[..., None], img


def _stub(name, **a):
    m = types.ModuleType(name)
    m.__dict__.update(a)
    sys.modules[name] = m
    return m


def run_reference(sal, img, prob_thr=0.5, ppmm_thr=250.0):
    for name in ("gdal", "rasterio", "geopandas", "spectral", "spectral.io", "skimage", "statsmodels", "statsmodels.robust"):
        _stub(name)
    _stub("osgeo", gdal=sys.modules["gdal"])
    sys.modules["gdal"].gdalconst = sys.modules["gdal"].ogr = sys.modules["gdal"].osr = None
    _stub("spectral.io.envi", open=lambda *a, **k: None)
    sys.modules["spectral"].SpyFile = type("SpyFile", (), {})
    # UTM -> lat/lon is a third-party module (absent): oracle/utm_oracle.py states its series; the reference indexes
    # the two results (``p[0]``, :109-110), so they are handed over as arrays
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import utm_oracle
    _stub("LatLongUTMconversion",
          UTMtoLL=lambda datum, n, e, zone: tuple(np.atleast_1d(v) for v in utm_oracle.UTMtoLL(datum, n, e, zone)),
          LLtoUTM=utm_oracle.LLtoUTM)
    sys.modules["skimage"].__path__ = []
    _stub("skimage.morphology", disk=lambda r, **k: np.ones((2 * r + 1, 2 * r + 1), bool))
    _stub("skimage.measure", label=lambda a, connectivity=None, **k: ndi.label(a, structure=ndi.generate_binary_structure(a.ndim, connectivity or a.ndim))[0])
    _stub("statsmodels.robust.scale", mad=lambda a, c=0.6745, axis=0, center=np.median: np.median(np.abs(a - (center(a) if callable(center) else center)), axis=axis) / c)
    sys.path.insert(0, REF)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import srcfinder_util as U
    SpyFile = sys.modules["spectral"].SpyFile

    class Img(SpyFile):
        metadata = {"map info": list(MAPINFO)}
    arrays = {"sal_img": sal, "ang20200101t000000_cmf_v1_img": img}
    U.openimgmm = lambda f, **k: (Img(), arrays[os.path.basename(f)])
    captured = []
    import pandas as pd
    orig = pd.DataFrame.from_records

    def spy(*a, **k):
        df = orig(*a, **k)
        captured.append(df)
        return df
    pd.DataFrame.from_records = staticmethod(spy)
    out = tempfile.mkdtemp()
    old = sys.argv
    sys.argv = ["salience_predictions.py", "--prob_thr", str(prob_thr), "--ppmm_thr", str(ppmm_thr), "--outdir", out,
                "sal_img", "ang20200101t000000_cmf_v1_img"]
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            runpy.run_path(os.path.join(REF, "salience_predictions.py"), run_name="__main__")
    except (SystemExit, AttributeError, ImportError, ModuleNotFoundError) as e:   # the .xlsx writer (pandas >= 2) after the table is built
        print("script stopped after the table:", type(e).__name__, e)
    finally:
        sys.argv = old
        pd.DataFrame.from_records = orig
    return captured[0]


def main():
    sal, img = scene()
    df = run_reference(sal, img)
    cols = [c for c in df.columns if c not in ("detid", "lid")]
    table = df[cols].to_numpy(dtype=np.float64)
    import scipy
    np.savez_compressed(os.path.join(HERE, "detections_golden.npz"), lines=LINES, samples=SAMPLES, seed=SEED,
                        prob_thr=0.5, ppmm_thr=250.0, columns=np.array(cols), table=table, detid=np.array(df["detid"], dtype=str),
                        mapinfo=np.array(MAPINFO), versions=np.array(["numpy " + np.__version__, "scipy " + scipy.__version__]))
    print(df.to_string())
    # a region larger than the GPU's LDS-resident sort (round 5)
    sal, img = scene_big()
    dfb = run_reference(sal, img)
    colsb = [c for c in dfb.columns if c not in ("detid", "lid")]
    np.savez_compressed(os.path.join(HERE, "detections_big_golden.npz"), lines=420, samples=330, seed=99, prob_thr=0.5,
                        ppmm_thr=250.0, columns=np.array(colsb), table=dfb[colsb].to_numpy(dtype=np.float64),
                        detid=np.array(dfb["detid"], dtype=str), mapinfo=np.array(MAPINFO),
                        versions=np.array(["numpy " + np.__version__, "scipy " + scipy.__version__]))
    print(dfb[["minr", "maxr", "minc", "maxc", "salnpix", "cmfnpix"]].to_string() if "salnpix" in dfb.columns else dfb.iloc[:, :12].to_string())


if __name__ == "__main__":
    main()
