#!/usr/bin/env python3
"""Golden vectors for the spectrometer masks, produced by EXECUTING the reference script.

Only runs in the development container (needs /root/reference).  ``spectrometer_masks/masks_sds.py`` is a top-level
script: it is run unmodified with ``runpy`` on a seeded synthetic radiance cube.  Stand-ins: ``spectral`` (the file
layer: ``envi.open`` hands the script an in-memory BIP cube, ``envi.save_image`` captures the product) and ``skimage``
(absent here; ``morphology.disk / binary_dilation`` and ``measure.label / regionprops`` are replaced by scipy.ndimage
equivalents of their published definitions -- see oracle/masks_oracle.py).  The per-pixel rules are the script's own code.

Stored per case: the cube's seed and geometry, the command-line flags, the product int16 [lines, samples, 4] and the
pre-dilation layers the script leaves in its globals.

    python tests/golden/gen_golden_masks.py
"""
import os
import runpy
import sys
import tempfile
import types

import numpy as np
import scipy.ndimage as ndi

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/spectrometer_masks/masks_sds.py"
BANDS = 425
WAVELENGTHS = np.round(np.linspace(376.86, 2500.12, BANDS), 2)       # AVIRIS-NG-like centres, 5 nm apart

CASES = {   # name: (lines, samples, seed, flags)
    "default_px": (120, 48, 11, ["-M", "6px", "-B", "3px"]),
    "minarea": (140, 40, 20, ["-M", "5px", "-B", "2px", "-A", "6"]),
    "minarea_none_qualifies": (140, 40, 12, ["-M", "5px", "-B", "2px", "-A", "40"]),
    "two_blocks": (260, 36, 13, ["-M", "7px", "-B", "4px", "--saturation-processing-block-length", "100"]),
    "meters": (90, 40, 14, ["-M", "12m", "-B", "9m"]),
}


def radiance_cube(lines, samples, seed):
    """BIP float32 [lines, samples, 425] with every rule triggered: bright falling spectra (cloud), bright rising ones
    (not cloud), SWIR flares of several sizes (some over bright 500 nm = sun glint, some specular), dark water, a
    NODATA border and a NaN."""
    rng = np.random.default_rng(seed)
    base = 8.0 * np.exp(-np.arange(BANDS) / 140.0) + 0.3
    cube = (base * rng.uniform(0.5, 1.2, (lines, samples, 1)) * rng.uniform(0.97, 1.03, (lines, samples, BANDS))).astype(np.float32)
    yy, xx = np.mgrid[0:lines, 0:samples]
    cloud = ((yy - lines * 0.3) ** 2 / 90 + (xx - samples * 0.6) ** 2 / 60) < 1
    cube[cloud] *= np.float32(3.2)                                    # band 15 > 15 with a falling slope to band 60
    rising = ((yy - lines * 0.7) ** 2 + (xx - samples * 0.2) ** 2) < 16
    cube[rising, :40] = 16.0 + np.arange(40, dtype=np.float32) * 0.2  # bright but rising 450 -> 670 nm: not a cloud
    for k in range(7):                                                # flares: SWIR saturation blobs of 1..12 pixels
        cy, cx = rng.integers(8, lines - 8), rng.integers(4, samples - 4)
        h, w = rng.integers(1, 4), rng.integers(1, 5)
        cube[cy:cy + h, cx:cx + w, 330:400] = np.float32(6.5 + k)
        if k % 3 == 0:
            cube[cy:cy + h, cx:cx + w, 20:30] = 11.0                  # bright at 500 nm: glint / specular
    cube[lines // 2:lines // 2 + 6, :7, 340:360] = 0.05               # dark surface at 2139 nm
    cube[:4] = -9999.0
    cube[-3:, :, :] = -9999.0
    cube[lines // 3, samples // 2, 100] = np.nan
    return cube


def _stub(name, **a):
    m = types.ModuleType(name)
    m.__dict__.update(a)
    sys.modules[name] = m
    return m


class _MM(np.ndarray):
    def flush(self):
        pass


def run_reference(cube, flags, pixel_m=3.0):
    saved = {}
    lines, samples, _ = cube.shape

    class Img:
        def __init__(s):
            s.metadata = {"map info": ["UTM", "1", "1", "0", "0", str(pixel_m), str(pixel_m), "11", "North", "WGS-84", "units=Meters"]}
            s.nrows, s.ncols = lines, samples
            s.bands = types.SimpleNamespace(centers=list(WAVELENGTHS))

        def open_memmap(s, **k):
            return cube.view(_MM)

        def read_subregion(s, rows, cols):
            return cube[rows[0]:rows[1], cols[0]:cols[1]]

    envi = _stub("spectral.io.envi", open=lambda p: Img(), dtype_to_envi={"h": 2},
                 save_image=lambda path, arr, **k: saved.update(arr=np.array(arr), meta=k.get("metadata"), path=path))
    spio = _stub("spectral.io", envi=envi)
    _stub("spectral", io=spio, envi=envi)

    def label(a, connectivity=None, return_num=False):
        lab, n = ndi.label(a, structure=ndi.generate_binary_structure(a.ndim, connectivity or a.ndim))
        return (lab, n) if return_num else lab

    class RP:
        def __init__(s, coords):
            s.coords, s.area = coords, len(coords)

    def disk(radius, dtype=np.uint8):
        r = int(radius)
        y, x = np.ogrid[-r:r + 1, -r:r + 1]
        return (x * x + y * y <= r * r).astype(dtype)

    def bdil(image, selem=None, **k):
        return ndi.binary_dilation(image, structure=ndi.generate_binary_structure(image.ndim, 1) if selem is None else selem)

    morph = _stub("skimage.morphology", disk=disk, binary_dilation=bdil)
    meas = _stub("skimage.measure", label=label, regionprops=lambda lab: [RP(np.argwhere(lab == i)) for i in range(1, lab.max() + 1)])
    _stub("skimage", morphology=morph, measure=meas)
    d = tempfile.mkdtemp()
    with open(os.path.join(d, "list.txt"), "w") as f:
        f.write("ang20200101t000000_rdn_v2x1_img\n")
    old = sys.argv
    sys.argv = ["masks_sds.py", "--txt", os.path.join(d, "list.txt"), "--inpath", d + "/", "--outpath", d + "/"] + flags
    try:
        g = runpy.run_path(REF, run_name="__main__")
    finally:
        sys.argv = old
    return saved["arr"], {k: np.array(g[k]) for k in ("sat_mask_full2", "spec_mask_full", "dark_mask_full")}


def main():
    out = {"wavelengths": WAVELENGTHS, "cases": np.array(list(CASES))}
    for name, (lines, samples, seed, flags) in CASES.items():
        prod, layers = run_reference(radiance_cube(lines, samples, seed), flags)
        out[name + "_geom"] = np.array([lines, samples, seed])
        out[name + "_flags"] = np.array(flags)
        out[name + "_product"] = prod
        out[name + "_cloud_raw"] = layers["sat_mask_full2"]
        print(name, prod.shape, [int((prod[..., k] > 0).sum()) for k in range(4)], "flare==2:", int((prod[..., 2] == 2).sum()),
              "border:", int((prod[..., 0] == -9999).sum()))
    import scipy
    out["versions"] = np.array(["numpy " + np.__version__, "scipy " + scipy.__version__])
    np.savez_compressed(os.path.join(HERE, "masks_golden.npz"), **out)


if __name__ == "__main__":
    main()
