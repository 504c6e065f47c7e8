"""Spectrometer masks and the shared image primitives: the numpy restatement against the products the
real ``spectrometer_masks/masks_sds.py`` wrote (CPU), and the HIP path against both (``-m gpu``)."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import masks_oracle as MO


def _gen(golden_dir):
    spec = importlib.util.spec_from_file_location("gen_golden_masks", os.path.join(golden_dir, "gen_golden_masks.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    return gen


def _flags(flags):
    """The script's flags -> keyword arguments of srcfinder_amd.masks.spectrometer_masks (same names as the flags)."""
    kw, f, i = {}, list(flags), 0
    names = {"-M": "maskgrowradius", "-B": "cldbfr", "-A": "mingrowarea",
             "--saturation-processing-block-length": "saturation_processing_block_length"}
    while i < len(f):
        k = names[f[i]]
        kw[k] = int(f[i + 1]) if k in ("mingrowarea", "saturation_processing_block_length") else f[i + 1]
        i += 2
    return kw


META = {"map info": ["UTM", "1", "1", "0", "0", "3.0", "3.0", "11", "North", "WGS-84", "units=Meters"]}


def _oracle_kw(kw):
    from srcfinder_amd import masks
    return dict(cloud_buffer_px=masks.cloud_buffer_passes(kw.get("cldbfr", "150m"), META),
                grow_radius_px=masks.radius_in_pixels(kw.get("maskgrowradius", "150m"), META),
                mingrowarea=kw.get("mingrowarea"), block=kw.get("saturation_processing_block_length", 500))


def test_masks_oracle_reproduces_the_reference_products(golden_dir):
    gen = _gen(golden_dir)
    g = np.load(os.path.join(golden_dir, "masks_golden.npz"))
    for name in g["cases"]:
        lines, samples, seed = (int(v) for v in g[name + "_geom"])
        cube = gen.radiance_cube(lines, samples, seed)
        got = MO.masks(cube, g["wavelengths"], **_oracle_kw(_flags(g[name + "_flags"])))
        assert np.array_equal(got, g[name + "_product"]), name
        assert np.array_equal(MO.pixel_rules(cube, g["wavelengths"])["cloud"], g[name + "_cloud_raw"] != 0), name
    assert (g["minarea_none_qualifies_product"][..., 2] > 0).sum() == 0      # the script's quirk is in the golden
    assert (g["minarea_product"][..., 2] == 2).sum() > 0 and (g["two_blocks_product"][..., 2] == 2).sum() > 0


def test_radius_parsing_follows_the_script():
    from srcfinder_amd import masks
    assert masks.radius_in_pixels("12px") == 12 and masks.radius_in_pixels("2.2px") == 3
    assert masks.radius_in_pixels("150m", META) == 50 and masks.radius_in_pixels("10m", META) == 4
    assert masks.cloud_buffer_passes("9m", META) == 3 and masks.cloud_buffer_passes("10m", META) == 4
    with pytest.raises(RuntimeError):
        masks.radius_in_pixels("150m", {})
    with pytest.raises(RuntimeError):
        masks.radius_in_pixels("150", META)


@pytest.mark.gpu
def test_masks_gpu_match_reference_products(golden_dir):
    import torch
    from srcfinder_amd import masks
    gen = _gen(golden_dir)
    g = np.load(os.path.join(golden_dir, "masks_golden.npz"))
    for name in g["cases"]:
        lines, samples, seed = (int(v) for v in g[name + "_geom"])
        bip = gen.radiance_cube(lines, samples, seed)
        bil = torch.as_tensor(np.ascontiguousarray(bip.transpose(0, 2, 1))).cuda()
        got = masks.spectrometer_masks(bil, g["wavelengths"], metadata=META, to_numpy=True, **_flags(g[name + "_flags"]))
        assert got.dtype == np.int16 and np.array_equal(got, g[name + "_product"]), name


@pytest.mark.gpu
def test_masks_gpu_random_cubes_against_oracle(golden_dir):
    """Ragged geometries, several blocks, radii larger than the image, NaN / inf values, with and without the buffers."""
    import torch
    from srcfinder_amd import masks
    gen = _gen(golden_dir)
    rng = np.random.default_rng(3)
    wl = gen.WAVELENGTHS
    for lines, samples, kw in ((97, 70, dict(maskgrowradius="9px", cldbfr="5px")),
                               (333, 65, dict(maskgrowradius="3px", cldbfr="1px", mingrowarea=3, saturation_processing_block_length=64)),
                               (64, 130, dict(maskgrowradius=None, cldbfr="0px")),
                               (50, 9, dict(maskgrowradius="80px", cldbfr="70px")),
                               (700, 33, dict(maskgrowradius="21m", cldbfr="16m", mingrowarea=5))):
        bip = gen.radiance_cube(lines, samples, int(rng.integers(1 << 30)))
        bip[rng.integers(lines), rng.integers(samples), 352] = np.inf
        bip[rng.integers(lines), rng.integers(samples), 15] = np.nan
        bil = torch.as_tensor(np.ascontiguousarray(bip.transpose(0, 2, 1))).cuda()
        got = masks.spectrometer_masks(bil, wl, metadata=META, to_numpy=True, **kw)
        okw = dict(cloud_buffer_px=masks.cloud_buffer_passes(kw["cldbfr"], META),
                   grow_radius_px=None if kw["maskgrowradius"] is None else masks.radius_in_pixels(kw["maskgrowradius"], META),
                   mingrowarea=kw.get("mingrowarea"), block=kw.get("saturation_processing_block_length", 500))
        want = MO.masks(bip, wl, **okw)
        assert np.array_equal(got, want), (lines, samples, kw)


@pytest.mark.gpu
def test_label8_matches_scipy_numbering():
    """Component ids in raster order of the first pixels -- scipy.ndimage.label with the 3 x 3 structure, the
    definition skimage.measure.label(connectivity=2) shares -- on random, striped, spiral and empty images."""
    import scipy.ndimage as ndi
    from srcfinder_amd import masks
    rng = np.random.default_rng(8)
    imgs = [rng.random((120, 77)) < d for d in (0.05, 0.3, 0.45, 0.6, 0.9)]
    imgs += [np.zeros((40, 33), bool), np.ones((31, 65), bool), np.indices((64, 64)).sum(0) % 2 == 0]
    spiral = np.zeros((97, 97), bool)
    for k in range(0, 48, 2):
        spiral[k, k:97 - k] = spiral[96 - k, k:97 - k] = spiral[k:97 - k, 96 - k] = True
        spiral[k + 2:97 - k, k] = True
        spiral[k + 2, k:k + 3] = True
    imgs.append(spiral)
    imgs.append(rng.random((2500, 598)) < 0.4)                 # a few hundred thousand components, long equivalence chains
    for im in imgs:
        lab, n = masks.label(im, to_numpy=True)
        want, nw = ndi.label(im, structure=np.ones((3, 3)))
        assert n == nw and np.array_equal(lab, want)


# ------------------------------------------------------------------------------------------------------------------
# saliency -> detections (first half)
# ------------------------------------------------------------------------------------------------------------------
def _gen_det(golden_dir):
    spec = importlib.util.spec_from_file_location("gen_golden_detections", os.path.join(golden_dir, "gen_golden_detections.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    return gen


_DET_GOLDENS = [("detections_golden.npz", "scene", 14), ("detections_big_golden.npz", "scene_big", 9)]   # big: one region of
#                     ~38 000 saliency / ~30 000 CMF pixels, more than the GPU's LDS-resident sort holds


@pytest.mark.parametrize("fname,scene,nrows", _DET_GOLDENS)
def test_detections_oracle_reproduces_the_reference_table(golden_dir, fname, scene, nrows):
    from oracle import detect_oracle as DO
    gen = _gen_det(golden_dir)
    g = np.load(os.path.join(golden_dir, fname))
    sal, img = getattr(gen, scene)(int(g["lines"]), int(g["samples"]), int(g["seed"]))
    mi = [float(v) for v in g["mapinfo"][3:7]]
    assert "rotation=17.0000000" in list(g["mapinfo"])          # the rotation of the reference's own sample product
    got = DO.detections(sal, img, float(g["prob_thr"]), float(g["ppmm_thr"]), *mi, rot=17.0, zone="11", hemi="North")
    assert list(g["columns"]) == DO.HEADER
    assert got.shape == g["table"].shape and len(got) == nrows
    assert np.array_equal(got, g["table"])


@pytest.mark.gpu
@pytest.mark.parametrize("fname,scene,nrows", _DET_GOLDENS)
def test_detections_gpu_match_reference_table(golden_dir, fname, scene, nrows):
    from srcfinder_amd import detections
    gen = _gen_det(golden_dir)
    g = np.load(os.path.join(golden_dir, fname))
    sal, img = getattr(gen, scene)(int(g["lines"]), int(g["samples"]), int(g["seed"]))
    mi = detections.mapinfo([str(v) for v in g["mapinfo"]])     # UTM zone 11 North, rotation 17 degrees
    assert mi["rotation"] == 17.0 and mi["proj"] == "UTM"
    df = detections.salience2detections(sal, img, float(g["prob_thr"]), float(g["ppmm_thr"]), "ang20200101t000000", mi)
    assert list(df["detid"]) == list(g["detid"]) and len(df) == nrows
    got = df[list(g["columns"])].to_numpy(dtype=np.float64)
    geo = [i for i, c in enumerate(g["columns"]) if c.endswith("lat") or c.endswith("lon")]
    rest = [i for i in range(got.shape[1]) if i not in geo]
    assert np.array_equal(got[:, rest], g["table"][:, rest])     # order statistics, positions: exact
    # lat / lon: the same series evaluated through the reference's rotxy (a 2 x 2 dot product) and here in scalar form
    assert np.abs(got[:, geo] - g["table"][:, geo]).max() < 1e-11


@pytest.mark.gpu
def test_detections_gpu_random_scenes_against_oracle(golden_dir):
    """Other sizes and thresholds (a full-width 598-sample strip among them), float32 products, no map info; a region
    without any CMF pixel above the threshold dies like the reference (extrema of an empty selection)."""
    from oracle import detect_oracle as DO
    from srcfinder_amd import detections
    gen = _gen_det(golden_dir)
    for lines, samples, seed, sthr, cthr in ((97, 70, 1, 0.5, 250.0), (400, 598, 2, 0.35, 100.0), (64, 33, 3, 0.6, 300.0)):
        sal, img = gen.scene(lines, samples, seed)
        want = DO.detections(sal, img, sthr, cthr, 0.0, 0.0, 1.0, 1.0)
        hdr, rows = detections.salience2detections(sal[..., 0], img, sthr, cthr, "x", dict(ulx=0, uly=0, xps=1, yps=1),
                                                   as_dataframe=False)
        got = np.array([r[2:] for r in rows], dtype=np.float64).reshape(len(rows), 20)
        assert np.array_equal(got, want), (lines, samples)
    sal, img = gen.scene(97, 70, 1)
    with pytest.raises(ValueError, match="zero-size array"):
        detections.salience2detections(sal, img, 0.5, 1e9, "x")
