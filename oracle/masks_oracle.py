"""TEST INFRASTRUCTURE -- numpy restatement of ``spectrometer_masks/masks_sds.py``.

Pinned by ``tests/golden/masks_golden.npz``: products written by the REAL script, executed unmodified
(``tests/golden/gen_golden_masks.py``).  The per-pixel rules (saturation :133-151, specular :153-163, dark :165-179,
cloud :181-232, border :341) are the script's own numpy code.  The script's morphology calls go to ``skimage``
(``morphology.binary_dilation``, ``morphology.disk``, ``measure.label``, ``measure.regionprops``), a dependency that is
not under /root/reference and not installed here (the reference pins scikit-image through its conda environment): the
generator puts ``scipy.ndimage`` equivalents of their published definitions in its place -- binary dilation by a
structuring element with a background border, ``disk(r)`` = {x^2 + y^2 <= r^2}, 2-connected (8-neighbour) labelling in
raster order -- so the morphology of the golden products is pinned to those definitions, not to skimage's binaries.

Only ``tests/`` may import this file.
"""
import numpy as np
import scipy.ndimage as ndi

SAT_THRESH_DEFAULT = 6.0          # masks_sds.py:49
SAT_THRESH_CLD = (15.0,)          # :51
CROSS = ndi.generate_binary_structure(2, 1)
EIGHT = ndi.generate_binary_structure(2, 2)


def disk(r):
    r = int(r)
    y, x = np.ogrid[-r:r + 1, -r:r + 1]
    return x * x + y * y <= r * r


def pixel_rules(bip, wavelengths, sat_thr=SAT_THRESH_DEFAULT, sat_window=(1945, 2485), cloud_thr=15.0,
                cloud_bands=(15, 60, 175), vis_thr=9.0, dark_thr=0.104):
    """bip: [lines, samples, bands] float32.  Returns dict of boolean [lines, samples] masks."""
    wave = np.asarray(wavelengths, np.float64)
    win = np.logical_and(wave >= sat_window[0], wave <= sat_window[1])
    sat = (bip[..., win] > sat_thr).any(axis=-1)                                          # :150
    r1, r2 = bip[..., cloud_bands[0]], bip[..., cloud_bands[1]]
    der_a = (r2 - r1) / (-(wave[cloud_bands[0]] - wave[cloud_bands[1]]))                  # :206-219
    with np.errstate(invalid="ignore"):
        cloud = (r1 > cloud_thr) & (der_a < 0)         # :230 -- the second slope is logical_and's `out`, not an operand
    spec = sat & (bip[..., 25] > vis_thr)                                                 # :160-162
    dk = bip[..., 352]
    dark = (dk < dark_thr) & ~(dk <= -9999)                                               # :175-178
    idx500 = int(np.argmin(np.abs(wave - 500)))                                           # :281
    return dict(sat=sat, cloud=cloud, spec=spec, dark=dark, idx500=idx500, border=bip[..., 0] == -9999)


def masks(bip, wavelengths, *, sat_thr=SAT_THRESH_DEFAULT, sat_window=(1945, 2485), cloud_thr=15.0,
          cloud_bands=(15, 60, 175), cloud_buffer_px=50, grow_radius_px=50, mingrowarea=None, vis_thr=9.0,
          dark_thr=0.104, block=500):
    """The product int16 [lines, samples, 4] (cloud, specular, flare, dark) of masks_sds.py:283-341, block loop included.
    ``grow_radius_px`` None = no flare buffer (``--maskgrowradius`` absent)."""
    lines, samples, _ = bip.shape
    pr = pixel_rules(bip, wavelengths, sat_thr, sat_window, cloud_thr, cloud_bands, vis_thr, dark_thr)
    flare = np.zeros((lines, samples), np.uint8)
    if grow_radius_px is not None:
        overlap = int(np.ceil((mingrowarea or 0) + grow_radius_px))                       # :288
        selem = disk(grow_radius_px)
        for a in range(0, lines, block):                                                   # :291-327
            b = min(lines, a + block + overlap)
            satb = pr["sat"][a:b]
            lab, n = ndi.label(satb, structure=EIGHT)
            grow = np.zeros_like(satb)
            hit = False
            for i in range(1, n + 1):
                reg = lab == i
                if mingrowarea is None or reg.sum() >= mingrowarea:
                    hit = True
                    grow |= reg & (bip[a:b, :, pr["idx500"]] < vis_thr)
            if hit:       # the assignments sit inside the loop over qualifying regions: a block without one writes nothing
                grown = ndi.binary_dilation(grow, structure=selem)
                flare[a:b][grown] = 2
                flare[a:b][satb & ~pr["spec"][a:b]] = 1
    cloud = pr["cloud"].copy()
    for _ in range(int(np.ceil(cloud_buffer_px))):                                        # :270-272
        cloud = ndi.binary_dilation(cloud, structure=CROSS)
    out = np.zeros((lines, samples, 4), np.int16)
    out[..., 0], out[..., 1], out[..., 2], out[..., 3] = cloud, pr["spec"], flare, pr["dark"]
    out[pr["border"]] = -9999
    return out
