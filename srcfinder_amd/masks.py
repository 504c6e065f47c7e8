"""Spectrometer masks (cloud / specular / flare / dark) of a radiance cube on MI355X.

Mirrors ``spectrometer_masks/masks_sds.py`` (a command-line script upstream; the keyword names below are its flags,
:66-107): the per-pixel rules run in one HIP kernel over the BIL cube that the matched filter already holds in HBM, the
cloud buffer is N passes of a cross dilation, the flare buffer a disk dilation of the "grow" pixels -- block by block
as the script does it (:283-327), because its minimum-area rule labels each block on its own.  Integer results,
bit-exact.  There is no CPU path.
"""
from __future__ import annotations

import numpy as np

from . import _ffi

SAT_THRESH_DEFAULT = 6.0            # masks_sds.py:49
SAT_THRESH_CLD = (15.0,)            # :51
BAND_NAMES = ["Cloud mask (dimensionless)", "Specular mask (dimensionless)", "Flare mask (dimensionless)",
              "Dark mask (dimensionless)"]                                         # :347


def radius_in_pixels(value_str, metadata=None):
    """``get_radius_in_pixels`` (:234-250): '12px' -> ceil(12); '150m' -> ceil(150 / pixel size from 'map info')."""
    if value_str.endswith("px"):
        return float(np.ceil(float(value_str.split("px")[0])))
    if value_str.endswith("m"):
        return float(np.ceil(_meters(value_str, metadata)))
    raise RuntimeError("Unknown unit specified.")


def _meters(value_str, metadata):
    if not metadata or "map info" not in metadata:
        raise RuntimeError("Image does not have resolution specified. Try giving values in pixels.")
    mi = metadata["map info"]
    if "meters" not in str(mi[10]).lower():
        raise RuntimeError("Unknown unit for image resolution.")
    mx, my = float(mi[5]), float(mi[6])
    if mx != my:
        mx = (my + mx) / 2.0
    return float(value_str.split("m")[0]) / mx


def cloud_buffer_passes(value_str, metadata=None):
    """``dilate_mask`` (:252-273): the number of cross-dilation passes, ``int(ceil(dil_u))``."""
    dil_u = float(np.ceil(float(value_str.split("px")[0]))) if value_str.endswith("px") else _meters(value_str, metadata)
    return int(np.ceil(dil_u))


def spectrometer_masks(cube_bil, wavelengths, *, saturationthreshold=None, saturationwindow=None, cldthreshold=SAT_THRESH_CLD,
                       cldbands=None, cldbfr="150m", maskgrowradius="150m", mingrowarea=None,
                       saturation_processing_block_length=500, visible_mask_growing_threshold=9.0, dark_threshold=0.104,
                       metadata=None, to_numpy=False):
    """cube_bil: [lines, bands, samples] float32 (torch tensor on the GPU, or ndarray).  Returns the product
    int16 [lines, samples, 4] = (cloud incl. buffer, specular, flare: 2 = buffer / 1 = flare, dark), -9999 on the image
    border -- the array the script hands to ``spectral.envi.save_image`` (:336-341)."""
    import torch
    if not torch.cuda.is_available():
        raise _ffi.SrcfinderError("no GPU visible: srcfinder_amd has no CPU fallback")
    t = cube_bil if torch.is_tensor(cube_bil) else torch.as_tensor(np.ascontiguousarray(cube_bil, dtype=np.float32))
    if t.dtype != torch.float32 or t.dim() != 3:
        raise TypeError("cube must be float32 [lines, bands, samples]")
    t = (t if t.is_cuda else t.cuda()).contiguous()
    dev = t.device
    lines, bands, samples = t.shape
    wave = np.asarray(wavelengths, dtype=np.float64)
    if wave.shape != (bands,):
        raise ValueError("one wavelength per band")
    lo, hi = saturationwindow if saturationwindow is not None else (1945, 2485)
    win = np.flatnonzero((wave >= lo) & (wave <= hi))
    if len(win) == 0:
        sat_b0 = sat_b1 = 0
    else:
        sat_b0, sat_b1 = int(win[0]), int(win[-1]) + 1
        if sat_b1 - sat_b0 != len(win):
            raise NotImplementedError("saturation window that is not one contiguous band range")
    thr = SAT_THRESH_DEFAULT if saturationthreshold is None else float(saturationthreshold)
    cb = tuple(int(b) for b in (cldbands if cldbands is not None else (15, 60, 175)))
    dwl = float(wave[cb[1]] - wave[cb[0]])
    idx500 = int(np.argmin(np.abs(wave - 500)))
    L = _ffi.lib()
    P, st = _ffi.ptr, _ffi.stream_ptr
    u8 = dict(dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        sat, cloud, spec, dark, grow, border = (torch.empty((lines, samples), **u8) for _ in range(6))
        _ffi.check(L.sf_masks_pixel(P(t), lines, bands, samples, sat_b0, sat_b1, thr, cb[0], cb[1], float(cldthreshold[0]),
                                    dwl, 25, float(visible_mask_growing_threshold), 352, float(dark_threshold), idx500,
                                    P(sat), P(cloud), P(spec), P(dark), P(grow), P(border), st()), "sf_masks_pixel")
        # ---- cloud buffer (:329-333)
        tmp = torch.empty_like(cloud)
        _ffi.check(L.sf_image_dilate_cross(P(cloud), P(tmp), lines, samples, cloud_buffer_passes(cldbfr, metadata), st()),
                   "sf_image_dilate_cross")
        # ---- flare buffer, block by block (:283-327)
        flare_buf = None
        sat_for_flare = sat
        if maskgrowradius is not None:
            r = int(radius_in_pixels(maskgrowradius, metadata))
            overlap = int(np.ceil((mingrowarea if mingrowarea is not None else 0) + r))
            step = int(saturation_processing_block_length)
            flare_buf = torch.zeros((lines, samples), **u8)
            sat_for_flare = torch.zeros((lines, samples), **u8)       # saturated pixels of the blocks that wrote (see below)
            nmax = (step + overlap) * samples
            scratch = torch.empty(max(L.sf_image_dilate_disk_scratch_bytes(step + overlap, samples, r),
                                      L.sf_image_label8_scratch_bytes(step + overlap, samples)), **u8)
            labels = torch.empty(nmax, dtype=torch.int32, device=dev)
            area = torch.empty(nmax + 1, dtype=torch.int32, device=dev)
            ncomp = torch.empty(1, dtype=torch.int32, device=dev)
            grown = torch.empty(nmax, **u8)
            for a in range(0, lines, step):
                b = min(lines, a + step + overlap)
                h = b - a
                sat_b, grow_b = sat[a:b], grow[a:b].clone()
                if mingrowarea is not None:
                    _ffi.check(L.sf_image_label8(P(sat_b), h, samples, P(labels), P(area), h * samples + 1, P(ncomp), P(scratch),
                                                 st()), "sf_image_label8")
                    _ffi.check(L.sf_image_filter_small_components(P(labels), P(area), int(mingrowarea), P(grow_b), h, samples,
                                                                  st()), "sf_image_filter_small_components")
                    # the script's assignments sit inside its loop over the qualifying regions (:310-327): a block in
                    # which no region reaches the minimum area writes nothing, not even its flare pixels
                    wrote = (area[1:h * samples + 1] >= int(mingrowarea)).any()
                else:
                    wrote = sat_b.any()
                _ffi.check(L.sf_image_dilate_disk(P(grow_b), P(grown), h, samples, r, P(scratch), st()), "sf_image_dilate_disk")
                g = grown[:h * samples].view(h, samples)
                flare_buf[a:b] |= g & wrote
                sat_for_flare[a:b] |= sat_b & wrote
        out = torch.empty((lines, samples, 4), dtype=torch.int16, device=dev)
        _ffi.check(L.sf_masks_compose(P(cloud), P(spec), P(sat_for_flare), P(flare_buf), P(dark), P(border), lines, samples,
                                      P(out), st()), "sf_masks_compose")
    return out.cpu().numpy() if to_numpy else out


def label(mask, to_numpy=False):
    """8-connected components of a boolean image on the GPU: (labels int32 [H, W], n).  Numbering follows the raster
    order of the components' first pixels, as ``skimage.measure.label(mask, connectivity=2)`` /
    ``scipy.ndimage.label(mask, ones((3, 3)))`` (srcfinder_util.imlabel, masks_sds.py:309)."""
    import torch
    if not torch.cuda.is_available():
        raise _ffi.SrcfinderError("no GPU visible: srcfinder_amd has no CPU fallback")
    m = mask if torch.is_tensor(mask) else torch.as_tensor(np.ascontiguousarray(mask))
    m = (m != 0).to(torch.uint8)
    m = (m if m.is_cuda else m.cuda()).contiguous()
    H, W = m.shape
    L = _ffi.lib()
    with torch.cuda.device(m.device):
        labels = torch.empty((H, W), dtype=torch.int32, device=m.device)
        ncomp = torch.empty(1, dtype=torch.int32, device=m.device)
        scratch = torch.empty(L.sf_image_label8_scratch_bytes(H, W), dtype=torch.uint8, device=m.device)
        _ffi.check(L.sf_image_label8(_ffi.ptr(m), H, W, _ffi.ptr(labels), None, 0, _ffi.ptr(ncomp), _ffi.ptr(scratch),
                                     _ffi.stream_ptr()), "sf_image_label8")
    n = int(ncomp.item())
    return (labels.cpu().numpy() if to_numpy else labels), n
