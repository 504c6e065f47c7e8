"""Saliency map -> plume detection list on MI355X (first half).

Mirrors ``salience_predictions.py`` ``salience2detections`` (:25-150): threshold the saliency map, label its
8-connected regions, and for every region report the bounding box and the statistics of the saliency and of the CMF
enhancement inside it.  Labelling and the per-region order statistics run on the GPU (``sf_image_label8``,
``sf_detect_region_stats``); the georeferencing of the two maxima is host arithmetic (``srcfinder_util.sl2xy``).
Not mirrored: the per-region PDF figures (:121-146) and the spreadsheet writer (``save_detections``: use the returned
table).  There is no CPU path.
"""
from __future__ import annotations

import numpy as np

from . import _ffi

# salience_predictions.py:32-37.  The reference fills the four bounding-box columns in the order
# (row start, col start, row stop, col stop) under these names (:113); the order is kept.
HEADER = ["detid", "lid", "detbbminr", "detbbmaxr", "detbbminc", "detbbmaxc",
          "salmax", "salmin", "salmed", "salmad", "salmaxrow", "salmaxcol", "salmaxlat", "salmaxlon",
          "cmfmax", "cmfmin", "cmfmed", "cmfmad", "cmfmaxrow", "cmfmaxcol", "cmfmaxlat", "cmfmaxlon"]


def sl2xy(s, l, mapinfo):
    """Map coordinates of (sample, line): srcfinder_util.sl2xy without rotation (``map info``: ulx, uly, xps, yps)."""
    ulx, uly, xps = float(mapinfo["ulx"]), float(mapinfo["uly"]), float(mapinfo["xps"])
    yps = float(mapinfo.get("yps", xps)) or xps
    if float(mapinfo.get("rotation", 0) or 0) != 0:
        raise NotImplementedError("rotated map info")
    return ulx + xps * s, uly - yps * l


def salience2detections(salimg, cmfimg, salthr, cmfthr, cmflid, cmfmap=None, latlon=None, as_dataframe=True):
    """salimg [H, W] or [H, W, C] float32 saliency (last channel; two channels are normalised by their sum, :41-42);
    cmfimg [H, W, 4] float64 product (R, G, B, CMF).  ``cmfmap``: dict with ulx, uly, xps, yps (map info) or None;
    ``latlon(x, y) -> (lat, lon)`` converts map coordinates (the reference calls a UTM library; without it the lat / lon
    columns hold (y, x) map coordinates, NaN when there is no map info).  Returns the table of :148 (DataFrame, or a
    (header, rows) pair with ``as_dataframe=False``)."""
    import torch
    if not torch.cuda.is_available():
        raise _ffi.SrcfinderError("no GPU visible: srcfinder_amd has no CPU fallback")
    sal = salimg if torch.is_tensor(salimg) else torch.as_tensor(np.ascontiguousarray(salimg))
    cmf = cmfimg if torch.is_tensor(cmfimg) else torch.as_tensor(np.ascontiguousarray(cmfimg, dtype=np.float64))
    assert cmf.dim() == 3 and cmf.shape[2] == 4                                      # :29
    sal = sal if sal.is_cuda else sal.cuda()
    cmf = (cmf if cmf.is_cuda else cmf.cuda()).to(torch.float64).contiguous()
    if sal.dim() == 3:
        salpos = sal[..., -1]
        if sal.shape[-1] == 2:
            salpos = salpos / sal.sum(dim=2)
    else:
        salpos = sal
    salpos = salpos.to(torch.float32).contiguous()
    H, W = salpos.shape
    dev = salpos.device
    L = _ffi.lib()
    P, st = _ffi.ptr, _ffi.stream_ptr
    with torch.cuda.device(dev):
        nodata = (cmf[..., 0] == -9999).to(torch.uint8).contiguous()                 # :45
        salmask = (salpos > float(salthr)).to(torch.uint8).contiguous()              # :60
        labels = torch.empty((H, W), dtype=torch.int32, device=dev)
        ncomp = torch.empty(1, dtype=torch.int32, device=dev)
        scratch = torch.empty(L.sf_image_label8_scratch_bytes(H, W), dtype=torch.uint8, device=dev)
        _ffi.check(L.sf_image_label8(P(salmask), H, W, P(labels), None, 0, P(ncomp), P(scratch), st()), "sf_image_label8")
        n = int(ncomp.item())
        rec = torch.zeros((n + 1, 20), dtype=torch.float64, device=dev)
        bbox = torch.empty((n + 1, 4), dtype=torch.int32, device=dev)
        _ffi.check(L.sf_detect_region_stats(P(labels), H, W, n, P(salpos), P(cmf), 4, 3, P(nodata), float(cmfthr), P(bbox),
                                            P(rec), st()), "sf_detect_region_stats")
        rec = rec.cpu().numpy()[1:]
    if n and rec[:, 18].any():
        raise _ffi.SrcfinderError("a saliency region exceeds the LDS-resident sort (32768 saliency / 16384 CMF pixels)")
    if n and (rec[:, 17] == 0).any():
        # the reference takes extrema() of an empty selection here and dies (numpy: zero-size array, :98)
        raise ValueError("zero-size array to reduction operation fmin which has no identity")
    rows = []
    for i in range(n):
        r = rec[i]
        imin, imax, jmin, jmax = int(r[0]), int(r[1]), int(r[2]), int(r[3])
        pmi, pmj, cmi, cmj = int(r[8]), int(r[9]), int(r[15]), int(r[16])
        if cmfmap is not None:
            px, py = sl2xy(pmj, pmi, cmfmap)
            cx, cy = sl2xy(cmj, cmi, cmfmap)
            plat, plon = latlon(px, py) if latlon else (py, px)
            clat, clon = latlon(cx, cy) if latlon else (cy, cx)
        else:
            plat = plon = clat = clon = float("nan")
        rows.append(["%s-%d" % (cmflid, i + 1), cmflid, imin, jmin, imax, jmax,
                     r[4], r[5], r[6], r[7], pmi, pmj, plat, plon, r[11], r[12], r[13], r[14], cmi, cmj, clat, clon])
    if not as_dataframe:
        return HEADER, rows
    from pandas import DataFrame
    return DataFrame.from_records(rows, columns=HEADER)
