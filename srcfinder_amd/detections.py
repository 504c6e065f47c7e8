"""Saliency map -> plume detection list on MI355X (first half).

Mirrors ``salience_predictions.py`` ``salience2detections`` (:25-150): threshold the saliency map, label its
8-connected regions, and for every region report the bounding box and the statistics of the saliency and of the CMF
enhancement inside it.  Labelling and the per-region order statistics run on the GPU (``sf_image_label8``,
``sf_detect_region_stats``); the georeferencing of the two maxima is host arithmetic: ``mapinfo`` / ``rotxy`` / ``sl2xy`` / ``utm2latlon`` /
``sl2latlon`` below mirror ``srcfinder_util.py:766-877,:987-1024`` (rotated map info as in the reference's sample product,
``rotation=17``), the UTM -> lat/lon series is the one of the third-party ``LatLongUTMconversion`` module it imports.
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


DEG2RAD = np.pi / 180.0          # srcfinder_util.py:77
DATUM_WGS84 = 23                 # srcfinder_util.py:75 (index into LatLongUTMconversion's ellipsoid table)
_WGS84_A, _WGS84_E2 = 6378137.0, 0.00669438   # that table's entry 23


def mapinfo(maplist):
    """``srcfinder_util.mapinfo`` (:987-1024) for an ENVI ``map info`` entry given as the header's list of strings or
    as its comma-separated text (``envi.read_header`` keeps the text): proj, xtie, ytie, ulx, uly, xps, yps, for UTM
    zone / hemi / datum, the ``key=value`` items (units, rotation), ``rotation`` as float (0 when absent)."""
    if isinstance(maplist, str):
        maplist = [v.strip() for v in maplist.strip().strip("{}").split(",")]
    mi = {"proj": maplist[0], "xtie": float(maplist[1]), "ytie": float(maplist[2]), "ulx": float(maplist[3]),
          "uly": float(maplist[4]), "xps": float(maplist[5]), "yps": float(maplist[6])}
    if mi["proj"] == "UTM":
        mi["zone"], mi["hemi"], mi["datum"] = maplist[7], maplist[8], maplist[9]
    meta = []
    for item in maplist[len(mi):]:
        if "=" in item:
            key, val = [t.strip() for t in item.split("=")]
            mi[key] = val
        else:
            meta.append(item)
    mi["rotation"] = float(mi.get("rotation", "0"))
    if meta:
        mi["metadata"] = meta
    return mi


def rotxy(x, y, adeg, xc, yc):
    """``srcfinder_util.rotxy`` (:766-787): rotate (x, y) about (xc, yc) by ``adeg`` degrees, counter-clockwise."""
    arad = DEG2RAD * adeg
    sinr, cosr = np.sin(arad), np.cos(arad)
    dx, dy = x - xc, y - yc
    return (cosr * dx - sinr * dy) + xc, (sinr * dx + cosr * dy) + yc


def sl2xy(s, l, mapinfo):
    """Map coordinates of (sample, line): ``srcfinder_util.sl2xy`` (:815-857) -- ``(ulx + xps s, uly - yps l)`` rotated
    about the upper-left corner by the header's ``rotation`` (degrees); ``yps == 0`` means ``xps`` (:847-848)."""
    if mapinfo.get("ulx") is None or mapinfo.get("uly") is None:
        raise ValueError("ulx or uly undefined")                                         # :841-842
    if mapinfo.get("xps") is None:
        raise ValueError("xps or yps undefined")                                         # :844-845
    ulx, uly, xps = float(mapinfo["ulx"]), float(mapinfo["uly"]), float(mapinfo["xps"])
    yps = float(mapinfo.get("yps", xps)) or xps
    rot = float(mapinfo.get("rotation", 0) or 0)
    xp, yp = ulx + xps * s, uly - yps * l
    if rot == 0:
        return xp, yp
    return rotxy(xp, yp, rot, ulx, uly)


def utm_to_latlon(northing, easting, zone, a=_WGS84_A, ecc2=_WGS84_E2):
    """(lat, lon) in degrees of a UTM point -- the series of ``LatLongUTMconversion.UTMtoLL`` (the third-party module
    ``srcfinder_util.py:27`` imports; equations of USGS Bulletin 1532 / Snyder 1987 pp. 57-64), WGS-84 by default.
    ``zone`` is the zone number followed by a latitude-band letter; letters below 'N' are southern (false northing)."""
    k0 = 0.9996
    e1 = (1 - np.sqrt(1 - ecc2)) / (1 + np.sqrt(1 - ecc2))
    x = np.asarray(easting, dtype=np.float64) - 500000.0
    y = np.asarray(northing, dtype=np.float64)
    if zone[-1] < "N":
        y = y - 10000000.0
    lon0 = (int(zone[:-1]) - 1) * 6 - 180 + 3
    eccp2 = ecc2 / (1 - ecc2)
    mu = (y / k0) / (a * (1 - ecc2 / 4 - 3 * ecc2 * ecc2 / 64 - 5 * ecc2 * ecc2 * ecc2 / 256))
    phi1 = (mu + (3 * e1 / 2 - 27 * e1 * e1 * e1 / 32) * np.sin(2 * mu)
            + (21 * e1 * e1 / 16 - 55 * e1 * e1 * e1 * e1 / 32) * np.sin(4 * mu) + (151 * e1 * e1 * e1 / 96) * np.sin(6 * mu))
    sp, cp, tp = np.sin(phi1), np.cos(phi1), np.tan(phi1)
    n1 = a / np.sqrt(1 - ecc2 * sp * sp)
    t1, c1 = tp * tp, eccp2 * cp * cp
    r1 = a * (1 - ecc2) / np.power(1 - ecc2 * sp * sp, 1.5)
    d = x / (n1 * k0)
    lat = phi1 - (n1 * tp / r1) * (d * d / 2 - (5 + 3 * t1 + 10 * c1 - 4 * c1 * c1 - 9 * eccp2) * d * d * d * d / 24
                                   + (61 + 90 * t1 + 298 * c1 + 45 * t1 * t1 - 252 * eccp2 - 3 * c1 * c1) * d * d * d * d * d * d / 720)
    lon = (d - (1 + 2 * t1 + c1) * d * d * d / 6
           + (5 - 2 * c1 + 28 * t1 - 3 * c1 * c1 + 8 * eccp2 + 24 * t1 * t1) * d * d * d * d * d / 120) / cp
    return np.degrees(lat), lon0 + np.degrees(lon)


def utm2latlon(easting, northing, zone, hemi="North", alpha=None):
    """``srcfinder_util.utm2latlon`` (:806-813), argument for argument: it hands ``(easting, northing)`` to
    ``UTMtoLL(datum, ...)``, whose parameters are ``(northing, easting)`` -- so the FIRST argument here is treated as the
    northing.  ``sl2latlon`` calls it with ``(y, x)`` (:874) and the two swaps cancel."""
    if hemi not in ("North", "South"):
        print("invalid hemisphere value=", hemi)
        return None, None
    zone_alpha = alpha or ("N" if hemi == "North" else "M")
    return utm_to_latlon(easting, northing, str(zone) + zone_alpha)


def sl2latlon(s, l, mapinfo):
    """``srcfinder_util.sl2latlon`` (:860-877): (lat, lon) of (sample, line) for ``proj`` UTM or Geographic Lat/Lon."""
    proj = mapinfo.get("proj")
    if not proj:
        raise ValueError("proj undefined")                                               # :864-865
    if proj not in ("UTM", "Geographic Lat/Lon"):
        print("unknown projection:", proj)                                               # :866-868
        return None
    x, y = sl2xy(s, l, mapinfo)
    if proj == "Geographic Lat/Lon":
        return y, x
    return utm2latlon(y, x, zone=mapinfo["zone"], hemi=mapinfo["hemi"])


def salience2detections(salimg, cmfimg, salthr, cmfthr, cmflid, cmfmap=None, latlon=None, as_dataframe=True):
    """salimg [H, W] or [H, W, C] float32 saliency (last channel; two channels are normalised by their sum, :41-42);
    cmfimg [H, W, 4] float64 product (R, G, B, CMF).  ``cmfmap``: the product's map info -- ``mapinfo(header list)`` or a
    dict with proj / ulx / uly / xps / yps / rotation (and zone / hemi for UTM) -- or None.  The two maxima of every region are
    georeferenced as the reference does (``sl2latlon``, :109-110: rotated map coordinates, UTM -> lat/lon on WGS-84).  A dict
    without ``proj`` (an unprojected test scene; the reference has no such case) gives the (y, x) map coordinates;
    ``latlon(x, y) -> (lat, lon)`` overrides the conversion; without map info the columns are NaN.  Returns the table
    of :148 (DataFrame, or a (header, rows) pair with ``as_dataframe=False``)."""
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
    if n and rec[:, 18].any():        # (never since round 5: a region of any size is served -- detect.hip selects the order
        raise _ffi.SrcfinderError("sf_detect_region_stats reported a failed region")   # statistics of a large region from global memory)
    if n and (rec[:, 17] == 0).any():
        # the reference takes extrema() of an empty selection here and dies (numpy: zero-size array, :98)
        raise ValueError("zero-size array to reduction operation fmin which has no identity")
    rows = []
    for i in range(n):
        r = rec[i]
        imin, imax, jmin, jmax = int(r[0]), int(r[1]), int(r[2]), int(r[3])
        pmi, pmj, cmi, cmj = int(r[8]), int(r[9]), int(r[15]), int(r[16])
        if cmfmap is not None and (latlon is not None or not cmfmap.get("proj")):
            px, py = sl2xy(pmj, pmi, cmfmap)
            cx, cy = sl2xy(cmj, cmi, cmfmap)
            plat, plon = latlon(px, py) if latlon else (py, px)
            clat, clon = latlon(cx, cy) if latlon else (cy, cx)
        elif cmfmap is not None:
            plat, plon = (float(v) for v in sl2latlon(pmj, pmi, cmfmap))
            clat, clon = (float(v) for v in sl2latlon(cmj, cmi, cmfmap))
        else:
            plat = plon = clat = clon = float("nan")
        rows.append(["%s-%d" % (cmflid, i + 1), cmflid, imin, jmin, imax, jmax,
                     r[4], r[5], r[6], r[7], pmi, pmj, plat, plon, r[11], r[12], r[13], r[14], cmi, cmj, clat, clon])
    if not as_dataframe:
        return HEADER, rows
    from pandas import DataFrame
    return DataFrame.from_records(rows, columns=HEADER)
