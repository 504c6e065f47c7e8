"""TEST INFRASTRUCTURE -- numpy restatement of ``salience_predictions.py:25-150`` (salience2detections), the table only.

Pinned by ``tests/golden/detections_golden.npz``: the table the REAL script assembled for a seeded scene
(``tests/golden/gen_golden_detections.py``; ``skimage.measure.label`` and ``statsmodels.robust.scale.mad`` are absent
here and were replaced by scipy / numpy statements of their definitions, the UTM -> lat/lon module by
``oracle/utm_oracle.py``; the map info is rotated by 17 degrees like the reference's sample product).
Only ``tests/`` may import this file."""
import numpy as np
import scipy.ndimage as ndi

from . import utm_oracle

HEADER = ["detbbminr", "detbbmaxr", "detbbminc", "detbbmaxc", "salmax", "salmin", "salmed", "salmad", "salmaxrow",
          "salmaxcol", "salmaxlat", "salmaxlon", "cmfmax", "cmfmin", "cmfmed", "cmfmad", "cmfmaxrow", "cmfmaxcol",
          "cmfmaxlat", "cmfmaxlon"]                                                      # :32-37 without detid, lid


def sl2xy(s, l, ulx, uly, xps, yps, rot=0.0):
    xp, yp = ulx + xps * s, uly - yps * l                                                 # srcfinder_util.py:850
    if rot == 0:
        return xp, yp
    arad = (np.pi / 180.0) * rot                                                          # rotxy, :782-787
    sinr, cosr = np.sin(arad), np.cos(arad)
    xyp = np.dot([[cosr, -sinr], [sinr, cosr]], np.c_[xp - ulx, yp - uly].T).squeeze()
    return xyp[0] + ulx, xyp[1] + uly


def detections(salimg, cmfimg, salthr, cmfthr, ulx, uly, xps, yps, rot=0.0, zone=None, hemi="North"):
    """``zone`` None: the (y, x) map coordinates stand in the lat / lon columns (scenes without a projection)."""
    def latlon(x, y):
        if zone is None:
            return y, x
        # sl2latlon -> utm2latlon(y, x, ...) -> UTMtoLL(datum, easting := y, northing := x, zone) whose parameters are
        # (northing, easting): srcfinder_util.py:874, :806-812
        la, lo = utm_oracle.UTMtoLL(23, y, x, str(zone) + ("N" if hemi == "North" else "M"))
        return float(la), float(lo)
    salpos = salimg[..., -1]                                                              # :40
    cmfdet = cmfimg[..., 3]
    nodata = cmfimg[..., 0] == -9999                                                      # :45
    cmfmask = cmfdet > cmfthr                                                             # :49
    salreg, n = ndi.label(salpos > salthr, structure=np.ones((3, 3)))                     # :60-61 (connectivity 2)
    rows = []
    for ri, robj in enumerate(ndi.find_objects(salreg)):                                  # :66
        plab = ri + 1
        imin, imax, jmin, jmax = robj[0].start, robj[0].stop, robj[1].start, robj[1].stop
        pmsk = (salreg[robj] == plab) & ~nodata[robj]
        pimg = salpos[robj]
        ppix = pimg[pmsk]
        pmed = np.median(ppix)
        pmad = np.median(np.abs(ppix - pmed))                                             # mad(ppix, medval=pmed), c = 1
        ppmn, ppmx = np.nanmin(ppix), np.nanmax(ppix)                                     # extrema
        pmi, pmj = np.int32(ndi.center_of_mass((pimg * pmsk) == ppmx)) + [imin, jmin]
        cmsk = cmfmask[robj] & pmsk
        cimg = cmfdet[robj]
        cpix = cimg[cmsk]
        cpmn, cpmx = np.nanmin(cpix), np.nanmax(cpix)
        cmed = np.median(cpix)
        cmad = np.median(np.abs(cpix - cmed))
        cmi, cmj = np.int32(ndi.center_of_mass((cimg * cmsk) == cpmx)) + [imin, jmin]
        plat, plon = latlon(*sl2xy(pmj, pmi, ulx, uly, xps, yps, rot))
        clat, clon = latlon(*sl2xy(cmj, cmi, ulx, uly, xps, yps, rot))
        rows.append([imin, jmin, imax, jmax, ppmx, ppmn, pmed, pmad, pmi, pmj, plat, plon,   # (the header names :32 do not
                     cpmx, cpmn, cmed, cmad, cmi, cmj, clat, clon])                           #  match this order :113)
    return np.array(rows, dtype=np.float64).reshape(len(rows), len(HEADER))
