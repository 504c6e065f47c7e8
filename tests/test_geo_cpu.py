"""Georeferencing of the detection list: ``srcfinder_amd.detections`` mirrors ``srcfinder_util``'s
``mapinfo`` / ``rotxy`` / ``sl2xy`` / ``utm2latlon`` / ``sl2latlon`` (:766-877, :987-1024); the UTM -> lat/lon series
itself belongs to a third-party module the reference imports and this image lacks -- ``oracle/utm_oracle.py`` states it and
is pinned here against a printed worked example and an independent projection.  CPU only (host arithmetic)."""
import math
import os

import numpy as np
import pytest

from oracle import utm_oracle as U
from srcfinder_amd import detections as D

SAMPLE_MAPINFO = "UTM, 1, 1, 272247.152557, 3992010.65018, 3.1, 3.1, 11, North, WGS-84, units=Meters, rotation=17.0000000"


def kruger_forward(lat, lon, lon0, a=6378137.0, f=1 / 298.257223563, k0=0.9996):
    """Transverse Mercator by the Krueger n-series (Karney 2011, eqs. 7-11 and 35, to n^6): an independent forward
    projection, accurate to nanometres inside a UTM zone.  Returns (easting, northing) with the 500 km false easting."""
    n = f / (2 - f)
    A = a / (1 + n) * (1 + n ** 2 / 4 + n ** 4 / 64 + n ** 6 / 256)
    al = [n / 2 - 2 * n ** 2 / 3 + 5 * n ** 3 / 16 + 41 * n ** 4 / 180 - 127 * n ** 5 / 288 + 7891 * n ** 6 / 37800,
          13 * n ** 2 / 48 - 3 * n ** 3 / 5 + 557 * n ** 4 / 1440 + 281 * n ** 5 / 630 - 1983433 * n ** 6 / 1935360,
          61 * n ** 3 / 240 - 103 * n ** 4 / 140 + 15061 * n ** 5 / 26880 + 167603 * n ** 6 / 181440,
          49561 * n ** 4 / 161280 - 179 * n ** 5 / 168 + 6601661 * n ** 6 / 7257600,
          34729 * n ** 5 / 80640 - 3418889 * n ** 6 / 1995840,
          212378941 * n ** 6 / 319334400]
    e = math.sqrt(f * (2 - f))
    phi, lam = math.radians(lat), math.radians(lon - lon0)
    tau = math.tan(phi)
    sig = math.sinh(e * math.atanh(e * tau / math.sqrt(1 + tau * tau)))
    taup = tau * math.sqrt(1 + sig * sig) - sig * math.sqrt(1 + tau * tau)
    xi = math.atan2(taup, math.cos(lam))
    eta = math.asinh(math.sin(lam) / math.sqrt(taup * taup + math.cos(lam) ** 2))
    x = eta + sum(al[j] * math.cos(2 * (j + 1) * xi) * math.sinh(2 * (j + 1) * eta) for j in range(6))
    y = xi + sum(al[j] * math.sin(2 * (j + 1) * xi) * math.cosh(2 * (j + 1) * eta) for j in range(6))
    return k0 * A * x + 500000.0, k0 * A * y


def test_utm_series_snyder_worked_example():
    """Snyder, Map Projections -- A Working Manual (USGS PP 1395), p. 269-270: Clarke 1866, central meridian 75 W,
    k0 = 0.9996, the point 40 30' N, 73 30' W has x = 127,106.5 m, y = 4,484,124.4 m (printed to 0.1 m)."""
    lat, lon = U.UTMtoLL(5, 4484124.4, 127106.5 + 500000.0, None, _lon0=-75.0)
    assert abs(lat - 40.5) < 1e-6 and abs(lon + 73.5) < 1e-6          # 0.1 m is 9e-7 degrees
    zone, e, n = U.LLtoUTM(5, 40.5, -73.5, _lon0=-75.0)
    assert abs(e - 500000.0 - 127106.5) < 0.06 and abs(n - 4484124.4) < 0.06


def test_utm_series_against_an_independent_projection():
    """WGS-84: project what the inverse series returns with the Krueger series; the truncated series of the module is good
    to a few tenths of a millimetre within a zone (its ellipsoid table rounds e^2 to 0.00669438: included in the bound)."""
    rng = np.random.default_rng(5)
    worst = worst_rt = 0.0
    for zone in (11, 12, 33, 55):
        lon0 = (zone - 1) * 6 - 180 + 3
        for _ in range(100):
            lat, lon = rng.uniform(0.0, 80.0), lon0 + rng.uniform(-3.0, 3.0)      # inside the zone
            e, n = kruger_forward(lat, lon, lon0)
            lat2, lon2 = U.UTMtoLL(23, n, e, "%dN" % zone)
            worst = max(worst, 111320.0 * math.hypot(float(lat2) - lat, (float(lon2) - lon) * math.cos(math.radians(lat))))
            z3, e3, n3 = U.LLtoUTM(23, lat, lon, _lon0=lon0)
            worst_rt = max(worst_rt, math.hypot(e3 - e, n3 - n))
    assert worst < 5e-3 and worst_rt < 5e-3, (worst, worst_rt)             # metres
    lat_s, lon_s = U.UTMtoLL(23, 10000000.0 - 3000000.0, 500000.0, "33M")   # southern hemisphere: false northing
    lat_n, lon_n = U.UTMtoLL(23, 3000000.0, 500000.0, "33N")
    assert abs(lat_s + lat_n) < 1e-12 and lon_s == lon_n == 15.0


def test_product_series_equals_the_oracle():
    rng = np.random.default_rng(6)
    e, n = rng.uniform(150000, 850000, 200), rng.uniform(0, 9000000, 200)
    for zone in ("11N", "55M"):
        a = D.utm_to_latlon(n, e, zone)
        b = U.UTMtoLL(23, n, e, zone)
        assert np.abs(a[0] - b[0]).max() < 1e-12 and np.abs(a[1] - b[1]).max() < 1e-12


def test_mapinfo_rotation_and_sl2latlon_on_the_sample_header():
    """The reference's only sample product (cnn/samples/ang20200924t211102_ch4mf_v2y1_img.hdr) is rotated by 17 degrees."""
    mi = D.mapinfo(SAMPLE_MAPINFO)
    assert mi["proj"] == "UTM" and mi["zone"] == "11" and mi["hemi"] == "North" and mi["rotation"] == 17.0
    assert mi["units"] == "Meters" and mi["xps"] == 3.1 and mi["ulx"] == 272247.152557
    assert D.sl2xy(0, 0, mi) == (mi["ulx"], mi["uly"])
    # rotxy: counter-clockwise about the upper-left corner (srcfinder_util.py:782-787)
    x, y = D.sl2xy(100, 0, mi)
    c, s = math.cos(math.radians(17)), math.sin(math.radians(17))
    assert abs(x - (mi["ulx"] + 310.0 * c)) < 1e-9 and abs(y - (mi["uly"] + 310.0 * s)) < 1e-9
    x, y = D.sl2xy(0, 100, mi)
    assert abs(x - (mi["ulx"] + 310.0 * s)) < 1e-9 and abs(y - (mi["uly"] - 310.0 * c)) < 1e-9
    lat, lon = D.sl2latlon(334, 1400, mi)                       # the middle of the 669 x 2801 sample product
    assert 35.9 < lat < 36.1 and -119.6 < lon < -119.4          # California's Central Valley
    e, n = kruger_forward(float(lat), float(lon), -117.0)
    xm, ym = D.sl2xy(334, 1400, mi)
    assert math.hypot(e - xm, n - ym) < 5e-3
    # error behaviour of the reference
    with pytest.raises(ValueError, match="proj undefined"):
        D.sl2latlon(0, 0, dict(ulx=0.0, uly=0.0, xps=1.0))
    with pytest.raises(ValueError, match="ulx or uly undefined"):
        D.sl2xy(0, 0, dict(proj="UTM", xps=1.0))
    assert D.sl2latlon(3, 4, dict(proj="Geographic Lat/Lon", ulx=10.0, uly=50.0, xps=0.5, yps=0.25)) == (49.0, 11.5)
    assert D.sl2latlon(3, 4, dict(proj="Arbitrary", ulx=10.0, uly=50.0, xps=0.5)) is None
    assert D.utm2latlon(1.0, 2.0, 11, hemi="East") == (None, None)
    assert D.sl2xy(2, 3, dict(ulx=1.0, uly=9.0, xps=2.0, yps=0)) == (5.0, 3.0)   # yps == 0 means xps


def test_envi_header_map_info_round_trip(tmp_path):
    from srcfinder_amd import envi
    hdr = tmp_path / "x.hdr"
    hdr.write_text("ENVI\nsamples = 669\nlines = 2801\nbands = 4\nheader offset = 0\ndata type = 5\ninterleave = bip\n"
                   "byte order = 0\nmap info = {%s}\n" % SAMPLE_MAPINFO)
    meta = envi.read_header(str(hdr))
    mi = D.mapinfo(meta["map info"])
    assert mi["rotation"] == 17.0 and mi["zone"] == "11"
