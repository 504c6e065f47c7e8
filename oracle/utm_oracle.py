"""TEST INFRASTRUCTURE -- restatement of the third-party module the reference imports for UTM <-> lat/lon.

``srcfinder_util.py:27`` does ``from LatLongUTMconversion import UTMtoLL, LLtoUTM``; that module is NOT in the reference
checkout and is not pinned anywhere in it (``environment.yml`` does not list it): it is the widely copied public
``LatLongUTMconversion.py`` (a Python transcription of Chuck Gantz's C++ routines, equations from USGS Bulletin 1532 /
Snyder, "Map Projections -- A Working Manual", 1987, pp. 57-64).  What is restated here is that published algorithm with
the module's call signatures: ``UTMtoLL(ReferenceEllipsoid, northing, easting, zone)`` -- northing FIRST -- and
``LLtoUTM(ReferenceEllipsoid, Lat, Long)``; ellipsoid 23 is WGS-84 (a = 6378137, e^2 = 0.00669438, the module's table).
The reference's call chain ``sl2latlon -> utm2latlon(y, x, ...) -> UTMtoLL(datum, easting, northing, zone)``
(``srcfinder_util.py:874``, ``:806-812``) swaps the pair twice, so the map's y ends up as ``northing`` and x as ``easting``.

Parity is pinned three ways (``tests/test_geo_cpu.py``): Snyder's printed numerical example for the transverse Mercator
(Clarke 1866, p. 269-270), an independent Krueger-series forward projection (Karney 2011, 6th order in the third
flattening) of the points this inverse returns, and the forward/inverse round trip of the module's own pair.
``tests/golden/gen_golden_detections.py`` hands THIS function to the real ``srcfinder_util`` in place of the absent module.
Only ``tests/`` (and the golden generator) may import this file."""
import math

import numpy as np

# the module's ellipsoid table, entries used here: index -> (name, equatorial radius, eccentricity squared)
ELLIPSOID = {5: ("Clarke 1866", 6378206.4, 0.006768658), 23: ("WGS-84", 6378137.0, 0.00669438)}
K0 = 0.9996


def UTMtoLL(ReferenceEllipsoid, northing, easting, zone, _a=None, _e2=None, _lon0=None):
    """(lat, lon) in degrees.  ``zone`` = number + latitude-band letter ('11N'); letters >= 'N' are northern."""
    a = _a if _a is not None else ELLIPSOID[ReferenceEllipsoid][1]
    ecc2 = _e2 if _e2 is not None else ELLIPSOID[ReferenceEllipsoid][2]
    e1 = (1 - math.sqrt(1 - ecc2)) / (1 + math.sqrt(1 - ecc2))
    x = np.asarray(easting, dtype=np.float64) - 500000.0        # remove the false easting
    y = np.asarray(northing, dtype=np.float64)
    if _lon0 is None:
        letter, number = zone[-1], int(zone[:-1])
        if letter < "N":
            y = y - 10000000.0                                   # false northing of the southern hemisphere
        lon0 = (number - 1) * 6 - 180 + 3                        # central meridian of the zone
    else:
        lon0 = _lon0
    eccp2 = ecc2 / (1 - ecc2)
    M = y / K0
    mu = M / (a * (1 - ecc2 / 4 - 3 * ecc2 * ecc2 / 64 - 5 * ecc2 * ecc2 * ecc2 / 256))
    phi1 = (mu + (3 * e1 / 2 - 27 * e1 * e1 * e1 / 32) * np.sin(2 * mu)
            + (21 * e1 * e1 / 16 - 55 * e1 * e1 * e1 * e1 / 32) * np.sin(4 * mu)
            + (151 * e1 * e1 * e1 / 96) * np.sin(6 * mu))
    N1 = a / np.sqrt(1 - ecc2 * np.sin(phi1) * np.sin(phi1))
    T1 = np.tan(phi1) * np.tan(phi1)
    C1 = eccp2 * np.cos(phi1) * np.cos(phi1)
    R1 = a * (1 - ecc2) / np.power(1 - ecc2 * np.sin(phi1) * np.sin(phi1), 1.5)
    D = x / (N1 * K0)
    lat = phi1 - (N1 * np.tan(phi1) / R1) * (D * D / 2 - (5 + 3 * T1 + 10 * C1 - 4 * C1 * C1 - 9 * eccp2) * D * D * D * D / 24
                                             + (61 + 90 * T1 + 298 * C1 + 45 * T1 * T1 - 252 * eccp2 - 3 * C1 * C1)
                                             * D * D * D * D * D * D / 720)
    lon = (D - (1 + 2 * T1 + C1) * D * D * D / 6
           + (5 - 2 * C1 + 28 * T1 - 3 * C1 * C1 + 8 * eccp2 + 24 * T1 * T1) * D * D * D * D * D / 120) / np.cos(phi1)
    return np.degrees(lat), lon0 + np.degrees(lon)


def LLtoUTM(ReferenceEllipsoid, Lat, Long, _a=None, _e2=None, _lon0=None):
    """(zone string, easting, northing); the forward series of the same module (used for the round-trip test only)."""
    a = _a if _a is not None else ELLIPSOID[ReferenceEllipsoid][1]
    ecc2 = _e2 if _e2 is not None else ELLIPSOID[ReferenceEllipsoid][2]
    LongTemp = (Long + 180) - int((Long + 180) / 360) * 360 - 180
    number = int((LongTemp + 180) / 6) + 1
    lon0 = (number - 1) * 6 - 180 + 3 if _lon0 is None else _lon0
    la, lo, l0 = math.radians(Lat), math.radians(LongTemp), math.radians(lon0)
    eccp2 = ecc2 / (1 - ecc2)
    N = a / math.sqrt(1 - ecc2 * math.sin(la) ** 2)
    T = math.tan(la) ** 2
    C = eccp2 * math.cos(la) ** 2
    A = math.cos(la) * (lo - l0)
    M = a * ((1 - ecc2 / 4 - 3 * ecc2 ** 2 / 64 - 5 * ecc2 ** 3 / 256) * la
             - (3 * ecc2 / 8 + 3 * ecc2 ** 2 / 32 + 45 * ecc2 ** 3 / 1024) * math.sin(2 * la)
             + (15 * ecc2 ** 2 / 256 + 45 * ecc2 ** 3 / 1024) * math.sin(4 * la)
             - (35 * ecc2 ** 3 / 3072) * math.sin(6 * la))
    easting = K0 * N * (A + (1 - T + C) * A ** 3 / 6 + (5 - 18 * T + T * T + 72 * C - 58 * eccp2) * A ** 5 / 120) + 500000.0
    northing = K0 * (M + N * math.tan(la) * (A * A / 2 + (5 - T + 9 * C + 4 * C * C) * A ** 4 / 24
                                             + (61 - 58 * T + T * T + 600 * C - 330 * eccp2) * A ** 6 / 720))
    if Lat < 0:
        northing += 10000000.0
    return "%d%s" % (number, "N" if Lat >= 0 else "M"), easting, northing
