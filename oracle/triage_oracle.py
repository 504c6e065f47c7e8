"""TEST INFRASTRUCTURE -- numpy restatement of the column profile of ``triage/cmf_profile.py:110-140``.

Pinned: ``tests/golden/triage_profile.npz`` holds the two CSV tables (plain and ``--robust``) that the REAL script wrote
for a seeded synthetic product -- ``cmf_profile.py`` executed unmodified with the real ``srcfinder_util`` (its file
reader replaced by an in-memory product, the absent third-party imports stubbed; ``tests/golden/gen_golden_triage.py``).
``tests/test_oracle_golden.py::test_triage_profile_golden`` checks this restatement against them bit for bit.
Follows the listing line by line: float32 cast :112, validity + positivity mask :113-114, nan-statistics :124-131."""
import warnings

import numpy as np


def column_profile(cmf_plane, nodata=-9999.0):
    nodatav = np.float32(nodata)
    cmf = np.float32(np.array(cmf_plane, copy=True))
    cmfnodata = (cmf == nodatav) | np.isnan(cmf)
    cmfmask = ~cmfnodata & (cmf > 0)
    cmf[~cmfmask] = np.nan
    colnum = np.count_nonzero(cmfmask, axis=0)
    with np.errstate(all="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        colavg = np.nanmean(cmf, axis=0)
        colstd = np.nanstd(cmf, axis=0)
        colmin = np.nanmin(cmf, axis=0)
        colmax = np.nanmax(cmf, axis=0)
    return np.stack([colnum, colavg, colstd, colmin, colmax]).astype(np.float64)


def column_profile_robust(cmf_plane, nodata=-9999.0, p=0.95):
    """use_robust_stats branch (:124-127): nanmedian, nanmedian(|x - med|), extrema(p) = nearest percentiles
    (srcfinder_util.py:647-653; ``interpolation='nearest'`` is ``method='nearest'`` in numpy >= 1.22)."""
    nodatav = np.float32(nodata)
    cmf = np.float32(np.array(cmf_plane, copy=True))
    cmfnodata = (cmf == nodatav) | np.isnan(cmf)
    cmfmask = ~cmfnodata & (cmf > 0)
    cmf[~cmfmask] = np.nan
    colnum = np.count_nonzero(cmfmask, axis=0)
    with np.errstate(all="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        colavg = np.nanmedian(cmf, axis=0)
        colstd = np.nanmedian(np.abs(cmf - colavg), axis=0)
        colmin = np.nanpercentile(cmf, axis=0, q=(1 - p) * 100, method="nearest")
        colmax = np.nanpercentile(cmf, axis=0, q=p * 100, method="nearest")
    return np.stack([colnum, colavg, colstd, colmin, colmax]).astype(np.float64)
