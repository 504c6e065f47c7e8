"""Column profile and columnwise-systematics flags of a CMF product.

Mirrors ``triage/cmf_profile.py`` ``summarize`` (:90-140, default non-robust statistics) and the rolling-median
test that follows it (:182-205; same rule in ``triage/COVID/COVID_systematics_ID_Deliver.py:37-38``): a column is
flagged when its mean exceeds the centred 3-column rolling median by more than k x the MAD of the column means.
The per-pixel reduction runs in a HIP kernel on the product while it is still in HBM; the 598-element flag
computation is host numpy.
"""
from __future__ import annotations

import numpy as np

from . import _ffi

STATCOLS = ["npix", "avg", "std", "min", "max"]          # triage/cmf_profile.py:97
ROBUST_STATCOLS = ["npix", "med", "mad", "p05", "p95"]    # triage/cmf_profile.py:95


def column_profile(out, nodata=-9999.0, band=-1, to_numpy=True, robust=False, p=0.95):
    """out: [lines, samples, nb] float64 product (device tensor or ndarray) -> profile[5, samples]
    (npix, avg, std, min, max over valid positive CMF pixels; NaN where a column has none), or with ``robust``
    (npix, median, MAD, (1-p) and p percentiles, 'nearest') as ``use_robust_stats`` selects (:124-127)."""
    import torch
    if not torch.cuda.is_available():
        raise _ffi.SrcfinderError("no GPU visible: srcfinder_amd has no CPU fallback")
    t = out if torch.is_tensor(out) else torch.as_tensor(np.ascontiguousarray(out, dtype=np.float64))
    t = t.contiguous() if t.is_cuda else t.cuda().contiguous()
    if t.dim() == 2:
        t = t[..., None]
    L_, S_, nb = t.shape
    b = band % nb
    prof = torch.empty((5, S_), dtype=torch.float64, device=t.device)
    if robust:
        with torch.cuda.device(t.device):
            _ffi.check(_ffi.lib().sf_cmf_column_profile_robust(_ffi.ptr(t), L_, S_, nb, b, float(nodata), float(p),
                                                               _ffi.ptr(prof), _ffi.stream_ptr()),
                       "sf_cmf_column_profile_robust")
        return prof.cpu().numpy() if to_numpy else prof
    scratch = torch.empty(((L_ + 255) // 256) * S_ * 5, dtype=torch.float64, device=t.device)
    with torch.cuda.device(t.device):
        _ffi.check(_ffi.lib().sf_cmf_column_profile(_ffi.ptr(t), L_, S_, nb, b, float(nodata), _ffi.ptr(prof),
                                                    _ffi.ptr(scratch), _ffi.stream_ptr()), "sf_cmf_column_profile")
    return prof.cpu().numpy() if to_numpy else prof


def systematics_flags(colavg, win=3, nsigma=(1, 2, 3)):
    """(coldiff, colsigma, counts): centred rolling median of the column means (edges = median of the first/last
    `win` values), colsigma = MAD of the finite means, counts[k] = #columns with coldiff > k * colsigma."""
    colavg = np.asarray(colavg, dtype=np.float64)
    n = len(colavg)
    rwin = np.full(n, np.nan)
    h = win // 2
    for i in range(h, n - h):
        w = colavg[i - h:i + h + 1]
        rwin[i] = np.median(w) if np.all(np.isfinite(w)) else np.nan      # pandas rolling: NaN in window -> NaN
    rwin[0] = np.nanmedian(colavg[:win])
    rwin[-1] = np.nanmedian(colavg[-win:])
    coldiff = colavg - rwin
    fin = colavg[colavg == colavg]
    colsigma = np.median(np.abs(fin - np.median(fin)))
    with np.errstate(invalid="ignore"):
        counts = [int(np.count_nonzero(coldiff > k * colsigma)) for k in nsigma]
    return coldiff, colsigma, counts


def write_column_stats_csv(path, profile, robust=False):
    """``<product>_column_stats.csv`` as triage/cmf_profile.py:136-139 writes it (one row per column)."""
    with open(path, "w") as f:
        f.write(",".join(ROBUST_STATCOLS if robust else STATCOLS) + "\n")
        for i in range(profile.shape[1]):
            f.write(",".join(repr(float(v)) for v in profile[:, i]) + "\n")
