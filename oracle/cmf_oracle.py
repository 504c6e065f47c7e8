"""TEST INFRASTRUCTURE -- CPU oracle for the columnwise robust matched filter.

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  ``srcfinder_amd`` never does.

It restates, function-shaped, the algorithm that the reference holds inline in
``cmf/robust_mf.py`` (``__main__`` body, lines 185-397, and ``looshrinkage`` 92-136), with the
same float64 arithmetic and the same numpy/scipy calls in the same order, so that on the same
numpy/scipy build it reproduces the reference bit for bit.  Parity pin: the committed golden
vectors under ``tests/golden/`` were produced by executing the *real* reference in the
development container (``tests/golden/gen_golden.py``); ``tests/test_oracle_golden.py``
checks this oracle against every one of them.

Reference semantics restated here (all ``cmf/robust_mf.py``):
  * active window from gas/units, 1-based inclusive            :185-194
  * alpha grid 10**arange(-10, 0+0.05, 0.05) (201 points)      :241-244
  * valid rows: all(~(x<0) & isfinite(x)) over ACTIVE bands    :282, :299
  * float64 promotion of the valid rows                        :301
  * mean, centring, LOO shrinkage, inverse, matched filter     :347-381
  * outputs: float64 BIP [lines, samples, (R,G,B,CMF)]         :212-228, :383-397
  * bgmeta int16 [lines, samples, (cluster, alpha index)]      :268-279, :365
  * column stats npix / mean / std                             :388-392
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla

PPM_SCALING = 100000.0          # cmf/robust_mf.py:38
STABILITY_SCALING = 100.0       # cmf/robust_mf.py:94

ACTIVE_WINDOWS = {              # cmf/robust_mf.py:186-191
    ("ch4", False): (351, 422),
    ("ch4", True): (5, 420),
    ("co2", False): (309, 391),
    ("co2", True): (309, 391),
}


def alpha_grid() -> np.ndarray:
    """cmf/robust_mf.py:242-243 -- 201 values, the last one is 1.0000000000003273."""
    astep, aminexp, amaxexp = 0.05, -10.0, 0.0
    return 10.0 ** np.arange(aminexp, amaxexp + astep, astep)


def cov(a: np.ndarray, **kw) -> np.ndarray:
    """Rows are samples, ddof defaults to 1 (cmf/robust_mf.py:52-70)."""
    kw.setdefault("ddof", 1)
    return np.cov(a.T, **kw)


def inv(a, **kw):
    """cmf/robust_mf.py:72-76."""
    kw.setdefault("overwrite_a", False)
    kw.setdefault("check_finite", False)
    return sla.inv(a, **kw)


def det(a, **kw):
    """cmf/robust_mf.py:86-90."""
    kw.setdefault("overwrite_a", False)
    kw.setdefault("check_finite", False)
    return sla.det(a, **kw)


def eig(a, **kw):
    """cmf/robust_mf.py:78-84."""
    kw.setdefault("overwrite_a", False)
    kw.setdefault("check_finite", False)
    kw.setdefault("left", False)
    kw.setdefault("right", True)
    return sla.eig(a, **kw)


def looshrinkage(i_zm: np.ndarray, alphas: np.ndarray, nll: np.ndarray, n, i_reg=()):
    """Theiler leave-one-out shrinkage, faithful form (cmf/robust_mf.py:92-136).

    ``nll`` is filled in place.  Returns ``(C, mindex)``; ``mindex == -1`` (and alpha = 0)
    when every candidate's NLL is +inf.
    """
    p = i_zm.shape[1]
    x = i_zm * STABILITY_SCALING
    s = cov(x)
    t = np.diag(np.diag(s)) if len(i_reg) == 0 else cov(np.asarray(i_reg) * STABILITY_SCALING)
    plog2pi = p * np.log(2.0 * np.pi)
    nll[:] = np.inf
    for i, alpha in enumerate(alphas):
        try:
            beta = (1.0 - alpha) / (n - 1.0)
            g = n * (beta * s) + (alpha * t)
            gdet = det(g)
            if gdet == 0:
                continue
            r = (x.dot(inv(g)) * x).sum(axis=1)
            q = 1.0 - beta * r
            with np.errstate(all="ignore"):
                nll[i] = 0.5 * (plog2pi + np.log(gdet)) + 1.0 / (2.0 * n) * (np.log(q) + (r / q)).sum()
        except sla.LinAlgError:
            pass
    mindex = int(np.argmin(nll))
    if nll[mindex] != np.inf:
        alpha = alphas[mindex]
    else:
        mindex, alpha = -1, 0.0
    s = cov(i_zm)
    t = np.diag(np.diag(s)) if len(i_reg) == 0 else cov(np.asarray(i_reg))
    c = (1.0 - alpha) * s + alpha * t
    return c, mindex


def looshrinkage_eig(i_zm: np.ndarray, alphas: np.ndarray, nll: np.ndarray, n):
    """The same NLL through ONE symmetric eigendecomposition.

    With T = diag(S), d = sqrt(diag S), R = S/(d d^T) = V L V^T and Y = (X/d) V:
    r_k(alpha) = sum_j y_kj^2 / (n beta l_j + alpha),
    log det G = 2 sum log d_j + sum_j log(n beta l_j + alpha).
    This is the algorithm the HIP kernels implement; the oracle keeps it to show (tests) that it
    selects the same alpha index as the faithful form.  det over/underflow is emulated from
    the total log-determinant (rule (i) of SURVEY.md §7.3 item 3).
    """
    p = i_zm.shape[1]
    x = i_zm * STABILITY_SCALING
    s = cov(x)
    d2 = np.diag(s).copy()
    nll[:] = np.inf
    if np.any(d2 <= 0) or not np.all(np.isfinite(d2)):
        mindex, alpha = -1, 0.0
    else:
        d = np.sqrt(d2)
        rmat = s / np.outer(d, d)
        lam, v = np.linalg.eigh(rmat)
        y2 = ((x / d) @ v) ** 2
        logd2 = 2.0 * np.log(d).sum()
        for i, alpha in enumerate(alphas):
            beta = (1.0 - alpha) / (n - 1.0)
            den = n * beta * lam + alpha
            with np.errstate(all="ignore"):
                logdet = logd2 + np.log(den).sum()
                if not (logdet > np.log(5e-324)):       # det underflowed to 0 -> skipped (:112-113)
                    continue
                if logdet >= np.log(np.finfo(np.float64).max):
                    continue                            # det == inf -> nll = inf
                r = y2 @ (1.0 / den)
                q = 1.0 - beta * r
                nll[i] = 0.5 * (p * np.log(2.0 * np.pi) + logdet) + 1.0 / (2.0 * n) * (np.log(q) + r / q).sum()
        mindex = int(np.argmin(nll))
        if nll[mindex] != np.inf:
            alpha = alphas[mindex]
        else:
            mindex, alpha = -1, 0.0
    s = cov(i_zm)
    c = (1.0 - alpha) * s + alpha * np.diag(np.diag(s))
    return c, mindex


def useidx(icol: np.ndarray) -> np.ndarray:
    """cmf/robust_mf.py:282."""
    return np.where(((~(icol < 0)) & np.isfinite(icol)).all(axis=1))[0]


def active_window(gas: str = "ch4", reflectance: bool = False):
    try:
        return ACTIVE_WINDOWS[(gas, bool(reflectance))]
    except KeyError:
        raise ValueError("could not set active range for gas %r" % (gas,))


def robust_mf_oracle(cube_bil: np.ndarray, library: np.ndarray, *, gas="ch4", reflectance=False,
                     rgb_bands=(60, 42, 24), nodata=-9999.0, active=None, columns=None,
                     shrinkage=looshrinkage, return_nll=False, model="looshrinkage"):
    """Unimodal (k=1) column loop of cmf/robust_mf.py:297-397 on an in-memory BIL cube.

    cube_bil : [lines, bands, samples] (any float dtype; the reference reads float32)
    library  : [bands, 3] table or the [bands] third column
    columns  : optional iterable of sample indices to process (others are left untouched:
               CMF band = nodata, RGB = 0) -- used by the bounded CPU-baseline timing.
    Returns dict(out[lines,samples,4] f64, bgmeta[lines,samples,2] i16, colstats[3,samples],
                 alphaidx[samples] int, nuse[samples] int, status[samples] int[, nll[samples,201]]).
    status: 0 ok, 1 = no valid rows (column skipped, :303-304), 2 = singular C (:371-374).
    """
    if nodata > 0:
        raise Exception("nodata value=%f > 0, values will not be masked" % nodata)   # :232-234
    lines, nbands, samples = cube_bil.shape
    lib = np.float64(np.asarray(library))
    abscf_full = lib[:, 2] if lib.ndim == 2 else lib
    a0, a1 = active if active is not None else active_window(gas, reflectance)
    abscf = abscf_full[a0 - 1:a1]
    alphas = alpha_grid()
    nll = np.zeros(len(alphas))
    nrgb = len(rgb_bands)
    if nrgb not in (0, 3):
        raise Exception("invalid value of rgb_bands argument: %s" % (rgb_bands,))     # :225-226
    out = np.zeros((lines, samples, 4 if nrgb == 3 else 1), np.float64)
    out[:, :, -1] = nodata                                                            # :266
    bgmeta = np.zeros((lines, samples, 2), np.int16)
    colstats = np.ones((3, samples)) * nodata                                         # :293-295
    alphaidx = np.full(samples, -2, np.int64)
    nuse_all = np.zeros(samples, np.int64)
    status = np.zeros(samples, np.int32)
    nll_all = np.full((samples, len(alphas)), np.inf) if return_nll else None
    for col in (range(samples) if columns is None else columns):
        icol_full = cube_bil[:, a0 - 1:a1, col]
        use = useidx(icol_full)
        icol = np.float64(icol_full[use, :].copy())
        nuse = icol.shape[0]
        nuse_all[col] = nuse
        if nuse == 0:
            status[col] = 1
            continue
        mu = np.mean(icol, axis=0)
        try:
            if model == "empirical":                       # -M empirical: the sample covariance itself (:350-351, :366-367)
                cinv = inv(cov(icol - mu))
            else:
                c, aidx = shrinkage(icol - mu, alphas, nll, nuse)
                alphaidx[col] = aidx
                if return_nll:
                    nll_all[col] = nll
                cinv = inv(c)
                bgmeta[use, col, 1] = aidx
        except sla.LinAlgError:
            out[use, col, -1] = 0
            status[col] = 2
            cinv = None
        if cinv is not None:
            xc = icol - mu
            target = abscf.copy()
            target = target - mu if reflectance else target * mu
            normalizer = target.dot(cinv).dot(target.T)
            mf = (xc.dot(cinv).dot(target.T)) / normalizer
            out[use, col, -1] = mf if reflectance else mf * PPM_SCALING
        colpix = out[use, col, -1]
        colstats[0, col] = nuse
        colstats[1, col] = np.mean(colpix)
        colstats[2, col] = np.std(colpix)
        if nrgb == 3:
            for oi, bi in enumerate(rgb_bands):
                out[:, col, oi] = cube_bil[:, bi, col]
    res = dict(out=out, bgmeta=bgmeta, colstats=colstats, alphaidx=alphaidx, nuse=nuse_all, status=status)
    if return_nll:
        res["nll"] = nll_all
    return res


def robust_mf_multimodal_oracle(cube_bil: np.ndarray, library: np.ndarray, labels: np.ndarray, *, gas="ch4",
                                reflectance=False, rgb_bands=(60, 42, 24), nodata=-9999.0, active=None,
                                shrinkage=looshrinkage, reject=False, full=False, model="looshrinkage"):
    """Multimodal (k > 1) column loop of cmf/robust_mf.py:297-397 with the cluster labels INJECTED
    (labels[lines, samples], ids >= 0; what the reference's unseeded MiniBatchKMeans chose, :312-313, as stored
    in its bgmeta image, :327).  Per cluster of a column (:336-386): its own mean, looshrinkage with n = the
    COLUMN's valid-row count (the reference passes `nuse`, :355-356, not the cluster size), its own inverse and
    matched filter.
    reject (-r, :317-332): a cluster with fewer than int((a1-a0)*1.2) rows is relabelled -l (label 0 cannot be:
    -0 == 0); its rows are never scored, and in its turn of the loop the model of ALL non-rejected rows is fitted and
    written over those rows (:341); the column statistics cover the non-rejected rows (:388).
    full (-f, :354): the shrinkage target is the covariance of the whole column instead of diag(S).
    Returns dict(out, bgmeta, colstats, alphaidx[samples, k] (-2 = cluster absent), status[samples, k])."""
    lines, nbands, samples = cube_bil.shape
    lib = np.float64(np.asarray(library))
    abscf_full = lib[:, 2] if lib.ndim == 2 else lib
    a0, a1 = active if active is not None else active_window(gas, reflectance)
    abscf = abscf_full[a0 - 1:a1]
    bgminsamp = int((a1 - a0) * 1.2)                                               # :200
    alphas = alpha_grid()
    nll = np.zeros(len(alphas))
    nrgb = len(rgb_bands)
    k = int(labels.max()) + 1
    out = np.zeros((lines, samples, 4 if nrgb == 3 else 1), np.float64)
    out[:, :, -1] = nodata
    bgmeta = np.zeros((lines, samples, 2), np.int16)
    colstats = np.ones((3, samples)) * nodata
    alphaidx = np.full((samples, k), -2, np.int64)
    status = np.zeros((samples, k), np.int32)
    for col in range(samples):
        icol_full = cube_bil[:, a0 - 1:a1, col]
        use = useidx(icol_full)
        icol = np.float64(icol_full[use, :].copy())
        nuse = icol.shape[0]
        if nuse == 0:
            continue
        bglabels = np.int64(labels[use, col]).copy()
        bgulab = np.unique(bglabels)
        for i, l in enumerate(bgulab.copy()):
            lmask = bglabels == l
            if reject and lmask.sum() < bgminsamp:                                 # :321-324
                bglabels[lmask] = -l
                bgulab[i] = -l
            bgmeta[use[lmask], col, 0] = bgulab[i]                                 # :327
        if (bgulab < 0).all():                                                     # :330-332
            bglabels, bgulab = abs(bglabels), abs(bgulab)
        for ki in bgulab:
            kmask = (bglabels == ki) if ki >= 0 else (bglabels >= 0)               # :341
            icol_ki = icol[kmask, :].copy()
            mu = np.mean(icol_ki, axis=0)
            icol_reg = icol - mu if full else []                                   # :354
            try:
                if model == "empirical":                                           # :350-351, :366-367
                    cinv = inv(cov(icol_ki - mu))
                    alphaidx[col, abs(ki)] = -1
                else:
                    # n = nuse of the column (:355)
                    c, aidx = shrinkage(icol_ki - mu, alphas, nll, nuse, icol_reg) if full else shrinkage(icol_ki - mu, alphas, nll, nuse)
                    alphaidx[col, abs(ki)] = aidx
                    cinv = inv(c)
                    bgmeta[use[kmask], col, 1] = aidx
            except sla.LinAlgError:
                out[use[kmask], col, -1] = 0
                status[col, abs(ki)] = 2
                continue
            xc = icol_ki - mu
            target = abscf.copy()
            target = target - mu if reflectance else target * mu
            normalizer = target.dot(cinv).dot(target.T)
            mf = (xc.dot(cinv).dot(target.T)) / normalizer
            out[use[kmask], col, -1] = mf if reflectance else mf * PPM_SCALING
        colpix = out[use[bglabels >= 0], col, -1]                                  # :388
        colstats[0, col] = nuse
        colstats[1, col] = np.mean(colpix)
        colstats[2, col] = np.std(colpix)
        if nrgb == 3:
            for oi, bi in enumerate(rgb_bands):
                out[:, col, oi] = cube_bil[:, bi, col]
    return dict(out=out, bgmeta=bgmeta, colstats=colstats, alphaidx=alphaidx, status=status)
