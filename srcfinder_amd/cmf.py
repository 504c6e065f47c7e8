"""Columnwise robust matched filter on MI355X -- host side of the drop-in boundary.

Mirrors the interface of the reference's ``cmf/robust_mf.py``:

* ``looshrinkage(I_zm, alphas, nll, n, I_reg=[]) -> (C, mindex)``   (robust_mf.py:92-136; fills ``nll``)
* ``cov(A, **kw)``, ``inv``, ``det``, ``eig``                        (robust_mf.py:52-90)
* ``robust_mf(cube_bil, library, ...)`` -- the body of the reference's ``__main__`` column loop
  (robust_mf.py:185-397), which has no callable form upstream; keyword names follow the CLI flags
  (robust_mf.py:142-166).
* ``alpha_grid()``, ``active_window()``, ``model_parameters()`` -- the constants the script derives
  (robust_mf.py:185-194, :241-259).

Everything numerical runs in hand-written HIP kernels behind ``libsrcfinder_amd.so``
(``include/srcfinder_amd.h``); PyTorch only owns device memory and the stream.  There is no CPU path:
calling these without the library or without a GPU raises.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _ffi

PPM_SCALING = 100000.0
NODATA_DEFAULT = -9999.0

_ACTIVE = {("ch4", False): (351, 422), ("ch4", True): (5, 420),
           ("co2", False): (309, 391), ("co2", True): (309, 391)}


def active_window(gas="ch4", reflectance=False):
    """1-based inclusive active channel window (robust_mf.py:185-194).

    The reference picks it from the library *file name*; unknown gases make it print and exit --
    here that is a ``ValueError``."""
    key = (str(gas).lower(), bool(reflectance))
    if key not in _ACTIVE:
        raise ValueError("could not set active range for gas %r" % (gas,))
    return _ACTIVE[key]


def gas_from_library_name(path):
    name = str(path)
    if "ch4" in name:
        return "ch4"
    if "co2" in name:
        return "co2"
    raise ValueError("could not set active range")


def alpha_grid():
    """robust_mf.py:242-243 -- evaluated with the same numpy expression (201 values)."""
    astep, aminexp, amaxexp = 0.05, -10.0, 0.0
    return 10.0 ** np.arange(aminexp, amaxexp + astep, astep)


def model_parameters(reflectance=False, active=(351, 422), modelname="looshrinkage", bgmodes=1, pcadim=6, reject=False,
                     regfull=False):
    """The ``model parameters`` header string the reference writes (robust_mf.py:246-259)."""
    bgmodel = "unimodal" if bgmodes == 1 else "multimodal"
    s = "modelname=%s, bgmodel=%s" % (modelname, bgmodel)
    if bgmodes > 1:
        s += ", bgmodes=%d, pcadim=%d, reject=%s" % (bgmodes, pcadim, bool(reject))
        if modelname == "looshrinkage":
            s += ", regfull=%s" % bool(regfull)
    if modelname == "looshrinkage":
        s += ", aminexp=-10.0, amaxexp=0.0, astep=0.05"
    s += ", reflectance=%s, active_bands=[%d, %d]" % (bool(reflectance), active[0], active[1])
    return "{ %s }" % s


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise _ffi.SrcfinderError("no GPU visible: srcfinder_amd has no CPU fallback")
    return torch


@dataclass
class CMFResult:
    """Outputs of :func:`robust_mf` (device tensors unless ``to_numpy`` was requested).

    out      [lines, samples, 4] float64 BIP, bands (R, G, B, CMF ppm*m); NODATA where a row is invalid
             ([lines, samples, 1] when ``rgb_bands=()``)                       robust_mf.py:212-228,:383-397
    bgmeta   [lines, samples, 2] int16 (cluster id = 0, alpha index) or None    robust_mf.py:268-279,:365
    colstats [3, samples] float64 = npix, mean, std of the written scores       robust_mf.py:388-392
    alphaidx [samples] int32 (-1: every NLL was inf; untouched columns -2)
    nuse     [samples] int32 valid rows per column
    status   [samples] int32: 0 ok, 1 no valid rows, 2 singular covariance
    nll      [samples, 201] float64 or None ([samples, k, 201] per cluster when kmeans > 1)
    """
    out: object
    bgmeta: object
    colstats: object
    alphaidx: object
    nuse: object
    status: object
    nll: object = None
    modelparms: str = ""
    labels: object = None      # multimodal only: [lines, samples] uint8 cluster of every row (255 = invalid row);
                               # alphaidx / status are then [samples, k] (status 1 = cluster absent in the column)


class _Workspace:
    """One growable device scratch buffer per (device, stream) -- the C ABI never allocates.  Flightlines in flight on
    different streams (inflight.py) must not share scratch."""
    _bufs = {}

    @classmethod
    def get(cls, nbytes, device):
        torch = _torch()
        key = (str(device), int(torch.cuda.current_stream(device).cuda_stream))
        buf = cls._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            cls._bufs[key] = None
            buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            cls._bufs[key] = buf
        return buf


_CONSTS = {}


def _device_const(arr, dev):
    """Small read-only host table -> device tensor, uploaded ONCE per (device, content).  A pageable host-to-device
    copy blocks the host until everything queued before it on the stream has finished; done per call it would put the
    host in lockstep with the GPU and nothing could be pipelined (tools/host_overhead_probe.py: 2.25 -> 0.1 ms
    of host time per robust_mf call)."""
    torch = _torch()
    arr = np.ascontiguousarray(arr)
    key = (str(dev), arr.dtype.str, arr.tobytes())
    t = _CONSTS.get(key)
    if t is None:
        if len(_CONSTS) > 64:
            # other streams' kernels (flightlines in flight, inflight.py) may still read cached tables: nothing is
            # released before the whole device has drained
            torch.cuda.synchronize(dev)
            _CONSTS.clear()
        t = torch.as_tensor(arr, device=dev)
        torch.cuda.current_stream(dev).synchronize()      # usable from any stream from now on
        _CONSTS[key] = t
    return t


def _abscf_from_library(library, a0, a1):
    lib = np.asarray(library, dtype=np.float64)
    col = lib[:, 2] if lib.ndim == 2 else lib
    return np.ascontiguousarray(col[a0 - 1:a1])


def robust_mf(cube_bil, library, *, gas="ch4", reflectance=False, kmeans=1, pcadim=6, reject=False, full=False,
              model="looshrinkage", rgb_bands=(60, 42, 24), nodata=NODATA_DEFAULT, active=None, metadata=False,
              columns=None, out=None, out_column0=0, return_nll=False, to_numpy=False, labels=None, kmeans_seed=0,
              kmeans_iters=30):
    """Columnwise matched filter of a BIL radiance cube (unimodal, or multimodal with ``kmeans`` > 1).

    cube_bil : [lines, bands, samples] float32, torch tensor on the GPU (preferred: stays resident) or
               ndarray (copied once).
    library  : [bands, 3] target table (column 3 = unit absorption) or its third column.
    columns  : optional (s0, s1) shard of samples to process (multi-GPU sharding); outputs then cover only
               those columns unless ``out`` (a preallocated [lines, S, nb] float64 tensor) and
               ``out_column0`` say where to put them.
    kmeans   : number of background modes per column (robust_mf.py:306-332).  The rows of a column are clustered
               on the device (``kmeans_seed``: the labels are a pure function of data, k and seed -- the reference's
               MiniBatchKMeans is unseeded) unless ``labels`` ([lines, samples] ints >= 0, e.g. the cluster band of
               a reference bgmeta image) injects them; ``result.labels`` holds the labels used.
    """
    torch = _torch()
    kmeans = int(kmeans)
    if kmeans < 1 or kmeans > 8:
        raise ValueError("kmeans must be in 1..8")
    if model not in ("looshrinkage", "empirical"):
        raise ValueError("model must be 'looshrinkage' or 'empirical'")
    if model == "empirical":
        if metadata:
            raise NameError("name 'alphas' is not defined")       # what the reference does with -M empirical -m (:241-244 vs :275)
    if nodata > 0:
        raise Exception("nodata value=%f > 0, values will not be masked" % nodata)       # robust_mf.py:232-234
    rgb_bands = tuple(int(b) for b in rgb_bands)
    if len(rgb_bands) not in (0, 3):
        raise Exception("invalid value of rgb_bands argument: %s" % (rgb_bands,))         # robust_mf.py:225-226
    a0, a1 = active if active is not None else active_window(gas, reflectance)
    p = a1 - a0 + 1
    # A host cube (ndarray / memmap / CPU tensor) is never copied whole: only the active window and the RGB bands are
    # staged into a compact device cube through pinned buffers (ingest.py; what the reference's memmap slicing does,
    # robust_mf.py:206-208, :298, :395-397).  A resident CUDA tensor is used where it lies.
    from . import ingest
    compact = None
    if isinstance(cube_bil, ingest.CompactCube):
        compact = cube_bil
        if compact.active != (int(a0), int(a1)) and active is not None:
            raise ValueError("compact cube holds window %s, not %s" % (compact.active, (a0, a1)))
        if len(rgb_bands) == 3 and compact.rgb_bands != rgb_bands:
            raise ValueError("compact cube holds RGB bands %s, not %s" % (compact.rgb_bands, rgb_bands))
        a0, a1 = compact.active
        p = a1 - a0 + 1
    elif not (torch.is_tensor(cube_bil) and cube_bil.is_cuda):
        host = cube_bil.numpy() if torch.is_tensor(cube_bil) else cube_bil
        if getattr(host, "ndim", 0) != 3:
            raise TypeError("cube must be float32 [lines, bands, samples]")
        # (a column shard of a host cube: only those columns are staged, and they become columns 0 .. of the compact cube)
        compact = ingest.stage_cube(host, (a0, a1), rgb_bands, columns=columns)
        if columns is not None:
            if labels is not None:                 # injected labels are full width: keep the shard's columns
                labels = labels[:, int(columns[0]):int(columns[1])]
            columns = (0, int(columns[1]) - int(columns[0]))
    if compact is not None:
        cube_bil = compact.tensor
        b0 = 0                                         # the window starts at band 0 of the compact cube
        rgb_dev = compact.compact_rgb if len(rgb_bands) == 3 else (0, 0, 0)
    else:
        b0 = a0 - 1
        rgb_dev = rgb_bands if len(rgb_bands) == 3 else (0, 0, 0)
    if cube_bil.dtype != torch.float32 or cube_bil.dim() != 3:
        raise TypeError("cube must be float32 [lines, bands, samples]")
    cube_bil = cube_bil.contiguous()
    dev = cube_bil.device
    lines, bands, samples = cube_bil.shape
    s0, s1 = (0, samples) if columns is None else (int(columns[0]), int(columns[1]))
    ncols = s1 - s0
    abscf = _device_const(_abscf_from_library(library, a0, a1), dev)
    alphas_np = alpha_grid()
    alphas = _device_const(alphas_np, dev)
    nalpha = len(alphas_np)
    nb = 4 if len(rgb_bands) == 3 else 1
    with torch.cuda.device(dev):
        if out is None:
            out_t = torch.empty((lines, ncols, nb), dtype=torch.float64, device=dev)
            out_samples, out_s0 = ncols, 0
        else:
            out_t = out
            if out_t.dtype != torch.float64 or tuple(out_t.shape[::2]) != (lines, nb) or not out_t.is_contiguous():
                raise TypeError("out must be contiguous float64 [lines, S, %d]" % nb)
            out_samples, out_s0 = out_t.shape[1], int(out_column0)
        alphaidx = torch.empty(ncols, dtype=torch.int32, device=dev)
        nuse = torch.empty(ncols, dtype=torch.int32, device=dev)
        status = torch.empty(ncols, dtype=torch.int32, device=dev)
        colstats = torch.empty((3, ncols), dtype=torch.float64, device=dev)
        bgmeta = torch.empty((lines, out_samples, 2), dtype=torch.int16, device=dev) if metadata else None
        if metadata and out is not None and out_samples != ncols:
            bgmeta.zero_()
        nll = torch.empty((ncols, nalpha), dtype=torch.float64, device=dev) if return_nll else None
        L = _ffi.lib()
        if kmeans > 1:
            r = rgb_dev
            res = _multimodal(torch, L, cube_bil, lines, bands, samples, s0, s1, b0 + 1, p, abscf, alphas, nalpha,
                              bool(reflectance), r, float(nodata), out_t, out_samples, out_s0, nb, bgmeta, kmeans,
                              int(pcadim), labels, int(kmeans_seed), int(kmeans_iters), bool(reject), bool(full),
                              int((a1 - a0) * 1.2), bool(return_nll), model == "empirical")   # bgminsamp, robust_mf.py:200
            res.modelparms = model_parameters(reflectance, (a0, a1), modelname=model, bgmodes=kmeans, pcadim=int(pcadim),
                                              reject=reject, regfull=full)
            if to_numpy:
                for k in ("out", "bgmeta", "colstats", "alphaidx", "nuse", "status", "labels", "nll"):
                    v = getattr(res, k)
                    if v is not None:
                        setattr(res, k, v.cpu().numpy())
            return res
        wsb = L.sf_cmf_workspace_bytes(lines, p, ncols, nalpha)
        ws = _Workspace.get(wsb, dev)
        r = rgb_dev
        if model == "empirical":
            _empirical(torch, L, cube_bil, lines, bands, samples, s0, s1, b0 + 1, p, abscf, alphas, bool(reflectance), r,
                       float(nodata), out_t, out_samples, out_s0, nb, alphaidx, nuse, status, colstats, ws)
            res = CMFResult(out=out_t, bgmeta=None, colstats=colstats, alphaidx=alphaidx, nuse=nuse, status=status, nll=None,
                            modelparms=model_parameters(reflectance, (a0, a1), modelname="empirical"))
            if to_numpy:
                for k in ("out", "colstats", "alphaidx", "nuse", "status"):
                    setattr(res, k, getattr(res, k).cpu().numpy())
            return res
        rc = L.sf_cmf_run(_ffi.ptr(cube_bil), lines, bands, samples, s0, s1, b0, p, _ffi.ptr(abscf),
                          _ffi.ptr(alphas), nalpha, int(bool(reflectance)), r[0], r[1], r[2], float(nodata),
                          _ffi.ptr(out_t), out_samples, out_s0, nb, _ffi.ptr(alphaidx), _ffi.ptr(nuse),
                          _ffi.ptr(status), _ffi.ptr(colstats), _ffi.ptr(bgmeta), _ffi.ptr(nll), _ffi.ptr(ws),
                          C.c_size_t(ws.numel()), _ffi.stream_ptr())
        _ffi.check(rc, "sf_cmf_run")
    res = CMFResult(out=out_t, bgmeta=bgmeta, colstats=colstats, alphaidx=alphaidx, nuse=nuse, status=status, nll=nll,
                    modelparms=model_parameters(reflectance, (a0, a1)))
    if to_numpy:
        for k in ("out", "bgmeta", "colstats", "alphaidx", "nuse", "status", "nll"):
            v = getattr(res, k)
            if v is not None:
                setattr(res, k, v.cpu().numpy())
    return res


def _empirical(torch, L, cube, lines, bands, samples, s0, s1, a0, p, abscf, alphas, reflectance, rgb, nodata, out_t,
               out_samples, out_s0, nb, alphaidx, nuse, status, colstats, ws):
    """-M empirical (robust_mf.py:350-351, :366-367): C is the sample covariance itself -- the stage entry points without
    stage 5; alpha index -1 makes stage 6 use alpha = 0, i.e. C = S."""
    if p > 512:
        raise NotImplementedError("active window of more than 512 bands")
    dev = cube.device
    ncols = s1 - s0
    ps = (p + 3) // 4 * 4
    P, st, check = _ffi.ptr, _ffi.stream_ptr(), _ffi.check
    f64 = dict(dtype=torch.float64, device=dev)
    xt = torch.empty((ncols, lines, ps), dtype=torch.float32, device=dev)
    mask = torch.empty((ncols, lines), dtype=torch.uint8, device=dev)
    mu, d, lam = (torch.empty((ncols, p), **f64) for _ in range(3))
    S, evec = torch.empty((ncols, p, p), **f64), torch.empty((ncols, p, p), **f64)
    filt, bias = torch.empty((ncols, p), **f64), torch.empty(ncols, **f64)
    check(L.sf_cmf_extract_columns(P(cube), lines, bands, samples, s0, s1, a0 - 1, p, P(xt), P(mask), st), "sf_cmf_extract_columns")
    check(L.sf_cmf_column_mean(P(xt), 0, P(mask), lines, p, ncols, P(nuse), P(mu), P(ws), st), "sf_cmf_column_mean")
    if p > 96:      # wide window (-R: 5..420): covariance + eigendecomposition by the batched-GEMM path (its sweep is unused)
        nalpha = alphas.numel()
        nll = torch.empty((ncols, nalpha), **f64)
        check(L.sf_cmf_wide_stats(P(xt), 0, P(mask), P(nuse), P(nuse), P(mu), P(alphas), nalpha, lines, p, ncols, P(S), P(d),
                                  P(lam), P(evec), P(status), P(nll), P(alphaidx), P(ws), st), "sf_cmf_wide_stats")
    else:
        check(L.sf_cmf_covariance(P(xt), 0, P(mask), P(nuse), P(mu), lines, p, ncols, P(S), P(ws), st), "sf_cmf_covariance")
        check(L.sf_cmf_eigh(P(S), P(nuse), p, ncols, P(d), P(lam), P(evec), P(status), P(ws), st), "sf_cmf_eigh")
    alphaidx.fill_(-1)
    check(L.sf_cmf_filter(P(mu), P(d), P(lam), P(evec), P(alphas), P(alphaidx), P(abscf), int(reflectance), p, ncols,
                          P(status), P(filt), P(bias), st), "sf_cmf_filter")
    alphaidx.fill_(-1)                  # (a single-row column comes back with index 0: the empirical model has none)
    check(L.sf_cmf_score(P(cube), lines, bands, samples, s0, s1, a0 - 1, p, P(filt), P(bias), P(status), P(alphaidx),
                         P(nuse), rgb[0], rgb[1], rgb[2], nodata, P(out_t), out_samples, out_s0, nb, None, P(colstats),
                         P(ws), st), "sf_cmf_score")


def _multimodal(torch, L, cube, lines, bands, samples, s0, s1, a0, p, abscf, alphas, nalpha, reflectance, rgb, nodata,
                out_t, out_samples, out_s0, nb, bgmeta, k, pcadim, labels, seed, iters, reject=False, full=False,
                bgminsamp=85, return_nll=False, empirical=False):
    """Multimodal column loop (robust_mf.py:306-386): stage entry points of the C ABI, once per cluster with the row
    mask  valid & (label == ki);  stage 5 gets the COLUMN's valid-row count as n (:355-356).
    reject (-r, :317-341): clusters of fewer than bgminsamp rows (never label 0: -0 == 0, :323) are relabelled -l in the
    cluster band and never scored; in their turn of the loop the model of ALL non-rejected rows is fitted and written
    over those rows.  If every cluster of a column is rejected none is (:330-332).
    full (-f, :354): the shrinkage target of every cluster is the covariance of the WHOLE column (numpy.cov removes the
    mean again, so the cluster mean subtracted at :354 drops out) -> sf_cmf_eigh_general instead of sf_cmf_eigh."""
    wide = p > 96                       # reflectance / full-band windows: statistics through sf_cmf_wide_stats
    if p > 512:
        raise NotImplementedError("active window of more than 512 bands")
    if empirical:                       # -M empirical: C = the cluster's sample covariance (:350-351); -f has no effect on it
        full, return_nll = False, False
    dev = cube.device
    ncols = s1 - s0
    ps = (p + 3) // 4 * 4
    P, st = _ffi.ptr, _ffi.stream_ptr()
    f64 = dict(dtype=torch.float64, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    det_window = 8                      # exact det() semantics on narrow windows: grid points per range crossing (two rounds)
    ws = _Workspace.get(max(L.sf_cmf_workspace_bytes(lines, p, ncols, nalpha),
                            0 if wide else L.sf_cmf_exact_det_scratch_bytes(p, ncols, nalpha, det_window)), dev)
    xt = torch.empty((ncols, lines, ps), dtype=torch.float32, device=dev)
    mask = torch.empty((ncols, lines), dtype=torch.uint8, device=dev)
    nuse_col, nuse_k, status_k, aidx_k = (torch.empty(ncols, **i32) for _ in range(4))
    mu = torch.empty((ncols, p), **f64)
    S = torch.empty((ncols, p, p), **f64)
    d = torch.empty((ncols, p), **f64)
    lam = torch.empty((ncols, p), **f64)
    evec = torch.empty((ncols, p, p), **f64)
    nll = torch.empty((ncols, nalpha), **f64)
    filt = torch.zeros((ncols, p), **f64)
    bias = torch.zeros(ncols, **f64)
    colstats = torch.empty((3, ncols), **f64)
    check = _ffi.check

    target = r_tmp = l_tmp = None

    def stats(m, n_rows, n_loo, status, want_alpha=True):
        check(L.sf_cmf_column_mean(P(xt), 0, P(m), lines, p, ncols, P(n_rows), P(mu), P(ws), st), "sf_cmf_column_mean")
        if wide:        # covariance + eigendecomposition + sweep in one call (the alpha index is a by-product)
            if target is not None:
                check(L.sf_cmf_wide_stats_target(P(xt), 0, P(m), P(n_rows), P(n_loo), P(mu), P(alphas), nalpha, lines, p, ncols,
                                                 P(target), P(S), P(d), P(lam), P(evec), P(status), P(nll), P(aidx_k), P(ws),
                                                 st), "sf_cmf_wide_stats_target")
            else:
                check(L.sf_cmf_wide_stats(P(xt), 0, P(m), P(n_rows), P(n_loo), P(mu), P(alphas), nalpha, lines, p, ncols, P(S),
                                          P(d), P(lam), P(evec), P(status), P(nll), P(aidx_k), P(ws), st), "sf_cmf_wide_stats")
            return
        check(L.sf_cmf_covariance(P(xt), 0, P(m), P(n_rows), P(mu), lines, p, ncols, P(S), P(ws), st), "sf_cmf_covariance")
        if target is not None:
            check(L.sf_cmf_eigh_general(P(S), P(target), P(n_rows), p, ncols, P(r_tmp), P(l_tmp), P(d), P(lam), P(evec),
                                        P(status), P(ws), st), "sf_cmf_eigh_general")
        else:
            check(L.sf_cmf_eigh(P(S), P(n_rows), p, ncols, P(d), P(lam), P(evec), P(status), P(ws), st), "sf_cmf_eigh")
        if want_alpha:
            check(L.sf_cmf_loocv(P(xt), 0, P(m), P(n_loo), P(mu), P(d), P(lam), P(evec), P(status), P(alphas), nalpha,
                                 lines, p, ncols, P(nll), P(aidx_k), P(ws), st), "sf_cmf_loocv")
            # a cluster of fewer rows than bands puts det(G_alpha) at the edge of the float64 range for the small alphas:
            # the grid points next to a lost one are factorised for real (scipy's running pivot product, :111-113)
            check(L.sf_cmf_exact_det(P(S), P(target) if target is not None else None, P(n_loo), P(status), P(alphas), nalpha, p,
                                     ncols, det_window, P(nll), P(aidx_k), P(ws), st), "sf_cmf_exact_det")

    check(L.sf_cmf_extract_columns(P(cube), lines, bands, samples, s0, s1, a0 - 1, p, P(xt), P(mask), st), "sf_cmf_extract_columns")
    if labels is None:
        stats(mask, nuse_col, nuse_col, status_k, want_alpha=False)      # eigenbasis of the whole column: the PCA axes
        labels_t = torch.empty((ncols, lines), dtype=torch.uint8, device=dev)
        scratch = torch.empty(ncols * lines * pcadim, dtype=torch.float32, device=dev)
        check(L.sf_cmf_kmeans(P(xt), P(mask), P(mu), P(d), P(lam), P(evec), lines, p, ncols, k, pcadim, seed, iters,
                              P(labels_t), P(scratch), st), "sf_cmf_kmeans")
    else:
        lab = labels if torch.is_tensor(labels) else torch.as_tensor(np.ascontiguousarray(labels))
        if tuple(lab.shape) != (lines, samples):
            raise ValueError("labels must be [lines, samples]")
        if int(lab.min()) < 0 or int(lab.max()) >= k:
            raise ValueError("labels must be cluster ids in 0..%d" % (k - 1))
        labels_t = lab.to(dev)[:, s0:s1].t().contiguous().to(torch.uint8)
        check(L.sf_cmf_column_mean(P(xt), 0, P(mask), lines, p, ncols, P(nuse_col), P(mu), P(ws), st), "sf_cmf_column_mean")
        if full and wide:           # (the wide route has no covariance-only entry: the whole-column statistics once)
            stats(mask, nuse_col, nuse_col, status_k, want_alpha=False)
        elif full:
            check(L.sf_cmf_covariance(P(xt), 0, P(mask), P(nuse_col), P(mu), lines, p, ncols, P(S), P(ws), st), "sf_cmf_covariance")
    if full:                        # S holds the whole column's covariance at this point on both routes
        target = S.clone()
        if not wide:
            r_tmp, l_tmp = torch.empty_like(S), torch.empty_like(S)
    # ---- the product starts as: NODATA on invalid rows, 0 on valid ones, RGB copied (a zero filter through stage 7)
    status0 = (nuse_col == 0).to(torch.int32)                           # 1 = column without a valid row: skipped (:303-304)
    zero_idx = torch.zeros(ncols, **i32)
    check(L.sf_cmf_score(P(cube), lines, bands, samples, s0, s1, a0 - 1, p, P(filt), P(bias), P(status0), P(zero_idx),
                         P(nuse_col), rgb[0], rgb[1], rgb[2], nodata, P(out_t), out_samples, out_s0, nb, P(bgmeta), None,
                         P(ws), st), "sf_cmf_score")
    alphaidx = torch.full((ncols, k), -2, **i32)
    status = torch.ones((ncols, k), **i32)
    nll_all = torch.full((ncols, k, nalpha), float("inf"), **f64) if return_nll else None   # per cluster (absent: inf)
    labels_valid = torch.where(mask != 0, labels_t, torch.full_like(labels_t, 255))
    keep = None
    if reject:
        counts = torch.stack([(labels_valid == ki).sum(1) for ki in range(k)], 1)          # [ncols, k]
        present = counts > 0
        rej = present & (counts < bgminsamp)
        rej[:, 0] = False                                                                  # -0 == 0 (:323)
        lab_idx = labels_valid.clamp(max=k - 1).long()
        if bgmeta is not None:      # cluster band: -l for the rows of a rejected cluster (written before :330's abs())
            lab16 = labels_valid.to(torch.int16)
            meta0 = torch.where(mask != 0, torch.where(torch.gather(rej, 1, lab_idx), -lab16, lab16), torch.zeros_like(lab16))
            bgmeta[:, out_s0:out_s0 + ncols, 0] = meta0.t()
        rej &= ~((rej | ~present).all(1, keepdim=True))                                    # all rejected -> none (:330-332)
        keep = ((mask != 0) & ~torch.gather(rej, 1, lab_idx)).to(torch.uint8)
    for ki in range(k):
        mask_k = (labels_valid == ki).to(torch.uint8)
        if reject:
            mask_k = torch.where(rej[:, ki, None], keep, mask_k)
        stats(mask_k, nuse_k, nuse_col, status_k, want_alpha=not empirical)
        if empirical:
            aidx_k.fill_(-1)            # alpha index -1 makes stage 6 use alpha = 0: C = S (:366-367)
        check(L.sf_cmf_filter(P(mu), P(d), P(lam), P(evec), P(alphas), P(aidx_k), P(abscf), int(reflectance), p, ncols,
                              P(status_k), P(filt), P(bias), st), "sf_cmf_filter")
        if empirical:
            aidx_k.fill_(-1)            # (a single-row cluster comes back with index 0: the empirical model has none)
        check(L.sf_cmf_score_cluster(P(cube), lines, bands, samples, s0, s1, a0 - 1, p, P(filt), P(bias), P(status_k),
                                     P(aidx_k), P(mask_k), -32768 if reject else ki, P(out_t), out_samples, out_s0, nb,
                                     P(bgmeta), st),
              "sf_cmf_score_cluster")
        alphaidx[:, ki] = torch.where(status_k == 1, torch.full_like(aidx_k, -2), aidx_k)
        status[:, ki] = status_k
        if nll_all is not None:
            nll_all[:, ki] = torch.where((status_k == 0)[:, None], nll, nll_all[:, ki])
    check(L.sf_cmf_colstats_rows(P(out_t), out_samples, out_s0, nb, P(mask if keep is None else keep), lines, ncols, nodata,
                                 P(colstats), st), "sf_cmf_colstats_rows")
    if keep is not None:            # rows of a rejected cluster are never written: they stay NODATA (:266, :341)
        out_t[:, out_s0:out_s0 + ncols, nb - 1].masked_fill_(((mask != 0) & (keep == 0)).t(), nodata)
    if keep is not None:            # the count column stays the number of valid rows (:389), mean/std skip rejected rows (:388)
        colstats[0] = torch.where(nuse_col > 0, nuse_col.to(torch.float64), colstats[0])
    return CMFResult(out=out_t, bgmeta=bgmeta, colstats=colstats, alphaidx=alphaidx, nuse=nuse_col, status=status,
                     labels=labels_valid.t().contiguous(), nll=nll_all)


def sweep_routes(cube_bil, library=None, *, gas="ch4", reflectance=False, active=None, columns=None):
    """Which sweep kernel each column of a flightline takes (diagnostic; bench.py reports it beside the step time).

    The sweep of the 4x4x4 windows (69..72 bands: CH4; 81..84: CO2; 93..96) multiplies by the rank-28 or rank-36
    factorisation of its coefficient matrix when the column's eigenvalue spectrum allows it (cmf_lowrank.hip) and by the full
    matrix otherwise (at 72 bands: 814 / 954 / 1260 MFMAs per 16-row tile; "full" columns of the wider windows take the
    16x16x4 kernel) -- the reference pays one cost for any data (robust_mf.py:105-117).  Runs stages 1-4 through the C ABI and
    the factorisation's test hook; returns counts {"rank24", "rank28", "rank36", "full", "skipped"} (skipped: status != 0).
    Windows of 257..432 bands (full-band, -R): {"factored", "unfactored", "skipped", "ranks": {rank: columns}} -- the wide sweep's second
    product through the rank factorisation of cmf_wlr.hip, or unfactored (a spectrum spread densely over decades)."""
    torch = _torch()
    if not (torch.is_tensor(cube_bil) and cube_bil.is_cuda):
        raise TypeError("sweep_routes needs the resident cube")
    dev = cube_bil.device
    lines, bands, samples = cube_bil.shape
    a0, a1 = active if active is not None else active_window(gas, reflectance)
    p = a1 - a0 + 1
    nj = (p + 3) // 4
    wide = 256 < p <= 432             # k_wsweep8's windows: the r phase through the rank factorisation of cmf_wlr.hip, or plain
    if nj not in (18, 21, 24) and not wide:
        return {"note": "windows other than 69..72, 81..84, 93..96 and 257..432 bands take one route (p = %d)" % p}
    nje = nj + (nj & 1)
    s0, s1 = (0, samples) if columns is None else (int(columns[0]), int(columns[1]))
    ncols = s1 - s0
    L = _ffi.lib()
    P, st, check = _ffi.ptr, _ffi.stream_ptr(), _ffi.check
    alphas_np = alpha_grid()
    al = _device_const(alphas_np, dev)
    nalpha = len(alphas_np)
    with torch.cuda.device(dev):
        ws = _Workspace.get(L.sf_cmf_workspace_bytes(lines, p, ncols, nalpha), dev)
        f64 = dict(dtype=torch.float64, device=dev)
        ps = (p + 3) // 4 * 4
        xt = torch.empty((ncols, lines, ps), dtype=torch.float32, device=dev)
        mask = torch.empty((ncols, lines), dtype=torch.uint8, device=dev)
        nuse = torch.empty(ncols, dtype=torch.int32, device=dev)
        status = torch.empty(ncols, dtype=torch.int32, device=dev)
        mu, d, lam = (torch.empty((ncols, p), **f64) for _ in range(3))
        S, evec = torch.empty((ncols, p, p), **f64), torch.empty((ncols, p, p), **f64)
        check(L.sf_cmf_extract_columns(P(cube_bil.contiguous()), lines, bands, samples, s0, s1, a0 - 1, p, P(xt), P(mask), st),
              "sf_cmf_extract_columns")
        check(L.sf_cmf_column_mean(P(xt), 0, P(mask), lines, p, ncols, P(nuse), P(mu), P(ws), st), "sf_cmf_column_mean")
        if wide:        # stages 3-5 of the wide windows are one call; the eigenvalues it leaves are what the factorisation sees
            nll = torch.empty((ncols, nalpha), **f64)
            aidx = torch.empty(ncols, dtype=torch.int32, device=dev)
            check(L.sf_cmf_wide_stats(P(xt), 0, P(mask), P(nuse), P(nuse), P(mu), P(al), nalpha, lines, p, ncols, P(S), P(d), P(lam),
                                      P(evec), P(status), P(nll), P(aidx), P(ws), st), "sf_cmf_wide_stats")
        else:
            check(L.sf_cmf_covariance(P(xt), 0, P(mask), P(nuse), P(mu), lines, p, ncols, P(S), P(ws), st), "sf_cmf_covariance")
            check(L.sf_cmf_eigh(P(S), P(nuse), p, ncols, P(d), P(lam), P(evec), P(status), P(ws), st), "sf_cmf_eigh")
        if wide:
            scratch = torch.empty(int(L.sf_debug_wlr_bytes(ncols)), dtype=torch.uint8, device=dev)
            wlr = torch.empty(ncols, dtype=torch.int32, device=dev)
            check(L.sf_debug_wlr(P(lam), P(nuse), P(status), P(al), nalpha, p, ncols, P(scratch), P(wlr), st), "sf_debug_wlr")
            ok = status == 0
            ranks = {int(k): int(((wlr == k) & ok).sum()) for k in torch.unique(wlr[ok]).tolist() if k > 0}
            return {"factored": int(((wlr > 0) & ok).sum()), "unfactored": int(((wlr == 0) & ok).sum()),
                    "skipped": int((~ok).sum()), "ranks": ranks}
        ufrag = torch.empty((ncols, nje * 9 * 16), **f64)
        wfrag = torch.empty((ncols, 13 * 9 * 64), **f64)
        lrok = torch.empty(ncols, dtype=torch.int32, device=dev)
        check(L.sf_debug_lowrank(P(lam), P(nuse), P(status), P(al), nalpha, p, ncols, P(ufrag), P(wfrag), P(lrok), st),
              "sf_debug_lowrank")
        ok = status == 0
        counts = {"rank24": int(((lrok == 3) & ok).sum()), "rank28": int(((lrok == 1) & ok).sum()), "rank36": int(((lrok == 2) & ok).sum()),
                  "full": int(((lrok == 0) & ok).sum()), "skipped": int((~ok).sum())}
    return counts


# ------------------------------------------------------------------------------------------------------
# function-level entries with the reference's signatures
# ------------------------------------------------------------------------------------------------------
def _upload_rows(a):
    """[n, p] float64 host matrix -> device xt[1][n][ps] float64 (zero padded), all rows valid."""
    torch = _torch()
    a = np.ascontiguousarray(a, dtype=np.float64)
    n, p = a.shape
    ps = (p + 3) // 4 * 4
    buf = np.zeros((1, n, ps), np.float64)
    buf[0, :, :p] = a
    xt = torch.as_tensor(buf).cuda()
    mask = torch.ones((1, n), dtype=torch.uint8, device=xt.device)
    return xt, mask, n, p


def _wide_stats(torch, L, xt, mask, nrows, nloo, mu, alphas, rows, p, S, ws, nll=None, aidx=None, target=None):
    """sf_cmf_wide_stats on ONE float64 matrix (function-level entries with more than 96 bands)."""
    dev = xt.device
    f64 = dict(dtype=torch.float64, device=dev)
    nalpha = alphas.numel()
    d, lam = torch.empty((1, p), **f64), torch.empty((1, p), **f64)
    evec = torch.empty((1, p, p), **f64)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    nll = torch.empty((1, nalpha), **f64) if nll is None else nll
    aidx = torch.empty(1, dtype=torch.int32, device=dev) if aidx is None else aidx
    if target is not None:
        _ffi.check(L.sf_cmf_wide_stats_target(_ffi.ptr(xt), 1, _ffi.ptr(mask), _ffi.ptr(nrows), _ffi.ptr(nloo), _ffi.ptr(mu),
                                              _ffi.ptr(alphas), nalpha, rows, p, 1, _ffi.ptr(target), _ffi.ptr(S), _ffi.ptr(d),
                                              _ffi.ptr(lam), _ffi.ptr(evec), _ffi.ptr(status), _ffi.ptr(nll), _ffi.ptr(aidx),
                                              _ffi.ptr(ws), _ffi.stream_ptr()), "sf_cmf_wide_stats_target")
        return nll, aidx
    _ffi.check(L.sf_cmf_wide_stats(_ffi.ptr(xt), 1, _ffi.ptr(mask), _ffi.ptr(nrows), _ffi.ptr(nloo), _ffi.ptr(mu),
                                   _ffi.ptr(alphas), nalpha, rows, p, 1, _ffi.ptr(S), _ffi.ptr(d), _ffi.ptr(lam),
                                   _ffi.ptr(evec), _ffi.ptr(status), _ffi.ptr(nll), _ffi.ptr(aidx), _ffi.ptr(ws),
                                   _ffi.stream_ptr()), "sf_cmf_wide_stats")
    return nll, aidx


def cov(A, **kwargs):
    """Sample covariance with MATLAB semantics: rows are samples, ``ddof=1`` (robust_mf.py:52-70).

    Computed on the GPU (masked mean + centred fp64-MFMA SYRK).  Only ``ddof`` (0 or 1) is honoured."""
    torch = _torch()
    ddof = kwargs.pop("ddof", 1)
    if kwargs:
        raise TypeError("unsupported numpy.cov keyword(s): %s" % sorted(kwargs))
    if isinstance(A, torch.Tensor):
        A = A.detach().cpu().numpy()
    xt, mask, n, p = _upload_rows(A)
    if p > 512:
        raise NotImplementedError("cov(): more than 512 features")
    L = _ffi.lib()
    dev = xt.device
    nuse = torch.empty(1, dtype=torch.int32, device=dev)
    mu = torch.empty((1, p), dtype=torch.float64, device=dev)
    S = torch.empty((1, p, p), dtype=torch.float64, device=dev)
    st = _ffi.stream_ptr()
    if p > 96:      # wide window: the batched-GEMM path computes S on the way to its eigendecomposition
        ws = _Workspace.get(L.sf_cmf_workspace_bytes(n, p, 1, 1), dev)
        _ffi.check(L.sf_cmf_column_mean(_ffi.ptr(xt), 1, _ffi.ptr(mask), n, p, 1, _ffi.ptr(nuse), _ffi.ptr(mu),
                                        _ffi.ptr(ws), st), "sf_cmf_column_mean")
        _wide_stats(torch, L, xt, mask, nuse, None, mu, torch.ones(1, dtype=torch.float64, device=dev), n, p, S, ws)
    else:
        ws = _Workspace.get(L.sf_cmf_workspace_bytes(n, p, 1, 1), dev)
        _ffi.check(L.sf_cmf_column_mean(_ffi.ptr(xt), 1, _ffi.ptr(mask), n, p, 1, _ffi.ptr(nuse), _ffi.ptr(mu),
                                        _ffi.ptr(ws), st), "sf_cmf_column_mean")
        _ffi.check(L.sf_cmf_covariance(_ffi.ptr(xt), 1, _ffi.ptr(mask), _ffi.ptr(nuse), _ffi.ptr(mu), n, p, 1,
                                       _ffi.ptr(S), _ffi.ptr(ws), st), "sf_cmf_covariance")
    S = S[0].cpu().numpy()
    if ddof != 1:
        S = S * ((n - 1.0) / (n - float(ddof)))
    return S


def _square(A, what):
    torch = _torch()
    if isinstance(A, torch.Tensor):
        A = A.detach().cpu().numpy()
    a = np.ascontiguousarray(A, dtype=np.float64)
    if a.ndim != 2 or a.shape[0] != a.shape[1]:
        raise ValueError("expected square matrix")
    if a.shape[0] > 1024:
        raise NotImplementedError("%s(): more than 1024 rows" % what)
    return a


def _lapack_kwargs(kwargs, allowed):
    """The reference's wrappers forward **kwargs to scipy.linalg after setting overwrite_a / check_finite (:72-90);
    those two have no effect here (the input is copied to the device, nothing is checked), anything else is refused."""
    for k in list(kwargs):
        if k in ("overwrite_a", "check_finite") or k in allowed:
            continue
        raise TypeError("unsupported scipy.linalg keyword: %s" % k)


def inv(A, **kwargs):
    """``scipy.linalg.inv(A, overwrite_a=False, check_finite=False)`` (robust_mf.py:72-76) on the GPU: LU with partial
    pivoting + triangular solves in float64; an exactly singular matrix raises ``numpy.linalg.LinAlgError`` like LAPACK's
    ``info > 0`` (the column loop catches it and writes zeros, :371-374)."""
    torch = _torch()
    _lapack_kwargs(kwargs, ())
    a = _square(A, "inv")
    n = a.shape[0]
    L = _ffi.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    ad = torch.as_tensor(a, device=dev)
    work, out = torch.empty_like(ad), torch.empty_like(ad)
    piv = torch.empty(n, dtype=torch.int32, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    _ffi.check(L.sf_linalg_inv(_ffi.ptr(ad), n, 1, _ffi.ptr(work), _ffi.ptr(piv), _ffi.ptr(out), _ffi.ptr(info),
                               _ffi.stream_ptr()), "sf_linalg_inv")
    if int(info.item()) > 0:
        raise np.linalg.LinAlgError("singular matrix")
    return out.cpu().numpy()


def det(A, **kwargs):
    """``scipy.linalg.det(A, overwrite_a=False, check_finite=False)`` (robust_mf.py:86-90) on the GPU: the running
    product of the LU pivots in index order -- once a prefix has reached inf or 0 it stays there, which is what decides
    ``log(det)`` and the ``det == 0`` test of looshrinkage (:111-113); 0.0 for an exactly singular matrix."""
    torch = _torch()
    _lapack_kwargs(kwargs, ())
    a = _square(A, "det")
    n = a.shape[0]
    L = _ffi.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    ad = torch.as_tensor(a, device=dev)
    work = torch.empty_like(ad)
    out = torch.empty(1, dtype=torch.float64, device=dev)
    _ffi.check(L.sf_linalg_det(_ffi.ptr(ad), n, 1, _ffi.ptr(work), _ffi.ptr(out), _ffi.stream_ptr()), "sf_linalg_det")
    return np.float64(out.item())


def eig(A, **kwargs):
    """``scipy.linalg.eig(A, left=False, right=True)`` (robust_mf.py:78-84) for the one use the reference makes of it:
    the eigendecomposition of a covariance matrix (:312).  Returns ``(w, vr)`` with complex128 ``w`` (imaginary parts 0)
    and unit-norm eigenvectors in the columns of ``vr``; the ORDER of the pairs is the eigensolver's (LAPACK's geev does
    not sort either -- the reference takes "the first six").  Symmetric matrices of up to 512 rows
    (one-sided Jacobi on the GPU: in LDS up to 96 rows, the blocked wide eigensolver above -- the -R -k 2 run calls
    eig(cov(...)) at p = 416, :310); anything else is refused."""
    torch = _torch()
    _lapack_kwargs(kwargs, ("left", "right"))
    if kwargs.get("left", False) or not kwargs.get("right", True):
        raise NotImplementedError("eig(): only left=False, right=True")
    a = _square(A, "eig")
    n = a.shape[0]
    if n > 512:
        raise NotImplementedError("eig(): more than 512 rows")
    if not np.allclose(a, a.T, rtol=1e-12, atol=1e-12 * np.abs(a).max()):
        raise NotImplementedError("eig(): nonsymmetric matrix")
    a = (a + a.T) * 0.5
    # shift to positive definite: the solver factorises its input (the eigenvectors do not change)
    shift = 0.0
    gersh = float(np.min(np.diag(a) - (np.abs(a).sum(1) - np.abs(np.diag(a)))))
    if gersh <= 0:
        shift = -gersh + 1e-3 * max(np.abs(a).max(), 1e-300)
    L = _ffi.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    f64 = dict(dtype=torch.float64, device=dev)
    S = torch.as_tensor((a + shift * np.eye(n))[None], device=dev)
    d, lam = torch.empty((1, n), **f64), torch.empty((1, n), **f64)
    evec = torch.empty((1, n, n), **f64)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    if n > 96:
        ws = _Workspace.get(L.sf_cmf_eigh_wide_scratch_bytes(n, 1), dev)
        _ffi.check(L.sf_cmf_eigh_wide(_ffi.ptr(S), n, 1, _ffi.ptr(lam), _ffi.ptr(evec), _ffi.ptr(status), _ffi.ptr(ws),
                                      _ffi.stream_ptr()), "sf_cmf_eigh_wide")
    else:
        T = torch.eye(n, **f64)[None].contiguous()
        nrows = torch.tensor([n + 2], dtype=torch.int32, device=dev)
        r_tmp, l_tmp = torch.empty((1, n, n), **f64), torch.empty((1, n, n), **f64)
        ws = _Workspace.get(L.sf_cmf_workspace_bytes(n + 2, n, 1, 1), dev)
        _ffi.check(L.sf_cmf_eigh_general(_ffi.ptr(S), _ffi.ptr(T), _ffi.ptr(nrows), n, 1, _ffi.ptr(r_tmp), _ffi.ptr(l_tmp),
                                         _ffi.ptr(d), _ffi.ptr(lam), _ffi.ptr(evec), _ffi.ptr(status), _ffi.ptr(ws),
                                         _ffi.stream_ptr()), "sf_cmf_eigh_general")
    if int(status.item()) != 0:
        raise np.linalg.LinAlgError("eig(): the eigensolver did not converge (status %d)" % int(status.item()))
    w = lam[0].cpu().numpy() - shift
    v = evec[0].cpu().numpy().T.copy()                     # rows -> columns
    v /= np.linalg.norm(v, axis=0, keepdims=True)
    return w.astype(np.complex128), v


def looshrinkage(I_zm, alphas, nll, n, I_reg=[]):
    """Leave-one-out shrinkage covariance (Theiler 2012), same contract as robust_mf.py:92-136:

    ``I_zm`` [rows, p] zero-mean float64 samples, ``alphas`` the candidate grid, ``nll`` a float64 array of
    the same length that is FILLED IN PLACE, ``n`` the sample count used in beta and 1/(2n) (the reference
    passes the column's ``nuse`` even for a cluster subset).  Returns ``(C, mindex)`` with
    ``C = (1-alpha) S + alpha T`` on the unscaled data, ``T = diag(S)`` or, with a non-empty ``I_reg`` [rows', p],
    ``T = cov(I_reg)`` (:99, :131), and ``mindex = -1`` (alpha = 0) when every candidate's NLL is +inf.
    """
    torch = _torch()
    alphas_np = np.ascontiguousarray(alphas, dtype=np.float64)
    xt, mask, rows, p = _upload_rows(I_zm)
    if p > 512:
        raise NotImplementedError("looshrinkage(): more than 512 bands")
    L = _ffi.lib()
    dev = xt.device
    nalpha = len(alphas_np)
    ws = _Workspace.get(L.sf_cmf_workspace_bytes(rows, p, 1, nalpha), dev)
    f64 = dict(dtype=torch.float64, device=dev)
    nrows = torch.tensor([rows], dtype=torch.int32, device=dev)      # numpy.cov's own row count (ddof=1)
    nloo = torch.tensor([int(n)], dtype=torch.int32, device=dev)     # the n of beta and of 1/(2n)
    mu = torch.zeros((1, p), **f64)                                   # data is already centred
    S = torch.empty((1, p, p), **f64)
    d = torch.empty((1, p), **f64)
    lam = torch.empty((1, p), **f64)
    evec = torch.empty((1, p, p), **f64)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    nll_d = torch.empty((1, nalpha), **f64)
    aidx = torch.empty(1, dtype=torch.int32, device=dev)
    al = torch.as_tensor(alphas_np, device=dev)
    st = _ffi.stream_ptr()
    T_np = T_d = None
    if len(I_reg) != 0:
        T_np = cov(I_reg)
        if T_np.shape != (p, p):
            raise ValueError("I_reg must have the same number of columns as I_zm")
        T_d = torch.as_tensor(np.ascontiguousarray(T_np[None]), device=dev)
    if p > 96:      # wide window (e.g. the reference's own -R run, p = 416; full-band p = 425): batched-GEMM path
        _wide_stats(torch, L, xt, mask, nrows, nloo, mu, al, rows, p, S, ws, nll_d, aidx, target=T_d)
        nll[:] = nll_d[0].cpu().numpy()
        mindex = int(aidx.item())
        alpha = float(alphas_np[mindex]) if mindex >= 0 else 0.0
        S = S[0].cpu().numpy()
        return (1.0 - alpha) * S + alpha * (np.diag(np.diag(S)) if T_np is None else T_np), mindex
    _ffi.check(L.sf_cmf_covariance(_ffi.ptr(xt), 1, _ffi.ptr(mask), _ffi.ptr(nrows), _ffi.ptr(mu), rows, p, 1,
                                   _ffi.ptr(S), _ffi.ptr(ws), st), "sf_cmf_covariance")
    if T_d is not None:
        r_tmp, l_tmp = torch.empty_like(S), torch.empty_like(S)
        _ffi.check(L.sf_cmf_eigh_general(_ffi.ptr(S), _ffi.ptr(T_d), _ffi.ptr(nrows), p, 1, _ffi.ptr(r_tmp), _ffi.ptr(l_tmp),
                                         _ffi.ptr(d), _ffi.ptr(lam), _ffi.ptr(evec), _ffi.ptr(status), _ffi.ptr(ws), st),
                   "sf_cmf_eigh_general")
    else:
        _ffi.check(L.sf_cmf_eigh(_ffi.ptr(S), _ffi.ptr(nrows), p, 1, _ffi.ptr(d), _ffi.ptr(lam), _ffi.ptr(evec),
                                 _ffi.ptr(status), _ffi.ptr(ws), st), "sf_cmf_eigh")
    _ffi.check(L.sf_cmf_loocv(_ffi.ptr(xt), 1, _ffi.ptr(mask), _ffi.ptr(nloo), _ffi.ptr(mu), _ffi.ptr(d),
                              _ffi.ptr(lam), _ffi.ptr(evec), _ffi.ptr(status), _ffi.ptr(al), nalpha, rows, p, 1,
                              _ffi.ptr(nll_d), _ffi.ptr(aidx), _ffi.ptr(ws), st), "sf_cmf_loocv")
    # det() over/underflow as the running pivot product has it: every grid point factorised for real (201 small LUs)
    dws = _Workspace.get(max(L.sf_cmf_workspace_bytes(rows, p, 1, nalpha), L.sf_cmf_exact_det_scratch_bytes(p, 1, nalpha, 0)), dev)
    _ffi.check(L.sf_cmf_exact_det(_ffi.ptr(S), _ffi.ptr(T_d) if T_d is not None else None, _ffi.ptr(nloo), _ffi.ptr(status),
                                  _ffi.ptr(al), nalpha, p, 1, 0, _ffi.ptr(nll_d), _ffi.ptr(aidx), _ffi.ptr(dws), st),
               "sf_cmf_exact_det")
    nll[:] = nll_d[0].cpu().numpy()
    mindex = int(aidx.item())
    alpha = float(alphas_np[mindex]) if mindex >= 0 else 0.0
    S = S[0].cpu().numpy()
    T = np.diag(np.diag(S)) if T_np is None else T_np
    Cmat = (1.0 - alpha) * S + alpha * T                               # robust_mf.py:130-134
    return Cmat, mindex
