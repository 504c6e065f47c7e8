"""TEST INFRASTRUCTURE -- the CPU oracle over many columns on all host cores.

The faithful restatement (``cmf_oracle.robust_mf_oracle``: 201 x det + inv + GEMM per column, cmf/robust_mf.py:92-136)
takes ~2 s per 20000-line column at p = 72 and ~45 s at p = 425 on one core.  The full-size parity tests and the
``cpu_baseline`` leg of ``bench.py`` check / time it over dozens of columns: one SPAWNED worker per usable core
(``OMP_NUM_THREADS=1`` each; never fork a process that has initialised HIP), only the active window of the sampled
columns travels to the workers.  Checker only: nothing under ``srcfinder_amd/`` imports this.
"""
from __future__ import annotations

import os
import time

import numpy as np


def usable_cores():
    """CPUs this process can actually run on: the affinity mask, capped by the cgroup CPU quota (v2 cpu.max, v1 cfs)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return max(1, n)


def _warm(_):
    os.environ["OMP_NUM_THREADS"] = "1"
    from oracle import cmf_oracle  # noqa: F401
    return 0


def _columns(job):
    """Worker: the oracle on a window-only sub-cube [lines, p, k] (float32), library window [p]."""
    sub, libw, reflectance = job
    from oracle import cmf_oracle as O
    t0 = time.perf_counter()
    o = O.robust_mf_oracle(sub, libw, active=(1, sub.shape[1]), rgb_bands=(), reflectance=reflectance)
    return o["out"][..., 0], o["alphaidx"], o["status"], o["nuse"], time.perf_counter() - t0


def oracle_columns(sub, libw, *, workers=None, reflectance=False, per_job=1):
    """sub [lines, p, ncols] float32 (the active window of the sampled columns), libw [p] float64.
    Returns dict(score [lines, ncols] f64 with NODATA -9999, alphaidx, status, nuse [ncols], seconds = wall time of the
    map with the pool already warm, workers).  Columns are dealt ``per_job`` at a time so that the pool stays balanced."""
    import multiprocessing as mp
    lines, p, ncols = sub.shape
    workers = max(1, min(workers or usable_cores(), ncols))
    jobs = [list(range(i, min(ncols, i + per_job))) for i in range(0, ncols, per_job)]
    score = np.empty((lines, ncols))
    aidx = np.empty(ncols, np.int64)
    status = np.empty(ncols, np.int32)
    nuse = np.empty(ncols, np.int64)
    libw = np.ascontiguousarray(libw, np.float64)
    payload = [(np.ascontiguousarray(sub[:, :, ch]), libw, bool(reflectance)) for ch in jobs]
    if workers == 1:
        t0 = time.perf_counter()
        outs = [_columns(j) for j in payload]
        dt = time.perf_counter() - t0
    else:
        ctx = mp.get_context("spawn")
        with ctx.Pool(workers) as pool:
            pool.map(_warm, range(workers))          # imports done before the clock starts
            t0 = time.perf_counter()
            outs = pool.map(_columns, payload, chunksize=1)
            dt = time.perf_counter() - t0
    for ch, (s_, a_, st_, n_, _t) in zip(jobs, outs):
        score[:, ch], aidx[ch], status[ch], nuse[ch] = s_, a_, st_, n_
    return {"score": score, "alphaidx": aidx, "status": status, "nuse": nuse, "seconds": dt, "workers": workers}
