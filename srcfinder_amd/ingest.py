"""File / host array -> HBM: only the bands the matched filter touches.

The reference never reads the whole cube: per column it takes ``img_mm[:, active[0]-1:active[1], col]`` from the
memory map (cmf/robust_mf.py:206-208, :298) and three RGB bands (:395-397).  :func:`stage_cube` does the same for the
GPU: of a 425-band flightline only the active window (72 bands for CH4 radiance) and the three RGB bands cross the
PCIe bus, line chunk by line chunk through two PINNED host buffers -- the host fills one from the file while the
other one's asynchronous copy is in flight -- into a compact device cube ``[lines, p + 3, samples]``
(3.6 GB instead of 20.3 GB for 598 x 20000 x 425).  :class:`CompactCube` carries the band bookkeeping, and
``cmf.robust_mf`` accepts it (or any host array, which it stages the same way) in place of a resident tensor.

:func:`fetch_product` is the way back: device product -> host array / memmap through the same two pinned buffers.
"""
from __future__ import annotations

import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np


class CompactCube:
    """Device cube ``[lines, p + len(rgb_bands), samples]`` float32 that holds, in this order, the active window
    ``active = (a0, a1)`` (1-based inclusive, robust_mf.py:185-194) and the RGB bands (0-based, :395-397) of a cube of
    ``bands_total`` bands."""

    def __init__(self, tensor, active, rgb_bands, bands_total, stats=None):
        self.tensor = tensor
        self.active = (int(active[0]), int(active[1]))
        self.rgb_bands = tuple(int(b) for b in rgb_bands)
        self.bands_total = int(bands_total)
        self.stats = stats or {}

    @property
    def p(self):
        return self.active[1] - self.active[0] + 1

    @property
    def compact_rgb(self):
        return tuple(self.p + i for i in range(len(self.rgb_bands)))

    @property
    def shape(self):
        return tuple(self.tensor.shape)


def band_plan(active, rgb_bands, bands_total):
    """0-based source band of every band of the compact cube."""
    a0, a1 = int(active[0]), int(active[1])
    if a0 < 1 or a1 > bands_total or a1 < a0:
        raise ValueError("bad active window [%d,%d] of %d bands" % (a0, a1, bands_total))
    rgb = [int(b) for b in rgb_bands]
    if len(rgb) not in (0, 3):
        raise Exception("invalid value of rgb_bands argument: %s" % (tuple(rgb),))       # robust_mf.py:225-226
    for b in rgb:
        if b < 0 or b >= bands_total:
            raise IndexError("rgb band %d out of range (cube has %d bands)" % (b, bands_total))
    return list(range(a0 - 1, a1)) + rgb


def _as_bil_view(src, interleave):
    il = str(interleave).lower()
    if il == "bil":
        return src
    if il == "bip":
        return src.transpose(0, 2, 1)
    if il == "bsq":
        return src.transpose(1, 0, 2)
    raise ValueError("unknown interleave %r" % (interleave,))


def _fill(dst, bil, l0, l1, a0, a1, rgb, pool, nthreads, s0=0, s1=None):
    """dst[:l1-l0] <- the active window and the RGB bands of lines l0..l1, samples s0..s1 (host copy; numpy releases the
    GIL for it)."""
    n, p = l1 - l0, a1 - a0 + 1

    def part(i0, i1):
        np.copyto(dst[i0:i1, :p, :], bil[l0 + i0:l0 + i1, a0 - 1:a1, s0:s1], casting="unsafe")
        for i, b in enumerate(rgb):
            np.copyto(dst[i0:i1, p + i, :], bil[l0 + i0:l0 + i1, b, s0:s1], casting="unsafe")

    if pool is None or n < 2 * nthreads:
        part(0, n)
        return
    step = (n + nthreads - 1) // nthreads
    list(pool.map(lambda i0: part(i0, min(n, i0 + step)), range(0, n, step)))


_PINNED = {}          # (shape, dtype) -> two page-locked buffers, reused by later calls (page-locking 200 MB costs ~50 ms)


def _pinned_pair(shape, dtype, pinned):
    import torch
    if not pinned:
        return [torch.empty(shape, dtype=dtype) for _ in range(2)]
    key = (tuple(shape), str(dtype))
    pair = _PINNED.get(key)
    if pair is None:
        if len(_PINNED) >= 4:
            _PINNED.clear()
        pair = [torch.empty(shape, dtype=dtype, pin_memory=True) for _ in range(2)]
        _PINNED[key] = pair
    return pair


def _default_threads():
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(16, n))


def stage_cube(src, active, rgb_bands=(60, 42, 24), *, interleave="bil", device=None, chunk_bytes=96 << 20, pinned=True,
               threads=None, columns=None):
    """Host array / memmap (any ENVI interleave) -> :class:`CompactCube` on ``device``.

    columns=(s0, s1): only that sample range -- what one rank of a column-sharded run needs (in a BIL file the
    shard is a 4 (s1 - s0)-byte fragment of every band row; the other ranks' columns are never read).

    pinned=True : two page-locked chunk buffers, asynchronous copies on a private stream, host fill of chunk i+1
                  overlapped with the copy of chunk i.
    pinned=False: the same chunks through pageable memory with synchronous copies (what ``tensor.cuda()`` does; kept
                  so that ``bench.py`` can time both).
    ``stats`` of the result: bytes moved, seconds, host-fill seconds, GB/s over the whole call."""
    import torch
    if not torch.cuda.is_available():
        from . import _ffi
        raise _ffi.SrcfinderError("no GPU visible: srcfinder_amd has no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    bil = _as_bil_view(src, interleave)
    if bil.ndim != 3:
        raise TypeError("cube must be [lines, bands, samples]")
    lines, bands, samples_all = bil.shape
    s0, s1 = (0, samples_all) if columns is None else (int(columns[0]), int(columns[1]))
    if not 0 <= s0 < s1 <= samples_all:
        raise ValueError("bad column shard [%d,%d) of %d samples" % (s0, s1, samples_all))
    samples = s1 - s0
    a0, a1 = int(active[0]), int(active[1])
    plan = band_plan((a0, a1), rgb_bands, bands)
    rgb = plan[a1 - a0 + 1:]
    nb = len(plan)
    line_bytes = nb * samples * 4
    chunk = max(1, min(lines, int(chunk_bytes) // max(line_bytes, 1)))
    t_begin = time.perf_counter()
    with torch.cuda.device(dev):
        out = torch.empty((lines, nb, samples), dtype=torch.float32, device=dev)
        bufs = _pinned_pair((chunk, nb, samples), torch.float32, pinned)
        views = [b.numpy() for b in bufs]
        events = [None, None]
        copy_stream = torch.cuda.Stream(device=dev) if pinned else None
        if pinned:
            # `out` may be a block the caching allocator just took back from the previous call's compact cube, whose score /
            # sweep kernels can still be reading it on the caller's stream (robust_mf does not synchronise on return): the
            # private stream must not write it before the allocating stream has reached this point
            copy_stream.wait_stream(torch.cuda.current_stream(dev))
        nthreads = max(1, int(threads if threads is not None else _default_threads()))
        pool = ThreadPoolExecutor(nthreads) if nthreads > 1 else None
        t_fill = 0.0
        try:
            for i, l0 in enumerate(range(0, lines, chunk)):
                l1 = min(lines, l0 + chunk)
                k = i & 1
                if events[k] is not None:
                    events[k].synchronize()              # the copy that last read this buffer has finished
                t0 = time.perf_counter()
                _fill(views[k], bil, l0, l1, a0, a1, rgb, pool, nthreads, s0, s1)
                t_fill += time.perf_counter() - t0
                if pinned:
                    with torch.cuda.stream(copy_stream):
                        out[l0:l1].copy_(bufs[k][:l1 - l0], non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(copy_stream)
                    events[k] = ev
                else:
                    out[l0:l1].copy_(bufs[k][:l1 - l0])
            for ev in events:
                if ev is not None:
                    ev.synchronize()                     # the pinned buffers go away with this frame
            if pinned:
                torch.cuda.current_stream(dev).wait_stream(copy_stream)
        finally:
            if pool is not None:
                pool.shutdown()
    dt = time.perf_counter() - t_begin
    nbytes = lines * line_bytes
    stats = {"bytes": nbytes, "seconds": dt, "host_fill_seconds": t_fill, "GBps": nbytes / dt / 1e9 if dt > 0 else 0.0,
             "pinned": bool(pinned), "chunk_lines": chunk, "bands_moved": nb, "bands_total": bands, "columns": (s0, s1)}
    return CompactCube(out, (a0, a1), rgb, bands, stats)


def fetch_product(tensor, dest=None, *, chunk_bytes=96 << 20, pinned=True):
    """Device tensor ``[lines, ...]`` -> host ``dest`` (ndarray / memmap of the same shape; allocated when None), line
    chunks through two pinned buffers with the copy of chunk i+1 in flight while chunk i is written to ``dest``."""
    import torch
    t = tensor.contiguous()
    lines = t.shape[0]
    if dest is None:
        dest = np.empty(tuple(t.shape), dtype=np.dtype(str(t.dtype).replace("torch.", "")))
    if tuple(dest.shape) != tuple(t.shape):
        raise ValueError("dest shape %s != tensor shape %s" % (tuple(dest.shape), tuple(t.shape)))
    if lines == 0:
        return dest
    if not pinned:
        dest[...] = t.cpu().numpy()
        return dest
    row_bytes = max(1, t[0].numel() * t.element_size())
    chunk = max(1, min(lines, int(chunk_bytes) // row_bytes))
    dev = t.device
    with torch.cuda.device(dev):
        bufs = _pinned_pair((chunk,) + tuple(t.shape[1:]), t.dtype, True)
        copy_stream = torch.cuda.Stream(device=dev)
        copy_stream.wait_stream(torch.cuda.current_stream(dev))      # the product is complete before the first copy
        spans = [(l0, min(lines, l0 + chunk)) for l0 in range(0, lines, chunk)]
        events = [None] * len(spans)

        def issue(i):
            l0, l1 = spans[i]
            with torch.cuda.stream(copy_stream):
                bufs[i & 1][:l1 - l0].copy_(t[l0:l1], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            events[i] = ev

        issue(0)
        for i, (l0, l1) in enumerate(spans):
            events[i].synchronize()
            if i + 1 < len(spans):
                # chunk i+1 goes into the OTHER buffer, whose previous content (chunk i-1) has been written out already
                issue(i + 1)
            dest[l0:l1] = bufs[i & 1][:l1 - l0].numpy()
        t.record_stream(copy_stream)
    return dest
