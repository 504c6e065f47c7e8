"""Several flightlines in flight on one GPU.

The reference is run over whole campaigns, one flightline after the other (one ``python cmf/robust_mf.py`` process per
file); on the GPU the stages of ONE flightline cannot fill the device all the time -- the eigensolver and the rank
factorisation are one workgroup per cross-track column (75 workgroups for a 75-column shard on 256 CUs), and every
stage ends in a partially filled last round.  :class:`FlightlinePipeline` keeps ``depth`` flightlines in flight, each
on its own HIP stream with its own scratch, so that one flightline's latency-bound stages run beside another's
bandwidth- or MFMA-bound ones.  Results are bit-identical to sequential calls (same kernels, same launch geometry;
nothing is shared between the slots but the read-only cube/library).

Measured (MI355X, 20000 lines, p = 72; tools/pipeline_probe.py): 598 columns 11.37 -> 10.53 ms per flightline at
depth 2; a 75-column shard 2.30 -> 1.93 (depth 2) -> 1.72 ms (depth 3).
"""
from __future__ import annotations

from . import cmf


class Ticket:
    """A submitted flightline: ``result`` (device tensors, valid once ``wait()`` has been called or the slot's stream
    is otherwise synchronised) and the event that marks its completion."""

    def __init__(self, result, event, slot):
        self.result, self.event, self.slot = result, event, slot

    def _mark_in_use(self, stream):
        import torch
        for name in ("out", "bgmeta", "colstats", "alphaidx", "nuse", "status", "nll", "labels"):
            t = getattr(self.result, name, None)
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(stream)

    def wait(self, stream=None):
        """Make ``stream`` (default: the current stream) wait for this flightline -- no host synchronisation.  The result
        tensors were allocated on the slot's stream: they are marked as in use by the waiting stream, so the caching
        allocator does not hand their memory to the slot's next flightline while the consumer still reads them."""
        import torch
        stream = stream or torch.cuda.current_stream()
        stream.wait_event(self.event)
        self._mark_in_use(stream)
        return self.result

    def synchronize(self):
        """Block the host until this flightline is done.  The results are marked as in use by the CURRENT stream, on which
        the caller will consume them (without that a consumer kernel on another stream raced the slot's next
        flightline once the host dropped its references)."""
        import torch
        self.event.synchronize()
        self._mark_in_use(torch.cuda.current_stream())
        return self.result


class FlightlinePipeline:
    def __init__(self, depth=2, device=None):
        import torch
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.depth = int(depth)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.depth)]
        self._n = 0

    def submit(self, cube_bil, library, **kw):
        """``cmf.robust_mf(cube_bil, library, **kw)`` on the next slot's stream.  The slot first waits for everything
        enqueued so far on the caller's current stream (the cube upload, the consumer of the buffer passed as ``out``).
        Pass a distinct ``out`` buffer per slot (``slot_of_next()``) or none."""
        import torch
        if kw.get("to_numpy"):
            raise ValueError("to_numpy synchronises the host: fetch the result from the ticket instead")
        slot = self._n % self.depth
        self._n += 1
        st = self.streams[slot]
        st.wait_stream(torch.cuda.current_stream(self.device))
        # the inputs were allocated on the caller's stream and are read on the slot's: the caller may drop or reuse them
        # right after submit() without the allocator handing their memory out under the slot
        for t in (cube_bil, library, kw.get("out"), kw.get("bgmeta")):
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(st)
        with torch.cuda.stream(st):
            res = cmf.robust_mf(cube_bil, library, **kw)
            ev = torch.cuda.Event()
            ev.record(st)
        return Ticket(res, ev, slot)

    def slot_of_next(self):
        return self._n % self.depth

    def pick_product_buffers(self, cube_bil, library, *, spares=3, shape=None, **kw):
        """A pool of ``depth`` product buffers ``[lines, ncols, 4]`` float64 for ``submit(..., out=...)``, chosen by measurement.

        The record-writing score kernel is HBM-bound and its time depends, persistently per buffer, on where the caching
        allocator put the 383 MB product buffer (0.735 .. 0.81 ms for the same launch, up to 8 %;
        profiles/r05_score_placement.md).  A long-running host pays for that once: ``depth + spares`` candidates are allocated,
        one flightline is run into each with the score kernel's HIP events on (``sf_cmf_score_timing``), and the fastest
        ``depth`` are kept; the rest go back to the allocator.  Opt-in set-up work of ~``depth + spares`` flightline steps --
        ``submit`` works with any buffer (or none).  ``kw`` is passed to ``robust_mf`` (``active=``, ``reflectance=`` ...).
        Returns ``(buffers, report)``; ``report`` holds the candidates' score-kernel ms and the kept ones."""
        import torch
        from . import _ffi
        L = _ffi.lib()
        if shape is None:
            shape = (cube_bil.shape[0], cube_bil.shape[2], 4)
        cands = [torch.empty(shape, dtype=torch.float64, device=self.device) for _ in range(self.depth + int(spares))]
        tms = []
        for c in cands:
            L.sf_cmf_score_timing(1)
            try:
                self.submit(cube_bil, library, out=c, out_column0=0, **kw)
                self.synchronize()
                tot, nl = _ffi.C.c_double(0.0), _ffi.C.c_int(0)
                L.sf_cmf_score_timing_read(_ffi.C.byref(tot), _ffi.C.byref(nl))
            finally:
                L.sf_cmf_score_timing(0)
            tms.append(tot.value / max(nl.value, 1))
        order = sorted(range(len(cands)), key=lambda i: tms[i])
        keep = [cands[i] for i in order[:self.depth]]
        report = {"candidate_score_ms": [round(t, 4) for t in tms], "kept": sorted(round(tms[i], 4) for i in order[:self.depth])}
        return keep, report

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def close(self):
        """Wait for everything in flight and release the scratch buffers of this pipeline's streams."""
        self.synchronize()
        for st in self.streams:
            cmf._Workspace._bufs.pop((str(self.device), int(st.cuda_stream)), None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
