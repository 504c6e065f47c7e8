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
