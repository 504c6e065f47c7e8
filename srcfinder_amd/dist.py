"""Column sharding of the matched filter over the GPUs of one node (one process per GPU).

Cross-track columns are independent in the reference's loop (``for col in arange(ncols)``,
cmf/robust_mf.py:297), so rank r processes the contiguous sample range ``shard_columns(samples, world, r)``
with exactly the arithmetic of a single-GPU run (results are bit-identical for every column), and the only
exchange is ONE gather of the finished score blocks to the destination rank (``gather_columns``; the library call
``robust_mf_sharded`` adds one <= 5 KB gather of the per-column records): ``torch.distributed.gather`` on the
``nccl`` backend is RCCL over xGMI; every peer sends its block straight to the root (7 links in parallel,
12 MB per rank for the float64 score band of a 598 x 20000 flightline; 48 MB with ``gather="product"``).  The helper is backend
agnostic -- the CPU tests drive it with ``gloo``.
"""
from __future__ import annotations

import numpy as np


def shard_columns(samples: int, world: int, rank: int):
    """Contiguous, balanced: sizes differ by at most one column (598 over 8 -> 74/75)."""
    return rank * samples // world, (rank + 1) * samples // world


def _host_staged(group, t):
    """True when ``t`` lives on a GPU but the group's backend moves host memory only (``gloo``: two ranks that share one
    GPU, a node without RCCL peers).  The collective then runs on a host copy of the block and the assembled result goes
    back to the block's device on ``dst`` -- the functional route, not the fast one: RCCL (``nccl``) gathers device buffers
    directly and is what a multi-GPU node uses."""
    import torch.distributed as dist
    return bool(t.is_cuda) and dist.get_backend(group) == "gloo"


class GatherHandle:
    """An in-flight column gather (``gather_columns(..., async_op=True)``): ``wait()`` returns the assembled array
    on ``dst`` (``None`` elsewhere).  The collective runs on the backend's own stream; the buffers live here."""

    def __init__(self, work, send, recv, meta, back=None):
        self.work, self.send, self.recv, self.meta, self.back = work, send, recv, meta, back

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        full = _assemble(self.recv, *self.meta)
        return full if (full is None or self.back is None) else full.to(self.back, non_blocking=True)


def _assemble(recv, shape, axis, samples, world, as_int16):
    import torch
    if recv is None:
        return None
    # one pass: every rank's block (padded along the column axis) is written straight into its column range
    full_shape = list(recv[0].shape)
    full_shape[axis] = samples
    full = torch.empty(full_shape, dtype=recv[0].dtype, device=recv[0].device)
    for r in range(world):
        a, b = shard_columns(samples, world, r)
        full.narrow(axis, a, b - a).copy_(recv[r].narrow(axis, 0, b - a))
    return full.view(torch.int16) if as_int16 else full


def gather_columns(block, samples: int, *, axis=None, group=None, dst: int = 0, async_op: bool = False):
    """Gather per-rank column blocks into the full array on ``dst``; other ranks get ``None``.

    ``axis`` is the axis of ``block`` that runs over this rank's columns.  Default: 1 for arrays of two or more
    dimensions (``[lines, ncols_r, ...]`` products, ``[3, ncols_r]`` column statistics), 0 for vectors; per-column
    records whose FIRST axis is the column (``alphaidx`` / ``status`` ``[ncols_r, k]`` of a multimodal run) must pass
    ``axis=0`` -- nothing is inferred from the shape beyond that default.

    Blocks are padded to the largest shard so a single fixed-size gather suffices.  With ``async_op`` a
    :class:`GatherHandle` is returned instead and the collective overlaps whatever the caller enqueues next
    (the next flightline's compute) until ``wait()``."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if axis is None:
        axis = 1 if block.dim() >= 2 else 0
    axis = axis % block.dim()
    as_int16 = block.dtype == torch.int16
    if as_int16:                        # not a collective dtype everywhere (gloo): ship the bytes
        if axis == block.dim() - 1:
            raise ValueError("int16 blocks are shipped as bytes: the column axis must not be the last one")
        block = block.contiguous().view(torch.uint8)
    a, b = shard_columns(samples, world, rank)
    if block.shape[axis] != b - a:
        raise ValueError("rank %d holds columns [%d, %d) but the block has %d along axis %d"
                         % (rank, a, b, block.shape[axis], axis))
    # the block padded ALONG its column axis to the largest shard (no transposition: plain strided copies on both sides)
    maxc = max(b - a for a, b in (shard_columns(samples, world, r) for r in range(world)))
    shp = list(block.shape)
    shp[axis] = maxc
    back = block.device if _host_staged(group, block) else None
    if back is not None:
        block = block.cpu()              # (waits for the stream that produced the block)
    send = torch.empty(shp, dtype=block.dtype, device=block.device)
    send.narrow(axis, 0, b - a).copy_(block)
    if b - a < maxc:
        send.narrow(axis, b - a, maxc - (b - a)).zero_()
    recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    work = dist.gather(send, recv, dst=dst, group=group, async_op=async_op)
    h = GatherHandle(work if async_op else None, send, recv, (tuple(block.shape), axis, samples, world, as_int16), back)
    return h if async_op else h.wait()


# which axis of each robust_mf() result runs over the columns (multimodal runs return alphaidx / status as [ncols, k])
_COLUMN_AXIS = {"out": 1, "bgmeta": 1, "labels": 1, "colstats": 1, "alphaidx": 0, "nuse": 0, "status": 0, "nll": 0}


def _as_bytes3(v, axis, lead):
    """``v`` as a uint8 tensor [lead, ncols_r, bytes]: its column axis in the middle, everything behind it (and the
    element bytes) folded into the last axis.  ``lead`` is the leading extent shared by the fields of one packed
    exchange (the line count of the image fields; 1 for per-column records, whose column axis is moved to the front)."""
    import torch
    v = v if torch.is_tensor(v) else torch.as_tensor(np.ascontiguousarray(v))
    if lead == 1:
        v = v.movedim(axis, 0).contiguous()
        nc = v.shape[0]
        return v.view(torch.uint8).reshape(1, nc, -1), tuple(v.shape[1:]), v.dtype
    assert axis == 1 and v.shape[0] == lead
    v = v.contiguous()
    nc = v.shape[1]
    return v.view(torch.uint8).reshape(lead, nc, -1), tuple(v.shape[2:]), v.dtype


def gather_packed(fields, samples: int, *, lead: int = 1, group=None, dst: int = 0):
    """ONE collective for several per-rank column blocks: ``fields`` maps a name to ``(block, column axis)``; every block
    becomes a byte slab ``[lead, ncols_r, bytes]`` (``_as_bytes3``), the slabs are concatenated along the byte axis, padded
    to the largest shard along the column axis and gathered once; ``dst`` gets ``{name: full array}`` (``None`` elsewhere).
    ``lead`` > 1 packs image-like fields ``[lines, ncols_r, ...]`` without transposing them (lead = lines); ``lead`` = 1
    packs per-column records (alpha indices, counts, statuses, column statistics, NLL curves).
    The consumers' layout is the reference's product / metadata image pair (``srcfinder_util.py:1624-1635``)."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    a, b = shard_columns(samples, world, rank)
    slabs, layout = [], []
    for name, (v, axis) in fields.items():
        sl, rest, dtype = _as_bytes3(v, axis, lead)
        if sl.shape[1] != b - a:
            raise ValueError("rank %d holds columns [%d, %d) but %s has %d of them" % (rank, a, b, name, sl.shape[1]))
        slabs.append(sl)
        layout.append((name, axis, rest, dtype, sl.shape[2]))
    maxc = max(q - p for p, q in (shard_columns(samples, world, r) for r in range(world)))
    width = sum(l[4] for l in layout)
    back = slabs[0].device if _host_staged(group, slabs[0]) else None
    send = torch.zeros((lead, maxc, width), dtype=torch.uint8, device="cpu" if back is not None else slabs[0].device)
    off = 0
    for sl in slabs:
        send[:, :b - a, off:off + sl.shape[2]].copy_(sl)
        off += sl.shape[2]
    recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    full = torch.empty((lead, samples, width), dtype=torch.uint8, device=send.device)
    for r in range(world):
        p, q = shard_columns(samples, world, r)
        full[:, p:q].copy_(recv[r][:, :q - p])
    if back is not None:
        full = full.to(back)
    out, off = {}, 0
    for name, axis, rest, dtype, nb in layout:
        v = full[:, :, off:off + nb].contiguous().view(dtype)
        off += nb
        if lead == 1:
            out[name] = v.reshape((samples,) + rest).movedim(0, axis).contiguous()
        else:
            out[name] = v.reshape((lead, samples) + rest)
    return out


def robust_mf_sharded(cube_shard, library, samples: int, *, group=None, dst: int = 0, compute=None, gather: str = "score",
                      **kw):
    """Run the matched filter on this rank's column slice ``cube_shard`` [lines, bands, ncols_r] and gather.

    ``gather="score"`` (default; the north star's "single gather of the final score image"): the image
    collective carries the float64 CMF band only -- 8 B/pixel, 12 MB per rank of a 598 x 20000 flightline (plus the 4 B/pixel
    of the int16 metadata image and 8 B/pixel of a label image when the run produced them, in the SAME collective).  ``dst``
    gets ``score`` [lines, samples] float64.  The RGB quick-look bands of the reference's 4-band product
    (cmf/robust_mf.py:212-228, :395-397) are a copy of three cube bands and stay with the rank that read those columns:
    EVERY rank -- ``dst`` or not -- gets its own block [lines, ncols_r, nb] as ``out_local``, so each rank can write its
    columns of the BIP product itself (or the caller asks for ``gather="product"``).
    ``gather="product"``: the whole [lines, samples, nb] float64 product (32 B/pixel, 48 MB per rank) as ``out`` on ``dst``,
    ``score`` a view of its last band.

    Either way TWO collectives: the image slab above, and ONE small record slab (``alphaidx``, ``nuse``, ``status`` [samples]
    -- [samples, k] for a multimodal run --, ``colstats`` [3, samples], ``nll`` when asked for: <= 5 KB per rank for the
    unimodal product).  Returns a dict on every rank: the gathered fields + ``out_local`` on ``dst``, ``{"out_local": ...}``
    alone elsewhere.  ``compute`` defaults to :func:`srcfinder_amd.cmf.robust_mf`; tests inject a CPU stand-in to exercise
    the sharding and the collectives without a GPU."""
    import torch.distributed as dist
    if gather not in ("score", "product"):
        raise ValueError("gather must be 'score' or 'product'")
    if compute is None:
        from .cmf import robust_mf as compute
    res = compute(cube_shard, library, **kw)
    get = (lambda n: res.get(n)) if isinstance(res, dict) else (lambda n: getattr(res, n, None))
    out = get("out")
    lines = out.shape[0]
    images, records = {}, {}
    for name, axis in _COLUMN_AXIS.items():
        v = get(name)
        if v is None:
            continue
        if axis == 1 and v.ndim >= 2 and v.shape[0] == lines and name != "colstats":
            if name == "out" and gather == "score":
                images["score"] = (out[..., out.shape[2] - 1], 1)      # the CMF band is the LAST band (robust_mf.py:212-228)
            else:
                images[name] = (v, axis)
        else:
            records[name] = (v, axis)
    full = gather_packed(images, samples, lead=lines, group=group, dst=dst)
    rec = gather_packed(records, samples, lead=1, group=group, dst=dst)
    if dist.get_rank(group) != dst:
        return {"out_local": out}
    full.update(rec)
    if gather == "product":
        full["score"] = full["out"][..., full["out"].shape[2] - 1]
    full["out_local"] = out
    return full


# ------------------------------------------------------------------------------------------------------
# CNN saliency map over the GPUs of one node (CNN row): replaces the reference's DataParallel,
# which replicates the model and scatters / gathers every batch (cnn/cnn_pred_pipeline.py:113-116)
# ------------------------------------------------------------------------------------------------------
def shard_rows(rows: int, world: int, rank: int):
    """Contiguous, balanced row ranges of the output image (20000 over 8 -> 2500 each)."""
    return rank * rows // world, (rank + 1) * rows // world


def gather_rows(block, rows: int, *, group=None, dst: int = 0):
    """Gather per-rank row blocks ``[rows_r, W]`` into ``[rows, W]`` on ``dst`` (``None`` elsewhere): row blocks of a
    row-major image are contiguous, so the assembly is one concatenation of the received (padded) blocks."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    maxr = max(b - a for a, b in (shard_rows(rows, world, r) for r in range(world)))
    back = block.device if _host_staged(group, block) else None
    send = torch.zeros((maxr,) + tuple(block.shape[1:]), dtype=block.dtype, device="cpu" if back is not None else block.device)
    send[:block.shape[0]].copy_(block)
    recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        a, b = shard_rows(rows, world, r)
        parts.append(recv[r][:b - a])
    full = torch.cat(parts, 0)
    return full if back is None else full.to(back)


def predict_flightline_sharded(cmf2d, *, group=None, dst: int = 0, compute=None, **kw):
    """Tile scorer: every rank holds the whole CMF plane (69 MB padded for a 598 x 20000 flightline) and its own weight
    copy, scores the rows ``shard_rows(H, world, rank)`` and the float32 blocks are gathered once (6 MB per rank).
    ``compute(cmf2d, rows=(r0, r1), **kw) -> [H, W]`` defaults to :func:`srcfinder_amd.cnn.predict_flightline`."""
    import torch.distributed as dist
    if compute is None:
        from .cnn import predict_flightline as compute
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    H = cmf2d.shape[0]
    r0, r1 = shard_rows(H, world, rank)
    sal = compute(cmf2d, rows=(r0, r1), **kw)
    return gather_rows(sal[r0:r1], H, group=group, dst=dst)


def fcn_predict_flightline_sharded(cmf2d, *, scale: int = 32, group=None, dst: int = 0, compute=None, **kw):
    """FCN shift-and-stitch: the scale^2 shifts are independent; rank r runs the shift range
    ``shard_rows(scale^2, world, r)``.  Each shift owns a disjoint set of pixels (one per scale x scale cell) and a
    rank leaves the others at 0, so ONE sum-reduce to ``dst`` assembles the map exactly (x + 0 == x in float32; a NODATA
    pixel is written by the rank whose shift owns it).
    ``compute(cmf2d, shifts=(s0, s1), **kw) -> [H, W]`` defaults to :func:`srcfinder_amd.cnn.fcn_predict_flightline`."""
    import torch.distributed as dist
    if compute is None:
        from .cnn import fcn_predict_flightline as compute
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    s0, s1 = shard_rows(scale * scale, world, rank)
    part = compute(cmf2d, shifts=(s0, s1), scale=scale, **kw).contiguous()
    if _host_staged(group, part):
        host = part.cpu()
        dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM, group=group)
        return host.to(part.device) if rank == dst else None
    dist.reduce(part, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return part if rank == dst else None
