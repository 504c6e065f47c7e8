"""world_size-2 (and 3) gloo runs of the column sharding + single gather, on CPU.

The compute of each rank is the CPU oracle standing in for the HIP path (no GPU here); what is under test is
srcfinder_amd.dist: shard boundaries with unequal sizes, padding, gather order, reassembly of the image and of the
per-column outputs -- compared with the oracle run on the whole cube."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, lines, samples, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    sys.path.insert(0, ROOT)
    from oracle import cmf_oracle as O
    from srcfinder_amd import dist as sd
    from srcfinder_amd.synth import make_cube_numpy

    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
    cube = make_cube_numpy(lines, samples, seed=77, abscf_full=lib[:, 2], nodata_lines=2)
    s0, s1 = sd.shard_columns(samples, world, rank)

    def compute(shard, library, **kw):
        r = O.robust_mf_oracle(shard, library)
        return {k: torch.as_tensor(v) for k, v in r.items()}

    # count the collectives and their payload: the default call is ONE image gather of the score band (+ the metadata bytes
    # of this run) and ONE small record gather
    calls = []
    real_gather = dist.gather

    def counting_gather(t, *a, **k):
        calls.append(t.numel() * t.element_size())
        return real_gather(t, *a, **k)

    dist.gather = counting_gather
    got = sd.robust_mf_sharded(np.ascontiguousarray(cube[:, :, s0:s1]), lib, samples, compute=compute)
    ncalls_score, bytes_score = len(calls), list(calls)
    del calls[:]
    prod = sd.robust_mf_sharded(np.ascontiguousarray(cube[:, :, s0:s1]), lib, samples, compute=compute, gather="product")
    ncalls_prod, bytes_prod = len(calls), list(calls)
    dist.gather = real_gather
    maxc = max(b - a for a, b in (sd.shard_columns(samples, world, r) for r in range(world)))
    assert ncalls_score == 2 and ncalls_prod == 2
    assert bytes_score[0] == lines * maxc * (8 + 4)            # float64 score + int16 x 2 metadata per pixel, nothing else
    assert bytes_prod[0] == lines * maxc * (32 + 4)
    assert bytes_score[1] <= 64 * maxc                         # the per-column records: a few dozen bytes per column
    # the overlapped form bench.py uses: two gathers in flight, waited in order
    mine = O.robust_mf_oracle(np.ascontiguousarray(cube[:, :, s0:s1]), lib)
    h1 = sd.gather_columns(torch.as_tensor(mine["out"][..., 3]), samples, dst=0, async_op=True)
    h2 = sd.gather_columns(torch.as_tensor(mine["bgmeta"]), samples, dst=0, async_op=True)
    a1, a2 = h1.wait(), h2.wait()
    if rank == 0:
        ref = O.robust_mf_oracle(cube, lib)
        ok = all(np.array_equal(got[k].numpy(), ref[k], equal_nan=True)
                 for k in ("alphaidx", "nuse", "status", "colstats", "bgmeta"))
        ok = ok and np.array_equal(got["score"].numpy(), ref["out"][..., 3]) and "out" not in got
        ok = ok and np.array_equal(got["out_local"].numpy(), ref["out"][:, s0:s1])
        ok = ok and all(np.array_equal(prod[k].numpy(), ref[k], equal_nan=True)
                        for k in ("out", "alphaidx", "nuse", "status", "colstats", "bgmeta"))
        ok = ok and np.array_equal(a1.numpy(), ref["out"][..., 3]) and np.array_equal(a2.numpy(), ref["bgmeta"])
        q.put(bool(ok))
    else:
        # a non-destination rank keeps its own block of the 4-band product (the RGB bands it read): ADVICE r5
        mine_out = O.robust_mf_oracle(np.ascontiguousarray(cube[:, :, s0:s1]), lib)["out"]
        assert set(got) == {"out_local"} and set(prod) == {"out_local"}
        assert np.array_equal(got["out_local"].numpy(), mine_out) and np.array_equal(prod["out_local"].numpy(), mine_out)
        assert a1 is None and a2 is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,samples", [(2, 9), (3, 10)])
def test_sharded_gather_matches_single_run(world, samples):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + world) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, 80, samples, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _mm_worker(rank, world, port, q):
    """Multimodal-shaped results (alphaidx / status [ncols, k], labels [lines, ncols], nll [ncols, k, A]) through
    robust_mf_sharded with uneven shards: the column axis of every field is explicit (a [10, 2] status
    block used to be gathered along the wrong axis)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    from srcfinder_amd import dist as sd

    dist.init_process_group("gloo", rank=rank, world_size=world)
    lines, samples, k, A = 12, 11, 2, 5
    rng = np.random.default_rng(5)
    full = {"out": rng.normal(size=(lines, samples, 4)),
            "bgmeta": rng.integers(-3, 200, size=(lines, samples, 2)).astype(np.int16),
            "labels": rng.integers(0, 3, size=(lines, samples)).astype(np.uint8),
            "colstats": rng.normal(size=(3, samples)),
            "alphaidx": rng.integers(-2, 200, size=(samples, k)).astype(np.int32),
            "status": rng.integers(0, 3, size=(samples, k)).astype(np.int32),
            "nuse": rng.integers(0, lines, size=samples).astype(np.int32),
            "nll": rng.normal(size=(samples, k, A))}
    s0, s1 = sd.shard_columns(samples, world, rank)

    def compute(shard, library, **kw):
        return {name: torch.as_tensor(np.ascontiguousarray(np.take(v, np.arange(s0, s1), axis=sd._COLUMN_AXIS[name])))
                for name, v in full.items()}

    got = sd.robust_mf_sharded(None, None, samples, compute=compute, gather="product")
    # a block whose column axis disagrees with the rank's shard is refused, not silently mis-assembled
    try:
        sd.gather_columns(torch.zeros(s1 - s0 + 1, k), samples, axis=0)
        raised = False
    except ValueError:
        raised = True
    if rank == 0:
        ok = raised and all(np.array_equal(got[name].numpy(), v) for name, v in full.items())
        ok = ok and np.array_equal(got["score"].numpy(), full["out"][..., 3])
        q.put(bool(ok))
    else:
        assert set(got) == {"out_local"} and raised
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_gather_of_multimodal_fields(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() + world) % 2000
    procs = [ctx.Process(target=_mm_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_shard_columns_cover_everything():
    from srcfinder_amd.dist import shard_columns
    for world in (1, 2, 4, 8):
        edges = [shard_columns(598, world, r) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == 598
        assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in edges]
        assert max(sizes) - min(sizes) <= 1


def _cnn_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    from srcfinder_amd import dist as sd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, W, scale = 37, 11, 4
    g = torch.Generator().manual_seed(5)
    plane = torch.rand((H, W), generator=g)
    plane[3, 4] = -9999.0
    full = torch.where(plane == -9999.0, plane, plane * 2 + 1)            # stand-in "saliency": a pixelwise function

    def tiles(p, rows=None, **kw):                                         # the contract of cnn.predict_flightline(rows=)
        out = torch.zeros_like(p)
        out[rows[0]:rows[1]] = full[rows[0]:rows[1]]
        return out

    def fcn(p, shifts=None, scale=32, **kw):                               # the contract of cnn.fcn_predict_flightline(shifts=)
        out = torch.zeros_like(p)
        for idx in range(*shifts):
            top, left = divmod(idx, scale)
            ys = torch.arange(H)[(torch.arange(H) + scale // 2) % scale == scale - top - 1]
            xs = torch.arange(W)[(torch.arange(W) + scale // 2) % scale == scale - left - 1]
            out[ys[:, None], xs[None, :]] = full[ys[:, None], xs[None, :]]
        return out

    a = sd.predict_flightline_sharded(plane, compute=tiles)
    b = sd.fcn_predict_flightline_sharded(plane, scale=scale, compute=fcn)
    if rank == 0:
        q.put(bool(torch.equal(a, full) and torch.equal(b, full)))
    else:
        assert a is None and b is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_cnn_row_shards_and_fcn_shift_shards(world):
    """Row sharding + gather of the tile scorer and shift sharding + sum-reduce of the FCN mode (uneven shards)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + world) % 2000
    procs = [ctx.Process(target=_cnn_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
