"""The N > 1 path on hardware, as far as a one-GPU box allows: TWO processes that share cuda:0, each running the real HIP path
on its own column (row) shard, a gloo process group whose collectives are staged through the host (srcfinder_amd.dist:
RCCL refuses two ranks on one device), and rank 0 comparing the assembled product with a single-process run of the whole
cube -- bit for bit (per-column arithmetic independent of the sharding; cmf/robust_mf.py:297)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from srcfinder_amd import cmf, cnn
    from srcfinder_amd import dist as sd
    from srcfinder_amd.cnn_weights import synthetic_plane, synthetic_state_dict
    from srcfinder_amd.synth import make_cube_numpy

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
    lines, samples = 700, 11
    cube = make_cube_numpy(lines, samples, seed=77, abscf_full=lib[:, 2], nodata_lines=2, nodata_column=4)
    s0, s1 = sd.shard_columns(samples, world, rank)
    shard = torch.as_tensor(np.ascontiguousarray(cube[:, :, s0:s1])).cuda()
    ok = True
    for kw in (dict(), dict(gas="co2"), dict(gather="product", metadata=True)):
        got = sd.robust_mf_sharded(shard, lib, samples, **kw)
        if rank == 0:
            run = {k: v for k, v in kw.items() if k != "gather"}
            ref = cmf.robust_mf(torch.as_tensor(cube).cuda(), lib, **run)
            ok = ok and got["score"].is_cuda and torch.equal(got["score"], ref.out[..., 3])
            ok = ok and torch.equal(got["alphaidx"].cpu(), ref.alphaidx.cpu()) and torch.equal(got["status"].cpu(), ref.status.cpu())
            if kw.get("gather") == "product":
                ok = ok and torch.equal(got["out"], ref.out) and torch.equal(got["bgmeta"].cpu(), ref.bgmeta.cpu())
            else:
                ok = ok and torch.equal(got["out_local"], ref.out[:, s0:s1])
        else:
            ok = ok and set(got) == {"out_local"} and tuple(got["out_local"].shape[:2]) == (lines, s1 - s0)
    # the tile scorer: row shards, one gather of the float32 blocks
    net = cnn.GoogLeNetHIP(synthetic_state_dict(seed=2024))
    plane = synthetic_plane(9, 6, seed=3)
    plane[4, 1] = -9999.0
    sal = sd.predict_flightline_sharded(torch.as_tensor(plane).cuda(), model=(0.0, 500.0), net=net, batch=16)
    if rank == 0:
        whole = cnn.predict_flightline(torch.as_tensor(plane).cuda(), (0.0, 500.0), net=net, batch=16)
        ok = ok and torch.equal(sal, whole)
    else:
        ok = ok and sal is None
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def test_two_ranks_share_one_gpu_and_reproduce_the_single_run():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29611 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r, ok = q.get(timeout=600)
        res[r] = ok
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res == {0: True, 1: True}


def test_bench_starts_its_own_ranks_when_no_launcher_is_around():
    """`python bench.py --gpus 2` with no WORLD_SIZE (the shape of the driver's command): the script starts its two ranks as a
    child torch.distributed.run, relays rank 0's JSON line as the last line and returns the child's exit code.  One GPU here,
    so the two ranks share it over host-staged gloo collectives (a functional run, not a scaling number)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SF_BENCH_SHARE_GPU="1", SF_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--lines", "1200",
           "--samples", "40", "--no-placement"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    line = json.loads(last)
    assert line["n_gpus"] == 2 and line["config"]["gather_verified"] is True
    assert line["value"] > 0 and line["scaling"] == "strong"


def test_bench_replicas_mode_one_flightline_per_rank():
    """BASELINE config 5 (`bench.py --gpus 2 --replicas`): every rank takes a whole flightline through CMF + CNN, no collective on the
    data path; rank 0 prints per-GPU and aggregate figures.  Two replicas share the one GPU here (functional run)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SF_BENCH_SHARE_GPU="1", SF_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--replicas", "--steps", "2", "--warmup", "1", "--lines", "600",
           "--samples", "40", "--strip-lines", "6"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and len(line["per_gpu"]) == 2
    assert abs(line["value"] - sum(p["value"] for p in line["per_gpu"])) < 1e-3 and all(p["cnn_windows_per_s"] > 0 for p in line["per_gpu"])
