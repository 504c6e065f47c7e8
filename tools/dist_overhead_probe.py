import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from srcfinder_amd.inflight import FlightlinePipeline
from srcfinder_amd.synth import make_cube_torch
dev = torch.device("cuda", 0)
lib = np.load('/root/repo/tests/golden/ch4_library.npz')["library"]
cube = make_cube_torch(20000, 75, seed=1, abscf_full=lib[:, 2], device=dev)
pipe = FlightlinePipeline(3, dev)
outs = [torch.empty((20000, 75, 4), dtype=torch.float64, device=dev) for _ in range(3)]
def run(N):
    torch.cuda.synchronize(); T0 = time.perf_counter()
    for i in range(N):
        slot = pipe.slot_of_next()
        pipe.submit(cube, lib, out=outs[slot], out_column0=0)
    pipe.synchronize(); torch.cuda.synchronize()
    return (time.perf_counter() - T0) / N * 1e3
run(10)
print("no dist: %.3f %.3f ms per step" % (run(60), run(60)))
if len(sys.argv) > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=dev)
    print("after init_process_group: %.3f %.3f ms per step" % (run(60), run(60)))
    t = torch.zeros(8, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
    print("after a collective: %.3f %.3f ms per step" % (run(60), run(60)))
    dist.destroy_process_group()
