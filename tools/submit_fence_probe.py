import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from srcfinder_amd import cmf
from srcfinder_amd.inflight import FlightlinePipeline, Ticket
from srcfinder_amd.synth import make_cube_torch
dev = torch.device("cuda", 0)
lib = np.load('/root/repo/tests/golden/ch4_library.npz')["library"]
for ncols in (75, 598):
    cube = make_cube_torch(20000, ncols, seed=1, abscf_full=lib[:, 2], device=dev)
    def run(N, mode):
        pipe = FlightlinePipeline(3, dev)
        outs = [torch.empty((20000, ncols, 4), dtype=torch.float64, device=dev) for _ in range(3)]
        for i in range(N + 6):
            if i == 6:
                torch.cuda.synchronize(); T0 = time.perf_counter()
            slot = pipe.slot_of_next()
            if mode == "std":
                pipe.submit(cube, lib, out=outs[slot], out_column0=0)
            else:
                pipe._n += 1
                st = pipe.streams[slot]
                with torch.cuda.stream(st):
                    cmf.robust_mf(cube, lib, out=outs[slot], out_column0=0)
                    if mode == "noevent":
                        pass
                    else:
                        ev = torch.cuda.Event(); ev.record(st)
        pipe.synchronize(); torch.cuda.synchronize()
        dt = (time.perf_counter() - T0) / N * 1e3
        pipe.close()
        return dt
    N = 60 if ncols == 75 else 15
    for mode in ("std", "nowait", "noevent", "std", "nowait", "noevent"):
        print(ncols, "%-8s %.3f ms per step" % (mode, run(N, mode)), flush=True)
