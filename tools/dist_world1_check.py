#!/usr/bin/env python3
"""The sharding helpers of srcfinder_amd.dist through a REAL RCCL process group of size 1 on the GPU (device tensors,
device collectives): robust_mf_sharded, predict_flightline_sharded, fcn_predict_flightline_sharded == the direct calls."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.distributed as dist
from srcfinder_amd import cmf, cnn, dist as sd
from srcfinder_amd.cnn_weights import synthetic_plane, synthetic_state_dict
from srcfinder_amd.synth import make_cube_torch
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
cube = make_cube_torch(400, 9, seed=3, abscf_full=lib[:, 2])
got = sd.robust_mf_sharded(cube, lib, 9, metadata=True, gather="product")
sc = sd.robust_mf_sharded(cube, lib, 9, metadata=True)
ref = cmf.robust_mf(cube, lib, metadata=True)
ok1 = all(torch.equal(got[k], getattr(ref, k)) for k in ("out", "alphaidx", "nuse", "status", "bgmeta")) and torch.equal(got["colstats"], ref.colstats) \
    and torch.equal(sc["score"], ref.out[..., 3]) and torch.equal(sc["bgmeta"], ref.bgmeta) and "out" not in sc
net = cnn.GoogLeNetHIP(synthetic_state_dict(seed=2024))
plane = torch.as_tensor(synthetic_plane(6, 5, seed=1)).cuda()
a = sd.predict_flightline_sharded(plane, net=net, batch=8)
b = cnn.predict_flightline(plane, net=net, batch=8)
c = sd.fcn_predict_flightline_sharded(plane, net=net, batch=16)
d = cnn.fcn_predict_flightline(plane, net=net, batch=16)
print("cmf sharded == direct:", ok1, "| tile CNN:", torch.equal(a, b), "| FCN:", torch.equal(c, d))
dist.destroy_process_group()
sys.exit(0 if (ok1 and torch.equal(a, b) and torch.equal(c, d)) else 1)
