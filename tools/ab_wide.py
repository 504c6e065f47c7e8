#!/usr/bin/env python3
"""The full-band window (p = 425) on the benchmark flightline: ms per flightline and the NLL / alpha / product hash of the run to an
.npz; --cmp a.npz b.npz compares two library builds bit for bit.  ab_wide.py out.npz [samples=598] [key=value knobs ...]"""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        same = np.array_equal(a[k], b[k], equal_nan=True) if a[k].dtype.kind == "f" else np.array_equal(a[k], b[k])
        print(k, "identical" if same else "DIFFERENT (max rel %.3e)" % np.nanmax(np.abs(a[k].astype(float) - b[k].astype(float)) / (np.abs(b[k].astype(float)) + 1e-300)))
    sys.exit(0)
import torch
from srcfinder_amd import cmf
from srcfinder_amd.synth import make_cube_torch
lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 598
from srcfinder_amd import _ffi
for kv in sys.argv[3:]:
    k, v = kv.split("="); _ffi.lib().sf_debug_set(int(k), int(v))
cube = make_cube_torch(20000, ns, seed=1234, abscf_full=lib[:, 2], nodata_column=ns // 3)
out = torch.empty((20000, ns, 4), dtype=torch.float64, device="cuda")
res = cmf.robust_mf(cube, lib, out=out, active=(1, 425), metadata=True, return_nll=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): cmf.robust_mf(cube, lib, out=out, active=(1, 425))
torch.cuda.synchronize()
print("%d samples, p = 425: %.1f ms per flightline" % (ns, (time.perf_counter() - t0) / 3 * 1e3))
h = hashlib.sha256(res.out.cpu().numpy().tobytes()).hexdigest()
np.savez(sys.argv[1], nll=res.nll.cpu().numpy(), alphaidx=res.alphaidx.cpu().numpy(), status=res.status.cpu().numpy(),
         outhash=np.frombuffer(bytes.fromhex(h), dtype=np.uint8))
