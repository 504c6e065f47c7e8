#!/usr/bin/env python3
"""Full-size parity of the sweep's three routes: columns of the benchmark flightline are replaced by columns whose
correlation spectrum spans ~3.5 decades (rank-36 factorisation) and ~6 decades (no factorisation: full-rank kernel), and
all of them are compared with the faithful oracle.  python tools/validate_wide_spectrum.py [columns per kind = 12]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "8")
import numpy as np, torch
from srcfinder_amd import cmf, _ffi
from srcfinder_amd.synth import make_cube_torch
from oracle import cmf_oracle as O

nk = int(sys.argv[1]) if len(sys.argv) > 1 else 12
lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
lines, samples, a0, a1 = 20000, 598, 351, 422
p = a1 - a0 + 1
cube = make_cube_torch(lines, samples, seed=1234, abscf_full=lib[:, 2], nodata_column=samples // 3)
g = torch.Generator(device="cuda"); g.manual_seed(99)
kinds = {}
for i in range(2 * nk):
    c = 5 + 13 * i
    lo = 3e-4 if i < nk else 1e-6
    q, _ = torch.linalg.qr(torch.randn((p, p), generator=g, device="cuda", dtype=torch.float64))
    sd = torch.sqrt(torch.exp(torch.linspace(0.0, float(np.log(lo)), p, device="cuda", dtype=torch.float64)))
    x = 10.0 + 0.5 * (torch.randn((lines, p), generator=g, device="cuda", dtype=torch.float64) * sd) @ q.T
    cube[:, a0 - 1:a1, c] = x.float()
    kinds[c] = "3.5 decades" if i < nk else "6 decades"
cube[:7] = -9999.0
res = cmf.robust_mf(cube, lib, metadata=True)
cols = sorted(kinds) + [0, 300, 597]
host = cube[:, :, cols].cpu().numpy()
t0 = time.time()
o = O.robust_mf_oracle(host, lib)
got = res.out[:, cols, 3].cpu().numpy()
ref = o["out"][..., 3]
nod = ref == -9999.0
tol = 1e-4 * np.abs(ref) + 1e-9 * np.abs(ref[~nod]).max()
rel = (np.abs(got - ref) / tol)[~nod]
ai = res.alphaidx.cpu().numpy()[cols]
print("%d modified + 3 plain columns x %d lines, oracle %.0f s: NODATA equal %s, alpha idx equal %s, max score error / tolerance %.2e"
      % (len(kinds), lines, time.time() - t0, np.array_equal(got == -9999.0, nod), np.array_equal(ai, o["alphaidx"]), rel.max()))
print("alpha indices:", dict(zip(cols, ai.tolist())))
