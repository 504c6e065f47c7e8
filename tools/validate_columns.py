#!/usr/bin/env python3
"""One-off wide parity check at full size: N evenly spaced columns of the benchmark flightline, GPU against the
faithful numpy oracle (alpha index exact, scores 1e-4).  python tools/validate_columns.py [ncols_to_check] [lines] [samples]
(lines beyond 20000: a long flightline -- 70000 x 598 x 425 is a 71 GB cube -- exercises the 64-bit index arithmetic and the
line-dependent split counts)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "8")
import numpy as np, torch
from srcfinder_amd import cmf
from srcfinder_amd.synth import make_cube_torch
from oracle import cmf_oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
lines = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
samples = int(sys.argv[3]) if len(sys.argv) > 3 else 598
cube = make_cube_torch(lines, samples, seed=1234, abscf_full=lib[:, 2], nodata_column=samples // 3)
res = cmf.robust_mf(cube, lib)
cols = sorted(set(int(round(i * (samples - 1) / (n - 1))) for i in range(n)) - {samples // 3})
host = cube[:, :, cols].cpu().numpy()
t0 = time.time()
o = O.robust_mf_oracle(host, lib)
got = res.out[:, cols, 3].cpu().numpy()
ref = o["out"][..., 3]
nod = ref == -9999.0
rel = np.abs(got[~nod] - ref[~nod]) / (1e-4 * np.abs(ref[~nod]) + 1e-9 * np.abs(ref[~nod]).max())
ai = res.alphaidx.cpu().numpy()[cols]
print("%d columns x %d lines, oracle %.0f s: NODATA placement equal %s, alpha idx equal %s (%s), max score error / tolerance %.2e"
      % (len(cols), lines, time.time() - t0, np.array_equal(got == -9999.0, nod), np.array_equal(ai, o["alphaidx"]),
         sorted(set(ai.tolist())), rel.max()))
