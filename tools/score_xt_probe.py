#!/usr/bin/env python3
"""the score launch from the transposed window (k_score_xt, sf_debug_set(1, 300)) against the production
kernel (k_score from the cube) on the benchmark flightline: products compared bit for bit (colstats to rounding), the launch
timed by the library's HIP events on its own stream inside the step (one flightline in flight), and under three in flight."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from srcfinder_amd import _ffi, cmf
    from srcfinder_amd.inflight import FlightlinePipeline
    from srcfinder_amd.synth import make_cube_torch, make_cube_numpy
    L = _ffi.lib()
    lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
    res = {}
    # ---- small ragged cubes first: every output, both product shapes, the CO2 window
    for kw in (dict(metadata=True), dict(rgb_bands=(), metadata=True), dict(gas="co2", metadata=True), dict(columns=(5, 60), metadata=True)):
        cube = torch.as_tensor(make_cube_numpy(301, 70, seed=3, abscf_full=lib[:, 2])).cuda()
        L.sf_debug_set(1, 0)
        a = cmf.robust_mf(cube, lib, **kw)
        L.sf_debug_set(1, 300)
        b = cmf.robust_mf(cube, lib, **kw)
        L.sf_debug_set(1, 0)
        ok = torch.equal(a.out, b.out) and torch.equal(a.bgmeta, b.bgmeta) and torch.allclose(a.colstats, b.colstats, rtol=1e-12, atol=0, equal_nan=True)
        print("small", kw, "identical:", bool(ok))
        assert ok
    lines, samples = 20000, 598
    cube = make_cube_torch(lines, samples, seed=1234, abscf_full=lib[:, 2], device="cuda", nodata_column=samples // 3)
    out = {}
    for name, variant in (("k_score", 0), ("k_score_xt", 300), ("k_score", 0), ("k_score_xt", 300)):
        L.sf_debug_set(1, variant)
        r = cmf.robust_mf(cube, lib, metadata=True)
        torch.cuda.synchronize()
        L.sf_cmf_score_timing(1)
        t0 = time.perf_counter()
        for _ in range(10):
            r = cmf.robust_mf(cube, lib, metadata=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        tot, n = _ffi.C.c_double(0.0), _ffi.C.c_int(0)
        L.sf_cmf_score_timing_read(_ffi.C.byref(tot), _ffi.C.byref(n))
        L.sf_cmf_score_timing(0)
        res.setdefault(name, []).append({"score_ms": round(tot.value / n.value, 4), "step_ms": round(dt * 1e3, 3)})
        out[name] = (r.out.clone(), r.bgmeta.clone(), r.colstats.clone())
        del r
    L.sf_debug_set(1, 0)
    same = torch.equal(out["k_score"][0], out["k_score_xt"][0]) and torch.equal(out["k_score"][1], out["k_score_xt"][1])
    cs = torch.allclose(out["k_score"][2], out["k_score_xt"][2], rtol=1e-12, atol=0, equal_nan=True)
    res["products_bit_identical"] = bool(same)
    res["colstats_equal_to_1e-12"] = bool(cs)
    del out
    # ---- three flightlines in flight (the headline's depth)
    for name, variant in (("k_score", 0), ("k_score_xt", 300)):
        L.sf_debug_set(1, variant)
        outs = [torch.empty((lines, samples, 4), dtype=torch.float64, device="cuda") for _ in range(3)]
        with FlightlinePipeline(3, cube.device) as pipe:
            for i in range(3):
                pipe.submit(cube, lib, out=outs[i], out_column0=0)
            pipe.synchronize()
            t0 = time.perf_counter()
            for i in range(30):
                pipe.submit(cube, lib, out=outs[i % 3], out_column0=0)
            pipe.synchronize()
            res[name].append({"three_in_flight_step_ms": round((time.perf_counter() - t0) / 30 * 1e3, 3)})
        del outs
    L.sf_debug_set(1, 0)
    alg = (4 * 72 + 8) * lines * samples
    for name in ("k_score", "k_score_xt"):
        ms = min(x["score_ms"] for x in res[name] if "score_ms" in x)
        res[name + "_frac_of_8TBs"] = round(alg / (ms * 1e-3) / 8e12, 4)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
