"""The spawned oracle pool (oracle/pool.py) returns what the in-process oracle returns, column by column."""
import numpy as np

from oracle import cmf_oracle as O
from oracle import pool as OP
from srcfinder_amd.synth import make_cube_numpy


def test_pool_matches_in_process_oracle(library):
    cube = make_cube_numpy(160, 6, seed=77, abscf_full=library[:, 2])
    a0, a1 = O.active_window("ch4", False)
    ref = O.robust_mf_oracle(cube, library)
    sub = np.ascontiguousarray(cube[:, a0 - 1:a1, :])
    got = OP.oracle_columns(sub, library[a0 - 1:a1, 2], workers=2, per_job=2)
    assert np.array_equal(got["score"], ref["out"][..., 3])
    assert np.array_equal(got["alphaidx"], ref["alphaidx"]) and np.array_equal(got["status"], ref["status"])
    assert np.array_equal(got["nuse"], ref["nuse"])
    assert got["workers"] == 2 and got["seconds"] > 0
    assert OP.usable_cores() >= 1
