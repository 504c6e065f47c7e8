"""Host-side pieces of the CLI mirrors (no GPU): ENVI round trips, header parsing, flag sets."""
import os

import numpy as np

from srcfinder_amd import cli_cnn_pred, cli_fcn_pred, cli_robust_mf, envi


def test_envi_roundtrip_all_interleaves(tmp_path):
    rng = np.random.default_rng(0)
    cube = rng.standard_normal((5, 7, 3)).astype(np.float32)              # BIL: lines, bands, samples
    for il, arr in (("bil", cube), ("bip", cube.transpose(0, 2, 1)), ("bsq", cube.transpose(1, 0, 2))):
        path = str(tmp_path / ("cube_" + il))
        mm = envi.create_image(path, {"lines": 5, "samples": 3, "bands": 7, "data ignore value": -9999,
                                      "description": "test cube"}, np.float32, il)
        mm[...] = arr
        mm.flush()
        back, meta = envi.open_memmap(path)
        assert meta["interleave"] == il and meta["data type"] == 4 and meta["lines"] == 5
        assert float(meta["data ignore value"]) == -9999
        assert np.array_equal(np.asarray(envi.to_bil(back, meta)), cube)


def test_parse_reference_style_header(tmp_path):
    hdr = tmp_path / "prod.hdr"
    hdr.write_text("""ENVI
description = {
  AVIRIS-NG product }
samples = 669
lines   = 2801
bands   = 4
header offset = 0
file type = ENVI Standard
data type = 5
interleave = bip
byte order = 0
map info = { UTM , 1.000 , 1.000 , 600000.0 , 4000000.0 , 3.2 , 3.2 , 13 , North , WGS-84 , units=Meters , rotation=-15.0 }
band names = { Red Radiance (uW/nm/sr/cm2) , Green Radiance (uW/nm/sr/cm2) , Blue Radiance (uW/nm/sr/cm2) , CH4 Absorption (ppm x m) }
data ignore value = -9999
model parameters = { modelname=looshrinkage, bgmodel=unimodal, aminexp=-10.0, amaxexp=0.0, astep=0.05, reflectance=False, active_bands=[351, 422] }
""")
    m = envi.read_header(str(hdr))
    assert (m["samples"], m["lines"], m["bands"], m["data type"], m["interleave"]) == (669, 2801, 4, 5, "bip")
    assert len(m["band names"]) == 4 and m["band names"][3] == "CH4 Absorption (ppm x m)"
    assert m["model parameters"].startswith("{ modelname=looshrinkage") and "active_bands=[351, 422]" in m["model parameters"]


def test_cli_flag_sets_match_the_reference():
    a = cli_robust_mf.build_parser().parse_args(["-m", "-R", "-k", "1", "--rgb_bands", "60,42,24", "in", "lib_ch4.txt", "out"])
    assert a.metadata and a.reflectance and a.kmeans == 1 and a.pcadim == 6 and a.model == "looshrinkage"
    assert (a.input, a.library, a.output) == ("in", "lib_ch4.txt", "out")
    c = cli_cnn_pred.build_parser().parse_args(["flight.img", "-m", "CalCH4_v8", "-g", "0", "1", "-b", "512", "-o", "out"])
    assert c.model == "CalCH4_v8" and c.gpus == [0, 1] and c.batch == 512 and c.output == "out" and c.band == 1


def test_cnn_cli_exits_1_without_weights(tmp_path):
    assert cli_cnn_pred.main([str(tmp_path / "x.img"), "--weights", str(tmp_path / "missing.pt")]) == 1


def test_fcn_cli_flag_set_and_missing_weights(tmp_path):
    """cnn/fcn_pred_pipeline.py:95-121: same flags, same defaults (band 1, scale 32, batch 8), exit 1 without weights."""
    f = cli_fcn_pred.build_parser().parse_args(["flight.img"])
    assert (f.band, f.scale, f.model, f.gpus, f.batch, f.output) == (1, 32, "COVID_QC", [-1], 8, ".")
    f = cli_fcn_pred.build_parser().parse_args(["flight.img", "-n", "4", "-s", "32", "-m", "Permian_QC", "-g", "0", "1", "-b", "4", "-o", "o"])
    assert (f.band, f.scale, f.model, f.gpus, f.batch, f.output) == (4, 32, "Permian_QC", [0, 1], 4, "o")
    assert cli_fcn_pred.main([str(tmp_path / "x.img"), "--weights", str(tmp_path / "missing.pt")]) == 1


def test_systematics_flags_host_logic(tmp_path):
    """N2 host side (triage/cmf_profile.py:182-205): rolling 3-column median, MAD threshold, CSV layout."""
    import numpy as np
    from srcfinder_amd import triage
    rng = np.random.default_rng(5)
    avg = 100 + rng.standard_normal(64)
    avg[20] += 40
    avg[33] = np.nan
    coldiff, sigma, counts = triage.systematics_flags(avg)
    assert np.nanargmax(coldiff) == 20 and counts[0] >= counts[1] >= counts[2] >= 1
    assert np.isnan(coldiff[32:35]).all() and np.isfinite(coldiff[0]) and np.isfinite(coldiff[-1])
    w = np.sort(avg[:3])
    assert coldiff[0] == avg[0] - w[1] and coldiff[5] == avg[5] - np.median(avg[4:7])
    fin = avg[np.isfinite(avg)]
    assert sigma == np.median(np.abs(fin - np.median(fin)))
    prof = np.stack([np.arange(4.0), np.ones(4), np.zeros(4), np.ones(4), np.ones(4)])
    triage.write_column_stats_csv(tmp_path / "c.csv", prof)
    rows = (tmp_path / "c.csv").read_text().strip().split("\n")
    assert rows[0] == "npix,avg,std,min,max" and len(rows) == 5 and rows[2].startswith("1.0,1.0,0.0")
