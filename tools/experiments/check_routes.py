import sys, numpy as np, torch
sys.path.insert(0, '.')
from srcfinder_amd import cnn
from srcfinder_amd.cnn_weights import synthetic_plane, synthetic_state_dict
net = cnn.GoogLeNetHIP(synthetic_state_dict(seed=2024))
for (H, W, rows) in ((321, 306, (141, 143)), (21, 140, (13, 15)), (300, 290, (148, 152))):
    plane = synthetic_plane(H, W, seed=5)
    res = {}
    for name in ("split", "split_conv3", "split_unshared", "winograd"):
        info = {}
        res[name] = cnn.predict_flightline(plane, "COVID_QC", net=net, batch=96, rows=rows, route=name, info=info)[rows[0]:rows[1]]
        print("   ", name, "info:", {k: v for k, v in info.items() if k != "scales"})
    b = res["split_unshared"]; v = b != -9999
    for name in ("split", "split_conv3", "winograd"):
        a = res[name]
        print(H, W, name, "bit-equal" if torch.equal(a, b) else "differs", "max rel %.2e" % float(((a[v]-b[v]).abs()/b[v].abs().clamp_min(1e-7)).max()), "values", float(b[v].min()), float(b[v].max()))
