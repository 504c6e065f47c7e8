#!/usr/bin/env python3
"""Generate the CNN golden vectors with the REAL reference classes (development container only).

    python tests/golden/gen_golden_cnn.py

``cnn/archs/googlenet1.py`` imports as is (torch only).  ``cnn/cnn_pred_pipeline.py`` needs ``torchvision.transforms``
and ``rasterio`` (absent here): three trivial stand-in classes (Compose / Normalize / Pad) and a ``rasterio.open``
that returns the in-memory plane are placed in ``sys.modules``; ``ClampCH4`` and
``FlightlineConvolve`` then run unmodified.  The script's ``__main__`` needs a weights file next to the read-only
reference, so its 10-line batch loop (:173-189) is restated here around the imported classes.
Weights: ``srcfinder_amd.cnn_weights.synthetic_state_dict`` (trained ones are not in the checkout).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
from srcfinder_amd.cnn_weights import synthetic_plane, synthetic_state_dict  # noqa: E402

PLANE = {}


def install_stubs():
    tv, tr, rio = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms"), types.ModuleType("rasterio")

    class Compose:
        def __init__(s, ts): s.ts = ts
        def __call__(s, x):
            for t in s.ts:
                x = t(x)
            return x

    class Normalize:
        def __init__(s, mean, std):
            s.m, s.s = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)
        def __call__(s, x): return (x - s.m) / s.s

    class Pad:
        def __init__(s, padding, fill=0, padding_mode="constant"): s.p, s.fill = padding, fill
        def __call__(s, x):
            l, t, r, b = s.p
            return torch.nn.functional.pad(x, (l, r, t, b), value=s.fill)

    tr.Compose, tr.Normalize, tr.Pad = Compose, Normalize, Pad
    tv.transforms = tr
    rio.open = lambda path, *a, **k: types.SimpleNamespace(read=lambda band: PLANE[path])
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "rasterio": rio,
                        "matplotlib": types.ModuleType("matplotlib"),
                        "matplotlib.pyplot": types.ModuleType("matplotlib.pyplot")})
    sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    return Compose, Normalize


def main():
    torch.set_num_threads(8)
    Compose, Normalize = install_stubs()
    sys.path.insert(0, os.path.join(REF, "cnn"))
    import cnn_pred_pipeline as P
    from archs.googlenet1 import googlenet

    sd_np = synthetic_state_dict(seed=2024)
    model = googlenet(pretrained=False, num_classes=2, init_weights=False).eval()
    full = model.state_dict()
    missing = [k for k in full if k not in sd_np and not k.startswith(("aux1.", "aux2.")) and not k.endswith("num_batches_tracked")]
    assert not missing, missing
    for k, v in sd_np.items():
        assert tuple(full[k].shape) == v.shape, k
        full[k] = torch.as_tensor(v)
    model.load_state_dict(full)

    mean, std = 110.6390, 183.9152          # COVID_QC, cnn_pred_pipeline.py:126-133
    tf = Compose([P.ClampCH4(vmin=0, vmax=4000), Normalize([mean], [std])])

    # (1) FlightlineConvolve on a 40 x 30 plane with NODATA: padded image + three tiles
    plane = synthetic_plane(40, 30, seed=7)
    PLANE["p40"] = plane
    ds = P.FlightlineConvolve("p40", transform=tf)
    assert len(ds) == 40 * 30 and ds.dim == 256 and tuple(ds.inshape) == (1, 40, 30)
    tiles_idx = [0, 17 * 30 + 5, 40 * 30 - 1]
    out = dict(plane40=plane, padded40=ds.x.numpy(), tiles_idx=np.array(tiles_idx),
               tiles=np.stack([ds[i].numpy() for i in tiles_idx]))

    # (2) logits + per-block activation checksums for 6 tiles of that plane
    idx6 = [0, 31, 17 * 30 + 5, 600, 911, 1199]
    batch = torch.stack([ds[i] for i in idx6])
    acts = {}
    hooks = []
    for name in ["conv1", "maxpool1", "conv2", "conv3", "maxpool2", "inception3a", "inception3b", "maxpool3",
                 "inception4a", "inception4b", "inception4c", "inception4d", "inception4e", "maxpool4",
                 "inception5a", "inception5b"]:
        hooks.append(getattr(model, name).register_forward_hook(lambda m, i, o, n=name: acts.__setitem__(n, o.detach())))
    with torch.no_grad():
        logits = model(batch)
    for h in hooks:
        h.remove()
    out["logits_idx"] = np.array(idx6)
    out["logits"] = logits.numpy()
    for n, a in acts.items():
        out["act_mean_" + n] = a.mean(dim=(0, 2, 3)).numpy()          # per-channel mean
        out["act_abs_" + n] = a.abs().mean().numpy()
    out["conv1_tile0_ch0"] = acts["conv1"][0, 0].numpy()               # one full 128x128 plane: pins padding/stride
    out["inception3a_tile0"] = acts["inception3a"][0, :, ::4, ::4].numpy()

    # (3) the batch loop of the script (:173-189) on a 24 x 20 plane
    plane2 = synthetic_plane(24, 20, seed=11)
    PLANE["p24"] = plane2
    ds2 = P.FlightlineConvolve("p24", transform=tf)
    loader = torch.utils.data.DataLoader(ds2, batch_size=32, shuffle=False, num_workers=0)
    allpred = []
    for b in loader:
        with torch.no_grad():
            preds = torch.nn.functional.softmax(model(b), dim=1)
            allpred += [x[1] for x in preds.cpu().detach().numpy()]
    allpred = np.array(allpred).reshape(plane2.shape)
    allpred[plane2 == -9999] = -9999
    out["plane24"] = plane2
    out["saliency24"] = allpred.astype(np.float32)
    out["versions"] = np.array("torch %s numpy %s" % (torch.__version__, np.__version__))
    np.savez_compressed(os.path.join(HERE, "cnn_googlenet_golden.npz"), **out)
    print("logits", logits.numpy())
    print("saliency range", allpred[allpred > -9999].min(), allpred.max())


if __name__ == "__main__":
    main()
