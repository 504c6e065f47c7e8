#!/usr/bin/env python3
"""Golden vectors of the reference's FCN shift-and-stitch fast mode (cnn/fcn_pred_pipeline.py), made with the REAL
reference classes (development container only).

    python tests/golden/gen_golden_fcn.py

Same stand-ins as gen_golden_cnn.py (torchvision.transforms / rasterio are absent here).  ``FlightlineShiftStitch`` and
``stitch_stack`` run unmodified; the script's ``__main__`` needs a weights file next to the read-only reference, so
its model conversion (:157-160) and batch loop (:225-243) are restated around the imported classes.
Weights: ``srcfinder_amd.cnn_weights.synthetic_state_dict`` (trained ones are not in the checkout).
"""
import os
import sys

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden_cnn as GC  # noqa: E402
from srcfinder_amd.cnn_weights import synthetic_plane, synthetic_state_dict  # noqa: E402


def main():
    torch.set_num_threads(8)
    Compose, Normalize = GC.install_stubs()
    sys.modules["tqdm"] = __import__("types").ModuleType("tqdm")
    sys.modules["tqdm"].tqdm = lambda it, **k: it
    sys.path.insert(0, os.path.join(GC.REF, "cnn"))
    import fcn_pred_pipeline as F
    from archs.googlenet1 import googlenet

    sd_np = synthetic_state_dict(seed=2024)
    model = googlenet(pretrained=False, num_classes=2, init_weights=False).eval()
    full = model.state_dict()
    for k, v in sd_np.items():
        full[k] = torch.as_tensor(v)
    model.load_state_dict(full)
    # fcn_pred_pipeline.py:157-160
    fcn = nn.Sequential(*list(model.children())[:-5])
    fcn.add_module("final_conv", nn.Conv2d(1024, 2, kernel_size=1))
    fcn.final_conv.weight.data.copy_(model.fc.weight.data[:, :, None, None])
    fcn.final_conv.bias.data.copy_(model.fc.bias.data)
    fcn.eval()

    mean, std = 110.6390, 183.9152          # COVID_QC, fcn_pred_pipeline.py:175-182
    tf = Compose([F.ClampCH4(vmin=0, vmax=4000), Normalize([mean], [std])])
    H, W, scale = 45, 70, 32
    plane = synthetic_plane(H, W, seed=77)
    plane[3:6, 10:14] = -9999.0
    GC.PLANE["mem://fcn"] = plane
    ds = F.FlightlineShiftStitch("mem://fcn", transform=tf, scale=scale)
    loader = torch.utils.data.DataLoader(ds, batch_size=16, shuffle=False, num_workers=0)
    allpred, ts, ls = None, [], []
    for (t, l), batch in loader:                                     # :225-243
        with torch.no_grad():
            preds = torch.nn.functional.softmax(fcn(batch), dim=1)
        p1 = preds.numpy()[:, 1, :, :]
        allpred = p1 if allpred is None else np.concatenate((allpred, p1), axis=0)
        ts += t
        ls += l
    stitched = F.stitch_stack(plane.shape, ts, ls, allpred, scale=scale)
    stitched[plane == -9999] = -9999
    out = stitched.astype(np.float32)
    # one shifted canvas and its raw prediction map, for the unit test of the prepare kernel / the trunk
    (t5, l5), canvas = ds[5 * scale + 9]
    print("stack", allpred.shape, "saliency", out.shape, "range", out[out != -9999].min(), out.max())
    np.savez_compressed(os.path.join(HERE, "cnn_fcn_golden.npz"), seed_weights=2024, seed_plane=77, H=H, W=W, scale=scale,
                        mean=mean, std=std, plane=plane, saliency=out, predstack=allpred.astype(np.float32),
                        canvas_5_9=canvas.numpy()[0], versions=np.array("torch %s numpy %s" % (torch.__version__, np.__version__)))


if __name__ == "__main__":
    main()
