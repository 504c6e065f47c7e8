"""CMF -> CNN end to end on one GPU (BASELINE config 4): the matched-filter plane goes straight from the score
kernel's output into the tile scorer without leaving HBM.

The reference chains the two scripts through files (and its CNN script reads band 1 of the raster, which on the
4-band CMF product is the red radiance band -- SURVEY.md D8); here the CMF band (index 3) is passed explicitly.
"""
from __future__ import annotations

from . import cmf, cnn


def cmf_then_cnn(cube_bil, library, weights, model="COVID_QC", batch=256, net=None, mode="tiles", **cmf_kw):
    """Returns (CMFResult, saliency[H, W] float32 on the GPU).  mode "tiles": one 256 x 256 window per pixel
    (cnn/cnn_pred_pipeline.py); "fcn": the reference's shift-and-stitch fast mode (cnn/fcn_pred_pipeline.py)."""
    res = cmf.robust_mf(cube_bil, library, **cmf_kw)
    plane = res.out[..., -1].float().contiguous()        # float64 ppm*m -> the float32 plane the CNN is fed
    if mode == "fcn":
        sal = cnn.fcn_predict_flightline(plane, model, weights=weights, batch=min(batch, 8), net=net)
    elif mode == "tiles":
        sal = cnn.predict_flightline(plane, model, weights=weights, batch=batch, net=net)
    else:
        raise ValueError("mode must be 'tiles' or 'fcn'")
    return res, sal
