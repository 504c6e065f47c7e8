"""CMF -> CNN end to end on one GPU (BASELINE config 4): the matched-filter plane goes straight from the score
kernel's output into the tile scorer without leaving HBM.

The reference chains the two scripts through files (and its CNN script reads band 1 of the raster, which on the
4-band CMF product is the red radiance band -- SURVEY.md D8); here the CMF band (index 3) is passed explicitly.
"""
from __future__ import annotations

from . import cmf, cnn


def cmf_then_cnn(cube_bil, library, weights, model="COVID_QC", batch=256, net=None, **cmf_kw):
    """Returns (CMFResult, saliency[H, W] float32 on the GPU)."""
    res = cmf.robust_mf(cube_bil, library, **cmf_kw)
    plane = res.out[..., -1].float().contiguous()        # float64 ppm*m -> the float32 plane the CNN is fed
    sal = cnn.predict_flightline(plane, model, weights=weights, batch=batch, net=net)
    return res, sal
