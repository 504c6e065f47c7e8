"""ctypes binding of libsrcfinder_amd.so (include/srcfinder_amd.h).

The library is the product: there is no CPU fallback.  Importing this module never touches the GPU;
:func:`lib` loads the shared object on first use and raises if it has not been built
(``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C srcfinder_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsrcfinder_amd.so")
_lib = None

vp, i32, f64, sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
f32, i64 = C.c_float, C.c_longlong

# name -> (restype, argtypes); mirrors include/srcfinder_amd.h line by line
SIGNATURES = {
    "sf_version": (i32, []),
    "sf_last_error_string": (C.c_char_p, []),
    "sf_cmf_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "sf_cmf_extract_columns": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    "sf_cmf_column_mean": (i32, [vp, i32, vp, i32, i32, i32, vp, vp, vp, vp]),
    "sf_cmf_covariance": (i32, [vp, i32, vp, vp, vp, i32, i32, i32, vp, vp, vp]),
    "sf_cmf_eigh": (i32, [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp]),
    "sf_cmf_wide_stats": (i32, [vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "sf_cmf_wide_stats_target": (i32, [vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                       vp]),
    "sf_cmf_exact_det_scratch_bytes": (sz, [i32, i32, i32, i32]),
    "sf_cmf_exact_det": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]),
    "sf_cmf_eigh_general": (i32, [vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "sf_cmf_eigh_wide_scratch_bytes": (C.c_size_t, [i32, i32]),
    "sf_cmf_eigh_wide": (i32, [vp, i32, i32, vp, vp, vp, vp, vp]),
    "sf_cmf_loocv": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]),
    "sf_cmf_filter": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]),
    "sf_cmf_score": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, i32, i32, f64,
                           vp, i32, i32, i32, vp, vp, vp, vp]),
    "sf_cmf_run": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32, i32, i32, i32, i32, f64,
                         vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "sf_cnn_prepare_plane": (i32, [vp, i32, i32, f32, f32, f32, f32, i32, vp, vp]),
    "sf_cnn_conv1": (i32, [vp, i32, i32, i32, i64, i32, vp, vp, vp, vp]),
    "sf_cnn_conv1_pool": (i32, [vp, i32, i32, i32, i64, i32, vp, vp, vp, vp]),
    "sf_cnn_maxpool": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, i32, vp]),
    "sf_cnn_conv": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp, i32, i32, vp]),
    "sf_cnn_wino_ok": (i32, [i32, i32, i32]),
    "sf_cnn_wino_weight_floats": (C.c_size_t, [i32, i32]),
    "sf_cnn_split_weights": (i32, [vp, i32, i32, vp, vp, vp, vp]),
    "sf_cnn_conv_split": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, i32, C.c_float, vp, i32, C.c_float, i32, i32, vp, vp]),
    "sf_cnn_absmax": (i32, [vp, sz, vp, vp]),
    "sf_cnn_num_scales": (i32, []),
    "sf_cnn_phase_canvas": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "sf_cnn_ring_pool1": (i32, [vp, i32, i32, i32, C.c_longlong, i32, vp, vp, vp, vp]),
    "sf_cnn_conv_ring": (i32, [vp, i32, C.c_longlong, i32, i32, i32, i32, i32, sz, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, i32,
                               i32, i32, C.c_float, vp, i32, i32, vp, i32, i32, vp, i32, i32, i32, C.c_float, C.c_float, vp, vp]),
    "sf_cnn_pool_gather": (i32, [vp, C.c_longlong, i32, i32, i32, i32, i32, sz, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "sf_cnn_calibrate": (i32, [vp, i32, i32, vp, i32, vp, sz, vp, vp]),
    "sf_cnn_conv_split3_split": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, i32, i32, C.c_float, vp, i32, i32, vp, i32, i32,
                                       vp, i32, i32, i32, C.c_float, C.c_float, vp, vp]),
    "sf_cnn_pool_conv_split_ok": (i32, [i32, i32, i32, i32, i32]),
    "sf_cnn_pool_conv_split": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, vp, i32, C.c_float, vp, i32, i32, vp, vp]),
    "sf_cnn_wino_weights": (i32, [vp, i32, i32, vp, vp]),
    "sf_cnn_conv3x3_wino": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, i32, i32, vp]),
    "sf_cnn_pool_conv": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, i32, i32, vp, vp]),
    "sf_cnn_conv_split3": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, i32, i32, i32, vp, i32, i32, vp, i32, i32,
                                 vp, i32, i32, vp]),
    "sf_cnn_head": (i32, [vp, i32, i32, i32, vp, vp, vp, i64, f32, vp, vp]),
    "sf_cnn_blob_floats": (sz, []),
    "sf_cnn_score_workspace_bytes": (C.c_size_t, [i32, i32, i32]),
    "sf_cnn_score_rows": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, i32, i32, vp, vp, vp, sz, vp]),
    "sf_cnn_fcn_prepare": (i32, [vp, i32, i32, f32, f32, f32, f32, i32, i32, i32, i32, i32, vp, vp]),
    "sf_cnn_conv1_image": (i32, [vp, i32, i32, i32, vp, vp, vp, i32, vp]),
    "sf_cnn_fcn_stitch": (i32, [vp, i32, i32, i32, i32, i32, vp, i32, i32, f32, vp, vp]),
    "sf_cnn_conv1_f16": (i32, [vp, i32, i32, i32, i64, i32, vp, vp, vp, vp]),
    "sf_cnn_maxpool_f16": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, i32, vp]),
    "sf_cnn_conv_f16": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp, i32, i32, vp]),
    "sf_cnn_conv_split3_f16": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, i32, i32, i32, vp, i32, i32, vp, i32, i32,
                                     vp, i32, i32, vp]),
    "sf_cnn_head_f16": (i32, [vp, i32, i32, i32, vp, vp, vp, i64, f32, vp, vp]),
    "sf_cmf_kmeans": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, C.c_ulonglong, i32, vp, vp, vp]),
    "sf_cmf_score_cluster": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, vp, vp]),
    "sf_cmf_colstats_rows": (i32, [vp, i32, i32, i32, vp, i32, i32, f64, vp, vp]),
    "sf_debug_lowrank": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]),
    "sf_debug_wlr_bytes": (C.c_size_t, [i32]),
    "sf_debug_wlr": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]),
    "sf_debug_wsweep_stamps": (i32, [vp, i32]),
    "sf_debug_wtri_scratch_bytes": (sz, [i32, i32]),
    "sf_debug_wtri": (i32, [vp, vp, i32, i32, vp, vp, vp, vp, vp]),
    "sf_debug_wtri_stamps": (i32, [vp, i32]),
    "sf_debug_lu_stamps": (i32, [vp, i32]),
    "sf_cmf_column_profile_robust": (i32, [vp, i32, i32, i32, i32, f64, f64, vp, vp]),
    "sf_cmf_column_profile": (i32, [vp, i32, i32, i32, i32, f64, vp, vp, vp]),
    "sf_masks_pixel": (i32, [vp, i32, i32, i32, i32, i32, f32, i32, i32, f32, f32, i32, f32, i32, f32, i32,
                             vp, vp, vp, vp, vp, vp, vp]),
    "sf_image_dilate_cross": (i32, [vp, vp, i32, i32, i32, vp]),
    "sf_image_dilate_disk_scratch_bytes": (sz, [i32, i32, i32]),
    "sf_image_dilate_disk": (i32, [vp, vp, i32, i32, i32, vp, vp]),
    "sf_image_label8_scratch_bytes": (sz, [i32, i32]),
    "sf_image_label8": (i32, [vp, i32, i32, vp, vp, i32, vp, vp, vp]),
    "sf_image_filter_small_components": (i32, [vp, vp, i32, vp, i32, i32, vp]),
    "sf_masks_compose": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]),
    "sf_linalg_det": (i32, [vp, i32, i32, vp, vp, vp]),
    "sf_linalg_inv": (i32, [vp, i32, i32, vp, vp, vp, vp, vp]),
    "sf_detect_region_stats": (i32, [vp, i32, i32, i32, vp, vp, i32, i32, vp, f64, vp, vp, vp]),
    "sf_cmf_score_timing": (i32, [i32]),
    "sf_debug_set": (i32, [i32, i32]),
    "sf_debug_get": (i32, [i32, C.POINTER(C.c_int)]),
    "sf_cmf_score_timing_read": (i32, [C.POINTER(f64), C.POINTER(i32)]),
}


class SrcfinderError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        # torch first: its wheel carries its own libamdhip64.so.7; loading ours before it would bind the
        # kernels to a second HIP runtime that sees no device.
        import torch  # noqa: F401
        if not os.path.isfile(LIB_PATH):
            raise SrcfinderError(
                "HIP library %s is missing: build it (make -C srcfinder_amd/csrc); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError here = header and library out of sync
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().sf_last_error_string()
        raise SrcfinderError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else ""))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
