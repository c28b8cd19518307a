"""ctypes binding of libmxdet_hip.so (the C-ABI in include/mxdet.h).

The product path has no CPU fallback: if the HIP library is missing, importing an op raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MXDET_LIB") or os.path.join(_HERE, "libmxdet_hip.so")   # MXDET_LIB: ablation builds

c_i32, c_i64, c_u32, c_f32, c_sz, c_vp = C.c_int32, C.c_int64, C.c_uint32, C.c_float, C.c_size_t, C.c_void_p


class PyramidT(C.Structure):
    _fields_ = [
        ("num_levels", c_i32), ("A", c_i32),
        ("H", c_i32 * 8), ("W", c_i32 * 8), ("stride", c_i32 * 8),
        ("cls", c_vp * 8), ("reg", c_vp * 8),
        ("cls_sn", c_i64 * 8), ("cls_sy", c_i64 * 8), ("cls_sx", c_i64 * 8), ("cls_sa", c_i64 * 8),
        ("reg_sn", c_i64 * 8), ("reg_sy", c_i64 * 8), ("reg_sx", c_i64 * 8), ("reg_sc", c_i64 * 8),
        ("dtype", c_i32), ("classes", c_i32),
        ("base_anchors", c_vp * 8),
    ]


class FeatPyramidT(C.Structure):
    _fields_ = [
        ("num_levels", c_i32), ("lvl_min", c_i32),
        ("H", c_i32 * 8), ("W", c_i32 * 8),
        ("spatial_scale", c_f32 * 8),
        ("feat", c_vp * 8),
    ]


class ImageDescT(C.Structure):
    """mxdet_image_desc_t (include/mxdet.h)."""
    _fields_ = [("src", C.c_void_p), ("src_h", C.c_int32), ("src_w", C.c_int32), ("dst_h", C.c_int32),
                ("dst_w", C.c_int32), ("flip", C.c_int32), ("pad_", C.c_int32), ("inv_scale", C.c_double)]


class WgradItemT(C.Structure):
    pass


class ConvItemT(C.Structure):
    pass


class ConvDescT(C.Structure):
    _fields_ = [
        ("N", c_i32), ("H", c_i32), ("W", c_i32), ("Cin", c_i32),
        ("Cout", c_i32), ("KH", c_i32), ("KW", c_i32),
        ("stride", c_i32), ("pad", c_i32),
        ("Ho", c_i32), ("Wo", c_i32),
        ("relu", c_i32), ("res_upsample", c_i32), ("accumulate", c_i32),
        ("prefetch", c_vp), ("prefetch_bytes", c_i64), ("relu_bits", c_vp),
    ]


WgradItemT._fields_ = [("desc", ConvDescT), ("x", c_vp), ("dy", c_vp), ("dw", c_vp), ("db", c_vp)]
ConvItemT._fields_ = [("desc", ConvDescT), ("src", c_vp), ("filt", c_vp), ("bias", c_vp), ("residual", c_vp),
                      ("relu_mask", c_vp), ("dst", c_vp)]

P = C.POINTER

# name -> (restype, argtypes). Every symbol include/mxdet.h declares must appear here
# (tests/test_abi.py checks the header against this table and the built library).
SIGNATURES = {
    "mxdet_last_error": (C.c_char_p, []),
    "mxdet_version": (C.c_char_p, []),
    "mxdet_box_iou": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "mxdet_generate_anchors": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "mxdet_nms_batched_workspace_bytes": (c_sz, [c_i32, c_i32]),
    "mxdet_nms_batched": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_i32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "mxdet_pixel_shuffle2_inv_relu": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "mxdet_mask_paste_workspace_bytes": (c_sz, [c_i64, c_i32]),
    "mxdet_mask_paste": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_sz, c_vp]),
    "mxdet_retina_detect_workspace_bytes": (c_sz, [c_vp, c_i32, c_i32]),
    "mxdet_retina_detect": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_f32, c_f32, c_i32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "mxdet_debug_preprocess_direct": (c_i32, [c_i32]),
    "mxdet_image_preprocess": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "mxdet_polygon_masks": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "mxdet_detection_postprocess_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32]),
    "mxdet_detection_postprocess": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32,
                                            P(c_f32), P(c_f32), c_f32, c_f32, c_i32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "mxdet_proposal_workspace_bytes": (c_sz, [P(PyramidT), c_i32, c_i32]),
    "mxdet_proposal": (c_i32, [P(PyramidT), c_i32, c_vp, c_i32, c_i32, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp,
                               c_vp, c_sz, c_vp]),
    "mxdet_anchor_target_workspace_bytes": (c_sz, [c_i32, c_i64, c_i32]),
    "mxdet_anchor_target": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i32, c_vp, c_f32, c_f32, c_f32, c_i32, c_f32,
                                    c_u32, c_u32, c_vp, c_u32, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "mxdet_proposal_target": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_f32, c_f32, c_f32, c_f32,
                                      c_i32, c_i32, P(c_f32), P(c_f32), c_u32, c_u32, c_vp, c_u32, c_vp, c_vp, c_vp,
                                      c_vp, c_vp, c_vp, c_vp]),
    "mxdet_fpn_level_map": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp]),
    "mxdet_roi_align_fwd": (c_i32, [P(FeatPyramidT), c_i32, c_i32, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp,
                                    c_vp]),
    "mxdet_roi_align_bwd": (c_i32, [P(FeatPyramidT), c_i32, c_i32, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp,
                                    c_vp]),
    "mxdet_roi_align_bwd_gather_workspace_bytes": (c_sz, [P(FeatPyramidT), c_i32, c_i64]),
    "mxdet_roi_align_bwd_gather": (c_i32, [P(FeatPyramidT), c_i32, c_i32, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp,
                                           c_i32, c_vp, c_sz, c_vp]),
    "mxdet_roi_align_bwd_gather_prepare": (c_i32, [P(FeatPyramidT), c_i32, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32,
                                                   c_vp, c_sz, c_vp]),
    "mxdet_roi_align_bwd_gather_prepared": (c_i32, [P(FeatPyramidT), c_i32, c_i32, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32,
                                                    c_vp, c_i32, c_vp, c_sz, c_vp]),
    "mxdet_smooth_l1_fwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_f32, c_vp, c_vp]),
    "mxdet_smooth_l1_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_i32, c_vp, c_vp]),
    "mxdet_loss_workspace_bytes": (c_sz, [c_i64]),
    "mxdet_focal_loss": (c_i32, [c_vp, c_i32, c_vp, c_i64, c_i32, c_f32, c_f32, c_f32, c_vp, c_vp, c_vp, c_sz,
                                 c_vp]),
    "mxdet_anchor_class_labels": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "mxdet_retina_loss_num_partials": (c_i32, [c_i32, c_i32, c_i32, c_i32]),
    "mxdet_retina_loss_level": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i64,
                                        c_i64, c_f32, c_f32, c_f32, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp]),
    "mxdet_rpn_loss_num_partials": (c_i32, [c_i32, c_i32, c_i32]),
    "mxdet_rpn_loss_level": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i64, c_i64, c_f32,
                                     c_f32, c_f32, c_vp, c_vp, c_vp]),
    "mxdet_loss_finalize": (c_i32, [c_vp, c_i32, c_i32, c_vp, c_vp]),
    "mxdet_rcnn_loss": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_f32,
                                c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "mxdet_mask_target": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "mxdet_pixel_shuffle2": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "mxdet_mask_loss_workspace_bytes": (c_sz, [c_i64, c_i32]),
    "mxdet_mask_loss": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "mxdet_conv2d_fwd": (c_i32, [P(ConvDescT), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mxdet_conv2d_fwd_chain": (c_i32, [P(ConvDescT), c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32,
                                       c_i32, c_vp, c_vp]),
    "mxdet_conv2d_fwd_splitk_workspace_bytes": (c_sz, [P(ConvDescT), c_i32]),
    "mxdet_conv2d_fwd_splitk": (c_i32, [P(ConvDescT), c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_sz, c_vp]),
    "mxdet_conv2d_dgrad": (c_i32, [P(ConvDescT), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mxdet_conv2d_wgrad_workspace_bytes": (c_sz, [P(ConvDescT)]),
    "mxdet_conv2d_wgrad": (c_i32, [P(ConvDescT), c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "mxdet_conv2d_grouped_table_bytes": (c_sz, [c_i32]),
    "mxdet_conv2d_grouped_plan": (c_i32, [P(ConvItemT), c_i32, c_i32, c_vp, c_sz, P(c_i32), P(c_i32)]),
    "mxdet_conv2d_grouped": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "mxdet_conv2d_wgrad_grouped_table_bytes": (c_sz, [c_i32]),
    "mxdet_conv2d_wgrad_grouped_plan": (c_i32, [P(WgradItemT), c_i32, c_vp, c_sz, P(c_sz), P(c_i32), P(c_i32), P(c_i32)]),
    "mxdet_conv2d_wgrad_grouped": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_sz, c_sz, c_vp]),
    "mxdet_conv2d_wgrad_grouped_parts": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_sz, c_sz, c_vp]),
    "mxdet_debug_force_conv_cfg": (c_i32, [c_i32]),
    "mxdet_debug_force_wgrad_ksplit": (c_i32, [c_i32]),
    "mxdet_filter_transpose": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "mxdet_filter_transpose_batched": (c_i32, [c_vp, c_i32, c_i32, c_vp]),
    "mxdet_stem_conv7x7": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "mxdet_stem_conv7x7_pool": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "mxdet_maxpool3x3s2": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "mxdet_subsample2": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "mxdet_subsample2_bwd": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "mxdet_upsample2_bwd": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "mxdet_add_bf16": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    "mxdet_relu_bwd_bf16": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    "mxdet_f32_to_bf16": (c_i32, [c_vp, c_i64, c_vp, c_vp]),
    "mxdet_f32_accum_to_bf16": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp]),
    "mxdet_nchw_to_nhwc_bf16": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "mxdet_nhwc_to_nchw_f32": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "mxdet_sgd_momentum_update": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_vp]),
    "mxdet_sgd_momentum_update_sched": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_f32, c_f32, c_f32, c_vp]),
    "mxdet_comm_unique_id": (c_i32, [c_vp]),
    "mxdet_comm_create": (c_i32, [c_vp, c_i32, c_i32, P(c_vp)]),
    "mxdet_comm_destroy": (c_i32, [c_vp]),
    "mxdet_allreduce_bucket": (c_i32, [c_vp, c_vp, c_i64, c_vp, P(c_i32)]),
    "mxdet_comm_wait": (c_i32, [c_vp, c_i32, c_vp]),
    "mxdet_comm_broadcast": (c_i32, [c_vp, c_vp, c_sz, c_i32, c_vp]),
    "mxdet_debug_set_tuning": (c_i32, [c_i32, c_i64]),
}

# entries declared in include/mxdet_debug.h (tuning / test hooks, not part of the drop-in boundary)
DEBUG_SYMBOLS = ("mxdet_debug_force_conv_cfg", "mxdet_debug_force_wgrad_ksplit", "mxdet_debug_preprocess_direct",
                 "mxdet_debug_set_tuning")
TUNING_KEYS = {"T64": 0, "T128": 1, "PAR64": 2, "WG_TARGET": 3, "WG_MINSTEPS": 4, "WG_MAXSTEPS": 5, "T3_ENABLE": 6,
               "T3_TARGET": 7, "T3_MINSTEPS": 8, "T3_NS": 9, "TAIL": 10, "WG_NS": 11, "ROI_TABLE": 12, "ROI_ROWS": 13, "STATIC_TAPS": 14,
               "T128W": 15, "T3_MIX": 16, "SPLITK_TILE": 17, "T3_PER_ITEM": 18}

_lib = None


class MxdetError(RuntimeError):
    pass


def load():
    """Load libmxdet_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MxdetError(
            "libmxdet_hip.so not found at %s -- build it with `python -m mxdetection_amd.build` "
            "(there is no CPU fallback)" % LIB_PATH)
    # torch owns device memory and streams on the host side and bundles its own libamdhip64/libhsa-runtime64 with
    # the same SONAME as /opt/rocm's. It must be mapped FIRST so that this library's DT_NEEDED entries resolve to
    # the runtime torch uses; two HIP/HSA runtimes in one process leave the second one without a device.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)  # tests/test_abi.py asserts that no declared symbol is missing
        if fn is None:
            continue
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    # sweep tooling (tools/tune_sweep.sh): MXDET_TUNE_<KEY>=value is read HERE, by the Python host side, and handed to
    # the debug hook; the library itself never reads the environment
    for key, idx in TUNING_KEYS.items():
        v = os.environ.get("MXDET_TUNE_" + key)
        if v:
            lib.mxdet_debug_set_tuning(idx, int(v))
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().mxdet_last_error().decode("utf-8", "replace")
        raise MxdetError("%s failed (%d): %s" % (what or "mxdet call", rc, msg))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
