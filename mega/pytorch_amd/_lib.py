"""ctypes binding of libmega_hip.so (include/mega_hip.h).

The product path has NO fallback: if the library is missing or a call fails, this raises.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmega_hip.so")

c_int, c_float, c_void_p, c_size_t = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t
c_longlong = ctypes.c_longlong

_ERR = {1: "bad argument", 2: "kernel launch failure", 3: "workspace too small"}

# name -> (restype, argtypes).  Must list every symbol declared in include/mega_hip.h.
SIGNATURES = {
    "mega_conv2d_nhwc": (c_int, [c_void_p] * 6 + [c_int] * 15 + [c_void_p]),
    "mega_conv2d_nhwc_ws": (c_int, [c_void_p] * 6 + [c_int] * 15 + [c_void_p, c_size_t, c_void_p]),
    "mega_conv2d_nhwc_workspace_bytes": (c_size_t, [c_int] * 3),
    "mega_stem_conv_bn_relu": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p]),
    "mega_stem_conv_bn_relu_bf16": (c_int, [c_void_p] * 5 + [c_int] * 3 + [c_void_p]),
    "mega_stem_conv_bn_relu_bf16_u8": (c_int, [c_void_p] * 5 + [c_int] * 3 + [c_float] * 3 + [c_int, c_void_p]),
    "mega_stem_pool_bf16": (c_int, [c_void_p, c_int] + [c_void_p] * 4 + [c_int] * 3 + [c_float] * 3 + [c_int, c_void_p]),
    "mega_stem_pool_dt": (c_int, [c_void_p, c_int] + [c_void_p] * 4 + [c_int] * 3 + [c_float] * 3 + [c_int, c_int, c_void_p]),
    "mega_maxpool3x3s2_nhwc": (c_int, [c_void_p] * 2 + [c_int] * 5 + [c_void_p]),
    "mega_avgpool2x2_ceil_nhwc": (c_int, [c_void_p] * 2 + [c_int] * 5 + [c_void_p]),
    "mega_roi_align_fwd": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_float] + [c_int] * 7 + [c_void_p]),
    "mega_nms_full_workspace_bytes": (c_size_t, [c_int]),
    "mega_nms": (c_int, [c_void_p, c_void_p, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mega_nms_workspace_bytes": (c_size_t, [c_int, c_int]),
    "mega_nms_sorted": (c_int, [c_void_p] * 4 + [c_int, c_int, c_float, c_int, c_int] + [c_void_p] * 4 +
                        [c_size_t, c_void_p]),
    "mega_rpn_select_workspace_bytes": (c_size_t, [c_int, c_int]),
    "mega_rpn_select": (c_int, [c_void_p, c_void_p] + [c_int] * 8 + [c_float, c_int, c_float, c_float, c_float] +
                        [c_void_p] * 4 + [c_size_t, c_void_p]),
    "mega_rpn_select_idx": (c_int, [c_void_p, c_void_p] + [c_int] * 8 + [c_float, c_int, c_float, c_float, c_float] +
                            [c_void_p] * 5 + [c_size_t, c_void_p]),
    "mega_postprocess_workspace_bytes": (c_size_t, [c_int, c_int]),
    "mega_postprocess": (c_int, [c_void_p] * 4 + [c_int, c_int] + [c_float] * 8 + [c_int, c_int] + [c_void_p] * 6 +
                         [c_size_t, c_void_p]),
    "mega_postprocess_batched_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "mega_postprocess_batched": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int] + [c_float] * 8 + [c_int, c_int] +
                                 [c_void_p] * 6 + [c_size_t, c_void_p]),
    "mega_position_logits": (c_int, [c_void_p] * 6 + [c_int] * 4 + [c_void_p]),
    "mega_position_logits_tiled": (c_int, [c_void_p] * 6 + [c_int] * 2 + [c_void_p]),
    "mega_relation_attention_tiled_pos": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                                  c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                                  c_float, c_void_p, c_size_t, c_void_p]),
    "mega_relation_attention_splits": (c_int, [c_int, c_int, c_int]),
    "mega_relation_attention_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "mega_relation_attention": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                        c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                        c_int, c_void_p, c_size_t, c_void_p]),
    "mega_relation_attention_batched": (c_int, [c_void_p, c_int, c_int, c_float, c_int, c_void_p]),
    "mega_position_logits_tiled_batched": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mega_position_logits_tiled_dt": (c_int, [c_void_p] * 6 + [c_int] * 3 + [c_void_p]),
    "mega_position_logits_tiled_batched_dt": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "mega_relation_attention_tiled_pos_dt": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                                     c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                                     c_float, c_int, c_void_p, c_size_t, c_void_p]),
    "mega_conv2d_nhwc_tile": (c_int, [c_int] * 3),
    "mega_conv2d_nhwc_plan": (c_int, [c_int] * 4),
    "mega_conv2d_nhwc_plan_ex": (c_int, [c_int] * 14),
    "mega_preprocess_frames": (c_int, [c_void_p, c_void_p] + [c_int] * 3 + [c_float] * 3 + [c_int, c_void_p]),
    "mega_dff_warp_scale": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    "mega_fgfa_pair_taps": (c_int, [c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "mega_resize_bilinear_u8": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                                                      c_int, c_void_p]),
    "mega_fgfa_warp_aggregate": (c_int, [c_void_p] * 4 + [c_int] * 7 + [c_void_p]),
    "mega_fgfa_warp_aggregate_ring": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p, c_int, c_void_p]),
    "mega_copy_segments": (c_int, [c_void_p, c_int, c_void_p]),
    "mega_copy_cast_segments": (c_int, [c_void_p, c_int, c_void_p]),
    "mega_bottleneck64_fwd": (c_int, [c_void_p] * 11 + [c_int] * 3 + [c_void_p]),
    "mega_bottleneck64_ds_fwd": (c_int, [c_void_p] * 14 + [c_int] * 3 + [c_void_p]),
    "mega_bottleneck64_fwd_dt": (c_int, [c_void_p] * 11 + [c_int] * 4 + [c_void_p]),
    "mega_bottleneck64_ds_fwd_dt": (c_int, [c_void_p] * 14 + [c_int] * 4 + [c_void_p]),
    "mega_cast_f32_to_half": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "mega_copy_cast_segments_dt": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "mega_cast_f32_to_bf16": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mega_split_f32_to_bf16x3": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mega_split_f32_to_planes": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mega_roi_align_fwd_planes": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_float] + [c_int] * 3 + [c_void_p]),
    "mega_conv2d_nhwc_sp": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                    c_int] + [c_int] * 11 + [c_void_p, c_size_t, c_void_p]),
    "mega_split_f32_to_planes_dt": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mega_roi_align_fwd_planes_dt": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_float] + [c_int] * 4 + [c_void_p]),
    "mega_conv2d_nhwc_sp_dt": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                       c_int] + [c_int] * 12 + [c_void_p, c_size_t, c_void_p]),
    "mega_conv2d_nhwc_ks_workspace_bytes": (c_size_t, [c_int] * 5),
    "mega_conv2d_nhwc_ks": (c_int, [c_void_p] * 6 + [c_int] * 16 + [c_void_p, c_size_t, c_void_p]),
    "mega_conv2d_nhwc_subpixel": (c_int, [c_void_p] * 4 + [c_int] * 13 + [c_void_p, c_size_t, c_void_p]),
    "mega_flow_level_assemble": (c_int, [c_void_p] * 5 + [c_int] * 10 + [c_void_p]),
    "mega_flow_pred_finish": (c_int, [c_void_p, c_int, c_void_p, c_float, c_void_p] + [c_int] * 4 + [c_void_p]),
    "mega_flow_conv1_combine": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_longlong, c_int, c_int, c_void_p]),
    "mega_fgfa_warp_aggregate_ring_pos": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p, c_int, c_int, c_void_p]),
    "mega_fgfa_warp_aggregate_ring_pos_batched": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p, c_int, c_int, c_int, c_void_p]),
    "mega_last_error_string": (ctypes.c_char_p, []),
}

_lib = None


def load():
    """Load the library (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libmega_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the MEGA hot path)" % LIB_PATH)
    # torch bundles its own libamdhip64.so.7; ours must resolve to THAT copy (one HIP runtime per process:
    # streams / device pointers are shared with torch).  Loading this library before torch would bind
    # /opt/rocm's copy instead and every launch would fail with hipErrorNoDevice.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        detail = ""
        if rc == 2 and _lib is not None:
            detail = ": " + _lib.mega_last_error_string().decode()
        raise RuntimeError("%s failed: %s (code %d)%s" % (what, _ERR.get(rc, "unknown"), rc, detail))
