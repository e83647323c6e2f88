"""ctypes binding of libsmos_hip.so (the C ABI declared in include/smos.h).

There is no CPU fallback behind these calls: if the library is missing or a kernel reports an
error the caller gets a RuntimeError, exactly like the reference's extension modules.
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMOS_HIP_LIB") or os.path.join(_PKG, "lib", "libsmos_hip.so")   # override: diagnostic builds

c_i64p = ctypes.POINTER(ctypes.c_int64)
c_f32p = ctypes.POINTER(ctypes.c_float)
c_f64p = ctypes.POINTER(ctypes.c_double)
vp = ctypes.c_void_p
i64 = ctypes.c_int64
i32 = ctypes.c_int32

# name -> argtypes, in the order of include/smos.h
SIGNATURES = {
    "smos_abi_version": [],
    "smos_debug_set_conv_grid_cap": [i32],
    "smos_voxel_maxpool_fwd": [vp, c_i64p, vp, vp, c_i64p, vp, i64, i64, i64, i32, c_i64p, c_f32p, i32, vp, vp],
    "smos_voxel_maxpool_bwd": [vp, c_i64p, vp, vp, vp, c_i64p, vp, i64, i64, i64, i32, c_i64p, c_f32p, i32, vp],
    "smos_bilinear_gather_fwd": [vp, c_i64p, vp, i32, vp, c_i64p, i64, i64, i64, i64, i64, c_f32p, vp],
    "smos_msda_fwd": [vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i32, vp],
    "smos_msda_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i32, vp],
    "smos_tta_argmax": [vp, i64, i64, i64, vp, vp, vp],
    "smos_vote_clear": [vp, vp],
    "smos_vote_accumulate": [vp, i64, i64, vp, c_f64p, i32, vp, vp],
    "smos_vote_accumulate_frames": [i32, ctypes.POINTER(vp), c_i64p, c_i64p, ctypes.POINTER(vp), ctypes.POINTER(c_f64p), i32, vp, vp],
    "smos_vote_resolve": [vp, i64, i64, vp, i32, vp, vp, vp, vp],
    "smos_dbscan_work_bytes": [i64],
    "smos_dbscan": [vp, i64, i64, ctypes.c_double, i32, vp, vp, i64, i32, vp],
    "smos_box_vote": [vp, i64, i64, vp, c_f64p, vp, i32, vp, vp],
    "smos_bias_act": [vp, i64, i64, vp, vp, i64, i64, vp, i64, i64, i64, i64, i64, i32, vp],
    "smos_downsample_epilogue": [vp, c_i64p, vp, c_i64p, vp, vp, i64, i64, i64, i64, i64, i64, i32, vp],
    "smos_channel_gate_residual": [vp, i64, i64, vp, vp, vp, vp, vp, vp, i64, i64, vp, i64, i64, vp, i64, i64, i64, i64, vp],
    "smos_pointnet_scatter": [vp, vp, i32, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i32, i32, i32, vp],
    "smos_stem_scan_state_words": [i64],
    "smos_stem_mark": [vp, i32, i64, i64, i64, i64, i64, vp, vp, vp],
    "smos_stem_scan": [vp, i64, i64, i64, vp, vp, vp, vp, vp, i64, vp],
    "smos_stem_gemm": [vp, vp, vp, ctypes.POINTER(vp), ctypes.POINTER(vp), i64, i64, vp],
    "smos_stem_epilogue": [ctypes.POINTER(vp), vp, vp, vp, vp, i64, i64, i64, i64, i64, vp],
    "smos_pointnet_scatter_rows": [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i32, i32, i32, vp],
    "smos_pointnet_scatter_rows_live": [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i32, i32, i32, vp, vp],
    "smos_point_head_weight_floats": [],
    "smos_point_head": [vp, i64, vp, vp, i64, i64, i64, i64, i64, i64, vp],
    "smos_point_head_live": [vp, i64, vp, vp, i64, i64, i64, i64, i64, i64, vp, vp],
    "smos_conv_cl_sum_chunks": [i64, i64],
    "smos_conv_cl": [vp, i64, vp, vp, vp, i64, vp, i64, i64, i64, i64, i64, i64, i32, i32, i32, i32, i32, i32, i32, vp, vp],
    "smos_conv_rows_cl": [vp, i64, vp, vp, vp, i64, vp, i64, i64, i64, i64, i64, i64, i32, i32, i32, i32, vp, vp],
    "smos_conv_wino_sum_chunks": [i64, i64],
    "smos_conv_wino_cl": [vp, i64, vp, vp, vp, i64, vp, i64, i64, i64, i64, i64, i64, i32, i32, vp, vp],
    "smos_basic_block_ws_floats": [i64, i64, i64, i64],
    "smos_basic_block_cl": [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i64, vp, i64, vp, i64, i64, i64, i64, i32, vp],
    "smos_unbalance_block_cl": [vp, i64, vp, vp, i64, i64, vp, vp, i64, i64, vp, vp, vp, i64, vp, i64, i64, i64, i64, i64, i32, vp],
    "smos_zero_views_cl": [i32, ctypes.POINTER(vp), c_i64p, c_i64p, c_i64p, vp],
    "smos_conv_wino_chain_ws_ints": [i64, i64, i64, i64],
    "smos_conv_wino_chain_cl": [i32, vp, i64, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(i32), ctypes.POINTER(vp), c_i64p,
                                ctypes.POINTER(i32), vp, vp, i32, i64, i64, i64, i64, i32, vp],
    "smos_msda_fwd_qp": [vp, vp, vp, i64, i64, i64, i64, i64, i64, vp],
    "smos_add_layer_norm": [vp, vp, vp, vp, vp, i64, i64, ctypes.c_float, vp],
    "smos_tfusion_project": [i32, ctypes.POINTER(vp), c_i64p, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), c_i64p, c_i64p, c_i64p, vp],
    "smos_tfusion_layer_param_floats": [i64],
    "smos_tfusion_layer_stream_floats": [i64, i32],
    "smos_tfusion_layer": [vp, vp, i64, vp, vp, vp, i64, vp, i64, i64, i64, ctypes.c_float, ctypes.c_float, vp],
    "smos_conv_wino1d_cl": [vp, i64, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, i32, i32, vp],
    "smos_upconv_xpass": [vp, vp, i64, i64, i64, i64, i64, vp],
    "smos_upconv_ypass": [vp, i64, vp, vp, i64, vp, i64, vp, i64, i64, i64, i64, i64, i32, vp],
    "smos_upconv_xy_ok": [i64, i64],
    "smos_upconv_xy": [vp, i64, vp, vp, i64, i64, vp, i64, i64, vp, i64, i64, i64, i64, i64, i32, vp],
    "smos_gather_scatter": [vp, c_i64p, vp, i32, c_f32p, vp, i32, c_f32p, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, i64, vp],
    "smos_nhwc_to_nchw": [vp, vp, i64, i64, i64, i64, i64, vp],
    "smos_prep_transform_mask": [vp, i64, c_f64p, c_f64p, vp, vp, vp],
    "smos_prep_emit": [vp, vp, vp, i64, i32, i32, i64, i32, c_f32p, c_f32p, c_f64p, c_i64p, c_f64p, vp, vp, vp, vp],
    "smos_prep_unpad_labels": [vp, i64, vp, vp, i64, vp, vp],
    "smos_bias_act_cl": [vp, i64, vp, vp, i64, vp, i64, i64, i64, i32, vp],
    "smos_downsample_epilogue_cl": [vp, i64, vp, i64, vp, vp, i64, i64, i64, i64, i64, i32, vp],
    "smos_channel_gate_residual_cl": [vp, i64, vp, vp, vp, vp, vp, vp, i64, vp, i64, vp, i64, i64, i64, i64, i64, vp],
    "smos_channel_gate_apply_cl": [vp, i64, vp, vp, vp, vp, vp, vp, i64, vp, i64, vp, i64, vp, i64, i64, i64, i64, vp],
    "smos_upsample_concat_cl": [ctypes.POINTER(vp), c_i64p, c_i64p, c_i64p, c_i64p, i32, vp, i64, i64, i64, vp],
    "smos_gather_scatter_cl": [vp, i64, vp, i32, c_f32p, vp, i32, c_f32p, vp, i64, vp, i64, i64, i64, i64, i64, i64, i64, i64, i64, vp],
    "smos_downsample_pool_branch": [vp, i64, vp, vp, i64, vp, vp, i64, i64, i64, i64, i64, i64, i32, vp],
    "smos_gather_scatter_cl_view": [vp, i64, vp, i32, i64, c_f32p, vp, i32, i64, c_f32p, vp, i64, vp, i64, i64, i64, i64, i64, i64, i64, i64, i64, vp, vp],
    "smos_gather_scatter_cl_live": [vp, i64, vp, i32, c_f32p, vp, i32, c_f32p, vp, i64, vp, i64, i64, i64, i64, i64, i64, i64, i64, i64, vp, vp],
    "smos_upsample_concat": [ctypes.POINTER(vp), c_i64p, c_i64p, c_i64p, c_i64p, c_i64p, i32, vp, i64, i64, i64, vp],
}

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            "streammos_amd: %s is missing -- the HIP kernels are not built. Run "
            "`python -m streammos_amd.build` (or __graft_entry__.build()). There is no CPU fallback." % LIB_PATH)
    # The kernels must run on the SAME HIP runtime instance as torch (stream handles and device pointers are
    # only meaningful inside one runtime).  torch bundles its own libamdhip64.so.7; loading it first makes the
    # dynamic linker resolve this library's DT_NEEDED libamdhip64.so.7 to that already-loaded copy instead of
    # /opt/rocm's, which would be a second, device-less runtime inside the process.
    import torch
    bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.isfile(bundled):
        ctypes.CDLL(bundled, mode=ctypes.RTLD_GLOBAL)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    lib.smos_dbscan_work_bytes.restype = ctypes.c_int64
    lib.smos_stem_scan_state_words.restype = ctypes.c_int64
    lib.smos_conv_cl_sum_chunks.restype = ctypes.c_int64
    lib.smos_conv_wino_sum_chunks.restype = ctypes.c_int64
    lib.smos_basic_block_ws_floats.restype = ctypes.c_int64
    lib.smos_conv_wino_chain_ws_ints.restype = ctypes.c_int64
    lib.smos_point_head_weight_floats.restype = ctypes.c_int64
    lib.smos_tfusion_layer_param_floats.restype = ctypes.c_int64
    lib.smos_tfusion_layer_stream_floats.restype = ctypes.c_int64
    lib.smos_last_error.argtypes = []
    lib.smos_last_error.restype = ctypes.c_char_p
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().smos_last_error().decode(errors="replace")
        raise RuntimeError("%s failed (smos status %d): %s" % (what, rc, msg))


def i64_array(values):
    return (ctypes.c_int64 * len(values))(*[int(v) for v in values])


def f32_array(values):
    return (ctypes.c_float * len(values))(*[float(v) for v in values])


def f64_array(values):
    return (ctypes.c_double * len(values))(*[float(v) for v in values])
