"""ctypes binding of libtinyfaces_hip.so (C ABI declared in include/tinyfaces_hip.h).

The library is loaded AFTER `import torch` so that it binds to the HIP runtime torch already
mapped (same soname libamdhip64.so.7) -- one runtime, torch's streams and allocations are
directly usable.  There is no fallback: if the library is missing every op raises.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must be imported before the .so is mapped)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtinyfaces_hip.so")

TF_F32, TF_BF16, TF_F16 = 0, 1, 2
TF_COMM_ID_BYTES = 128
EPI_AFFINE, EPI_RES, EPI_RELU, EPI_STATS, EPI_MASK, EPI_STATS2, EPI_JOIN, EPI_MASK2, EPI_STATS3 = 1, 2, 4, 8, 16, 32, 64, 128, 256
ERRORS = {-1: "TF_ERR_ARG", -2: "TF_ERR_LAUNCH", -3: "TF_ERR_UNSUPPORTED", -4: "TF_ERR_WORKSPACE"}

vp, i32, i64, u64, f32, f64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_double, C.c_size_t


class HipLibraryMissing(RuntimeError):
    pass


class ImagePrepareArgs(C.Structure):
    """tf_image_prepare_args (include/tinyfaces_hip.h)."""
    _fields_ = [("img", vp), ("H", i32), ("W", i32), ("RH", i32), ("RW", i32), ("crop_y", i32), ("crop_x", i32), ("crop_h", i32),
                ("crop_w", i32), ("paste_y", i32), ("paste_x", i32), ("flip", i32), ("OH", i32), ("OW", i32),
                ("mean", f32 * 3), ("std", f32 * 3), ("bg", C.c_ubyte * 3), ("out", vp)]


class Pack2Job(C.Structure):
    """tf_pack2_job (include/tinyfaces_hip.h)."""
    _fields_ = [("src", vp), ("dst", vp), ("dst_t", vp), ("cout", i32), ("cin", i32), ("taps", i32), ("rows_pad", i32), ("cols_pad", i32),
                ("rows_pad_t", i32), ("cols_pad_t", i32)]


class BnFwdDesc(C.Structure):
    """tf_bn_fwd_desc (include/tinyfaces_hip.h)."""
    _fields_ = [("stat", vp), ("gamma", vp), ("beta", vp), ("scale", vp), ("shift", vp), ("mean", vp), ("invstd", vp),
                ("running_mean", vp), ("running_var", vp), ("stat_shift", vp)]


class BnBwdDesc(C.Structure):
    """tf_bn_bwd_desc (include/tinyfaces_hip.h)."""
    _fields_ = [("stat", vp), ("gamma", vp), ("mean", vp), ("invstd", vp), ("dgamma", vp), ("dbeta", vp), ("nk", i32), ("kidx", i32)]


class DetnetHooks(C.Structure):
    """tf_detnet_hooks (include/tinyfaces_hip.h): the gradient-ready hooks of ONE backward call."""
    _fields_ = [("blocks", C.POINTER(i32)), ("events", C.POINTER(vp)), ("n", i32), ("fn", vp), ("user", vp), ("single_stream", i32)]


class CommPlan(C.Structure):
    """tf_comm_plan (include/tinyfaces_hip.h): block -> element range of the flat gradient, for tf_comm_allreduce_hook."""
    _fields_ = [("comm", vp), ("grad_flat", vp), ("n", i32), ("blocks", C.POINTER(i32)), ("start", C.POINTER(i64)), ("end", C.POINTER(i64)),
                ("rc", i32), ("issued", i32), ("status", C.POINTER(i32))]


class ConvArgs(C.Structure):
    _fields_ = [("dtype", i32), ("mode", i32),
                ("N", i32), ("H", i32), ("W", i32), ("Cin", i32), ("OH", i32), ("OW", i32), ("Cout", i32),
                ("KH", i32), ("KW", i32), ("stride", i32), ("pad", i32),
                ("ldy", i32), ("epi", i32), ("pro_relu", i32),
                ("x", vp), ("w", vp), ("y", vp),
                ("pro_scale", vp), ("pro_shift", vp), ("epi_scale", vp), ("epi_shift", vp),
                ("aux", vp), ("aux2", vp), ("aux3", vp), ("mask_scale", vp), ("mask_shift", vp),
                ("stat_out", vp), ("tile", i32), ("stat_shift", vp), ("stat_shift_out", vp),
                ("bnf", vp), ("bnf_out", vp), ("bnf_rows", i32), ("bnf_count", f32), ("bnf_eps", f32), ("bnf_momentum", f32),
                ("alg_k", i32), ("alg_n", i32)]


class WgradArgs(C.Structure):
    _fields_ = [("dtype", i32),
                ("N", i32), ("H", i32), ("W", i32), ("Cin", i32), ("OH", i32), ("OW", i32), ("Cout", i32),
                ("KH", i32), ("KW", i32), ("stride", i32), ("pad", i32),
                ("ldx", i32), ("lddy", i32), ("pro_relu", i32),
                ("x", vp), ("dy", vp), ("dw_oihw", vp), ("pro_scale", vp), ("pro_shift", vp),
                ("dw_ld", i32), ("splitk", i32), ("tile", i32), ("packed", i32), ("partial_ws", vp), ("partial_ws_bytes", sz)]


_SIGNATURES = {
    "tf_version": (i32, []),
    "tf_build_id": (C.c_char_p, []),
    "tf_symbol_count": (i32, []),
    "tf_symbol_name": (C.c_char_p, [i32]),
    "tf_targets_workspace_bytes": (sz, [i32]),
    "tf_dense_overlap_targets": (i32, [vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, u64, f64, f64,
                                       vp, vp, vp, sz, vp]),
    "tf_dense_overlap_iou": (i32, [vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "tf_pairwise_iou_distance": (i32, [vp, i32, vp, vp]),
    "tf_nms_workspace_bytes": (sz, [i32]),
    "tf_nms_f64": (i32, [vp, vp, i32, f64, vp, vp, vp, sz, vp]),
    "tf_nms_batched_workspace_bytes": (sz, [C.POINTER(i32), i32]),
    "tf_nms_f64_batched": (i32, [vp, vp, C.POINTER(i32), i32, f64, vp, vp, vp, sz, vp]),
    "tf_decode_workspace_bytes": (sz, [i32, i32, i32]),
    "tf_decode_compact": (i32, [vp, i32, i32, i32, vp, i32, vp, vp, f32, f64, i32, i32, i32, i32, vp, vp, i32, vp, sz, vp]),
    "tf_criterion_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "tf_criterion_fwd_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, f32, i32, i32, f32, vp, vp, u64, vp, vp, vp, vp, vp, sz, vp]),
    "tf_image_prepare": (i32, [C.POINTER(ImagePrepareArgs), vp]),
    "tf_sgd_step": (i32, [vp, vp, vp, i64, f32, f32, f32, f32, vp]),
    "tf_conv_mtiles": (i32, [C.POINTER(ConvArgs)]),
    "tf_conv2d": (i32, [C.POINTER(ConvArgs), vp]),
    "tf_pack_weight": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, i32, i32, vp]),
    "tf_conv2d_wgrad": (i32, [C.POINTER(WgradArgs), vp]),
    "tf_conv2d_wgrad_group": (i32, [C.POINTER(WgradArgs), i32, vp]),
    "tf_wgrad_workspace_bytes": (sz, [C.POINTER(WgradArgs)]),
    "tf_unpack_dw": (i32, [vp, i32, i32, i32, vp, vp]),
    "tf_stem_im2col": (i32, [vp, i32, i32, i32, i32, vp, i32, vp]),
    "tf_stem_wgrad": (i32, [i32, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
    "tf_stem_conv": (i32, [i32, vp, i32, i32, i32, vp, i32, vp, i32, vp, vp, vp, C.POINTER(i32), vp]),
    "tf_maxpool_fwd": (i32, [i32, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    "tf_maxpool_bwd": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "tf_maxpool_bwd_stats": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, C.POINTER(i32), vp]),
    "tf_colstats_blocks": (i32, [i32, i32, i32]),
    "tf_colstats": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, vp, vp]),
    "tf_bn_finalize": (i32, [vp, i32, i32, i32, f32, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, i32, vp]),
    "tf_bn_fold": (i32, [vp, vp, vp, vp, f32, i32, vp, vp, vp]),
    "tf_bn_bwd_finalize": (i32, [vp, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "tf_bn_bwd_apply": (i32, [i32, vp, vp, vp, vp, vp, vp, i64, i32, vp, vp]),
    "tf_bn_relu": (i32, [i32, vp, vp, vp, i64, i32, vp, vp]),
    "tf_bn_add_relu": (i32, [i32, vp, vp, vp, vp, vp, vp, i64, i32, vp, vp]),
    "tf_bn_relu_fused": (i32, [i32, vp, C.POINTER(BnFwdDesc), i32, i64, i32, f32, f32, f32, vp, vp]),
    "tf_bn_add_relu_fused": (i32, [i32, vp, C.POINTER(BnFwdDesc), vp, C.POINTER(BnFwdDesc), i32, i64, i32, f32, f32, f32, vp, vp]),
    "tf_bn_bwd_apply_fused": (i32, [i32, vp, vp, vp, C.POINTER(BnBwdDesc), i32, i64, i32, f32, vp, vp]),
    "tf_upsample_add_crop": (i32, [i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "tf_upsample_add_crop_bwd": (i32, [i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    "tf_reduce_partials": (i32, [vp, i32, i32, i32, i32, i32, vp, i32, vp]),
    "tf_detnet_num_params": (i32, []),
    "tf_detnet_param_name": (C.c_char_p, [i32]),
    "tf_detnet_param_numel": (i64, [i32, i32]),
    "tf_detnet_workspace_bytes": (sz, [i32, i32, i32, i32, i32, i32]),
    "tf_detnet_out_shape": (i32, [i32, i32, C.POINTER(i32), C.POINTER(i32)]),
    "tf_detnet_param_region_bytes": (sz, [i32, i32, i32]),
    "tf_detnet_set_grad_events": (i32, [vp, vp, i32]),
    "tf_detnet_forward": (i32, [i32, i32, vp, i32, i32, i32, i32, vp, f32, f32, vp, vp, sz, i32, vp]),
    "tf_detnet_backward": (i32, [i32, vp, i32, i32, i32, i32, vp, vp, vp, vp, sz, vp, sz, vp]),
    "tf_comm_available": (i32, []),
    "tf_comm_unique_id": (i32, [vp]),
    "tf_comm_init": (i32, [vp, i32, i32, C.POINTER(vp)]),
    "tf_comm_destroy": (i32, [vp]),
    "tf_comm_rank": (i32, [vp]),
    "tf_comm_world": (i32, [vp]),
    "tf_allreduce_bucket": (i32, [vp, vp, sz, vp]),
    "tf_comm_join": (i32, [vp, vp]),
    "tf_comm_allreduce_hook": (None, [i32, vp, vp]),
    "tf_detnet_ctx_create": (i32, [C.POINTER(vp)]),
    "tf_detnet_ctx_destroy": (i32, [vp]),
    "tf_detnet_forward_ctx": (i32, [vp, i32, i32, i32, vp, i32, i32, i32, i32, vp, f32, f32, vp, vp, sz, i32, vp]),
    "tf_detnet_backward_ctx": (i32, [vp, C.POINTER(DetnetHooks), i32, vp, i32, i32, i32, i32, vp, vp, vp, vp, sz, vp, sz, vp]),
    "tf_pack_weights_batched": (i32, [i32, vp, i32, vp]),
    "tf_pack_weights_tiled": (i32, [i32, vp, i32, vp]),
    "tf_detnet_set_dual_stream": (i32, [i32]),
    "tf_detnet_set_grad_callback": (i32, [vp, vp]),
    "tf_set_stat_rows": (i32, [i32]),
    "tf_get_stat_rows": (i32, []),
    "tf_profile_enable": (i32, [i32]),
    "tf_profile_shapes": (i32, [C.POINTER(C.c_double), i32]),
    "tf_profile_collect": (i32, [C.POINTER(C.c_double), i32]),
}
# debugging / measurement probes (csrc/debug_api.h): exported by the library, NOT part of the C ABI of include/tinyfaces_hip.h
_DEBUG_SIGNATURES = {
    "tf_probe_tr16": (i32, [vp, vp]),
    "tf_debug_conv3x3h_trace": (i32, [vp]),
    "tf_debug_conv3x3h_tile_rows": (i32, []),
    "tf_debug_probe": (i32, [i32, i32, i32, vp, sz, i32, vp]),
    "tf_debug_probe_chain": (i32, [i32, i32, i32, vp, sz, i32, i32, vp]),
}

_lib = None


# entry points of the EXPERIMENTAL build only (build.py --experimental; include/tinyfaces_hip.h `#ifdef TF_EXPERIMENTAL`): bound when the library has them
_EXPERIMENTAL_SIGNATURES = {
    "tf_conv2d_bnbwd": (i32, [C.POINTER(ConvArgs), C.POINTER(BnBwdDesc), vp, vp, i32, f32, vp]),
    "tf_conv2d_bnfwd": (i32, [C.POINTER(ConvArgs), C.POINTER(BnFwdDesc), vp, C.POINTER(BnFwdDesc), vp, i32, f32, f32, f32, vp]),
}


def experimental():
    """True when the loaded library was built with TF_EXPERIMENTAL (the measured-and-lost kernels compiled in)."""
    return lib().tf_build_id().decode().endswith("+x")


def lib():
    """The loaded library (raises HipLibraryMissing with build instructions if absent)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build it with `python tiny-faces-pytorch_amd/build.py` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the hot path.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError here == ABI mismatch with include/tinyfaces_hip.h
            fn.restype, fn.argtypes = res, args
        for name, (res, args) in list(_DEBUG_SIGNATURES.items()) + list(_EXPERIMENTAL_SIGNATURES.items()):
            fn = getattr(l, name, None)
            if fn is not None:
                fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def identity():
    """What a committed profile is stamped with (scripts/pmc_traffic.py, scripts/stamp_profiles.py) and what bench.py compares it to:
    the ABI version, the digest of the sources the loaded library was built from, and the sha256 of the library file itself."""
    import hashlib
    l = lib()
    with open(LIB_PATH, "rb") as f:
        sha = hashlib.sha256(f.read()).hexdigest()
    return {"tf_version": int(l.tf_version()), "build_id": l.tf_build_id().decode(), "so_sha256": sha}


def symbols():
    l = lib()
    return [l.tf_symbol_name(i).decode() for i in range(l.tf_symbol_count())]


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {ERRORS.get(rc, rc)}")


def require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor is on {t.device}; the tiny-faces hot path only exists as HIP kernels "
                           "for MI355X (gfx950) -- there is no CPU fallback.")


def ptr(t):
    return 0 if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def tf_dtype(dtype):
    if dtype in (torch.float32, "fp32", "f32", TF_F32):
        return TF_F32
    if dtype in (torch.bfloat16, "bf16", TF_BF16):
        return TF_BF16
    if dtype in (torch.float16, "fp16", "f16", TF_F16):       # inference only (BASELINE.json configs[4])
        return TF_F16
    raise ValueError(f"unsupported compute dtype {dtype}")


def torch_dtype(tfd):
    return {TF_F32: torch.float32, TF_BF16: torch.bfloat16, TF_F16: torch.float16}[tfd]
