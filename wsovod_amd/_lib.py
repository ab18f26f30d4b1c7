"""ctypes binding of the C-ABI library `libwsovod_hip.so` (see include/wsovod_hip.h).

This is the reference-side binding a maintainer would add in place of the pybind module
`wsovod._C` (/root/reference/wsovod/layers/vision.cpp:9-13).  The product path has no CPU
fallback: if the library is missing or a call fails, a RuntimeError is raised (the
reference's convention: AT_ERROR -> RuntimeError).
"""
import ctypes as C
import os

import torch  # must be imported first: the library resolves libamdhip64.so.7 to torch's copy

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libwsovod_hip.so")

F32, BF16, BF16X2, BF16X2P, F16MX = 0, 1, 2, 3, 4  # wsovod_dtype (bf16x2 / its planar form / f16mx: include/wsovod_hip.h)
NCHW, NHWC = 0, 1


class ConvGeom(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("n_img", "H", "W", "Cin", "Ho", "Wo", "KH", "KW", "stride", "pad", "dil", "pool")]


class GemmDesc(C.Structure):
    _fields_ = [
        ("dtype_in", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("A", C.c_void_p), ("lda", C.c_longlong),
        ("B", C.c_void_p), ("ldb", C.c_longlong),
        ("C", C.c_void_p), ("ldc", C.c_longlong), ("dtype_c", C.c_int),
        ("Ct", C.c_void_p), ("ldct", C.c_longlong), ("dtype_ct", C.c_int),
        ("alpha", C.c_float),
        ("row_scale", C.c_void_p), ("bias", C.c_void_p),
        ("residual", C.c_void_p), ("ldr", C.c_longlong), ("dtype_r", C.c_int),
        ("relu", C.c_int),
        ("dropout_p", C.c_float), ("dropout_seed", C.c_ulonglong),
        ("row_group", C.c_void_p), ("group_add", C.c_void_p), ("ld_ga", C.c_longlong),
        ("mask_src", C.c_void_p), ("ldm", C.c_longlong), ("dtype_m", C.c_int),
        ("mask_scale", C.c_float),
        ("accumulate", C.c_int),
        ("conv", C.c_int), ("geom", ConvGeom),
        ("tile_hint", C.c_int), ("prof_tag", C.c_int),
        ("A2", C.c_void_p), ("Cin2", C.c_int),
        ("dropout_seed_add", C.c_void_p),
        ("a_plane_bytes", C.c_longlong),
    ]


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char_p), ("launches", C.c_longlong), ("ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double)]


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_longlong, C.c_float
ABI_VERSION = 9  # == wsovod_abi_version() of the library this file's struct layouts and signatures were written for

# name -> argtypes; must list every symbol include/wsovod_hip.h declares (tests check this).
SIGNATURES = {
    "wsovod_last_error": [],
    "wsovod_abi_version": [],
    "wsovod_profile_enable": [_I],
    "wsovod_profile_reset": [],
    "wsovod_profile_collect": [C.POINTER(ProfEntry), _I],
    "wsovod_roi_pool_forward": [_P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _I, _P, _P],
    "wsovod_roi_pool_forward_x2hi": [_P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _I, _P, _P, _P],
    "wsovod_roi_align_forward_x2hi": [_P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _P, _I, _P, _P],
    "wsovod_roi_pool_backward": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "wsovod_roi_loop_pool_forward": [_P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P, _P, _P],
    "wsovod_roi_align_forward": [_P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _P, _I, _P],
    "wsovod_roi_align_backward": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _I, _P, _P],
    "wsovod_gemm_nt": [C.POINTER(GemmDesc), _P],
    "wsovod_preprocess_image": [_P, _P, _P, _P, _I, _I, _I, _P, _P],
    "wsovod_stem_im2col": [_P, _P, _P, _P, _I, _I, _I, _P, _I, _P],
    "wsovod_stem_conv1": [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P],
    "wsovod_maxpool2x2_nhwc": [_P, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "wsovod_maxpool2x2_nhwc_backward": [_P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "wsovod_global_avgpool_nhwc": [_P, _I, _I, _I, _I, _P, _P, _P],
    "wsovod_transpose_cast": [_P, _I, _L, _I, _I, _P, _I, _L, _P],
    "wsovod_cast": [_P, _I, _P, _I, _L, _P],
    "wsovod_row_l2norm_scale": [_P, _I, _L, _I, _I, _F, _F, _P, _P],
    "wsovod_row_l2norm_backward": [_P, _I, _L, _P, _L, _I, _I, _F, _F, _I, _P, _L, _P],
    "wsovod_segment_colsum": [_P, _I, _L, _P, _I, _I, _I, _P, _L, _I, _P, _P],
    "wsovod_colsum_workspace_floats": [_I, _I, _I],
    "wsovod_scale_by_device_scalar": [_P, _L, _P, _P, _P],
    "wsovod_sgd_momentum": [_P, _P, _P, _L, _F, _F, _F, _F, _P, _P],
    "wsovod_mil_forward": [_P, _L, _P, _I, _I, _P, _P, _P, _I, _P],
    "wsovod_mil_backward": [_P, _P, _P, _P, _I, _I, _P, _L, _I, _P],
    "wsovod_image_bce_forward": [_P, _P, _I, _I, _P, _F, _P, _P, _P, _P],
    "wsovod_image_bce_backward": [_P, _P, _I, _I, _P, _P, _P],
    "wsovod_weighted_ce_forward": [_P, _L, _I, _I, _P, _P, _I, _P, _L, _P, _P, _P],
    "wsovod_weighted_l1_box_forward": [_P, _L, _P, _P, _P, _P, _I, _I, _P, _F, _I, _P, _P, _P, _P, _P],
    "wsovod_mask_transpose": [_P, _L, _P, _L, _I, _I, _I, _F, _P, _L, _P, _L, _I, _P],
    "wsovod_mask_transpose_colsum": [_P, _L, _P, _L, _I, _I, _I, _F, _P, _L, _P, _L, _I, _P, _P],
    "wsovod_add_group_rows": [_P, _L, _I, _P, _P, _L, _I, _I, _P, _L, _P],
    "wsovod_scale_rows": [_P, _L, _P, _I, _I, _P, _L, _I, _P],
    "wsovod_data_aware_forward": [_P, _I, _I, _P, _P, _I, _P, _P, _I, _P, _I, _P, _P, _P, _P],
    "wsovod_data_aware_backward": [_P, _I, _P, _I, _P, _P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P],
    "wsovod_pgt_mine_and_label": [_P, _L, _P, _P, _I, _P, _P, _P, _I, _F] + [_P] * 11 + [_P],
    "wsovod_split3_bf16": [_P, _L, _I, _I, _P, _L, _L, _I, _P],
    "wsovod_subsample_labels": [_P, _P, _P, _I, _I, _I, _I, _L, _P, _P],
    "wsovod_sgd_momentum_multi": [_P, _I, _F, _F, _P],
    "wsovod_grad_clip_workspace_floats": [_P, _I],
    "wsovod_grad_clip_coef": [_P, _I, _F, _F, _I, _P, _P, _P],
    "wsovod_roi_pool_workspace_bytes": [_I, _I, _I, _I, _I, _I, _I, _I, _I, _I],
    "wsovod_roi_pool_forward_ws": [_P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _I, _P, _P, _P, _L, _P],
    "wsovod_roi_pool_forward_m2": [_P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _I, _P, _P, _P, _L, _P],
    "wsovod_max2x2_gap_workspace_floats": [_I, _I, _I, _I, _I],
    "wsovod_max2x2_gap_nhwc": [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "wsovod_pack_bf16_multi": [_P, _I, _P],
    "wsovod_sum_shards_bf16": [_P, _I, _L, _P, _P],
    "wsovod_format_rois": [_P, _P, _I, _I, _P, _P, _P, _P],
    "wsovod_gemm_tn": [_P, _L, _P, _L, _I, _I, _I, _P, _L, _F, _I, _P],
    "wsovod_gemm_tn_ex": [_P, _L, _P, _L, _I, _I, _I, _I, _P, _L, _F, _I, _P],
    "wsovod_gemm_tn_sgd": [_P, _L, _P, _L, _I, _I, _I, _I, _F, _P, _P],
    "wsovod_f16mx_encode": [_P, _L, _I, _I, _I, _P, _L, _P, _P],
    "wsovod_f16mx_encode_with": [_P, _L, _I, _I, _P, _L, _P, _P, _P],
    "wsovod_f16mx_from_bf16x2": [_P, _P, _L, _P],
    "wsovod_gemm_f16mx": [_P, _P, _I, _P, _I, _P, _L, _P],
    "wsovod_mask_transpose_ex": [_P, _L, _I, _P, _L, _I, _I, _I, _F, _P, _L, _P, _L, _I, _P, _P],
    "wsovod_bf16x2_encode": [_P, _L, _I, _I, _P, _L, _P],
    "wsovod_bf16x2_decode": [_P, _L, _I, _I, _P, _L, _P],
    "wsovod_stem_conv1_x2": [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P],
    "wsovod_nms_segments": [_P, _P, _P, _I, _I, _I, _F, _I, _P, _P, _P, _P],
    "wsovod_rpn_label_anchors": [_P, _I, _P, _P, _P, _I, _I, _F, _F, _P, _P, _P, _P, _P],
    "wsovod_im2col_rows": [_P, _I, _P, _I] + [_I] * 10 + [_P, _P],
    "wsovod_rpn_decode": [_P, _P, _P, _I, _I, _L, _P, C.POINTER(C.c_float), _F, _F, _P, _P, _P],
}

class SgdTensor(C.Structure):
    """wsovod_sgd_tensor (include/wsovod_hip.h)."""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("momentum_buf", C.c_void_p), ("bf16_shadow", C.c_void_p),
                ("numel", C.c_longlong), ("lr", C.c_float), ("weight_decay", C.c_float), ("grad_is_bf16", C.c_int),
                ("shadow_is_bf16x2", C.c_int), ("used_flag", C.c_void_p), ("grad_coef", C.c_void_p),
                ("clip_value", C.c_float), ("lr_dev", C.c_void_p), ("mx_scale", C.c_void_p)]


class TnSgd(C.Structure):
    """wsovod_tn_sgd (include/wsovod_hip.h)."""
    _fields_ = [("param", C.c_void_p), ("momentum_buf", C.c_void_p), ("shadow", C.c_void_p), ("shadow_is_bf16x2", C.c_int),
                ("lr", C.c_float), ("weight_decay", C.c_float), ("momentum", C.c_float), ("grad_scale", C.c_float),
                ("lr_dev", C.c_void_p), ("mx_scale", C.c_void_p)]


class PackTensor(C.Structure):
    """wsovod_pack_tensor (include/wsovod_hip.h)."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("numel", C.c_longlong)]


_lib = None


def lib():
    """Load (once) and return the shared library; fail loudly if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"wsovod_amd: HIP extension {LIB_PATH} is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc); "
                "there is no CPU fallback on the product path."
            )
        _lib = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(_lib, name)  # AttributeError here = library older than the header
            fn.argtypes = argtypes
            fn.restype = C.c_int
        _lib.wsovod_last_error.restype = C.c_char_p
        for name in ("wsovod_colsum_workspace_floats", "wsovod_grad_clip_workspace_floats", "wsovod_roi_pool_workspace_bytes",
                     "wsovod_max2x2_gap_workspace_floats"):
            getattr(_lib, name).restype = C.c_longlong
        got = _lib.wsovod_abi_version()
        if got != ABI_VERSION:  # the structs of include/wsovod_hip.h (gemm desc, sgd tensor) changed size across versions
            raise RuntimeError(f"wsovod_amd: {LIB_PATH} has ABI version {got}, this package needs {ABI_VERSION}: rebuild "
                               "it (python -c 'import __graft_entry__ as g; g.build()')")
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().wsovod_last_error().decode(errors="replace")
        raise RuntimeError(f"wsovod_hip {what} failed (status {rc}): {msg}")


def dtype_code(t):
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    raise RuntimeError(f"wsovod_hip: unsupported dtype {t} (float32 / bfloat16 only)")


def ptr(t):
    """Device pointer of a tensor as c_void_p (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())


def stream():
    """The current torch HIP stream as a raw hipStream_t (the raw C getters: torch.cuda.current_stream() costs ~9 us
    per call in Python, which is 0.6 ms of a 3 ms host-bound step)."""
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "wsovod_hip: tensors must live on an MI355X device (got a CPU tensor); "
                "the HIP path has no CPU fallback"
            )


PROFILING = [False]  # per-launch hipEvent brackets are on: captured graphs are bypassed (a replay launches nothing here)


def profile_enable(on=True):
    PROFILING[0] = bool(on)
    return lib().wsovod_profile_enable(1 if on else 0)


def profile_reset():
    lib().wsovod_profile_reset()


def profile_collect():
    buf = (ProfEntry * 256)()
    n = lib().wsovod_profile_collect(buf, 256)
    return [
        dict(name=buf[i].name.decode(), launches=buf[i].launches, ms=buf[i].ms,
             flops=buf[i].flops, bytes=buf[i].bytes)
        for i in range(n)
    ]
