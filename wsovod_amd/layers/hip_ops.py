"""Raw (non-autograd) Python fronts of the C-ABI kernels: tensors in, tensors out.

Every function enqueues on torch's current HIP stream and never synchronises.  There is no
CPU fallback: CPU tensors or a missing library raise RuntimeError.
"""
import ctypes as C

import torch

from .. import _lib
from .._lib import BF16, F32, NCHW, NHWC, GemmDesc, check, dtype_code, lib, ptr, require_gpu, stream


def feature_layout(feat):
    """(layout code, N, C, H, W) of a logical-NCHW feature tensor; NHWC == torch.channels_last."""
    if feat.dim() != 4:
        raise RuntimeError(f"wsovod_hip: expected a 4-D (N,C,H,W) feature map, got {tuple(feat.shape)}")
    N, Cc, H, W = feat.shape
    if feat.is_contiguous():
        return NCHW, N, Cc, H, W
    if feat.is_contiguous(memory_format=torch.channels_last):
        return NHWC, N, Cc, H, W
    raise RuntimeError("wsovod_hip: feature map must be contiguous (NCHW) or channels_last (NHWC)")


def _rois_f32(rois):
    if rois.dim() != 2 or rois.size(1) != 5:
        raise RuntimeError(f"wsovod_hip: rois must be (R,5), got {tuple(rois.shape)}")
    return rois.to(torch.float32).contiguous()


def roi_pool_forward(feat, rois, spatial_scale, output_size, roi_scale=None, out_dtype=None, need_argmax=True):
    """RoI max pool -> (out (R,C,ph,pw), argmax int32 or None)."""
    require_gpu(feat, rois, roi_scale)
    layout, N, Cc, H, W = feature_layout(feat)
    rois = _rois_f32(rois)
    ph, pw = output_size
    R = rois.size(0)
    out_dtype = out_dtype or feat.dtype
    out = torch.empty((R, Cc, ph, pw), dtype=out_dtype, device=feat.device)
    argmax = torch.empty((R, Cc, ph, pw), dtype=torch.int32, device=feat.device) if need_argmax else None
    if roi_scale is not None:
        roi_scale = roi_scale.to(torch.float32).contiguous()
    check(lib().wsovod_roi_pool_forward(
        ptr(feat), dtype_code(feat.dtype), layout, ptr(rois), ptr(roi_scale), R, N, Cc, H, W, ph, pw,
        C.c_float(spatial_scale), ptr(out), dtype_code(out_dtype), ptr(argmax), stream()), "roi_pool_forward")
    return out, argmax


def roi_pool_backward(grad_out, rois, argmax, input_shape, channels_last=False, roi_scale=None):
    require_gpu(grad_out, rois, argmax)
    N, Cc, H, W = input_shape
    rois = _rois_f32(rois)
    grad_out = grad_out.to(torch.float32).contiguous()
    R, _, ph, pw = grad_out.shape
    mf = torch.channels_last if channels_last else torch.contiguous_format
    grad_in = torch.zeros((N, Cc, H, W), dtype=torch.float32, device=grad_out.device).contiguous(memory_format=mf)
    check(lib().wsovod_roi_pool_backward(
        ptr(grad_out), ptr(rois), ptr(roi_scale), ptr(argmax), R, N, Cc, H, W, ph, pw,
        NHWC if channels_last else NCHW, ptr(grad_in), stream()), "roi_pool_backward")
    return grad_in


def roi_align_forward(feat, rois, spatial_scale, output_size, sampling_ratio, aligned, roi_scale=None,
                      out_dtype=None):
    require_gpu(feat, rois, roi_scale)
    layout, N, Cc, H, W = feature_layout(feat)
    rois = _rois_f32(rois)
    ph, pw = output_size
    R = rois.size(0)
    out_dtype = out_dtype or feat.dtype
    out = torch.empty((R, Cc, ph, pw), dtype=out_dtype, device=feat.device)
    if roi_scale is not None:
        roi_scale = roi_scale.to(torch.float32).contiguous()
    check(lib().wsovod_roi_align_forward(
        ptr(feat), dtype_code(feat.dtype), layout, ptr(rois), ptr(roi_scale), R, N, Cc, H, W, ph, pw,
        C.c_float(spatial_scale), int(sampling_ratio), int(bool(aligned)), ptr(out), dtype_code(out_dtype),
        stream()), "roi_align_forward")
    return out


def roi_align_backward(grad_out, rois, spatial_scale, sampling_ratio, aligned, input_shape, channels_last=False,
                       roi_scale=None):
    require_gpu(grad_out, rois)
    N, Cc, H, W = input_shape
    rois = _rois_f32(rois)
    grad_out = grad_out.to(torch.float32).contiguous()
    R, _, ph, pw = grad_out.shape
    mf = torch.channels_last if channels_last else torch.contiguous_format
    grad_in = torch.zeros((N, Cc, H, W), dtype=torch.float32, device=grad_out.device).contiguous(memory_format=mf)
    check(lib().wsovod_roi_align_backward(
        ptr(grad_out), ptr(rois), ptr(roi_scale), R, N, Cc, H, W, ph, pw, C.c_float(spatial_scale),
        int(sampling_ratio), int(bool(aligned)), NHWC if channels_last else NCHW, ptr(grad_in), stream()),
        "roi_align_backward")
    return grad_in


def _ld(t):
    if t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError("wsovod_hip gemm: operands must be 2-D with a contiguous last dim")
    return t.stride(0)


def gemm_nt(A, B, *, out=None, out_dtype=None, out_t=None, alpha=1.0, row_scale=None, bias=None, residual=None,
            relu=False, dropout_p=0.0, dropout_seed=0, row_group=None, group_add=None, mask_src=None,
            mask_scale=1.0, accumulate=False, M=None, N=None, K=None, conv=None, tile_hint=0, want_c=True):
    """C[M][N] = epilogue(sum_k A[m][k]*B[n][k]); see include/wsovod_hip.h for the epilogue order.

    A: (M,K) or, with `conv` (a dict of geometry), the NHWC input tensor.  B: (N,K).
    `out_t` is an optional (N, >=M) tensor that receives the transposed copy.
    Returns `out` (or None if want_c is False).
    """
    require_gpu(A, B, out, out_t, row_scale, bias, residual, row_group, group_add, mask_src)
    d = GemmDesc()
    d.dtype_in = dtype_code(B.dtype)
    if A.dtype != B.dtype:
        raise RuntimeError(f"wsovod_hip gemm: A is {A.dtype} but B is {B.dtype}")
    d.N = B.size(0) if N is None else N
    d.K = B.size(1) if K is None else K
    d.B, d.ldb = B.data_ptr(), _ld(B)
    if conv is not None:
        g = d.geom
        for k, v in conv.items():
            setattr(g, k, int(v))
        d.conv = 1
        d.M = g.n_img * g.Ho * g.Wo
        d.A, d.lda = A.data_ptr(), g.Cin
    else:
        d.M = A.size(0) if M is None else M
        d.A, d.lda = A.data_ptr(), _ld(A)
    if want_c:
        if out is None:
            out = torch.empty((d.M, d.N), dtype=out_dtype or A.dtype, device=B.device)
        d.C, d.ldc, d.dtype_c = out.data_ptr(), _ld(out), dtype_code(out.dtype)
    if out_t is not None:
        d.Ct, d.ldct, d.dtype_ct = out_t.data_ptr(), _ld(out_t), dtype_code(out_t.dtype)
    d.alpha = alpha
    if row_scale is not None:
        d.row_scale = row_scale.data_ptr()
    if bias is not None:
        if bias.dtype != torch.float32:
            raise RuntimeError("wsovod_hip gemm: bias must be fp32")
        d.bias = bias.data_ptr()
    if residual is not None:
        d.residual, d.ldr, d.dtype_r = residual.data_ptr(), _ld(residual), dtype_code(residual.dtype)
    d.relu = int(bool(relu))
    d.dropout_p, d.dropout_seed = float(dropout_p), int(dropout_seed)
    if group_add is not None:
        d.row_group, d.group_add, d.ld_ga = row_group.data_ptr(), group_add.data_ptr(), _ld(group_add)
    if mask_src is not None:
        d.mask_src, d.ldm, d.dtype_m = mask_src.data_ptr(), _ld(mask_src), dtype_code(mask_src.dtype)
    d.mask_scale = float(mask_scale)
    d.accumulate = int(bool(accumulate))
    d.tile_hint = int(tile_hint)
    check(lib().wsovod_gemm_nt(C.byref(d), stream()), "gemm_nt")
    return out
