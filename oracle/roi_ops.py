"""ORACLE -- test infrastructure only.  ctypes front of oracle/roi_ops_ref.c (our plain-C
restatement of /root/reference/wsovod/layers/ROILoopPool/ROILoopPool_cpu.cpp:13-123 and of
torchvision roi_align) and of oracle/_ref (the reference's own RoIPool, compiled where it
lies).  All tensors are CPU, fp32, NCHW, contiguous.
"""
import ctypes as C

import numpy as np
import torch

from . import build as _build

_c = None
_ref = None


def _clib():
    global _c
    if _c is None:
        _c = C.CDLL(_build.build_c())
    return _c


def ref_available():
    return _build.build_ref() is not None


def _reflib():
    global _ref
    if _ref is None:
        path = _build.build_ref()
        if path is None:
            raise RuntimeError("oracle/_ref is not built and /root/reference is absent")
        _ref = C.CDLL(path)
    return _ref


def _p(t):
    return C.c_void_p(t.data_ptr())


def _prep(feat, rois):
    feat = feat.detach().to(torch.float32).contiguous()
    rois = rois.detach().to(torch.float32).contiguous()
    assert feat.dim() == 4 and rois.dim() == 2 and rois.size(1) == 5
    return feat, rois


def roi_pool_forward(feat, rois, spatial_scale, output_size):
    """-> (out (R,C,ph,pw) fp32, argmax int32)."""
    feat, rois = _prep(feat, rois)
    ph, pw = output_size
    N, Cc, H, W = feat.shape
    R = rois.shape[0]
    out = torch.zeros(R, Cc, ph, pw, dtype=torch.float32)
    arg = torch.zeros(R, Cc, ph, pw, dtype=torch.int32)
    if R:
        _clib().roi_pool_forward(_p(feat), C.c_float(spatial_scale), Cc, H, W, ph, pw, _p(rois), R,
                                 _p(out), _p(arg))
    return out, arg


def roi_pool_backward(grad, rois, argmax, input_shape):
    N, Cc, H, W = input_shape
    grad = grad.detach().to(torch.float32).contiguous()
    rois = rois.detach().to(torch.float32).contiguous()
    R, _, ph, pw = grad.shape
    gi = torch.zeros(N, Cc, H, W, dtype=torch.float32)
    if R:
        _clib().roi_pool_backward(_p(grad), _p(argmax.contiguous()), R, Cc, H, W, ph, pw, _p(gi), _p(rois))
    return gi


def roi_align_forward(feat, rois, spatial_scale, output_size, sampling_ratio, aligned):
    feat, rois = _prep(feat, rois)
    ph, pw = output_size
    N, Cc, H, W = feat.shape
    R = rois.shape[0]
    out = torch.zeros(R, Cc, ph, pw, dtype=torch.float32)
    if R:
        _clib().roi_align_forward(_p(feat), C.c_float(spatial_scale), Cc, H, W, ph, pw,
                                  int(sampling_ratio), int(bool(aligned)), _p(rois), R, _p(out))
    return out


def roi_align_backward(grad, rois, spatial_scale, sampling_ratio, aligned, input_shape):
    N, Cc, H, W = input_shape
    grad = grad.detach().to(torch.float32).contiguous()
    rois = rois.detach().to(torch.float32).contiguous()
    R, _, ph, pw = grad.shape
    gi = torch.zeros(N, Cc, H, W, dtype=torch.float32)
    if R:
        _clib().roi_align_backward(_p(grad), C.c_float(spatial_scale), Cc, H, W, ph, pw,
                                   int(sampling_ratio), int(bool(aligned)), _p(rois), R, _p(gi))
    return gi


class _RoIPoolFn(torch.autograd.Function):
    """Differentiable front of the C restatement (forward + the argmax scatter of ROILoopPool_cpu.cpp:82-123): the
    oracle's pooling when a backbone stage is trainable (MODEL.BACKBONE.FREEZE_AT < 5)."""

    @staticmethod
    def forward(ctx, feat, rois, spatial_scale, output_size):
        out, arg = roi_pool_forward(feat, rois, spatial_scale, output_size)
        ctx.save_for_backward(rois, arg)
        ctx.shape = tuple(feat.shape)
        return out

    @staticmethod
    def backward(ctx, grad):
        rois, arg = ctx.saved_tensors
        return roi_pool_backward(grad, rois, arg, ctx.shape), None, None, None


class _RoIAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, spatial_scale, output_size, sampling_ratio, aligned):
        ctx.save_for_backward(rois)
        ctx.cfg = (spatial_scale, sampling_ratio, aligned, tuple(feat.shape))
        return roi_align_forward(feat, rois, spatial_scale, output_size, sampling_ratio, aligned)

    @staticmethod
    def backward(ctx, grad):
        (rois,) = ctx.saved_tensors
        scale, sr, aligned, shape = ctx.cfg
        return roi_align_backward(grad, rois, scale, sr, aligned, shape), None, None, None, None, None


def roi_pool(feat, rois, spatial_scale, output_size):
    return _RoIPoolFn.apply(feat, rois, spatial_scale, tuple(output_size))


def roi_align(feat, rois, spatial_scale, output_size, sampling_ratio, aligned):
    return _RoIAlignFn.apply(feat, rois, spatial_scale, tuple(output_size), sampling_ratio, aligned)


# ---- the reference's own compiled RoIPool (oracle/_ref) ----
def ref_roi_pool_forward(feat, rois, spatial_scale, output_size):
    feat, rois = _prep(feat, rois)
    ph, pw = output_size
    N, Cc, H, W = feat.shape
    R = rois.shape[0]
    out = torch.zeros(R, Cc, ph, pw, dtype=torch.float32)
    arg = torch.zeros(R, Cc, ph, pw, dtype=torch.int32)
    if R:
        _reflib().ref_roi_pool_forward(_p(feat), N, Cc, H, W, _p(rois), R, C.c_float(spatial_scale),
                                       ph, pw, _p(out), _p(arg))
    return out, arg


def ref_roi_pool_backward(grad, rois, argmax, spatial_scale, input_shape):
    N, Cc, H, W = input_shape
    grad = grad.detach().to(torch.float32).contiguous()
    rois = rois.detach().to(torch.float32).contiguous()
    R, _, ph, pw = grad.shape
    gi = torch.zeros(N, Cc, H, W, dtype=torch.float32)
    if R:
        _reflib().ref_roi_pool_backward(_p(grad), _p(rois), _p(argmax.contiguous()), R,
                                        C.c_float(spatial_scale), ph, pw, N, Cc, H, W, _p(gi))
    return gi


def nms_segments(boxes, seg_offsets, iou_threshold, max_keep=0, valid=None):
    """oracle/det_ops_ref.c: greedy NMS per segment over score-sorted boxes.
    -> list (per segment) of int64 tensors of kept positions relative to the segment start."""
    boxes = boxes.detach().to(torch.float32).contiguous()
    seg = torch.as_tensor(seg_offsets, dtype=torch.int32).contiguous()
    G, N = seg.numel() - 1, boxes.shape[0]
    keep = torch.zeros(max(N, 1), dtype=torch.int32)
    count = torch.zeros(max(G, 1), dtype=torch.int32)
    v = None
    if valid is not None:
        v = valid.detach().to(torch.uint8).contiguous()
    if G:
        _clib().nms_segments(_p(boxes), _p(seg), _p(v) if v is not None else None, G, C.c_float(iou_threshold),
                             int(max_keep), _p(keep), _p(count))
    return [keep[int(seg[g]):int(seg[g]) + int(count[g])].to(torch.int64) for g in range(G)]


def roi_loop_pool_forward(feat, rois, spatial_scale, output_size, context_ratio=1.8):
    """The reference's 3-output ROILoopPool (restated from its CUDA kernel) -> (out (3R,C,ph,pw), argmax int32)."""
    feat, rois = _prep(feat, rois)
    ph, pw = output_size
    N, Cc, H, W = feat.shape
    R = rois.shape[0]
    out = torch.zeros(3 * R, Cc, ph, pw, dtype=torch.float32)
    arg = torch.full((3 * R, Cc, ph, pw), -1, dtype=torch.int32)
    if R:
        _clib().roi_loop_pool_forward(_p(feat), C.c_float(spatial_scale), Cc, H, W, ph, pw, _p(rois), R,
                                      C.c_float(context_ratio), _p(out), _p(arg))
    return out, arg
