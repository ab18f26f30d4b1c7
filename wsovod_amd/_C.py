"""`wsovod._C` replacement: the reference's native-op module surface over the C-ABI library.

The reference binds its CUDA op with pybind (`/root/reference/wsovod/layers/vision.cpp:9-12`):

    roi_loop_pool_forward(input, rois, spatial_scale, pooled_height, pooled_width) -> (output, argmax)
    roi_loop_pool_backward(grad, rois, argmax, spatial_scale, pooled_height, pooled_width,
                           batch_size, channels, height, width) -> grad_input

(`layers/ROILoopPool/ROILoopPool.h:48-106`, host side `ROILoopPool_cuda.cu:250-390`).  A maintainer switches
`from wsovod import _C` to `from wsovod_amd import _C` in `layers/roi_loop_pool.py:6` and the autograd wrapper there
(:9-35) runs unchanged.  Same positional signatures, shapes, dtypes, zero-initialised outputs, early return on empty
inputs and RuntimeError on CPU tensors / mixed dtypes (AT_ASSERTM / checkAllSameType).  `csc_forward`
(`vision.cpp:12`) is outside the hot path and raises.
"""
import torch

from .layers import hip_ops as H


def _check_cuda(**tensors):
    for name, t in tensors.items():
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA tensor")  # ROILoopPool_cuda.cu:258-259
    devs = {t.device for t in tensors.values()}
    if len(devs) != 1:
        raise RuntimeError(f"expected all tensors on one GPU, got {sorted(map(str, devs))}")  # checkAllSameGPU


def roi_loop_pool_forward(input, rois, spatial_scale, pooled_height, pooled_width):
    """(N,C,H,W) features + (R,5) [batch_idx,x0,y0,x1,y1] -> (output (3R,C,ph,pw) in input's dtype =
    [region | frame | context], argmax int32 of the same shape; -1 marks an empty bin)."""
    _check_cuda(input=input, rois=rois)
    if input.dtype != rois.dtype:
        raise RuntimeError(f"expected input and rois of the same type, got {input.dtype} and {rois.dtype}")  # :264
    num_rois, channels = rois.size(0), input.size(1)
    if num_rois * channels * pooled_height * pooled_width == 0:  # :288-291
        shape = (num_rois * 3, channels, pooled_height, pooled_width)
        return input.new_zeros(shape), torch.zeros(shape, dtype=torch.int32, device=input.device)
    feat = input if (input.is_contiguous() or input.is_contiguous(memory_format=torch.channels_last)) \
        else input.contiguous()
    out, argmax = H.roi_loop_pool_forward(feat, rois.contiguous(), float(spatial_scale),
                                          (int(pooled_height), int(pooled_width)))
    return out.to(input.dtype), argmax


def roi_loop_pool_backward(grad, rois, argmax, spatial_scale, pooled_height, pooled_width, batch_size, channels,
                           height, width):
    """Scatter-add of grad (3R,C,ph,pw; any strides) through argmax -> grad_input (N,C,H,W) in grad's dtype
    (`RoILoopPoolBackward`, ROILoopPool_cuda.cu:207-243: pooled row n belongs to rois[n % R])."""
    _check_cuda(grad=grad, rois=rois, argmax=argmax)
    if grad.dtype != rois.dtype:
        raise RuntimeError(f"expected grad and rois of the same type, got {grad.dtype} and {rois.dtype}")  # :336
    shape = (int(batch_size), int(channels), int(height), int(width))
    if grad.numel() == 0:  # :353-356
        return grad.new_zeros(shape)
    reps = grad.size(0) // max(rois.size(0), 1)
    gi = H.roi_pool_backward(grad, rois.repeat(reps, 1), argmax.contiguous(), shape)
    return gi.to(grad.dtype)


def csc_forward(*args, **kwargs):
    raise RuntimeError("wsovod_amd._C.csc_forward: the CSC op is outside the hot path (SURVEY.md 2.2); not built")
