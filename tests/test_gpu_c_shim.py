"""`wsovod_amd._C` = the reference's pybind module surface (layers/vision.cpp:9-12) over the C-ABI library,
driven through an autograd.Function of the same pattern as the reference's wrapper (layers/roi_loop_pool.py:9-35:
save_for_backward(roi, argmax), mark_non_differentiable(argmax), once_differentiable backward)."""
import pytest
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from oracle import roi_ops
from tests.util import random_rois

pytestmark = pytest.mark.gpu


class _LoopPool(Function):
    @staticmethod
    def forward(ctx, feat, roi, size, scale):
        from wsovod_amd import _C

        ctx.size, ctx.scale, ctx.shape = size, scale, feat.size()
        out, argmax = _C.roi_loop_pool_forward(feat, roi, scale, size[0], size[1])
        ctx.save_for_backward(roi, argmax)
        ctx.mark_non_differentiable(argmax)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        from wsovod_amd import _C

        roi, argmax = ctx.saved_tensors
        n, c, h, w = ctx.shape
        return _C.roi_loop_pool_backward(grad_out, roi, argmax, ctx.scale, ctx.size[0], ctx.size[1], n, c, h, w), \
            None, None, None


@pytest.mark.parametrize("size", [(7, 7), (3, 5)])
def test_c_shim_forward_backward_match_oracle(gpu, size):
    feat = torch.randn(2, 12, 38, 50, generator=torch.Generator().manual_seed(3))
    rois = random_rois(40, 2, 300, 400, seed=9)
    want, want_arg = roi_ops.roi_loop_pool_forward(feat, rois, 0.125, size)
    x = feat.to(gpu).requires_grad_(True)
    out = _LoopPool.apply(x, rois.to(gpu), size, 0.125)
    assert out.shape == (120, 12) + size and out.dtype == torch.float32
    assert torch.equal(out.detach().cpu(), want)  # max pooling copies values: bit-exact
    g = torch.randn(out.shape, generator=torch.Generator().manual_seed(4))
    # non-contiguous upstream gradient: the reference reads grad strides (ROILoopPool_cuda.cu:358-361)
    g_dev = g.to(gpu).permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)
    assert not g_dev.is_contiguous() or size[0] == size[1] == 1
    out.backward(g_dev)
    gi = roi_ops.roi_pool_backward(g, rois.repeat(3, 1), want_arg, tuple(feat.shape))
    torch.testing.assert_close(x.grad.cpu(), gi, rtol=1e-5, atol=1e-5)  # fp32 atomics: summation order differs


def test_c_shim_conventions(gpu):
    from wsovod_amd import _C

    feat = torch.randn(1, 4, 10, 12, device=gpu)
    out, arg = _C.roi_loop_pool_forward(feat, torch.zeros(0, 5, device=gpu), 0.125, 7, 7)
    assert out.shape == (0, 4, 7, 7) and arg.dtype == torch.int32
    gi = _C.roi_loop_pool_backward(torch.zeros(0, 4, 7, 7, device=gpu), torch.zeros(0, 5, device=gpu), arg, 0.125, 7, 7,
                                   1, 4, 10, 12)
    assert gi.shape == (1, 4, 10, 12) and float(gi.abs().sum()) == 0.0
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        _C.roi_loop_pool_forward(feat.cpu(), torch.zeros(1, 5), 0.125, 7, 7)
    with pytest.raises(RuntimeError, match="same type"):
        _C.roi_loop_pool_forward(feat, torch.zeros(1, 5, device=gpu, dtype=torch.float64), 0.125, 7, 7)
    with pytest.raises(RuntimeError, match="outside the hot path"):
        _C.csc_forward()
    # bf16 features: output in the input's dtype, values = the fp32 pool of the same bf16 numbers
    rois = random_rois(8, 1, 80, 96, seed=2)
    fb = feat.to(torch.bfloat16)
    ob, ab = _C.roi_loop_pool_forward(fb, rois.to(gpu).to(torch.bfloat16), 0.125, 7, 7)
    assert ob.dtype == torch.bfloat16 and ab.shape == (24, 4, 7, 7)
