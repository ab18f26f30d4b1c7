"""WSL ResNet ("DC5", output stride 8) backbone on the HIP implicit-GEMM convolution.

Host-side mirror of /root/reference/wsovod/modeling/backbone/resnet_wsl.py: same classes
(`BasicStem`, `BasicBlock`, `BottleneckBlock`, `ResNet`, `build_wsl_resnet_backbone`), same
constructor arguments, same parameter/buffer names (so reference d2 checkpoints load), same
`forward(x) -> {"res5": Tensor}` / `output_shape()` / `freeze()` surface.  The compute is not
torch: every conv (+ folded FrozenBN affine + ReLU + residual add) is one launch of the MFMA
implicit-GEMM kernel over NHWC activations, pools are the NHWC max-pool kernel.  The returned
feature map is logically NCHW but stored NHWC (torch.channels_last), which the ROIPooler reads
directly.

Scope: every shipped WSR config freezes the whole backbone (FREEZE_AT: 5): the HIP kernels are the
FORWARD of every conv.  A trainable stage (FREEZE_AT < 5, resnet_wsl.py:530-552) runs its forward on those kernels and --
since round 6 -- its BACKWARD too (`_TrainableStage`, `_TrainableStem`): the stage's activations are recomputed by the
forward kernels, ReLU masks and pool routing come from those bits, input gradients are implicit-GEMM convs on the rotated
weights, weight gradients the transposed-read contraction over im2col rows (stem conv1: over the normalised im2col rows of
the image).  The "bf16x3" modes keep the earlier form (the stage re-evaluated with torch's GPU convolution under autograd).
Never a CPU path; tests/test_gpu_freeze_at.py pins it to the reference (G19), to the oracle (FREEZE_AT = 0) and to the
torch re-evaluation (FREEZE_AT 1 - 4).
"""
import os
import warnings

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from ..config import BACKBONE_REGISTRY
from ..layers import hip_ops as H
from ..structures import ShapeSpec

__all__ = ["BasicStem", "BasicBlock", "BottleneckBlock", "ResNet", "FrozenBatchNorm2d", "Conv2d",
           "build_wsl_resnet_backbone", "make_stage"]


class FrozenBatchNorm2d(nn.Module):
    """detectron2.layers.FrozenBatchNorm2d: fixed statistics + affine, eps 1e-5 (buffers, not params)."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def scale_shift(self):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        return scale, self.bias - self.running_mean * scale


def get_norm(norm, out_channels):
    if norm is None or (isinstance(norm, str) and len(norm) == 0):
        return None
    if norm == "FrozenBN":
        return FrozenBatchNorm2d(out_channels)
    raise NotImplementedError(f"wsovod_amd backbone supports NORM 'FrozenBN' or '' (got {norm!r})")


def c2_msra_fill(module):
    nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


class Conv2d(nn.Module):
    """Parameter holder with detectron2.layers.Conv2d's state-dict layout (weight, bias, norm.*)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=True, norm=None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation = kernel_size, stride, padding, dilation
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.norm = norm
        self._folded = None

    def folded(self, dtype, cin_pad=None):
        """(weight [Cout][kh*kw*Cin] in `dtype` with the FrozenBN scale folded in, fp32 bias)."""
        key = (dtype, cin_pad, self.weight._version, self.weight.device)
        if self._folded is None or self._folded[0] != key:
            with torch.no_grad():
                w = self.weight.float()
                b = self.bias.float() if self.bias is not None else torch.zeros(self.out_channels, device=w.device)
                if self.norm is not None:
                    scale, shift = self.norm.scale_shift()
                    w = w * scale.view(-1, 1, 1, 1)
                    b = b * scale + shift
                w = w.permute(0, 2, 3, 1)  # [Cout][kh][kw][Cin]
                if cin_pad is not None and cin_pad != self.in_channels:
                    w = F.pad(w, (0, cin_pad - self.in_channels))
                wq = w.reshape(self.out_channels, -1).to(dtype).contiguous()
                self._folded = (key, wq, b.contiguous())
        return self._folded[1], self._folded[2]


def _x2():
    """MODEL.HIP.PRECISION = "parity": feature maps between the convs are bf16x2 NHWC tensors (hip_ops.X2)."""
    return H.x3_active() == "x2"


def _folded_x2(conv, cin_pad=None):
    """bf16x2 encoding of the folded fp32 weight rows ([Cout][kh*kw*Cin], Cin a multiple of 32: a tap's channels are
    whole 32-value groups, so encoding the flat rows encodes every tap), cached with the fold."""
    w, b = conv.folded(torch.float32, cin_pad=cin_pad)
    return H.x2_cached(w), b


def _folded_with_shortcut(conv, shortcut, dtype):
    """[W (kh*kw*Cin) | Wshortcut (Cin2)] rows + the summed folded biases: the operand of the conv that contracts the
    block's 1x1 projection shortcut in the same accumulation (wsovod_gemm_desc.A2)."""
    key = (dtype, conv.weight._version, shortcut.weight._version, conv.weight.device)
    c = getattr(conv, "_folded_sc", None)
    if c is None or c[0] != key:
        w, b = conv.folded(dtype)
        ws, bs = shortcut.folded(dtype)
        conv._folded_sc = c = (key, torch.cat([w, ws], dim=1).contiguous(), (b + bs).contiguous())
    return c[1], c[2]


CONV_MAX_OPERAND_BYTES = (1 << 31) - 1  # one buffer resource per NHWC operand (tests lower it to exercise the image blocks)


def hip_conv(x, conv, relu=False, residual=None, pool2=False, shortcut=None, out_fp32=False):
    """x: (N,H,W,Cin) NHWC contiguous in the compute dtype -> (N,Ho,Wo,Cout); with pool2 the MaxPool2d(2, 2) that
    follows the conv in the stem / block tail is applied too -> (N,Ho//2,Wo//2,Cout).  The 64-channel bf16 kernel pools
    in its epilogue (the full-resolution map is never written); every other conv is followed by the pool kernel."""
    N, Hh, Ww, Cin = x.shape
    k, s, p, d = conv.kernel_size, conv.stride, conv.padding, conv.dilation
    Ho = (Hh + 2 * p - d * (k - 1) - 1) // s + 1
    Wo = (Ww + 2 * p - d * (k - 1) - 1) // s + 1
    # the kernels address the NHWC input through one buffer resource (< 2 GiB): larger batches (> 139 images of
    # 800x600 at the 64-channel stem maps) go through in image blocks -- images are independent
    per_image = max(Hh * Ww * Cin, Ho * Wo * conv.out_channels) * x.element_size()
    if H.x3_active() in ("full", "fwd") and x.dtype == torch.float32:
        per_image = max(per_image, Hh * Ww * 3 * Cin * 2)  # the operand the kernel addresses is the [hi | hi | lo] bf16 split
    max_n = max(1, CONV_MAX_OPERAND_BYTES // max(per_image, 1))
    if N > max_n:
        parts = []
        for i in range(0, N, max_n):
            j = min(N, i + max_n)
            parts.append(hip_conv(x[i:j], conv, relu=relu, residual=None if residual is None else residual[i:j],
                                  pool2=pool2, shortcut=None if shortcut is None else (shortcut[0][i:j], shortcut[1]),
                                  out_fp32=out_fp32))
        return torch.cat(parts)
    geom = dict(n_img=N, H=Hh, W=Ww, Cin=Cin, Ho=Ho, Wo=Wo, KH=k, KW=k, stride=s, pad=p, dil=d)
    if _x2() and H.mx_of(x):
        # "parity_mx": x (and residual / shortcut input) are unit-scale f16mx maps; fp16 hi*hi + block-scaled e4m3 cross terms
        # on the f16mx weights (per-row scales, encoded once: the stages are frozen or re-encoded per optimizer step); the
        # output is f16mx again, or real fp32 for the map that leaves the backbone
        assert not pool2
        fmt = torch.float32 if out_fp32 else H.MX
        if shortcut is not None:
            x2in, sc = shortcut
            assert residual is None and H.mx_of(x2in) and x2in.shape[:3] == (N, Ho, Wo) and x2in.shape[3] == sc.in_channels
            wq, b = _folded_with_shortcut(conv, sc, torch.float32)
            wm, ws = H.mx_cached(wq)
            out = H.gemm_mx(x, None, wm, ws, conv=geom, A2=x2in, bias=b, relu=relu, out_dtype=fmt)
        else:
            wq, b = conv.folded(torch.float32, cin_pad=Cin)
            wm, ws = H.mx_cached(wq)
            res2d = residual.view(N * Ho * Wo, conv.out_channels) if residual is not None else None
            assert res2d is None or H.mx_of(residual)
            out = H.gemm_mx(x, None, wm, ws, conv=geom, bias=b, relu=relu, residual=res2d,
                            residual_fmt=H.MX if res2d is not None else None, out_dtype=fmt)
        out = out.view(N, Ho, Wo, conv.out_channels)
        if not out_fp32:
            out._mx = True
        return out
    if _x2():
        # x (and residual / shortcut input) are bf16x2 maps; three-MFMA products on the bf16x2 weights; the output is
        # bf16x2 again, or real fp32 for the map that leaves the backbone (out_fp32)
        fmt = torch.float32 if out_fp32 else H.X2
        if shortcut is not None:
            x2in, sc = shortcut
            assert residual is None and not pool2 and x2in.shape[:3] == (N, Ho, Wo) and x2in.shape[3] == sc.in_channels
            wq, b = _folded_with_shortcut(conv, sc, torch.float32)
            out = H.gemm_nt(x, H.x2_cached(wq), conv=geom, x2=True, bias=b, relu=relu, out_dtype=fmt, A2=x2in)
        else:
            wq, b = _folded_x2(conv, cin_pad=Cin)
            res2d = residual.view(N * Ho * Wo, conv.out_channels) if residual is not None else None
            if pool2 and not out_fp32 and Cin == 64 and conv.out_channels == 64 and (k, s, p, d) == (3, 1, 1, 1):
                geom["pool"] = 2  # the 64-channel halo kernel pools in its epilogue: the full-resolution map is never written
                out = H.gemm_nt(x, wq, conv=geom, x2=True, bias=b, relu=relu, residual=res2d, residual_x2=True, out_dtype=fmt)
                return out.view(N, Ho // 2, Wo // 2, conv.out_channels)
            out = H.gemm_nt(x, wq, conv=geom, x2=True, bias=b, relu=relu, residual=res2d, residual_x2=True, out_dtype=fmt)
        out = out.view(N, Ho, Wo, conv.out_channels)
        return H.maxpool2x2_nhwc(out, 2, x2=not out_fp32) if pool2 else out
    wq, b = conv.folded(x.dtype, cin_pad=Cin)
    if shortcut is not None:
        # `shortcut` = (block input, its 1x1 projection conv): out = conv(x) + projection(input), one accumulation
        x2, sc = shortcut
        assert residual is None and not pool2 and x2.shape[:3] == (N, Ho, Wo) and x2.shape[3] == sc.in_channels
        wq, b = _folded_with_shortcut(conv, sc, x.dtype)
        out = H.gemm_nt(x, wq, conv=geom, bias=b, relu=relu, out_dtype=x.dtype, A2=x2)
        return out.view(N, Ho, Wo, conv.out_channels)
    res2d = residual.view(N * Ho * Wo, conv.out_channels) if residual is not None else None
    fused = (pool2 and x.dtype == torch.bfloat16 and Cin == 64 and conv.out_channels == 64 and (k, s, p, d) == (3, 1, 1, 1)
             and (residual is None or residual.dtype == torch.bfloat16))
    if fused:
        geom["pool"] = 2
        out = H.gemm_nt(x, wq, conv=geom, bias=b, relu=relu, residual=res2d, out_dtype=x.dtype)
        return out.view(N, Ho // 2, Wo // 2, conv.out_channels)
    out = H.gemm_nt(x, wq, conv=geom, bias=b, relu=relu, residual=res2d, out_dtype=x.dtype)
    out = out.view(N, Ho, Wo, conv.out_channels)
    return H.maxpool2x2_nhwc(out, 2) if pool2 else out


def _fusable_shortcut(sc, x):
    """The block's projection shortcut can ride in its last conv's accumulation: 1x1, stride 1, a whole number of
    K-steps of channels, bf16 / exact-fp32 operands (the bf16x3 modes split their operands and keep the separate launch)."""
    if sc is None or (H.x3_active() and not _x2()) or os.environ.get("WSOVOD_FUSE_SHORTCUT", "1") == "0":
        return False
    kstep = 64 if x.dtype == torch.bfloat16 else 32  # (bf16x2: 32 values = 64 bf16 slots)
    return sc.kernel_size == 1 and sc.stride == 1 and sc.padding == 0 and sc.in_channels % kstep == 0 and x.is_contiguous()


class CNNBlockBase(nn.Module):
    def __init__(self, in_channels, out_channels, stride):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        return self


class _PoolMixin:
    def _init_pool(self, has_pool, pool_stride):
        self.has_pool, self.pool_stride = has_pool, pool_stride

    def _pool(self, out):
        if not self.has_pool:
            return out
        # stride 1: ZeroPad2d((0,1,0,1)) + MaxPool2d(2, 1); else MaxPool2d(2, stride)  (resnet_wsl.py:85-92)
        return H.maxpool2x2_nhwc(out, self.pool_stride, zero_pad_br=self.pool_stride == 1, x2=_x2())


class BasicBlock(CNNBlockBase, _PoolMixin):
    """resnet_wsl.py:24-110: two 3x3 convs; the block stride lives in the trailing max pool."""

    def __init__(self, in_channels, out_channels, *, stride=1, norm="BN", dilation=1, has_pool=False):
        super().__init__(in_channels, out_channels, stride)
        self._init_pool(has_pool, stride)
        stride = 1
        if in_channels != out_channels:
            self.shortcut = Conv2d(in_channels, out_channels, 1, stride=stride, bias=False,
                                   norm=get_norm(norm, out_channels))
        else:
            self.shortcut = None
        self.conv1 = Conv2d(in_channels, out_channels, 3, stride=stride, padding=dilation, dilation=dilation,
                            bias=False, norm=get_norm(norm, out_channels))
        self.conv2 = Conv2d(out_channels, out_channels, 3, stride=1, padding=dilation, dilation=dilation,
                            bias=False, norm=get_norm(norm, out_channels))
        for layer in [self.conv1, self.conv2, self.shortcut]:
            if layer is not None:
                c2_msra_fill(layer)

    def forward(self, x):
        last = getattr(self, "_emits_fp32", False) and not self.has_pool  # "parity": the map that leaves the backbone is real fp32
        out = hip_conv(x, self.conv1, relu=True)
        if _fusable_shortcut(self.shortcut, x) and not (self.has_pool and self.pool_stride == 2):
            # projection shortcut contracted inside conv2 (K = 9*C + Cin): no separate 1x1 launch, its output is neither
            # written nor rounded nor re-read as a residual
            return self._pool(hip_conv(out, self.conv2, relu=True, shortcut=(x, self.shortcut), out_fp32=last))
        shortcut = hip_conv(x, self.shortcut) if self.shortcut is not None else x
        if self.has_pool and self.pool_stride == 2:  # stride-2 tail pool (res2): fused where the kernel has it
            return hip_conv(out, self.conv2, relu=True, residual=shortcut, pool2=True)
        out = hip_conv(out, self.conv2, relu=True, residual=shortcut, out_fp32=last)  # out += shortcut; relu
        return self._pool(out)


class BottleneckBlock(CNNBlockBase, _PoolMixin):
    """resnet_wsl.py:113-241: 1x1 -> 3x3 (dilated) -> 1x1 + shortcut."""

    def __init__(self, in_channels, out_channels, *, bottleneck_channels, stride=1, num_groups=1, norm="BN",
                 stride_in_1x1=False, dilation=1, has_pool=False):
        super().__init__(in_channels, out_channels, stride)
        if num_groups != 1:
            raise NotImplementedError("grouped convolution is not on the WSR hot path")
        self._init_pool(has_pool, stride)
        stride = 1
        if in_channels != out_channels:
            self.shortcut = Conv2d(in_channels, out_channels, 1, stride=stride, bias=False,
                                   norm=get_norm(norm, out_channels))
        else:
            self.shortcut = None
        stride_1x1, stride_3x3 = (stride, 1) if stride_in_1x1 else (1, stride)
        self.conv1 = Conv2d(in_channels, bottleneck_channels, 1, stride=stride_1x1, bias=False,
                            norm=get_norm(norm, bottleneck_channels))
        self.conv2 = Conv2d(bottleneck_channels, bottleneck_channels, 3, stride=stride_3x3, padding=dilation,
                            dilation=dilation, bias=False, norm=get_norm(norm, bottleneck_channels))
        self.conv3 = Conv2d(bottleneck_channels, out_channels, 1, bias=False, norm=get_norm(norm, out_channels))
        for layer in [self.conv1, self.conv2, self.conv3, self.shortcut]:
            if layer is not None:
                c2_msra_fill(layer)

    def forward(self, x):
        last = getattr(self, "_emits_fp32", False) and not self.has_pool
        out = hip_conv(x, self.conv1, relu=True)
        out = hip_conv(out, self.conv2, relu=True)
        if _fusable_shortcut(self.shortcut, x) and out.shape[:3] == x.shape[:3]:
            return self._pool(hip_conv(out, self.conv3, relu=True, shortcut=(x, self.shortcut), out_fp32=last))
        shortcut = hip_conv(x, self.shortcut) if self.shortcut is not None else x
        out = hip_conv(out, self.conv3, relu=True, residual=shortcut, out_fp32=last)
        return self._pool(out)


class BasicStem(CNNBlockBase):
    """resnet_wsl.py:361-421: conv3x3 s2 -> conv3x3 -> conv3x3 -> maxpool k2 s2 (stride 4)."""

    def __init__(self, in_channels=3, out_channels=64, norm="BN"):
        super().__init__(in_channels, out_channels, 4)
        self.conv1 = Conv2d(in_channels, out_channels, 3, stride=2, padding=1, bias=False,
                            norm=get_norm(norm, out_channels))
        self.conv2 = Conv2d(out_channels, out_channels, 3, stride=1, padding=1, bias=False,
                            norm=get_norm(norm, out_channels))
        self.conv3 = Conv2d(out_channels, out_channels, 3, stride=1, padding=1, bias=False,
                            norm=get_norm(norm, out_channels))
        for c in (self.conv1, self.conv2, self.conv3):
            c2_msra_fill(c)

    def _tail(self, x):
        x = hip_conv(x, self.conv2, relu=True)
        return hip_conv(x, self.conv3, relu=True, pool2=True)

    def forward(self, x):
        """x: NHWC with Cin zero-padded to the kernel's K-step (generic float entry)."""
        return self._tail(hip_conv(x, self.conv1, relu=True))

    def _im2col_weight(self, dtype):
        wq, b = self.conv1.folded(dtype)  # [Cout][27]
        wpad = getattr(self, "_w_im2col", None)
        if wpad is None or wpad[0] is not wq:
            w32 = torch.zeros((wq.size(0), 32), dtype=wq.dtype, device=wq.device)
            w32[:, :27] = wq
            self._w_im2col = wpad = (wq, w32)
        return wpad[1], b

    def forward_uint8(self, images_u8, sizes, pixel_mean, pixel_std):
        if _x2():  # "parity": the fused kernel on the bf16x2 encoding of the (64, 32) fp32 weight, bf16x2 output
            w32, b = self._im2col_weight(torch.float32)
            return self._tail(H.stem_conv1_x2(images_u8, sizes, pixel_mean, pixel_std, H.x2_cached(w32), b))
        w32, b = self._im2col_weight(torch.bfloat16)
        return self._tail(H.stem_conv1(images_u8, sizes, pixel_mean, pixel_std, w32, b))

    def forward_im2col(self, a, n, ho, wo):
        """a: (n*ho*wo, 32) fused normalise+im2col operand of conv1 (K = 27 padded to 32)."""
        assert self.in_channels == 3
        wq, b = self.conv1.folded(a.dtype)  # [Cout][27]
        wpad = getattr(self, "_w_im2col", None)
        if wpad is None or wpad[0] is not wq:
            w32 = torch.zeros((wq.size(0), 32), dtype=wq.dtype, device=wq.device)
            w32[:, :27] = wq
            self._w_im2col = wpad = (wq, w32)
        x = H.gemm_nt(a, wpad[1], bias=b, relu=True, out_dtype=a.dtype).view(n, ho, wo, self.out_channels)
        return self._tail(x)


# ---------------------------------------------------------------------------------------------------------------
# trainable stages (MODEL.BACKBONE.FREEZE_AT < 5): HIP forward, torch-autograd backward by re-evaluation
# ---------------------------------------------------------------------------------------------------------------
def _torch_conv(conv, x):
    """conv + FrozenBN as torch ops on NCHW fp32 (resnet_wsl.py / detectron2 Conv2d.forward): the backward's restatement."""
    y = F.conv2d(x, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation)
    if conv.norm is not None:
        scale, shift = conv.norm.scale_shift()
        y = y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    return y


def _torch_block(block, x):
    """resnet_wsl.py:94-110 (BasicBlock) / :224-241 (BottleneckBlock) in torch ops."""
    out = F.relu(_torch_conv(block.conv1, x))
    if isinstance(block, BottleneckBlock):
        out = _torch_conv(block.conv3, F.relu(_torch_conv(block.conv2, out)))
    else:
        out = _torch_conv(block.conv2, out)
    out = F.relu(out + (_torch_conv(block.shortcut, x) if block.shortcut is not None else x))
    if block.has_pool:  # resnet_wsl.py:85-92
        out = F.max_pool2d(F.pad(out, (0, 1, 0, 1)), 2, 1) if block.pool_stride == 1 else F.max_pool2d(out, 2, block.pool_stride)
    return out


def forward_precision(name):
    """MODEL.HIP.PRECISION "parity_train" = the "parity" forward (bf16x2 activations, three products) + a backward that
    keeps the split too (layers/functions.py:backward_split): every module sees "parity", the meta-arch sets the flag.
    "parity_mx" likewise: the parity forward with its big contractions on the f16mx kernels (hip_ops.mx_mode)."""
    return "parity" if name in ("parity_train", "parity_mx") else name


# ---------------------------------------------------------------------------------------------------------------
# round 6: the backward of a trainable stage on the HIP kernels themselves (MODEL.BACKBONE.FREEZE_AT = 1 .. 4; reference:
# resnet_wsl.py:94-110,221-241 under autograd, :530-552)
#   tail pool     res2 / res3: the gradient goes to the first maximum of every 2x2 window      wsovod_maxpool2x2_nhwc_backward
#   mask          dL/d(pre-activation) = dL/d(out) * [out > 0]              wsovod_mask_transpose
#   input grad    a k x k, stride-1, same-size conv IS a conv of the output gradient with the kernel rotated by 180 deg and
#                 its channel roles swapped ([Cin][kh'][kw'][Cout]): the implicit-GEMM kernel of the forward pass; 1x1: a GEMM
#   weight grad   dW'[co][tap][ci] = sum_p g[p][co] * x[p + tap][ci] = g^T @ im2col(x): the transposed-read contraction
#                 (wsovod_gemm_tn) over patch rows (wsovod_im2col_rows); FrozenBN folds w' = w * scale[co], so dw = dW' * scale
# Arithmetic: the precision's backward grade -- plain bf16 MFMA products with fp32 accumulation for "bf16" / "parity" (on
# the hi halves of bf16x2 maps), exact-fp32 MFMA for "fp32".  The stage's activations are RE-COMPUTED by the same forward
# kernels (bit-identical to the forward that produced the loss: the ReLU masks are the forward's own -- the fp32 torch
# re-evaluation this replaces could pick other winners where two candidates lie within the forward's rounding).
# ---------------------------------------------------------------------------------------------------------------
def _hip_backward_ok(stage, x3):
    if os.environ.get("WSOVOD_HIP_CONV_BACKWARD", "1") == "0" or x3 in ("full", "fwd"):
        return False  # (the bf16x3 modes keep their operands in fp32 tensors and split on the fly: torch re-evaluation)
    for b in stage.children():
        if not isinstance(b, (BasicBlock, BottleneckBlock)):
            return False
        for c in (b.conv1, b.conv2, getattr(b, "conv3", None), b.shortcut):
            if c is not None and (c.stride != 1 or c.bias is not None or 2 * c.padding != c.dilation * (c.kernel_size - 1)
                                  or c.in_channels % 64 or c.out_channels % 64):
                return False
    return True


WGRAD_PATCH_BYTES = 1 << 30  # (tests lower it to exercise the row blocks)


def _masked(dy, y, cd):
    """dL/d(pre-activation) of y = relu(.): (P, C) in the compute dtype `cd` (dy fp32, y the forward's own output map)."""
    C = dy.shape[-1]
    P = dy.numel() // C
    if y.dtype == torch.bfloat16 and dy.dtype != torch.bfloat16:
        dy = dy.to(torch.bfloat16)  # (the mask kernel takes dy in y's dtype; "bf16" precision: bf16 gradients anyway)
    return H.mask_transpose(dy.reshape(P, C), y.reshape(P, C), 1.0, cd, want_plain=True, want_t=False,
                            y_x2=(y.dtype == torch.float32 and _is_x2_map(y, C)))[0]


def _is_x2_map(t, channels):
    """A bf16x2 carrier and a real fp32 map have the same dtype and shape; the blocks tag what they emit."""
    return bool(getattr(t, "_x2_map", False))


def _conv_dgrad(g2d, conv, N, Hh, Ww, cd):
    """g2d: (N*H*W, Cout) in cd -> dL/d(input) (N*H*W, Cin) fp32."""
    k, d = conv.kernel_size, conv.dilation
    w, _ = conv.folded(torch.float32)  # [Cout][kh*kw*Cin]
    Co, Ci = conv.out_channels, conv.in_channels
    if k == 1:
        return H.gemm_nt(g2d, w.t().contiguous().to(cd), out_dtype=torch.float32)
    wt = w.view(Co, k, k, Ci).flip(1, 2).permute(3, 1, 2, 0).reshape(Ci, k * k * Co).contiguous().to(cd)
    geom = dict(n_img=N, H=Hh, W=Ww, Cin=Co, Ho=Hh, Wo=Ww, KH=k, KW=k, stride=1, pad=conv.padding, dil=d)
    return H.gemm_nt(g2d.view(N, Hh, Ww, Co), wt, conv=geom, out_dtype=torch.float32)


def _conv_wgrad(g2d, xin, conv, cd):
    """g2d (P, Cout) in cd; xin: the conv's NHWC input as the forward left it (bf16 / fp32 / bf16x2 carrier) -> dL/dw in
    the parameter's own layout (Cout, Cin, kh, kw), FrozenBN scale applied."""
    k = conv.kernel_size
    N, Hh, Ww, Ci = xin.shape
    P = N * Hh * Ww
    x2 = _is_x2_map(xin, Ci)
    # patch rows are materialised in blocks of at most ~1 GiB (the stem's 64-channel convs at 32 images would be 8.8 GB at
    # once): the blocks' contributions accumulate into dW
    step = P if k == 1 else max(64, (WGRAD_PATCH_BYTES // (k * k * Ci * xin.element_size())) // 64 * 64)
    dw = torch.empty((conv.out_channels, k * k * Ci), dtype=torch.float32, device=xin.device)
    for a in range(0, P, step):
        b = min(P, a + step)
        if k == 1:
            patches = xin.reshape(P, Ci)
        else:
            rows = torch.arange(a, b, dtype=torch.int64, device=xin.device)
            patches = H.im2col_rows(xin, rows, k, 1, conv.padding, conv.dilation)  # (b - a, k*k*Ci), tap-major then channel
        gb = g2d[a:b]
        if cd == torch.float32:
            Pp = (b - a + 63) // 64 * 64
            H.gemm_nt(H.transpose_cast(gb, torch.float32, ld_dst=Pp), H.transpose_cast(patches, torch.float32, ld_dst=Pp),
                      out=dw, accumulate=a > 0)
        else:
            H.gemm_tn(gb, patches, out=dw, accumulate=a > 0, q_x2=x2)  # of a bf16x2 map the hi halves are read
        del patches
    dw = dw.view(conv.out_channels, k, k, Ci).permute(0, 3, 1, 2)
    if conv.norm is not None:
        dw = dw * conv.norm.scale_shift()[0].view(-1, 1, 1, 1)
    return dw.contiguous()


def _block_forward_saving(block, x):
    """The block's forward on the HIP kernels (exactly BasicBlock.forward / BottleneckBlock.forward), keeping what the
    backward reads: -> (out, [inputs of conv1, conv2(, conv3)])."""
    last = getattr(block, "_emits_fp32", False) and not block.has_pool
    tag = lambda t, real_fp32=False: (setattr(t, "_x2_map", _x2() and not real_fp32) or t)
    tag(x)
    h = tag(hip_conv(x, block.conv1, relu=True))
    ins = [x, h]
    tail = block.conv2
    if isinstance(block, BottleneckBlock):
        h = tag(hip_conv(h, block.conv2, relu=True))
        ins.append(h)
        tail = block.conv3
    if _fusable_shortcut(block.shortcut, x) and h.shape[:3] == x.shape[:3] and \
            not (isinstance(block, BasicBlock) and block.has_pool and block.pool_stride == 2):  # (the forwards' own rules)
        out = hip_conv(h, tail, relu=True, shortcut=(x, block.shortcut), out_fp32=last)
    else:
        sc = hip_conv(x, block.shortcut) if block.shortcut is not None else x
        out = hip_conv(h, tail, relu=True, residual=sc, out_fp32=last)
    tag(out, real_fp32=last)
    if block.has_pool:
        # the map the tail pool reads (the forward's fused 64-channel conv + pool never writes it: same bits, gemm.hip) and
        # the pooled map the next block takes
        return tag(block._pool(out)), ins, out
    return out, ins, out


def _block_backward(block, ins, out, dy, cd, need_dx):
    """dy: dL/d(out) (N,H,W,C) fp32 -> (dL/d(block input) (N,H,W,Cin) fp32 or None, {conv module: dL/dw})."""
    N, Hh, Ww, _ = out.shape
    convs = [block.conv1, block.conv2] + ([block.conv3] if isinstance(block, BottleneckBlock) else [])
    grads = {}
    if block.has_pool:  # `out` is the map the tail pool read: route dy back through the pool first
        dy = H.maxpool2x2_nhwc_backward(out, dy.contiguous(), block.pool_stride, zero_pad_br=block.pool_stride == 1,
                                        x2=_is_x2_map(out, out.shape[-1]))
    g = _masked(dy.contiguous(), out, cd)  # through the block's last ReLU: gradient of conv_tail(h) + shortcut(x)
    g_tail = g
    for i in range(len(convs) - 1, -1, -1):
        conv, xin = convs[i], ins[i]
        if conv.weight.requires_grad:
            grads[conv] = _conv_wgrad(g, xin, conv, cd)
        if i == 0 and not need_dx:
            dx = None
            break
        dx = _conv_dgrad(g, conv, N, Hh, Ww, cd)  # fp32 (P, Cin of this conv)
        if i > 0:
            g = _masked(dx.view(N, Hh, Ww, -1), xin, cd)  # through the ReLU that produced this conv's input
    sc = block.shortcut
    if sc is not None:
        if sc.weight.requires_grad:
            grads[sc] = _conv_wgrad(g_tail, ins[0], sc, cd)
        if need_dx:
            dx = dx + _conv_dgrad(g_tail, sc, N, Hh, Ww, cd)
    elif need_dx:
        dx = dx + (g_tail.float() if g_tail.dtype != torch.float32 else g_tail)
    return (dx.view(N, Hh, Ww, -1) if dx is not None else None), grads


_WARNED_TRAINABLE = set()


def _warn_trainable_stage_once(name):
    if name not in _WARNED_TRAINABLE:
        _WARNED_TRAINABLE.add(name)
        warnings.warn(f"wsovod_amd: backbone stage {name} is trainable (MODEL.BACKBONE.FREEZE_AT < 5): the step leaves the "
                      "optimised path -- no frozen-forward overlap, no step graph, no backbone graph; its backward runs on "
                      "the HIP kernels too (the bf16x3 modes: a torch re-evaluation of the stage; DESIGN.md section 7)",
                      stacklevel=3)


class _TrainableStage(torch.autograd.Function):
    """One backbone stage with trainable weights.  forward: the HIP kernels (as for a frozen stage).  backward: the
    stage re-evaluated from its saved input in fp32 torch ops on the GPU under autograd -> d input, d weights.
    The re-evaluation is fp32 while the forward that produced the loss ran in the model's precision (bf16 / bf16x2): a ReLU
    mask or max-pool winner of the recomputation can differ from the forward's where two candidates lie within the
    forward's rounding, so the gradient is that of a slightly different function (bf16: up to ~15 % on single elements,
    tests/test_gpu_freeze_at.py; fp32 / parity: at the oracle's tolerance)."""

    @staticmethod
    def forward(ctx, stage, x3, x, *params):
        with torch.no_grad(), H.x3_mode(x3):
            y = stage(x)
        ctx.stage, ctx.x3 = stage, x3
        ctx.save_for_backward(x, *params)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, *params = ctx.saved_tensors
        if _hip_backward_ok(ctx.stage, ctx.x3):
            return _TrainableStage._backward_hip(ctx, dy, x, params)
        with torch.no_grad():  # the saved map in its on-device format (bf16x2 carrier / bf16 / fp32 NHWC) -> fp32 NCHW
            x32 = H.x2_decode(x.reshape(-1, x.shape[-1])).view(x.shape) if ctx.x3 == "x2" else x.float()
            x32 = x32.permute(0, 3, 1, 2).contiguous()
        need_dx = ctx.needs_input_grad[2]
        x32.requires_grad_(need_dx)
        with torch.enable_grad():
            y = x32
            for block in ctx.stage.children():
                y = _torch_block(block, y)
        wanted = ([x32] if need_dx else []) + [p for p in params if p.requires_grad]
        grads = list(torch.autograd.grad(y, wanted, dy.float().permute(0, 3, 1, 2), allow_unused=True))
        dx = grads.pop(0).permute(0, 2, 3, 1).contiguous().to(x.dtype if ctx.x3 != "x2" else torch.float32) if need_dx else None
        it = iter(grads)
        return (None, None, dx, *[next(it) if p.requires_grad else None for p in params])


class _TrainableStem(torch.autograd.Function):
    """MODEL.BACKBONE.FREEZE_AT = 0 (round 6): the stem with trainable weights.  forward: the fused uint8 -> conv1 kernel and
    the 64-channel convs as for a frozen stem.  backward: the stem's activations recomputed by the same kernels, then pool
    backward -> mask -> weight / input gradients of conv3 and conv2 as in the residual stages -> conv1's weight gradient as
    g^T @ (normalised im2col rows of the image, wsovod_stem_im2col); the image itself takes no gradient."""

    @staticmethod
    def forward(ctx, net, x3, images_u8, sizes, mean, std, *params):
        with torch.no_grad(), H.x3_mode(x3):
            out = net._stem_uint8(images_u8, sizes, mean, std)
        ctx.net, ctx.x3, ctx.norm = net, x3, (mean, std)
        ctx.save_for_backward(images_u8, sizes, *params)
        return out

    @staticmethod
    def backward(ctx, dy):
        images_u8, sizes, *params = ctx.saved_tensors
        net, stem = ctx.net, ctx.net.stem
        mean, std = ctx.norm
        cd = torch.float32 if (ctx.x3 is False and net.compute_dtype == torch.float32) else torch.bfloat16
        tag = lambda t, on: (setattr(t, "_x2_map", on) or t)
        with torch.no_grad():
            with H.x3_mode(ctx.x3):
                x2 = _x2()
                a1 = tag(net._stem_conv1(images_u8, sizes, mean, std), x2)
                a2 = tag(hip_conv(a1, stem.conv2, relu=True), x2)
                a3 = tag(hip_conv(a2, stem.conv3, relu=True), x2)  # (the forward pools in this conv's epilogue: same bits)
            grads = {}
            with H.x3_mode(False):
                N, Hh, Ww, _ = a3.shape
                g = H.maxpool2x2_nhwc_backward(a3, (dy.float() if dy.dtype != torch.float32 else dy).contiguous(), 2, x2=x2)
                g = _masked(g, a3, cd)
                for conv, xin, yin in ((stem.conv3, a2, a2), (stem.conv2, a1, a1)):
                    if conv.weight.requires_grad:
                        grads[id(conv.weight)] = _conv_wgrad(g, xin, conv, cd)
                    g = _masked(_conv_dgrad(g, conv, N, Hh, Ww, cd).view(N, Hh, Ww, -1), yin, cd)
                if stem.conv1.weight.requires_grad:
                    patches, _, _ = H.stem_im2col(images_u8, sizes, mean, std, cd)  # (P, 32): [kh][kw][cin] + 5 zero columns
                    if cd == torch.float32:
                        Pp = (patches.size(0) + 63) // 64 * 64
                        dw = H.gemm_nt(H.transpose_cast(g, torch.float32, ld_dst=Pp),
                                       H.transpose_cast(patches, torch.float32, ld_dst=Pp), out_dtype=torch.float32)
                    else:
                        dw = H.gemm_tn(g, patches)
                    dw = dw[:, :27].reshape(stem.conv1.out_channels, 3, 3, 3).permute(0, 3, 1, 2)
                    if stem.conv1.norm is not None:
                        dw = dw * stem.conv1.norm.scale_shift()[0].view(-1, 1, 1, 1)
                    grads[id(stem.conv1.weight)] = dw.contiguous()
        return (None, None, None, None, None, None, *[grads.get(id(p)) if p.requires_grad else None for p in params])


def _stage_backward_hip(ctx, dy, x, params):
    stage = ctx.stage
    cd = torch.float32 if (ctx.x3 is False and x.dtype == torch.float32) else torch.bfloat16
    blocks = list(stage.children())
    with torch.no_grad():
        with H.x3_mode(ctx.x3):  # the forward's own kernels again: bit-identical activations, hence the forward's own masks
            acts, cur = [], x
            for b in blocks:
                nxt, ins, out = _block_forward_saving(b, cur)
                acts.append((ins, out))
                cur = nxt
        grads = {}
        g = dy.float() if dy.dtype != torch.float32 else dy
        with H.x3_mode(False):
            for bi in range(len(blocks) - 1, -1, -1):
                ins, out = acts[bi]
                need_dx = bi > 0 or ctx.needs_input_grad[2]
                g, gb = _block_backward(blocks[bi], ins, out, g, cd, need_dx)
                grads.update({id(c.weight): v for c, v in gb.items()})
    dx = None
    if ctx.needs_input_grad[2] and g is not None:
        dx = g if x.dtype == torch.float32 else g.to(x.dtype)
    return (None, None, dx, *[grads.get(id(p)) if p.requires_grad else None for p in params])


_TrainableStage._backward_hip = staticmethod(_stage_backward_hip)


class ResNet(nn.Module):
    """resnet_wsl.py:424-607."""

    def __init__(self, stem, stages, num_classes=None, out_features=None, freeze_at=0, precision="bf16"):
        super().__init__()
        if num_classes is not None:
            raise NotImplementedError("classification head is not on the detection hot path")
        self.stem = stem
        self.num_classes = num_classes
        self.precision = precision
        current_stride = self.stem.stride
        self._out_feature_strides = {"stem": current_stride}
        self._out_feature_channels = {"stem": self.stem.out_channels}
        self.stage_names, self.stages = [], []
        if out_features is not None:
            num_stages = max([{"res2": 1, "res3": 2, "res4": 3, "res5": 4}.get(f, 0) for f in out_features])
            stages = stages[:num_stages]
        for i, blocks in enumerate(stages):
            assert len(blocks) > 0, len(blocks)
            name = "res" + str(i + 2)
            stage = nn.Sequential(*blocks)
            self.add_module(name, stage)
            self.stage_names.append(name)
            self.stages.append(stage)
            self._out_feature_strides[name] = current_stride = int(
                current_stride * np.prod([k.stride for k in blocks]))
            self._out_feature_channels[name] = blocks[-1].out_channels
        self.stage_names = tuple(self.stage_names)
        if out_features is None:
            out_features = [name]
        self._out_features = out_features
        assert len(self._out_features)
        if precision == "parity":  # bf16x2 maps inside, real fp32 for the one map that leaves the backbone
            if list(out_features) != [self.stage_names[-1]]:
                raise NotImplementedError('MODEL.HIP.PRECISION "parity" returns the last stage only (bf16x2 maps inside)')
            list(self.stages[-1].children())[-1]._emits_fp32 = True
        children = [x[0] for x in self.named_children()]
        for out_feature in self._out_features:
            assert out_feature in children, "Available children: {}".format(", ".join(children))
        self.freeze(freeze_at)

    @property
    def size_divisibility(self):
        return 0

    @property
    def compute_dtype(self):
        return torch.bfloat16 if self.precision == "bf16" else torch.float32  # "fp32" and "bf16x3" carry fp32 tensors

    def _check_frozen(self):
        """The generic float entry (`forward(x)`, a normalised float image) has no trainable-stem path: the stem's first conv
        trains through the fused uint8 entry (`forward_uint8`, what the meta-arch calls; _TrainableStem)."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.stem.parameters()):
            raise NotImplementedError(
                "wsovod_amd: MODEL.BACKBONE.FREEZE_AT = 0 (a trainable stem) trains through ResNet.forward_uint8 (the uint8 "
                "entry the meta-arch uses); the float entry ResNet.forward(x) runs the stem forward-only")

    @property
    def has_trainable_stage(self):
        """True when a residual stage is trainable (FREEZE_AT < 5): the backbone forward then reads weights the optimizer
        updates, and the trainers may no longer run it ahead of the previous step's update."""
        return any(p.requires_grad for p in self._param_list())

    def _param_list(self):
        cached = getattr(self, "_params_cache", None)
        if cached is None:  # the module tree is fixed after construction: walk it once, not every step
            cached = self._params_cache = list(self.parameters())
        return cached

    def _stage_params(self, stage):
        cache = self.__dict__.setdefault("_stage_params_cache", {})
        got = cache.get(id(stage))
        if got is None:  # (fixed after construction, as _param_list)
            got = cache[id(stage)] = list(stage.parameters())
        return got

    MX_MIN_TILES = int(os.environ.get("WSOVOD_MX_MIN_TILES", "200"))

    def _mx_from(self):
        """"parity_mx": index of the first stage that runs on the f16mx kernels -- the trailing run of FROZEN stages of
        BasicBlocks without pools whose convs are at least 256 channels wide (res4 / res5 of WSR_18: what the numerics gate
        covered, profiles/r06_mx_gate.md; the BottleneckBlocks of WSR_50's res4 / res5 -- 1x1 / 3x3 / 1x1, 256 - 2048 channels -- are
        the same contractions and take the same kernel: tests/test_gpu_full_size.py holds them to the same bar); len(stages) = none.  The f16mx kernel has ONE tile shape (256 x 256, a workgroup per
        CU): below ~200 tiles per conv (fewer than 8 images of 800 x 600) the bf16x2 path's smaller tiles / split-K forms win
        and the stages stay on it -- same precision mode, same bound (both formats were gated alone and together)."""
        first = len(self.stages)
        for i in range(len(self.stages) - 1, -1, -1):
            stage = self.stages[i]
            def block_ok(b):  # every conv of the block at least 256 wide, whole 32-value groups, no pool, a 1x1 / stride-1 shortcut
                if not isinstance(b, (BasicBlock, BottleneckBlock)) or b.has_pool:
                    return False
                convs = [c for c in (b.conv1, b.conv2, getattr(b, "conv3", None)) if c is not None]
                return (all(c.in_channels % 32 == 0 and c.out_channels % 32 == 0 and c.out_channels >= 256 and c.stride == 1
                            for c in convs)
                        and (b.shortcut is None or (b.shortcut.kernel_size == 1 and b.shortcut.stride == 1)))

            ok = all(block_ok(b) for b in stage.children())
            if not ok or any(p.requires_grad for p in self._stage_params(stage)):
                break
            first = i
        return first

    def _run(self, x):
        outputs = {}
        if "stem" in self._out_features:
            outputs["stem"] = x.permute(0, 3, 1, 2)
        mx_from = self._mx_from() if (H.mx_active() and list(self._out_features) == [self.stage_names[-1]]) else len(self.stages)
        for si, (name, stage) in enumerate(zip(self.stage_names, self.stages)):
            params = self._stage_params(stage)
            if si == mx_from and si > 0 and -(-(x.shape[0] * x.shape[1] * x.shape[2]) // 256) >= self.MX_MIN_TILES:
                # the map that crosses from the bf16x2 layers to the f16mx ones (enough 256-row tiles: see _mx_from)
                with torch.no_grad():
                    x = H.mx_from_x2(x)
                    x._mx = True
            if torch.is_grad_enabled() and any(p.requires_grad for p in params):
                _warn_trainable_stage_once(name)
                x = _TrainableStage.apply(stage, H.x3_active(), x, *params)
            else:
                with torch.no_grad():
                    x = stage(x)
            if name in self._out_features:
                outputs[name] = x.permute(0, 3, 1, 2)  # logical NCHW, NHWC memory (channels_last)
        return outputs

    def forward(self, x):
        """x: (N,C,H,W) normalised float image batch -> {name: (N,C',H/8,W/8) channels_last}."""
        assert x.dim() == 4, f"ResNet takes an input of shape (N, C, H, W). Got {x.shape} instead!"
        self._check_frozen()
        cd = self.compute_dtype
        x3 = {"bf16x3": "full", "bf16x3f": "fwd", "parity": "fwd"}.get(self.precision, False)  # (float entry: no bf16x2 stem)
        kstep = 64 if (cd == torch.bfloat16 or x3) else 32
        with torch.no_grad():
            xn = x.permute(0, 2, 3, 1).to(cd)
            xn = F.pad(xn, (0, kstep - xn.size(-1))).contiguous()  # Cin 3 -> one K-step (generic float entry)
        with H.x3_mode(x3):
            with torch.no_grad():
                xs = self.stem(xn)
            return self._run(xs)

    def forward_uint8(self, images_u8, sizes, pixel_mean, pixel_std, allow_graph=False):
        """Fused entry used by the meta-arch: uint8 canvas -> normalise + im2col -> stem conv1 GEMM.
        allow_graph: the caller consumes the maps before its next call with this shape (the training step's frozen
        forward): small batches may then come from a captured HIP graph, whose outputs are that graph's STATIC buffers
        -- overwritten by the next replay.  inference() / TTA keep the eager launches (fresh tensors)."""
        with H.x3_mode({"bf16x3": "full", "bf16x3f": "fwd", "parity": "x2"}.get(self.precision, False)):
            if allow_graph and self.graph_max_batch and images_u8.is_cuda and images_u8.size(0) <= self.graph_max_batch \
                    and not self.has_trainable_stage:
                g = self._graph_for(images_u8, sizes, pixel_mean, pixel_std)
                if g is not None:
                    return g(images_u8)
            return self._forward_uint8(images_u8, sizes, pixel_mean, pixel_std)

    # ---- the frozen forward as a HIP graph (small batches: the ~25 launches of the backbone cost more host time than
    # device time; one replay instead).  Opt-in (`graph_max_batch`, set by the overlapped trainer): the returned maps
    # are the graph's static buffers, valid until the next call with the same input shape ----
    graph_max_batch = 0
    GRAPH_CACHE = 4  # graphs kept (each holds the backbone's activations of its shape)
    GRAPH_AFTER = 3  # calls with a shape before it is captured

    def _graph_fingerprint(self):
        ps = getattr(self, "_graph_tensors", None)
        if ps is None:
            ps = self._graph_tensors = list(self.parameters()) + list(self.buffers())
        return sum(t._version for t in ps), ps[0].data_ptr() if ps else 0

    def _graph_for(self, images_u8, sizes, pixel_mean, pixel_std):
        from .._lib import PROFILING

        if PROFILING[0] or torch.cuda.is_current_stream_capturing():
            return None
        fp = self._graph_fingerprint()
        cache = self.__dict__.setdefault("_graphs", {})
        if any(v and v.fingerprint != fp for v in cache.values()):
            cache.clear()  # a weight changed (load_state_dict, broadcast): the folded copies the graphs point at are stale
        key = (tuple(images_u8.shape), sizes.data_ptr(), tuple(pixel_mean), tuple(pixel_std), H.x3_active(), H.mx_active())
        g = cache.get(key)
        if g is None:
            # capture on the third call with a shape: with multi-scale inputs most shapes never repeat, and a capture
            # costs three forwards
            seen = self.__dict__.setdefault("_graph_seen", {})
            if len(seen) > 64:
                seen.clear()
            seen[key] = seen.get(key, 0) + 1
            if seen[key] < self.GRAPH_AFTER:
                return None
            if len(cache) >= self.GRAPH_CACHE:
                cache.pop(next(iter(cache)))
            try:
                g = _BackboneGraph(self, images_u8, sizes, pixel_mean, pixel_std, fp)
            except Exception as e:  # noqa: BLE001 -- out of memory in the graph's pool, an API call refused under capture
                import warnings

                warnings.warn(f"wsovod_amd: HIP graph capture of the frozen backbone failed for input shape "
                              f"{tuple(images_u8.shape)} ({type(e).__name__}: {e}); this shape keeps the eager launches")
                g = False  # remembered: never retried for this key
            cache[key] = g
        return g or None

    def _stem_conv1(self, images_u8, sizes, pixel_mean, pixel_std):
        """relu(conv1 (normalised image)) as the stem's forward produces it."""
        stem = self.stem
        if (self.compute_dtype == torch.bfloat16 or _x2()) and stem.out_channels == 64 and stem.in_channels == 3:
            if _x2():
                w32, b = stem._im2col_weight(torch.float32)
                return H.stem_conv1_x2(images_u8, sizes, pixel_mean, pixel_std, H.x2_cached(w32), b)
            w32, b = stem._im2col_weight(torch.bfloat16)
            return H.stem_conv1(images_u8, sizes, pixel_mean, pixel_std, w32, b)
        a, ho, wo = H.stem_im2col(images_u8, sizes, pixel_mean, pixel_std, self.compute_dtype)
        w32, b = stem._im2col_weight(a.dtype)
        return H.gemm_nt(a, w32, bias=b, relu=True, out_dtype=a.dtype).view(images_u8.size(0), ho, wo, stem.out_channels)

    def _stem_uint8(self, images_u8, sizes, pixel_mean, pixel_std):
        if (self.compute_dtype == torch.bfloat16 or _x2()) and self.stem.out_channels == 64 and self.stem.in_channels == 3:
            # bf16: one kernel from the uint8 canvas to relu(conv1) (bit-identical to im2col + GEMM, no operand pass)
            return self.stem.forward_uint8(images_u8, sizes, pixel_mean, pixel_std)
        a, ho, wo = H.stem_im2col(images_u8, sizes, pixel_mean, pixel_std, self.compute_dtype)
        return self.stem.forward_im2col(a, images_u8.size(0), ho, wo)

    def _forward_uint8(self, images_u8, sizes, pixel_mean, pixel_std):
        stem_params = self._stage_params(self.stem)
        if torch.is_grad_enabled() and any(p.requires_grad for p in stem_params):
            # MODEL.BACKBONE.FREEZE_AT = 0: the stem trains too (its backward on the HIP kernels, _TrainableStem)
            x3 = H.x3_active()
            if x3 in ("full", "fwd") or not (self.stem.out_channels == 64 and self.stem.in_channels == 3):
                raise NotImplementedError("wsovod_amd: MODEL.BACKBONE.FREEZE_AT = 0 (a trainable stem) is supported in the "
                                          "bf16 / fp32 / parity precisions on the 3 -> 64 stem")
            _warn_trainable_stage_once("stem")
            xs = _TrainableStem.apply(self, x3, images_u8, sizes, pixel_mean, pixel_std, *stem_params)
        else:
            with torch.no_grad():
                xs = self._stem_uint8(images_u8, sizes, pixel_mean, pixel_std)
        return self._run(xs)

    def output_shape(self):
        return {name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name])
                for name in self._out_features}

    def freeze(self, freeze_at=0):
        if freeze_at >= 1:
            self.stem.freeze()
        for idx, stage in enumerate(self.stages, start=2):
            if freeze_at >= idx:
                for block in stage.children():
                    block.freeze()
        return self

    @staticmethod
    def make_stage(block_class, num_blocks, *, in_channels, out_channels, **kwargs):
        blocks = []
        for i in range(num_blocks):
            curr_kwargs = {}
            for k, v in kwargs.items():
                if k.endswith("_per_block"):
                    assert len(v) == num_blocks
                    curr_kwargs[k[: -len("_per_block")]] = v[i]
                else:
                    curr_kwargs[k] = v
            blocks.append(block_class(in_channels=in_channels, out_channels=out_channels, **curr_kwargs))
            in_channels = out_channels
        return blocks


class _BackboneGraph:
    """One captured frozen forward for one input shape: static input / output buffers, replayed per call."""

    def __init__(self, net, images_u8, sizes, pixel_mean, pixel_std, fingerprint):
        self.fingerprint = fingerprint
        self.sizes = sizes  # (kept alive: the kernels of the graph read it)
        self.static_in = torch.empty_like(images_u8)
        self.static_in.copy_(images_u8)
        main = torch.cuda.current_stream(images_u8.device)
        side = torch.cuda.Stream(device=images_u8.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):  # warm-up outside the capture: weight folds, function attributes, allocator
            for _ in range(2):
                net._forward_uint8(self.static_in, sizes, pixel_mean, pixel_std)
        main.wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: a HIP call from another thread (a DataLoader's pin-memory thread, an eval thread) during the
        # capture -- it happens mid-training, on the third sighting of a shape -- must not abort it
        import gc

        gc_was_on = gc.isenabled()  # (no cyclic collection inside a capture: see engine/trainer.py:_StepGraph._capture)
        gc.disable()
        try:
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.out = net._forward_uint8(self.static_in, sizes, pixel_mean, pixel_std)
        finally:
            if gc_was_on:
                gc.enable()

    def __call__(self, images_u8):
        if images_u8.data_ptr() != self.static_in.data_ptr():
            self.static_in.copy_(images_u8)
        self.graph.replay()
        return self.out


def make_stage(*args, **kwargs):
    return ResNet.make_stage(*args, **kwargs)


@BACKBONE_REGISTRY.register()
def build_wsl_resnet_backbone(cfg, input_shape):
    """resnet_wsl.py:623-707 (stage wiring: stride-by-maxpool, dilated res4/res5)."""
    norm = cfg.MODEL.RESNETS.NORM
    stem = BasicStem(in_channels=input_shape.channels, out_channels=cfg.MODEL.RESNETS.STEM_OUT_CHANNELS, norm=norm)
    freeze_at = cfg.MODEL.BACKBONE.FREEZE_AT
    out_features = cfg.MODEL.RESNETS.OUT_FEATURES
    depth = cfg.MODEL.RESNETS.DEPTH
    num_groups = cfg.MODEL.RESNETS.NUM_GROUPS
    width_per_group = cfg.MODEL.RESNETS.WIDTH_PER_GROUP
    bottleneck_channels = num_groups * width_per_group
    in_channels = cfg.MODEL.RESNETS.STEM_OUT_CHANNELS
    out_channels = cfg.MODEL.RESNETS.RES2_OUT_CHANNELS
    stride_in_1x1 = cfg.MODEL.RESNETS.STRIDE_IN_1X1
    res5_dilation = cfg.MODEL.RESNETS.RES5_DILATION
    assert res5_dilation in {1, 2}, "res5_dilation cannot be {}.".format(res5_dilation)
    if any(cfg.MODEL.RESNETS.DEFORM_ON_PER_STAGE):
        raise NotImplementedError("deformable conv is not used by the WSR configs")
    num_blocks_per_stage = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3],
                            152: [3, 8, 36, 3]}[depth]
    if depth in [18, 34]:
        assert out_channels == 64, "Must set MODEL.RESNETS.RES2_OUT_CHANNELS = 64 for R18/R34"
        assert num_groups == 1, "Must set MODEL.RESNETS.NUM_GROUPS = 1 for R18/R34"
    stages = []
    for idx, stage_idx in enumerate(range(2, 6)):
        dilation = res5_dilation if stage_idx == 5 or stage_idx == 4 else 1
        first_stride = 2 if idx == 0 or (stage_idx == 3 and res5_dilation == 1) else 1
        has_pool = stage_idx == 2 or stage_idx == 3
        nb = num_blocks_per_stage[idx]
        stage_kargs = {"num_blocks": nb, "stride_per_block": [1] * (nb - 1) + [first_stride],
                       "has_pool_per_block": [False] * (nb - 1) + [has_pool], "in_channels": in_channels,
                       "out_channels": out_channels, "norm": norm, "dilation": dilation}
        if depth in [18, 34]:
            stage_kargs["block_class"] = BasicBlock
        else:
            stage_kargs.update(block_class=BottleneckBlock, bottleneck_channels=bottleneck_channels,
                               stride_in_1x1=stride_in_1x1, num_groups=num_groups)
        stages.append(ResNet.make_stage(**stage_kargs))
        in_channels = out_channels
        out_channels *= 2
        bottleneck_channels *= 2
    return ResNet(stem, stages, out_features=out_features, freeze_at=freeze_at,
                  precision=forward_precision(cfg.MODEL.HIP.PRECISION))
