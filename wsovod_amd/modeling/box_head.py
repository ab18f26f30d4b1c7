"""DiscriminativeAdaptationNeck on the MFMA contraction kernel.

Mirror of /root/reference/wsovod/modeling/roi_heads/box_head.py:18-106: Flatten -> fc1 -> ReLU ->
Dropout(0.5) -> fc2 -> ReLU -> Dropout(0.5), same child-module names (state-dict keys
`fc1.weight`, ...), same initialisers (N(0, 0.005), bias 0.1).  Each FC + bias + ReLU + dropout is
ONE launch (epilogue fused); the backward is the mask/transpose prologue + two contractions.
"""
from typing import List

import torch
from torch import nn

from ..config import ROI_BOX_HEAD_REGISTRY, configurable
from ..layers import functions as Fn
from ..layers import hip_ops as H
from ..structures import ShapeSpec

__all__ = ["DiscriminativeAdaptationNeck", "build_box_head"]


@ROI_BOX_HEAD_REGISTRY.register()
class DiscriminativeAdaptationNeck(nn.Sequential):
    @configurable
    def __init__(self, input_shape: ShapeSpec, *, conv_dims: List[int], fc_dims: List[int], conv_norm="", seed: int = -1):
        super().__init__()
        if len(conv_dims) + len(fc_dims) == 0:
            raise ValueError("DiscriminativeAdaptationNeck needs at least one layer")
        if conv_dims:
            raise NotImplementedError("NUM_CONV > 0 is not used by any WSOVOD config (hot path: FC neck only)")
        # Child names and order are the checkpoint contract of the reference (box_head.py:60-71): flatten, then per
        # layer k = 1.. `fc{k}` (N(0, 0.005) weights, bias 0.1), `fc_relu{k}`, `fc_dropout{k}` (p = 0.5).
        self.conv_norm_relus, self.fcs = [], []
        width = input_shape.channels * (input_shape.height or 1) * (input_shape.width or 1)
        self._output_size = (input_shape.channels, input_shape.height, input_shape.width)
        self.add_module("flatten", nn.Flatten())
        for k, out_width in enumerate(fc_dims, start=1):
            layer = nn.Linear(width, out_width)
            for name, child in ((f"fc{k}", layer), (f"fc_relu{k}", nn.ReLU(inplace=True)),
                                (f"fc_dropout{k}", nn.Dropout(p=0.5, inplace=False))):
                self.add_module(name, child)
            self.fcs.append(layer)
            width = self._output_size = out_width
        for layer in self.fcs:  # (after all layers exist: the same draws from the generator as the reference's constructor)
            nn.init.normal_(layer.weight, std=0.005)
            nn.init.constant_(layer.bias, 0.1)
        # Counter-based dropout (mask = hash(seed, step, layer, row, unit)); the reference draws from per-process
        # torch RNG streams seeded `cfg.SEED + rank` (detectron2 seed_all_rng): the base seed mixes cfg.SEED and the
        # data-parallel rank (read at the first training forward, the process group may not exist yet), the step
        # counter advances in training mode only and can be restored with `set_step(iteration)` on resume.
        self._step = 0
        self._step_dev = None
        self._cfg_seed = int(seed)
        self.dropout_seed = None

    def set_step(self, iteration: int):
        self._step = int(iteration)
        if self._step_dev is not None:
            self._step_dev.fill_(16 * self._step)

    def _step_term(self, device):
        """16 x (training step) as a 1-element int64 DEVICE tensor, advanced by an in-stream add at every training
        forward: the kernels add it to the layer's base seed, so eager launches and replays of a captured step graph
        (frozen kernel arguments) draw the same mask for the same step."""
        if self._step_dev is None or self._step_dev.device != device:
            self._step_dev = torch.full((1,), 16 * self._step, dtype=torch.int64, device=device)
        return self._step_dev

    def _base_seed(self):
        if self.dropout_seed is None:
            import torch.distributed as dist

            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
            base = self._cfg_seed if self._cfg_seed >= 0 else 0x5EED
            self.dropout_seed = (base * 0x9E3779B1 + rank * 0x85EBCA77 + 0x5EED) & 0xFFFFFFFF
        return self.dropout_seed

    @classmethod
    def from_config(cls, cfg, input_shape):
        return {"input_shape": input_shape, "conv_dims": [cfg.MODEL.ROI_BOX_HEAD.CONV_DIM] * cfg.MODEL.ROI_BOX_HEAD.NUM_CONV,
                "fc_dims": cfg.MODEL.ROI_BOX_HEAD.DAN_DIM, "conv_norm": cfg.MODEL.ROI_BOX_HEAD.NORM,
                "seed": cfg.SEED}

    def forward(self, x):
        if x.dim() > 2:
            x = torch.flatten(x, start_dim=1)
        step_term = None
        if self.training:
            if x.is_cuda:
                step_term = self._step_term(x.device)  # (created, if need be, from the step count BEFORE this forward)
                step_term.add_(16)  # in stream order: part of a captured step graph as well
            self._step += 1
        for k, fc in enumerate(self.fcs):
            drop = getattr(self, "fc_dropout{}".format(k + 1))
            p = drop.p if (self.training and drop.training) else 0.0
            # seed = base * 1000003 + 16 * step + layer; on the device path the step term is added by the kernel
            if step_term is not None:
                seed = (self._base_seed() * 1000003 + k) & 0x7FFFFFFFFFFFFFFF if p > 0 else 0
            else:
                seed = (self._base_seed() * 1000003 + self._step * 16 + k) & 0x7FFFFFFFFFFFFFFF if p > 0 else 0
            # "parity" precision: x arrives as bf16x2 (the pooler wrote it) and every FC hands bf16x2 on; "parity_mx": x arrives
            # as f16mx, the FC layers hand f16mx on among themselves and the last one bf16x2 to the heads
            fmt = None
            if H.x3_active() == "x2":
                fmt = H.MX if (H.mx_of(x) and k + 1 < len(self.fcs)) else H.X2
            x = Fn.linear(x, fc.weight, fc.bias, relu=True, dropout_p=p, seed=seed, out_dtype=fmt,
                          seed_add=step_term if p > 0 else None)
        return x

    @property
    def output_shape(self):
        o = self._output_size
        if isinstance(o, int):
            return ShapeSpec(channels=o)
        return ShapeSpec(channels=o[0], height=o[1], width=o[2])


def build_box_head(cfg, input_shape):
    return ROI_BOX_HEAD_REGISTRY.get(cfg.MODEL.ROI_BOX_HEAD.NAME)(cfg, input_shape)
