"""detectron2.modeling.anchor_generator.DefaultAnchorGenerator restated (un-vendored; call site
/root/reference/wsovod/modeling/proposal_generator/rpn.py:200-204).  Host-side index work: the anchor grid of a
feature-map size is computed once and cached on the device.

Cell anchors: for every size s and aspect ratio a (h / w): w = sqrt(s*s / a), h = a * w, box (-w/2, -h/2, w/2, h/2).
Grid: shifts (x, y) = (offset + i) * stride over the map, flattened (H, W, A) -- the order in which the RPN head's
NHWC outputs are laid out.
"""
import math
from typing import List

import torch
from torch import nn

from ..config import Registry, configurable
from ..structures import Boxes, ShapeSpec

ANCHOR_GENERATOR_REGISTRY = Registry("ANCHOR_GENERATOR")


def _broadcast_params(params, num_features, name):
    assert isinstance(params, (list, tuple)), f"{name} in anchor generator has to be a list! Got {params}."
    assert len(params), f"{name} in anchor generator cannot be empty!"
    if not isinstance(params[0], (list, tuple)):  # params is list[float]
        return [params] * num_features
    if len(params) == 1:
        return list(params) * num_features
    assert len(params) == num_features, (
        f"Got {name} of length {len(params)} in anchor generator, but the number of input features is {num_features}!")
    return params


@ANCHOR_GENERATOR_REGISTRY.register()
class DefaultAnchorGenerator(nn.Module):
    box_dim: int = 4

    @configurable
    def __init__(self, *, sizes, aspect_ratios, strides, offset=0.5):
        super().__init__()
        self.strides = strides
        self.num_features = len(self.strides)
        sizes = _broadcast_params(sizes, self.num_features, "sizes")
        aspect_ratios = _broadcast_params(aspect_ratios, self.num_features, "aspect_ratios")
        self._cell = [self.generate_cell_anchors(s, a) for s, a in zip(sizes, aspect_ratios)]
        self.offset = offset
        assert 0.0 <= self.offset < 1.0, self.offset
        self._cache = {}

    @classmethod
    def from_config(cls, cfg, input_shape: List[ShapeSpec]):
        return {"sizes": cfg.MODEL.ANCHOR_GENERATOR.SIZES, "aspect_ratios": cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS,
                "strides": [x.stride for x in input_shape], "offset": cfg.MODEL.ANCHOR_GENERATOR.OFFSET}

    @property
    def num_anchors(self):
        return [len(c) for c in self._cell]

    @property
    def num_cell_anchors(self):
        return self.num_anchors

    @staticmethod
    def generate_cell_anchors(sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1, 2)):
        anchors = []
        for size in sizes:
            area = size ** 2.0
            for aspect_ratio in aspect_ratios:
                w = math.sqrt(area / aspect_ratio)
                h = aspect_ratio * w
                anchors.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
        return torch.tensor(anchors)

    def _grid(self, level, h, w, device):
        key = (level, h, w, str(device))
        if key not in self._cache:
            stride, base = self.strides[level], self._cell[level]
            sx = torch.arange(self.offset * stride, w * stride, step=stride, dtype=torch.float32)
            sy = torch.arange(self.offset * stride, h * stride, step=stride, dtype=torch.float32)
            shift_y, shift_x = torch.meshgrid(sy, sx, indexing="ij")
            shift_x, shift_y = shift_x.reshape(-1), shift_y.reshape(-1)
            shifts = torch.stack((shift_x, shift_y, shift_x, shift_y), dim=1)
            grid = (shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4)
            self._cache[key] = grid.contiguous().to(device)
        return self._cache[key]

    def forward(self, features: List[torch.Tensor]):
        """features: NCHW-logical maps -> list[Boxes], one (H*W*A, 4) grid per level."""
        return [Boxes(self._grid(i, f.shape[-2], f.shape[-1], f.device)) for i, f in enumerate(features)]


def build_anchor_generator(cfg, input_shape):
    return ANCHOR_GENERATOR_REGISTRY.get(cfg.MODEL.ANCHOR_GENERATOR.NAME)(cfg, input_shape)
