"""ROIPooler on the HIP RoIPool / ROIAlign kernels.

Mirror of /root/reference/wsovod/modeling/poolers.py:119-337 (same constructor, same
`forward(x: List[Tensor], box_lists: List[Boxes], level_ids=None) -> (M,C,out,out)`).  The level
poolers are the C-ABI kernels; feature maps may be NCHW-contiguous (reference layout) or
channels_last (what the HIP backbone emits).  `roi_scale` is this implementation's extension: the
per-proposal objectness scaling of roi_heads.py:733-739 fused into the pooling epilogue.
"""
import math
from typing import List

import torch
from torch import nn

from ..layers import functions as Fn
from ..structures import Boxes

__all__ = ["ROIPooler", "RoIPool", "ROILoopPool", "ROIAlign", "convert_boxes_to_pooler_format", "assign_boxes_to_levels"]


class RoIPool(nn.Module):
    """torchvision.ops.RoIPool(output_size, spatial_scale) on wsovod_roi_pool_forward/backward."""

    def __init__(self, output_size, spatial_scale):
        super().__init__()
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        self.spatial_scale = spatial_scale

    def forward(self, input, rois, roi_scale=None, out_dtype=None):
        assert rois.dim() == 2 and rois.size(1) == 5
        return Fn.roi_pool(input, rois, self.output_size, self.spatial_scale, roi_scale, out_dtype)


class ROILoopPool(nn.Module):
    """wsovod.layers.ROILoopPool(output_size, spatial_scale): (3R, C, ph, pw) = [region | frame | context]."""

    def __init__(self, output_size, spatial_scale):
        super().__init__()
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        self.spatial_scale = spatial_scale

    def forward(self, input, rois, roi_scale=None, out_dtype=None):
        assert rois.dim() == 2 and rois.size(1) == 5
        assert roi_scale is None, "the objectness scale is applied by the caller for ROILoopPool"
        out = Fn.roi_loop_pool(input, rois, self.output_size, self.spatial_scale)
        return out if out_dtype is None else out.to(out_dtype)


class ROIAlign(nn.Module):
    """detectron2.layers.ROIAlign(output_size, spatial_scale, sampling_ratio, aligned)."""

    def __init__(self, output_size, spatial_scale, sampling_ratio, aligned=True):
        super().__init__()
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        self.spatial_scale, self.sampling_ratio, self.aligned = spatial_scale, sampling_ratio, aligned

    def forward(self, input, rois, roi_scale=None, out_dtype=None):
        assert rois.dim() == 2 and rois.size(1) == 5
        return Fn.roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio, self.aligned,
                            roi_scale, out_dtype)


def assign_boxes_to_levels(box_lists, min_level, max_level, canonical_box_size, canonical_level):
    box_sizes = torch.sqrt(torch.cat([boxes.area() for boxes in box_lists]))
    level_assignments = torch.floor(canonical_level + torch.log2(box_sizes / canonical_box_size + 1e-8))
    level_assignments = torch.clamp(level_assignments, min=min_level, max=max_level)
    return level_assignments.to(torch.int64) - min_level


def _fmt_box_list(box_tensor, batch_index: int):
    repeated_index = torch.full_like(box_tensor[:, :1], batch_index, dtype=box_tensor.dtype,
                                     device=box_tensor.device)
    return torch.cat((repeated_index, box_tensor), dim=1)


def convert_boxes_to_pooler_format(box_lists: List[Boxes]):
    """(M,5) fp32 [batch index, x0, y0, x1, y1]  (poolers.py:81-108).  Device boxes: one concatenation + one
    `wsovod_format_rois` launch instead of a full_like + cat per image."""
    if box_lists and box_lists[0].tensor.is_cuda:
        from ..layers import hip_ops as H
        from .fast_rcnn_open_vocabulary import segment_offsets

        boxes = H.cat_rows([b.tensor for b in box_lists])
        return H.format_rois(boxes, segment_offsets([len(b) for b in box_lists], boxes.device))[0]
    return torch.cat([_fmt_box_list(box_list.tensor, i) for i, box_list in enumerate(box_lists)], dim=0)


class ROIPooler(nn.Module):
    def __init__(self, output_size, scales, sampling_ratio, pooler_type, canonical_box_size=224, canonical_level=4):
        super().__init__()
        if isinstance(output_size, int):
            output_size = (output_size, output_size)
        assert len(output_size) == 2
        assert isinstance(output_size[0], int) and isinstance(output_size[1], int)
        self.output_size = output_size
        if pooler_type == "ROIAlign":
            self.level_poolers = nn.ModuleList(
                ROIAlign(output_size, spatial_scale=scale, sampling_ratio=sampling_ratio, aligned=False)
                for scale in scales)
        elif pooler_type == "ROIAlignV2":
            self.level_poolers = nn.ModuleList(
                ROIAlign(output_size, spatial_scale=scale, sampling_ratio=sampling_ratio, aligned=True)
                for scale in scales)
        elif pooler_type == "ROIPool":
            self.level_poolers = nn.ModuleList(RoIPool(output_size, spatial_scale=scale) for scale in scales)
        elif pooler_type == "ROILoopPool":  # poolers.py:183-186 of the reference (single level)
            self.level_poolers = nn.ModuleList(ROILoopPool(output_size, spatial_scale=scale) for scale in scales)
        elif pooler_type == "ROIAlignRotated":
            raise NotImplementedError(f"pooler type {pooler_type} is outside the hot path (no WSOVOD config selects it)")
        else:
            raise ValueError("Unknown pooler type: {}".format(pooler_type))
        min_level = -(math.log2(scales[0]))
        max_level = -(math.log2(scales[-1]))
        assert math.isclose(min_level, int(min_level)) and math.isclose(max_level, int(max_level)), \
            "Featuremap stride is not power of 2!"
        self.min_level, self.max_level = int(min_level), int(max_level)
        assert 0 <= self.min_level and self.min_level <= self.max_level
        self.canonical_level = canonical_level
        assert canonical_box_size > 0
        self.canonical_box_size = canonical_box_size

    def forward(self, x: List[torch.Tensor], box_lists: List[Boxes], level_ids=None, roi_scale=None, out_dtype=None,
                rois=None):
        """rois: optional precomputed pooler-format boxes of `box_lists` (single-level poolers only)."""
        num_level_assignments = len(self.level_poolers)
        assert isinstance(x, list) and isinstance(box_lists, list), "Arguments to pooler must be lists"
        assert len(x) == num_level_assignments, \
            "unequal value, num_level_assignments={}, but x is list of {} Tensors".format(num_level_assignments, len(x))
        assert len(box_lists) == x[0].size(0), \
            "unequal value, x[0] batch dim 0 is {}, but box_list has length {}".format(x[0].size(0), len(box_lists))
        if len(box_lists) == 0:
            return torch.zeros((0, x[0].shape[1]) + self.output_size, device=x[0].device, dtype=x[0].dtype)
        pooler_fmt_boxes = rois if (rois is not None and num_level_assignments == 1) else \
            convert_boxes_to_pooler_format(box_lists)
        if num_level_assignments == 1:
            return self.level_poolers[0](x[0], pooler_fmt_boxes, roi_scale, out_dtype)
        level_assignments = assign_boxes_to_levels(box_lists, self.min_level, self.max_level,
                                                   self.canonical_box_size, self.canonical_level)
        if level_ids is not None:
            level_assignments = torch.cat(level_ids).to(torch.int64)
        num_boxes = pooler_fmt_boxes.size(0)
        num_channels = x[0].shape[1]
        output = torch.zeros((num_boxes, num_channels) + self.output_size, dtype=out_dtype or x[0].dtype,
                             device=x[0].device)
        for level, pooler in enumerate(self.level_poolers):
            inds = (level_assignments == level).nonzero().squeeze(1)
            sc = roi_scale[inds] if roi_scale is not None else None
            output.index_put_((inds,), pooler(x[level], pooler_fmt_boxes[inds], sc, out_dtype))
        return output
