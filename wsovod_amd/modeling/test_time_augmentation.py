"""Test-time augmentation around the HIP inference path.

Mirrors /root/reference/wsovod/modeling/test_time_augmentation_avg.py:67-334 and
test_time_augmentation_union.py (same mapper, different merge): every image is run at several short-edge sizes, each
optionally flipped; AVG maps the per-proposal boxes of every view back to the original frame, averages boxes and class
scores over the views and applies ONE threshold / per-class NMS / top-k pass; UNION pools the detections of the
views and lets the NMS pick.  detectron2's ResizeShortestEdge / RandomFlip(1.0) / TransformList (un-vendored) are
restated in wsovod_amd/data/proposals.py; image resampling is PIL bilinear on the host (what detectron2's
ResizeTransform does for uint8 images).  The model runs through `GeneralizedRCNN_WSOVOD.inference`
(do_postprocess=False), the merge through `fast_rcnn_inference_single_image` (HIP NMS).
"""
import copy
from itertools import count
from typing import List

import numpy as np
import torch
from torch import nn

from ..config import configurable
from ..data.proposals import HFlipTransform, ResizeTransform, TransformList, transform_proposals
from ..structures import Boxes, Instances
from .fast_rcnn_open_vocabulary import fast_rcnn_inference_single_image

__all__ = ["DatasetMapperTTAAVG", "DatasetMapperTTAUNION", "GeneralizedRCNNWithTTAAVG", "GeneralizedRCNNWithTTAUNION"]


def _shortest_edge_size(h, w, size, max_size):
    """detectron2 ResizeShortestEdge.get_output_shape."""
    scale = size * 1.0 / min(h, w)
    newh, neww = (size, scale * w) if h < w else (scale * h, size)
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)


def _resize_image(img_hwc, new_h, new_w):
    from PIL import Image

    pil = Image.fromarray(img_hwc)
    return np.asarray(pil.resize((new_w, new_h), Image.BILINEAR))


class _InvertibleList(TransformList):
    def inverse(self):
        inv = []
        for t in reversed(self.transforms):
            if isinstance(t, ResizeTransform):
                inv.append(ResizeTransform(t.new_h, t.new_w, t.h, t.w))
            else:  # a horizontal flip is its own inverse
                inv.append(t)
        return _InvertibleList(inv)


class DatasetMapperTTAAVG:
    """One dataset dict -> list of augmented dicts (`image`, transformed `proposals`, `transforms`)."""

    @configurable
    def __init__(self, min_sizes: List[int], max_size: int, flip: bool, proposal_topk: int):
        self.min_sizes, self.max_size, self.flip, self.proposal_topk = min_sizes, max_size, flip, proposal_topk

    @classmethod
    def from_config(cls, cfg):
        return {"min_sizes": cfg.TEST.AUG.MIN_SIZES, "max_size": cfg.TEST.AUG.MAX_SIZE, "flip": cfg.TEST.AUG.FLIP,
                "proposal_topk": cfg.DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TEST if cfg.MODEL.LOAD_PROPOSALS else 0}

    def __call__(self, dataset_dict):
        numpy_image = dataset_dict["image"].permute(1, 2, 0).numpy()
        shape = numpy_image.shape
        orig_shape = (dataset_dict["height"], dataset_dict["width"])
        pre = [ResizeTransform(orig_shape[0], orig_shape[1], shape[0], shape[1])] if shape[:2] != orig_shape else []
        ret = []
        for min_size in self.min_sizes:
            nh, nw = _shortest_edge_size(shape[0], shape[1], min_size, self.max_size)
            resized = _resize_image(numpy_image, nh, nw)
            for flip in ([False, True] if self.flip else [False]):
                tf = pre + [ResizeTransform(shape[0], shape[1], nh, nw)] + ([HFlipTransform(nw)] if flip else [])
                new_image = resized[:, ::-1] if flip else resized
                dic = copy.deepcopy(dataset_dict)
                dic["transforms"] = _InvertibleList(tf)
                dic["image"] = torch.from_numpy(np.ascontiguousarray(new_image.transpose(2, 0, 1)))
                if self.proposal_topk and "proposal_boxes" in dic:
                    transform_proposals(dic, new_image.shape[:2], dic["transforms"], proposal_topk=self.proposal_topk)
                elif "proposals" in dic:  # already-mapped Instances: move the boxes with the view, keep their order
                    p = dic["proposals"]
                    q = Instances(tuple(new_image.shape[:2]), **p.get_fields())
                    view = _InvertibleList(tf[len(pre):])  # the mapped boxes live in the input image's frame
                    boxes = Boxes(torch.from_numpy(view.apply_box(p.proposal_boxes.tensor.cpu().numpy())).float())
                    boxes.clip(q.image_size)
                    q.proposal_boxes = boxes
                    dic["proposals"] = q
                ret.append(dic)
        return ret


class DatasetMapperTTAUNION(DatasetMapperTTAAVG):
    """test_time_augmentation_union.py:66-145: the same views; with `proposal_topk` > 0 the already-mapped `proposals` of
    the input follow each view through the mapper's own `transform_proposals` (:25-63): view transforms (without the
    pre-transform), clip to the view, boxes with an empty side dropped, the first proposal_topk kept."""

    def __call__(self, dataset_dict):
        views = super().__call__(dataset_dict)
        if not self.proposal_topk or "proposals" not in dataset_dict:
            return views
        src = dataset_dict["proposals"]
        shape = tuple(dataset_dict["image"].shape[1:])
        orig_shape = (dataset_dict["height"], dataset_dict["width"])
        n_pre = 1 if shape != orig_shape else 0
        for v in views:
            view_tf = _InvertibleList(list(v["transforms"].transforms)[n_pre:])
            hw = tuple(v["image"].shape[1:])
            boxes = Boxes(torch.from_numpy(view_tf.apply_box(src.proposal_boxes.tensor.cpu().numpy())).float())
            boxes.clip(hw)
            keep = boxes.nonempty(threshold=0)
            q = Instances(hw)
            q.proposal_boxes = boxes[keep][:self.proposal_topk]
            q.objectness_logits = src.objectness_logits[keep][:self.proposal_topk]
            v["proposals"] = q
        return views


class _TTABase(nn.Module):
    _mapper_cls = DatasetMapperTTAAVG

    def __init__(self, cfg, model, tta_mapper=None, batch_size=1):
        super().__init__()
        self.cfg = cfg.clone()
        self.model = model
        self.tta_mapper = tta_mapper if tta_mapper is not None else self._mapper_cls(cfg)
        self.batch_size = batch_size

    def _run_model(self, batched_inputs):
        outputs, all_scores, all_boxes = [], [], []
        for i in range(0, len(batched_inputs), self.batch_size):
            out, sc, bx = self.model.inference(batched_inputs[i:i + self.batch_size], do_postprocess=False)
            outputs.extend(out)
            all_scores.extend(sc)
            all_boxes.extend(bx)
        return outputs, all_scores, all_boxes

    def __call__(self, batched_inputs):
        return [self._inference_one_image(copy.copy(x)) for x in batched_inputs]

    def _inference_one_image(self, inp):
        if "height" not in inp and "width" not in inp:
            inp["height"], inp["width"] = inp["image"].shape[1], inp["image"].shape[2]
        orig_shape = (inp["height"], inp["width"])
        augmented = self.tta_mapper(inp)
        tfms = [x.pop("transforms") for x in augmented]
        boxes, scores, classes = self._get_augmented_boxes(augmented, tfms)
        return {"instances": self._merge_detections(boxes, scores, classes, orig_shape)}


class GeneralizedRCNNWithTTAAVG(_TTABase):
    """Average per-proposal boxes and class scores over the views, then one detection pass."""

    def _get_augmented_boxes(self, augmented, tfms):
        _, all_scores, all_boxes = self._run_model(augmented)
        back = []
        for pred_boxes, tfm in zip(all_boxes, tfms):
            num_img, num_pred, num_col = pred_boxes.shape
            assert num_img == 1
            orig = tfm.inverse().apply_box(pred_boxes.reshape(num_pred * num_col // 4, 4).cpu().numpy())
            back.append(torch.from_numpy(orig).to(pred_boxes.device).reshape(1, num_pred, num_col))
        boxes = torch.mean(torch.cat(back, dim=0), dim=0)
        scores = torch.mean(torch.cat(all_scores, dim=0), dim=0)
        return boxes, scores, None

    def _merge_detections(self, all_boxes, all_scores, all_classes, shape_hw):
        merged, *_ = fast_rcnn_inference_single_image(
            all_boxes, all_scores, shape_hw, self.cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST,
            self.cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST, self.cfg.TEST.DETECTIONS_PER_IMAGE)
        return merged


class GeneralizedRCNNWithTTAUNION(_TTABase):
    """Pool the detections of every view (mapped back to the original frame) and let the NMS choose."""
    _mapper_cls = DatasetMapperTTAUNION

    def _get_augmented_boxes(self, augmented, tfms):
        outputs, _, _ = self._run_model(augmented)
        all_boxes, all_scores, all_classes = [], [], []
        for output, tfm in zip(outputs, tfms):
            pred_boxes = output.pred_boxes.tensor
            orig = tfm.inverse().apply_box(pred_boxes.cpu().numpy())
            all_boxes.append(torch.from_numpy(orig).to(pred_boxes.device).to(pred_boxes.dtype))
            all_scores.extend(output.scores)
            all_classes.extend(output.pred_classes)
        return torch.cat(all_boxes, dim=0), all_scores, all_classes

    def _merge_detections(self, all_boxes, all_scores, all_classes, shape_hw):
        num_boxes, num_classes = len(all_boxes), self.cfg.MODEL.ROI_HEADS.NUM_CLASSES
        scores_2d = torch.zeros(num_boxes, num_classes + 1, device=all_boxes.device)
        if num_boxes:
            idx = torch.arange(num_boxes, device=all_boxes.device)
            scores_2d[idx, torch.stack(list(all_classes)).to(all_boxes.device)] = torch.stack(list(all_scores))
        merged, *_ = fast_rcnn_inference_single_image(all_boxes, scores_2d, shape_hw, 1e-8,
                                                      self.cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST,
                                                      self.cfg.TEST.DETECTIONS_PER_IMAGE)
        return merged
