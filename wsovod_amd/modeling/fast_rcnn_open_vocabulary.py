"""Object-mining (WSDDN MIL) and instance-refinement output layers on HIP kernels.

Mirror of /root/reference/wsovod/modeling/roi_heads/fast_rcnn_open_vocabulary.py: same classes
(`ObjectMiningOutputLayers` :220-618, `InstanceRefinementOutputLayers` :621-1062), same child
names (`cls`, `det`, `bbox_pred`), initialisers, `forward` / `losses` / `predict_*` surface and
loss-dict keys.  Training-path compute (the two softmaxes over ragged per-image segments, image
BCE, cosine-similarity logits, weighted CE, weighted smooth-L1) runs on the C-ABI kernels.
The eval tail (`fast_rcnn_inference`: threshold + per-class NMS + top-k) is a SURVEY 8f "next"
row and is kept as plain torch ops on the device.
"""
from typing import Dict, List, Tuple, Union

import torch
from torch import nn
from torch.nn import functional as F

from ..config import configurable
from ..layers import functions as Fn
from ..layers import hip_ops as H
from ..layers.functions import scale_losses
from ..structures import Boxes, Instances, ShapeSpec
from .box_regression import Box2BoxTransform

__all__ = ["fast_rcnn_inference", "BatchedDetections", "ObjectMiningOutputLayers", "InstanceRefinementOutputLayers",
           "segment_offsets"]


def segment_offsets(nums, device):
    """int32 (G+1) prefix offsets of the per-image proposal counts (host ints -> one tiny H2D copy)."""
    from ..layers.hip_ops import const_tensor

    offs = [0]
    for n in nums:
        offs.append(offs[-1] + int(n))
    return const_tensor(offs, torch.int32, device)


def batched_nms(boxes, scores, idxs, iou_threshold):
    """detectron2.layers.batched_nms (-> torchvision): NMS inside each category `idxs`, result sorted by
    descending score.  Boxes are ordered by (category, score), each category becomes one segment of the HIP
    segment-NMS kernel (the exact per-category form; torchvision's coordinate-offset shortcut approximates it)."""
    from ..layers import hip_ops as H

    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    by_score = scores.argsort(descending=True, stable=True)
    order = by_score[idxs[by_score].argsort(stable=True)]
    _, counts = torch.unique_consecutive(idxs[order], return_counts=True)
    seg = torch.zeros(counts.numel() + 1, dtype=torch.int32, device=boxes.device)
    seg[1:] = counts.cumsum(0)
    keep_rel, keep_count = H.nms_segments(boxes.float()[order], seg, int(counts.max()), float(iou_threshold))
    pos = torch.arange(order.numel(), device=boxes.device, dtype=torch.int32)
    start = torch.repeat_interleave(seg[:-1], counts)  # segment start of every slot
    kept_slots = pos - start < torch.repeat_interleave(keep_count, counts)  # slot s of segment g holds its s-th keep
    kept = order[(start + keep_rel)[kept_slots].long()]
    return kept[scores[kept].argsort(descending=True, stable=True)]


def fast_rcnn_inference_single_image(boxes, scores, image_shape, score_thresh, nms_thresh, topk_per_image):
    """fast_rcnn_open_vocabulary.py:149-217: finite filter, clip, score threshold, per-class NMS, top-k.
    Index plumbing in torch on the device; the suppression itself is the HIP segment-NMS kernel."""
    all_scores, all_boxes = scores.clone().unsqueeze(0), boxes.clone().unsqueeze(0)
    pred_inds = torch.arange(scores.size(0), device=scores.device, dtype=torch.long).unsqueeze(1).repeat(
        1, scores.size(1))
    valid_mask = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
    if not valid_mask.all():
        boxes, scores, pred_inds = boxes[valid_mask], scores[valid_mask], pred_inds[valid_mask]
    scores = scores[:, :-1]
    num_bbox_reg_classes = boxes.shape[1] // 4
    boxes = Boxes(boxes.reshape(-1, 4))
    boxes.clip(image_shape)
    boxes = boxes.tensor.view(-1, num_bbox_reg_classes, 4)
    pred_inds = pred_inds[:, :-1]
    filter_mask = scores > score_thresh
    filter_inds = filter_mask.nonzero()
    boxes = boxes[filter_inds[:, 0], 0] if num_bbox_reg_classes == 1 else boxes[filter_mask]
    scores = scores[filter_mask]
    pred_inds = pred_inds[filter_mask]
    keep = batched_nms(boxes, scores, filter_inds[:, 1], nms_thresh)
    if topk_per_image >= 0:
        keep = keep[:topk_per_image]
    boxes, scores, filter_inds, pred_inds = boxes[keep], scores[keep], filter_inds[keep], pred_inds[keep]
    result = Instances(image_shape)
    result.pred_boxes = Boxes(boxes)
    result.scores = scores
    result.pred_classes = filter_inds[:, 1]
    result.pred_inds = pred_inds
    return result, filter_inds[:, 0], all_scores, all_boxes


def fast_rcnn_inference(boxes, scores, image_shapes, score_thresh, nms_thresh, topk_per_image):
    if (len(scores) and scores[0].is_cuda and topk_per_image >= 0 and all(b.dim() == 2 and b.shape[1] == 4 for b in boxes)
            and all(s.shape[0] > 0 for s in scores) and max(s.shape[0] for s in scores) <= 16384):
        # bound the (images, classes, proposals) candidate tensors: large vocabularies go through in groups of images
        per_image = (scores[0].shape[1] - 1) * max(s.shape[0] for s in scores)
        group = max(1, min(len(scores), (8 << 20) // max(per_image, 1)))
        if group >= len(scores):
            return _fast_rcnn_inference_batched(boxes, scores, image_shapes, score_thresh, nms_thresh, topk_per_image)
        out = ([], [], [], [])
        for i in range(0, len(scores), group):
            part = _fast_rcnn_inference_batched(boxes[i:i + group], scores[i:i + group], image_shapes[i:i + group],
                                                score_thresh, nms_thresh, topk_per_image)
            for acc, p in zip(out, part):
                acc.extend(p)
        return out  # (a plain list: the packed post-processing path applies to single-group batches only)
    r = [fast_rcnn_inference_single_image(b, s, shp, score_thresh, nms_thresh, topk_per_image)
         for s, b, shp in zip(scores, boxes, image_shapes)]
    return [x[0] for x in r], [x[1] for x in r], [x[2] for x in r], [x[3] for x in r]


class BatchedDetections(list):
    """list[Instances] of the batched tail; `packed` = (boxes (N,k,4), scores, classes, proposal ids (N,k), counts
    list) are the batch-level tensors the per-image Instances are slices of, so that a caller that post-processes every
    image the same way (detector_postprocess) can do it in one pass."""
    packed = None


def _fast_rcnn_inference_batched(boxes, scores, image_shapes, score_thresh, nms_thresh, topk_per_image):
    """The tail of fast_rcnn_inference_single_image (fast_rcnn_open_vocabulary.py:149-217) for ALL images of the batch
    at once, class-agnostic boxes: no per-image Python loop over index ops, ONE host read (the detection counts).

    Every (image, class) pair is one fixed-length segment of the HIP segment-NMS kernel: the image's proposals sorted
    by that class' score (stable), with a validity byte instead of a compaction for `score > thresh` / non-finite rows
    (an invalid box is never kept and never suppresses -- the same keep set as filtering first).  The final per-image
    ranking is a stable descending sort over the (class-major, rank) layout, i.e. the reference's tie order
    (class, then proposal index).  Same detections in the same order as the per-image form (tests compare both with
    the oracle)."""
    from ..layers import hip_ops as H

    dev = scores[0].device
    N, K = len(scores), scores[0].shape[1] - 1
    nums = [int(s.shape[0]) for s in scores]
    Rm = max(nums)
    all_scores = [s.unsqueeze(0) for s in scores]
    all_boxes = [b.unsqueeze(0) for b in boxes]
    if all(n == Rm for n in nums):
        S = H.cat_rows(list(scores)).view(N, Rm, K + 1)
        B = H.cat_rows(list(boxes)).view(N, Rm, 4)
        pad = None
    else:  # ragged batch: pad the short images with rows that can never pass the threshold
        S = torch.full((N, Rm, K + 1), float("-inf"), dtype=scores[0].dtype, device=dev)
        B = torch.zeros((N, Rm, 4), dtype=boxes[0].dtype, device=dev)
        for i, (s, b) in enumerate(zip(scores, boxes)):
            S[i, :nums[i]], B[i, :nums[i]] = s, b
        pad = H.const_tensor([int(r < n) for n in nums for r in range(Rm)], torch.bool, dev).view(N, Rm)
    S, B = S.float(), B.float()
    row_ok = torch.isfinite(B).all(dim=2) & torch.isfinite(S).all(dim=2) if pad is None else \
        torch.isfinite(B).all(dim=2) & torch.isfinite(torch.where(pad[..., None], S, torch.zeros_like(S))).all(dim=2) & pad
    sc = torch.where(row_ok[..., None], S[..., :K], torch.full_like(S[..., :K], float("-inf")))
    sorted_sc, order = torch.sort(sc.permute(0, 2, 1), dim=2, descending=True, stable=True)  # (N, K, Rm)
    hw = H.const_tensor([float(v) for shp in image_shapes for v in (shp[1], shp[0], shp[1], shp[0])], torch.float32,
                        dev).view(N, 1, 4)
    Bc = torch.minimum(B.clamp(min=0), hw)  # Boxes.clip: x in [0, w], y in [0, h]
    Bc = torch.where(row_ok[..., None], Bc, torch.zeros_like(Bc))
    bs = torch.gather(Bc[:, None].expand(N, K, Rm, 4), 2, order[..., None].expand(N, K, Rm, 4)).contiguous()
    valid = (sorted_sc > score_thresh).contiguous()
    seg = torch.arange(N * K + 1, device=dev, dtype=torch.int32) * Rm
    keep_rel, keep_count = H.nms_segments(bs.view(-1, 4), seg, Rm, float(nms_thresh), valid=valid.view(-1))
    slot_used = torch.arange(Rm, device=dev, dtype=torch.int32)[None] < keep_count[:, None]  # (N*K, Rm)
    kept = torch.zeros((N * K, Rm), dtype=torch.int32, device=dev)
    kept.scatter_add_(1, keep_rel.view(N * K, Rm).clamp(0, Rm - 1).long(), slot_used.to(torch.int32))
    final_sc = torch.where(kept > 0, sorted_sc.reshape(N * K, Rm), torch.full_like(sorted_sc.reshape(N * K, Rm),
                                                                                  float("-inf"))).view(N, K * Rm)
    top_sc, top_idx = torch.sort(final_sc, dim=1, descending=True, stable=True)
    k = min(topk_per_image, K * Rm)
    top_sc, top_idx = top_sc[:, :k], top_idx[:, :k]
    counts = (top_sc > float("-inf")).sum(dim=1).tolist()  # the one host read of the tail
    det_cls = torch.div(top_idx, Rm, rounding_mode="floor")
    det_prop = order.reshape(N, K * Rm).gather(1, top_idx)
    det_box = bs.view(N, K * Rm, 4).gather(1, top_idx[..., None].expand(N, k, 4))
    results, kept_props = BatchedDetections(), []
    results.packed = (det_box, top_sc, det_cls, det_prop, counts)  # (N,k,..) tensors behind the per-image slices
    for i, n in enumerate(counts):
        r = Instances(image_shapes[i])
        r.pred_boxes = Boxes(det_box[i, :n])
        r.scores = top_sc[i, :n]
        r.pred_classes = det_cls[i, :n]
        r.pred_inds = det_prop[i, :n]
        results.append(r)
        kept_props.append(det_prop[i, :n])
    return results, kept_props, all_scores, all_boxes


class ObjectMiningOutputLayers(nn.Module):
    @configurable
    def __init__(self, input_shape: ShapeSpec, *, box2box_transform, num_classes: int, class_head: nn.Module = None,
                 test_score_thresh: float = 0.0, test_nms_thresh: float = 0.5, test_topk_per_image: int = 100,
                 cls_agnostic_bbox_reg: bool = False, smooth_l1_beta: float = 0.0,
                 box_reg_loss_type: str = "smooth_l1", loss_weight: Union[float, Dict[str, float]] = 1.0,
                 mean_loss: bool = True):
        super().__init__()
        if isinstance(input_shape, int):
            input_shape = ShapeSpec(channels=input_shape)
        self.num_classes = num_classes
        input_size = input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
        self.num_bbox_reg_classes = 1 if cls_agnostic_bbox_reg else num_classes
        self.box_dim = len(box2box_transform.weights)
        self.det = nn.Linear(input_size, num_classes)
        nn.init.xavier_uniform_(self.det.weight)
        nn.init.constant_(self.det.bias, 0)
        if class_head is not None:
            raise NotImplementedError("class_head != None is never built by the reference (roi_heads.py:588-590)")
        self.cls = nn.Linear(input_size, num_classes)
        nn.init.xavier_uniform_(self.cls.weight)
        nn.init.constant_(self.cls.bias, 0)
        self.box2box_transform = box2box_transform
        self.smooth_l1_beta = smooth_l1_beta
        self.test_score_thresh, self.test_nms_thresh = test_score_thresh, test_nms_thresh
        self.test_topk_per_image = test_topk_per_image
        self.box_reg_loss_type = box_reg_loss_type
        self.loss_weight = loss_weight
        self.mean_loss = mean_loss

    @classmethod
    def from_config(cls, cfg, input_shape, class_head=None, num_classes=None):
        return {
            "input_shape": input_shape,
            "box2box_transform": Box2BoxTransform(weights=cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS),
            "num_classes": num_classes if num_classes else cfg.MODEL.ROI_HEADS.NUM_CLASSES,
            "class_head": class_head,
            "cls_agnostic_bbox_reg": cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG,
            "smooth_l1_beta": cfg.MODEL.ROI_BOX_HEAD.SMOOTH_L1_BETA,
            "test_score_thresh": cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST,
            "test_nms_thresh": cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST,
            "test_topk_per_image": cfg.TEST.DETECTIONS_PER_IMAGE,
            "box_reg_loss_type": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE,
            "loss_weight": {"loss_box_reg_object_mining": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_WEIGHT,
                            "loss_cls_object_mining": cfg.WSOVOD.OBJECT_MINING.WEIGHT},
            "mean_loss": cfg.WSOVOD.OBJECT_MINING.MEAN_LOSS,
        }

    def stacked_params(self):
        """rows [cls | det] of one (2K, F) weight: both heads are one contraction."""
        return (torch.cat([self.cls.weight, self.det.weight], dim=0), torch.cat([self.cls.bias, self.det.bias], dim=0))

    def forward(self, x, proposals=None, context=False, logits=None):
        """-> (scores (R,K) = softmax_classes(C) * softmax_proposals(D) per image, zero deltas).
        logits: optional precomputed (R, 2K) = [C | D] (batched with the other heads on x by the ROI heads)."""
        K = self.num_classes
        if context:  # forward_contextlocnet (fast_rcnn_open_vocabulary.py:369-390): C = cls(x), D = det(Fx) - det(Cx)
            x, fx, cx = [t.flatten(start_dim=1) if t.dim() > 2 else t for t in x]
            c = Fn.linear(x, self.cls.weight, self.cls.bias, out_dtype=torch.float32)
            d = Fn.linear(fx, self.det.weight, self.det.bias, out_dtype=torch.float32) - \
                Fn.linear(cx, self.det.weight, self.det.bias, out_dtype=torch.float32)
            logits = torch.cat([c, d], dim=1)
        if x.dim() > 2:
            x = torch.flatten(x, start_dim=1)
        if logits is None:
            w, b = self.stacked_params()
            logits = Fn.linear(x, w, b, out_dtype=torch.float32)  # (R, 2K) = [C | D]
        if K == 1:  # fast_rcnn_open_vocabulary.py:338-340
            z = torch.zeros_like(logits[:, :1])
            logits = torch.cat((logits[:, :1], z, logits[:, 1:], z), dim=1)
        nums = [x.size(0)] if (proposals is None or len(proposals) == 1) else [len(p) for p in proposals]
        self._seg = segment_offsets(nums, x.device)
        scores = Fn.mil_scores(logits, self._seg, max(K, 2) if K == 1 else K)
        if K == 1:
            scores = scores[:, :1]
        proposal_deltas = torch.zeros(scores.shape[0], self.num_bbox_reg_classes * self.box_dim, dtype=scores.dtype,
                                      device=scores.device, requires_grad=False)
        return scores, proposal_deltas

    def losses(self, predictions, proposals, gt_classes_img_oh):
        scores, _ = predictions
        assert gt_classes_img_oh.dim() == 2
        seg = segment_offsets([len(p) for p in proposals], scores.device)
        G, K = gt_classes_img_oh.shape
        # mean: BCE mean over (N,K); else sum / N   (fast_rcnn_open_vocabulary.py:411-425)
        norm = float(G * K) if self.mean_loss else float(G)
        loss, img = Fn.image_bce(scores, seg, gt_classes_img_oh, norm)
        self._pred_class_img_logits = img
        losses = {"loss_cls_object_mining": loss}
        return scale_losses(losses, self.loss_weight)

    def predict_boxes(self, predictions, proposals):
        return [p.proposal_boxes.tensor for p in proposals]

    def predict_probs(self, predictions, proposals):
        scores, _ = predictions
        num_inst_per_image = [len(p) for p in proposals]
        probs_bg = torch.zeros(scores.shape[0], 1, dtype=scores.dtype, device=scores.device)
        return torch.cat((scores, probs_bg), 1).split(num_inst_per_image, dim=0)

    def predict_probs_img(self, predictions, proposals):
        """Clamped image-level scores (N,K); computed by the BCE kernel when losses() ran first."""
        cached = getattr(self, "_pred_class_img_logits", None)
        if cached is not None and self.training:
            return cached
        scores, _ = predictions
        seg = segment_offsets([len(p) for p in proposals], scores.device)
        zeros = torch.zeros((len(proposals), scores.size(1)), device=scores.device)
        _, img = Fn.image_bce(scores.detach(), seg, zeros, 1.0)
        return img


class InstanceRefinementOutputLayers(nn.Module):
    @configurable
    def __init__(self, input_shape: ShapeSpec, *, box2box_transform, num_classes: int, class_head: nn.Module,
                 test_score_thresh: float = 0.0, test_nms_thresh: float = 0.5, test_topk_per_image: int = 100,
                 smooth_l1_beta: float = 0.0, box_reg_loss_type: str = "smooth_l1",
                 loss_weight: Union[float, Dict[str, float]] = 1.0, refine_k: int = None, refine_reg=False,
                 cross_entropy_weighted: bool = True):
        super().__init__()
        if isinstance(input_shape, int):
            input_shape = ShapeSpec(channels=input_shape)
        self.num_classes = num_classes
        input_size = input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
        self.cls = class_head
        self.num_bbox_reg_classes = 1
        self.box_dim = len(box2box_transform.weights)
        self.bbox_pred = nn.Linear(input_size, self.num_bbox_reg_classes * self.box_dim)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        nn.init.constant_(self.bbox_pred.bias, 0)
        self.box2box_transform = box2box_transform
        self.smooth_l1_beta = smooth_l1_beta
        self.test_score_thresh, self.test_nms_thresh = test_score_thresh, test_nms_thresh
        self.test_topk_per_image = test_topk_per_image
        self.box_reg_loss_type = box_reg_loss_type
        if isinstance(loss_weight, float):
            loss_weight = {"loss_cls_r" + str(refine_k): loss_weight, "loss_box_reg_r" + str(refine_k): loss_weight}
        self.loss_weight = loss_weight
        self.refine_k, self.refine_reg = refine_k, refine_reg
        self.cross_entropy_weighted = cross_entropy_weighted
        if not self.refine_reg[self.refine_k]:
            del self.bbox_pred

    @classmethod
    def from_config(cls, cfg, input_shape, refine_k, class_head):
        return {
            "input_shape": input_shape,
            "box2box_transform": Box2BoxTransform(weights=cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS),
            "num_classes": cfg.MODEL.ROI_HEADS.NUM_CLASSES,
            "class_head": class_head,
            "smooth_l1_beta": cfg.MODEL.ROI_BOX_HEAD.SMOOTH_L1_BETA,
            "test_score_thresh": cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST,
            "test_nms_thresh": cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST,
            "test_topk_per_image": cfg.TEST.DETECTIONS_PER_IMAGE,
            "box_reg_loss_type": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE,
            "loss_weight": {"loss_box_reg_r" + str(refine_k): cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_WEIGHT,
                            "loss_cls_r" + str(refine_k): cfg.WSOVOD.INSTANCE_REFINEMENT.WEIGHT},
            "refine_k": refine_k,
            "refine_reg": cfg.WSOVOD.INSTANCE_REFINEMENT.REFINE_REG,
            "cross_entropy_weighted": cfg.WSOVOD.INSTANCE_REFINEMENT.CROSS_ENTROPY_WEIGHTED,
        }

    def forward(self, x, classifier=None, append_background=True, pre=None):
        """-> (logits (R,K+1) with background logit == 0, class-agnostic deltas (R,4)).
        pre: optional (hidden, deltas) precomputed by the ROI heads' grouped contraction on x."""
        if x.dim() > 2:
            x = torch.flatten(x, start_dim=1)
        hidden, deltas = pre if pre is not None else (None, None)
        scores = self.cls(x, classifier, append_background=append_background, hidden=hidden)
        if self.refine_reg[self.refine_k]:
            proposal_deltas = deltas if deltas is not None else \
                Fn.linear(x, self.bbox_pred.weight, self.bbox_pred.bias, out_dtype=torch.float32)
        else:
            proposal_deltas = torch.zeros(scores.shape[0], self.num_bbox_reg_classes * self.box_dim,
                                          dtype=scores.dtype, device=scores.device, requires_grad=False)
        return scores, proposal_deltas

    def losses(self, predictions, proposals, num_classes=None, proposal_boxes=None):
        """proposal_boxes: optional (sum R, 4) concatenation of the proposals' boxes the caller already holds.  The
        per-proposal targets are slices of the labelling kernel's packed output, which `cat_rows` re-joins as views."""
        scores, proposal_deltas = predictions
        dev = scores.device
        gt_classes = H.cat_rows([p.gt_classes for p in proposals]) if len(proposals) else \
            torch.empty(0, dtype=torch.int64, device=dev)
        if len(proposals):
            if proposal_boxes is None:
                proposal_boxes = H.cat_rows([p.proposal_boxes.tensor for p in proposals])
            assert not proposal_boxes.requires_grad, "Proposals should not require gradients!"
            gt_boxes = H.cat_rows([(p.gt_boxes if p.has("gt_boxes") else p.proposal_boxes).tensor for p in proposals])
        else:
            proposal_boxes = gt_boxes = torch.empty((0, 4), device=dev)
        weighted_box = self.box_reg_loss_type == "smooth_l1_weighted"
        weights = None
        if self.cross_entropy_weighted or weighted_box:
            weights = H.cat_rows([p.gt_weights for p in proposals])
        k = str(self.refine_k)
        losses = {"loss_cls_r" + k: Fn.weighted_cross_entropy(scores, gt_classes, weights,
                                                            weighted=self.cross_entropy_weighted)}
        if self.refine_reg[self.refine_k]:
            if self.box_reg_loss_type not in ("smooth_l1", "smooth_l1_weighted"):
                raise NotImplementedError(f"box_reg_loss_type {self.box_reg_loss_type} is not on the hot path")
            losses["loss_box_reg_r" + k] = Fn.weighted_l1_box_loss(
                proposal_deltas, proposal_boxes, gt_boxes, gt_classes, weights,
                num_classes if num_classes is not None else self.num_classes, self.box2box_transform.weights,
                self.smooth_l1_beta, weighted=weighted_box)
        return scale_losses(losses, self.loss_weight)

    # ---- eval tail ("next" row n2): plain torch ops on the device ----
    def inference(self, predictions, proposals):
        if isinstance(predictions[0], tuple):
            boxes = self.predict_boxes_K(predictions, proposals)
            scores = self.predict_probs_K(predictions, proposals)
        else:
            boxes = self.predict_boxes(predictions, proposals)
            scores = self.predict_probs(predictions, proposals)
        image_shapes = [x.image_size for x in proposals]
        return fast_rcnn_inference(boxes, scores, image_shapes, self.test_score_thresh, self.test_nms_thresh,
                                   self.test_topk_per_image)

    def predict_boxes(self, predictions, proposals):
        if not len(proposals):
            return []
        _, proposal_deltas = predictions
        num_prop_per_image = [len(p) for p in proposals]
        proposal_boxes = torch.cat([p.proposal_boxes.tensor for p in proposals], dim=0)
        return self.box2box_transform.apply_deltas(proposal_deltas, proposal_boxes).split(num_prop_per_image)

    def predict_boxes_K(self, predictions, proposals):
        if not len(proposals):
            return []
        deltas = torch.stack([p[1] for p in predictions]).mean(0)
        return self.predict_boxes((None, deltas), proposals)

    def predict_probs(self, predictions, proposals):
        scores, _ = predictions
        return F.softmax(scores, dim=-1).split([len(p) for p in proposals], dim=0)

    def predict_probs_K(self, predictions, proposals):
        probs = torch.stack([F.softmax(p[0], dim=-1) for p in predictions]).mean(0)
        return probs.split([len(p) for p in proposals], dim=0)
