"""WSOVODROIHeads on the HIP kernels.

Mirror of /root/reference/wsovod/modeling/roi_heads/roi_heads.py (`get_image_level_gt` :158-174,
`ROIHeads` :177-427, `WSOVODROIHeads` :430-1857): same registry name, `from_config` keys,
`forward(images, features, proposals, data_aware_features=None, targets=None, classifier=None,
append_background=True, file_names=None, loaded_proposals=None)` and return values
(train: `(proposals, losses)`; eval: `(instances, {}, all_scores, all_boxes)`), plus the
`proposal_targets` attribute the meta-arch reads.

What is on the HIP path: RoI pooling with the fused objectness scale, the neck, object mining,
pseudo-GT mining + proposal labelling (one kernel, no host syncs, replacing the python list
comprehensions of get_pgt_top_k :1043-1343 and label_and_sample_proposals_wsl :1722-1825) and the
instance-refinement branch.  Out of scope (raise): MRRP, MIST
refinement, in-loop SAM box refinement, `_vis_*` debug dumps.
"""
import inspect
import os
from typing import Dict, List, Optional

import torch
from torch import nn

from ..config import ROI_HEADS_REGISTRY, configurable
from ..layers import functions as Fn
from ..layers import hip_ops as H
from ..structures import Boxes, ImageList, Instances, ShapeSpec
from .box_head import build_box_head
from .class_heads import OpenVocabularyClassifier
from .fast_rcnn_open_vocabulary import (InstanceRefinementOutputLayers, ObjectMiningOutputLayers,
                                        segment_offsets)
from .matcher import Matcher
from .poolers import ROIPooler


def build_roi_heads(cfg, input_shape):
    return ROI_HEADS_REGISTRY.get(cfg.MODEL.ROI_HEADS.NAME)(cfg, input_shape)


@torch.no_grad()
def get_image_level_gt(targets, num_classes):
    """roi_heads.py:158-174: per-image sorted unique GT classes + one-hot (N,K)."""
    if targets is None:
        return None, None, None
    gt_classes_img = [torch.unique(t.gt_classes, sorted=True) for t in targets]
    gt_classes_img_int = [gt.to(torch.int64) for gt in gt_classes_img]
    dev = targets[0].gt_classes.device
    gt_classes_img_oh = torch.cat(
        [torch.zeros((1, num_classes), dtype=torch.float, device=dev).scatter_(1, torch.unsqueeze(gt, dim=0), 1)
         for gt in gt_classes_img_int], dim=0)
    return gt_classes_img, gt_classes_img_int, gt_classes_img_oh


class ROIHeads(nn.Module):
    @configurable
    def __init__(self, *, num_classes, batch_size_per_image, positive_fraction, proposal_matcher,
                 proposal_append_gt=True, pixel_mean, pixel_std):
        super().__init__()
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction
        self.num_classes = num_classes
        self.proposal_matcher = proposal_matcher
        self.proposal_append_gt = proposal_append_gt
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std).view(-1, 1, 1), False)

    @classmethod
    def from_config(cls, cfg):
        return {
            "batch_size_per_image": cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE,
            "positive_fraction": cfg.MODEL.ROI_HEADS.POSITIVE_FRACTION,
            "num_classes": cfg.MODEL.ROI_HEADS.NUM_CLASSES,
            "proposal_append_gt": cfg.MODEL.ROI_HEADS.PROPOSAL_APPEND_GT,
            "proposal_matcher": Matcher(cfg.MODEL.ROI_HEADS.IOU_THRESHOLDS, cfg.MODEL.ROI_HEADS.IOU_LABELS,
                                        allow_low_quality_matches=False),
            "pixel_mean": cfg.MODEL.PIXEL_MEAN,
            "pixel_std": cfg.MODEL.PIXEL_STD,
        }


class PseudoTargets:
    """list[Instances] view of the mining kernel's output.  `packed` = (boxes (T,4), start (G) int32, count (G)
    int32) on the device: image g owns rows [start[g], start[g] + count[g])."""

    def __init__(self, mined, gt_off, image_sizes):
        self._o, self._off, self._sizes, self._items = mined, gt_off, image_sizes, None
        self.packed = (mined["pgt_boxes"], gt_off[:-1].contiguous(), mined["pgt_count"])

    def _materialise(self):
        if self._items is None:
            G = len(self._sizes)
            host = torch.cat([self._o["pgt_count"], self._off]).tolist()
            counts, offs = host[:G], host[G:]
            self._items = []
            for g, size in enumerate(self._sizes):
                sl = slice(offs[g], offs[g] + counts[g])
                self._items.append(Instances(size, gt_boxes=Boxes(self._o["pgt_boxes"][sl]),
                                             gt_classes=self._o["pgt_classes"][sl], gt_scores=self._o["pgt_scores"][sl],
                                             gt_weights=self._o["pgt_weights"][sl]))
        return self._items

    def __len__(self):
        return len(self._sizes)

    def __getitem__(self, i):
        return self._materialise()[i]

    def __iter__(self):
        return iter(self._materialise())


@ROI_HEADS_REGISTRY.register()
class WSOVODROIHeads(ROIHeads):
    @configurable
    def __init__(self, *, box_in_features: List[str], box_pooler: ROIPooler, box_head: nn.Module,
                 object_miner: nn.Module, sam=None, train_on_pred_boxes: bool = False, output_dir: str = None,
                 vis_test: bool = False, vis_period: int = 0, mrrp_on: bool = False, mrrp_num_branch: int = 3,
                 mrrp_fast: bool = False, refine_K: int = 4, refine_mist: bool = False,
                 refine_reg: List[bool] = [False, False, False, False],
                 box_refinery: List[nn.Module] = [None, None, None, None], sampling_on: bool = False,
                 proposal_matchers: List[Matcher] = [None, None, None, None],
                 batch_size_per_images: List[int] = [512, 512, 512, 512],
                 positive_sample_fractions: List[float] = [0.25, 0.25, 0.25, 0.25],
                 cls_agnostic_bbox_known: bool = False, pooler_type: str = "ROIPool", rpn_on: bool = False,
                 metadata: Dict = None, precision: str = "bf16", **kwargs):
        super().__init__(**kwargs)
        if mrrp_on or refine_mist or sam is not None or train_on_pred_boxes:
            raise NotImplementedError("MRRP / MIST refinement / in-loop SAM / TRAIN_ON_PRED_BOXES are outside the "
                                      "hot path (SURVEY section 2, F7)")
        self.in_features = self.box_in_features = box_in_features
        self.box_pooler = box_pooler
        self.box_head = box_head
        self.object_miner = object_miner
        self.sam = None
        self.iter = 0
        self.iter_test = 0
        self.epoch_test = 0
        self.refine_K = refine_K
        self.refine_reg = refine_reg
        self.box_refinery = box_refinery
        for k in range(self.refine_K):
            self.add_module("box_refinery_{}".format(k), self.box_refinery[k])
        self.sampling_on = sampling_on
        self.proposal_matchers = proposal_matchers
        self.batch_size_per_images = batch_size_per_images
        self.positive_sample_fractions = positive_sample_fractions
        self.cls_agnostic_bbox_known = cls_agnostic_bbox_known
        self.pooler_type = pooler_type
        self.rpn_on = rpn_on
        self.metadata = metadata
        self.precision = precision
        self.proposal_targets = None
        self.image_level_gt = None  # optional (cls_cat int64, offsets int32, onehot) precomputed without syncs

    @property
    def compute_dtype(self):
        return torch.bfloat16 if self.precision == "bf16" else torch.float32

    @property
    def pool_dtype(self):
        """What the pooler writes: the compute dtype, bf16x2 (hip_ops.X2) under the "parity" precision, unit-scale f16mx
        (hip_ops.MX) under "parity_mx"."""
        return H.X2 if H.x3_active() == "x2" else self.compute_dtype

    MX_MIN_ROWS = int(os.environ.get("WSOVOD_MX_MIN_ROWS", "4096"))

    def _pool_dtype_for(self, rows):
        """"parity_mx": the box head's FC layers take the f16mx kernel (one 256 x 256 tile shape) from 4096 pooled rows up --
        16 row tiles x 16 column tiles = one workgroup per CU; below, the bf16x2 path's split-K forms win (same mode, same
        bound)."""
        if H.mx_active() and self.pooler_type in ("ROIPool", "ROIAlignV2", "ROIAlign") and rows >= self.MX_MIN_ROWS:
            return H.MX
        return self.pool_dtype

    @classmethod
    def from_config(cls, cfg, input_shape):
        ret = super().from_config(cfg)
        ret["train_on_pred_boxes"] = cfg.MODEL.ROI_BOX_HEAD.TRAIN_ON_PRED_BOXES
        if cfg.WSOVOD.BBOX_REFINE.ENABLE:
            raise NotImplementedError("WSOVOD.BBOX_REFINE.ENABLE (SAM ViT in the training step) is out of scope; "
                                      "the hot path runs with BBOX_REFINE.ENABLE=False (SURVEY F7)")
        ret["sam"] = None
        if inspect.ismethod(cls._init_box_head):
            ret.update(cls._init_box_head(cfg, input_shape))
        return ret

    @classmethod
    def _init_box_head(cls, cfg, input_shape):
        in_features = cfg.MODEL.ROI_HEADS.IN_FEATURES
        pooler_resolution = cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION
        pooler_scales = tuple(1.0 / input_shape[k].stride for k in in_features)
        sampling_ratio = cfg.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO
        pooler_type = cfg.MODEL.ROI_BOX_HEAD.POOLER_TYPE
        in_channels = [input_shape[f].channels for f in in_features]
        assert len(set(in_channels)) == 1, in_channels
        in_channels = in_channels[0]
        box_pooler = ROIPooler(output_size=pooler_resolution, scales=pooler_scales, sampling_ratio=sampling_ratio,
                               pooler_type=pooler_type)
        box_head = build_box_head(cfg, ShapeSpec(channels=in_channels, height=pooler_resolution,
                                                 width=pooler_resolution))
        object_miner = ObjectMiningOutputLayers(cfg, box_head.output_shape, None)
        refine_K = cfg.WSOVOD.INSTANCE_REFINEMENT.REFINE_NUM
        refine_reg = cfg.WSOVOD.INSTANCE_REFINEMENT.REFINE_REG
        box_refinery = []
        for k in range(refine_K):
            head = OpenVocabularyClassifier(cfg, box_head.output_shape)
            box_refinery.append(InstanceRefinementOutputLayers(cfg, box_head.output_shape, k, head))
        sampling_on = cfg.WSOVOD.SAMPLING.SAMPLING_ON
        proposal_matchers = [None for _ in range(refine_K)]
        if sampling_on:
            for k in range(refine_K):
                proposal_matchers[k] = Matcher(cfg.WSOVOD.SAMPLING.IOU_THRESHOLDS[k],
                                               cfg.WSOVOD.SAMPLING.IOU_LABELS[k], allow_low_quality_matches=False)
        return {
            "box_in_features": in_features, "box_pooler": box_pooler, "box_head": box_head,
            "object_miner": object_miner, "output_dir": cfg.OUTPUT_DIR, "vis_test": cfg.VIS_TEST,
            "vis_period": cfg.VIS_PERIOD, "mrrp_on": cfg.MODEL.MRRP.MRRP_ON,
            "mrrp_num_branch": cfg.MODEL.MRRP.NUM_BRANCH, "mrrp_fast": cfg.MODEL.MRRP.TEST_BRANCH_IDX != -1,
            "refine_K": refine_K, "refine_mist": cfg.WSOVOD.INSTANCE_REFINEMENT.REFINE_MIST,
            "refine_reg": refine_reg, "box_refinery": box_refinery, "sampling_on": sampling_on,
            "proposal_matchers": proposal_matchers,
            "batch_size_per_images": cfg.WSOVOD.SAMPLING.BATCH_SIZE_PER_IMAGE,
            "positive_sample_fractions": cfg.WSOVOD.SAMPLING.POSITIVE_FRACTION,
            "cls_agnostic_bbox_known": cfg.WSOVOD.CLS_AGNOSTIC_BBOX_KNOWN, "pooler_type": pooler_type,
            "rpn_on": cfg.MODEL.PROPOSAL_GENERATOR.NAME != "PrecomputedProposals", "metadata": None,
            "precision": "parity" if cfg.MODEL.HIP.PRECISION in ("parity_train", "parity_mx") else cfg.MODEL.HIP.PRECISION,
        }

    # ------------------------------------------------------------------------------
    def forward(self, images: ImageList, features: Dict[str, torch.Tensor], proposals: List[Instances],
                data_aware_features=None, targets: Optional[List[Instances]] = None, classifier=None,
                append_background=True, file_names=None, loaded_proposals=None, pooled=None):
        return self._forward_impl(images, features, proposals, data_aware_features, targets, classifier,
                                  append_background, pooled)

    def _forward_impl(self, images, features, proposals, data_aware_features, targets, classifier, append_background,
                      pooled):
        if self.training:
            assert targets, "'targets' argument is required during training"
            if self.image_level_gt is not None:
                self._gt_cat, self._gt_off, self.gt_classes_img_oh = self.image_level_gt
                self.image_level_gt = None
            else:  # drop-in path: host round trip (torch.unique sizes), as in the reference
                _, gt_int, self.gt_classes_img_oh = get_image_level_gt(targets, self.num_classes)
                self._gt_cat = torch.cat(gt_int)
                self._gt_off = segment_offsets([len(g) for g in gt_int], self._gt_cat.device)
            # The reference first labels proposals against the dataset boxes
            # (label_and_sample_proposals, roi_heads.py:664); with weak supervision every field it
            # sets is overwritten by label_and_sample_proposals_wsl before any loss reads it, so
            # it is dead on this path and skipped.
        del targets
        self.images = images
        if self.training:
            losses = self._forward_box(features, proposals, data_aware_features, classifier, append_background,
                                       pooled=pooled)
            self.iter = self.iter + 1
            if self.iter_test > 0:
                self.epoch_test = self.epoch_test + 1
            self.iter_test = 0
            return proposals, losses
        pred_instances, all_scores, all_boxes = self._forward_box(features, proposals, data_aware_features,
                                                                  classifier, append_background)
        self.iter_test = self.iter_test + 1
        return pred_instances, {}, all_scores, all_boxes

    def pool_features(self, features, proposals):
        """Parameter-free part of the box branch: RoI pooling with the fused `* (objectness + 1)` scale
        (roi_heads.py:727-739)."""
        feats = [features[f] for f in self.box_in_features]
        # pooler-format rois and the scale `objectness + 1` in one launch on the concatenated boxes
        boxes = self.boxes_cat(proposals)
        rois, roi_scale = H.format_rois(boxes, segment_offsets([len(p) for p in proposals], boxes.device),
                                        H.cat_rows([x.objectness_logits for x in proposals]))
        if self.pooler_type == "ROILoopPool" and H.x3_active() == "x2":
            raise NotImplementedError('MODEL.HIP.PRECISION "parity" does not cover POOLER_TYPE ROILoopPool (use "bf16x3f")')
        if self.pooler_type == "ROILoopPool":  # (3R, C, 7, 7) = [region | frame | context], roi_heads.py:727-739
            out = self.box_pooler(feats, [x.proposal_boxes for x in proposals], out_dtype=torch.float32, rois=rois)
            return (out * roi_scale.repeat(3).view(-1, 1, 1, 1)).to(self.compute_dtype)
        Fn._WANT_HI.on = self.training and os.environ.get("WSOVOD_X2_HI", "1") != "0"  # (bf16x2 pooling only) a plain bf16 copy for fc1's dW
        try:
            return self.box_pooler(feats, [x.proposal_boxes for x in proposals], roi_scale=roi_scale,
                                   out_dtype=self._pool_dtype_for(int(rois.shape[0])), rois=rois)
        finally:
            Fn._WANT_HI.on = False

    def boxes_cat(self, proposals):
        """(sum R, 4) boxes of all images; concatenated once per step (pooling, mining and the box loss read it)."""
        tensors = [p.proposal_boxes.tensor for p in proposals]
        c = getattr(self, "_boxes_cat", None)
        if c is not None and len(c[0]) == len(tensors) and all(a is b for a, b in zip(c[0], tensors)):
            return c[1]
        cat = H.cat_rows(tensors)
        self._boxes_cat = (tensors, cat)
        return cat

    def get_features(self, features, proposals, data_aware_features=None, pooled=None):
        """roi_heads.py:1827-1857: pooled -> objectness scale -> neck -> (+ data-aware features)."""
        box_features = pooled if pooled is not None else self.pool_features(features, proposals)
        box_features = self.box_head(box_features)
        if self.pooler_type == "ROILoopPool":  # contextlocnet: the neck ran on region, frame and context rows
            parts = list(torch.chunk(box_features, 3, dim=0))
            if data_aware_features is not None:
                nums = [len(p) for p in proposals]
                daf = data_aware_features.to(torch.float32)
                if daf.size(0) == len(proposals) and daf.size(0) != sum(nums):
                    daf = torch.cat([daf[i:i + 1].expand(n, -1) for i, n in enumerate(nums)])
                parts = [(q.float() + daf).to(q.dtype) for q in parts]
            return parts
        if data_aware_features is not None:
            nums = [len(p) for p in proposals]
            dev = box_features.device
            if data_aware_features.size(0) == len(proposals) and data_aware_features.size(0) != sum(nums):
                seg = segment_offsets(nums, dev)  # per-image rows, broadcast in-kernel
                row_group = H.const_tensor([i for i, n in enumerate(nums) for _ in range(n)], torch.int32, dev)
            else:  # reference form: one row per proposal
                seg = torch.arange(sum(nums) + 1, dtype=torch.int32, device=dev)
                row_group = torch.arange(sum(nums), dtype=torch.int32, device=dev)
            box_features = Fn.add_group_rows(box_features, data_aware_features.to(torch.float32), row_group, seg)
        return box_features

    def _forward_box(self, features, proposals, data_aware_features=None, classifier=None, append_background=True,
                     pooled=None):
        box_features = self.get_features(features, proposals, data_aware_features, pooled=pooled)
        if self.pooler_type == "ROILoopPool":  # roi_heads.py:748-760: mining sees [region, frame, context]
            pre = None
            predictions = self.object_miner(box_features, proposals, context=True)
            box_features = box_features[0]
        else:
            pre = self._grouped_heads(box_features) if self.training else None
            predictions = self.object_miner(box_features, proposals, logits=pre[0] if pre else None)
        if not self.training:
            if self.refine_K <= 0:
                raise NotImplementedError("REFINE_NUM=0 inference is not used by any WSOVOD config")
            predictions_K = [self.box_refinery[k](box_features, classifier, append_background)
                             for k in range(self.refine_K)]
            pred_instances, _, all_scores, all_boxes = self.box_refinery[-1].inference(predictions_K, proposals)
            return pred_instances, all_scores, all_boxes

        losses = self.object_miner.losses(predictions, proposals, self.gt_classes_img_oh)
        self.pred_class_img_logits = self.object_miner.predict_probs_img(predictions, proposals).detach()
        prev_pred_scores = predictions[0].detach()  # (R,K); the appended zero bg column is never read
        proposal_boxes = prev_pred_boxes = self.boxes_cat(proposals)
        nums = [len(p) for p in proposals]
        seg = segment_offsets(nums, box_features.device)
        for k in range(self.refine_K):
            targets, proposals_k = self.mine_and_label(k, prev_pred_scores, prev_pred_boxes, proposals, seg, nums)
            predictions_k = self.box_refinery[k](box_features, classifier=classifier,
                                                 append_background=append_background, pre=pre[1 + k] if pre else None)
            losses.update(self.box_refinery[k].losses(predictions_k, proposals_k, self.num_classes,
                                                      proposal_boxes=proposal_boxes))
            if k + 1 < self.refine_K or self.rpn_on:
                prev_pred_scores = torch.softmax(predictions_k[0].detach(), dim=-1)
                prev_pred_boxes = torch.cat(self.box_refinery[k].predict_boxes(
                    (None, predictions_k[1].detach()), proposals_k), dim=0)
        if self.rpn_on:
            self.proposal_targets = self.rpn_targets(prev_pred_scores, prev_pred_boxes, proposals, seg)
        self._boxes_cat = None
        return losses

    def _grouped_heads(self, box_features):
        """Every Linear that reads the box features directly -- object mining [cls | det], each refinement head's
        box regression and first classifier projection -- as ONE autograd node (layers/functions.py:_LinearGroup):
        separate forward GEMMs, but a single input-gradient GEMM and a single weight-gradient contraction.
        Returns (mining logits, [(hidden_k, deltas_k)])."""
        if self.object_miner.num_classes == 1 or os.environ.get("WSOVOD_DISABLE_GROUP", "0") == "1":
            return None  # the K == 1 padding path keeps the per-module form (env switch: A/B measurements)
        om = self.object_miner
        # fp32 heads first: [cls | det] (two modules, one head of 2K rows) and every box regression ride in ONE forward
        # GEMM (the box features are streamed once for them); then the ReLU projections
        heads = [([om.cls.weight, om.det.weight], [om.cls.bias, om.det.bias], False, torch.float32)]
        regs = []
        for k in range(self.refine_K):
            r = self.box_refinery[k]
            regs.append(None)
            if r.refine_reg[r.refine_k]:
                heads.append((r.bbox_pred.weight, r.bbox_pred.bias, False, torch.float32))
                regs[-1] = len(heads) - 1
        join = list(range(len(heads)))
        biased = [h[1] is not None and all(b is not None for b in (h[1] if isinstance(h[1], list) else [h[1]])) for h in heads]
        joins = [join] if all(biased) and os.environ.get("WSOVOD_JOIN_HEADS", "1") != "0" else []  # (env: A/B runs)
        hids = []
        for k in range(self.refine_K):
            l1 = self.box_refinery[k].cls.projection[0]
            heads.append((l1.weight, l1.bias, True, H.X2 if H.x3_active() == "x2" else None))  # feeds the 2nd projection
            hids.append(len(heads) - 1)
        outs = Fn.linear_group(box_features, heads, joins=joins)
        return [outs[0]] + [(outs[hid], outs[reg] if reg is not None else None) for hid, reg in zip(hids, regs)]

    @torch.no_grad()
    def rpn_targets(self, prev_pred_scores, prev_pred_boxes, proposals, seg):
        """roi_heads.py:862-881: the pseudo ground truth the RPN is trained on = get_pgt_top_k(top_k=1) of the LAST
        refinement head's boxes and class probabilities.  Same mining kernel.  The result stays on the device
        (`.packed`); the per-image Instances of the reference's interface are cut only if somebody indexes the
        list (one host read of the counts)."""
        o = H.pgt_mine_and_label(prev_pred_scores.to(torch.float32), prev_pred_boxes, seg, self._gt_cat, self._gt_off,
                                 self.pred_class_img_logits, self.num_classes, 0.5)
        return PseudoTargets(o, self._gt_off, [p.image_size for p in proposals])

    def _sample_keys(self, num_rows, device):
        """Sort keys of the random sub-sampling (uniform keys = the reference's randperm; tests substitute the row
        index = a deterministic first-n rule on both sides)."""
        return torch.rand((num_rows,), device=device)

    @torch.no_grad()
    def mine_and_label(self, k, prev_pred_scores, prev_pred_boxes, proposals, seg, nums):
        """get_pgt_top_k (top_k=1, no SAM) + label_and_sample_proposals_wsl fused in one kernel.

        Returns (targets: per-image Instances{gt_boxes, gt_classes, gt_scores, gt_weights} as the
        reference builds at roi_heads.py:1316-1341, lazily sliced; proposals_k: per-image Instances with
        gt_classes / gt_boxes / gt_scores / gt_weights per proposal)."""
        if not self.sampling_on:
            raise NotImplementedError("WSOVOD.SAMPLING.SAMPLING_ON=False is not used by the WSR configs")
        m = self.proposal_matchers[k]
        assert len(m.thresholds) == 3 and m.labels == [0, 1], "hot path supports Matcher([thr], [0, 1])"
        o = H.pgt_mine_and_label(prev_pred_scores.to(torch.float32), prev_pred_boxes, seg, self._gt_cat,
                                 self._gt_off, self.pred_class_img_logits, self.num_classes, m.thresholds[1])
        if max(nums) > self.batch_size_per_images[k] or self.positive_sample_fractions[k] < 1.0:
            # _sample_proposals_wsl (roi_heads.py:1597-1610): rows outside the random sample are ignored (-1); every
            # row stays in place, so the boxes / scores / weights above are untouched
            o["gt_classes_all"] = o["gt_classes"]
            o["gt_classes"] = H.subsample_labels(o["gt_classes"], self._sample_keys(sum(nums), seg.device), seg,
                                                 max(nums), self.batch_size_per_images[k],
                                                 self.positive_sample_fractions[k], self.num_classes)
        self._last_pgt = o
        proposals_k = []
        start = 0
        for p, n in zip(proposals, nums):
            q = Instances(p.image_size, **p.get_fields())
            sl = slice(start, start + n)
            q.gt_classes = o["gt_classes"][sl]
            if not self.cls_agnostic_bbox_known:
                q.gt_boxes = Boxes(o["gt_boxes"][sl])
            q.gt_scores = o["gt_scores"][sl]
            q.gt_weights = o["gt_weights"][sl]
            proposals_k.append(q)
            start += n
        return o, proposals_k


@ROI_HEADS_REGISTRY.register()
class WSOVODMixedDatasetsROIHeads(WSOVODROIHeads):
    """Mixed-dataset variant (reference roi_heads.py:1860-3324): one object miner per dataset
    family (`object_miners[source_id]`, voc/coco/lvis share by name), the class count and the text-embedding
    `classifier` of the refinement head chosen per call.  Everything else is the single-dataset path."""

    @configurable
    def __init__(self, *, object_miners: nn.ModuleList, num_classes_list: List[int], **kwargs):
        self._source_id = 0
        super().__init__(object_miner=None, **kwargs)
        self.object_miners = object_miners
        self.num_classes_list = list(num_classes_list)

    @property
    def object_miner(self):
        return self.object_miners[self._source_id]

    @object_miner.setter
    def object_miner(self, value):  # the base class assigns None; the active miner is selected per call
        pass

    @classmethod
    def _init_box_head(cls, cfg, input_shape):
        ret = super()._init_box_head(cfg, input_shape)
        box_head = ret["box_head"]
        keys_map = {}
        for name in cfg.DATASETS.MIXED_DATASETS.NAMES:
            for fam in ("voc", "coco", "lvis"):
                if fam in name:
                    keys_map[name] = fam
        miner_dict = {}
        for name, num_classes in zip(cfg.DATASETS.MIXED_DATASETS.NAMES, cfg.DATASETS.MIXED_DATASETS.NUM_CLASSES):
            k = keys_map[name]
            if k not in miner_dict:
                miner_dict[k] = ObjectMiningOutputLayers(cfg, box_head.output_shape, class_head=None,
                                                         num_classes=num_classes)
        ret.pop("object_miner")
        ret["object_miners"] = nn.ModuleList([miner_dict[keys_map[n]] for n in cfg.DATASETS.MIXED_DATASETS.NAMES])
        ret["num_classes_list"] = cfg.DATASETS.MIXED_DATASETS.NUM_CLASSES
        return ret

    def select_source(self, source_id):
        self._source_id = int(source_id)
        self.num_classes = self.num_classes_list[self._source_id]

    def forward(self, images, features, proposals, data_aware_features=None, targets=None, classifier=None,
                source_id=0, append_background=True, file_names=None, loaded_proposals=None, pooled=None):
        if self.training:
            self.select_source(source_id)
        return self._forward_impl(images, features, proposals, data_aware_features, targets, classifier,
                                  append_background, pooled)
