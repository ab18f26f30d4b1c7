"""GeneralizedRCNN_WSOVOD on the HIP hot path.

Mirror of /root/reference/wsovod/modeling/meta_arch/rcnn_wsovod.py:27-344: same registry name,
`forward(batched_inputs) -> dict[str, loss]` in training / `inference(...)` in eval, same
`batched_inputs` format (`image` uint8 CHW BGR, `instances`, `proposals`, `height`/`width`), same
sub-module names (`backbone`, `data_aware_head`, `roi_heads`) so state dicts are interchangeable.
The proposals-only branch (rcnn_wsovod.py:198-204) is the hot path; an RPN
(`MODEL.PROPOSAL_GENERATOR.NAME != "PrecomputedProposals"`) is a SURVEY 8f "next" row.
"""
import contextlib
import logging
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import nn

from ..config import BACKBONE_REGISTRY, META_ARCH_REGISTRY, configurable
from ..layers import hip_ops as H
from ..structures import Boxes, ImageList, Instances, ShapeSpec
from .class_heads import DataAwareFeaturesHead
from .fast_rcnn_open_vocabulary import segment_offsets
from .proposal_generator import build_proposal_generator
from .roi_heads import build_roi_heads

__all__ = ["GeneralizedRCNN_WSOVOD", "GeneralizedRCNN_WSOVOD_MixedDatasets", "build_model", "build_backbone"]


def build_backbone(cfg, input_shape=None):
    if input_shape is None:
        input_shape = ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))
    return BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, input_shape)


def build_model(cfg):
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model


def detector_postprocess(results: Instances, output_height: int, output_width: int):
    """wsovod/modeling/postprocessing.py:8-82 (boxes only): rescale to the requested output size."""
    scale_x, scale_y = output_width / results.image_size[1], output_height / results.image_size[0]
    out = Instances((output_height, output_width), **results.get_fields())
    boxes = out.pred_boxes.clone()
    boxes.scale(scale_x, scale_y)
    boxes.clip(out.image_size)
    out.pred_boxes = boxes
    return out[boxes.nonempty()]


@META_ARCH_REGISTRY.register()
class GeneralizedRCNN_WSOVOD(nn.Module):
    @configurable
    def __init__(self, *, cfg=None, backbone, data_aware_head=None, proposal_generator, roi_heads,
                 pixel_mean: Tuple[float], pixel_std: Tuple[float], input_format: Optional[str] = None,
                 vis_period: int = 0):
        super().__init__()
        self.cfg = cfg
        self.backbone = backbone
        self.data_aware_head = data_aware_head
        self.proposal_generator = proposal_generator
        self.roi_heads = roi_heads
        self.input_format = input_format
        self.vis_period = 0
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std).view(-1, 1, 1), False)
        self._mean, self._std = [float(v) for v in pixel_mean], [float(v) for v in pixel_std]
        self.logger = logging.getLogger(__name__)
        self.classifier = None
        # MODEL.HIP.PRECISION = "parity_train": the heads' backward keeps the hi/lo split (layers/functions.py)
        self.backward_split = bool(cfg is not None and cfg.MODEL.HIP.PRECISION == "parity_train")
        # "parity_mx" (round 6): the parity forward with the res4 / res5 convs and the box head's FC layers on the block-scaled
        # f16mx kernels (layers/hip_ops.py:mx_mode); every module sees "parity", this flag selects the kernels
        self.mx = bool(cfg is not None and cfg.MODEL.HIP.PRECISION == "parity_mx")

    @classmethod
    def from_config(cls, cfg):
        backbone = build_backbone(cfg)
        return {
            "cfg": cfg, "backbone": backbone,
            "data_aware_head": DataAwareFeaturesHead(cfg, backbone.output_shape())
            if cfg.MODEL.ROI_BOX_HEAD.OPEN_VOCABULARY.DATA_AWARE else None,
            "proposal_generator": build_proposal_generator(cfg, backbone.output_shape()), "roi_heads": build_roi_heads(cfg, backbone.output_shape()),
            "input_format": cfg.INPUT.FORMAT, "vis_period": cfg.VIS_PERIOD,
            "pixel_mean": cfg.MODEL.PIXEL_MEAN, "pixel_std": cfg.MODEL.PIXEL_STD,
        }

    @property
    def device(self):
        return self.pixel_mean.device

    # ---- input staging (H2D + padding are plumbing; normalisation is fused into the stem) ----
    def _canvas(self, batched_inputs):
        images = [x["image"] for x in batched_inputs]
        sizes = [(int(im.shape[-2]), int(im.shape[-1])) for im in images]
        Hp, Wp = max(s[0] for s in sizes), max(s[1] for s in sizes)
        if all(im.device == self.device for im in images) and all(s == sizes[0] for s in sizes):
            canvas = self._adjacent(images)  # already resident as consecutive slices of one batch tensor: no pass at all
            if canvas is None:
                canvas = torch.stack(images)  # already resident: one gather pass
        elif all(s == sizes[0] for s in sizes):
            # host images (the DatasetMapper's format): each one is copied straight into its slot of the batch tensor --
            # no per-image device temporary and no second (stack) pass over the batch
            canvas = torch.empty((len(images), 3, Hp, Wp), dtype=images[0].dtype, device=self.device)
            for i, im in enumerate(images):
                canvas[i].copy_(im, non_blocking=True)
        else:
            canvas = torch.zeros((len(images), 3, Hp, Wp), dtype=torch.uint8, device=self.device)
            for i, im in enumerate(images):
                canvas[i, :, : sizes[i][0], : sizes[i][1]].copy_(im, non_blocking=True)
        if canvas.dtype != torch.uint8:
            raise RuntimeError("GeneralizedRCNN_WSOVOD expects uint8 CHW images (DatasetMapper format)")
        sizes_t = H.const_tensor([v for s in sizes for v in s], torch.int32, self.device).view(-1, 2)
        return canvas.contiguous(), sizes_t, sizes

    @staticmethod
    def _adjacent(images):
        """The (N,3,H,W) view over `images` when they are contiguous, equally shaped slices i = 0..N-1 of one allocation
        (a collated batch handed over as per-image views, the reference's list-of-dicts format), else None."""
        im0 = images[0]
        n = im0.numel()
        if not im0.is_contiguous() or n == 0:
            return None
        base, esz, st = im0.data_ptr(), im0.element_size(), im0.untyped_storage()
        for i, im in enumerate(images):
            if im.shape != im0.shape or not im.is_contiguous() or im.dtype != im0.dtype \
                    or im.data_ptr() != base + i * n * esz or im.untyped_storage().data_ptr() != st.data_ptr():
                return None
        if im0.storage_offset() + len(images) * n > st.nbytes() // esz:
            return None
        return im0.as_strided((len(images),) + tuple(im0.shape), (n,) + tuple(im0.stride()), im0.storage_offset())

    def preprocess_image(self, batched_inputs):
        """rcnn_wsovod.py:321-328: normalise, pad and batch -> ImageList (fp32 NCHW)."""
        canvas, sizes_t, sizes = self._canvas(batched_inputs)
        return ImageList(H.preprocess_image(canvas, sizes_t, self._mean, self._std), sizes)

    def _proposals(self, batched_inputs):
        assert "proposals" in batched_inputs[0]
        proposals = [x["proposals"].to(self.device) for x in batched_inputs]
        if self.proposal_generator is None:  # rcnn_wsovod.py:198-203; with an RPN the loaded boxes stay as they are
            level_ids = torch.zeros((sum(len(p) for p in proposals),), dtype=torch.int64, device=self.device)
            for p, ids in zip(proposals, level_ids.split([len(p) for p in proposals])):  # one fill, per-image views
                p.level_ids = ids
        return proposals

    def _rpn_proposals(self, images, features, batched_inputs, loaded, gt_instances=None):
        """rcnn_wsovod.py:177-197 / :267-283: RPN boxes (objectness = sigmoid, ramped by iter / MAX_ITER in
        training) followed by the loaded SAM boxes of the image."""
        proposals, _ = self.proposal_generator(images, features, gt_instances)
        ramp = self.roi_heads.iter / self.cfg.SOLVER.MAX_ITER if self.training else 1.0
        for p in proposals:
            p.objectness_logits = torch.sigmoid(p.objectness_logits) * ramp
        self.rpn_proposals = proposals
        if loaded is not None:
            proposals = [Instances.cat([p1, p2]) for p1, p2 in zip(proposals, loaded)]
        return proposals

    _step_meta = None  # set while a whole-step HIP graph is captured: the image-level labels are static input buffers

    def _image_level_gt(self, batched_inputs):
        """get_image_level_gt (roi_heads.py:158-174) on the host copies: no device sync."""
        if self._step_meta is not None:
            return self._step_meta.image_level_gt()
        K = self.roi_heads.num_classes
        cls_list, oh = [], torch.zeros((len(batched_inputs), K), dtype=torch.float32)
        for i, x in enumerate(batched_inputs):
            gc = x["instances"].gt_classes
            if gc.is_cuda:
                return None
            u = torch.unique(gc, sorted=True).to(torch.int64)
            cls_list.append(u)
            oh[i, u] = 1.0
        cat = H.h2d_small(torch.cat(cls_list), self.device)
        off = segment_offsets([len(u) for u in cls_list], self.device)
        return cat, off, H.h2d_small(oh, self.device)

    def forward(self, batched_inputs, classifier=None):
        if not self.training:
            return self.inference(batched_inputs, classifier=classifier)
        return self.forward_trainable(self.forward_frozen(batched_inputs))

    @property
    def x3(self):
        """MODEL.HIP.PRECISION "bf16x3" / "bf16x3f": fp32 tensors, contractions on the bf16 MFMA kernels over hi/lo-split
        operands (layers/hip_ops.py:x3_mode; "f" = in the forward pass only) -- the fast modes that keep the north star's
        1e-3 logit bound.  "parity" ("x2"): the forward split on bf16x2 activations (hip_ops.X2), produced by the kernels
        themselves -- no stand-alone split passes, no fp32 activation traffic -- and a plain bf16 backward."""
        return {"bf16x3": "full", "bf16x3f": "fwd", "parity": "x2"}.get(getattr(self.backbone, "precision", "bf16"), False)

    @torch.no_grad()
    def forward_frozen(self, batched_inputs):
        with H.x3_mode(self.x3), H.mx_mode(getattr(self, "mx", False)):
            return self._forward_frozen(batched_inputs)

    def forward_trainable(self, st):
        from ..layers.functions import backward_split

        with H.x3_mode(self.x3), backward_split(self.backward_split), H.mx_mode(getattr(self, "mx", False)):
            return self._forward_trainable(st)

    @torch.no_grad()
    def inference(self, batched_inputs, detected_instances=None, do_postprocess=True, classifier=None):
        pre = getattr(self, "_pre_inference", None)  # an overlapped trainer applies its pending update first (eval / TTA
        if pre is not None:                          # hooks between steps must see the weights after optimizer.step())
            pre()
        with H.x3_mode(self.x3), H.mx_mode(getattr(self, "mx", False)):
            return self._inference(batched_inputs, detected_instances, do_postprocess, classifier)

    def _forward_frozen(self, batched_inputs):
        """Everything of the training forward that touches NO trainable parameter: input staging, the
        frozen backbone, GAP for the data-aware head, RoI pooling with the objectness scale.  Because it
        does not depend on the weights being updated, a data-parallel trainer may run it while the previous
        step's gradient all-reduce is still in flight (wsovod_amd/engine/trainer.py)."""
        canvas, sizes_t, sizes = self._canvas(batched_inputs)
        st = {"canvas": canvas, "sizes": sizes, "gt_instances": None, "image_level_gt": None}
        self._select_source(batched_inputs, st)
        if "instances" in batched_inputs[0]:
            st["image_level_gt"] = self._image_level_gt(batched_inputs)
            # With the image-level labels already extracted on the host, the heads never read the dataset
            # boxes (weak supervision), so the instances are not copied to the device (a pageable H2D copy
            # would block the host until the stream drains).
            st["gt_instances"] = [x["instances"] if st["image_level_gt"] is not None else x["instances"].to(self.device)
                                  for x in batched_inputs]
        st["proposals"] = self._proposals(batched_inputs) if "proposals" in batched_inputs[0] else None
        if getattr(self.backbone, "has_trainable_stage", False):
            # MODEL.BACKBONE.FREEZE_AT < 5: the backbone reads weights the optimizer updates and takes part in autograd --
            # it runs in the trainable part (behind the pending update of an overlapped trainer), with gradients enabled
            st["sizes_t"] = sizes_t
            st["features"] = st["gaps"] = st["pooled"] = None
            return st
        features = self.backbone.forward_uint8(canvas, sizes_t, self._mean, self._std, allow_graph=True)
        st["features"] = features
        pool_here = self.proposal_generator is None  # with an RPN the box set depends on trainable weights: pooling moves
        fuse = contextlib.nullcontext()               # to the trainable part
        if pool_here and self.data_aware_head is not None and getattr(self.roi_heads, "pooler_type", "") == "ROIPool" \
                and st["proposals"] is not None:
            # the GAP of the data-aware head and the RoIPool pre-pass read the same res5 map: one pass (layers/hip_ops.py)
            res = self.roi_heads.box_pooler.output_size
            fuse = H.gap_with_pool_prepass(sum(len(p) for p in st["proposals"]), (res, res) if isinstance(res, int) else res)
        with fuse:
            st["gaps"] = self.data_aware_head.pooled_stats(features) if self.data_aware_head is not None else None
            st["pooled"] = self.roi_heads.pool_features(features, st["proposals"]) if pool_here else None
        return st

    def _heads_kwargs(self, st):
        return {}

    def _select_source(self, batched_inputs, st):
        pass

    def _forward_trainable(self, st):
        """The trainable remainder: data-aware MLP, neck, object mining, refinement, losses."""
        self.roi_heads.image_level_gt = st["image_level_gt"]
        if st["features"] is None:  # a trainable backbone stage: see _forward_frozen
            st["features"] = self.backbone.forward_uint8(st["canvas"], st["sizes_t"], self._mean, self._std)
            st["gaps"] = self.data_aware_head.pooled_stats(st["features"]) if self.data_aware_head is not None else None
            if self.proposal_generator is None:
                st["pooled"] = self.roi_heads.pool_features(st["features"], st["proposals"])
        daf = self.data_aware_head.from_stats(st["gaps"]) if self.data_aware_head is not None else None
        images = ImageList(st["canvas"], st["sizes"])
        proposals = st["proposals"]
        if self.proposal_generator is not None:
            proposals = self._rpn_proposals(images, st["features"], None, st["proposals"], st["gt_instances"])
        _, detector_losses = self.roi_heads(images, st["features"], proposals, daf, st["gt_instances"],
                                            append_background=True, loaded_proposals=st["proposals"],
                                            pooled=st["pooled"], **self._heads_kwargs(st))
        losses = {}
        losses.update(detector_losses)
        if self.proposal_generator is not None:  # rcnn_wsovod.py:222-223: trained from the heads' pseudo GT
            losses.update(self.proposal_generator.get_losses(self.roi_heads.proposal_targets))
        return losses

    def _inference(self, batched_inputs, detected_instances=None, do_postprocess=True, classifier=None):
        assert not self.training
        assert detected_instances is None, "forward_with_given_boxes is not on the hot path"
        canvas, sizes_t, sizes = self._canvas(batched_inputs)
        features = self.backbone.forward_uint8(canvas, sizes_t, self._mean, self._std)
        proposals = self._proposals(batched_inputs) if "proposals" in batched_inputs[0] else None
        if self.proposal_generator is not None:
            proposals = self._rpn_proposals(ImageList(canvas, sizes), features, batched_inputs, proposals)
        daf = self.data_aware_head.forward_per_image(features) if self.data_aware_head is not None else None
        if classifier is not None:
            self.classifier = classifier
        elif self.classifier is None:
            weight_path = self.cfg.MODEL.ROI_BOX_HEAD.OPEN_VOCABULARY.WEIGHT_PATH_TEST
            self.classifier = torch.as_tensor(np.load(weight_path, encoding="bytes", allow_pickle=True)).to(
                torch.float32).contiguous().to(self.device)
        results, _, all_scores, all_boxes = self.roi_heads(ImageList(canvas, sizes), features, proposals, daf, None,
                                                           self.classifier, append_background=True)
        if do_postprocess:
            return GeneralizedRCNN_WSOVOD._postprocess(results, batched_inputs, sizes)
        return results, all_scores, all_boxes

    @staticmethod
    def _postprocess(instances, batched_inputs, image_sizes):
        packed = getattr(instances, "packed", None)
        if packed is not None and len(instances):
            return GeneralizedRCNN_WSOVOD._postprocess_packed(instances, packed, batched_inputs, image_sizes)
        processed_results = []
        for results_per_image, input_per_image, image_size in zip(instances, batched_inputs, image_sizes):
            height = input_per_image.get("height", image_size[0])
            width = input_per_image.get("width", image_size[1])
            processed_results.append({"instances": detector_postprocess(results_per_image, height, width)})
        return processed_results


def _postprocess_packed(instances, packed, batched_inputs, image_sizes):
    """detector_postprocess (postprocessing.py:8-82, boxes only) for every image of the batch in one pass over the packed
    detections of the batched tail: rescale to the requested output size, clip, drop empty boxes (order kept) -- the
    same arithmetic per box as the per-image form, one host read (the surviving counts) instead of two per image."""
    det_box, det_sc, det_cls, det_prop, counts = packed
    N, k = det_sc.shape
    dev = det_box.device
    out_hw = [(int(inp.get("height", sz[0])), int(inp.get("width", sz[1]))) for inp, sz in zip(batched_inputs, image_sizes)]
    scale = H.const_tensor([v for (oh, ow), r in zip(out_hw, instances)
                            for v in (ow / r.image_size[1], oh / r.image_size[0]) * 2], torch.float32, dev).view(N, 1, 4)
    lim = H.const_tensor([float(v) for oh, ow in out_hw for v in (ow, oh, ow, oh)], torch.float32, dev).view(N, 1, 4)
    bx = torch.minimum((det_box * scale).clamp(min=0), lim)
    live = torch.arange(k, device=dev)[None] < H.const_tensor(counts, torch.int64, dev)[:, None]
    keep = live & ((bx[..., 2] - bx[..., 0]) > 0) & ((bx[..., 3] - bx[..., 1]) > 0)
    perm = torch.sort((~keep).to(torch.uint8), dim=1, stable=True).indices  # kept detections first, order preserved
    n_keep = keep.sum(dim=1).tolist()
    bx = bx.gather(1, perm[..., None].expand(N, k, 4))
    sc, cl, pi = det_sc.gather(1, perm), det_cls.gather(1, perm), det_prop.gather(1, perm)
    out = []
    for i, n in enumerate(n_keep):
        r = Instances(out_hw[i])
        r.pred_boxes = Boxes(bx[i, :n])
        r.scores, r.pred_classes, r.pred_inds = sc[i, :n], cl[i, :n], pi[i, :n]
        out.append({"instances": r})
    return out


GeneralizedRCNN_WSOVOD._postprocess_packed = staticmethod(_postprocess_packed)


@META_ARCH_REGISTRY.register()
class GeneralizedRCNN_WSOVOD_MixedDatasets(GeneralizedRCNN_WSOVOD):
    """meta_arch/rcnn_wsovod_mixed_datasets.py:28-367: a batch comes from ONE dataset (`dataset_id`); its class
    count selects the object miner and its CLIP text embeddings are handed to the refinement head per call."""

    @configurable
    def __init__(self, *, classifier_train=(), classifier_test=None, **kwargs):
        super().__init__(**kwargs)
        self.classifier_train = list(classifier_train)
        self.classifier = classifier_test

    @property
    def classifier_test(self):  # the reference's attribute name (rcnn_wsovod_mixed_datasets.py:81,319-331)
        return self.classifier

    @classmethod
    def from_config(cls, cfg):
        ret = super().from_config(cfg)

        def load(path):
            return torch.as_tensor(np.load(path, encoding="bytes", allow_pickle=True)).to(torch.float32).contiguous().to(
                cfg.MODEL.DEVICE)

        ret["classifier_train"] = [load(p) for p in cfg.DATASETS.MIXED_DATASETS.WEIGHT_PATH_TRAINS]
        ret["classifier_test"] = load(cfg.MODEL.ROI_BOX_HEAD.OPEN_VOCABULARY.WEIGHT_PATH_TEST)
        return ret

    def _select_source(self, batched_inputs, st):
        st["source_id"] = int(batched_inputs[0].get("dataset_id", 0))
        if self.training:
            self.roi_heads.select_source(st["source_id"])

    def _heads_kwargs(self, st):
        return {"classifier": self.classifier_train[st["source_id"]], "source_id": st["source_id"]}
