"""RPN branch of the shipped configs on the HIP kernels.

Mirrors /root/reference/wsovod/modeling/proposal_generator/rpn.py:90-515 (`WSOVODRPN_V2`),
proposal_utils.py:26-144 (`find_top_rpn_proposals`) and detectron2's `StandardRPNHead` (un-vendored; SURVEY
Appendix A): same class / registry names, constructor arguments, state-dict keys (`rpn_head.conv`,
`rpn_head.objectness_logits`, `rpn_head.anchor_deltas`) and loss names.

Compute: the 3x3 conv (512 -> 512, ReLU) is the implicit-GEMM conv kernel on the NHWC feature map; the two 1x1
heads are ONE GEMM over the (N*H*W, C) rows with the weights stacked (A objectness + 4A delta columns), whose
NHWC output already is the (N, H*W*A[, 4]) layout the reference reaches by permuting NCHW tensors.  Proposal
selection = device sort + `wsovod_rpn_decode` (decode, clip, min-size flags for the pre-NMS top-k only) +
`wsovod_nms_segments` (one segment per image) + ONE host read of the per-image keep counts.  Backward: only the
sampled anchors (<= BATCH_SIZE_PER_IMAGE per image) carry a loss, so the weight gradients are GEMMs over those
rows alone -- patch rows gathered by `wsovod_im2col_rows`; the backbone is frozen, no input gradient.
"""
from typing import Dict, List, Optional, Tuple, Union

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function

from ..config import Registry, configurable
from ..layers import hip_ops as H
from ..layers.functions import scale_losses
from ..structures import Boxes, ImageList, Instances, ShapeSpec
from .anchor_generator import build_anchor_generator
from .backbone import Conv2d, hip_conv
from .box_regression import Box2BoxTransform
from .matcher import Matcher

PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")
RPN_HEAD_REGISTRY = Registry("RPN_HEAD")

__all__ = ["StandardRPNHead", "WSOVODRPN_V2", "build_proposal_generator", "build_rpn_head", "find_top_rpn_proposals"]


def _pad(n, m):
    return (n + m - 1) // m * m


class _RPNHeadFn(Function):
    """conv3x3+ReLU -> stacked 1x1 heads, one image batch.  Returns (N*H*W, NP) fp32 (NP = 5A padded to 8)."""

    @staticmethod
    def forward(ctx, x_nhwc, conv, w_conv, b_conv, w_obj, b_obj, w_del, b_del, max_active):  # conv: module (folded cache)
        with torch.no_grad():
            cd = x_nhwc.dtype
            if H.x3_active() == "x2":  # "parity": res5 leaves the backbone as real fp32; the conv kernel reads bf16x2
                xe = H.x2_encode(x_nhwc.view(-1, x_nhwc.shape[-1])).view(x_nhwc.shape)
                h = hip_conv(xe, conv, relu=True, out_fp32=True)
            else:
                h = hip_conv(x_nhwc, conv, relu=True)  # (N,H,W,C)
            Cc = h.shape[-1]
            h2 = h.view(-1, Cc)
            A = w_obj.shape[0]
            NP = _pad(5 * A, 8)
            wcat = torch.zeros((NP, Cc), dtype=cd, device=h.device)
            wcat[:A] = w_obj.view(A, Cc)
            wcat[A:5 * A] = w_del.view(4 * A, Cc)
            bcat = torch.zeros((NP,), dtype=torch.float32, device=h.device)
            bcat[:A] = b_obj
            bcat[A:5 * A] = b_del
            out = H.gemm_nt(h2, wcat, bias=bcat, out_dtype=torch.float32)
        ctx.save_for_backward(x_nhwc, h2, wcat)
        ctx.conv, ctx.A, ctx.max_active = conv, A, max_active
        ctx.shapes = (w_obj.shape, w_del.shape)
        ctx.x3 = H.x3_active()
        return out

    @staticmethod
    def backward(ctx, dout):
        with H.x3_mode(bool(ctx.x3)):  # (the RPN head keeps the split in its backward also in the "fwd" mode)
            return _RPNHeadFn._backward(ctx, dout)

    @staticmethod
    def _backward(ctx, dout):
        x, h2, wcat = ctx.saved_tensors
        conv, A = ctx.conv, ctx.A
        cd = x.dtype
        Cc = h2.shape[1]
        NP = wcat.shape[0]
        dout = dout.contiguous()
        # rows that carry a gradient (the sampled anchors' pixels); fixed-size list, no host sync
        active = (dout != 0).any(dim=1)
        rows = torch.nonzero_static(active, size=min(ctx.max_active, dout.shape[0]), fill_value=-1).view(-1)
        live = (rows >= 0)
        safe = rows.clamp(min=0)
        d_act = (dout[safe] * live[:, None]).to(cd)  # (n, NP)
        h_act = h2[safe]  # (n, C)
        n = rows.numel()
        Mp = _pad(n, 64)
        # stacked 1x1 heads: dW = d^T h, db = colsum(d)
        dT = H.transpose_cast(d_act, cd, ld_dst=Mp)  # (NP, Mp)
        hT = H.transpose_cast(h_act, cd, ld_dst=Mp)  # (C, Mp)
        dwcat = H.gemm_nt(dT, hT, out_dtype=torch.float32)  # (NP, C)
        seg = H.const_tensor((0, n), torch.int32, x.device)
        dbcat = H.segment_colsum(d_act, seg).view(NP)
        # hidden gradient at the active rows, ReLU mask fused: dH = (d W) * [h > 0]
        wcatT = H.transpose_cast(wcat, cd)  # (C, NP)
        dh = H.gemm_nt(d_act, wcatT, out_dtype=cd, mask_src=h_act, mask_scale=1.0)  # (n, C)
        # 3x3 conv: dW[co][tap][ci] = sum_rows dH[row][co] * patch[row][tap][ci]
        patches = H.im2col_rows(x, rows, conv.kernel_size, conv.stride, conv.padding, conv.dilation)  # (n, 9C)
        dhT = H.transpose_cast(dh, cd, ld_dst=Mp)  # (C, Mp)
        pT = H.transpose_cast(patches, cd, ld_dst=Mp)  # (9C, Mp)
        dwc = H.gemm_nt(dhT, pT, out_dtype=torch.float32)  # (Cout, 9*Cin) in [kh][kw][ci] order
        k = conv.kernel_size
        dw_conv = dwc.view(Cc, k, k, x.shape[-1]).permute(0, 3, 1, 2).contiguous()
        db_conv = H.segment_colsum(dh, seg).view(Cc)
        so, sd = ctx.shapes
        return (None, None, dw_conv, db_conv, dwcat[:A].reshape(so), dbcat[:A].clone(),
                dwcat[A:5 * A].reshape(sd), dbcat[A:5 * A].clone(), None)


@RPN_HEAD_REGISTRY.register()
class StandardRPNHead(nn.Module):
    """detectron2 StandardRPNHead: 3x3 conv + ReLU, then 1x1 objectness (A) and 1x1 anchor deltas (4A)."""

    @configurable
    def __init__(self, *, in_channels: int, num_anchors: int, box_dim: int = 4, conv_dims: List[int] = (-1,),
                 batch_rows: int = 512):
        super().__init__()
        if len(conv_dims) != 1 or conv_dims[0] not in (-1, in_channels):
            raise NotImplementedError("RPN.CONV_DIMS other than [-1] is not used by any WSOVOD config")
        assert box_dim == 4
        self.conv = Conv2d(in_channels, in_channels, 3, stride=1, padding=1, bias=True)
        self.objectness_logits = nn.Conv2d(in_channels, num_anchors, kernel_size=1, stride=1)
        self.anchor_deltas = nn.Conv2d(in_channels, num_anchors * box_dim, kernel_size=1, stride=1)
        for layer in (self.conv, self.objectness_logits, self.anchor_deltas):
            nn.init.normal_(layer.weight, std=0.01)
            nn.init.constant_(layer.bias, 0)
        self.num_anchors = num_anchors
        self.batch_rows = batch_rows

    @classmethod
    def from_config(cls, cfg, input_shape):
        in_channels = [s.channels for s in input_shape]
        assert len(set(in_channels)) == 1, "Each level must have the same channel!"
        anchor_generator = build_anchor_generator(cfg, input_shape)
        num_anchors, box_dim = anchor_generator.num_anchors, anchor_generator.box_dim
        assert len(set(num_anchors)) == 1, "Each level must have the same number of anchors per spatial position"
        return {"in_channels": in_channels[0], "num_anchors": num_anchors[0], "box_dim": box_dim,
                "conv_dims": cfg.MODEL.RPN.CONV_DIMS, "batch_rows": cfg.MODEL.RPN.BATCH_SIZE_PER_IMAGE}

    def forward_nhwc(self, feature):
        """feature: NCHW-logical / channels_last map -> ((N, H*W*A) logits, (N, H*W*A, 4) deltas), fp32."""
        x = feature.permute(0, 2, 3, 1)
        if not x.is_contiguous():
            x = x.contiguous()
        N, Hh, Ww, _ = x.shape
        A = self.num_anchors
        out = _RPNHeadFn.apply(x, self.conv, self.conv.weight, self.conv.bias, self.objectness_logits.weight,
                               self.objectness_logits.bias, self.anchor_deltas.weight, self.anchor_deltas.bias,
                               N * self.batch_rows)
        logits = out[:, :A].reshape(N, Hh * Ww * A)
        deltas = out[:, A:5 * A].reshape(N, Hh * Ww * A, 4)
        return logits, deltas

    def forward(self, features: List[torch.Tensor]):
        """detectron2 surface: per level (N, A, H, W) logits and (N, 4A, H, W) deltas (views of the NHWC result)."""
        lo, de = [], []
        for f in features:
            N, _, Hh, Ww = f.shape
            logits, deltas = self.forward_nhwc(f)
            A = self.num_anchors
            lo.append(logits.view(N, Hh, Ww, A).permute(0, 3, 1, 2))
            de.append(deltas.view(N, Hh, Ww, A * 4).permute(0, 3, 1, 2))
        return lo, de


def build_rpn_head(cfg, input_shape):
    return RPN_HEAD_REGISTRY.get(cfg.MODEL.RPN.HEAD_NAME)(cfg, input_shape)


def find_top_rpn_proposals(anchors: List[Boxes], pred_objectness_logits: List[torch.Tensor],
                           pred_anchor_deltas: List[torch.Tensor], image_sizes: List[Tuple[int, int]],
                           box2box_transform, nms_thresh: float, pre_nms_topk: int, post_nms_topk: int,
                           min_box_size: float, training: bool):
    """proposal_utils.py:26-144 with the decode of rpn.py:495-515 folded in (only the pre-NMS top-k anchors are
    decoded).  Per level: sort logits, keep the top-k, decode + clip + size test; per image: NMS inside each level,
    keep post_nms_topk by score.  Returns list[Instances{proposal_boxes, objectness_logits}] sorted by score."""
    assert len(anchors) == 1, "single-level RPN (res5) is the only form the WSOVOD configs use"
    num_images = len(image_sizes)
    dev = pred_objectness_logits[0].device
    sizes_t = H.const_tensor([float(v) for hw in image_sizes for v in hw], torch.float32, dev).view(-1, 2)
    logits_i, deltas_i, anchors_i = pred_objectness_logits[0], pred_anchor_deltas[0], anchors[0].tensor
    k = min(logits_i.shape[1], pre_nms_topk)
    sorted_logits, idx = logits_i.detach().sort(descending=True, dim=1)
    topk_scores, topk_idx = sorted_logits[:, :k].contiguous(), idx[:, :k].contiguous()
    boxes, valid = H.rpn_decode(anchors_i, deltas_i.detach(), topk_idx, sizes_t, box2box_transform.weights,
                                box2box_transform.scale_clamp, min_box_size)
    finite_scores = torch.isfinite(topk_scores)
    valid = valid & finite_scores
    seg = H.const_tensor([n * k for n in range(num_images + 1)], torch.int32, dev)
    keep, count = H.nms_segments(boxes.view(-1, 4), seg, k, nms_thresh, post_nms_topk, valid=valid)
    bad = (~torch.isfinite(deltas_i.detach())).any() | (~finite_scores).any()
    host = torch.cat([count, bad.to(torch.int32).view(1)]).tolist()  # the one host read of this stage
    if training and host[-1]:
        raise FloatingPointError("Predicted boxes or scores contain Inf/NaN. Training has diverged.")
    results = []
    for n, image_size in enumerate(image_sizes):
        kept = keep[n * k:n * k + host[n]].long()
        res = Instances(image_size)
        res.proposal_boxes = Boxes(boxes[n][kept])
        res.objectness_logits = topk_scores[n][kept]
        results.append(res)
    return results


@PROPOSAL_GENERATOR_REGISTRY.register()
class WSOVODRPN_V2(nn.Module):
    """rpn.py:90-515.  Trained from the ROI heads' pseudo ground truth: `forward` keeps the predictions,
    `get_losses(proposal_targets)` is called by the meta-architecture after the ROI heads ran."""

    @configurable
    def __init__(self, *, in_features: List[str], head: nn.Module, anchor_generator: nn.Module, anchor_matcher: Matcher,
                 box2box_transform: Box2BoxTransform, batch_size_per_image: int, positive_fraction: float,
                 pre_nms_topk: Tuple[float, float], post_nms_topk: Tuple[float, float], nms_thresh: float = 0.7,
                 min_box_size: float = 0.0, anchor_boundary_thresh: float = -1.0,
                 loss_weight: Union[float, Dict[str, float]] = 1.0, box_reg_loss_type: str = "smooth_l1",
                 smooth_l1_beta: float = 0.0, mrrp_on: bool = False, mrrp_num_branch: int = 3, mrrp_fast: bool = False):
        super().__init__()
        if mrrp_on:
            raise NotImplementedError("MRRP is off in every WSR config (out of hot-path scope)")
        if box_reg_loss_type != "smooth_l1":
            raise NotImplementedError(f"RPN.BBOX_REG_LOSS_TYPE={box_reg_loss_type}: shipped configs use smooth_l1")
        self.in_features = in_features
        self.rpn_head = head
        self.anchor_generator = anchor_generator
        self.anchor_matcher = anchor_matcher
        self.box2box_transform = box2box_transform
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction
        self.pre_nms_topk = {True: pre_nms_topk[0], False: pre_nms_topk[1]}
        self.post_nms_topk = {True: post_nms_topk[0], False: post_nms_topk[1]}
        self.nms_thresh = nms_thresh
        self.min_box_size = float(min_box_size)
        self.anchor_boundary_thresh = anchor_boundary_thresh
        if isinstance(loss_weight, float):
            loss_weight = {"loss_rpn_cls": loss_weight, "loss_rpn_loc": loss_weight}
        self.loss_weight = loss_weight
        self.box_reg_loss_type = box_reg_loss_type
        self.smooth_l1_beta = smooth_l1_beta
        self.mrrp_on = False

    @classmethod
    def from_config(cls, cfg, input_shape: Dict[str, ShapeSpec]):
        in_features = cfg.MODEL.RPN.IN_FEATURES
        shapes = [input_shape[f] for f in in_features]
        return {
            "in_features": in_features,
            "min_box_size": cfg.MODEL.PROPOSAL_GENERATOR.MIN_SIZE,
            "nms_thresh": cfg.MODEL.RPN.NMS_THRESH,
            "batch_size_per_image": cfg.MODEL.RPN.BATCH_SIZE_PER_IMAGE,
            "positive_fraction": cfg.MODEL.RPN.POSITIVE_FRACTION,
            "loss_weight": {"loss_rpn_cls": cfg.MODEL.RPN.LOSS_WEIGHT,
                            "loss_rpn_loc": cfg.MODEL.RPN.BBOX_REG_LOSS_WEIGHT * cfg.MODEL.RPN.LOSS_WEIGHT},
            "anchor_boundary_thresh": cfg.MODEL.RPN.BOUNDARY_THRESH,
            "box2box_transform": Box2BoxTransform(weights=cfg.MODEL.RPN.BBOX_REG_WEIGHTS),
            "box_reg_loss_type": cfg.MODEL.RPN.BBOX_REG_LOSS_TYPE,
            "smooth_l1_beta": cfg.MODEL.RPN.SMOOTH_L1_BETA,
            "mrrp_on": cfg.MODEL.MRRP.MRRP_ON, "mrrp_num_branch": cfg.MODEL.MRRP.NUM_BRANCH,
            "mrrp_fast": cfg.MODEL.MRRP.TEST_BRANCH_IDX != -1,
            "pre_nms_topk": (cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN, cfg.MODEL.RPN.PRE_NMS_TOPK_TEST),
            "post_nms_topk": (cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN, cfg.MODEL.RPN.POST_NMS_TOPK_TEST),
            "anchor_generator": build_anchor_generator(cfg, shapes),
            "anchor_matcher": Matcher(cfg.MODEL.RPN.IOU_THRESHOLDS, cfg.MODEL.RPN.IOU_LABELS,
                                      allow_low_quality_matches=True),
            "head": build_rpn_head(cfg, shapes),
        }

    @staticmethod
    def _pack_targets(gt_instances):
        """list[Instances] -> (boxes (T,4), start (G) int32, count (G) int32) on the device."""
        boxes = [x.gt_boxes.tensor.to(torch.float32) for x in gt_instances]
        dev = boxes[0].device
        counts = [len(b) for b in boxes]
        starts = [sum(counts[:i]) for i in range(len(counts))]
        cat = torch.cat(boxes) if sum(counts) else torch.zeros((0, 4), device=dev)
        return cat, H.const_tensor(starts, torch.int32, dev), H.const_tensor(counts, torch.int32, dev)

    @torch.no_grad()
    def label_and_sample_anchors(self, anchors: List[Boxes], gt_instances: List[Instances]):
        """The reference's interface (rpn.py:237-293): per image a label vector in {-1, 0, 1} over all anchors and the
        matched ground-truth box of every anchor.  Thin view of the batched kernel path."""
        packed = getattr(gt_instances, "packed", None) or self._pack_targets(list(gt_instances))
        sample = self.label_and_sample_anchors_packed(anchors, packed)
        sampled, best_gt = sample[5], sample[6]
        matched = packed[0][best_gt.clamp(min=0).long()]  # (B, A, 4)
        matched = matched * (best_gt >= 0).unsqueeze(-1)  # no box in the image: zeros, as the reference returns
        return list(sampled.unbind(0)), list(matched.unbind(0))

    def _sample_keys(self, num_images, num_anchors, device):
        """Sort keys of the random sub-sampling: the k smallest keys among the positives / negatives of an image are
        its sample (uniform without replacement, what subsample_labels' randperm does).  Tests replace this by the
        anchor index (= the deterministic first-k rule the golden fixtures were generated with)."""
        return torch.rand((num_images, num_anchors), device=device)

    @torch.no_grad()
    def label_and_sample_anchors_packed(self, anchors: List[Boxes], packed):
        """Batched, sync-free form of label_and_sample_anchors on the device-resident pseudo GT: one labelling kernel
        for the whole batch, then two batched top-k selections.  Returns fixed-size index sets
        (pos_idx (B,P), pos_valid, neg_idx (B,S), neg_valid, matched boxes of the positives (B,P,4), labels (B,A),
        best-box index per anchor (B,A))."""
        gt_boxes, gt_start, gt_count = packed
        anchors_t = Boxes.cat(anchors).tensor
        m = self.anchor_matcher
        assert m.labels == [0, -1, 1] and m.allow_low_quality_matches, "RPN anchor matcher: [0, -1, 1] + low-quality"
        labels, best_gt, _ = H.rpn_label_anchors(anchors_t, gt_boxes, gt_start, gt_count, m.thresholds[1], m.thresholds[2])
        B, A = labels.shape
        S = min(self.batch_size_per_image, A)
        P = min(int(self.batch_size_per_image * self.positive_fraction), A)
        keys = self._sample_keys(B, A, labels.device).to(torch.float32)
        inf = torch.full_like(keys, float("inf"))
        pos_vals, pos_idx = torch.topk(torch.where(labels == 1, keys, inf), P, dim=1, largest=False, sorted=True)
        pos_valid = torch.isfinite(pos_vals)
        num_neg = S - pos_valid.sum(dim=1, keepdim=True)
        neg_vals, neg_idx = torch.topk(torch.where(labels == 0, keys, inf), S, dim=1, largest=False, sorted=True)
        neg_valid = torch.isfinite(neg_vals) & (torch.arange(S, device=keys.device)[None, :] < num_neg)
        matched = gt_boxes[best_gt.gather(1, pos_idx).clamp(min=0).long()]  # (B,P,4)
        sampled = torch.full_like(labels, -1)
        sampled.scatter_(1, neg_idx, torch.where(neg_valid, 0, -1).to(torch.int8))
        sampled.scatter_(1, pos_idx, torch.where(pos_valid, 1, sampled.gather(1, pos_idx).to(torch.int64)).to(torch.int8))
        return pos_idx, pos_valid, neg_idx, neg_valid, matched, sampled, best_gt

    def losses(self, anchors, pred_objectness_logits, gt_labels, pred_anchor_deltas, gt_boxes):
        """The reference's list interface (rpn.py:296-375): labels / matched boxes as label_and_sample_anchors returns
        them.  Re-expressed as the fixed-size index sets of losses_packed (one code path for the arithmetic)."""
        labels = torch.stack(gt_labels)
        matched_all = torch.stack(gt_boxes)
        B, A = labels.shape
        P = min(int(self.batch_size_per_image * self.positive_fraction), A)
        S = min(self.batch_size_per_image, A)
        order = torch.arange(A, device=labels.device, dtype=torch.float32).expand(B, A)
        inf = torch.full_like(order, float("inf"))
        pv, pos_idx = torch.topk(torch.where(labels == 1, order, inf), P, dim=1, largest=False)
        nv, neg_idx = torch.topk(torch.where(labels == 0, order, inf), S, dim=1, largest=False)
        matched = matched_all.gather(1, pos_idx.unsqueeze(-1).expand(-1, -1, 4))
        sample = (pos_idx, torch.isfinite(pv), neg_idx, torch.isfinite(nv), matched, labels)
        return self.losses_packed(anchors, pred_objectness_logits, pred_anchor_deltas, sample)

    def _smooth_l1(self, diff):
        diff = diff.abs()
        if self.smooth_l1_beta < 1e-5:
            return diff
        b = self.smooth_l1_beta
        return torch.where(diff < b, 0.5 * diff * diff / b, diff - 0.5 * b)

    def losses_packed(self, anchors, pred_objectness_logits, pred_anchor_deltas, sample):
        """The same two sums over the fixed-size sample (invalid slots weigh zero): no boolean-mask indexing, so no
        host synchronisation.  A few thousand elements -- torch on the device."""
        pos_idx, pos_valid, neg_idx, neg_valid, matched = sample[:5]
        logits = torch.cat(pred_objectness_logits, dim=1)
        deltas = torch.cat(pred_anchor_deltas, dim=1)
        B = logits.shape[0]
        anchors_t = Boxes.cat(anchors).tensor
        a_pos = anchors_t[pos_idx]  # (B,P,4)
        target = self.box2box_transform.get_deltas(a_pos.reshape(-1, 4), matched.reshape(-1, 4), check=False).view_as(a_pos)
        pv = pos_valid.unsqueeze(-1)
        finite = torch.isfinite(torch.where(pv, target, torch.zeros_like(target))).all()
        pred = deltas.gather(1, pos_idx.unsqueeze(-1).expand(-1, -1, 4))
        loc = (self._smooth_l1(pred - torch.where(pv, target, pred.detach())) * pv).sum()
        loc = torch.where(finite, loc, loc * 0.0)  # rpn.py:337-343: non-finite targets drop the term
        lp, ln = logits.gather(1, pos_idx), logits.gather(1, neg_idx)
        obj = (F.binary_cross_entropy_with_logits(lp, torch.ones_like(lp), reduction="none") * pos_valid).sum() + \
              (F.binary_cross_entropy_with_logits(ln, torch.zeros_like(ln), reduction="none") * neg_valid).sum()
        normalizer = self.batch_size_per_image * B
        losses = {"loss_rpn_cls": obj / normalizer, "loss_rpn_loc": loc / normalizer}
        return scale_losses(losses, self.loss_weight)

    def forward(self, images: ImageList, features: Dict[str, torch.Tensor],
                gt_instances: Optional[List[Instances]] = None):
        features = [features[f] for f in self.in_features]
        anchors = self.anchor_generator(features)
        pred_objectness_logits, pred_anchor_deltas = [], []
        for f in features:
            lo, de = self.rpn_head.forward_nhwc(f)
            pred_objectness_logits.append(lo)
            pred_anchor_deltas.append(de)
        if self.training:
            self.anchors = anchors
            self.pred_objectness_logits = pred_objectness_logits
            self.pred_anchor_deltas = pred_anchor_deltas
        proposals = self.predict_proposals(anchors, pred_objectness_logits, pred_anchor_deltas, images.image_sizes)
        return proposals, {}

    def get_losses(self, gt_instances):
        assert gt_instances is not None, "RPN requires gt_instances in training!"
        packed = getattr(gt_instances, "packed", None)
        if packed is not None:  # pseudo GT still on the device: batched, sync-free path
            sample = self.label_and_sample_anchors_packed(self.anchors, packed)
            self.sampled_labels = sample[5]
            return self.losses_packed(self.anchors, self.pred_objectness_logits, self.pred_anchor_deltas, sample)
        gt_labels, gt_boxes = self.label_and_sample_anchors(self.anchors, gt_instances)
        return self.losses(self.anchors, self.pred_objectness_logits, gt_labels, self.pred_anchor_deltas, gt_boxes)

    @torch.no_grad()
    def predict_proposals(self, anchors, pred_objectness_logits, pred_anchor_deltas, image_sizes):
        return find_top_rpn_proposals(anchors, pred_objectness_logits, pred_anchor_deltas, image_sizes,
                                      self.box2box_transform, self.nms_thresh, self.pre_nms_topk[self.training],
                                      self.post_nms_topk[self.training], self.min_box_size, self.training)


def build_proposal_generator(cfg, input_shape):
    name = cfg.MODEL.PROPOSAL_GENERATOR.NAME
    if name == "PrecomputedProposals":
        return None
    return PROPOSAL_GENERATOR_REGISTRY.get(name)(cfg, input_shape)
