"""ORACLE -- test infrastructure only (never imported by wsovod_amd/).

Functional PyTorch-CPU fp32 restatement of the reference's per-image detection hot path
(SURVEY.md section 8a).  Parameters come in as a state dict with the reference's key names, so
the same tensors can be loaded into the HIP model and into the reference modules (under shims,
tests/golden/make_golden.py) and all three compared.  Every function cites the reference lines
it follows; paths are relative to /root/reference/.

Parity pinning: tests/golden/*.npz hold outputs of the REFERENCE's own code on seeded inputs
(generated here by tests/golden/make_golden.py); tests/test_oracle_golden.py checks this file
against them.  Un-vendored pieces (detectron2 / fvcore / torchvision semantics) are restated from
SURVEY.md Appendix A and cross-checked against torch built-ins where one exists.
"""
import math

import torch
import torch.nn.functional as F

from . import roi_ops

# ----------------------------------------------------------------------------------------
# backbone: wsovod/modeling/backbone/resnet_wsl.py
# ----------------------------------------------------------------------------------------
_BLOCKS = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}


def conv_frozen_bn(x, sd, prefix, stride=1, padding=0, dilation=1, eps=1e-5):
    """detectron2 Conv2d(norm=FrozenBatchNorm2d): F.conv2d then F.batch_norm(training=False)."""
    x = F.conv2d(x, sd[prefix + "weight"], sd.get(prefix + "bias"), stride, padding, dilation)
    if prefix + "norm.weight" in sd:
        x = F.batch_norm(x, sd[prefix + "norm.running_mean"], sd[prefix + "norm.running_var"],
                         sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], training=False, eps=eps)
    return x


def _tail_pool(out, pool_stride):
    # resnet_wsl.py:85-92
    if pool_stride == 1:
        out = F.pad(out, (0, 1, 0, 1))
        return F.max_pool2d(out, kernel_size=2, stride=1, padding=0)
    return F.max_pool2d(out, kernel_size=2, stride=pool_stride, padding=0)


def basic_block(x, sd, prefix, dilation, has_pool, pool_stride):
    """resnet_wsl.py:94-110."""
    out = F.relu(conv_frozen_bn(x, sd, prefix + "conv1.", 1, dilation, dilation))
    out = conv_frozen_bn(out, sd, prefix + "conv2.", 1, dilation, dilation)
    shortcut = conv_frozen_bn(x, sd, prefix + "shortcut.") if prefix + "shortcut.weight" in sd else x
    out = F.relu(out + shortcut)
    return _tail_pool(out, pool_stride) if has_pool else out


def bottleneck_block(x, sd, prefix, dilation, has_pool, pool_stride):
    """resnet_wsl.py:221-241."""
    out = F.relu(conv_frozen_bn(x, sd, prefix + "conv1."))
    out = F.relu(conv_frozen_bn(out, sd, prefix + "conv2.", 1, dilation, dilation))
    out = conv_frozen_bn(out, sd, prefix + "conv3.")
    shortcut = conv_frozen_bn(x, sd, prefix + "shortcut.") if prefix + "shortcut.weight" in sd else x
    out = F.relu(out + shortcut)
    return _tail_pool(out, pool_stride) if has_pool else out


def backbone_forward(sd, x, depth=18, res5_dilation=2, prefix="backbone."):
    """resnet_wsl.py:410-421 (stem), :497-520 (stages), :674-706 (stride/dilation/pool wiring)."""
    out = {}
    x = F.relu(conv_frozen_bn(x, sd, prefix + "stem.conv1.", 2, 1))
    x = F.relu(conv_frozen_bn(x, sd, prefix + "stem.conv2.", 1, 1))
    x = F.relu(conv_frozen_bn(x, sd, prefix + "stem.conv3.", 1, 1))
    x = F.max_pool2d(x, kernel_size=2, stride=2, padding=0)
    out["stem"] = x
    block = basic_block if depth in (18, 34) else bottleneck_block
    for idx, stage_idx in enumerate(range(2, 6)):
        dilation = res5_dilation if stage_idx in (4, 5) else 1
        first_stride = 2 if idx == 0 or (stage_idx == 3 and res5_dilation == 1) else 1
        has_pool = stage_idx in (2, 3)
        nb = _BLOCKS[depth][idx]
        for b in range(nb):
            last = b == nb - 1
            x = block(x, sd, f"{prefix}res{stage_idx}.{b}.", dilation, has_pool and last, first_stride if last else 1)
        out[f"res{stage_idx}"] = x
    return out


def preprocess_image(images_u8, pixel_mean, pixel_std):
    """meta_arch/rcnn_wsovod.py:321-328 + ImageList.from_tensors(size_divisibility=0)."""
    mean = torch.tensor(pixel_mean).view(-1, 1, 1)
    std = torch.tensor(pixel_std).view(-1, 1, 1)
    imgs = [(x.float() - mean) / std for x in images_u8]
    Hm, Wm = max(i.shape[-2] for i in imgs), max(i.shape[-1] for i in imgs)
    out = imgs[0].new_zeros((len(imgs), 3, Hm, Wm))
    for i, im in enumerate(imgs):
        out[i, :, : im.shape[-2], : im.shape[-1]] = im
    return out


# ----------------------------------------------------------------------------------------
# pooling: wsovod/modeling/poolers.py
# ----------------------------------------------------------------------------------------
def pooler_format(boxes_list):
    """poolers.py:74-108."""
    return torch.cat([torch.cat((torch.full_like(b[:, :1], i), b), dim=1) for i, b in enumerate(boxes_list)], dim=0)


def roi_pooler(feat, boxes_list, pooler_type="ROIPool", output_size=7, scale=0.125, sampling_ratio=0):
    """poolers.py:169-197,277-284 (single level)."""
    rois = pooler_format(boxes_list)
    size = (output_size, output_size)
    if feat.requires_grad:  # a trainable backbone stage (FREEZE_AT < 5): the differentiable fronts
        if pooler_type == "ROIPool":
            return roi_ops.roi_pool(feat, rois, scale, size)
        if pooler_type in ("ROIAlignV2", "ROIAlign"):
            return roi_ops.roi_align(feat, rois, scale, size, sampling_ratio, pooler_type == "ROIAlignV2")
        raise ValueError(f"{pooler_type}: no differentiable oracle")
    if pooler_type == "ROIPool":
        return roi_ops.roi_pool_forward(feat, rois, scale, size)[0]
    if pooler_type == "ROILoopPool":  # (3R, C, ph, pw) = [region | frame | context]
        return roi_ops.roi_loop_pool_forward(feat, rois, scale, size)[0]
    if pooler_type == "ROIAlignV2":
        return roi_ops.roi_align_forward(feat, rois, scale, size, sampling_ratio, True)
    if pooler_type == "ROIAlign":
        return roi_ops.roi_align_forward(feat, rois, scale, size, sampling_ratio, False)
    raise ValueError(pooler_type)


# ----------------------------------------------------------------------------------------
# heads
# ----------------------------------------------------------------------------------------
def neck_forward(sd, pooled, prefix="roi_heads.box_head.", dropout_masks=None):
    """roi_heads/box_head.py:90-93 (Flatten, fc1, ReLU, Dropout, fc2, ReLU, Dropout).
    dropout_masks: None (eval) or two 0/1 keep masks (inverted dropout, p = 0.5)."""
    x = torch.flatten(pooled, 1)
    for k in (1, 2):
        x = F.relu(F.linear(x, sd[f"{prefix}fc{k}.weight"], sd[f"{prefix}fc{k}.bias"]))
        if dropout_masks is not None:
            x = x * dropout_masks[k - 1] * 2.0
    return x


def data_aware_forward(sd, res5, prefix="data_aware_head."):
    """class_heads/data_aware_features_head.py:123-129 -> one row per image."""
    x = F.adaptive_avg_pool2d(res5, 1).flatten(start_dim=1)
    x = F.relu(F.linear(x, sd[prefix + "linear1.weight"], sd[prefix + "linear1.bias"]))
    x = torch.tanh(F.linear(x, sd[prefix + "linear2.weight"], sd[prefix + "linear2.bias"]))
    return torch.matmul(x, sd[prefix + "datasets_feat.weight"])


def mining_forward(sd, x, nums, prefix="roi_heads.object_miner."):
    """roi_heads/fast_rcnn_open_vocabulary.py:333-357."""
    C = F.linear(x, sd[prefix + "cls.weight"], sd[prefix + "cls.bias"])
    D = F.linear(x, sd[prefix + "det.weight"], sd[prefix + "det.bias"])
    K = C.shape[1]
    if K == 1:
        C = torch.cat((C, torch.zeros_like(C)), dim=1)
        D = torch.cat((D, torch.zeros_like(D)), dim=1)
    scores = torch.cat([F.softmax(c, dim=1) * F.softmax(d, dim=0) for c, d in zip(C.split(nums), D.split(nums))], dim=0)
    if K == 1:
        scores, _ = torch.split(scores, 1, dim=1)
    return scores


def mining_forward_contextlocnet(sd, x, fx, cx, nums, prefix="roi_heads.object_miner."):
    """fast_rcnn_open_vocabulary.py:369-390 (forward_contextlocnet) + :345-357: C = cls(x), D = det(Fx) - det(Cx)."""
    C = F.linear(x, sd[prefix + "cls.weight"], sd[prefix + "cls.bias"])
    D = F.linear(fx, sd[prefix + "det.weight"], sd[prefix + "det.bias"]) - \
        F.linear(cx, sd[prefix + "det.weight"], sd[prefix + "det.bias"])
    return torch.cat([F.softmax(c, dim=1) * F.softmax(d, dim=0) for c, d in zip(C.split(nums), D.split(nums))], dim=0)


def predict_probs_img(scores, nums):
    """fast_rcnn_open_vocabulary.py:604-618."""
    s = torch.cat([t.sum(dim=0, keepdim=True) for t in scores.split(nums)], dim=0)
    return torch.clamp(s, min=1e-6, max=1.0 - 1e-6)


def mining_loss(scores, nums, gt_oh, mean_loss=True):
    """fast_rcnn_open_vocabulary.py:392-437."""
    img = predict_probs_img(scores, nums)
    if mean_loss:
        return F.binary_cross_entropy(img, gt_oh.float(), reduction="mean")
    return F.binary_cross_entropy(img, gt_oh.float(), reduction="sum") / gt_oh.size(0)


def ov_classifier_forward(sd, x, prefix, temperature=50.0, norm_weight=True, classifier=None,
                          append_background=True, cls_bias=None):
    """class_heads/open_vocabulary_classifier.py:79-105.  sd[prefix+'class_weight'] is the (D,K) buffer
    already normalised at construction (:59-60)."""
    x = F.relu(F.linear(x, sd[prefix + "projection.0.weight"], sd[prefix + "projection.0.bias"]))
    x = F.relu(F.linear(x, sd[prefix + "projection.2.weight"], sd[prefix + "projection.2.bias"]))
    if classifier is not None:
        class_weight = classifier.permute(1, 0).contiguous()
        class_weight = F.normalize(class_weight, p=2, dim=0) if norm_weight else class_weight
    else:
        class_weight = sd[prefix + "class_weight"]
    if norm_weight:
        x = temperature * F.normalize(x, p=2, dim=1)
    if append_background:
        class_weight = torch.cat([class_weight, class_weight.new_zeros((class_weight.size(0), 1))], dim=1)
    x = torch.mm(x, class_weight)
    if cls_bias is not None:
        x = x + cls_bias
    return x


def box2box_get_deltas(src, tgt, weights=(10.0, 10.0, 5.0, 5.0)):
    """detectron2 Box2BoxTransform.get_deltas."""
    sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
    sx, sy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
    tw, th = tgt[:, 2] - tgt[:, 0], tgt[:, 3] - tgt[:, 1]
    tx, ty = tgt[:, 0] + 0.5 * tw, tgt[:, 1] + 0.5 * th
    wx, wy, ww, wh = weights
    return torch.stack((wx * (tx - sx) / sw, wy * (ty - sy) / sh, ww * torch.log(tw / sw), wh * torch.log(th / sh)),
                       dim=1)


SCALE_CLAMP = math.log(1000.0 / 16)  # detectron2 box_regression._DEFAULT_SCALE_CLAMP


def box2box_apply_deltas(deltas, boxes, weights=(10.0, 10.0, 5.0, 5.0), scale_clamp=SCALE_CLAMP):
    """detectron2 Box2BoxTransform.apply_deltas (un-vendored)."""
    deltas = deltas.float()
    boxes = boxes.to(deltas.dtype)
    widths = boxes[:, 2] - boxes[:, 0]
    heights = boxes[:, 3] - boxes[:, 1]
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx = deltas[:, 0::4] / wx
    dy = deltas[:, 1::4] / wy
    dw = torch.clamp(deltas[:, 2::4] / ww, max=scale_clamp)
    dh = torch.clamp(deltas[:, 3::4] / wh, max=scale_clamp)
    pred_ctr_x = dx * widths[:, None] + ctr_x[:, None]
    pred_ctr_y = dy * heights[:, None] + ctr_y[:, None]
    pred_w = torch.exp(dw) * widths[:, None]
    pred_h = torch.exp(dh) * heights[:, None]
    x1 = pred_ctr_x - 0.5 * pred_w
    y1 = pred_ctr_y - 0.5 * pred_h
    x2 = pred_ctr_x + 0.5 * pred_w
    y2 = pred_ctr_y + 0.5 * pred_h
    return torch.stack((x1, y1, x2, y2), dim=-1).reshape(deltas.shape)


def rpn_decode_clip(anchors, deltas, image_size, weights=(1.0, 1.0, 1.0, 1.0), min_size=0.0):
    """rpn.py:495-515 (_decode_proposals) + proposal_utils.py:101-121 (finite check, Boxes.clip,
    nonempty(threshold=min_size)) for one image.  -> (clipped boxes, keep flags)."""
    boxes = box2box_apply_deltas(deltas, anchors, weights)
    finite = torch.isfinite(boxes).all(dim=1)
    h, w = image_size
    boxes = boxes.clone()
    boxes[:, 0::2] = boxes[:, 0::2].clamp(min=0, max=w)
    boxes[:, 1::2] = boxes[:, 1::2].clamp(min=0, max=h)
    keep = finite & ((boxes[:, 2] - boxes[:, 0]) > min_size) & ((boxes[:, 3] - boxes[:, 1]) > min_size)
    return boxes, keep


def batched_nms(boxes, scores, idxs, iou_threshold):
    """detectron2.layers.batched_nms -> torchvision.ops.batched_nms (un-vendored): per-category greedy NMS,
    result sorted by descending score (oracle/det_ops_ref.c does the suppression)."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    by_score = scores.argsort(descending=True, stable=True)
    order = by_score[idxs[by_score].argsort(stable=True)]
    cats, counts = torch.unique_consecutive(idxs[order], return_counts=True)
    offs = [0] + counts.cumsum(0).tolist()
    keeps = roi_ops.nms_segments(boxes.float()[order], offs, iou_threshold)
    kept = torch.cat([order[offs[g] + k] for g, k in enumerate(keeps)])
    return kept[scores[kept].argsort(descending=True, stable=True)]


def fast_rcnn_inference_single_image(boxes, scores, image_shape, score_thresh, nms_thresh, topk_per_image):
    """fast_rcnn_open_vocabulary.py:149-217 -> (boxes, scores, classes, proposal index) of the detections."""
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
    pred_inds = torch.arange(scores.size(0))[valid]
    boxes, scores = boxes[valid], scores[valid]
    scores = scores[:, :-1]
    nreg = boxes.shape[1] // 4
    boxes = boxes.reshape(-1, 4).clone()
    h, w = image_shape
    boxes[:, 0::2] = boxes[:, 0::2].clamp(min=0, max=w)
    boxes[:, 1::2] = boxes[:, 1::2].clamp(min=0, max=h)
    boxes = boxes.view(-1, nreg, 4)
    filter_mask = scores > score_thresh
    filter_inds = filter_mask.nonzero()
    boxes = boxes[filter_inds[:, 0], 0] if nreg == 1 else boxes[filter_mask]
    scores = scores[filter_mask]
    keep = batched_nms(boxes, scores, filter_inds[:, 1], nms_thresh)
    if topk_per_image >= 0:
        keep = keep[:topk_per_image]
    return boxes[keep], scores[keep], filter_inds[keep, 1], pred_inds[filter_inds[keep, 0]]


# ----------------------------------------------------------------------------------------
# inference: rcnn_wsovod.py:236-319 -> roi_heads.py eval branch -> predict_probs_K / predict_boxes_K
# (fast_rcnn_open_vocabulary.py:987-1058) -> fast_rcnn_inference_single_image -> detector_postprocess.
# Pinned by tests/golden/g14_eval_tail.npz and g15_tta_avg.npz (outputs of the reference's own files).
# ----------------------------------------------------------------------------------------
@torch.no_grad()
def eval_forward(sd, batch, *, depth=18, classifier=None, pooler_type="ROIPool", temperature=50.0,
                 pixel_mean=(102.9801, 115.9465, 122.7717), pixel_std=(1.0, 1.0, 1.0), data_aware=True, refine_K=1,
                 sampling_ratio=0, rpn=None):
    """Per image: (scores (R, K+1) = mean over the refinement heads of softmax(logits), boxes (R, 4) =
    apply_deltas(mean deltas, proposals)) -- the `all_scores` / `all_boxes` the detection tail consumes.
    rpn: None, or a dict of find_top_rpn_proposals keyword arguments (rcnn_wsovod.py:267-283: the RPN's boxes, with
    sigmoid objectness, ahead of the loaded ones); the per-image RPN proposals are then left in rpn["proposals"]."""
    x = preprocess_image([b["image"] for b in batch], pixel_mean, pixel_std)
    res5 = backbone_forward(sd, x, depth)["res5"]
    boxes_list = [b["boxes"] for b in batch]
    obj_list = [b["objectness"] for b in batch]
    if rpn is not None:
        anchors = anchor_grid(res5.shape[-2], res5.shape[-1])
        lo, de = rpn_head_forward(sd, res5)
        kw = {k: v for k, v in rpn.items() if k != "proposals"}
        props = find_top_rpn_proposals(anchors, lo, de, [tuple(b["image"].shape[-2:]) for b in batch], **kw)
        rpn["proposals"] = props
        boxes_list = [torch.cat([pb, b]) for (pb, _), b in zip(props, boxes_list)]
        obj_list = [torch.cat([torch.sigmoid(ps), o]) for (_, ps), o in zip(props, obj_list)]
    nums = [len(b) for b in boxes_list]
    pooled = roi_pooler(res5, boxes_list, pooler_type, 7, 0.125, sampling_ratio)
    pooled = pooled * torch.cat([o + 1 for o in obj_list]).view(-1, 1, 1, 1)
    feat = neck_forward(sd, pooled)
    if data_aware:
        daf = data_aware_forward(sd, res5)
        feat = feat + torch.cat([daf[i].repeat(n, 1) for i, n in enumerate(nums)])
    probs, deltas = 0, 0
    for k in range(refine_K):
        prefix = f"roi_heads.box_refinery_{k}."
        logits = ov_classifier_forward(sd, feat, prefix + "cls.", temperature, classifier=classifier)
        probs = probs + torch.softmax(logits, dim=-1)
        deltas = deltas + F.linear(feat, sd[prefix + "bbox_pred.weight"], sd[prefix + "bbox_pred.bias"])
    boxes = box2box_apply_deltas(deltas / refine_K, torch.cat(boxes_list))
    return list(zip((probs / refine_K).split(nums), boxes.split(nums)))


def detector_postprocess(boxes, image_size, output_height, output_width):
    """postprocessing.py:8-82 for box-only results: scale to the requested size, clip, keep non-empty boxes.
    Returns (boxes, keep mask)."""
    sx, sy = output_width / image_size[1], output_height / image_size[0]
    b = boxes.clone()
    b[:, 0::2] *= sx
    b[:, 1::2] *= sy
    b[:, 0::2] = b[:, 0::2].clamp(min=0, max=output_width)
    b[:, 1::2] = b[:, 1::2].clamp(min=0, max=output_height)
    keep = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
    return b[keep], keep


def tta_avg_merge(view_boxes, view_scores, inverse_apply_box, shape_hw, score_thresh, nms_thresh, topk_per_image):
    """test_time_augmentation_avg.py:279-318: every view's per-proposal boxes go back to the original frame through
    the inverse of that view's transforms (`inverse_apply_box[i]`: (n,4) numpy -> (n,4) numpy), boxes and class
    scores are averaged over the views, then ONE detection tail pass."""
    back = [torch.from_numpy(inv(b.numpy())).to(b.dtype) for b, inv in zip(view_boxes, inverse_apply_box)]
    boxes = torch.stack(back).mean(dim=0)
    scores = torch.stack(list(view_scores)).mean(dim=0)
    return (boxes, scores) + tuple(fast_rcnn_inference_single_image(boxes, scores, shape_hw, score_thresh, nms_thresh,
                                                                     topk_per_image))


def tta_union_view_proposals(boxes, objectness, apply_box, image_shape, proposal_topk, min_box_size=0):
    """test_time_augmentation_union.py:25-63 (`transform_proposals` of the UNION mapper, proposal_topk > 0): the loaded
    boxes follow the view's transforms (`apply_box`: (n,4) numpy -> numpy), are clipped to the view, boxes with a side
    <= min_box_size are dropped, the first proposal_topk stay."""
    b = torch.from_numpy(apply_box(boxes.numpy())).to(boxes.dtype)
    h, w = image_shape
    b[:, 0::2] = b[:, 0::2].clamp(min=0, max=w)
    b[:, 1::2] = b[:, 1::2].clamp(min=0, max=h)
    keep = ((b[:, 2] - b[:, 0]) > min_box_size) & ((b[:, 3] - b[:, 1]) > min_box_size)
    return b[keep][:proposal_topk], objectness[keep][:proposal_topk]


def tta_union_merge(view_dets, inverse_apply_box, shape_hw, num_classes, nms_thresh, topk_per_image):
    """test_time_augmentation_union.py:273-309: every view's DETECTIONS (boxes, scores, classes) go back to the original
    frame through the inverse of that view's transforms and are pooled; a (n, K+1) score matrix holds each detection's
    score in its class column; one more detection tail pass at score threshold 1e-8.
    -> (pooled boxes, boxes, scores, classes)."""
    back = [torch.from_numpy(inv(b.numpy())).to(b.dtype) for (b, _, _), inv in zip(view_dets, inverse_apply_box)]
    boxes = torch.cat(back)
    scores = torch.cat([s for _, s, _ in view_dets])
    classes = torch.cat([c for _, _, c in view_dets])
    sc2d = torch.zeros(len(boxes), num_classes + 1)
    sc2d[torch.arange(len(boxes)), classes] = scores
    rb, rs, rc, _ = fast_rcnn_inference_single_image(boxes, sc2d, shape_hw, 1e-8, nms_thresh, topk_per_image)
    return boxes, rb, rs, rc


def refinement_losses(logits, deltas, gt_classes, gt_weights, proposal_boxes, gt_boxes, num_classes,
                      bbox_weights=(10.0, 10.0, 5.0, 5.0), beta=0.0, cross_entropy_weighted=True,
                      box_loss_type="smooth_l1_weighted"):
    """fast_rcnn_open_vocabulary.py:754-892 (weighted CE :813-820; weighted smooth-L1 :864-878)."""
    w = gt_weights.clone()
    w[gt_classes == -1] = 0.0
    valid = torch.zeros_like(w)
    valid[w > 1e-12] = 1.0
    if cross_entropy_weighted:
        ce = F.cross_entropy(logits, gt_classes, reduction="none", ignore_index=-1)
        loss_cls = (ce * w).sum() / valid.sum()
    else:
        loss_cls = F.cross_entropy(logits, gt_classes, reduction="mean", ignore_index=-1)
    fg = ((gt_classes >= 0) & (gt_classes < num_classes)).nonzero().squeeze(1)
    tgt = box2box_get_deltas(proposal_boxes[fg], gt_boxes[fg], bbox_weights)
    diff = (deltas[fg] - tgt).abs()
    l = diff if beta < 1e-5 else torch.where(diff < beta, 0.5 * diff * diff / beta, diff - 0.5 * beta)
    if box_loss_type == "smooth_l1_weighted":
        if torch.isnan(tgt).any():
            loss_box = torch.zeros(())
        else:
            loss_box = (l * w[fg, None]).sum() / max(gt_classes.numel(), 1.0)
    else:
        loss_box = l.sum() / max(gt_classes.numel(), 1.0)
    return loss_cls, loss_box


# ----------------------------------------------------------------------------------------
# pseudo-GT mining + labelling (no grad): roi_heads/roi_heads.py
# ----------------------------------------------------------------------------------------
def pairwise_iou(b1, b2):
    """detectron2.structures.pairwise_iou."""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    wh = (torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])).clamp(min=0)
    inter = wh.prod(dim=2)
    return torch.where(inter > 0, inter / (a1[:, None] + a2 - inter), torch.zeros(1))


@torch.no_grad()
def get_pgt_top_k(prev_pred_boxes, prev_pred_scores, gt_classes_img_int, pred_class_img_logits, num_classes):
    """roi_heads.py:1043-1343 with top_k=1, thres=0, need_weight, sam=None.
    prev_pred_boxes: list of (R,4); prev_pred_scores: list of (R,>=K).  Returns per image a dict with
    gt_boxes (G,4), gt_classes (G), gt_scores (G), gt_weights (G)."""
    out = []
    for i, (boxes, scores, gt_int) in enumerate(zip(prev_pred_boxes, prev_pred_scores, gt_classes_img_int)):
        R = boxes.size(0)
        b = boxes.unsqueeze(1).expand(R, num_classes, 4)
        s = torch.index_select(scores, 1, gt_int)
        b = torch.index_select(b, 1, gt_int)
        keep = (b[:, :, 2] - b[:, :, 0]) * (b[:, :, 3] - b[:, :, 1]) > 20  # :1090-1096
        b = b.masked_select(keep.unsqueeze(2).expand(-1, gt_int.numel(), 4)).view(-1, gt_int.numel(), 4)
        s = s.masked_select(keep).view(-1, gt_int.numel())
        top_k = min(s.size(0), 1)
        pgt_scores, pgt_idx = torch.topk(s, top_k, dim=0)
        pgt_boxes = torch.gather(b, 0, pgt_idx.unsqueeze(2).expand(top_k, gt_int.numel(), 4))
        pgt_classes = gt_int.unsqueeze(0).expand(top_k, gt_int.numel())
        pgt_weights = torch.index_select(pred_class_img_logits[i:i + 1], 1, gt_int).expand(top_k, gt_int.numel())
        pgt_scores, pgt_boxes = pgt_scores.reshape(-1), pgt_boxes.reshape(-1, 4)
        pgt_classes, pgt_weights = pgt_classes.reshape(-1), pgt_weights.reshape(-1)
        if pgt_weights.numel() == 0:  # :1181-1207 fallbacks
            pgt_weights = torch.tensor([1], dtype=pgt_weights.dtype)
        if pgt_scores.numel() == 0:
            pgt_scores = torch.tensor([1], dtype=pgt_scores.dtype)
        if pgt_boxes.numel() == 0:
            pgt_boxes = torch.tensor([[-10000, -10000, 10000, 10000]], dtype=pgt_boxes.dtype)
        if pgt_classes.numel() == 0:
            pgt_classes = torch.tensor([0], dtype=pgt_classes.dtype)
        out.append(dict(gt_boxes=pgt_boxes, gt_classes=pgt_classes, gt_scores=pgt_scores, gt_weights=pgt_weights))
    return out


def subsample_labels_keyed(labels, num_samples, positive_fraction, bg_label, keys):
    """detectron2 subsample_labels (un-vendored; call site roi_heads.py:1597-1602) with the
    random permutations expressed as sort keys: `randperm(n)[:k]` of a group = its k rows with the smallest keys
    (ties by row index).  Returns (pos_idx, neg_idx)."""
    positive = ((labels != -1) & (labels != bg_label)).nonzero().flatten()
    negative = (labels == bg_label).nonzero().flatten()
    num_pos = min(positive.numel(), int(num_samples * positive_fraction))
    num_neg = min(negative.numel(), num_samples - num_pos)
    pos = positive[torch.argsort(keys[positive], stable=True)[:num_pos]]
    neg = negative[torch.argsort(keys[negative], stable=True)[:num_neg]]
    return pos, neg


@torch.no_grad()
def label_and_sample_proposals_wsl(proposal_boxes_list, targets, num_classes, iou_thr=0.5, batch_size_per_image=4096,
                                   positive_fraction=1.0, keys=None):
    """roi_heads.py:1722-1825 + _sample_proposals_wsl :1566-1610 with Matcher([thr],[0,1]).  subsample_labels keeps
    every proposal when R <= batch_size_per_image and positive_fraction == 1; otherwise `keys` (one float tensor per
    image) stand for the random permutations and rows outside the sample are labelled -1 (:1604-1607)."""
    res = []
    for i, (pb, t) in enumerate(zip(proposal_boxes_list, targets)):
        iou = pairwise_iou(t["gt_boxes"], pb)  # (G,R)
        matched_vals, matched_idxs = iou.max(dim=0)
        labels = (matched_vals >= iou_thr).to(torch.int8)
        gt_classes = t["gt_classes"][matched_idxs].clone()
        gt_classes[labels == 0] = num_classes
        if keys is not None:
            pos, neg = subsample_labels_keyed(gt_classes, batch_size_per_image, positive_fraction, num_classes, keys[i])
            sampled = torch.full_like(gt_classes, -1)
            idx = torch.cat([pos, neg])
            sampled[idx] = gt_classes[idx]
            gt_classes = sampled
        else:
            assert len(pb) <= batch_size_per_image and positive_fraction >= 1.0, "sub-sampling needs keys"
        res.append(dict(gt_classes=gt_classes, gt_boxes=t["gt_boxes"][matched_idxs],
                        gt_scores=t["gt_scores"][matched_idxs], gt_weights=t["gt_weights"][matched_idxs],
                        matched_idxs=matched_idxs))
    return res


def get_image_level_gt(gt_classes_list, num_classes):
    """roi_heads.py:158-174."""
    ints = [torch.unique(g, sorted=True).to(torch.int64) for g in gt_classes_list]
    oh = torch.cat([torch.zeros((1, num_classes)).scatter_(1, g.unsqueeze(0), 1) for g in ints], dim=0)
    return ints, oh


# ----------------------------------------------------------------------------------------
# RPN branch: proposal_generator/rpn.py, proposal_utils.py, detectron2 pieces restated
# ----------------------------------------------------------------------------------------
def anchor_grid(h, w, stride=8, sizes=(32, 64, 128, 256, 512, 768), aspect_ratios=(1.0, 2.0, 0.5), offset=0.0):
    """detectron2 DefaultAnchorGenerator (un-vendored): (H*W*A, 4), A = sizes x ratios."""
    cell = []
    for size in sizes:
        area = size ** 2.0
        for ar in aspect_ratios:
            ww = math.sqrt(area / ar)
            hh = ar * ww
            cell.append([-ww / 2.0, -hh / 2.0, ww / 2.0, hh / 2.0])
    cell = torch.tensor(cell)
    sx = torch.arange(offset * stride, w * stride, step=stride, dtype=torch.float32)
    sy = torch.arange(offset * stride, h * stride, step=stride, dtype=torch.float32)
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    xx, yy = xx.reshape(-1), yy.reshape(-1)
    shifts = torch.stack((xx, yy, xx, yy), dim=1)
    return (shifts.view(-1, 1, 4) + cell.view(1, -1, 4)).reshape(-1, 4)


def rpn_head_forward(sd, feat, prefix="proposal_generator.rpn_head."):
    """detectron2 StandardRPNHead + the permutes of rpn.py:403-417 -> ((N, H*W*A), (N, H*W*A, 4))."""
    t = F.relu(F.conv2d(feat, sd[prefix + "conv.weight"], sd[prefix + "conv.bias"], padding=1))
    lo = F.conv2d(t, sd[prefix + "objectness_logits.weight"], sd[prefix + "objectness_logits.bias"])
    de = F.conv2d(t, sd[prefix + "anchor_deltas.weight"], sd[prefix + "anchor_deltas.bias"])
    N = feat.shape[0]
    lo = lo.permute(0, 2, 3, 1).flatten(1)
    de = de.view(N, -1, 4, de.shape[-2], de.shape[-1]).permute(0, 3, 4, 1, 2).flatten(1, -2)
    return lo, de


@torch.no_grad()
def find_top_rpn_proposals(anchors, logits, deltas, image_sizes, nms_thresh=0.7, pre_nms_topk=2048,
                           post_nms_topk=1024, min_box_size=40.0, weights=(1.0, 1.0, 1.0, 1.0)):
    """rpn.py:495-515 + proposal_utils.py:26-144, single level.  -> per image (boxes, objectness logits)."""
    out = []
    k = min(logits.shape[1], pre_nms_topk)
    for n, image_size in enumerate(image_sizes):
        sc, idx = logits[n].sort(descending=True)
        sc, idx = sc[:k], idx[:k]
        boxes, keep = rpn_decode_clip(anchors[idx], deltas[n][idx], image_size, weights, min_box_size)
        keep = keep & torch.isfinite(sc)
        boxes, sc = boxes[keep], sc[keep]
        kept = roi_ops.nms_segments(boxes, [0, len(boxes)], nms_thresh, post_nms_topk)[0]
        out.append((boxes[kept], sc[kept]))
    return out


def rpn_match_anchors(anchors, gt_boxes, thresholds=(0.2, 0.6), labels=(0, -1, 1)):
    """detectron2 Matcher(allow_low_quality_matches=True) on pairwise_iou(gt, anchors) (rpn.py:268-269)."""
    iou = pairwise_iou(gt_boxes, anchors)
    vals, matches = iou.max(dim=0)
    lab = torch.ones_like(matches, dtype=torch.int8)
    bounds = [-float("inf")] + list(thresholds) + [float("inf")]
    for l, lo, hi in zip(labels, bounds[:-1], bounds[1:]):
        lab[(vals >= lo) & (vals < hi)] = l
    best = iou.max(dim=1).values
    lab[(iou == best[:, None]).nonzero()[:, 1]] = 1
    return matches, lab


def rpn_losses(anchors, logits, deltas, targets, subsample, batch_size_per_image=512, positive_fraction=0.5,
               weights=(1.0, 1.0, 1.0, 1.0), thresholds=(0.2, 0.6)):
    """rpn.py:237-375 (label_and_sample_anchors + losses, smooth_l1 beta 0).  `subsample(labels, n, frac, bg)` is
    detectron2's subsample_labels (random there; the tests inject a deterministic one on both sides)."""
    gt_labels, gt_deltas = [], []
    for t in targets:
        matches, lab = rpn_match_anchors(anchors, t["gt_boxes"], thresholds)
        pos_idx, neg_idx = subsample(lab, batch_size_per_image, positive_fraction, 0)
        lab = torch.full_like(lab, -1)
        lab[pos_idx] = 1
        lab[neg_idx] = 0
        gt_labels.append(lab)
        gt_deltas.append(box2box_get_deltas(anchors, t["gt_boxes"][matches], weights))
    gt_labels, gt_deltas = torch.stack(gt_labels), torch.stack(gt_deltas)
    pos = gt_labels == 1
    loc = (deltas[pos] - gt_deltas[pos]).abs().sum()
    valid = gt_labels >= 0
    obj = F.binary_cross_entropy_with_logits(logits[valid], gt_labels[valid].float(), reduction="sum")
    norm = batch_size_per_image * len(targets)
    return {"loss_rpn_cls": obj / norm, "loss_rpn_loc": loc / norm}, gt_labels


# ----------------------------------------------------------------------------------------
# the whole training step (rcnn_wsovod.py:137-234 -> roi_heads.py:648-907), proposals-only mode
# ----------------------------------------------------------------------------------------
def train_forward(sd, batch, *, depth=18, num_classes=20, pooler_type="ROIPool", temperature=50.0,
                  pixel_mean=(102.9801, 115.9465, 122.7717), pixel_std=(1.0, 1.0, 1.0), data_aware=True,
                  mean_loss=True, sampling_ratio=0, dropout_masks=None, refine_prefix="roi_heads.box_refinery_0.",
                  miner_prefix="roi_heads.object_miner.", classifier=None, rpn=None, sampling=None, backbone_grad=False):
    """batch: list of dicts {image uint8 (3,H,W), boxes (R,4), objectness (R), gt_classes (G)}.
    Returns (losses dict, intermediates dict).  REFINE_NUM=1, REFINE_REG=[True], SAMPLING_ON.
    sampling: None (every proposal kept: R <= 4096, fraction 1) or dict(batch_size_per_image, positive_fraction,
    keys = callable n -> (n,) float sort keys standing for subsample_labels' random permutation)."""
    inter = {}
    x = preprocess_image([b["image"] for b in batch], pixel_mean, pixel_std)
    feats = backbone_forward(sd, x, depth)
    res5 = feats["res5"]
    inter["res5"] = res5
    boxes_list = [b["boxes"] for b in batch]
    obj_list = [b["objectness"] for b in batch]
    if rpn is not None:
        # rcnn_wsovod.py:177-197: RPN boxes (sigmoid objectness ramped by iter / MAX_ITER) ahead of the loaded ones.
        # rpn = dict(cur_iter, max_iter, subsample[, proposals = precomputed (boxes, logits) per image])
        anchors = anchor_grid(res5.shape[-2], res5.shape[-1])
        rpn_logits, rpn_deltas = rpn_head_forward(sd, res5.detach())
        inter["rpn_logits"], inter["rpn_deltas"] = rpn_logits, rpn_deltas
        sizes = [tuple(b["image"].shape[-2:]) for b in batch]
        props = rpn.get("proposals") or find_top_rpn_proposals(anchors, rpn_logits.detach(), rpn_deltas.detach(), sizes)
        inter["rpn_proposals"] = props
        ramp = rpn["cur_iter"] / rpn["max_iter"]
        boxes_list = [torch.cat([pb, b]) for (pb, _), b in zip(props, boxes_list)]
        obj_list = [torch.cat([torch.sigmoid(ps) * ramp, o]) for (_, ps), o in zip(props, obj_list)]
    nums = [len(b) for b in boxes_list]
    # backbone_grad: a backbone stage is trainable (MODEL.BACKBONE.FREEZE_AT < 5, resnet_wsl.py:530-552): the gradient
    # runs through the pooling into res5 (the shipped configs freeze all five stages: the detach is then a no-op)
    pooled = roi_pooler(res5 if backbone_grad else res5.detach(), boxes_list, pooler_type, 7, 0.125, sampling_ratio)
    objectness = torch.cat([o + 1 for o in obj_list], dim=0)
    loop = pooler_type == "ROILoopPool"
    if loop:
        objectness = objectness.repeat(3)
    pooled = pooled * objectness.view(-1, 1, 1, 1)  # roi_heads.py:733-739
    inter["pooled"] = pooled
    feat = neck_forward(sd, pooled, dropout_masks=dropout_masks)
    if loop:
        feat, feat_f, feat_c = torch.chunk(feat, 3, dim=0)  # roi_heads.py:748-760
    if data_aware:
        daf = data_aware_forward(sd, res5)
        inter["daf"] = daf
        rep = torch.cat([daf[i].repeat(n, 1) for i, n in enumerate(nums)])
        feat = feat + rep  # roi_heads.py:762-763
        if loop:
            feat_f, feat_c = feat_f + rep, feat_c + rep
    inter["box_features"] = feat
    if loop:
        scores = mining_forward_contextlocnet(sd, feat, feat_f, feat_c, nums, prefix=miner_prefix)
    else:
        scores = mining_forward(sd, feat, nums, prefix=miner_prefix)
    inter["mining_scores"] = scores
    gt_int, gt_oh = get_image_level_gt([b["gt_classes"] for b in batch], num_classes)
    losses = {"loss_cls_object_mining": mining_loss(scores, nums, gt_oh, mean_loss)}
    img_logits = predict_probs_img(scores, nums).detach()
    inter["pred_class_img_logits"] = img_logits
    targets = get_pgt_top_k(boxes_list, list(scores.detach().split(nums)), gt_int, img_logits, num_classes)
    if sampling is None:
        labelled = label_and_sample_proposals_wsl(boxes_list, targets, num_classes)
    else:
        labelled = label_and_sample_proposals_wsl(boxes_list, targets, num_classes,
                                                  batch_size_per_image=sampling["batch_size_per_image"],
                                                  positive_fraction=sampling["positive_fraction"],
                                                  keys=[sampling["keys"](n) for n in nums])
    inter["targets"], inter["labelled"] = targets, labelled
    # mixed-dataset mode hands the dataset's raw text embeddings in per call (rcnn_wsovod_mixed_datasets.py:237)
    logits = ov_classifier_forward(sd, feat, refine_prefix + "cls.", temperature, classifier=classifier)
    deltas = F.linear(feat, sd[refine_prefix + "bbox_pred.weight"], sd[refine_prefix + "bbox_pred.bias"])
    inter["refine_logits"], inter["refine_deltas"] = logits, deltas
    lc, lb = refinement_losses(logits, deltas, torch.cat([l["gt_classes"] for l in labelled]),
                               torch.cat([l["gt_weights"] for l in labelled]), torch.cat(boxes_list),
                               torch.cat([l["gt_boxes"] for l in labelled]), num_classes)
    losses["loss_cls_r0"], losses["loss_box_reg_r0"] = lc, lb
    if rpn is not None:
        # roi_heads.py:862-881: pseudo GT for the RPN from the refinement head's own predictions
        probs = torch.softmax(logits.detach(), dim=-1)
        pred_boxes = box2box_apply_deltas(deltas.detach(), torch.cat(boxes_list))
        rpn_targets = get_pgt_top_k(list(pred_boxes.split(nums)), list(probs.split(nums)), gt_int, img_logits,
                                    num_classes)
        inter["rpn_targets"] = rpn_targets
        rl, rpn_labels = rpn_losses(anchors, rpn_logits, rpn_deltas, rpn_targets, rpn["subsample"])
        inter["rpn_labels"] = rpn_labels
        losses.update(rl)
    return losses, inter


def batch_from_inputs(batched_inputs):
    """DatasetMapper-format dicts (with Instances) -> the plain-tensor batch of train_forward."""
    return [dict(image=x["image"], boxes=x["proposals"].proposal_boxes.tensor.float(),
                 objectness=x["proposals"].objectness_logits.float(), gt_classes=x["instances"].gt_classes)
            for x in batched_inputs]


def sgd_step(params, grads, bufs, lr, momentum, weight_decay):
    """torch.optim.SGD (dampening 0) as configured by engine/defaults.py:274-318."""
    for k in params:
        g = grads[k] + weight_decay * params[k]
        bufs[k] = momentum * bufs[k] + g if k in bufs else g.clone()
        params[k] = params[k] - lr * bufs[k]
