"""CPU: pin the oracle (oracle/wsovod_ref.py, oracle/roi_ops_ref.c) against the golden vectors that
tests/golden/make_golden.py produced by running the REFERENCE's own code."""
import os

import numpy as np
import pytest
import torch

from oracle import roi_ops
from oracle import wsovod_ref as R
from tests.golden import gen

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    d = np.load(os.path.join(G, name + ".npz"), allow_pickle=False)
    return {k: torch.from_numpy(d[k]) if d[k].dtype.kind in "fiub" else d[k] for k in d.files}


def seeded_sd(seed=1):
    d = np.load(os.path.join(G, "shapes_r18_k20.npz"))
    shapes = {str(k): eval(str(s)) for k, s in zip(d["keys"], d["shapes"])}
    return gen.seeded_state(shapes, seed)


def test_roi_pool_c_oracle_matches_reference_op():
    g = load("g2_roi_pool")
    out, arg = roi_ops.roi_pool_forward(g["feat"], g["rois"], 0.125, (7, 7))
    assert torch.equal(arg, g["argmax"]) and torch.equal(out, g["out"])  # bit-exact
    gi = roi_ops.roi_pool_backward(g["grad_out"], g["rois"], arg, tuple(g["feat"].shape))
    assert torch.equal(gi, g["grad_in"])


@pytest.mark.skipif(not roi_ops.ref_available(), reason="oracle/_ref not built and /root/reference absent")
def test_roi_pool_c_oracle_matches_compiled_reference_random():
    from tests.util import random_rois

    feat = torch.randn(3, 5, 40, 50)
    rois = random_rois(128, 3, 320, 400, seed=77)
    for size in [(7, 7), (3, 5), (1, 1)]:
        a = roi_ops.roi_pool_forward(feat, rois, 0.125, size)
        b = roi_ops.ref_roi_pool_forward(feat, rois, 0.125, size)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_backbone_matches_reference():
    g = load("g1_backbone_small")
    res5 = R.backbone_forward(seeded_sd(), g["x"], depth=18)["res5"]
    torch.testing.assert_close(res5, g["res5"], rtol=1e-4, atol=1e-5)


def test_ov_classifier_matches_reference():
    g = load("g5_ov_classifier")
    sd = seeded_sd()
    p = "roi_heads.box_refinery_0.cls."
    torch.testing.assert_close(R.ov_classifier_forward(sd, g["x"], p), g["logits_default"], rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(R.ov_classifier_forward(sd, g["x"], p, append_background=False), g["logits_nobg"],
                               rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(R.ov_classifier_forward(sd, g["x"], p, classifier=g["classifier"]),
                               g["logits_classifier"], rtol=1e-5, atol=1e-4)
    assert torch.all(g["logits_default"][:, -1] == 0)  # background logit is identically zero


def test_data_aware_head_matches_reference():
    g = load("g9_data_aware")
    per_image = R.data_aware_forward(seeded_sd(), g["res5"])
    rep = torch.cat([per_image[i].repeat(int(n), 1) for i, n in enumerate(g["nums"])])
    torch.testing.assert_close(rep, g["daf"], rtol=1e-5, atol=1e-6)


def test_full_training_step_matches_reference():
    """Losses, MIL scores, refinement logits (<=1e-5), pseudo-GT / labels (exact) and parameter
    gradients of one whole step at the plumbing scale (4 ragged images, R=64/57, K=20)."""
    g = load("g8_train_step_r18_k20")
    sd = seeded_sd()
    train_keys = [k[len("gradnorm/"):] for k in g if k.startswith("gradnorm/")]
    for k in train_keys:
        sd[k].requires_grad_(True)
    batch = gen.seeded_batch(4, 64, 20, 320, 416, seed=2)
    losses, inter = R.train_forward(sd, batch, depth=18, num_classes=20, pixel_std=gen.PIXEL_STD)
    for k in ("loss_cls_object_mining", "loss_cls_r0", "loss_box_reg_r0"):
        torch.testing.assert_close(losses[k].detach(), g["loss/" + k], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(inter["mining_scores"].detach(), g["mining_scores"], rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(inter["refine_logits"].detach(), g["refine_logits"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(inter["refine_deltas"].detach(), g["refine_deltas"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(inter["pred_class_img_logits"], g["pred_class_img_logits"], rtol=1e-5, atol=1e-8)
    # indices / labels: exact
    lab = inter["labelled"]
    assert torch.equal(torch.cat([l["gt_classes"] for l in lab]), g["label/gt_classes"])
    assert torch.equal(torch.cat([l["gt_boxes"] for l in lab]), g["label/gt_boxes"])
    torch.testing.assert_close(torch.cat([l["gt_weights"] for l in lab]), g["label/gt_weights"], rtol=1e-5, atol=1e-8)
    tg = inter["targets"]
    assert [len(t["gt_classes"]) for t in tg] == g["pgt/num"].tolist()
    assert torch.equal(torch.cat([t["gt_boxes"] for t in tg]), g["pgt/gt_boxes"])
    assert torch.equal(torch.cat([t["gt_classes"] for t in tg]), g["pgt/gt_classes"])
    total = sum(losses.values())
    grads = torch.autograd.grad(total, [sd[k] for k in train_keys])
    for k, gr in zip(train_keys, grads):
        torch.testing.assert_close(gr.norm(), g["gradnorm/" + k], rtol=2e-4, atol=1e-9)
        torch.testing.assert_close(gen.strided_sample(gr, 2048), g["gradsample/" + k], rtol=2e-3, atol=1e-7)


def test_mixed_datasets_step_matches_reference():
    """G10: the reference's mixed-dataset meta-arch/ROI heads; dataset 2 (K=80, own miner) and dataset 0
    (K=20, the shared voc miner), per-call text embeddings."""
    from wsovod_amd.data import make_class_embeddings

    g = load("g10_mixed_datasets_step")
    d = np.load(os.path.join(G, "shapes_mixed_r18.npz"))
    shapes = {str(k): eval(str(s)) for k, s in zip(d["keys"], d["shapes"])}
    sd = gen.mixed_seeded_state(shapes, 17)
    Ks = (20, 20, 80)
    for source_id in (2, 0):
        K, p = Ks[source_id], f"s{source_id}/"
        batch = gen.seeded_batch(2, 40, K, 256, 352, seed=19 + source_id)
        sdc = {k: v.clone() for k, v in sd.items()}
        losses, inter = R.train_forward(sdc, batch, depth=18, num_classes=K, pixel_std=gen.PIXEL_STD,
                                        miner_prefix=f"roi_heads.object_miners.{source_id}.",
                                        classifier=make_class_embeddings(K, 512, seed=100 + K))
        assert inter["refine_logits"].shape[1] == K + 1
        for k in ("loss_cls_object_mining", "loss_cls_r0", "loss_box_reg_r0"):
            torch.testing.assert_close(losses[k].detach(), g[p + "loss/" + k], rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(inter["mining_scores"].detach(), g[p + "mining_scores"], rtol=1e-4, atol=1e-8)
        torch.testing.assert_close(inter["refine_logits"].detach(), g[p + "refine_logits"], rtol=1e-4, atol=1e-4)
        lab = inter["labelled"]
        assert torch.equal(torch.cat([l["gt_classes"] for l in lab]), g[p + "label/gt_classes"])
        assert torch.equal(torch.cat([l["gt_boxes"] for l in lab]), g[p + "label/gt_boxes"])
        # the reference leaves the other family's miner without a gradient (find_unused_parameters)
        other = 0 if source_id == 2 else 2
        assert float(g[p + f"gradnorm/roi_heads.object_miners.{other}.cls.weight"]) == -1.0
        assert float(g[p + f"gradnorm/roi_heads.object_miners.{source_id}.cls.weight"]) > 0


def test_rpn_train_step_matches_reference():
    """G12: the reference's RPN branch (WSOVODRPN_V2, find_top_rpn_proposals, meta-arch + ROI-heads glue) on one
    training step: proposals, anchor labels and pseudo-GT exact, losses and every gradient norm."""
    g = load("g12_rpn_train_step")
    d = np.load(os.path.join(G, "shapes_rpn_r18.npz"))
    shapes = {str(k): eval(str(s)) for k, s in zip(d["keys"], d["shapes"])}
    sd = gen.seeded_state(shapes, 41)
    keys = [k[len("gradnorm/"):] for k in g if k.startswith("gradnorm/")]
    for k in keys:
        sd[k].requires_grad_(True)
    batch = gen.seeded_batch(2, 40, 20, 256, 352, seed=43)
    losses, inter = R.train_forward(sd, batch, depth=18, num_classes=20, pixel_std=gen.PIXEL_STD,
                                    rpn=dict(cur_iter=1000, max_iter=4000, subsample=gen.first_k_subsample))
    assert set(losses) == {"loss_cls_object_mining", "loss_cls_r0", "loss_box_reg_r0", "loss_rpn_cls", "loss_rpn_loc"}
    for k, v in losses.items():
        torch.testing.assert_close(v.detach(), g["loss/" + k], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(inter["rpn_logits"].detach(), g["rpn_logits"], rtol=1e-4, atol=1e-5)
    for i, (b, s) in enumerate(inter["rpn_proposals"]):
        assert torch.equal(b, g[f"prop{i}/boxes"]) and torch.equal(s, g[f"prop{i}/logits"])
        assert 0 < len(b) <= 1024
    assert torch.equal(inter["rpn_labels"], g["anchor_labels"])
    for i, t in enumerate(inter["rpn_targets"]):
        assert torch.equal(t["gt_boxes"], g[f"target{i}/gt_boxes"])
        assert torch.equal(t["gt_classes"], g[f"target{i}/gt_classes"])
    grads = torch.autograd.grad(sum(losses.values()), [sd[k] for k in keys])
    for k, gr in zip(keys, grads):
        ref = float(g["gradnorm/" + k])
        assert abs(float(gr.double().norm()) - ref) <= 2e-4 * ref + 1e-9, k


def test_r50_bottleneck_backbone_matches_reference():
    g = load("g1_backbone_r50_small")
    d = np.load(os.path.join(G, "shapes_r50_backbone.npz"))
    shapes = {str(k): eval(str(s)) for k, s in zip(d["keys"], d["shapes"])}
    sd = gen.seeded_state(shapes, 3)
    res5 = R.backbone_forward(sd, g["x"], depth=50)["res5"]
    assert res5.shape[1] == 2048
    torch.testing.assert_close(res5, g["res5"], rtol=1e-4, atol=1e-5)


def test_eval_tail_matches_reference():
    """G14: the reference's own inference path (model.inference -> predict_probs_K/boxes_K ->
    fast_rcnn_inference_single_image -> detector_postprocess).  The oracle's eval forward reproduces the per-proposal
    scores / boxes; its tail and post-processing reproduce the detections EXACTLY from the reference's tail inputs.
    The NMS behind the fixture is an independent brute-force greedy NMS (make_golden.py), so oracle/det_ops_ref.c is
    pinned by it as well."""
    g = load("g14_eval_tail")
    sd = seeded_sd(1)
    batch = gen.seeded_batch(3, 200, 20, 256, 352, seed=15)
    out = R.eval_forward(sd, batch, depth=18, classifier=g["classifier"], pixel_std=gen.PIXEL_STD)
    total = 0
    for i, (b, (scores, boxes)) in enumerate(zip(batch, out)):
        torch.testing.assert_close(scores, g[f"img{i}/all_scores"], rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(boxes, g[f"img{i}/all_boxes"], rtol=1e-4, atol=1e-3)
        size = tuple(b["image"].shape[-2:])
        rb, rs, rc, ri = R.fast_rcnn_inference_single_image(g[f"img{i}/all_boxes"], g[f"img{i}/all_scores"], size,
                                                            1e-5, 0.3, 100)
        assert torch.equal(rb, g[f"img{i}/raw_boxes"]) and torch.equal(rs, g[f"img{i}/raw_scores"])
        assert torch.equal(rc, g[f"img{i}/raw_classes"]) and torch.equal(ri, g[f"img{i}/raw_inds"])
        oh, ow = (int(v) for v in g[f"img{i}/out_size"])
        assert (oh, ow) == (300 + 40 * i, 500 - 30 * i)
        pb, keep = R.detector_postprocess(rb, size, oh, ow)
        assert torch.equal(pb, g[f"img{i}/out_boxes"]) and torch.equal(rs[keep], g[f"img{i}/out_scores"])
        assert torch.equal(rc[keep], g[f"img{i}/out_classes"]) and torch.equal(ri[keep], g[f"img{i}/out_inds"])
        total += len(rc)
    assert total > 100
    for name, topk in (("agnostic", 40), ("specific", -1)):  # non-finite rows, exact ties, class-specific boxes, top-k
        p = f"tail_{name}/"
        rb, rs, rc, ri = R.fast_rcnn_inference_single_image(g[p + "boxes_in"], g[p + "scores_in"], (210, 330), 0.05,
                                                            0.3, topk)
        assert torch.equal(rb, g[p + "boxes"]) and torch.equal(rs, g[p + "scores"])
        assert torch.equal(rc, g[p + "classes"]) and torch.equal(ri, g[p + "inds"])
        assert len(rc) > 10 and not bool((ri == 20).any())  # the NaN-score row never reaches the output
        assert name == "agnostic" or not bool((ri == 30).any())  # nor the row with an infinite class-specific box


def test_tta_avg_matches_reference():
    """G15: the reference's DatasetMapperTTAAVG + GeneralizedRCNNWithTTAAVG around the reference model (4 views).
    Oracle: per-view eval forward on the restated views, inverse transforms, mean, one tail pass."""
    from tests.helpers import to_inputs
    from wsovod_amd.modeling.test_time_augmentation import DatasetMapperTTAAVG

    g = load("g15_tta_avg")
    sd = seeded_sd(1)
    inp = to_inputs(gen.seeded_batch(1, 60, 20, 256, 352, seed=21))[0]
    views = DatasetMapperTTAAVG([192, 256], 4000, True, 0)(inp)
    assert len(views) == 4
    vb, vs, inv = [], [], []
    for i, v in enumerate(views):
        assert tuple(v["image"].shape) == tuple(int(x) for x in g[f"view{i}/shape"])
        assert float(v["image"].double().sum()) == float(g[f"view{i}/image_checksum"])
        assert torch.equal(v["proposals"].proposal_boxes.tensor, g[f"view{i}/proposal_boxes"])
        b = dict(image=v["image"], boxes=v["proposals"].proposal_boxes.tensor,
                 objectness=v["proposals"].objectness_logits)
        (scores, boxes), = R.eval_forward(sd, [b], depth=18, classifier=g["classifier"], pixel_std=gen.PIXEL_STD)
        vb.append(boxes)
        vs.append(scores)
        inv.append(v["transforms"].inverse().apply_box)
    boxes, scores, rb, rs, rc, _ = R.tta_avg_merge(vb, vs, inv, (256, 352), 1e-5, 0.3, 100)
    torch.testing.assert_close(scores, g["avg_scores"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(boxes, g["avg_boxes"], rtol=1e-4, atol=1e-3)
    # the merge itself, from the reference's averaged tensors: exact
    _, _, rb, rs, rc, _ = R.tta_avg_merge([g["avg_boxes"]], [g["avg_scores"]], [lambda x: x], (256, 352), 1e-5, 0.3, 100)
    assert torch.equal(rb, g["boxes"]) and torch.equal(rs, g["scores"]) and torch.equal(rc, g["classes"])
    assert len(rc) == 100


def test_tta_union_matches_reference():
    """G17: the reference's DatasetMapperTTAUNION + GeneralizedRCNNWithTTAUNION (test_time_augmentation_union.py:66-330)
    around the reference model WITH its RPN branch, 4 views.  The product's mapper reproduces the views (images and the
    loaded boxes moved by the mapper's own transform_proposals) exactly; the oracle's eval forward (RPN boxes + loaded
    boxes -> heads -> tail) reproduces every view's detections; the merge of the reference's own per-view detections
    is exact."""
    from tests.helpers import to_inputs
    from wsovod_amd.modeling.test_time_augmentation import DatasetMapperTTAUNION

    g = load("g17_tta_union")
    d = np.load(os.path.join(G, "shapes_rpn_r18.npz"))
    sd = gen.seeded_state({str(k): eval(str(s)) for k, s in zip(d["keys"], d["shapes"])}, 41)
    inp = to_inputs(gen.seeded_batch(1, 60, 20, 256, 352, seed=23))[0]
    views = DatasetMapperTTAUNION([192, 256], 4000, True, 4000)(inp)
    assert len(views) == 4
    dets, inv = [], []
    for i, v in enumerate(views):
        assert tuple(v["image"].shape) == tuple(int(x) for x in g[f"view{i}/shape"])
        assert float(v["image"].double().sum()) == float(g[f"view{i}/image_checksum"])
        assert torch.equal(v["proposals"].proposal_boxes.tensor, g[f"view{i}/proposal_boxes"])
        assert torch.equal(v["proposals"].objectness_logits, g[f"view{i}/objectness"])
        # the oracle's statement of the mapper's transform_proposals agrees with the product's mapper
        ob, oo = R.tta_union_view_proposals(inp["proposals"].proposal_boxes.tensor, inp["proposals"].objectness_logits,
                                            v["transforms"].apply_box, tuple(v["image"].shape[1:]), 4000)
        assert torch.equal(ob, g[f"view{i}/proposal_boxes"]) and torch.equal(oo, g[f"view{i}/objectness"])
        b = dict(image=v["image"], boxes=ob, objectness=oo)
        rpn = dict(nms_thresh=0.7, pre_nms_topk=2048, post_nms_topk=1024, min_box_size=40.0)
        (scores, boxes), = R.eval_forward(sd, [b], depth=18, classifier=g["classifier"], pixel_std=gen.PIXEL_STD, rpn=rpn)
        pb, pl = rpn["proposals"][0]
        assert torch.equal(pb, g[f"view{i}/rpn_boxes"]) and torch.equal(pl, g[f"view{i}/rpn_logits"])
        rb, rs, rc, _ = R.fast_rcnn_inference_single_image(boxes, scores, tuple(v["image"].shape[1:]), 1e-5, 0.3, 100)
        assert torch.equal(rc, g[f"view{i}/det_classes"])
        torch.testing.assert_close(rb, g[f"view{i}/det_boxes"], rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(rs, g[f"view{i}/det_scores"], rtol=1e-4, atol=1e-6)
        dets.append((g[f"view{i}/det_boxes"], g[f"view{i}/det_scores"], g[f"view{i}/det_classes"]))
        inv.append(v["transforms"].inverse().apply_box)
    pooled, rb, rs, rc = R.tta_union_merge(dets, inv, (256, 352), 20, 0.3, 100)
    assert torch.equal(pooled, g["pooled_boxes"])
    assert torch.equal(rb, g["boxes"]) and torch.equal(rs, g["scores"]) and torch.equal(rc, g["classes"])
    assert len(rc) == 100 and len(pooled) == 400


def test_subsample_matches_reference():
    """G16: the reference's _sample_proposals_wsl beyond BATCH_SIZE_PER_IMAGE / below POSITIVE_FRACTION 1 with the
    deterministic first-k stand-in for subsample_labels; the oracle's keyed form with keys = row index is that rule."""
    g = load("g16_subsample")
    i = 0
    while f"case{i}/params" in g:
        R_, num, frac, K = g[f"case{i}/params"].tolist()
        lab = g[f"case{i}/labels_in"]
        pos, neg = R.subsample_labels_keyed(lab, int(num), frac, int(K), torch.arange(int(R_), dtype=torch.float32))
        out = torch.full_like(lab, -1)
        idx = torch.cat([pos, neg])
        out[idx] = lab[idx]
        assert torch.equal(out, g[f"case{i}/labels_out"])
        kept = int((out != -1).sum())
        assert kept == min(int(num), len(lab)) or frac < 1.0 or int((lab != K).sum()) + int((lab == K).sum()) < num
        i += 1
    assert i == 5


# ----------------------------------------------------------------------------------------
# G18: the oracle's edge branches against the reference's own code (tests/golden/make_golden.py:golden_edges)
# ----------------------------------------------------------------------------------------
def test_edge_whole_step_with_empty_pseudo_gt_matches_reference():
    """Image 1 of the batch has only boxes <= 20 px^2: the miner's candidate list is empty and the reference falls back
    to one dummy target (box +-10000, class 0, score 1, weight 1: roi_heads.py:1181-1207), every proposal of that image
    becomes background; a third of image 2's boxes are filtered too.  Losses, scores, labels (exact), gradients."""
    g = load("g18_edge_branches")
    sd = seeded_sd()
    p = "step/"
    train_keys = [k[len(p + "gradnorm/"):] for k in g if k.startswith(p + "gradnorm/")]
    for k in train_keys:
        sd[k].requires_grad_(True)
    batch = gen.edge_batch(20)
    losses, inter = R.train_forward(sd, batch, depth=18, num_classes=20, pixel_std=gen.PIXEL_STD)
    for k in ("loss_cls_object_mining", "loss_cls_r0", "loss_box_reg_r0"):
        torch.testing.assert_close(losses[k].detach(), g[p + "loss/" + k], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(inter["mining_scores"].detach(), g[p + "mining_scores"], rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(inter["refine_logits"].detach(), g[p + "refine_logits"], rtol=1e-4, atol=1e-4)
    tg, lab = inter["targets"], inter["labelled"]
    assert [len(t["gt_classes"]) for t in tg] == g[p + "pgt/num"].tolist()
    assert torch.equal(torch.cat([t["gt_boxes"] for t in tg]), g[p + "pgt/gt_boxes"])
    assert torch.equal(torch.cat([t["gt_classes"] for t in tg]), g[p + "pgt/gt_classes"])
    torch.testing.assert_close(torch.cat([t["gt_weights"] for t in tg]), g[p + "pgt/gt_weights"], rtol=1e-5, atol=1e-8)
    torch.testing.assert_close(torch.cat([t["gt_scores"] for t in tg]), g[p + "pgt/gt_scores"], rtol=1e-4, atol=1e-8)
    assert tg[1]["gt_boxes"].tolist() == [[-10000.0, -10000.0, 10000.0, 10000.0]] and tg[1]["gt_classes"].tolist() == [0]
    assert torch.equal(torch.cat([l["gt_classes"] for l in lab]), g[p + "label/gt_classes"])
    assert torch.equal(torch.cat([l["gt_boxes"] for l in lab]), g[p + "label/gt_boxes"])
    torch.testing.assert_close(torch.cat([l["gt_weights"] for l in lab]), g[p + "label/gt_weights"], rtol=1e-5, atol=1e-8)
    assert bool((lab[1]["gt_classes"] == 20).all())  # the fallback image: all background
    grads = torch.autograd.grad(sum(losses.values()), [sd[k] for k in train_keys])
    for k, gr in zip(train_keys, grads):
        torch.testing.assert_close(gr.norm(), g[p + "gradnorm/" + k], rtol=2e-4, atol=1e-9)
        torch.testing.assert_close(gen.strided_sample(gr, 1024), g[p + "gradsample/" + k], rtol=2e-3, atol=1e-7)


def _edge_direct_inputs(g):
    p = "direct/"
    nums = g[p + "nums"].tolist()
    boxes = list(g[p + "boxes"].split(nums))
    scores = list(g[p + "scores"].split(nums))
    gt_int = list(g[p + "gt_int"].split(g[p + "gt_int_num"].tolist()))
    return nums, boxes, scores, gt_int


def test_edge_pseudo_gt_mining_on_filtered_candidates_matches_reference():
    """get_pgt_top_k called directly: an image whose boxes all have area EXACTLY 20 (the cut is `> 20`), an image whose
    best-scoring box is filtered (the runner-up must win: the filter precedes the top-k), an image with one survivor."""
    g = load("g18_edge_branches")
    p = "direct/"
    nums, boxes, scores, gt_int = _edge_direct_inputs(g)
    tg = R.get_pgt_top_k(boxes, scores, gt_int, g[p + "img_logits"], 20)
    assert [len(t["gt_classes"]) for t in tg] == g[p + "pgt/num"].tolist() == [2, 1, 1, 2]
    assert torch.equal(torch.cat([t["gt_boxes"] for t in tg]), g[p + "pgt/gt_boxes"])
    assert torch.equal(torch.cat([t["gt_classes"] for t in tg]), g[p + "pgt/gt_classes"])
    assert torch.equal(torch.cat([t["gt_scores"] for t in tg]), g[p + "pgt/gt_scores"])
    assert torch.equal(torch.cat([t["gt_weights"] for t in tg]), g[p + "pgt/gt_weights"])
    lab = R.label_and_sample_proposals_wsl(boxes, tg, 20)
    assert torch.equal(torch.cat([l["gt_classes"] for l in lab]), g[p + "label/gt_classes"])
    assert torch.equal(torch.cat([l["gt_boxes"] for l in lab]), g[p + "label/gt_boxes"])
    assert torch.equal(torch.cat([l["gt_weights"] for l in lab]), g[p + "label/gt_weights"])
    assert torch.equal(torch.cat([l["gt_scores"] for l in lab]), g[p + "label/gt_scores"])


@pytest.mark.parametrize("case", ["ignores", "all_background", "zero_weights", "one_foreground"])
def test_edge_refinement_losses_match_reference(case):
    """Weighted CE / weighted smooth-L1 with -1 ignores (weight forced to 0, excluded from the valid count), an
    all-background batch (no foreground: box loss exactly 0), weights at and below the 1e-12 validity threshold, and a
    single foreground row: values and the gradients w.r.t. logits and deltas."""
    g = load("g18_edge_branches")
    lg = g["loss/logits"].clone().requires_grad_(True)
    dl = g["loss/deltas"].clone().requires_grad_(True)
    lc, lb = R.refinement_losses(lg, dl, g[f"loss/{case}/gt_classes"], g[f"loss/{case}/gt_weights"],
                                 g["loss/proposal_boxes"], g["loss/gt_boxes"], 20)
    torch.testing.assert_close(lc.detach(), g[f"loss/{case}/loss_cls"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(lb.detach().reshape(()), g[f"loss/{case}/loss_box"].reshape(()), rtol=1e-5, atol=1e-8)
    (lc + lb).backward()
    torch.testing.assert_close(lg.grad, g[f"loss/{case}/dlogits"], rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(dl.grad if dl.grad is not None else torch.zeros_like(dl), g[f"loss/{case}/ddeltas"],
                               rtol=1e-4, atol=1e-9)
    if case == "all_background":
        assert float(g[f"loss/{case}/loss_box"]) == 0.0 and float(g[f"loss/{case}/ddeltas"].abs().max()) == 0.0


def test_edge_mil_head_with_one_class_matches_reference():
    """num_classes == 1: a zero column is appended to C and D before the two softmaxes and dropped afterwards
    (fast_rcnn_open_vocabulary.py:338-340,356-357) -- two images, one image, `proposals=None`; BCE loss and gradients."""
    g = load("g18_edge_branches")
    x = gen.edge_features("k1", 40)
    sd = {"m.cls.weight": g["k1/cls_w"].clone().requires_grad_(True), "m.cls.bias": g["k1/cls_b"],
          "m.det.weight": g["k1/det_w"].clone().requires_grad_(True), "m.det.bias": g["k1/det_b"]}
    with torch.no_grad():
        torch.testing.assert_close(R.mining_forward(sd, x, [25, 15], prefix="m."), g["k1/scores_two_images"], rtol=1e-5, atol=1e-9)
        torch.testing.assert_close(R.mining_forward(sd, x[:25], [25], prefix="m."), g["k1/scores_one_image"], rtol=1e-5, atol=1e-9)
        torch.testing.assert_close(R.mining_forward(sd, x, [40], prefix="m."), g["k1/scores_no_proposals"], rtol=1e-5, atol=1e-9)
    assert g["k1/scores_two_images"].shape == (40, 1)
    xs = x.clone().requires_grad_(True)
    loss = R.mining_loss(R.mining_forward(sd, xs, [25, 15], prefix="m."), [25, 15], g["k1/gt_oh"])
    torch.testing.assert_close(loss.detach(), g["k1/loss"], rtol=1e-5, atol=1e-8)
    loss.backward()
    torch.testing.assert_close(gen.strided_sample(xs.grad, 2048), g["k1/dx_sample"], rtol=1e-4, atol=1e-10)
    torch.testing.assert_close(sd["m.cls.weight"].grad, g["k1/dcls_w"], rtol=1e-4, atol=1e-9)
    torch.testing.assert_close(sd["m.det.weight"].grad, g["k1/ddet_w"], rtol=1e-4, atol=1e-9)


def test_edge_whole_step_with_one_class_matches_reference():
    g = load("g18_edge_branches")
    d = np.load(os.path.join(G, "shapes_r18_k1.npz"))
    shapes = {str(k): eval(str(s)) for k, s in zip(d["keys"], d["shapes"])}
    sd = gen.seeded_state(shapes, 5)
    p = "k1/step/"
    train_keys = [k[len(p + "gradnorm/"):] for k in g if k.startswith(p + "gradnorm/")]
    for k in train_keys:
        sd[k].requires_grad_(True)
    losses, inter = R.train_forward(sd, gen.seeded_batch(2, 20, 1, 128, 160, seed=9), depth=18, num_classes=1,
                                    pixel_std=gen.PIXEL_STD)
    assert inter["mining_scores"].shape[1] == 1 and inter["refine_logits"].shape[1] == 2
    for k in ("loss_cls_object_mining", "loss_cls_r0", "loss_box_reg_r0"):
        torch.testing.assert_close(losses[k].detach(), g[p + "loss/" + k], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(inter["pred_class_img_logits"], g[p + "pred_class_img_logits"], rtol=1e-5, atol=1e-8)
    grads = torch.autograd.grad(sum(losses.values()), [sd[k] for k in train_keys])
    for k, gr in zip(train_keys, grads):
        torch.testing.assert_close(gr.norm(), g[p + "gradnorm/" + k], rtol=2e-4, atol=1e-9)


@pytest.mark.parametrize("tag,use_bias,norm", [("bias", -2.0, True), ("nobias", 0.0, True), ("bias_nonorm", 0.75, False)])
def test_edge_ov_classifier_bias_and_coco_shapes_match_reference(tag, use_bias, norm):
    """OpenVocabularyClassifier at BASELINE config 3's shapes (K = 80 classes, D = 768 ViT-L/14 embeddings) with
    `use_bias != 0` (a learnable scalar added to EVERY column, background included: open_vocabulary_classifier.py:35-37,
    103-104) and with norm_weight off (raw dot products, no temperature)."""
    g = load("g18_edge_branches")
    emb, clsf = gen.edge_embeddings(80, 768)
    x = gen.edge_features("ovc", 33)
    sd = gen.seeded_state({"cls.projection.0.weight": (1024, 4096), "cls.projection.0.bias": (1024,),
                           "cls.projection.2.weight": (768, 1024), "cls.projection.2.bias": (768,)}, 23)
    cw = emb.t().contiguous()
    sd["cls.class_weight"] = torch.nn.functional.normalize(cw, p=2, dim=0) if norm else cw
    bias = torch.full((1,), use_bias, requires_grad=True) if use_bias else None
    xg = x.clone().requires_grad_(True)
    out = R.ov_classifier_forward(sd, xg, "cls.", norm_weight=norm, cls_bias=bias)
    tol = dict(rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(out.detach(), g[f"ovc/{tag}/logits_bg"], **tol)
    assert out.shape == (33, 81)
    if use_bias:
        assert torch.all(g[f"ovc/{tag}/logits_bg"][:, -1] == use_bias)  # the background logit is the bias itself
    out.square().mean().backward()
    torch.testing.assert_close(gen.strided_sample(xg.grad, 2048), g[f"ovc/{tag}/dx_sample"], rtol=2e-3, atol=1e-7)
    if use_bias:
        torch.testing.assert_close(bias.grad, g[f"ovc/{tag}/dcls_bias"], rtol=1e-4, atol=1e-7)
    with torch.no_grad():
        torch.testing.assert_close(R.ov_classifier_forward(sd, x, "cls.", norm_weight=norm, cls_bias=bias, append_background=False),
                                   g[f"ovc/{tag}/logits_nobg"], **tol)
        torch.testing.assert_close(R.ov_classifier_forward(sd, x, "cls.", norm_weight=norm, cls_bias=bias, classifier=clsf),
                                   g[f"ovc/{tag}/logits_classifier"], **tol)


def test_trainable_res5_step_matches_reference():
    """G19: MODEL.BACKBONE.FREEZE_AT = 4 -- the reference's whole step with res5 trainable (gradient through its own RoIPool
    backward and through the GAP of the data-aware head into the stage).  The oracle with `backbone_grad=True`: losses,
    scores, logits, labels and the gradient of every trainable tensor, the res5 convs included."""
    g = load("g19_freeze_at_4")
    sd = seeded_sd()
    train_keys = [str(k) for k in g["train_keys"]]
    assert sum(k.startswith("backbone.res5.") for k in train_keys) >= 5
    for k in train_keys:
        sd[k].requires_grad_(True)
    batch = gen.seeded_batch(2, 24, 20, 160, 208, seed=11)
    losses, inter = R.train_forward(sd, batch, depth=18, num_classes=20, pixel_std=gen.PIXEL_STD, backbone_grad=True)
    for k in ("loss_cls_object_mining", "loss_cls_r0", "loss_box_reg_r0"):
        torch.testing.assert_close(losses[k].detach(), g["loss/" + k], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(inter["mining_scores"].detach(), g["mining_scores"], rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(inter["refine_logits"].detach(), g["refine_logits"], rtol=1e-4, atol=1e-4)
    assert torch.equal(torch.cat([l["gt_classes"] for l in inter["labelled"]]), g["label/gt_classes"])
    grads = torch.autograd.grad(sum(losses.values()), [sd[k] for k in train_keys])
    for k, gr in zip(train_keys, grads):
        torch.testing.assert_close(gr.norm(), g["gradnorm/" + k], rtol=3e-4, atol=1e-9, msg=lambda m: f"{k}: {m}")
        torch.testing.assert_close(gen.strided_sample(gr, 1024), g["gradsample/" + k], rtol=3e-3,
                                   atol=1e-5 * float(g["gradsample/" + k].abs().max()) + 1e-9, msg=lambda m: f"{k}: {m}")
