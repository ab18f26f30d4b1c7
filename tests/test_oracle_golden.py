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
