"""GPU: the HIP path on the reference-generated EDGE fixtures (tests/golden/g18_edge_branches.npz -- produced by
tests/golden/make_golden.py:golden_edges from the reference's own code; tests/test_oracle_golden.py pins the oracle on
the same vectors): empty pseudo-GT fallbacks, the filter-before-top-k order, -1 ignores / all-background / threshold
weights in the refinement losses, the MIL head at one class, OpenVocabularyClassifier with a bias at K = 80 / D = 768.
Indices and labels bit-exact; floating point within the north star's 1e-3 on logits / scores (tighter where stated)."""
import os

import numpy as np
import pytest
import torch

from tests.golden import gen
from tests.helpers import G, build_seeded_hip_model, load_golden, to_inputs

pytestmark = pytest.mark.gpu


def _compact(o, gt_counts):
    """The mining kernel's packed layout (image g owns rows [start[g], start[g] + count[g]) of (T, .) buffers, start = the
    prefix sum of the images' GT-class counts; modeling/roi_heads.py:PseudoTargets) -> the reference's concatenation."""
    starts = np.concatenate([[0], np.cumsum(gt_counts)])[:-1]
    counts = o["pgt_count"].cpu().tolist()
    rows = torch.cat([torch.arange(s, s + c) for s, c in zip(starts, counts)])
    return {k: o[k].cpu()[rows] for k in ("pgt_boxes", "pgt_classes", "pgt_scores", "pgt_weights")}


def _run(model, batch):
    captured = {}
    rh = model.roi_heads
    orig_m, orig_r = rh.object_miner.forward, rh.box_refinery[0].forward

    def cap(name, fn):
        def w(*a, **k):
            o = fn(*a, **k)
            captured[name] = o
            return o
        return w

    rh.object_miner.forward = cap("miner", orig_m)
    rh.box_refinery[0].forward = cap("refine", orig_r)
    losses = model(to_inputs(batch))
    rh.object_miner.forward, rh.box_refinery[0].forward = orig_m, orig_r
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    return losses, captured, rh._last_pgt


@pytest.mark.parametrize("precision", ["fp32", "parity"])
def test_whole_step_with_empty_pseudo_gt_matches_reference_golden(gpu, precision):
    g = load_golden("g18_edge_branches")
    p = "step/"
    cfg, model, sd = build_seeded_hip_model(precision)
    batch = gen.edge_batch(20)
    losses, cap, pgt = _run(model, batch)
    scores, logits = cap["miner"][0].detach().float().cpu(), cap["refine"][0].detach().float().cpu()
    assert (scores - g[p + "mining_scores"]).abs().max() < 1e-3
    assert (logits - g[p + "refine_logits"]).abs().max() < 1e-3
    for k in ("loss_cls_object_mining", "loss_cls_r0", "loss_box_reg_r0"):
        torch.testing.assert_close(losses[k].detach().cpu(), g[p + "loss/" + k], rtol=1e-3, atol=1e-5)
    # mining + labelling: bit-exact, the fallback target of image 1 included
    assert pgt["pgt_count"].cpu().tolist() == g[p + "pgt/num"].tolist()
    mined = _compact(pgt, [len(torch.unique(b["gt_classes"])) for b in batch])
    assert torch.equal(mined["pgt_boxes"], g[p + "pgt/gt_boxes"])
    assert torch.equal(mined["pgt_classes"], g[p + "pgt/gt_classes"])
    wtol = dict(rtol=1e-4, atol=1e-7) if precision == "fp32" else dict(rtol=1e-3, atol=1e-4)  # image scores: 1e-3 bound
    torch.testing.assert_close(mined["pgt_weights"], g[p + "pgt/gt_weights"], **wtol)
    assert torch.equal(pgt["gt_classes"].cpu(), g[p + "label/gt_classes"])
    assert torch.equal(pgt["gt_boxes"].cpu(), g[p + "label/gt_boxes"])
    torch.testing.assert_close(pgt["gt_weights"].cpu(), g[p + "label/gt_weights"], **wtol)
    n0 = int(g[p + "pgt/num"][0])
    assert mined["pgt_boxes"][n0].tolist() == [-10000.0, -10000.0, 10000.0, 10000.0] and int(mined["pgt_classes"][n0]) == 0
    rtol = 2e-3 if precision == "fp32" else 1e-2
    for k, q in model.named_parameters():
        if q.requires_grad:
            torch.testing.assert_close(q.grad.detach().float().cpu().norm(), g[p + "gradnorm/" + k], rtol=rtol, atol=1e-7,
                                       msg=lambda m: f"{k}: {m}")


def test_pgt_kernel_on_filtered_candidates_matches_reference_golden(gpu):
    """The one-launch mining + labelling kernel on the reference's direct get_pgt_top_k / label_and_sample vectors: boxes of
    area exactly 20 are filtered (`> 20`), a filtered best box gives way to the runner-up, the empty image gets the
    reference's dummy target -- counts, boxes, classes, scores, weights and every per-proposal field bit for bit."""
    from wsovod_amd.layers import hip_ops as H

    g = load_golden("g18_edge_branches")
    p = "direct/"
    nums = g[p + "nums"].tolist()
    seg = torch.tensor(np.concatenate([[0], np.cumsum(nums)]), dtype=torch.int32, device=gpu)
    goff = torch.tensor(np.concatenate([[0], np.cumsum(g[p + "gt_int_num"].tolist())]), dtype=torch.int32, device=gpu)
    o = H.pgt_mine_and_label(g[p + "scores"].to(gpu), g[p + "boxes"].to(gpu), seg, g[p + "gt_int"].to(gpu), goff,
                             g[p + "img_logits"].to(gpu), 20, 0.5)
    torch.cuda.synchronize()
    assert o["pgt_count"].cpu().tolist() == g[p + "pgt/num"].tolist()
    mined = _compact(o, g[p + "gt_int_num"].tolist())
    assert torch.equal(mined["pgt_boxes"], g[p + "pgt/gt_boxes"])
    assert torch.equal(mined["pgt_classes"], g[p + "pgt/gt_classes"])
    assert torch.equal(mined["pgt_scores"], g[p + "pgt/gt_scores"])
    assert torch.equal(mined["pgt_weights"], g[p + "pgt/gt_weights"])
    for f in ("gt_classes", "gt_boxes", "gt_weights", "gt_scores"):
        assert torch.equal(o[f].cpu(), g[p + "label/" + f]), f


@pytest.mark.parametrize("case", ["ignores", "all_background", "zero_weights", "one_foreground"])
def test_refinement_loss_kernels_on_edge_labels_match_reference_golden(gpu, case):
    from wsovod_amd.layers import functions as Fn

    g = load_golden("g18_edge_branches")
    gc, w = g[f"loss/{case}/gt_classes"], g[f"loss/{case}/gt_weights"].clone()
    w[gc == -1] = 0.0  # the heads zero the weights of ignored rows before the kernels (fast_rcnn_open_vocabulary.py:814)
    lg = g["loss/logits"].to(gpu).requires_grad_(True)
    dl = g["loss/deltas"].to(gpu).requires_grad_(True)
    lc = Fn.weighted_cross_entropy(lg, gc.to(gpu), w.to(gpu), True)
    lb = Fn.weighted_l1_box_loss(dl, g["loss/proposal_boxes"].to(gpu), g["loss/gt_boxes"].to(gpu), gc.to(gpu), w.to(gpu), 20,
                                 (10.0, 10.0, 5.0, 5.0), 0.0)
    (lc + lb).backward()
    torch.testing.assert_close(lc.detach().cpu(), g[f"loss/{case}/loss_cls"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(lb.detach().cpu().reshape(()), g[f"loss/{case}/loss_box"].reshape(()), rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(lg.grad.cpu(), g[f"loss/{case}/dlogits"], rtol=1e-3, atol=1e-7)
    torch.testing.assert_close(dl.grad.cpu(), g[f"loss/{case}/ddeltas"], rtol=1e-4, atol=1e-8)


def test_mil_head_with_one_class_matches_reference_golden(gpu):
    from wsovod_amd.modeling.box_regression import Box2BoxTransform
    from wsovod_amd.modeling.fast_rcnn_open_vocabulary import ObjectMiningOutputLayers
    from wsovod_amd.structures import Boxes, Instances

    g = load_golden("g18_edge_branches")
    layer = ObjectMiningOutputLayers(4096, box2box_transform=Box2BoxTransform((10., 10., 5., 5.)), num_classes=1).to(gpu)
    layer.load_state_dict({"cls.weight": g["k1/cls_w"], "cls.bias": g["k1/cls_b"], "det.weight": g["k1/det_w"],
                           "det.bias": g["k1/det_b"]}, strict=False)
    x = gen.edge_features("k1", 40).to(gpu)
    pl = [Instances((96, 128), proposal_boxes=Boxes(torch.zeros(n, 4, device=gpu))) for n in (25, 15)]
    with torch.no_grad():
        s2, _ = layer(x, pl)
        s1, _ = layer(x[:25].contiguous(), pl[:1])
        s0, _ = layer(x, None)
    assert s2.shape == (40, 1)
    torch.testing.assert_close(s2.cpu(), g["k1/scores_two_images"], rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(s1.cpu(), g["k1/scores_one_image"], rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(s0.cpu(), g["k1/scores_no_proposals"], rtol=1e-4, atol=1e-7)
    xs = x.clone().requires_grad_(True)
    loss = layer.losses(layer(xs, pl), pl, g["k1/gt_oh"].to(gpu))["loss_cls_object_mining"]
    loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), g["k1/loss"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(gen.strided_sample(xs.grad.cpu(), 2048), g["k1/dx_sample"], rtol=2e-3, atol=1e-8)
    torch.testing.assert_close(layer.cls.weight.grad.cpu(), g["k1/dcls_w"], rtol=2e-3, atol=1e-7)
    torch.testing.assert_close(layer.det.weight.grad.cpu(), g["k1/ddet_w"], rtol=2e-3, atol=1e-7)


@pytest.mark.parametrize("precision", ["fp32", "parity"])
@pytest.mark.parametrize("tag,use_bias,norm", [("bias", -2.0, True), ("nobias", 0.0, True), ("bias_nonorm", 0.75, False)])
def test_ov_classifier_bias_and_coco_shapes_match_reference_golden(gpu, precision, tag, use_bias, norm):
    """The region x text head at config 3's shapes (K = 80, D = 768) with use_bias / norm_weight variants against the
    reference's own outputs: logits within 1e-3 in fp32 AND in the benchmarked `parity` precision."""
    import pickle
    import tempfile

    from wsovod_amd.layers import hip_ops as H
    from wsovod_amd.modeling.class_heads import OpenVocabularyClassifier
    from wsovod_amd.structures import ShapeSpec

    g = load_golden("g18_edge_branches")
    emb, clsf = gen.edge_embeddings(80, 768)
    path = os.path.join(tempfile.mkdtemp(prefix="edge_"), "emb.pkl")
    with open(path, "wb") as f:
        pickle.dump(emb.numpy(), f)
    head = OpenVocabularyClassifier(ShapeSpec(channels=4096), num_classes=80, weight_path=path, weight_dim=768,
                                    use_bias=use_bias, norm_weight=norm, norm_temperature=50.0)
    st = gen.seeded_state({"cls.projection.0.weight": (1024, 4096), "cls.projection.0.bias": (1024,),
                           "cls.projection.2.weight": (768, 1024), "cls.projection.2.bias": (768,)}, 23)
    head.projection.load_state_dict({k[len("cls.projection."):]: v for k, v in st.items()})
    head = head.to(gpu)
    x = gen.edge_features("ovc", 33).to(gpu)
    mode = {"fp32": False, "parity": "x2"}[precision]
    tol = 1e-3
    with H.x3_mode(mode):
        def enc(t):
            return H.x2_encode(t) if precision == "parity" else t
        xg = x.clone().requires_grad_(True)
        out = head(enc(xg) if precision == "fp32" else enc(x), None, append_background=True)
        assert out.shape == (33, 81)
        assert (out.detach().float().cpu() - g[f"ovc/{tag}/logits_bg"]).abs().max() < tol
        if use_bias:
            assert torch.all(out[:, -1].detach().float().cpu() == use_bias)
        if precision == "fp32":
            out.float().square().mean().backward()
            torch.testing.assert_close(gen.strided_sample(xg.grad.cpu(), 2048), g[f"ovc/{tag}/dx_sample"], rtol=5e-3,
                                       atol=2e-3 * float(g[f"ovc/{tag}/dx_sample"].abs().max()))
            if use_bias:
                torch.testing.assert_close(head.cls_bias.grad.cpu(), g[f"ovc/{tag}/dcls_bias"], rtol=1e-3, atol=1e-6)
        with torch.no_grad():
            nobg = head(enc(x), None, append_background=False)
            call = head(enc(x), clsf.to(gpu), append_background=True)
        assert (nobg.float().cpu() - g[f"ovc/{tag}/logits_nobg"]).abs().max() < tol
        assert (call.float().cpu() - g[f"ovc/{tag}/logits_classifier"]).abs().max() < tol


def test_whole_step_with_one_class_matches_reference_golden(gpu):
    from wsovod_amd.testing import build_hot_path_model

    g = load_golden("g18_edge_branches")
    d = np.load(os.path.join(G, "shapes_r18_k1.npz"))
    shapes = {str(k): eval(str(s)) for k, s in zip(d["keys"], d["shapes"])}
    cfg, model = build_hot_path_model(seed=0, precision="fp32", K=1, device="cuda:0", calibrate_synthetic=False)
    cfg.MODEL.PIXEL_STD = list(gen.PIXEL_STD)
    model._std = [float(v) for v in gen.PIXEL_STD]
    model.load_state_dict(gen.seeded_state(shapes, 5), strict=True)
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    p = "k1/step/"
    losses, cap, pgt = _run(model, gen.seeded_batch(2, 20, 1, 128, 160, seed=9))
    assert cap["miner"][0].shape[1] == 1 and cap["refine"][0].shape[1] == 2
    for k in ("loss_cls_object_mining", "loss_cls_r0", "loss_box_reg_r0"):
        torch.testing.assert_close(losses[k].detach().cpu(), g[p + "loss/" + k], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(model.roi_heads.pred_class_img_logits.cpu(), g[p + "pred_class_img_logits"], rtol=1e-4, atol=1e-7)
    for k, q in model.named_parameters():
        if q.requires_grad:
            torch.testing.assert_close(q.grad.detach().float().cpu().norm(), g[p + "gradnorm/" + k], rtol=2e-3, atol=1e-7,
                                       msg=lambda m: f"{k}: {m}")
