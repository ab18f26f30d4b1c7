"""-m gpu: the RPN branch on the HIP kernels against the reference's golden step (g12) and the
oracle.  fp32 parity mode; detectron2's random anchor sub-sampling is replaced by gen.first_k_subsample on both
sides (the golden was generated the same way)."""
import numpy as np
import os
import pytest
import torch

from oracle import wsovod_ref as R
from tests.conftest import *  # noqa: F401,F403
from tests.golden import gen
from tests.helpers import G, load_golden, to_inputs

pytestmark = pytest.mark.gpu


def build_rpn_model(precision="fp32"):
    from wsovod_amd.modeling import build_model, sampling
    from wsovod_amd.testing import hot_path_cfg

    cfg = hot_path_cfg(precision=precision, device="cuda:0", rpn=True)
    cfg.SOLVER.MAX_ITER = 4000
    torch.manual_seed(0)
    model = build_model(cfg)
    model._std = [float(v) for v in gen.PIXEL_STD]
    d = np.load(os.path.join(G, "shapes_rpn_r18.npz"))
    shapes = {str(k): eval(str(s)) for k, s in zip(d["keys"], d["shapes"])}
    assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == {k: tuple(v) for k, v in shapes.items()}
    sd = gen.seeded_state(shapes, 41)
    model.load_state_dict(sd, strict=True)
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    model.roi_heads.iter = 1000  # the reference reads the trainer's iteration counter (rcnn_wsovod.py:181-184)
    return cfg, model, sd, sampling


def first_k_keys(model, monkeypatch):
    """Sampling keys = anchor index: the batched sampler then picks the first-k anchors, the rule the golden step
    and the oracle use (gen.first_k_subsample) in place of detectron2's random permutation."""
    monkeypatch.setattr(model.proposal_generator, "_sample_keys",
                        lambda B, A, dev: torch.arange(A, device=dev, dtype=torch.float32).expand(B, A))


def match_fraction(a, b, tol=0.05):
    """fraction of rows of a that have a row of b within tol (max abs coordinate difference)."""
    if len(a) == 0:
        return 1.0
    d = (a[:, None, :] - b[None, :, :]).abs().amax(dim=2)
    return float((d.min(dim=1).values < tol).float().mean())


def test_rpn_step_matches_reference_golden(gpu, monkeypatch):
    g = load_golden("g12_rpn_train_step")
    cfg, model, sd, sampling = build_rpn_model("fp32")
    first_k_keys(model, monkeypatch)
    batch = gen.seeded_batch(2, 40, 20, 256, 352, seed=43)
    losses = model(to_inputs(batch))
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    pg = model.proposal_generator
    # anchor labels after sampling: identical up to the few anchors whose IoU with a pseudo-GT box (decoded through
    # expf on the device) sits within rounding of a threshold
    assert (pg.sampled_labels.cpu() != g["anchor_labels"]).sum() <= 4
    torch.testing.assert_close(pg.pred_objectness_logits[0].detach().cpu(), g["rpn_logits"], rtol=1e-4, atol=2e-4)
    torch.testing.assert_close(gen.strided_sample(pg.pred_anchor_deltas[0].detach().cpu(), 8192), g["rpn_deltas_sample"],
                               rtol=1e-4, atol=2e-4)
    # proposals: decoded through expf and selected by sort + NMS from logits that differ in the last bits, so the
    # set is compared geometrically (the index kernels themselves are bit-exact: tests/test_gpu_proposals.py)
    for i, p in enumerate(model.rpn_proposals):
        ref_b, ref_s = g[f"prop{i}/boxes"], g[f"prop{i}/logits"]
        got_b = p.proposal_boxes.tensor.cpu()
        assert abs(len(got_b) - len(ref_b)) <= 2
        assert match_fraction(ref_b, got_b) >= 0.95 and match_fraction(got_b, ref_b) >= 0.95
        assert float(p.objectness_logits.max()) <= 0.25 + 1e-6  # sigmoid * iter / MAX_ITER
    for i, t in enumerate(model.roi_heads.proposal_targets):
        torch.testing.assert_close(t.gt_boxes.tensor.cpu(), g[f"target{i}/gt_boxes"], rtol=1e-4, atol=1e-2)
        assert torch.equal(t.gt_classes.cpu(), g[f"target{i}/gt_classes"])
    for k in ("loss_cls_object_mining", "loss_cls_r0", "loss_box_reg_r0", "loss_rpn_cls", "loss_rpn_loc"):
        torch.testing.assert_close(losses[k].detach().cpu(), g["loss/" + k], rtol=5e-3, atol=1e-5,
                                   msg=lambda m: f"{k}: {m}")
    for k, q in model.named_parameters():
        if q.requires_grad:
            ref = float(g["gradnorm/" + k])
            got = float(q.grad.double().norm())
            assert abs(got - ref) <= 1e-2 * ref + 1e-6, (k, got, ref)


def test_rpn_head_gradients_match_oracle(gpu, monkeypatch):
    """Given identical proposals (the oracle is fed the HIP path's own RPN boxes), every loss and the RPN-head
    gradients agree tightly -- isolates the sparse-row weight-gradient path (im2col rows + GEMMs)."""
    cfg, model, sd, sampling = build_rpn_model("fp32")
    first_k_keys(model, monkeypatch)
    batch = gen.seeded_batch(2, 40, 20, 256, 352, seed=47)
    losses = model(to_inputs(batch))
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    pg = model.proposal_generator
    k = pg.pre_nms_topk[True]
    props = []
    for p in model.rpn_proposals:  # undo sigmoid * ramp: the oracle applies it itself
        s = p.objectness_logits.cpu() / 0.25
        props.append((p.proposal_boxes.tensor.cpu(), torch.log(s / (1 - s))))
    sdc = {kk: v.clone() for kk, v in sd.items()}
    keys = [kk for kk in sdc if kk.startswith("proposal_generator.")]
    for kk in keys:
        sdc[kk].requires_grad_(True)
    ref_losses, inter = R.train_forward(sdc, batch, depth=18, num_classes=20, pixel_std=gen.PIXEL_STD,
                                        rpn=dict(cur_iter=1000, max_iter=4000, subsample=gen.first_k_subsample,
                                                 proposals=props))
    for name, v in ref_losses.items():
        torch.testing.assert_close(losses[name].detach().cpu(), v.detach(), rtol=2e-3, atol=1e-5,
                                   msg=lambda m: f"{name}: {m}")
    grads = torch.autograd.grad(ref_losses["loss_rpn_cls"] + ref_losses["loss_rpn_loc"], [sdc[kk] for kk in keys])
    P = dict(model.named_parameters())
    for kk, gr in zip(keys, grads):
        got = P[kk].grad.detach().float().cpu()
        assert (got - gr).abs().max() <= 2e-3 * gr.abs().max() + 1e-7, kk


def test_anchor_labels_and_losses_match_oracle_on_the_same_targets(gpu, monkeypatch):
    """The batched labelling kernel + top-k sampling + fixed-size losses against the oracle's detectron2-style
    restatement (Matcher with low-quality matches, first-k sub-sampling, masked sums) on the SAME pseudo-GT boxes and
    predictions; and the reference-shaped list interface gives the same numbers as the packed one."""
    cfg, model, sd, sampling = build_rpn_model("fp32")
    first_k_keys(model, monkeypatch)
    batch = gen.seeded_batch(3, 30, 20, 256, 352, seed=53)
    losses = model(to_inputs(batch))
    pg = model.proposal_generator
    targets = model.roi_heads.proposal_targets
    anchors = pg.anchors[0].tensor.cpu()
    ref_targets = [dict(gt_boxes=t.gt_boxes.tensor.cpu()) for t in targets]
    ref, ref_labels = R.rpn_losses(anchors, pg.pred_objectness_logits[0].detach().cpu(),
                                   pg.pred_anchor_deltas[0].detach().cpu(), ref_targets, gen.first_k_subsample,
                                   thresholds=(0.2, 0.6))
    assert torch.equal(pg.sampled_labels.cpu(), ref_labels)
    for k in ("loss_rpn_cls", "loss_rpn_loc"):
        torch.testing.assert_close(losses[k].detach().cpu(), ref[k], rtol=1e-5, atol=1e-7)
    gt_labels, gt_boxes = pg.label_and_sample_anchors(pg.anchors, list(targets))  # list interface
    assert torch.equal(torch.stack(gt_labels), pg.sampled_labels)
    via_list = pg.losses(pg.anchors, pg.pred_objectness_logits, gt_labels, pg.pred_anchor_deltas, gt_boxes)
    for k in ("loss_rpn_cls", "loss_rpn_loc"):
        torch.testing.assert_close(via_list[k], losses[k], rtol=1e-6, atol=1e-8)
    assert int((pg.sampled_labels >= 0).sum(dim=1).max()) <= pg.batch_size_per_image


def test_rpn_weights_are_live_across_optimizer_steps(gpu, monkeypatch):
    """The RPN conv runs on a folded / re-laid-out copy of its weight that is cached on the parameter's version: after an
    optimizer step (raw-pointer update) the next forward must see the NEW weights."""
    from wsovod_amd.engine import build_optimizer

    cfg, model, sd, sampling = build_rpn_model("fp32")
    first_k_keys(model, monkeypatch)
    cfg.SOLVER.BASE_LR = 0.05
    opt = build_optimizer(cfg, model)
    batch = to_inputs(gen.seeded_batch(2, 30, 20, 256, 352, seed=71))
    pg = model.proposal_generator
    logits = []
    for it in range(2):
        opt.zero_grad(set_to_none=True)
        losses = model(batch)
        logits.append(pg.pred_objectness_logits[0].detach().clone())
        sum(losses.values()).backward()
        opt.step()
    conv = pg.rpn_head.conv
    x = torch.randn(1, 8, 8, 512, device=gpu)
    from wsovod_amd.modeling.backbone import hip_conv
    got = hip_conv(x, conv, relu=False).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), conv.weight.detach(), conv.bias.detach(), padding=1)
    torch.testing.assert_close(got, ref, rtol=1e-3, atol=1e-3)  # folded copy == current parameter
    assert not torch.equal(logits[0], logits[1])


def test_rpn_eval_inference_runs(gpu):
    cfg, model, sd, _ = build_rpn_model("fp32")
    model.eval()
    batch = gen.seeded_batch(2, 30, 20, 256, 352, seed=49)
    out = model(to_inputs(batch), classifier=torch.randn(20, 512, device=gpu))
    assert len(out) == 2 and len(model.rpn_proposals[0]) > 0
    inst = out[0]["instances"]
    assert inst.pred_boxes.tensor.shape[1] == 4 and len(inst.scores) == len(inst.pred_classes)


def test_bf16_rpn_step_is_finite_and_close(gpu, monkeypatch):
    g = load_golden("g12_rpn_train_step")
    cfg, model, sd, sampling = build_rpn_model("bf16")
    first_k_keys(model, monkeypatch)
    batch = gen.seeded_batch(2, 40, 20, 256, 352, seed=43)
    losses = model(to_inputs(batch))
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    lo = model.proposal_generator.pred_objectness_logits[0].detach().float().cpu()
    assert (lo - g["rpn_logits"]).abs().max() < 0.05 * g["rpn_logits"].abs().max()  # bf16 conv inputs: ~1e-2 relative
    # the localisation loss depends on WHICH proposal the heads pick as pseudo ground truth; bf16 scores may pick a
    # neighbouring box, so it is only bounded loosely (observed 10 %)
    torch.testing.assert_close(losses["loss_rpn_cls"].detach().cpu(), g["loss/loss_rpn_cls"], rtol=0.1, atol=1e-3)
    torch.testing.assert_close(losses["loss_rpn_loc"].detach().cpu(), g["loss/loss_rpn_loc"], rtol=0.3, atol=1e-3)
    for k, q in model.named_parameters():
        if q.requires_grad:
            assert torch.isfinite(q.grad).all(), k


def test_shipped_proposal_counts_with_rpn_subsample_like_the_reference(gpu, monkeypatch):
    """The shipped form of the config (Base-RCNN-DilatedC5.yaml:12,58,84): 4000 loaded boxes + up to 1024 RPN boxes
    per image against SAMPLING.BATCH_SIZE_PER_IMAGE = 4096 -> the ROI heads sub-sample (roi_heads.py:1566-1610; this
    case raised NotImplementedError in round 1).  The oracle is fed the HIP path's own RPN boxes; first-k keys on both
    sides: per-proposal labels (incl. which rows are ignored) exact, losses as in the other fp32 RPN tests."""
    cfg, model, sd, sampling = build_rpn_model("fp32")
    first_k_keys(model, monkeypatch)
    monkeypatch.setattr(model.roi_heads, "_sample_keys",
                        lambda n, dev: torch.arange(n, dtype=torch.float32, device=dev))
    # (at this small image size only ~40 RPN boxes survive NMS, so the loaded set is sized to cross 4096 with them)
    batch = [gen.seeded_batch(1, 4087, 20, 256, 352, seed=61, edge_cases=False)[0],
             gen.seeded_batch(1, 4007, 20, 256, 352, seed=62, edge_cases=False)[0]]
    assert [len(b["boxes"]) for b in batch] == [4080, 4000]
    losses = model(to_inputs(batch))
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    props = []
    for p in model.rpn_proposals:  # undo sigmoid * ramp: the oracle applies it itself
        s = p.objectness_logits.cpu() / 0.25
        props.append((p.proposal_boxes.tensor.cpu(), torch.log(s / (1 - s))))
    nums = [len(b["boxes"]) + len(pb) for b, (pb, _) in zip(batch, props)]
    assert max(nums) > 4096, nums  # the case that needs sub-sampling
    sdc = {kk: v.clone() for kk, v in sd.items()}
    ref_losses, inter = R.train_forward(sdc, batch, depth=18, num_classes=20, pixel_std=gen.PIXEL_STD,
                                        rpn=dict(cur_iter=1000, max_iter=4000, subsample=gen.first_k_subsample,
                                                 proposals=props),
                                        sampling=dict(batch_size_per_image=4096, positive_fraction=1.0,
                                                      keys=lambda n: torch.arange(n, dtype=torch.float32)))
    lab = torch.cat([l["gt_classes"] for l in inter["labelled"]])
    got = model.roi_heads._last_pgt["gt_classes"].cpu()
    assert torch.equal(got, lab)
    off = 0
    for n in nums:
        assert int((lab[off:off + n] != -1).sum()) == min(n, 4096)
        off += n
    for k in ("loss_cls_object_mining", "loss_cls_r0", "loss_box_reg_r0"):
        torch.testing.assert_close(losses[k].detach().cpu(), ref_losses[k].detach(), rtol=5e-3, atol=1e-5,
                                   msg=lambda m: f"{k}: {m}")


def test_bf16x3_rpn_step_matches_reference_golden(gpu, monkeypatch):
    """The RPN branch in bf16x3 (fp32 tensors, bf16 MFMA over hi/lo-split operands; incl. the sparse-row weight-gradient
    GEMMs of the RPN head): the reference's golden step to the fp32-mode tolerances."""
    g = load_golden("g12_rpn_train_step")
    cfg, model, sd, sampling = build_rpn_model("bf16x3")
    first_k_keys(model, monkeypatch)
    batch = gen.seeded_batch(2, 40, 20, 256, 352, seed=43)
    losses = model(to_inputs(batch))
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    pg = model.proposal_generator
    torch.testing.assert_close(pg.pred_objectness_logits[0].detach().cpu(), g["rpn_logits"], rtol=1e-3, atol=1e-3)
    for k in ("loss_cls_object_mining", "loss_cls_r0", "loss_box_reg_r0", "loss_rpn_cls", "loss_rpn_loc"):
        torch.testing.assert_close(losses[k].detach().cpu(), g["loss/" + k], rtol=5e-3, atol=1e-5,
                                   msg=lambda m: f"{k}: {m}")
    for k, q in model.named_parameters():
        if q.requires_grad:
            ref = float(g["gradnorm/" + k])
            got = float(q.grad.double().norm())
            assert abs(got - ref) <= 2e-2 * ref + 1e-6, (k, got, ref)


def test_tta_union_matches_reference_golden(gpu):
    """G17 (the reference's DatasetMapperTTAUNION + GeneralizedRCNNWithTTAUNION around the reference model with its RPN
    branch, 4 views): the HIP model under the product's TTA-UNION wrapper.  Views (images, moved loaded boxes) exact;
    each view's RPN proposals and detections against the reference's (fp32 mode: same classes in the same order, boxes
    1e-2 px, scores 1e-4); the merge of the reference's own per-view detections through the HIP tail is exact; the end
    to end result has the reference's detections."""
    from wsovod_amd.modeling import GeneralizedRCNNWithTTAUNION
    from wsovod_amd.modeling.test_time_augmentation import DatasetMapperTTAUNION

    g = load_golden("g17_tta_union")
    cfg, model, sd, _ = build_rpn_model("fp32")
    model.eval()
    model.classifier = g["classifier"].to(gpu)
    cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST = 0.3
    cfg.TEST.DETECTIONS_PER_IMAGE = 100
    tta = GeneralizedRCNNWithTTAUNION(cfg, model, DatasetMapperTTAUNION([192, 256], 4000, True, 4000))
    inp = to_inputs(gen.seeded_batch(1, 60, 20, 256, 352, seed=23))[0]
    seen = {}
    orig_run = tta._run_model

    def run(aug):
        for i, a in enumerate(aug):
            assert float(a["image"].double().sum()) == float(g[f"view{i}/image_checksum"])
            assert torch.equal(a["proposals"].proposal_boxes.tensor.cpu(), g[f"view{i}/proposal_boxes"])
            assert torch.equal(a["proposals"].objectness_logits.cpu(), g[f"view{i}/objectness"])
        out = orig_run(aug)
        seen["dets"] = out[0]
        return out

    tta._run_model = run
    orig_get = tta._get_augmented_boxes

    def get(aug, tfms):
        o = orig_get(aug, tfms)
        seen["tfms"] = tfms
        return o

    tta._get_augmented_boxes = get
    out = tta([inp])[0]["instances"]
    assert len(seen["dets"]) == 4
    for i, det in enumerate(seen["dets"]):
        n = len(g[f"view{i}/det_scores"])
        assert len(det) == n
        same = det.pred_classes.cpu() == g[f"view{i}/det_classes"]
        assert float(same.float().mean()) > 0.97, i  # (near-tied scores may swap neighbours between fp32 implementations)
        torch.testing.assert_close(det.scores.cpu(), g[f"view{i}/det_scores"], rtol=1e-3, atol=1e-4)
        assert float((det.pred_boxes.tensor.cpu()[same] - g[f"view{i}/det_boxes"][same]).abs().max()) < 5e-2
    # the merge itself, fed the reference's per-view detections: exact (boxes, scores, classes, order)
    back = [torch.from_numpy(t.inverse().apply_box(g[f"view{i}/det_boxes"].numpy())) for i, t in enumerate(seen["tfms"])]
    pooled = torch.cat(back).to(gpu)
    assert torch.equal(pooled.cpu(), g["pooled_boxes"])
    merged = tta._merge_detections(pooled, list(g["pooled_scores"].to(gpu)), list(g["pooled_classes"].to(gpu)), (256, 352))
    assert torch.equal(merged.pred_boxes.tensor.cpu(), g["boxes"]) and torch.equal(merged.scores.cpu(), g["scores"])
    assert torch.equal(merged.pred_classes.cpu(), g["classes"])
    # end to end
    assert len(out) == len(g["scores"]) == 100
    torch.testing.assert_close(out.scores.cpu(), g["scores"], rtol=1e-3, atol=1e-4)
    assert float((out.pred_classes.cpu() == g["classes"]).float().mean()) > 0.95


def test_parity_mode_rpn_step_matches_reference_golden(gpu, monkeypatch):
    """The shipped form (RPN branch on, g12) in the parity precision: the trunk runs on bf16x2 maps, the RPN head on the
    fp32 res5 map through the hi/lo operand split, pooling moves behind the RPN; losses against the reference's."""
    g = load_golden("g12_rpn_train_step")
    cfg, model, sd, sampling = build_rpn_model("parity")
    first_k_keys(model, monkeypatch)
    batch = gen.seeded_batch(2, 40, 20, 256, 352, seed=43)
    losses = model(to_inputs(batch))
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    torch.testing.assert_close(model.proposal_generator.pred_objectness_logits[0].detach().cpu(), g["rpn_logits"], rtol=1e-3,
                               atol=1e-3)
    for k in ("loss_cls_object_mining", "loss_cls_r0", "loss_box_reg_r0"):
        torch.testing.assert_close(losses[k].detach().cpu(), g["loss/" + k], rtol=1e-2, atol=1e-5, msg=lambda m: f"{k}: {m}")
    targets = model.roi_heads.proposal_targets
    for i, t in enumerate(targets):  # the pseudo ground truth the RPN is trained on: same proposals picked, boxes to 0.1 px
        assert torch.equal(t.gt_classes.cpu(), g[f"target{i}/gt_classes"])
        torch.testing.assert_close(t.gt_boxes.tensor.cpu(), g[f"target{i}/gt_boxes"], rtol=0, atol=0.1)
    pg = model.proposal_generator
    if torch.equal(pg.sampled_labels.cpu(), g["anchor_labels"]):
        for k in ("loss_rpn_cls", "loss_rpn_loc"):
            torch.testing.assert_close(losses[k].detach().cpu(), g["loss/" + k], rtol=1e-2, atol=1e-5, msg=lambda m: f"{k}: {m}")
    else:
        # Matcher(allow_low_quality_matches=True) makes EVERY anchor that ties for a pseudo-GT box's best IoU positive: a
        # small box lies inside dozens of equal-area anchors, and whether that tie set or one partly overlapping anchor is
        # "best" flips with a 0.02-px change of the box (this batch: 44 vs 77 positives).  The labels are then compared with
        # the oracle's matcher on the model's OWN pseudo-GT boxes (exact) and the losses with the oracle's on those labels.
        ref, ref_labels = R.rpn_losses(pg.anchors[0].tensor.cpu(), pg.pred_objectness_logits[0].detach().float().cpu(),
                                       pg.pred_anchor_deltas[0].detach().float().cpu(),
                                       [dict(gt_boxes=t.gt_boxes.tensor.cpu()) for t in targets], gen.first_k_subsample,
                                       thresholds=(0.2, 0.6))
        assert torch.equal(pg.sampled_labels.cpu(), ref_labels)
        for k in ("loss_rpn_cls", "loss_rpn_loc"):
            torch.testing.assert_close(losses[k].detach().cpu(), ref[k], rtol=1e-4, atol=1e-6, msg=lambda m: f"{k}: {m}")
    for k, q in model.named_parameters():
        if q.requires_grad:
            assert q.grad is not None and bool(torch.isfinite(q.grad).all()), k
