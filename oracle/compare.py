"""ORACLE -- test infrastructure only (never imported by wsovod_amd/).

One training step of the CPU restatement (`oracle/wsovod_ref.py:train_forward`, pinned to the reference's golden
vectors by tests/test_oracle_golden.py) at ANY size, packed the way the GPU capture of
`wsovod_amd/testing.py:capture_full_step` is packed, and the comparison of the two.  Used by the `-m gpu` full-size
parity tests (tests/test_gpu_full_size.py) and by bench.py's `parity` block (checker, never the thing measured).

Reference path followed: wsovod/modeling/meta_arch/rcnn_wsovod.py:137-234 -> roi_heads/roi_heads.py:696-907 ->
fast_rcnn_open_vocabulary.py:318-367,726-820.
"""
import torch

from . import wsovod_ref as R


def oracle_step(state_dict, host_batch, train_keys, *, depth=18, num_classes=20, pooler_type="ROIPool", classifier=None,
                miner_prefix=None):
    """state_dict: CPU fp32 tensors under the reference's key names; host_batch: DatasetMapper-format dicts on the
    host; train_keys: names of the trainable tensors (their gradients are returned).
    -> dict(losses, mining_scores, refine_logits, refine_deltas, img_scores, gt_classes, gt_boxes, gt_weights,
            pgt_boxes, pgt_classes, pgt_num, grads {name: tensor})."""
    sd = {k: v.detach().float().cpu().clone() for k, v in state_dict.items()}
    for k in train_keys:
        sd[k].requires_grad_(True)
    extra = {} if miner_prefix is None else {"miner_prefix": miner_prefix}  # mixed-dataset models: the active miner
    losses, inter = R.train_forward(sd, R.batch_from_inputs(host_batch), depth=depth, num_classes=num_classes,
                                    pooler_type=pooler_type, classifier=classifier, **extra)
    grads = torch.autograd.grad(sum(losses.values()), [sd[k] for k in train_keys], allow_unused=True)
    lab, tg = inter["labelled"], inter["targets"]
    return {
        "losses": {k: float(v.detach()) for k, v in losses.items()},
        "mining_scores": inter["mining_scores"].detach(),
        "refine_logits": inter["refine_logits"].detach(),
        "refine_deltas": inter["refine_deltas"].detach(),
        "img_scores": inter["pred_class_img_logits"].detach(),
        "res5": inter["res5"].detach(),
        "box_features": inter["box_features"].detach(),
        "gt_classes": torch.cat([l["gt_classes"] for l in lab]),
        "gt_boxes": torch.cat([l["gt_boxes"] for l in lab]),
        "gt_weights": torch.cat([l["gt_weights"] for l in lab]),
        "pgt_boxes": torch.cat([t["gt_boxes"] for t in tg]),
        "pgt_classes": torch.cat([t["gt_classes"] for t in tg]),
        "pgt_num": [int(t["gt_classes"].numel()) for t in tg],
        "grads": {k: (g.detach() if g is not None else None) for k, g in zip(train_keys, grads)},
    }


def compare(got, want):
    """got: capture_full_step(...) moved to the host; want: oracle_step(...).  -> report of plain floats / bools."""
    rep = {
        "max_abs_logit_err": float((got["refine_logits"] - want["refine_logits"]).abs().max()),
        "max_abs_score_err": float((got["mining_scores"] - want["mining_scores"]).abs().max()),
        "max_abs_delta_err": float((got["refine_deltas"] - want["refine_deltas"]).abs().max()),
        "max_abs_img_score_err": float((got["img_scores"] - want["img_scores"]).abs().max()),
        "max_rel_loss_err": max(abs(got["losses"][k] - want["losses"][k]) / max(abs(want["losses"][k]), 1e-12)
                                for k in want["losses"]),
        "labels_exact": bool(torch.equal(got["gt_classes"], want["gt_classes"])),
        "label_boxes_exact": bool(torch.equal(got["gt_boxes"], want["gt_boxes"])),
        "pgt_exact": bool(got["pgt_num"] == want["pgt_num"] and torch.equal(got["pgt_boxes"], want["pgt_boxes"])
                          and torch.equal(got["pgt_classes"], want["pgt_classes"])),
        "max_rel_weight_err": float(((got["gt_weights"] - want["gt_weights"]).abs()
                                     / want["gt_weights"].abs().clamp(min=1e-6)).max()),
    }
    # gradient norms: relative, with an absolute floor of 1e-5 x the step's largest norm -- a tensor whose true gradient
    # is zero (the bias of `det`: softmax over the proposals is shift invariant) carries only rounding noise
    worst, worst_key = 0.0, None
    norms = {k: float(g.double().norm()) for k, g in want["grads"].items() if g is not None}  # fp64 sums (103 M elements)
    top = max(norms.values())
    for k, nk in norms.items():
        e = abs(float(got["grad_norms"][k]) - nk) / max(nk, 1e-5 * top, 1e-12)
        if e > worst:
            worst, worst_key = e, k
    rep["max_rel_gradnorm_err"], rep["worst_grad"] = worst, worst_key
    # element-wise, for the tensors the capture kept: max |g - g_ref| over the tensor / max |g_ref| over the tensor
    elem, elem_key = 0.0, None
    for k, g in (got.get("grads") or {}).items():
        w = want["grads"].get(k)
        if w is None:
            continue
        e = float((g.double() - w.double()).abs().max()) / max(float(w.double().abs().max()), 1e-5 * top / max(w.numel(), 1) ** 0.5,
                                                                 1e-30)
        if e > elem:
            elem, elem_key = e, k
    if got.get("grads"):
        rep["max_rel_grad_elem_err"], rep["worst_grad_elem"] = elem, elem_key
    rep["meets_1e-3_logit_bound"] = bool(rep["max_abs_logit_err"] < 1e-3 and rep["max_abs_score_err"] < 1e-3)
    return rep
