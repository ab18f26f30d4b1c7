"""Which contractions decide the north star's 1e-3 logit bound?  

    python tools/precision_table.py [n_images] > profiles/r03_precision_table.json

Runs the "parity" precision (bf16x2 activations, three-MFMA products) at the headline size against the oracle with the
split switched OFF for exactly one layer group at a time, and ON for exactly one group at a time (all others plain
bf16 products), plus all-on / all-off.  A group is "off" when its contractions use only the hi halves of both operands
(the lo halves are zeroed on copies before the launch): exactly the products of the plain bf16 mode, measured on the
same kernels, weights and images.  Forward pass only (the bound is on forward quantities).

Groups: stem+res2-3 | res4-5 | fc1 | fc2 | projection+cos-sim | cls/det/bbox.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import wsovod_ref as R
from wsovod_amd.data import make_batch
from wsovod_amd.layers import hip_ops as H
from wsovod_amd.testing import build_hot_path_model

GROUPS = ["stem+res2-3", "res4-5", "fc1", "fc2", "projection+cos-sim", "cls/det/bbox"]
OFF = set()  # groups whose contractions run as plain bf16 products
_real_gemm, _real_stem = H.gemm_nt, H.stem_conv1_x2


def drop_lo(t):
    """copy of a bf16x2 tensor with the lo halves zeroed: hi + 0 = the bf16 rounding of the value."""
    c = t.detach().clone()
    v = c.view(torch.bfloat16).view(-1, 2, 32)
    v[:, 1, :] = 0
    return c


def group_of(A, B, conv, K1):
    if conv is not None:
        return "stem+res2-3" if B.size(0) <= 128 else "res4-5"
    N, K = B.shape
    if K >= 20000:
        return "fc1"
    if N == 4096 and K == 4096:
        return "fc2"
    if N in (K1 - 1, 2 * (K1 - 1), 4) and K == 4096:
        return "cls/det/bbox"
    return "projection+cos-sim"


def gemm_nt(A, B, *a, **kw):
    if kw.get("x2") and group_of(A, B, kw.get("conv"), gemm_nt.K1) in OFF:
        A, B = drop_lo(A), drop_lo(B)
        if kw.get("A2") is not None:
            kw["A2"] = drop_lo(kw["A2"])
    return _real_gemm(A, B, *a, **kw)


def stem_conv1_x2(images_u8, sizes, mean, std, w32_x2, bias):
    if "stem+res2-3" in OFF:  # the plain bf16 stem kernel on the bf16 rounding of the weight, re-encoded
        w32 = H.x2_decode(w32_x2).to(torch.bfloat16).contiguous()
        out = H.stem_conv1(images_u8, sizes, mean, std, w32, bias)
        n, ho, wo, c = out.shape
        return H.x2_encode(out.float().view(-1, c)).view(n, ho, wo, c)
    return _real_stem(images_u8, sizes, mean, std, w32_x2, bias)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    K = 20
    gemm_nt.K1 = K + 1
    H.gemm_nt, H.stem_conv1_x2 = gemm_nt, stem_conv1_x2
    host = make_batch(n, 512, K, seed=4321)
    cfg, model = build_hot_path_model(seed=0, K=K, precision="parity", device="cuda:0")
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        _, inter = R.train_forward(sd, R.batch_from_inputs(host), depth=18, num_classes=K)
    want_logits, want_scores = inter["refine_logits"], inter["mining_scores"]
    batch = [{"image": x["image"].cuda(), "proposals": x["proposals"].to("cuda"), "instances": x["instances"],
              "height": x["height"], "width": x["width"]} for x in host]
    cap = {}
    rh = model.roi_heads
    om, rf = rh.object_miner.forward, rh.box_refinery[0].forward
    rh.object_miner.forward = lambda *a, **k: cap.__setitem__("m", om(*a, **k)) or cap["m"]
    rh.box_refinery[0].forward = lambda *a, **k: cap.__setitem__("r", rf(*a, **k)) or cap["r"]

    def run(off):
        OFF.clear()
        OFF.update(off)
        for t in list(model.parameters()) + list(model.buffers()):  # weight caches (folded / encoded) are value-keyed: fine
            pass
        with torch.no_grad():
            model(batch)
        torch.cuda.synchronize()
        return {"max_abs_logit_err": float((cap["r"][0].float().cpu() - want_logits).abs().max()),
                "max_abs_score_err": float((cap["m"][0].float().cpu() - want_scores).abs().max())}

    out = {"workload": f"{n} x 800x600 images x 512 proposals, WSR_18, K = {K}, forward pass, vs oracle/wsovod_ref.py",
           "bound": 1e-3, "all_split (parity)": run([]), "none_split (bf16 products everywhere)": run(GROUPS),
           "split_off_for_one_group": {}, "split_on_for_one_group": {}}
    for g in GROUPS:
        out["split_off_for_one_group"][g] = run([g])
        out["split_on_for_one_group"][g] = run([x for x in GROUPS if x != g])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
