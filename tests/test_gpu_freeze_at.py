"""GPU: MODEL.BACKBONE.FREEZE_AT < 5 (reference: backbone/resnet_wsl.py:530-552).  The trainable stages run their FORWARD
on the HIP kernels; their backward (`modeling/backbone.py:_TrainableStage`) too since round 6 for the stages without a tail
pool (res4 / res5: mask passes, input gradients as implicit-GEMM convs on the rotated weights, weight gradients as the
transposed-read contraction over im2col rows); res2 / res3 keep the torch-autograd re-evaluation (MIOpen).  Pinned to the
REFERENCE's own step at FREEZE_AT = 4 (tests/golden/g19_freeze_at_4.npz, make_golden.py:golden_trainable_stage): the
gradient runs through the HIP RoIPool backward (argmax scatter) and the GAP of the data-aware head into res5."""
import numpy as np
import pytest
import torch

from tests.golden import gen
from tests.helpers import load_golden, seeded_sd, to_inputs

pytestmark = pytest.mark.gpu


def _model(precision, freeze_at):
    from wsovod_amd.testing import build_hot_path_model

    cfg, model = build_hot_path_model(seed=0, precision=precision, device="cuda:0", calibrate_synthetic=False,
                                      freeze_at=freeze_at)
    cfg.MODEL.PIXEL_STD = list(gen.PIXEL_STD)
    model._std = [float(v) for v in gen.PIXEL_STD]
    model.load_state_dict(seeded_sd(1), strict=True)
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    return cfg, model


@pytest.mark.parametrize("precision", ["fp32", "parity", "bf16"])
def test_res5_trainable_step_matches_reference_golden(gpu, precision):
    g = load_golden("g19_freeze_at_4")
    cfg, model = _model(precision, 4)
    train = [k for k, p in model.named_parameters() if p.requires_grad]
    assert sorted(train) == sorted(str(k) for k in g["train_keys"])  # the same tensors are trainable as in the reference
    assert model.backbone.has_trainable_stage
    captured = {}
    rh = model.roi_heads
    om, rf = rh.object_miner.forward, rh.box_refinery[0].forward
    rh.object_miner.forward = lambda *a, **k: captured.setdefault("miner", om(*a, **k))
    rh.box_refinery[0].forward = lambda *a, **k: captured.setdefault("refine", rf(*a, **k))
    losses = model(to_inputs(gen.seeded_batch(2, 24, 20, 160, 208, seed=11)))
    rh.object_miner.forward, rh.box_refinery[0].forward = om, rf
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    scores, logits = captured["miner"][0].detach().float().cpu(), captured["refine"][0].detach().float().cpu()
    if precision != "bf16":
        assert (scores - g["mining_scores"]).abs().max() < 1e-3 and (logits - g["refine_logits"]).abs().max() < 1e-3
        assert torch.equal(rh._last_pgt["gt_classes"].cpu(), g["label/gt_classes"])
        for k in ("loss_cls_object_mining", "loss_cls_r0", "loss_box_reg_r0"):
            torch.testing.assert_close(losses[k].detach().cpu(), g["loss/" + k], rtol=1e-3, atol=1e-5)
    # gradients: the heads as with a frozen backbone; the res5 convs through pooling backward + the re-evaluated stage
    rtol = {"fp32": 3e-3, "parity": 2e-2, "bf16": 0.15}[precision]
    for k, p in model.named_parameters():
        if not p.requires_grad:
            assert p.grad is None, k
            continue
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k
        gr = p.grad.detach().float().cpu()
        torch.testing.assert_close(gr.norm(), g["gradnorm/" + k], rtol=rtol, atol=1e-7, msg=lambda m: f"{k}: {m}")
        if precision == "fp32":
            ref = g["gradsample/" + k]
            assert (gen.strided_sample(gr, 1024) - ref).abs().max() <= 5e-3 * ref.abs().max() + 1e-8, k


@pytest.mark.parametrize("precision,freeze_at,tol", [("fp32", 3, 2e-3), ("parity", 3, 2e-2), ("bf16", 4, 0.1),
                                                     ("fp32", 1, 2e-3), ("parity", 1, 3e-2), ("bf16", 2, 0.15)])
def test_hip_conv_backward_agrees_with_the_torch_re_evaluation(gpu, monkeypatch, precision, freeze_at, tol):
    """Round 6: the HIP dgrad / wgrad path of the trainable stages against the torch (MIOpen, fp32) re-evaluation it replaces
    (WSOVOD_HIP_CONV_BACKWARD=0), same model, same batch: every trainable tensor's gradient -- res4 AND res5 at
    FREEZE_AT = 3, i.e. the input gradient crosses a stage boundary, blocks with a projection shortcut fused into their last
    conv and blocks with an identity shortcut; at FREEZE_AT = 1 / 2 also the stages with a tail pool (res2: stride 2, fused
    into the 64-channel conv in the forward pass; res3: ZeroPad2d + stride 1) through the pool-backward kernel -- agrees in
    norm and element-wise to the precision's backward grade."""
    from wsovod_amd.modeling import backbone as BB

    batch = to_inputs(gen.seeded_batch(2, 24, 20, 160, 208, seed=13))
    if freeze_at == 1:  # the weight gradients in several row blocks of patch rows (accumulated), as at large batches
        monkeypatch.setattr(BB, "WGRAD_PATCH_BYTES", 1 << 20)
    grads = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("WSOVOD_HIP_CONV_BACKWARD", flag)
        cfg, model = _model(precision, freeze_at)
        sum(model(batch).values()).backward()
        torch.cuda.synchronize()
        grads[flag] = {k: p.grad.detach().float().clone() for k, p in model.named_parameters() if p.requires_grad}
        del model
    bb = [k for k in grads["1"] if k.startswith("backbone.")]
    # R18: res2 = 2 blocks x 2 convs; res3 / res4 / res5 = 2 blocks (4 convs + 1 projection shortcut) each
    assert len(bb) == {1: 19, 2: 15, 3: 10, 4: 5}[freeze_at]
    for k in grads["1"]:
        a, b = grads["1"][k], grads["0"][k]
        assert bool(torch.isfinite(a).all()), k
        assert abs(float(a.norm()) - float(b.norm())) <= tol * float(b.norm()) + 1e-9, (k, float(a.norm()), float(b.norm()))
        assert float((a - b).abs().max()) <= 4 * tol * float(b.abs().max()) + 1e-9, k


def test_trainer_and_optimizer_with_a_trainable_stage(gpu):
    """HotPathTrainer + HipSGD at FREEZE_AT = 3 (res4 and res5 trainable: the gradient crosses a bf16x2 stage boundary in
    the headline precision): the backbone runs BEHIND the pending update (no frozen-forward overlap, no step graph), the
    loss goes down on a fixed batch, res3 stays untouched, inference afterwards sees the trained stage."""
    from wsovod_amd.data import make_batch
    from wsovod_amd.engine import HotPathTrainer, build_optimizer

    from wsovod_amd.testing import build_hot_path_model

    cfg, model = build_hot_path_model(seed=0, precision="parity", device="cuda:0", freeze_at=3)
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    cfg.SOLVER.BASE_LR = 2e-4
    frozen = {k: v.detach().clone() for k, v in model.named_parameters() if k.startswith("backbone.res3.")}
    w0 = model.backbone.res4[0].conv1.weight.detach().clone()
    tr = HotPathTrainer(model, build_optimizer(cfg, model))
    batch = make_batch(2, 48, 20, H=192, W=256, seed=5)
    hist = [sum(float(v) for v in tr.run_step(batch).values()) for _ in range(8)]
    tr.flush()
    assert not tr._graphs and all(np.isfinite(hist)) and hist[-1] < hist[0], hist
    assert not torch.equal(model.backbone.res4[0].conv1.weight.detach(), w0)
    assert all(torch.equal(dict(model.named_parameters())[k].detach(), v) for k, v in frozen.items())
    model.eval()
    out = model.inference(make_batch(1, 32, 20, H=192, W=256, seed=6))
    assert len(out) == 1 and "instances" in out[0]
    tr.close()


def test_trainable_stem_matches_the_oracle(gpu):
    """MODEL.BACKBONE.FREEZE_AT = 0 (round 6; reference: resnet_wsl.py:530-552 with nothing frozen): the whole backbone
    trains -- the stem's conv1 is fused with the uint8 normalisation in the forward pass and takes its weight gradient from
    the normalised im2col rows of the image.  fp32 against the ORACLE's autograd (oracle/wsovod_ref.py with the gradient
    running through its differentiable RoIPool into res5 ... stem.conv1): every trainable tensor's gradient norm within 3e-3,
    sampled elements within 1 %; the parity precision against the fp32 run at its backward grade."""
    from oracle import wsovod_ref as R

    host = gen.seeded_batch(2, 16, 20, 96, 128, seed=21)
    grads = {}
    for precision in ("fp32", "parity"):
        cfg, model = _model(precision, 0)
        train = [k for k, p in model.named_parameters() if p.requires_grad]
        assert "backbone.stem.conv1.weight" in train and len([k for k in train if k.startswith("backbone.")]) == 22
        sum(model(to_inputs(host)).values()).backward()
        torch.cuda.synchronize()
        grads[precision] = {k: p.grad.detach().float().cpu() for k, p in model.named_parameters() if p.requires_grad}
        if precision == "fp32":
            sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
        del model
    for k in train:
        sd[k].requires_grad_(True)
    losses, _ = R.train_forward(sd, host, depth=18, num_classes=20, pixel_std=tuple(gen.PIXEL_STD),
                                backbone_grad=True)
    want = dict(zip(train, torch.autograd.grad(sum(losses.values()), [sd[k] for k in train], allow_unused=True)))
    top = max(float(g.norm()) for g in want.values() if g is not None)
    for k in train:
        w, g32, gp = want[k], grads["fp32"][k], grads["parity"][k]
        if w is None or float(w.norm()) < 1e-6 * top:
            continue
        assert abs(float(g32.norm()) - float(w.norm())) <= 3e-3 * float(w.norm()), (k, float(g32.norm()), float(w.norm()))
        assert float((g32 - w).abs().max()) <= 1e-2 * float(w.abs().max()) + 1e-9, k
        assert abs(float(gp.norm()) - float(w.norm())) <= 3e-2 * float(w.norm()), (k, float(gp.norm()), float(w.norm()))


def test_float_entry_with_a_trainable_stem_is_refused(gpu):
    """ResNet.forward(x) (the generic float entry) has no trainable-stem path; it says so instead of silently freezing."""
    cfg, model = _model("fp32", 0)
    with pytest.raises(NotImplementedError, match="FREEZE_AT = 0"):
        model.backbone(torch.randn(1, 3, 64, 96, device=gpu))


@pytest.mark.parametrize("stride,pad", [(2, False), (1, True)])
@pytest.mark.parametrize("fmt", ["fp32", "bf16", "x2"])
def test_maxpool_backward_routes_to_torchs_first_maximum(gpu, stride, pad, fmt):
    """wsovod_maxpool2x2_nhwc_backward against torch's max_pool2d backward (resnet_wsl.py:85-92 under autograd) on maps with
    MANY ties (post-ReLU zeros; quantised values): the gradient goes to the first maximum in scan order, the zero cells of
    ZeroPad2d((0,1,0,1)) take part and swallow their share; odd sizes (a last row / column no stride-2 window covers)."""
    import torch.nn.functional as F
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(3)
    N, Hh, Ww, C = 2, 13, 11, 64
    x = torch.relu(torch.randn(N, Hh, Ww, C, device=gpu)).mul(4).round().div(4)  # ties and zeros
    if fmt == "bf16":
        xin = x.to(torch.bfloat16)
    elif fmt == "x2":
        xin = H.x2_encode(x.view(-1, C)).view(N, Hh, Ww, C)
    else:
        xin = x
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    y = F.max_pool2d(F.pad(xr, (0, 1, 0, 1)), 2, 1) if pad else F.max_pool2d(xr, 2, stride)
    dy = torch.randn_like(y)
    y.backward(dy)
    want = xr.grad.permute(0, 2, 3, 1)
    got = H.maxpool2x2_nhwc_backward(xin.contiguous(), dy.permute(0, 2, 3, 1).contiguous(), stride, zero_pad_br=pad, x2=fmt == "x2")
    torch.testing.assert_close(got, want, rtol=0, atol=0)
    fwd = H.maxpool2x2_nhwc(xin.contiguous(), stride, zero_pad_br=pad, x2=fmt == "x2")
    fwd32 = H.x2_to_f32(fwd) if fmt == "x2" else fwd.float()
    torch.testing.assert_close(fwd32, y.detach().permute(0, 2, 3, 1), rtol=0, atol=0)
