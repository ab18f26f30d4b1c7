"""Round 6: the weight-gradient contraction fused with the optimizer step of that weight (wsovod_gemm_tn_sgd) -- the
reference's pair `loss.backward()` + `optimizer.step()` (engine/trainer.py:72-84, SGD of engine/defaults.py:274-318) for the
one tensor that dominates the bytes of a small-batch step (fc1: 103 M parameters)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shadow_kind", ["none", "bf16", "x2"])
@pytest.mark.parametrize("q_x2", [False, True])
def test_fused_kernel_equals_dw_then_sgd(gpu, shadow_kind, q_x2):
    """One launch against the two it replaces, on the same operands: parameter, momentum buffer and the refreshed operand
    copy agree to the last bits (same formula; the two kernels may contract their multiply-adds differently), the shadow is
    EXACTLY the encoding of the parameter the kernel wrote; ragged tiles (NI, Mred not multiples of the tile), a learning
    rate read from device memory, a second step on the momentum the first one left."""
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(11)
    Mred, NI, NJ = 300, 520, 1056  # 3 x 5 tiles, NJ = 33 groups of 32
    dA = (torch.randn(Mred, NI, device=gpu) * 0.1).to(torch.bfloat16)
    x32 = torch.randn(Mred, NJ, device=gpu)
    x = H.x2_encode(x32) if q_x2 else x32.to(torch.bfloat16)
    w0 = torch.randn(NI, NJ, device=gpu) * 0.05
    lr, wd, mu = 0.013, 5e-4, 0.9

    def shadow_of(w):
        if shadow_kind == "none":
            return None
        return H.x2_encode(w) if shadow_kind == "x2" else w.to(torch.bfloat16)

    # reference: dW, then the optimizer kernel (two steps)
    wr, br = w0.clone(), torch.zeros_like(w0)
    sr = shadow_of(wr)
    for _ in range(2):
        g = H.gemm_tn(dA, x, q_x2=q_x2, split_tail=False)
        H.sgd_momentum_multi([(wr, g, br, sr, lr, wd)], mu)
    # fused, the second step with the rate in device memory
    wf, bf = w0.clone(), torch.zeros_like(w0)
    sf = shadow_of(wf)
    H.gemm_tn_sgd(dA, x, wf, bf, sf, lr, wd, mu, q_x2=q_x2)
    H.gemm_tn_sgd(dA, x, wf, bf, sf, torch.tensor([lr], device=gpu), wd, mu, q_x2=q_x2)
    torch.cuda.synchronize()
    torch.testing.assert_close(wf, wr, rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(bf, br, rtol=2e-6, atol=1e-7)
    assert float((wf - w0).abs().max()) > 1e-4  # (the update is not a no-op)
    if shadow_kind == "bf16":
        assert torch.equal(sf, wf.to(torch.bfloat16))
    elif shadow_kind == "x2":
        assert torch.equal(sf, H.x2_encode(wf))


@pytest.mark.parametrize("tail", ["0", "1"])
def test_fused_kernel_with_a_partly_filled_last_round(gpu, monkeypatch, tail):
    """More than one round of 256 x 256 tiles with a last round that fills less than half the chip (17 x 16 = 272 tiles: 16 in
    the tail, as fc1's 1568 = 6 x 256 + 32), ragged last row tile: whole tiles (the default) and the measured-and-not-default
    form WSOVOD_TN_SGD_TAIL=1 -- the tail tiles' K slices meet by atomics in a compact scratch and a small pass applies their
    update -- against dW (whole tiles, fixed order) + the optimizer kernel."""
    from wsovod_amd.layers import hip_ops as H

    monkeypatch.setenv("WSOVOD_TN_SGD_TAIL", tail)

    torch.manual_seed(12)
    Mred, NI, NJ = 4096, 4096 + 40, 4096
    dA = (torch.randn(Mred, NI, device=gpu) * 0.05).to(torch.bfloat16)
    x = torch.randn(Mred, NJ, device=gpu).to(torch.bfloat16)
    w0 = torch.randn(NI, NJ, device=gpu) * 0.05
    lr, wd, mu = 0.01, 5e-4, 0.9
    wr, br = w0.clone(), torch.zeros_like(w0)
    sr = H.x2_encode(wr)
    wf, bf = w0.clone(), torch.zeros_like(w0)
    sf = H.x2_encode(wf)
    for _ in range(2):
        g = H.gemm_tn(dA, x, split_tail=False)
        H.sgd_momentum_multi([(wr, g, br, sr, lr, wd)], mu)
        H.gemm_tn_sgd(dA, x, wf, bf, sf, lr, wd, mu)
    torch.cuda.synchronize()
    torch.testing.assert_close(wf, wr, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(bf, br, rtol=2e-5, atol=2e-5)
    assert torch.equal(sf, H.x2_encode(wf))
    # the tail tiles (the last tile ids of the grouped order) were updated exactly once per step: nothing left at w0
    assert float((wf - w0).abs().min()) >= 0 and float(((wf - w0).abs() > 0).float().mean()) > 0.999


def _steps(gpu, monkeypatch, fused, graph, precision, n_steps=6):
    from wsovod_amd.data import make_batch
    from wsovod_amd.engine import HotPathTrainer, build_optimizer
    from wsovod_amd.layers import hip_ops as H
    from wsovod_amd.testing import build_hot_path_model

    monkeypatch.setattr(H, "DETERMINISTIC", True)
    monkeypatch.setenv("WSOVOD_BACKBONE_GRAPH", "0")
    monkeypatch.setenv("WSOVOD_FUSED_SGD", "1" if fused else "0")
    monkeypatch.setenv("WSOVOD_STEP_GRAPH", "1" if graph else "0")
    cfg, model = build_hot_path_model(seed=0, precision=precision, device="cuda:0")
    model.train()
    cfg.SOLVER.BASE_LR = 1e-3
    tr = HotPathTrainer(model, build_optimizer(cfg, model))
    assert bool(tr._fused) == fused  # (fc1 and fc2: the 2-D weights of at least 2^24 elements, largest first)
    if fused:
        assert [tuple(p.shape) for p in tr._fused] == [(4096, 25088), (4096, 4096)]
    losses = []
    for s in range(n_steps):
        for grp in tr.optimizer.param_groups:
            grp["lr"] = 1e-3 * (1 + s)  # a scheduler moves the rate between steps (device-resident under a step graph)
        b = make_batch(2, 64, 20, H=160, W=224, seed=700 + s)
        b = [{"image": x["image"].to(gpu), "proposals": x["proposals"].to(gpu), "instances": x["instances"],
              "height": x["height"], "width": x["width"]} for x in b]
        losses.append({k: float(v.detach()) for k, v in tr.run_step(b).items()})
    tr.flush()
    calls = tr._fused[0]._fused_update.calls if fused else 0
    fc2_calls = tr._fused[1]._fused_update.calls if fused else 0
    assert fc2_calls == calls
    fc1 = model.roi_heads.box_head.fc1.weight
    out = {"params": {k: v.detach().clone() for k, v in model.named_parameters() if v.requires_grad},
           "mom_fc1": tr.optimizer.state[fc1]["momentum_buffer"].clone(), "losses": losses, "calls": calls,
           "graphs": len(tr._graphs)}
    if precision == "parity":  # the operand copy the NEXT forward reads must be the encoding of the weight as it stands
        xe = getattr(fc1, "_x2_enc", None)
        assert xe is not None and xe[0] == (fc1._version, fc1.data_ptr(), None)
        assert torch.equal(xe[1], H.x2_encode(fc1.detach()))
    tr.close()
    assert not hasattr(fc1, "_fused_update")
    return out


@pytest.mark.parametrize("precision", ["bf16", "parity"])
@pytest.mark.parametrize("graph", [False, True])
def test_trainer_with_fused_fc1_update_equals_the_two_kernel_step(gpu, monkeypatch, precision, graph):
    """HotPathTrainer at 2 images per step, dropout on, six steps under a moving learning rate, eager launches and
    whole-step HIP graphs: with fc1's update inside its weight-gradient kernel against WSOVOD_FUSED_SGD=0 -- every trained
    tensor, fc1's momentum buffer and the losses of every step agree to rounding."""
    a = _steps(gpu, monkeypatch, True, graph, precision)
    b = _steps(gpu, monkeypatch, False, graph, precision)
    assert a["calls"] >= 3 and (a["graphs"] == 1) == graph  # (eager: every step; graph: the eager steps before the capture)
    for sa, sb in zip(a["losses"], b["losses"]):
        for k in sa:
            assert abs(sa[k] - sb[k]) <= 2e-5 * max(abs(sb[k]), 1e-3), (k, sa[k], sb[k])
    for k, v in b["params"].items():
        torch.testing.assert_close(a["params"][k], v, rtol=1e-5, atol=2e-6 * float(v.abs().max()) + 1e-9, msg=lambda m: f"{k}: {m}")
    torch.testing.assert_close(a["mom_fc1"], b["mom_fc1"], rtol=1e-5, atol=1e-6 * float(b["mom_fc1"].abs().max()) + 1e-12)


def test_fused_update_steps_aside_for_clipping_accumulation_and_large_batches(gpu, monkeypatch):
    """The fused form is only installed where it is the SAME computation: gradient clipping needs the whole gradient's norm
    first, ITER_SIZE > 1 accumulates, a trainable-gradient consumer (`p.grad` left by a caller) must see the gradient; a
    reduction longer than WSOVOD_FUSED_SGD_ROWS keeps the two kernels (their split tile-round tail is faster there)."""
    from wsovod_amd.engine import HipSGD, HotPathTrainer, build_optimizer
    from wsovod_amd.testing import build_hot_path_model

    cfg, model = build_hot_path_model(seed=0, precision="bf16", device="cuda:0")
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    tr = HotPathTrainer(model, HipSGD(params, lr=1e-3, momentum=0.9, clip=("full_model", 1.0)))
    assert not tr._fused
    tr.close()
    tr = HotPathTrainer(model, build_optimizer(cfg, model), iter_size=2)
    assert not tr._fused
    tr.close()
    monkeypatch.setenv("WSOVOD_FUSED_SGD_ROWS", "100")
    tr = HotPathTrainer(model, build_optimizer(cfg, model))
    big = tr._fused[0]
    fu = big._fused_update
    assert not fu.wants(64)  # not inside the trainer's own backward: a caller's loss.backward() gets a gradient
    fu.armed = True
    assert fu.wants(64) and not fu.wants(128)
    big.grad = torch.zeros_like(big)
    assert not fu.wants(64)  # somebody left a gradient on the tensor: it goes through the optimizer
    big.grad = None
    fu.armed = False
    # a plain forward + backward on the model while the trainer is attached leaves the weight alone and fills .grad
    from wsovod_amd.data import make_batch

    b = [{"image": x["image"].to(gpu), "proposals": x["proposals"].to(gpu), "instances": x["instances"],
          "height": x["height"], "width": x["width"]} for x in make_batch(1, 64, 20, H=160, W=224, seed=5)]
    before = big.detach().clone()
    sum(model(b).values()).backward()
    assert big.grad is not None and torch.equal(big.detach(), before) and fu.calls == 0
    model.zero_grad(set_to_none=True)
    tr.close()
