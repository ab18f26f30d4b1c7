"""Captured HIP graphs at small batches (the reference's own 1 - 2 images per GPU are host-bound): the frozen backbone
alone (opt-in, `forward_uint8(allow_graph=True)`), and the WHOLE training step -- frozen forward, heads, backward, fused
SGD -- through HotPathTrainer.  Replays must reproduce the eager launches: bit for bit where the eager path itself is
bit-reproducible (the backbone, the dropout masks, every integer output), to the eager path's own run-to-run jitter
elsewhere (a few reductions meet by float atomics)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _inputs(n, H, W, seed, dev):
    g = torch.Generator().manual_seed(seed)
    return [{"image": torch.randint(0, 256, (3, H, W), dtype=torch.uint8, generator=g).to(dev)} for _ in range(n)]


@pytest.mark.parametrize("precision", ["bf16", "parity"])
def test_backbone_graph_replay_equals_eager(gpu, precision):
    from wsovod_amd.testing import build_hot_path_model

    cfg, model = build_hot_path_model(seed=0, precision=precision, device="cuda:0")
    bb = model.backbone
    batches = [_inputs(2, 160, 224, s, gpu) for s in (1, 2)]

    def run(inp):
        canvas, sizes_t, _ = model._canvas(inp)
        return bb.forward_uint8(canvas, sizes_t, model._mean, model._std, allow_graph=True)["res5"]

    eager = [run(b).clone() for b in batches]
    bb.graph_max_batch = 8
    canvas, sizes_t, _ = model._canvas(batches[0])  # without allow_graph (inference, TTA): always fresh eager tensors
    for _ in range(bb.GRAPH_AFTER + 1):
        assert torch.equal(bb.forward_uint8(canvas, sizes_t, model._mean, model._std)["res5"], eager[0])
    assert not bb.__dict__.get("_graphs") and not bb.__dict__.get("_graph_seen")
    try:
        for _ in range(bb.GRAPH_AFTER - 1):  # a shape is captured on its third call
            assert torch.equal(run(batches[0]), eager[0]) and not bb.__dict__.get("_graphs")
        got = [run(b).clone() for b in batches]
        again = run(batches[0])
        assert len(bb._graphs) == 1  # one shape, one graph; the second batch replayed it
        assert torch.equal(again, eager[0]) and all(torch.equal(a, b) for a, b in zip(got, eager))
        assert again.data_ptr() == run(batches[1]).data_ptr()  # the maps are the graph's static buffers
        # a weight change (load_state_dict, broadcast ...) drops the graphs: they point at the old folded copies
        with torch.no_grad():
            bb.stem.conv1.weight.mul_(1.25)
        changed = run(batches[0]).clone()
        assert len(bb._graphs) == 1  # (the shape had been seen: recaptured at once)
        bb.graph_max_batch = 0
        assert torch.equal(changed, run(batches[0])) and not torch.equal(changed, eager[0])
        # another shape -> another graph; a batch above the limit stays eager
        bb.graph_max_batch = 2
        other = _inputs(1, 128, 192, 3, gpu)
        for _ in range(bb.GRAPH_AFTER):
            a = run(other).clone()
        assert len(bb._graphs) == 2
        big = _inputs(3, 128, 192, 4, gpu)
        n_graphs = len(bb._graphs)
        b3 = run(big).clone()
        assert len(bb._graphs) == n_graphs
        bb.graph_max_batch = 0
        assert torch.equal(a, run(other)) and torch.equal(b3, run(big))
    finally:
        bb.graph_max_batch = 0
        bb.__dict__.pop("_graphs", None)


def test_trainer_steps_with_backbone_graph_equal_eager_steps(gpu, monkeypatch):
    """The overlapped trainer turns the graph on for batches of up to 8 images: same losses, step for step, as with
    WSOVOD_BACKBONE_GRAPH=0 (the backbone is frozen and its output bit-identical -- the test above; the rest of the
    step is unchanged)."""
    from wsovod_amd.data import make_batch
    from wsovod_amd.engine import HotPathTrainer, build_optimizer
    from wsovod_amd.testing import build_hot_path_model

    hist = {}
    monkeypatch.setenv("WSOVOD_STEP_GRAPH", "0")  # (the whole-step graph would take over: tested below)
    for flag in ("0", "1"):
        monkeypatch.setenv("WSOVOD_BACKBONE_GRAPH", flag)
        cfg, model = build_hot_path_model(seed=0, precision="bf16", device="cuda:0")
        model.train()
        cfg.SOLVER.BASE_LR = 1e-3
        tr = HotPathTrainer(model, build_optimizer(cfg, model))
        assert model.backbone.graph_max_batch == (8 if flag == "1" else 0)
        batch = make_batch(2, 64, 20, H=320, W=416, seed=3)
        hist[flag] = [{k: float(v.detach()) for k, v in tr.run_step(batch).items()} for _ in range(4)]
        tr.close()
        assert model.backbone.graph_max_batch == 0
    for a, b in zip(hist["0"], hist["1"]):  # up to the run-to-run jitter of the float atomics in the loss / gradient
        for k in a:                          # sums (two eager runs differ by the same last bits)
            assert abs(a[k] - b[k]) <= 1e-4 * max(abs(a[k]), 1e-3), (k, a[k], b[k])


@pytest.mark.parametrize("precision", ["bf16", "parity"])
@pytest.mark.parametrize("shape", [(300, 256), (512, 4096)])
def test_dropout_step_term_on_the_device_draws_the_host_seeds_mask(gpu, precision, shape):
    """The dropout seed of a layer is base + 16 * step: with the step term in device memory (`seed_add`, what a captured
    graph replays against) the kernels draw exactly the mask of the host-computed seed -- both GEMM tiles, both modes."""
    from wsovod_amd.layers import functions as Fn
    from wsovod_amd.layers import hip_ops as H

    M, N = shape
    g = torch.Generator().manual_seed(3)
    x = torch.randn(M, 512, generator=g).to(gpu)
    w = (torch.randn(N, 512, generator=g) * 0.05).to(gpu)
    b = torch.randn(N, generator=g).to(gpu)
    step = torch.tensor([16 * 7], dtype=torch.int64, device=gpu)
    base = 0x1234567 * 1000003 + 1
    with H.x3_mode("x2" if precision == "parity" else False):
        xin = H.x2_encode(x) if precision == "parity" else x.to(torch.bfloat16)
        od = H.X2 if precision == "parity" else None
        a = Fn.linear(xin, w, b, relu=True, dropout_p=0.5, seed=base + 16 * 7, out_dtype=od)
        c = Fn.linear(xin, w, b, relu=True, dropout_p=0.5, seed=base, out_dtype=od, seed_add=step)
        other = Fn.linear(xin, w, b, relu=True, dropout_p=0.5, seed=base + 16 * 8, out_dtype=od)
    assert torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a, c.view(torch.int32) if c.dtype == torch.float32 else c)
    assert not torch.equal(a, other)
    step.add_(16)
    with H.x3_mode("x2" if precision == "parity" else False):
        d = Fn.linear(xin, w, b, relu=True, dropout_p=0.5, seed=base, out_dtype=od, seed_add=step)
    assert torch.equal(d, other)


def _varying_batches(n_steps, nums_list, K=20, H=320, W=416):
    """One batch per step, every step other images / boxes / labels; the per-image proposal counts follow `nums_list`
    (same total: the layout a step graph is keyed on)."""
    from wsovod_amd.data import make_batch

    out = []
    for s in range(n_steps):
        nums = nums_list[s % len(nums_list)]
        big = make_batch(len(nums), max(nums), K, H=H, W=W, seed=100 + s)
        for x, n in zip(big, nums):
            x["proposals"] = x["proposals"][:n]
        out.append(big)
    return out


@pytest.mark.parametrize("precision", ["bf16", "parity", "parity_mx"])
def test_whole_step_graph_reproduces_the_eager_steps(gpu, monkeypatch, precision):
    """HotPathTrainer at 2 images per step, dropout ON, eight steps on eight different batches whose per-image proposal
    counts AND totals change from step to step (inside one row bucket): with the whole-step HIP graph (captured on the third step, replayed
    from then on) against WSOVOD_STEP_GRAPH=0.  Every integer output -- per-proposal labels, pseudo-GT indices -- is
    equal step by step, the losses agree to the eager path's own jitter (float atomics in a few reductions; weight-gradient
    tails kept in fixed order here), and so do the trained parameters: same dropout masks, same updates, same data."""
    from wsovod_amd.engine import HotPathTrainer, build_optimizer
    from wsovod_amd.engine.trainer import _StepGraph
    from wsovod_amd.layers import hip_ops as H
    from wsovod_amd.testing import build_hot_path_model

    monkeypatch.setattr(H, "DETERMINISTIC", True)
    monkeypatch.setenv("WSOVOD_BACKBONE_GRAPH", "0")
    if precision == "parity_mx":  # (two small images: below the mode's tile-count thresholds -- lowered, the f16mx kernels run)
        from wsovod_amd.modeling.backbone import ResNet
        from wsovod_amd.modeling.roi_heads import WSOVODROIHeads

        monkeypatch.setattr(ResNet, "MX_MIN_TILES", 1)
        monkeypatch.setattr(WSOVODROIHeads, "MX_MIN_ROWS", 1)
    # totals 128 / 120 / 113 / 97: one bucket of 128 rows (trainer.row_bucket) -- the graph runs the short steps with
    # padding rows behind the last image (an empty box, label -1, zero gradient, the box loss normalised by the real count)
    batches = _varying_batches(8, [(64, 64), (60, 60), (70, 43), (33, 64)])
    runs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("WSOVOD_STEP_GRAPH", flag)
        cfg, model = build_hot_path_model(seed=0, precision=precision, device="cuda:0")
        model.train()
        cfg.SOLVER.BASE_LR = 1e-3
        tr = HotPathTrainer(model, build_optimizer(cfg, model))
        hist = []
        for b in batches:
            losses = tr.run_step(b)
            pgt = model.roi_heads._last_pgt
            t = int(sum(len(torch.unique(x["instances"].gt_classes)) for x in b))
            rows = sum(len(x["proposals"]) for x in b)
            assert bool((pgt["gt_classes"][rows:] == -1).all()) and bool((pgt["gt_weights"][rows:] == 0).all())
            hist.append(({k: float(v.detach()) for k, v in losses.items()}, pgt["gt_classes"][:rows].cpu().clone(),
                         pgt["pgt_index"][:t].cpu().clone(), pgt["pgt_classes"][:t].cpu().clone()))
        if flag == "1":
            assert [type(g) for g in tr._graphs.values()] == [_StepGraph]  # one layout, one graph, five replays
            assert model.roi_heads.box_head._step == len(batches) and tr.iter == len(batches)
            assert int(model.roi_heads.box_head._step_dev) == 16 * len(batches)
        else:
            assert not tr._graphs
        tr.flush()
        runs[flag] = (hist, {k: v.detach().clone() for k, v in model.named_parameters() if v.requires_grad})
        tr.close()
    for s, (e, g) in enumerate(zip(runs["0"][0], runs["1"][0])):
        assert torch.equal(e[1], g[1]) and torch.equal(e[2], g[2]) and torch.equal(e[3], g[3]), s
        for k in e[0]:
            assert abs(e[0][k] - g[0][k]) <= 2e-5 * max(abs(e[0][k]), 1e-3), (s, k, e[0][k], g[0][k])
    for k, v in runs["0"][1].items():
        torch.testing.assert_close(runs["1"][1][k], v, rtol=1e-5, atol=2e-6 * float(v.abs().max()) + 1e-9, msg=lambda m: f"{k}: {m}")


@pytest.mark.parametrize("precision", ["bf16", "parity"])
def test_mixed_dataset_model_replays_one_graph_per_source(gpu, monkeypatch, precision):
    """BASELINE config 5's model (rcnn_wsovod_mixed_datasets.py:188-191,237-238): a batch's `dataset_id` picks the object
    miner, the class count and the text embeddings of the step.  The step graph is keyed on (source, layout): twelve steps
    alternating between a 20-class and an 80-class source (different miners, different K, dropout ON) with the graphs
    against WSOVOD_STEP_GRAPH=0 -- labels / pseudo-GT indices equal step by step, losses and trained parameters (BOTH miners,
    the shared neck and refinement head) to the eager path's jitter; the miner of the source that is NOT in the batch gets no
    update in either run."""
    from wsovod_amd.data import make_batch
    from wsovod_amd.engine import HotPathTrainer, build_optimizer
    from wsovod_amd.engine.trainer import _StepGraph
    from wsovod_amd.layers import hip_ops as H
    from wsovod_amd.testing import build_mixed_model

    monkeypatch.setattr(H, "DETERMINISTIC", True)
    monkeypatch.setenv("WSOVOD_BACKBONE_GRAPH", "0")
    Ks = (20, 20, 80)
    order = [0, 2, 0, 2, 2, 0, 0, 2, 0, 2, 2, 0]
    batches = []
    for s, src in enumerate(order):
        b = make_batch(2, 64, Ks[src], H=160, W=224, seed=300 + s)
        for x in b:
            x["dataset_id"] = src
        batches.append(b)
    runs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("WSOVOD_STEP_GRAPH", flag)
        cfg, model = build_mixed_model(seed=0, Ks=Ks, precision=precision, device="cuda:0")
        model.train()
        cfg.SOLVER.BASE_LR = 1e-3
        tr = HotPathTrainer(model, build_optimizer(cfg, model))
        hist = []
        for b, src in zip(batches, order):
            losses = tr.run_step(b)
            assert model.roi_heads.num_classes == Ks[src]  # (a replay leaves the heads on its source, as an eager step does)
            pgt = model.roi_heads._last_pgt
            t = int(sum(len(torch.unique(x["instances"].gt_classes)) for x in b))
            hist.append(({k: float(v.detach()) for k, v in losses.items()}, pgt["gt_classes"][:128].cpu().clone(),
                         pgt["pgt_index"][:t].cpu().clone(), pgt["pgt_classes"][:t].cpu().clone()))
        if flag == "1":
            assert sorted(k[-2] for k in tr._graphs) == [0, 2] and all(type(g) is _StepGraph for g in tr._graphs.values())
        else:
            assert not tr._graphs
        tr.flush()
        runs[flag] = (hist, {k: v.detach().clone() for k, v in model.named_parameters() if v.requires_grad})
        tr.close()
    for s, (e, g) in enumerate(zip(runs["0"][0], runs["1"][0])):
        assert torch.equal(e[1], g[1]) and torch.equal(e[2], g[2]) and torch.equal(e[3], g[3]), s
        for k in e[0]:
            assert abs(e[0][k] - g[0][k]) <= 2e-5 * max(abs(e[0][k]), 1e-3), (s, k, e[0][k], g[0][k])
    for k, v in runs["0"][1].items():
        torch.testing.assert_close(runs["1"][1][k], v, rtol=1e-5, atol=2e-6 * float(v.abs().max()) + 1e-9, msg=lambda m: f"{k}: {m}")


def test_step_graph_falls_back_and_respects_its_limits(gpu, monkeypatch):
    """A layout is captured on its third sighting; another total proposal count is another graph; nine images stay
    eager; state_dict() between replays sees the applied update; a failing capture leaves the layout on the eager path
    (remembered, not retried) with the step still executed."""
    from wsovod_amd.engine import HotPathTrainer, build_optimizer
    from wsovod_amd.engine import trainer as T
    from wsovod_amd.testing import build_hot_path_model

    cfg, model = build_hot_path_model(seed=0, precision="bf16", device="cuda:0")
    model.train()
    cfg.SOLVER.BASE_LR = 1e-3
    tr = HotPathTrainer(model, build_optimizer(cfg, model))
    a = _varying_batches(5, [(48, 48)], H=256, W=320)
    for i, b in enumerate(a):
        tr.run_step(b)
        assert len(tr._graphs) == (1 if i >= 2 else 0)
    before = model.state_dict()["roi_heads.box_head.fc2.bias"].clone()
    tr.run_step(a[0])
    assert not torch.equal(before, model.state_dict()["roi_heads.box_head.fc2.bias"])  # the replayed update is visible
    # an LR scheduler moves the rates between steps: the captured SGD launch reads them from memory
    lrs = [g["lr"] for g in tr.optimizer.param_groups]
    for g in tr.optimizer.param_groups:
        g["lr"] = 0.0
    frozen = {k: v.detach().clone() for k, v in model.named_parameters() if v.requires_grad}
    tr.run_step(a[1])
    assert all(torch.equal(v, frozen[k]) for k, v in model.named_parameters() if v.requires_grad)
    for g, lr in zip(tr.optimizer.param_groups, lrs):
        g["lr"] = lr
    tr.run_step(a[2])
    assert not any(torch.equal(v, frozen[k]) for k, v in model.named_parameters() if v.requires_grad and v.numel() > 4)
    assert len(tr._graphs) == 1
    for b in _varying_batches(3, [(40, 40)], H=256, W=320):  # 80 rows: the same 128-row bucket, the same graph
        tr.run_step(b)
    assert len(tr._graphs) == 1
    for b in _varying_batches(3, [(100, 92)], H=256, W=320):  # 192 rows: another bucket, another graph
        tr.run_step(b)
    assert len(tr._graphs) == 2
    nine = _varying_batches(1, [(8,) * 9], H=256, W=320)[0]
    for _ in range(4):
        losses = tr.run_step(nine)
    assert len(tr._graphs) == 2 and all(torch.isfinite(v) for v in losses.values())

    def boom(self, *a_, **k_):
        raise RuntimeError("capture refused")

    monkeypatch.setattr(T._StepGraph, "_capture", boom)
    c = _varying_batches(4, [(150, 150)], H=256, W=320)
    with pytest.warns(UserWarning, match="capture of the training step failed"):
        for b in c[:3]:
            losses = tr.run_step(b)
    assert all(torch.isfinite(v) for v in losses.values()) and list(tr._graphs.values()).count(False) == 1
    tr.run_step(c[3])  # remembered: no second attempt, no second warning
    tr.close()


@pytest.mark.parametrize("pooler,nums_list", [("ROIAlignV2", [(96,), (81,), (70,), (96,)]), ("ROIPool", [(100,), (128,), (65,), (77,)])])
def test_step_graph_single_image_and_roialign(gpu, monkeypatch, pooler, nums_list):
    """The reference's own per-GPU batch -- ONE image per step -- and the north star's pooler through the step graph:
    six steps (two eager, the capture, three replays; the proposal count changes inside its bucket) against the eager
    trainer: labels equal, losses to the eager path's jitter."""
    from wsovod_amd.engine import HotPathTrainer, build_optimizer
    from wsovod_amd.layers import hip_ops as H
    from wsovod_amd.testing import build_hot_path_model

    monkeypatch.setattr(H, "DETERMINISTIC", True)
    monkeypatch.setenv("WSOVOD_BACKBONE_GRAPH", "0")
    batches = _varying_batches(6, nums_list, H=256, W=352)
    runs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("WSOVOD_STEP_GRAPH", flag)
        cfg, model = build_hot_path_model(seed=0, precision="bf16", pooler=pooler, device="cuda:0")
        model.train()
        cfg.SOLVER.BASE_LR = 1e-3
        tr = HotPathTrainer(model, build_optimizer(cfg, model))
        hist = []
        for b in batches:
            losses = tr.run_step(b)
            rows = len(b[0]["proposals"])
            hist.append(({k: float(v) for k, v in losses.items()}, model.roi_heads._last_pgt["gt_classes"][:rows].cpu().clone()))
        assert len(tr._graphs) == (1 if flag == "1" else 0)
        tr.close()
        runs[flag] = hist
    for s, (e, g) in enumerate(zip(runs["0"], runs["1"])):
        assert torch.equal(e[1], g[1]), s
        for k in e[0]:
            assert abs(e[0][k] - g[0][k]) <= 5e-5 * max(abs(e[0][k]), 1e-3), (s, k, e[0][k], g[0][k])


def test_eager_and_replayed_layouts_interleave_in_update_order(gpu, monkeypatch):
    """Two row buckets in rotation: X (128 rows) is captured and replayed, Y (192 rows) is still on the eager path when it
    comes between two replays of X.  The eager step leaves its SGD update deferred (overlap=True): the next replay must
    apply it BEFORE the graph reads the weights and runs its own update, or the update order drifts from the eager
    trainer's (and the reference's).  Step for step against WSOVOD_STEP_GRAPH=0."""
    from wsovod_amd.engine import HotPathTrainer, build_optimizer
    from wsovod_amd.layers import hip_ops as H
    from wsovod_amd.testing import build_hot_path_model

    monkeypatch.setattr(H, "DETERMINISTIC", True)
    monkeypatch.setenv("WSOVOD_BACKBONE_GRAPH", "0")
    X = _varying_batches(7, [(64, 64), (60, 57)], H=256, W=320)
    Y = _varying_batches(3, [(100, 92)], H=256, W=320)
    order = [X[0], X[1], X[2], X[3], Y[0], X[4], Y[1], X[5], X[6], Y[2]]  # X captured at its 3rd sighting; Y never (2 eager + capture at the end)
    runs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("WSOVOD_STEP_GRAPH", flag)
        cfg, model = build_hot_path_model(seed=0, precision="bf16", device="cuda:0")
        model.train()
        cfg.SOLVER.BASE_LR = 2e-2  # large steps: a missed / reordered update moves the next step's losses visibly
        tr = HotPathTrainer(model, build_optimizer(cfg, model))
        hist, kept = [], []
        for b in order:
            out = tr.run_step(b)
            kept.append(out)  # held across steps: must not alias the graph's static loss buffers
            hist.append({k: float(v) for k, v in out.items()})
        for h, o in zip(hist, kept):
            assert {k: float(v) for k, v in o.items()} == h
        tr.flush()
        runs[flag] = (hist, {k: v.detach().clone() for k, v in model.named_parameters() if v.requires_grad})
        tr.close()
    for s, (e, g) in enumerate(zip(runs["0"][0], runs["1"][0])):
        for k in e:
            assert abs(e[k] - g[k]) <= 5e-5 * max(abs(e[k]), 1e-3), (s, k, e[k], g[k])
    for k, v in runs["0"][1].items():
        torch.testing.assert_close(runs["1"][1][k], v, rtol=1e-4, atol=1e-5 * float(v.abs().max()) + 1e-9, msg=lambda m: f"{k}: {m}")


def test_step_graph_cache_is_lru_and_stops_capturing_when_it_thrashes(gpu, monkeypatch):
    """The cache keeps the most recently REPLAYED layouts; an evicted layout has to be sighted again (twice as often)
    before it is recaptured, and once recaptures exceed the budget no further layout is captured: multi-scale training with
    more hot layouts than the cache holds runs eager instead of re-tracing a graph per step."""
    from wsovod_amd.engine import HotPathTrainer, build_optimizer
    from wsovod_amd.testing import build_hot_path_model

    cfg, model = build_hot_path_model(seed=0, precision="bf16", device="cuda:0")
    model.train()
    tr = HotPathTrainer(model, build_optimizer(cfg, model))
    monkeypatch.setattr(tr, "GRAPH_CACHE", 2, raising=False)
    monkeypatch.setattr(tr, "GRAPH_RECAPTURES", 1, raising=False)
    shapes = [(128, 160), (128, 192), (160, 160)]
    data = {s: _varying_batches(1, [(32,)], H=s[0], W=s[1])[0] for s in shapes}

    def key(s):
        return tr._graph_key(data[s])

    A, B, C = shapes
    for _ in range(3):
        tr.run_step(data[A])
    for _ in range(3):
        tr.run_step(data[B])
    assert list(tr._graphs) == [key(A), key(B)]
    tr.run_step(data[A])  # a replay refreshes A: B is now the least recently used
    assert list(tr._graphs) == [key(B), key(A)]
    for _ in range(3):
        tr.run_step(data[C])
    assert list(tr._graphs) == [key(A), key(C)]  # B went, not A
    for _ in range(5):
        tr.run_step(data[B])  # evicted: not back at its third sighting ...
    assert key(B) not in tr._graphs
    tr.run_step(data[B])      # ... but at its sixth (recapture 1 of 1)
    assert key(B) in tr._graphs and tr._graph_recaptures == 1
    evicted = [k for k in (key(A), key(C)) if k not in tr._graphs][0]
    lay = A if evicted == key(A) else C
    with pytest.warns(UserWarning, match="more training-step layouts in rotation"):
        for _ in range(6):
            losses = tr.run_step(data[lay])
    assert evicted not in tr._graphs and all(torch.isfinite(v) for v in losses.values())
    n = len(tr._graphs)
    for _ in range(4):
        tr.run_step(_varying_batches(1, [(32,)], H=192, W=192)[0])  # budget spent: new layouts stay eager too
    assert len(tr._graphs) == n
    tr.close()
