"""The frozen backbone as a captured HIP graph (small batches; opt-in, set by the overlapped trainer): replays must be
bit-identical to the eager launches, per input, across weight changes, in both kernel precisions that have a fused stem."""
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
        return bb.forward_uint8(canvas, sizes_t, model._mean, model._std)["res5"]

    eager = [run(b).clone() for b in batches]
    bb.graph_max_batch = 8
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
