"""Bisect what a HIP graph capture of the training step can hold: stages frozen / +heads / +backward / +sgd, in a
subprocess each (a failing capture may take the process down).  python tools/graph_probe.py [stage] [mode]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_trainer(variant):
    """stage 4+: the real thing through HotPathTrainer. variant 4: device-resident constant batch; 5: host batch, constant
    counts; 6: host batches with varying per-image counts."""
    import faulthandler
    import torch
    faulthandler.enable()
    from wsovod_amd.data import make_batch
    from wsovod_amd.engine import HotPathTrainer, build_optimizer
    from wsovod_amd.testing import build_hot_path_model

    os.environ["WSOVOD_BACKBONE_GRAPH"] = "0"
    cfg, model = build_hot_path_model(seed=0, precision="bf16", device="cuda:0")
    model.train()
    cfg.SOLVER.BASE_LR = 1e-3
    tr = HotPathTrainer(model, build_optimizer(cfg, model))
    dev = model.device
    for s in range(5):
        nums = [(64, 64), (60, 68), (70, 58), (33, 95), (64, 64)][s] if variant == 6 else (64, 64)
        host = make_batch(2, max(nums), 20, H=320, W=416, seed=3 + (s if variant >= 5 else 0))
        for x, n in zip(host, nums):
            x["proposals"] = x["proposals"][:n]
        batch = host if variant >= 5 else [{"image": x["image"].to(dev), "proposals": x["proposals"].to(dev),
                                            "instances": x["instances"], "height": x["height"], "width": x["width"]} for x in host]
        out = tr.run_step(batch)
        torch.cuda.synchronize()
        print("step", s, len(tr._graphs), {k: round(float(v.detach()), 5) for k, v in out.items()}, flush=True)


def run_static(stage):
    """stages 10..13: as 0..3 but on the static batch of _StepGraph (views of its input buffers)."""
    import faulthandler
    import torch
    faulthandler.enable()
    from wsovod_amd.data import make_batch
    from wsovod_amd.engine import HotPathTrainer, build_optimizer
    from wsovod_amd.engine.trainer import _StepGraph, _StepMeta
    from wsovod_amd.layers import hip_ops as H
    from wsovod_amd.testing import build_hot_path_model

    os.environ["WSOVOD_STEP_GRAPH"] = "0"
    os.environ["WSOVOD_BACKBONE_GRAPH"] = "0"
    cfg, model = build_hot_path_model(seed=0, precision="bf16", device="cuda:0")
    model.train()
    cfg.SOLVER.BASE_LR = 1e-3
    tr = HotPathTrainer(model, build_optimizer(cfg, model))
    dev = model.device
    host = make_batch(2, 64, 20, H=320, W=416, seed=3)
    for _ in range(2):
        tr.run_step(host)
    tr.flush()
    sg = _StepGraph.__new__(_StepGraph)
    sg.tr = tr
    sg.canvas = torch.empty((2, 3, 320, 416), dtype=torch.uint8, device=dev)
    sg.boxes = torch.zeros((128, 4), device=dev)
    sg.objectness = torch.zeros((128,), device=dev)
    sg.meta = _StepMeta(2, 128, 20, dev)
    assert sg._load(host)
    static = sg._static_batch(host, [64, 64])
    model._step_meta = sg.meta
    st = stage - 10
    g = torch.cuda.CUDAGraph()
    with H.const_override(sg.meta.overrides()):
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            s_ = model.forward_frozen(static)
            if st >= 1:
                loss = model.forward_trainable(s_)
            if st >= 2:
                tr._backward(loss)
            if st >= 3:
                tr.optimizer.step()
                tr.optimizer.zero_grad(set_to_none=True)
    print("captured", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print("replayed", stage, flush=True)


def run_ctor(stage):
    """20: _StepGraph(tr, host, key) called directly; 21: the same with a strong reference instead of the proxy;
    22: without the _dw_split loop; 23: capture_error_mode global."""
    import faulthandler
    import torch
    faulthandler.enable()
    from wsovod_amd.data import make_batch
    from wsovod_amd.engine import HotPathTrainer, build_optimizer
    from wsovod_amd.engine import trainer as T
    from wsovod_amd.testing import build_hot_path_model

    os.environ["WSOVOD_STEP_GRAPH"] = "0"
    os.environ["WSOVOD_BACKBONE_GRAPH"] = "0"
    cfg, model = build_hot_path_model(seed=0, precision="bf16", device="cuda:0")
    model.train()
    cfg.SOLVER.BASE_LR = 1e-3
    tr = HotPathTrainer(model, build_optimizer(cfg, model))
    host = make_batch(2, 64, 20, H=320, W=416, seed=3)
    for _ in range(2):
        tr.run_step(host)
    if stage == 21:
        import weakref
        weakref.proxy = lambda x: x
    if stage == 23:
        real = torch.cuda.graph
        torch.cuda.graph = lambda g, **k: real(g, **{**k, "capture_error_mode": "global"})
    sg = T._StepGraph(tr, host, ("k",))
    print("captured", flush=True)
    out = sg.step(host)
    torch.cuda.synchronize()
    print("replayed", stage, {k: float(v) for k, v in out.items()}, flush=True)


def run_mix(stage):
    """30: trainer path (graph on), one host batch reused, results not read; 31: graph off for two steps on fresh batches,
    then the constructor on a fresh batch; 32: as 30 but results read (float) after every step."""
    import faulthandler
    import torch
    faulthandler.enable()
    from wsovod_amd.data import make_batch
    from wsovod_amd.engine import HotPathTrainer, build_optimizer
    from wsovod_amd.engine import trainer as T
    from wsovod_amd.testing import build_hot_path_model

    os.environ["WSOVOD_BACKBONE_GRAPH"] = "0"
    if stage == 31:
        os.environ["WSOVOD_STEP_GRAPH"] = "0"
    cfg, model = build_hot_path_model(seed=0, precision="bf16", device="cuda:0")
    model.train()
    cfg.SOLVER.BASE_LR = 1e-3
    tr = HotPathTrainer(model, build_optimizer(cfg, model))
    host = make_batch(2, 64, 20, H=320, W=416, seed=3)
    if stage in (30, 32):
        for s_ in range(4):
            out = tr.run_step(host)
            if stage == 32:
                print({k: float(v.detach()) for k, v in out.items()}, flush=True)
            print("step", s_, len(tr._graphs), "host step", model.roi_heads.box_head._step, "device term",
                  int(model.roi_heads.box_head._step_dev), flush=True)
    else:
        for s_ in range(2):
            tr.run_step(make_batch(2, 64, 20, H=320, W=416, seed=3 + s_))
        sg = T._StepGraph(tr, make_batch(2, 64, 20, H=320, W=416, seed=9), ("k",))
        print("captured", flush=True)
    torch.cuda.synchronize()
    print("done", stage, flush=True)


def run(stage, mode, precision="bf16"):
    if stage >= 30:
        return run_mix(stage)
    if stage >= 20:
        return run_ctor(stage)
    if stage >= 10:
        return run_static(stage)
    if stage >= 4:
        return run_trainer(stage)
    import torch
    from wsovod_amd.data import make_batch
    from wsovod_amd.engine import HotPathTrainer, build_optimizer
    from wsovod_amd.engine.trainer import _StepMeta
    from wsovod_amd.layers import hip_ops as H
    from wsovod_amd.testing import build_hot_path_model

    os.environ["WSOVOD_STEP_GRAPH"] = "0"
    os.environ["WSOVOD_BACKBONE_GRAPH"] = "0"
    cfg, model = build_hot_path_model(seed=0, precision=precision, device="cuda:0")
    model.train()
    cfg.SOLVER.BASE_LR = 1e-3
    tr = HotPathTrainer(model, build_optimizer(cfg, model))
    dev = model.device
    host = make_batch(2, 64, 20, H=320, W=416, seed=3)
    batch = [{"image": x["image"].to(dev), "proposals": x["proposals"].to(dev), "instances": x["instances"],
              "height": x["height"], "width": x["width"]} for x in host]
    for _ in range(2):
        tr.run_step(batch)
    tr.flush()
    torch.cuda.synchronize()
    meta = _StepMeta(2, 128, 20, dev)
    assert meta.fill(batch)
    model._step_meta = meta
    g = torch.cuda.CUDAGraph()
    with H.const_override(meta.overrides()):
        with torch.cuda.graph(g, capture_error_mode=mode):
            st = model.forward_frozen(batch)
            if stage >= 1:
                loss = model.forward_trainable(st)
            if stage >= 2:
                tr._backward(loss)
            if stage >= 3:
                tr.optimizer.step()
                tr.optimizer.zero_grad(set_to_none=True)
    model._step_meta = None
    print("captured", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print("replayed", stage, mode, {k: float(v) for k, v in loss.items()} if stage >= 1 else "", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(int(sys.argv[1]), sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "bf16")
    else:
        for mode in ("thread_local", "global"):
            for stage in (range(4) if mode == "global" else (4, 5, 6)):
                p = subprocess.run([sys.executable, os.path.abspath(__file__), str(stage), mode], capture_output=True, text=True)
                tail = (p.stdout + p.stderr).strip().splitlines()[-3:]
                print(f"stage {stage} mode {mode}: rc={p.returncode}", " | ".join(t[:160] for t in tail), flush=True)
