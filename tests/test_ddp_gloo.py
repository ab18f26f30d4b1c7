"""CPU, world_size 2, gloo: the multi-GPU leg of the path is plain data parallelism -- images sharded over
ranks, one gradient all-reduce per step (reference: engine/defaults.py:143-148, data/build.py:314-320).
The HIP model itself needs a GPU, so this drives the same engine code (wrap_model_with_ddp + run_step) with a
CPU stand-in module that has the path's loss-dict interface."""
import os

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


class _LossDictModel(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.fc = nn.Linear(8, 4)

    def forward(self, batch):
        if hasattr(self, "forward_frozen"):
            return self.forward_trainable(self.forward_frozen(batch))
        x = torch.stack([b["x"] for b in batch])
        y = self.fc(x)
        return {"loss_a": y.pow(2).mean(), "loss_b": y.abs().mean() * 0.5}


class _TwoPhaseModel(_LossDictModel):
    """Stand-in with the hot path's forward_frozen / forward_trainable split."""

    def forward_frozen(self, batch):
        return {"x": torch.stack([b["x"] for b in batch]) * 2.0}  # parameter-free work

    def forward_trainable(self, st):
        y = self.fc(st["x"])
        return {"loss_a": y.pow(2).mean(), "loss_b": y.abs().mean() * 0.5}


def _worker_overlap(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wsovod_amd.engine import HotPathTrainer

    model = _TwoPhaseModel()
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
    tr = HotPathTrainer(model, opt, overlap=True)
    tr.broadcast_parameters()
    g = torch.Generator().manual_seed(1234 + rank)
    batch = [{"x": torch.randn(8, generator=g)} for _ in range(3)]
    for it in range(3):
        tr.run_step(batch)
    tr.flush()
    q.put((rank, model.fc.weight.detach().tolist(), [b["x"].tolist() for b in batch]))
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wsovod_amd.engine import run_step, wrap_model_with_ddp

    model = _LossDictModel()
    ddp = wrap_model_with_ddp(model, rank)
    assert ddp is not model
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
    g = torch.Generator().manual_seed(1234 + rank)  # bench.py seeds each rank's shard the same way
    batch = [{"x": torch.randn(8, generator=g)} for _ in range(3)]
    for it in range(3):
        run_step(ddp, opt, batch, it=it)
    q.put((rank, model.fc.weight.detach().tolist(), [b["x"].tolist() for b in batch]))  # plain lists: no shm handles
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_step_matches_single_process_average():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w0, w1 = torch.tensor(res[0][1]), torch.tensor(res[1][1])
    assert torch.equal(w0, w1)  # replicas stay in lock-step
    assert res[0][2][0] != res[1][2][0]  # each rank trained on its own shard
    # single-process reference: gradient = mean over ranks of the per-rank mean loss gradients
    model = _LossDictModel()
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
    for it in range(3):
        opt.zero_grad()
        total = 0
        for r in range(2):
            total = total + sum(model([{"x": torch.tensor(x)} for x in res[r][2]]).values()) / 2
        total.backward()
        opt.step()
    torch.testing.assert_close(model.fc.weight.detach(), w0, rtol=1e-5, atol=1e-6)


def test_overlapped_trainer_matches_plain_data_parallel():
    """HotPathTrainer (async per-tensor all-reduce finished after the next frozen forward) == averaged SGD."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_overlap, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w0, w1 = torch.tensor(res[0][1]), torch.tensor(res[1][1])
    assert torch.equal(w0, w1)
    model = _TwoPhaseModel()
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
    for it in range(3):
        opt.zero_grad()
        total = 0
        for r in range(2):
            total = total + sum(model([{"x": torch.tensor(x)} for x in res[r][2]]).values()) / 2
        total.backward()
        opt.step()
    torch.testing.assert_close(model.fc.weight.detach(), w0, rtol=1e-5, atol=1e-6)


class _TwoSourceModel(nn.Module):
    """Mixed-dataset stand-in: the batch's dataset id picks one of two heads, the other gets no gradient."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.heads = nn.ModuleList([nn.Linear(8, 4), nn.Linear(8, 6)])

    def forward_frozen(self, batch):
        return {"x": torch.stack([b["x"] for b in batch]), "source": batch[0]["dataset_id"]}

    def forward_trainable(self, st):
        return {"loss": self.heads[st["source"]](st["x"]).pow(2).mean()}


def _worker_mixed(rank, world, port, q, same_source=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wsovod_amd.engine import HotPathTrainer

    model = _TwoSourceModel()
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=0.05 if same_source else 0.0)
    tr = HotPathTrainer(model, opt, overlap=True, reduce_unused=True)
    tr.broadcast_parameters()
    g = torch.Generator().manual_seed(77 + rank)
    xs = [torch.randn(8, generator=g) for _ in range(3)]
    for it in range(4 if same_source else 3):
        # different datasets on the two ranks every step / the same dataset on both (the other miner unused everywhere)
        src = (0, 0, 1, 0)[it] if same_source else (rank + it) % 2
        tr.run_step([{"x": x, "dataset_id": src} for x in xs])
    tr.flush()
    q.put((rank, [h.weight.detach().tolist() for h in model.heads], [x.tolist() for x in xs]))
    dist.barrier()
    dist.destroy_process_group()


def test_mixed_dataset_ranks_with_different_sources_stay_in_lockstep():
    """Ranks whose batches come from different datasets touch different object miners; with
    reduce_unused the exchange still lines up (no hang) and equals the averaged single-process update."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_mixed, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(torch.tensor(a), torch.tensor(b))
    model = _TwoSourceModel()
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
    for it in range(3):
        opt.zero_grad()
        total = 0
        for r in range(2):
            st = model.forward_frozen([{"x": torch.tensor(x), "dataset_id": (r + it) % 2} for x in res[r][2]])
            total = total + model.forward_trainable(st)["loss"] / 2
        total.backward()
        opt.step()
    for h, w in zip(model.heads, res[0][1]):
        torch.testing.assert_close(h.weight.detach(), torch.tensor(w), rtol=1e-5, atol=1e-6)


def test_tensors_unused_on_every_rank_are_skipped_like_sgd_none_grad():
    """DDP(find_unused_parameters=True) + SGD in the reference (engine/defaults.py:146-148): a parameter that no rank
    used keeps `grad is None`, so SGD applies neither weight decay nor momentum to it.  With reduce_unused the ranks
    exchange a per-tensor used flag next to the zero gradients and skip such tensors."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 34100 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_mixed, args=(r, 2, port, q, True)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model = _TwoSourceModel()
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=0.05)
    for it in range(4):
        opt.zero_grad(set_to_none=True)
        total = 0
        for r in range(2):
            st = model.forward_frozen([{"x": torch.tensor(x), "dataset_id": (0, 0, 1, 0)[it]} for x in res[r][2]])
            total = total + model.forward_trainable(st)["loss"] / 2
        total.backward()
        opt.step()  # the other head's grad is None: skipped
    for h, w0, w1 in zip(model.heads, res[0][1], res[1][1]):
        assert torch.equal(torch.tensor(w0), torch.tensor(w1))
        torch.testing.assert_close(h.weight.detach(), torch.tensor(w0), rtol=1e-5, atol=1e-6)


def test_state_read_between_steps_sees_the_applied_update_and_iter_size():
    """The overlapped trainer applies step t's update lazily; state_dict() (checkpoint / eval hooks run after run_step
    in the reference, engine/trainer.py:82-84) must see the weights AFTER optimizer.step().  ITER_SIZE > 1: same
    accumulate-and-step rule as the plain run_step (`iter % iter_size == 0`)."""
    from wsovod_amd.engine import HotPathTrainer, run_step

    g = torch.Generator().manual_seed(5)
    batches = [[{"x": torch.randn(8, generator=g)} for _ in range(3)] for _ in range(5)]
    for iter_size in (1, 2, 3):
        plain = _LossDictModel()
        popt = torch.optim.SGD(plain.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-3)
        lazy = _TwoPhaseModel()
        lopt = torch.optim.SGD(lazy.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-3)
        # _TwoPhaseModel doubles x in its frozen half: feed the plain model the doubled inputs
        tr = HotPathTrainer(lazy, lopt, overlap=True, iter_size=iter_size)
        for it, b in enumerate(batches):
            run_step(plain, popt, [{"x": d["x"] * 2.0} for d in b], iter_size=iter_size, it=it)
            tr.run_step(b)
            sd = lazy.state_dict()  # pre-hook flushes the pending update
            torch.testing.assert_close(sd["fc.weight"], plain.fc.weight.detach(), rtol=1e-6, atol=1e-7)
            mom = lopt.state_dict()["state"]
            pm = popt.state_dict()["state"]
            assert set(mom.keys()) == set(pm.keys())
            for k in mom:
                torch.testing.assert_close(mom[k]["momentum_buffer"], pm[k]["momentum_buffer"], rtol=1e-6, atol=1e-7)
        tr.synchronize()


def test_lr_scheduler_between_steps_does_not_move_the_deferred_update():
    """The reference steps its LR scheduler AFTER run_step (an after_step hook: optimizer.step() of iteration t has used
    lr_t by then).  The overlapped trainer applies step t's update inside run_step(t + 1): it must still be applied with
    lr_t (and weight_decay_t), whatever the scheduler wrote into the param groups in between."""
    from wsovod_amd.engine import HotPathTrainer, run_step

    g = torch.Generator().manual_seed(9)
    batches = [[{"x": torch.randn(8, generator=g)} for _ in range(3)] for _ in range(6)]
    lrs = [0.01 * (0.001 * (1 - i / 200) + i / 200) * 500 for i in range(6)]  # a warm-up ramp (scaled up: visible steps)
    plain, lazy = _LossDictModel(), _TwoPhaseModel()
    popt = torch.optim.SGD(plain.parameters(), lr=lrs[0], momentum=0.9, weight_decay=1e-3)
    lopt = torch.optim.SGD(lazy.parameters(), lr=lrs[0], momentum=0.9, weight_decay=1e-3)
    tr = HotPathTrainer(lazy, lopt, overlap=True)
    for it, b in enumerate(batches):
        run_step(plain, popt, [{"x": d["x"] * 2.0} for d in b])
        tr.run_step(b)
        if it + 1 < len(lrs):  # the scheduler hook, after the step
            for opt in (popt, lopt):
                for grp in opt.param_groups:
                    grp["lr"] = lrs[it + 1]
                    grp["weight_decay"] = 1e-3 * (it + 2)
    tr.flush()
    assert lopt.param_groups[0]["lr"] == lrs[-1]  # the scheduler's value is back in place after the deferred step
    torch.testing.assert_close(lazy.fc.weight.detach(), plain.fc.weight.detach(), rtol=1e-6, atol=1e-7)
    tr.close()


def _worker_bf16_wire(rank, world, port, q):
    """grad_wire="bf16" on CPU/gloo: the two HIP kernels of that path (gradient pack, SGD on bf16 slices) are replaced
    by torch stand-ins with the same contract, so that the trainer's own logic -- flat buffer slices, one collective,
    zero contribution for untouched tensors, 1/world scaling, release of the fp32 gradients -- runs under 2 ranks."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wsovod_amd.engine import HotPathTrainer
    from wsovod_amd.engine.trainer import HipSGD
    from wsovod_amd.layers import hip_ops as H

    def pack(pairs):
        for src, dst in pairs:
            assert src.dtype == torch.float32 and dst.dtype == torch.bfloat16 and dst.data_ptr() % 16 == 0
            dst.copy_(src.reshape(-1))

    def sgd(entries, momentum, grad_scale=1.0, clip=None):
        for p, g, buf, shadow, lr, wd, used in entries:
            assert g.dtype == torch.bfloat16 and used is not None
            if float(used) == 0.0:  # no rank produced a gradient: the kernel leaves parameter and momentum alone
                continue
            d = g.float().view_as(p) * grad_scale + wd * p
            buf.mul_(momentum).add_(d)
            p.sub_(lr * buf)

    H.pack_bf16_multi, H.sgd_momentum_multi = pack, sgd
    model = _TwoPhaseModel()
    model.unused = nn.Linear(3, 3)  # never touched by the loss: takes part with zeros (reduce_unused)
    unused0 = model.unused.weight.detach().clone()
    # weight decay ON: a tensor no rank touched must still stay put (SGD skips `grad is None` in the reference)
    opt = HipSGD([{"params": [p], "lr": 0.1, "weight_decay": 0.0 if p is model.fc.weight or p is model.fc.bias else 0.1}
                  for p in model.parameters()], 0.1, momentum=0.9)
    tr = HotPathTrainer(model, opt, overlap=True, reduce_unused=True, grad_wire="bf16")
    tr.broadcast_parameters()
    g = torch.Generator().manual_seed(1234 + rank)
    batch = [{"x": torch.randn(8, generator=g)} for _ in range(3)]
    for it in range(3):
        tr.run_step(batch)
        assert all(p.grad is None for p in tr.params)
    tr.flush()
    assert torch.equal(model.unused.weight.detach(), unused0), "a tensor unused on every rank must not decay"
    q.put((rank, model.fc.weight.detach().tolist(), model.unused.weight.detach().tolist(),
           [b["x"].tolist() for b in batch]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bf16_gradient_wire():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_bf16_wire, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w0, w1 = torch.tensor(res[0][1]), torch.tensor(res[1][1])
    assert torch.equal(w0, w1), "replicas diverged"
    # reference: the same three steps in one process, gradients averaged in fp32 and rounded to bf16 per rank first
    model = _TwoPhaseModel()
    buf = {n: torch.zeros_like(p) for n, p in model.named_parameters()}
    batches = [[{"x": torch.tensor(x)} for x in r[3]] for r in res]
    for it in range(3):
        grads = []
        for b in batches:
            model.zero_grad()
            sum(model(b).values()).backward()
            grads.append({n: p.grad.to(torch.bfloat16) for n, p in model.named_parameters()})
        with torch.no_grad():
            for n, p in model.named_parameters():
                gsum = (grads[0][n].float() + grads[1][n].float()).to(torch.bfloat16)  # gloo sums in bf16
                buf[n].mul_(0.9).add_(gsum.float() * 0.5)
                p.sub_(0.1 * buf[n])
    torch.testing.assert_close(w0, model.fc.weight.detach(), rtol=0, atol=2e-3)
    assert (w0 - model.fc.weight.detach()).abs().max() < 2e-3
    # the untouched tensor saw zero gradients on both ranks: unchanged (no weight decay here), identical on both
    assert res[0][2] == res[1][2]


class _SplitLinearFn(torch.autograd.Function):
    """CPU stand-in with the protocol of layers/functions.py:_Linear: the weight gradient is produced in two row
    blocks and `weight._dw_split = (rows, callback)` sees the first one before the second exists."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        ctx.dw_split = getattr(weight, "_dw_split", None)
        return x @ weight.t()

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dw = torch.empty_like(weight)
        ra = ctx.dw_split[0]
        dw[:ra] = dy[:, :ra].t() @ x
        ctx.dw_split[1](dw[:ra])
        dw[ra:] = dy[:, ra:].t() @ x
        return None, dw


class _SplitModel(nn.Module):
    def __init__(self, cols=16):
        super().__init__()
        torch.manual_seed(0)
        self.big = nn.Parameter(torch.randn(24, cols) * 0.1)
        self.small = nn.Linear(24, 3)

    def forward_frozen(self, batch):
        return {"x": torch.stack([b["x"] for b in batch])}

    def forward_trainable(self, st):
        h = _SplitLinearFn.apply(st["x"], self.big)
        return {"loss": self.small(torch.relu(h)).pow(2).mean()}


def _worker_early_block(rank, world, port, q):
    """The split-weight path of the bf16 wire: first row block packed and all-reduced from inside backward, the rest
    (and every other tensor) after it; two collectives per step on every rank."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wsovod_amd.engine import HotPathTrainer
    from wsovod_amd.engine.trainer import HipSGD
    from wsovod_amd.layers import hip_ops as H

    def pack(pairs):
        for src, dst in pairs:
            assert src.dtype == torch.float32 and dst.dtype == torch.bfloat16 and src.numel() == dst.numel()
            dst.copy_(src.reshape(-1))

    def sgd(entries, momentum, grad_scale=1.0, clip=None):
        for p, g, buf, shadow, lr, wd, used in entries:
            assert g.dtype == torch.bfloat16 and g.numel() == p.numel() and used is None  # (no reduce_unused here)
            buf.mul_(momentum).add_(g.float().view_as(p) * grad_scale + wd * p)
            p.sub_(lr * buf)

    H.pack_bf16_multi, H.sgd_momentum_multi = pack, sgd
    HotPathTrainer.split_on_cpu = True
    HotPathTrainer.split_rows = staticmethod(lambda n_rows, n_cols, cus=256, tile=256: 8)
    calls = []
    real = dist.all_reduce

    def counted(t, *a, **k):
        calls.append(t.numel())
        return real(t, *a, **k)

    dist.all_reduce = counted
    model = _SplitModel()
    opt = HipSGD([{"params": [p], "lr": 0.1, "weight_decay": 0.0} for p in model.parameters()], 0.1, momentum=0.9)
    tr = HotPathTrainer(model, opt, overlap=True, grad_wire="bf16")
    assert tr._split is not None and tr._split[1] == 8 and model.big._dw_split[0] == 8
    tr.broadcast_parameters()
    calls.clear()
    g = torch.Generator().manual_seed(99 + rank)
    batch = [{"x": torch.randn(16, generator=g)} for _ in range(4)]
    for it in range(3):
        tr.run_step(batch)
    tr.flush()
    total = tr._wire_slices()[0].numel()
    assert calls == [8 * 16, total - 8 * 16] * 3, calls  # early head, then the rest, every step
    q.put((rank, model.big.detach().tolist(), model.small.weight.detach().tolist(), [b["x"].tolist() for b in batch]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bf16_wire_early_block():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_early_block, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2], "replicas diverged"
    model = _SplitModel()
    model.big._dw_split = (8, lambda rows: None)
    buf = {n: torch.zeros_like(p) for n, p in model.named_parameters()}
    batches = [[{"x": torch.tensor(x)} for x in r[3]] for r in res]
    for it in range(3):
        grads = []
        for b in batches:
            model.zero_grad()
            sum(model.forward_trainable(model.forward_frozen(b)).values()).backward()
            grads.append({n: p.grad.to(torch.bfloat16) for n, p in model.named_parameters()})
        with torch.no_grad():
            for n, p in model.named_parameters():
                gsum = (grads[0][n].float() + grads[1][n].float()).to(torch.bfloat16)
                buf[n].mul_(0.9).add_(gsum.float() * 0.5)
                p.sub_(0.1 * buf[n])
    torch.testing.assert_close(torch.tensor(res[0][1]), model.big.detach(), rtol=0, atol=2e-3)
    torch.testing.assert_close(torch.tensor(res[0][2]), model.small.weight.detach(), rtol=0, atol=2e-3)


def _worker_direct(rank, world, port, q):
    """exchange="direct" on the bf16 wire: all-to-all of the shards, fp32 sum of the `world` copies with ONE rounding,
    all-gather -- for the early row block of the split weight and for the rest.  The HIP kernels are replaced by torch
    stand-ins with the same contract; three ranks, so that shards, padding and the rounding count are all non-trivial."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wsovod_amd.engine import HotPathTrainer
    from wsovod_amd.engine.trainer import HipSGD
    from wsovod_amd.layers import hip_ops as H

    def pack(pairs):
        for src, dst in pairs:
            dst.copy_(src.reshape(-1))

    def sgd(entries, momentum, grad_scale=1.0, clip=None):
        for p, g, buf, shadow, lr, wd, used in entries:
            assert g.dtype == torch.bfloat16 and g.numel() == p.numel()
            buf.mul_(momentum).add_(g.float().view_as(p) * grad_scale + wd * p)
            p.sub_(lr * buf)

    def sum_shards(src, n, dst):
        assert src.dtype == torch.bfloat16 and src.numel() == n * dst.numel() and dst.numel() % 8 == 0
        dst.copy_(src.view(n, -1).float().sum(0))
        return dst

    H.pack_bf16_multi, H.sgd_momentum_multi, H.sum_shards_bf16 = pack, sgd, sum_shards
    HotPathTrainer.split_on_cpu = True
    HotPathTrainer.split_rows = staticmethod(lambda n_rows, n_cols, cus=256, tile=256: 8)
    calls = []
    real_a2a, real_ag, real_ar = dist.all_to_all_single, dist.all_gather_into_tensor, dist.all_reduce
    dist.all_to_all_single = lambda out, inp, *a, **k: (calls.append(("a2a", inp.numel())), real_a2a(out, inp, *a, **k))[1]
    dist.all_gather_into_tensor = lambda out, inp, *a, **k: (calls.append(("ag", out.numel())), real_ag(out, inp, *a, **k))[1]
    dist.all_reduce = lambda t, *a, **k: (calls.append(("ar", t.numel())), real_ar(t, *a, **k))[1]
    model = _SplitModel(cols=48)
    opt = HipSGD([{"params": [p], "lr": 0.1, "weight_decay": 0.0} for p in model.parameters()], 0.1, momentum=0.9)
    tr = HotPathTrainer(model, opt, overlap=True, grad_wire="bf16", exchange="auto")
    assert tr.exchange_algo == "direct" and tr._split is not None and tr._split[2] == 8 * 48
    tr.broadcast_parameters()
    calls.clear()
    g = torch.Generator().manual_seed(7 + rank)
    batch = [{"x": torch.randn(48, generator=g)} for _ in range(4)]
    for it in range(3):
        tr.run_step(batch)
    tr.flush()
    total = tr._wire_slices()[0].numel()
    assert total % (8 * world) == 0 and total >= 24 * 48 + 3 * 24 + 3
    head = 8 * 48
    assert calls == [("a2a", head), ("ag", head), ("a2a", total - head), ("ag", total - head)] * 3, calls
    q.put((rank, model.big.detach().tolist(), model.small.weight.detach().tolist(), [b["x"].tolist() for b in batch]))
    dist.barrier()
    dist.destroy_process_group()


def test_three_rank_direct_exchange_sums_in_fp32_and_rounds_once():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    world = 3
    procs = [ctx.Process(target=_worker_direct, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] == res[0][1] and r[2] == res[0][2] for r in res), "replicas diverged"
    model = _SplitModel(cols=48)
    model.big._dw_split = (8, lambda rows: None)
    buf = {n: torch.zeros_like(p) for n, p in model.named_parameters()}
    batches = [[{"x": torch.tensor(x)} for x in r[3]] for r in res]
    for it in range(3):
        grads = []
        for b in batches:
            model.zero_grad()
            sum(model.forward_trainable(model.forward_frozen(b)).values()).backward()
            grads.append({n: p.grad.to(torch.bfloat16) for n, p in model.named_parameters()})
        with torch.no_grad():
            for n, p in model.named_parameters():
                gsum = sum(g[n].float() for g in grads).to(torch.bfloat16)  # fp32 sum of the bf16 copies, ONE rounding
                buf[n].mul_(0.9).add_(gsum.float() * (1.0 / world))
                p.sub_(0.1 * buf[n])
    torch.testing.assert_close(torch.tensor(res[0][1]), model.big.detach(), rtol=0, atol=1e-6)
    torch.testing.assert_close(torch.tensor(res[0][2]), model.small.weight.detach(), rtol=0, atol=1e-6)


def _worker_direct_unused_iter(rank, world, port, q):
    """Direct exchange together with reduce_unused (a tensor no rank touches keeps parameter and momentum, weight decay
    on) and ITER_SIZE = 2 (gradients accumulate locally, exchange + update every second iteration): three ranks."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wsovod_amd.engine import HotPathTrainer
    from wsovod_amd.engine.trainer import HipSGD
    from wsovod_amd.layers import hip_ops as H

    def pack(pairs):
        for src, dst in pairs:
            dst.copy_(src.reshape(-1))

    def sgd(entries, momentum, grad_scale=1.0, clip=None):
        for p, g, buf, shadow, lr, wd, used in entries:
            assert used is not None
            if float(used) == 0.0:
                continue
            buf.mul_(momentum).add_(g.float().view_as(p) * grad_scale + wd * p)
            p.sub_(lr * buf)

    def sum_shards(src, n, dst):
        dst.copy_(src.view(n, -1).float().sum(0))
        return dst

    H.pack_bf16_multi, H.sgd_momentum_multi, H.sum_shards_bf16 = pack, sgd, sum_shards
    calls = []
    real = dist.all_to_all_single
    dist.all_to_all_single = lambda out, inp, *a, **k: (calls.append(inp.numel()), real(out, inp, *a, **k))[1]
    model = _TwoPhaseModel()
    model.unused = nn.Linear(5, 3)
    unused0 = model.unused.weight.detach().clone()
    opt = HipSGD([{"params": [p], "lr": 0.1, "weight_decay": 0.05} for p in model.parameters()], 0.1, momentum=0.9)
    tr = HotPathTrainer(model, opt, overlap=True, reduce_unused=True, grad_wire="bf16", exchange="direct", iter_size=2)
    assert tr.exchange_algo == "direct" and tr._split is None  # accumulation: no early block
    tr.broadcast_parameters()
    calls.clear()
    g = torch.Generator().manual_seed(300 + rank)
    batches = [[{"x": torch.randn(8, generator=g)} for _ in range(3)] for _ in range(4)]
    for b in batches:
        tr.run_step(b)
    tr.flush()
    total = tr._wire_slices()[0].numel()
    assert total % (8 * world) == 0 and calls == [total, total], calls  # iterations 0 and 2 exchange, 1 and 3 accumulate
    assert torch.equal(model.unused.weight.detach(), unused0), "a tensor unused on every rank must not move"
    q.put((rank, model.fc.weight.detach().tolist(), [[d["x"].tolist() for d in b] for b in batches]))
    dist.barrier()
    dist.destroy_process_group()


def test_three_rank_direct_exchange_with_unused_tensors_and_iter_size():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36500 + os.getpid() % 2000
    world = 3
    procs = [ctx.Process(target=_worker_direct_unused_iter, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] == res[0][1] for r in res), "replicas diverged"
    # reference: iteration 0 steps on its own gradient; iterations 1 and 2 accumulate (each loss / 2) and step at 2;
    # iteration 3 accumulates and is never applied.  Per step: bf16 per rank, fp32 sum, one rounding, / world.
    model = _TwoPhaseModel()
    buf = {n: torch.zeros_like(p) for n, p in model.named_parameters()}

    def rank_grads(its):
        out = []
        for r in res:
            model.zero_grad()
            for it in its:
                (sum(model([{"x": torch.tensor(x)} for x in r[2][it]]).values()) / 2).backward()
            out.append({n: p.grad.to(torch.bfloat16) for n, p in model.named_parameters()})
        return out

    for its in ([0], [1, 2]):
        grads = rank_grads(its)
        with torch.no_grad():
            for n, p in model.named_parameters():
                gsum = sum(gr[n].float() for gr in grads).to(torch.bfloat16)
                buf[n].mul_(0.9).add_(gsum.float() / world + 0.05 * p)
                p.sub_(0.1 * buf[n])
    torch.testing.assert_close(torch.tensor(res[0][1]), model.fc.weight.detach(), rtol=0, atol=1e-6)


def _worker_pick_exchange(rank, world, port, q, break_direct):
    """bench.py's same-run A/B of the two bf16-wire exchanges (`pick_exchange`) on three gloo ranks with the torch
    stand-ins of the HIP kernels: both forms are timed in one process and the faster one is kept; with a direct path
    that raises (simulated RCCL failure inside the all-to-all) every rank falls back to the ring IN THIS PROCESS and
    training goes on with identical replicas."""
    import sys

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from wsovod_amd.engine import HotPathTrainer
    from wsovod_amd.engine.trainer import HipSGD
    from wsovod_amd.layers import hip_ops as H

    def pack(pairs):
        for src, dst in pairs:
            dst.copy_(src.reshape(-1))

    def sgd(entries, momentum, grad_scale=1.0, clip=None):
        for p, g, buf, shadow, lr, wd, used in entries:
            buf.mul_(momentum).add_(g.float().view_as(p) * grad_scale + wd * p)
            p.sub_(lr * buf)

    def sum_shards(src, n, dst):
        dst.copy_(src.view(n, -1).float().sum(0))
        return dst

    H.pack_bf16_multi, H.sgd_momentum_multi, H.sum_shards_bf16 = pack, sgd, sum_shards
    HotPathTrainer.split_on_cpu = True
    HotPathTrainer.split_rows = staticmethod(lambda n_rows, n_cols, cus=256, tile=256: 8)
    if break_direct:
        def broken(*a, **k):
            raise RuntimeError("simulated RCCL failure in all_to_all_single")
        dist.all_to_all_single = broken
    model = _SplitModel(cols=48)
    opt = HipSGD([{"params": [p], "lr": 0.01, "weight_decay": 0.0} for p in model.parameters()], 0.01, momentum=0.9)
    tr = HotPathTrainer(model, opt, overlap=True, grad_wire="bf16", exchange="ring")
    tr.broadcast_parameters()
    g = torch.Generator().manual_seed(7 + rank)
    batch = [{"x": torch.randn(48, generator=g)} for _ in range(4)]
    res = bench.pick_exchange(tr, lambda: tr.run_step(batch), dist.barrier, torch.device("cpu"), warm=2, steps=3)
    for _ in range(2):
        tr.run_step(batch)
    tr.flush()
    q.put((rank, res["chosen"], sorted(res["errors"]), {k: v is not None for k, v in res["ms"].items()}, tr.exchange_algo,
           model.big.detach().tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("break_direct", [False, True])
def test_bench_times_both_exchanges_and_falls_back_in_process(break_direct):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36500 + os.getpid() % 2000 + (7 if break_direct else 0)
    world = 3
    procs = [ctx.Process(target=_worker_pick_exchange, args=(r, world, port, q, break_direct)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len({r[1] for r in res}) == 1 and all(r[4] == r[1] for r in res)  # every rank chose (and runs) the same form
    assert all(r[5] == res[0][5] for r in res), "replicas diverged"
    if break_direct:
        assert res[0][1] == "ring" and res[0][2] == ["direct"] and res[0][3] == {"ring": True, "direct": False}
    else:
        assert res[0][1] in ("ring", "direct") and res[0][2] == [] and res[0][3] == {"ring": True, "direct": True}
