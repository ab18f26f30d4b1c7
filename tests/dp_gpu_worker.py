"""Worker of tests/test_gpu_data_parallel.py: one data-parallel rank of the REAL HIP model.  Two of these share
cuda:0 on the 1-GPU box (gloo backend; the collective code path of HotPathTrainer is the same one RCCL runs).

    python -m tests.dp_gpu_worker <rank> <world> <port> <out.pt> <mode>      mode: single | mixed | direct | mx
(direct = single with exchange="direct": all-to-all + fp32 shard sum + all-gather instead of the all-reduce;
 mx = single in the "parity_mx" precision -- the launcher lowers the f16mx thresholds through the environment)
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

STEPS, LR = 5, 1e-4  # (the whole-step HIP graph takes over on the third step: two eager steps, a capture, two replays)


def sample(t, n=65536):
    flat = t.detach().reshape(-1)
    if flat.numel() <= n:
        return flat.float().cpu().clone()
    idx = (torch.arange(n, dtype=torch.int64, device=flat.device) * (flat.numel() - 1)) // (n - 1)
    return flat[idx].float().cpu()


def fingerprint(t):
    """Order-independent exact fingerprint of the bits of a tensor (replicas must agree bit for bit)."""
    return int(t.detach().contiguous().view(torch.int32).to(torch.int64).sum().item())


def shard(rank, mode, it=0):
    from tests.golden import gen
    from tests.helpers import to_inputs

    if mode == "mixed":
        src = (2, 0)[(rank + it) % 2]  # the two ranks draw from DIFFERENT datasets in every step
        K = (20, 20, 80)[src]
        batch = gen.seeded_batch(2, 32, K, 256, 352, seed=50 + 10 * rank + it)
        for b in batch:
            b["dataset_id"] = src
        return to_inputs(batch)
    return to_inputs(gen.seeded_batch(2, 32, 20, 256, 352, seed=6 + rank))  # single, direct


def build(mode):
    from tests.golden import gen
    from tests.helpers import build_seeded_hip_model

    if mode == "mixed":
        from wsovod_amd.modeling import build_model
        from wsovod_amd.testing import mixed_datasets_cfg

        cfg = mixed_datasets_cfg(Ks=(20, 20, 80), precision="bf16", device="cuda:0")
        torch.manual_seed(0)
        model = build_model(cfg)
        model._std = [float(v) for v in gen.PIXEL_STD]
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict(gen.mixed_seeded_state(shapes, seed=17), strict=True)
        model.train()
    else:
        cfg, model, _ = build_seeded_hip_model("parity_mx" if mode == "mx" else "bf16")
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    cfg.SOLVER.BASE_LR = LR
    return cfg, model


def main():
    import faulthandler

    faulthandler.dump_traceback_later(240, exit=True)  # a hung rank says where (the launcher prints the log on a timeout)
    rank, world, port, out, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wsovod_amd.engine import HotPathTrainer, build_optimizer

    calls = []
    orig = dist.all_reduce

    def traced(t, *a, **k):
        calls.append((str(t.dtype), t.numel()))
        return orig(t, *a, **k)

    dist.all_reduce = traced
    for name in ("all_to_all_single", "all_gather_into_tensor"):
        def wrap(out_t, in_t, *a, _f=getattr(dist, name), _n=name, **k):
            calls.append((_n, in_t.numel()))
            return _f(out_t, in_t, *a, **k)
        setattr(dist, name, wrap)
    cfg, model = build(mode)
    tr = HotPathTrainer(model, build_optimizer(cfg, model), grad_wire="bf16", reduce_unused=(mode == "mixed"),
                        exchange="direct" if mode == "direct" else "ring")
    assert tr._split is not None, "the early fc1 block must be active (bf16 wire, TN weight gradient)"
    tr.broadcast_parameters()
    early = 0
    for it in range(STEPS):
        tr.run_step(shard(rank, mode, it))
        early += int(tr._pending is not None and len(tr._pending) >= 2)
    sd = model.state_dict()  # the pre-hook applies the last update
    torch.cuda.synchronize()
    params = {k: v for k, v in model.named_parameters() if v.requires_grad}
    torch.save({"rank": rank, "calls": calls, "early_steps": early,
                "fingerprint": {k: fingerprint(v) for k, v in params.items()},
                "sample": {k: sample(v) for k, v in params.items()},
                "state_keys": len(sd)}, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
