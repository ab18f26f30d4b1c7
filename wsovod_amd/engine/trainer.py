"""Training-step plumbing of the hot path: optimizer, DDP wrap, run_step.

Mirrors /root/reference/wsovod/engine/trainer.py:37-84 (`run_step`: forward, sum of the loss dict,
backward, optimizer step every ITER_SIZE) and engine/defaults.py:135-153,274-318
(`wrap_model_with_ddp`, `build_optimizer`: one param group per tensor, SGD momentum).  The
parameter update itself is the fused HIP kernel `wsovod_sgd_momentum`; gradient exchange is
torch DistributedDataParallel over RCCL ("nccl" backend on ROCm) -- the only collective on the path.
"""
import torch
import torch.distributed as dist

from ..layers import hip_ops as H


class HipSGD(torch.optim.Optimizer):
    """torch.optim.SGD semantics (momentum, weight decay, dampening 0) on wsovod_sgd_momentum."""

    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                state = self.state[p]
                if "momentum_buffer" not in state:
                    state["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                H.sgd_momentum(p.data, g, state["momentum_buffer"], group["lr"], group["momentum"],
                               group["weight_decay"])


def build_optimizer(cfg, model):
    """engine/defaults.py:274-318: every trainable tensor is its own group with BASE_LR / WEIGHT_DECAY."""
    params, memo = [], set()
    for key, value in model.named_parameters(recurse=True):
        if not value.requires_grad or value in memo:
            continue
        memo.add(value)
        lr = cfg.SOLVER.BASE_LR * (cfg.SOLVER.BACKBONE_MULTIPLIER if "backbone" in key else 1.0)
        params.append({"params": [value], "lr": lr, "weight_decay": cfg.SOLVER.WEIGHT_DECAY})
    if cfg.SOLVER.OPTIMIZER != "SGD":
        raise NotImplementedError("hot path optimizer is SGD (every WSR config)")
    return HipSGD(params, cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM)


def wrap_model_with_ddp(model, local_rank, find_unused_parameters=False, bucket_cap_mb=128):
    """engine/defaults.py:135-153.  Single-dataset mode touches every parameter each step, so the
    unused-parameter graph walk of the reference is unnecessary."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return model
    from torch.nn.parallel import DistributedDataParallel

    kw = dict(broadcast_buffers=False, find_unused_parameters=find_unused_parameters, bucket_cap_mb=bucket_cap_mb,
              gradient_as_bucket_view=True)
    if next(model.parameters()).is_cuda:
        kw["device_ids"] = [local_rank]
    return DistributedDataParallel(model, **kw)


def run_step(model, optimizer, data, iter_size=1, it=0):
    """One iteration of DefaultTrainer_WSOVOD.run_step.  Returns the loss dict (device tensors)."""
    loss_dict = model(data)
    losses = sum(loss_dict.values())
    if iter_size > 1:
        losses = losses / iter_size
    losses.backward()
    if it % iter_size == 0:
        optimizer.step()
        optimizer.zero_grad(set_to_none=True)
    return loss_dict
