"""Which ATen operators still run inside one steady-state training step, and from where?  

    python tools/aten_trace.py [--precision bf16|parity] [--batch 32] > gpurun_out/aten_trace.txt

Runs the bench's step (HotPathTrainer on resident synthetic inputs) a few times, then once more under a
TorchDispatchMode that records every ATen call together with the innermost frame inside wsovod_amd/ that issued it
(the autograd engine's worker thread inherits the mode, so backward calls are seen as well).  Calls that launch no
kernel (views, metadata) are filtered by name.  The library's own kernels do not appear: they are ctypes calls.
"""
import argparse
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from wsovod_amd.data import make_batch
from wsovod_amd.engine import HotPathTrainer, build_optimizer
from wsovod_amd.testing import build_hot_path_model

VIEWS = {"view", "_unsafe_view", "reshape", "permute", "transpose", "t", "slice", "select", "expand", "as_strided",
         "detach", "alias", "unsqueeze", "squeeze", "narrow", "split", "split_with_sizes", "chunk", "unbind", "size",
         "stride", "is_contiguous", "_local_scalar_dense", "empty", "empty_like", "empty_strided", "new_empty",
         "lift_fresh", "record_stream", "is_pinned", "unfold", "view_as_real", "resize_", "set_", "new_empty_strided",
         "is_nonzero", "sym_size", "sym_stride", "sym_numel", "is_same_size", "_reshape_alias", "unsafe_split", "flatten"}


class Trace(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.calls = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        if name not in VIEWS:
            site = "?"
            for fr in reversed(traceback.extract_stack()):
                if "wsovod_amd" in fr.filename or fr.filename.endswith("bench.py"):
                    site = f"{os.path.relpath(fr.filename)}:{fr.lineno} {fr.name}"
                    break
            shapes = tuple(tuple(a.shape) if isinstance(a, torch.Tensor) else None for a in args[:2])
            on_gpu = any(isinstance(a, torch.Tensor) and a.is_cuda for a in args) or "device" in (kwargs or {})
            self.calls[(name, site, shapes, on_gpu)] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg, model = build_hot_path_model(seed=0, depth=18, K=20, D=512, precision=a.precision, device=str(dev))
    model.train()
    cfg.SOLVER.BASE_LR = 1e-3
    trainer = HotPathTrainer(model, build_optimizer(cfg, model), grad_wire="bf16")
    trainer.broadcast_parameters()
    host = make_batch(a.batch, 512, 20, seed=1234)
    batch = [{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in x.items()} for x in host]
    for x, h in zip(batch, host):
        x["proposals"] = h["proposals"].to(dev)
    for _ in range(4):
        trainer.run_step(batch)
    torch.cuda.synchronize()
    tr = Trace()
    with tr:
        trainer.run_step(batch)
    torch.cuda.synchronize()
    n = 0
    for (name, site, shapes, on_gpu), c in sorted(tr.calls.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        print(f"{c:3d}  {name:28s} {'gpu' if on_gpu else 'cpu'}  {site:70s} {shapes}")
        n += c
    print("total ATen calls (non-view):", n)


if __name__ == "__main__":
    main()
