"""Run a few training steps of the hot path on cuda:0 and print losses / step time (dev probe)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsovod_amd.data import make_batch  # noqa: E402
from wsovod_amd.engine import build_optimizer, run_step  # noqa: E402
from wsovod_amd.testing import build_hot_path_model  # noqa: E402

for prec in ("fp32", "bf16"):
    for b in (1, 4):
        cfg, model = build_hot_path_model(precision=prec)
        model.train()
        opt = build_optimizer(cfg, model)
        data = make_batch(b, 512, 20)
        for it in range(4):
            torch.cuda.synchronize()
            t0 = time.time()
            ld = run_step(model, opt, data)
            torch.cuda.synchronize()
            dt = time.time() - t0
            print(prec, "b", b, "it", it, {k: round(float(v), 5) for k, v in ld.items()}, f"{dt*1e3:.1f} ms", flush=True)
        del model, opt
