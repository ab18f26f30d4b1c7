"""Which backward tensor's rounding drives the 6.7e-3 of the `parity` precision's five-step trajectory?  

    python tools/parity_train_ablation.py > profiles/r06_parity_train_ablation.json      (one MI355X)

Runs tests/test_gpu_full_size.py's five-step trajectory (five optimizer steps at the reference's warm-up rates, 2 x 800x600 x
512 proposals, against the oracle taking the same five SGD steps) for:
  parity                         plain bf16 backward on the hi halves (dA rounded to bf16, x = hi half, W = bf16 shadow)
  parity_train, WSOVOD_PT_SPLIT=dw   only the weight-gradient contractions keep the hi/lo split (fp32 dA, decoded x)
  parity_train, WSOVOD_PT_SPLIT=dx   only the input-gradient contractions keep it (fp32 dA, fp32 W)
  parity_train                   both
and reports the logit / score deviation after the five updates, the worst trained-weight deviation relative to the distance the
oracle moved that tensor, and images/s of the mode at 2 images per step.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import compare as OC
from oracle import wsovod_ref as R
from wsovod_amd.data import make_batch
from wsovod_amd.engine import HotPathTrainer, build_optimizer
from wsovod_amd.testing import build_hot_path_model, capture_full_step


def main():
    gpu = torch.device("cuda", 0)
    steps = 5
    lrs = [0.01 * (0.001 * (1 - i / 200) + i / 200) for i in range(steps)]
    hosts = [make_batch(2, 512, 20, seed=900 + s) for s in range(steps + 1)]
    dev = lambda host: [{"image": x["image"].to(gpu), "proposals": x["proposals"].to(gpu), "instances": x["instances"],
                         "height": x["height"], "width": x["width"]} for x in host]
    oracle = {}

    def run(precision, split=None):
        if split is None:
            os.environ.pop("WSOVOD_PT_SPLIT", None)
        else:
            os.environ["WSOVOD_PT_SPLIT"] = split
        cfg, model = build_hot_path_model(seed=0, precision=precision, device="cuda:0")
        cfg.SOLVER.BASE_LR = lrs[0]
        model.train()
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.eval()
        sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
        train_keys = [k for k, p in model.named_parameters() if p.requires_grad]
        tr = HotPathTrainer(model, build_optimizer(cfg, model))
        tr.graph_max_batch = 0
        for s in range(steps):
            for grp in tr.optimizer.param_groups:
                grp["lr"] = lrs[s]
            tr.run_step(dev(hosts[s]))
        tr.flush()
        probe = capture_full_step(model, dev(hosts[steps]))
        trained = {k: v.detach().float().cpu().clone() for k, v in model.named_parameters() if v.requires_grad}
        # speed of the mode (eager launches, 2 images per step)
        b = dev(hosts[0])
        for _ in range(3):
            tr.run_step(b)
        tr.flush()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            tr.run_step(b)
        tr.flush()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        tr.close()
        del model, tr
        torch.cuda.empty_cache()
        if not oracle:
            params = {k: v.clone() for k, v in sd.items()}
            bufs = {}
            mom, wd = float(cfg.SOLVER.MOMENTUM), float(cfg.SOLVER.WEIGHT_DECAY)
            for s in range(steps):
                w = OC.oracle_step(params, hosts[s], train_keys)
                tp = {k: params[k] for k in train_keys}
                R.sgd_step(tp, {k: w["grads"][k] for k in train_keys}, bufs, lrs[s], mom, wd)
                params.update(tp)
            oracle.update(params=params, sd=sd, want=OC.oracle_step(params, hosts[steps], train_keys), keys=train_keys)
        rep = OC.compare(probe, oracle["want"])
        worst, worst_key = 0.0, None
        for k in oracle["keys"]:
            moved = float((oracle["params"][k] - oracle["sd"][k]).abs().max())
            err = float((trained[k] - oracle["params"][k]).abs().max())
            if moved > 1e-6 * float(oracle["params"][k].abs().max()) and err / moved > worst:  # (det.bias: true gradient 0)
                worst, worst_key = err / moved, k
        return {"max_abs_logit_err_after_5_steps": rep["max_abs_logit_err"], "max_abs_score_err": rep["max_abs_score_err"],
                "labels_exact": rep["labels_exact"], "pgt_exact": rep["pgt_exact"],
                "worst_trained_weight_err / distance_moved": round(worst, 5), "worst_tensor": worst_key,
                "ms_per_step_2_images_eager": round(ms, 3)}

    out = {"workload": "five optimizer steps, 2 x 800x600 x 512 proposals, WSR_18, reference warm-up rates, vs the oracle's five steps",
           "parity (plain bf16 backward)": run("parity"),
           "parity_train, split kept in dW only": run("parity_train", "dw"),
           "parity_train, split kept in dX only": run("parity_train", "dx"),
           "parity_train, split kept in dW and dX": run("parity_train", "dw,dx"),
           "parity_train (default = dX only)": run("parity_train")}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
