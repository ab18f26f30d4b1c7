"""The device-side budget of the gradient exchange next to the window it hides behind.

    python tools/exchange_budget.py [precision] > profiles/r06_exchange_budget.json      (one MI355X, one process)

No 8-GPU node can be driven from here, so the WIRE time of DESIGN.md section 6 stays an estimate; everything the exchange
costs ON THE DEVICE is measurable on one GPU with a 1-rank RCCL group (`dist.init_process_group("nccl")`, world = 1: every
collective is issued, scheduled and executed by RCCL, it only has nobody to talk to):

  pack      all fp32 gradients -> their slices of the flat bf16 wire buffer (one launch, 249 MB for WSR_18)
  ring      dist.all_reduce of the buffer (what RCCL does at world = 1: its own copy / launch cost)
  direct    all_to_all_single -> sum_shards_bf16 -> all_gather_into_tensor (the point-to-point form)
  sgd_wire  the fused SGD update reading the reduced bf16 slices; sgd_fp32 = the same update from fp32 gradients
  window    the frozen forward (backbone + GAP + RoI pooling) of a step at 1 / 8 / 32 images per GPU: what the exchange of
            the previous step runs behind (`HotPathTrainer.stats()`: overlap_window_ms), with the exchange wait the compute
            stream actually saw (exchange_wait_ms) and the step time with and without the 1-rank group

A budget line per batch size: window - (pack is inside backward, not in the window) vs the expected wire time at N = 2 / 4 / 8
(bytes / per-link bandwidth) -> how much of the exchange is hidden if RCCL reaches the link rate.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def ev_time(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return {"median_ms": round(ts[len(ts) // 2], 4), "min_ms": round(ts[0], 4), "max_ms": round(ts[-1], 4)}


def main():
    precision = sys.argv[1] if len(sys.argv) > 1 else "parity"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from wsovod_amd.data import make_batch
    from wsovod_amd.engine import HotPathTrainer, build_optimizer
    from wsovod_amd.layers import hip_ops as H
    from wsovod_amd.testing import build_hot_path_model

    def to_dev(host):
        return [{"image": x["image"].to(dev), "proposals": x["proposals"].to(dev), "instances": x["instances"],
                 "height": x["height"], "width": x["width"]} for x in host]

    def steps_ms(tr, batch, n, warm):
        for _ in range(warm):
            tr.run_step(batch)
        tr.flush()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            tr.run_step(batch)
        tr.flush()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    out = {"precision": precision, "model": "WSR_18_DC5, 512 proposals / image, K = 20", "world": 1,
           "note": "1-rank RCCL group on one MI355X: device-side cost of every piece of the exchange; no wire time"}
    # ---- step time and windows WITHOUT a group (the N = 1 product path), per batch size
    nogroup = {}
    for b in (1, 8, 32):
        cfg, model = build_hot_path_model(seed=0, precision=precision, device="cuda:0")
        model.train()
        tr = HotPathTrainer(model, build_optimizer(cfg, model))
        batch = to_dev(make_batch(b, 512, 20, seed=1234))
        nogroup[b] = round(steps_ms(tr, batch, 20 if b < 32 else 10, 8), 4)
        tr.close()
        del tr, model, batch
        torch.cuda.empty_cache()
    out["step_ms_without_group"] = nogroup

    dist.init_process_group("nccl", device_id=dev)
    out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None
    per_batch = {}
    pieces = None
    for b in (1, 8, 32):
        for algo in ("ring", "direct"):
            cfg, model = build_hot_path_model(seed=0, precision=precision, device="cuda:0")
            model.train()
            tr = HotPathTrainer(model, build_optimizer(cfg, model), grad_wire="bf16", exchange=algo)
            batch = to_dev(make_batch(b, 512, 20, seed=1234))
            steps_ms(tr, batch, 4, 8)  # captures the step graphs at b <= 8, fills the allocator pools
            tr.stats_enable(True)
            ms = steps_ms(tr, batch, 20 if b < 32 else 10, 0)
            st = tr.stats()
            per_batch.setdefault(b, {})[algo] = {
                "step_ms_with_1_rank_group": round(ms, 4),
                "overlap_window_ms (frozen forward)": round(st["overlap_window_ms"], 4),
                "exchange_wait_ms (exposed on the compute stream)": round(st["exchange_wait_ms"], 4),
                "wire_bytes_per_step": st["wire_bytes_per_step"],
            }
            if pieces is None and b == 1:
                # ---- the pieces alone, on this trainer's own buffers
                flat, slices = tr._wire_slices()
                grads = [torch.randn_like(p, dtype=torch.float32) * 1e-3 for p in tr.params]
                pairs = [(g.reshape(-1), s) for g, s in zip(grads, slices)]
                pieces = {"wire_buffer_bytes": flat.numel() * 2,
                          "pack (fp32 gradients -> bf16 wire buffer, one launch)": ev_time(lambda: H.pack_bf16_multi(pairs)),
                          "ring: dist.all_reduce(buffer), world = 1": ev_time(lambda: dist.all_reduce(flat))}
                recv, mine = torch.empty_like(flat), torch.empty_like(flat)

                def direct():
                    dist.all_to_all_single(recv, flat)
                    H.sum_shards_bf16(recv, 1, mine)
                    dist.all_gather_into_tensor(flat, mine)
                pieces["direct: all_to_all_single + sum_shards_bf16 + all_gather_into_tensor, world = 1"] = ev_time(direct)

                def sgd(wire):
                    for p, g, s in zip(tr.params, grads, slices):
                        p._wire_grad = s if wire else None
                        p.grad = None if wire else g
                    tr.optimizer.step()
                    for p in tr.params:
                        p._wire_grad = None
                        p.grad = None
                pieces["sgd_wire (fused update reading the bf16 slices)"] = ev_time(lambda: sgd(True))
                pieces["sgd_fp32 (the same update from fp32 gradients: the N = 1 path)"] = ev_time(lambda: sgd(False))
                del grads, pairs, recv, mine
            tr.close()
            del tr, model, batch
            torch.cuda.empty_cache()
    out["pieces_alone"] = pieces
    out["per_images_per_gpu"] = per_batch
    # ---- the budget: expected wire time (249 MB, 153 GB/s per xGMI link and direction) vs the window
    wire = pieces["wire_buffer_bytes"]
    link = 153e9
    est = {f"ring N={n} (ONE ring on one link per direction: an upper bound, RCCL runs rings on several links)":
               2 * (n - 1) / n * wire / link * 1e3 for n in (2, 4, 8)}
    est["direct N=8 (every pair on its own link, 7 links at once)"] = 2 * (7 / 8) * wire / (7 * link) * 1e3
    est["direct N=4 (3 links at once)"] = 2 * (3 / 4) * wire / (3 * link) * 1e3
    out["expected_wire_ms (bytes / link rate; NOT measured: one GPU per call here)"] = {k: round(v, 3) for k, v in est.items()}
    budget = {}
    for b, d in per_batch.items():
        w = d["direct"]["overlap_window_ms (frozen forward)"]
        budget[b] = {"window_ms": w,
                     "device_side_exchange_ms (direct pieces at world = 1)":
                         pieces["direct: all_to_all_single + sum_shards_bf16 + all_gather_into_tensor, world = 1"]["median_ms"],
                     "hidden_if_wire_ms_below": round(w, 3),
                     "fits": {k: bool(v <= w) for k, v in est.items()}}
    out["budget"] = budget
    dist.destroy_process_group()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
