#!/usr/bin/env python3
"""Benchmark of the WSOVOD hot path on MI355X (contract: see the task statement / DESIGN.md).

One "step" = one full training iteration of WSOVOD_WSR_18_DC5 in proposals-only mode on a batch of
synthetic 800x600 images with 512 proposals each: uint8 image -> fused normalise+stem -> frozen
backbone -> RoIPool(+objectness) -> neck -> object mining (MIL) -> pseudo-GT mining/labelling ->
instance refinement (cosine-similarity head) -> losses -> backward -> fused SGD step
(reference: wsovod/engine/trainer.py:37-84).  Inputs are resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0.  `value` = images/sec over all N GPUs (weak scaling: per-GPU batch fixed).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK = {"bf16": 2500.0, "fp32": 157.3}  # dense MFMA TFLOP/s (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for debugging)")
    ap.add_argument("--share-device", action="store_true",
                    help="debug: every rank uses cuda:0 (lets the N>1 code path run on a 1-GPU box with --backend gloo)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--depth", type=int, default=18)
    ap.add_argument("--proposals", type=int, default=512)
    ap.add_argument("--classes", type=int, default=20)
    ap.add_argument("--embed-dim", type=int, default=512)
    ap.add_argument("--pooler", default="ROIPool")
    ap.add_argument("--rpn", action="store_true",
                    help="the shipped form of the config: RPN branch on next to the loaded proposals (SURVEY 8f n1); "
                         "not the north-star workload, so no CPU baseline is taken")
    ap.add_argument("--grad-wire", default="auto", choices=["auto", "fp32", "bf16"],
                    help="gradient all-reduce format at N>1: auto = the compute precision (bf16 run -> bf16 wire, the "
                         "counterpart of the reference's fp16 compression hook; master weights/momentum stay fp32)")
    ap.add_argument("--one-rank-group", action="store_true",
                    help="debug/measurement: at N=1 still create a 1-rank RCCL group, so that the whole exchange path "
                         "(gradient pack, split weight-gradient launch, collectives, SGD on the wire slices) runs and "
                         "its cost without any wire time can be read off one GPU")
    ap.add_argument("--eval", action="store_true",
                    help="side measurement, not the headline metric: images/s of model.inference() (eval mode: frozen "
                         "backbone, RoI pooling, heads, score threshold + per-class NMS tail) on the same synthetic batch")
    ap.add_argument("--launch-check", action="store_true",
                    help="launcher self-test (no GPU work): every rank joins the group, all-reduces its rank and rank 0 "
                         "prints {n_gpus, ranks_seen}; with --backend gloo it runs on a CPU-only host (tests/test_bench_launcher.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    return ap.parse_args()


def pmc_traffic(kernel_name, args):
    """HBM/fabric bytes per launch of `kernel_name` from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE cannot be collected from inside this process).  Only returned when the run matches the
    profiled workload; otherwise null."""
    import re

    path = os.path.join(ROOT, "profiles", f"r01_pmc_traffic_b{args.batch}_bf16.json")  # written by tools/collect_profiles.sh
    if not (os.path.exists(path) and args.precision == "bf16" and args.depth == 18 and args.proposals == 512):
        return None
    if getattr(args, "rpn", False):
        return None
    m = re.match(r"(gemm_nt|conv_igemm)_(bf16|f32)_(\d+)x(\d+)(_dma|_8ph)?$", kernel_name)
    if not m:
        return None
    conv = m.group(1) == "conv_igemm"
    if m.group(5) == "_8ph":  # rocprofv3 prints this one demangled
        tags = ("gemm256_8ph_kernelILb%d" % conv, "gemm256_8ph_kernel<%s>" % ("true" if conv else "false"))
    else:
        tags = ("gemm_nt_kernelI%sLi%sELi%sELb%d" % ("DF16b" if m.group(2) == "bf16" else "f", m.group(3), m.group(4),
                                                    conv),)
    with open(path) as f:
        table = json.load(f)["kernels"]
    hits = [v for k, v in table.items() if any(t in k for t in tags)]
    return hits[0]["traffic_bytes_per_launch"] if len(hits) == 1 else None


def to_device_batch(batch, dev):
    out = []
    for x in batch:
        out.append({"image": x["image"].to(dev), "proposals": x["proposals"].to(dev),
                    "instances": x["instances"],  # image-level labels stay on the host (no sync to read them)
                    "height": x["height"], "width": x["width"]})
    return out


def cpu_baseline(model, sd, batch, args):
    """The oracle (CPU restatement of the reference path) timed on the host cores: baseline only."""
    from oracle import wsovod_ref as R

    host_cores = os.cpu_count() or 1
    train_keys = [k for k, p in model.named_parameters() if p.requires_grad]
    for k in train_keys:
        sd[k].requires_grad_(True)
    sample = R.batch_from_inputs(batch[:1])
    bufs = {}

    def step():
        losses, _ = R.train_forward(sd, sample, depth=args.depth, num_classes=args.classes, pooler_type=args.pooler)
        total = sum(losses.values())
        grads = torch.autograd.grad(total, [sd[k] for k in train_keys], allow_unused=True)
        with torch.no_grad():
            for k, g in zip(train_keys, grads):
                if g is None:
                    continue
                g = g + 5e-4 * sd[k]
                bufs[k] = g if k not in bufs else bufs[k].mul_(0.9).add_(g)
                sd[k].sub_(0.01 * bufs[k])

    # pick the thread count the host runs this path fastest at (all cores oversubscribe the small ops)
    best = None
    for nt in sorted({host_cores, min(host_cores, 64), min(host_cores, 32), min(host_cores, 16)}, reverse=True):
        torch.set_num_threads(nt)
        if best is None:
            step()  # warm-up (also builds/loads the C oracle)
        t0 = time.time()
        step()
        dt = time.time() - t0
        if best is None or dt < best[1]:
            best = (nt, dt)
    ncores = best[0]
    torch.set_num_threads(ncores)
    n, t0 = 0, time.time()
    while True:
        step()
        n += 1
        if time.time() - t0 > args.cpu_seconds or n >= 8:
            break
    dt = time.time() - t0
    return {"value": n / dt, "unit": "images/sec", "cores": ncores, "kind": "port",
            "sample": f"{n} full fp32 training steps of 1 image x {args.proposals} proposals (oracle/wsovod_ref.py, "
                      f"torch {torch.__version__} CPU, {ncores} of {host_cores} host threads = fastest of 16/32/64/all; RoIPool = single-thread C oracle)"}


def eval_bench(args, cfg, model, dev, result_fd):
    """Inference throughput (1 GPU): model.inference(batch) with the class text embeddings given per call."""
    from wsovod_amd.data import make_batch

    model.eval()
    batch = to_device_batch(make_batch(args.batch, args.proposals, args.classes, seed=1234), dev)
    g = torch.Generator().manual_seed(7)
    classifier = torch.randn(args.classes, args.embed_dim, generator=g).to(dev)
    for _ in range(args.warmup):
        out = model.inference(batch, classifier=classifier)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = model.inference(batch, classifier=classifier)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t1 = time.perf_counter()
    for _ in range(args.steps):
        model.inference(batch, classifier=classifier, do_postprocess=False)
    torch.cuda.synchronize()
    elapsed_np = time.perf_counter() - t1
    res = {"metric": "inference images/sec (side measurement)", "value": args.batch * args.steps / elapsed,
           "unit": "images/sec", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "dtype": args.precision,
           "data": "synthetic",
           "config": {"workload": f"WSR_{args.depth}_DC5 inference, {args.proposals} proposals/img, {args.classes} classes, "
                                  f"{args.pooler}, score threshold + per-class NMS + top-100 tail", "images_per_step": args.batch,
                      "ms_per_step_without_detector_postprocess": elapsed_np / args.steps * 1e3,
                      "detections_first_image": int(len(out[0]["instances"]))}}
    os.write(result_fd, (json.dumps(res) + "\n").encode())
    return 0


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks here, one process per GPU, as
    the reference's `launch(main, num_gpus)` does (/root/reference/tools/train_net.py:80-90).  The parent never
    touches the GPU (no HIP call, no exec after one): it only spawns children, relays rank 0's JSON line and
    returns non-zero if any rank fails."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WSOVOD_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    rc, line = 0, b""
    try:
        pending = set(range(args.gpus))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if r == 0:
                    line = procs[0].stdout.read()
                if code != 0:
                    rc = rc or code
                    print(f"bench.py: rank {r} exited with code {code}", file=sys.stderr)
            if rc:  # a dead rank leaves the others waiting in a collective: stop exactly the children started here
                for r in pending:
                    procs[r].terminate()
                for r in pending:
                    try:
                        procs[r].wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        procs[r].kill()
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    lines = [x for x in line.decode().splitlines() if x.strip()]
    if rc == 0 and (len(lines) != 1 or json.loads(lines[0]).get("n_gpus") != args.gpus):
        print(f"bench.py: expected one JSON line with n_gpus={args.gpus} from rank 0, got {lines!r}", file=sys.stderr)
        rc = 1
    if rc == 0:
        sys.stdout.write(lines[0] + "\n")
        sys.stdout.flush()
    return rc


def launch_check(args, world, rank, result_fd):
    """The launcher path without GPU work: group creation, one all-reduce, one JSON line."""
    if os.environ.get("WSOVOD_BENCH_FAIL_RANK") == str(rank):  # tests: a rank that dies before joining the group
        raise SystemExit(3)
    dist.init_process_group(args.backend)
    t = torch.tensor([1 << rank], dtype=torch.int64)
    if args.backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        t = t.cuda()
    dist.all_reduce(t)
    if rank == 0:
        os.write(result_fd, (json.dumps({"metric": "launch-check", "n_gpus": dist.get_world_size(),
                                         "ranks_seen": int(t.item()), "backend": args.backend}) + "\n").encode())
    dist.barrier()
    dist.destroy_process_group()
    return 0


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            return launch_ranks(args)
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: the launcher and "
                         f"the flag must agree (the reported n_gpus is the size of the process group)")
    # stdout carries exactly ONE line, the JSON result: libraries that print to fd 1 (RCCL's version banner at
    # communicator creation does) are sent to stderr for the duration of the run.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.launch_check:
        return launch_check(args, world, rank, result_fd)
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X (no CPU fallback on the product path)")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.one_rank_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from wsovod_amd import _lib
    from wsovod_amd.data import make_batch
    from wsovod_amd.engine import HotPathTrainer, build_optimizer
    from wsovod_amd.testing import build_hot_path_model

    cfg, model = build_hot_path_model(seed=0, depth=args.depth, K=args.classes, D=args.embed_dim,
                                      precision=args.precision, pooler=args.pooler, device=f"cuda:{local_rank}",
                                      rpn=args.rpn)
    if args.rpn:
        args.no_cpu_baseline = True
        model.roi_heads.iter = cfg.SOLVER.MAX_ITER // 2  # mid-training objectness ramp (rcnn_wsovod.py:181-184)
    if args.eval:
        return eval_bench(args, cfg, model, dev, result_fd)
    model.train()
    optimizer = build_optimizer(cfg, model)
    wire = args.precision if args.grad_wire == "auto" else args.grad_wire
    # async gradient all-reduce overlapped with the next step's frozen forward
    trainer = HotPathTrainer(model, optimizer, grad_wire=wire)
    trainer.broadcast_parameters()

    def run_step(_m, _o, data):
        return trainer.run_step(data)

    ddp = model
    cpu_state = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # snapshot of the untrained weights for the CPU leg
        cpu_state = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    host_batch = make_batch(args.batch, args.proposals, args.classes, seed=1234 + rank)
    batch = to_device_batch(host_batch, dev)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run_step(ddp, optimizer, batch)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step(ddp, optimizer, batch)
    sync()
    elapsed = time.perf_counter() - t0
    last = run_step(ddp, optimizer, batch)  # outside the timed region: the run must have stayed finite
    final_losses = {k: float(v.detach()) for k, v in last.items()}
    if not all(v == v and abs(v) != float("inf") for v in final_losses.values()):
        raise RuntimeError(f"training diverged during the benchmark: {final_losses}")
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    roofline = None
    if not args.no_roofline:
        # second pass over the same K steps with every launch bracketed by hipEvents on its stream
        _lib.profile_reset()
        _lib.profile_enable(True)
        for _ in range(args.steps):
            run_step(ddp, optimizer, batch)
        torch.cuda.synchronize()
        table = _lib.profile_collect()
        _lib.profile_enable(False)
        table = [e for e in table if e["launches"] > 0]
        table.sort(key=lambda e: -e["ms"])
        if rank == 0 and table:
            top = table[0]
            mfma = top["flops"] > 0 and ("gemm" in top["name"] or "conv" in top["name"])
            avg_ms = top["ms"] / top["launches"]
            if mfma:
                ach = top["flops"] / top["launches"] / (avg_ms * 1e-3) / 1e12
                roofline = {"bound": "mfma", "achieved": ach, "peak": PEAK[args.precision], "unit": "TFLOP/s",
                            "frac": ach / PEAK[args.precision], "traffic": None}
            else:
                ach = top["bytes"] / top["launches"] / (avg_ms * 1e-3) / 1e9
                roofline = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": ach / HBM_PEAK_GBS, "traffic": None}
            roofline["traffic"] = pmc_traffic(top["name"], args)
            roofline["algorithmic_per_launch"] = (top["flops"] if mfma else top["bytes"]) / top["launches"]
            roofline.update(kernel=top["name"], launches_per_step=top["launches"] / args.steps,
                            avg_launch_ms=avg_ms, share_of_kernel_time=top["ms"] / sum(e["ms"] for e in table),
                            kernels=[{"name": e["name"], "ms_per_step": e["ms"] / args.steps,
                                      "launches_per_step": e["launches"] / args.steps,
                                      "tflops": (e["flops"] / (e["ms"] * 1e-3) / 1e12) if e["flops"] and e["ms"] else None,
                                      "gbs": (e["bytes"] / (e["ms"] * 1e-3) / 1e9) if e["bytes"] and e["ms"] else None}
                                     for e in table[:12]])

    if rank == 0:
        images = world * args.batch * args.steps
        out = {
            "metric": "images/sec (512 proposals/img) WSR_18_DC5 fwd+bwd at 1/2/4/8 MI355X",
            "value": images / elapsed, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"VOC07 WSOVOD_WSR_{args.depth}_DC5_1x, {args.proposals} proposals/img, "
                                   f"{args.classes}-class embeddings (D={args.embed_dim}), 800x600 images, "
                                   f"{'RPN + loaded proposals' if args.rpn else 'proposals-only mode'}, {args.pooler}, "
                                   f"full training step (fwd+bwd+SGD)",
                       "images_per_gpu_per_step": args.batch, "global_batch": world * args.batch,
                       "parallelism": f"dp{world}", "grad_allreduce": f"{wire} over RCCL" if dist.is_initialized() else "none (1 GPU)",
                       "final_losses": final_losses},
        }
        if roofline is not None:
            out["roofline"] = roofline
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(model, cpu_state, host_batch, args)
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
