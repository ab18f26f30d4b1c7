import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from wsovod_amd.layers import hip_ops as H
dev = torch.device("cuda:0")
def bench(fns, rounds=5, inner=10):
    times = {k: [] for k in fns}
    for k, f in list(fns.items()):
        try: f()
        except RuntimeError as e:
            del fns[k], times[k]
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(inner): f()
            e1.record(); torch.cuda.synchronize(); times[k].append(e0.elapsed_time(e1) / inner)
    return {k: sorted(v)[len(v) // 2] for k, v in times.items()}
for (M, N, K) in ((16384, 44, 4096), (16384, 40, 4096), (16384, 4, 4096), (16384, 24, 4096)):
    a = H.x2_encode(torch.randn(M, K, device=dev)); b = H.x2_encode(torch.randn(N, K, device=dev) * 0.01)
    bias = torch.randn(N, device=dev)
    fns = {t: (lambda t=t: H.gemm_nt(a, b, x2=True, bias=bias, out_dtype=torch.float32, tile_hint=t)) for t in (0, 64064, 1128064, 1256064)}
    r = bench(fns)
    print(M, N, K, {t: f"{ms*1e3:.0f}us" for t, ms in r.items()}, flush=True)
