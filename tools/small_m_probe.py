"""Tile choice for the FC contractions at small per-GPU batches (M = 512 * images): time per tile hint."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsovod_amd.layers import hip_ops as H
hints = [0, 64064, 128128, 1128128, 3128128, 4128128, 256128, 3256128, 256256, 8256256]
for (M, N, K) in ((512, 4096, 25088), (512, 4096, 4096), (512, 1024, 4096), (1024, 4096, 25088), (2048, 4096, 25088),
                  (4096, 4096, 25088)):
    A = (torch.rand(M, K, device="cuda") - 0.5).to(torch.bfloat16)
    B = (torch.rand(N, K, device="cuda") - 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = []
    for t in hints:
        try:
            f = lambda: H.gemm_nt(A, B, out=out, bias=bias, relu=True, tile_hint=t)
            for _ in range(3): f()
            torch.cuda.synchronize()
            ts = []
            for r in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); f(); f(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 3)
            res.append((t, sorted(ts)[2]))
        except Exception as e:
            res.append((t, float("nan")))
    gb = (M * K + N * K + M * N) * 2 / 1e9
    print((M, N, K), "  ".join(f"{t}:{ms*1e3:.0f}us" for t, ms in res), f"| bytes {gb*1e3:.0f} MB", flush=True)
