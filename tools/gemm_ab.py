"""A/B the GEMM tiles on the hot path's shapes: interleaved rounds in one process (random operands), plus a
correctness check of every tile against an fp32 torch matmul of the same bf16 operands."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsovod_amd.layers import hip_ops  # noqa: E402

tiles = [int(t) for t in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["256256", "8256256"])]
shapes = [(8192, 4096, 25088), (4096, 25088, 8192), (8192, 4096, 4096), (4096, 4096, 8192), (8192, 1024, 4096),
          (8000, 300, 1000)]
if len(sys.argv) > 2:
    shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[2].split(",")]
rounds = 7
for (M, N, K) in shapes:
    A = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    B = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    outs = {}
    ref = None
    if M * N <= 64 * 1024 * 1024:
        ref = A[:2048].float() @ B.float().t()
    times = {t: [] for t in tiles}
    for t in tiles:
        out = torch.empty(M, N, device="cuda", dtype=torch.float32)
        hip_ops.gemm_nt(A, B, out=out, tile_hint=t)
        torch.cuda.synchronize()
        if ref is not None:
            err = (out[:2048] - ref).abs().max().item() / ref.abs().max().item()
            assert err < 2e-3, (t, M, N, K, err)
        outs[t] = out
    for r in range(rounds):
        for t in tiles:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                hip_ops.gemm_nt(A, B, out=outs[t], tile_hint=t)
            e1.record()
            torch.cuda.synchronize()
            times[t].append(e0.elapsed_time(e1) / 3)
    base = None
    for t in tiles:
        med = sorted(times[t])[len(times[t]) // 2]
        tf = 2.0 * M * N * K / med / 1e9
        base = base or med
        same = torch.equal(outs[t], outs[tiles[0]])
        print(f"{M}x{N}x{K} tile {t}: median {med:.3f} ms  min {min(times[t]):.3f}  {tf:.0f} TF  x{base / med:.3f}  "
              f"bit-equal-to-first={same}", flush=True)
