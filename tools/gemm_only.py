"""Run one GEMM shape/tile repeatedly (for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsovod_amd.layers import hip_ops  # noqa: E402

M, N, K, tile = (int(v) for v in sys.argv[1:5])
dt = torch.bfloat16 if (len(sys.argv) < 6 or sys.argv[5] == "bf16") else torch.float32
A = torch.randn(M, K, device="cuda").to(dt)
B = torch.randn(N, K, device="cuda").to(dt)
out = torch.empty(M, N, device="cuda", dtype=torch.float32)
for _ in range(6):
    hip_ops.gemm_nt(A, B, out=out, tile_hint=tile)
torch.cuda.synchronize()
