import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsovod_amd.layers import hip_ops
M = N = 4096
for K in (64, 128, 512):
    A = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16); B = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    for odt in (torch.float32, torch.bfloat16):
        for relu_bias in (False, True):
            bias = torch.randn(N, device="cuda") if relu_bias else None
            for t in (256256, 8256256):
                out = torch.empty(M, N, device="cuda", dtype=odt)
                ts = []
                for r in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        hip_ops.gemm_nt(A, B, out=out, tile_hint=t, bias=bias, relu=relu_bias)
                    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10)
                print(f"K={K} out={odt} bias+relu={relu_bias} tile {t}: {sorted(ts)[2]*1e3:.1f} us", flush=True)
