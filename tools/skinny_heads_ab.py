"""A/B of the skinny head GEMMs on bf16x2 operands (N <= 64): the dispatcher's choice against the 64x64 grid.  python tools/skinny_heads_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsovod_amd.layers import hip_ops as H
dev = torch.device('cuda:0'); torch.manual_seed(0)
for M in (16384, 4096, 512):
    for N, K in ((40, 4096), (21, 512), (64, 4096)):
        a = H.x2_encode(torch.randn(M, K, device=dev)); b = H.x2_encode(torch.randn(N, K, device=dev) * 0.02); bias = torch.randn(N, device=dev)
        fns = {'auto': lambda: H.gemm_nt(a, b, x2=True, bias=bias, out_dtype=torch.float32), '64': lambda: H.gemm_nt(a, b, x2=True, bias=bias, out_dtype=torch.float32, tile_hint=64064)}
        ya, yb = fns['auto'](), fns['64']()
        err = float((ya - yb).abs().max()) / float(yb.abs().max())
        res = {}
        for k, f in fns.items():
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            res[k] = round(e0.elapsed_time(e1) / 20 * 1000, 1)
        print(M, N, K, res, 'relerr', err, flush=True)
