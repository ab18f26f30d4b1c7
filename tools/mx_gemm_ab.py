"""The f16mx forward GEMM (wsovod_gemm_f16mx) against the bf16x2 three-product lean tile on the FC shapes (round 6).
    python tools/mx_gemm_ab.py [MxNxK,...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from wsovod_amd.layers import hip_ops as H

shapes = [(16384, 4096, 25088), (16384, 4096, 4096), (4096, 4096, 25088), (512, 4096, 25088)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1].split(",")]


def t(fn, n=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 3)
    return sorted(ts)[len(ts) // 2]


for M, N, K in shapes:
    a = torch.randn(M, K, device="cuda")
    b = torch.randn(N, K, device="cuda") * 0.01
    bias = torch.randn(N, device="cuda")
    ax, bx = H.x2_encode(a), H.x2_encode(b)
    am, sa = H.mx_encode(a)
    bm, sb = H.mx_encode(b)
    del a, b
    o1 = torch.empty(M, N, device="cuda")
    o2 = torch.empty(M, N, device="cuda")
    kw = dict(bias=bias, relu=True, dropout_p=0.5, dropout_seed=5)
    t_x2 = t(lambda: H.gemm_nt(ax, bx, x2=True, out=o1, out_dtype=H.X2, **kw))
    t_mx = t(lambda: H.gemm_mx(am, sa, bm, sb, out=o2, out_dtype=H.X2, **kw))
    fl = 2.0 * M * N * K
    d1, d2 = H.x2_decode(o1), H.x2_decode(o2)
    same_mask = bool(torch.equal(d1 == 0, d2 == 0))
    err = float((d1 - d2).abs().max() / d1.abs().max())
    print(f"{M}x{N}x{K}: bf16x2 {t_x2:.3f} ms ({fl / t_x2 / 1e9:.0f} TF algorithmic)   f16mx {t_mx:.3f} ms ({fl / t_mx / 1e9:.0f} TF)   "
          f"x{t_x2 / t_mx:.3f}   same dropout/ReLU pattern: {same_mask}   max |diff| / max = {err:.2e}", flush=True)
    del ax, bx, am, bm, o1, o2, d1, d2
    torch.cuda.empty_cache()
