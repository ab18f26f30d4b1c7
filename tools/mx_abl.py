"""Timing ablations of the f16mx tile (results meaningless in the ablated builds): WSOVOD_LIB=<abl lib> python tools/mx_abl.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import wsovod_amd._lib as _L
if os.environ.get("WSOVOD_LIB"):
    _L.LIB_PATH = os.environ["WSOVOD_LIB"]
from wsovod_amd.layers import hip_ops as H
M, N, K = 16384, 4096, 25088
a, sa = H.mx_encode(torch.randn(M, K, device="cuda")); b, sb = H.mx_encode(torch.randn(N, K, device="cuda") * 0.01)
bias = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda")
run = lambda: H.gemm_mx(a, sa, b, sb, bias=bias, relu=True, out=out, out_dtype=H.X2)
run(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        run()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 3)
ms = sorted(ts)[2]
print(f"{os.path.basename(os.environ.get('WSOVOD_LIB', 'product'))}: {ms:.3f} ms ({2.0 * M * N * K / ms / 1e9:.0f} TF)", flush=True)
