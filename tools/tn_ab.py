"""dW contraction: transposed-read TN kernel vs (two transposes + NT 8-phase tile), interleaved rounds."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsovod_amd.layers import hip_ops as H
for (Mred, NI, NJ) in ((8192, 4096, 25088), (8192, 4096, 4096), (8192, 1024, 4096), (8192, 512, 1024)):
    P = (torch.rand(Mred, NI, device="cuda") * 2 - 1).to(torch.bfloat16)
    Q = (torch.rand(Mred, NJ, device="cuda") * 2 - 1).to(torch.bfloat16)
    out = torch.empty(NI, NJ, device="cuda")
    def tn(): H.gemm_tn(P, Q, out=out)
    def nt_only(pt, qt): H.gemm_nt(pt, qt, out=out, tile_hint=8256256)
    def nt_full():
        pt = H.transpose_cast(P, torch.bfloat16); qt = H.transpose_cast(Q, torch.bfloat16); H.gemm_nt(pt, qt, out=out, tile_hint=8256256)
    pt = H.transpose_cast(P, torch.bfloat16); qt = H.transpose_cast(Q, torch.bfloat16)
    variants = {"tn": tn, "nt_gemm_only": lambda: nt_only(pt, qt), "transposes+nt": nt_full}
    times = {k: [] for k in variants}
    for k, f in variants.items(): f()
    torch.cuda.synchronize()
    for r in range(7):
        for k, f in variants.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): f()
            e1.record(); torch.cuda.synchronize(); times[k].append(e0.elapsed_time(e1) / 3)
    fl = 2.0 * Mred * NI * NJ
    print(f"{Mred}x{NI}x{NJ}: " + "  ".join(f"{k} {sorted(v)[3]:.3f} ms ({fl / sorted(v)[3] / 1e9:.0f} TF)" for k, v in times.items()), flush=True)
