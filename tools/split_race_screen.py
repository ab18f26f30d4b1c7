import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from wsovod_amd.layers import hip_ops as H
g = torch.Generator(device="cuda").manual_seed(3)
side = torch.cuda.Stream(); junk = torch.empty(128 * 1024 * 1024, device="cuda")
bad = 0; n = 0
for (Mred, NI, NJ) in ((4096, 4096, 25088), (1024, 2304, 8448), (16384, 4096, 25088)):
    P = (torch.rand(Mred, NI, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
    Q = (torch.rand(Mred, NJ, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
    ref = H.gemm_tn(P, Q, split_tail=False)
    tol = 2e-6 * Mred ** 0.5 * 8
    for rep in range(40):
        if rep % 3 == 0:
            with torch.cuda.stream(side): junk.mul_(1.0001)
        out = H.gemm_tn(P, Q)
        err = float((out - ref).abs().max()); n += 1
        if not (err <= tol): bad += 1; print("MISMATCH", (Mred, NI, NJ), rep, err, tol)
    # split-K 8-phase NT
    A = (torch.rand(512, 25088, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
    B = (torch.rand(4096, 25088, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
    r2 = H.gemm_nt(A, B, out_dtype=torch.float32, tile_hint=8256256)
    for rep in range(40):
        o2 = H.gemm_nt(A, B, out_dtype=torch.float32)
        err = float((o2 - r2).abs().max()); n += 1
        if not (err <= 1e-2): bad += 1; print("MISMATCH splitk", rep, err)
print(n, "launches", bad, "mismatches")
