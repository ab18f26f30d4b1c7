"""Long race screen of the counted-wait kernels (8-phase NT, transposed-read TN): many launches per shape under
concurrent HBM traffic, every output compared bit for bit with the __syncthreads-ordered 16-wavefront tile."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsovod_amd.layers import hip_ops as H
g = torch.Generator(device="cuda").manual_seed(7)
side = torch.cuda.Stream()
junk = torch.empty(256 * 1024 * 1024, device="cuda")
shapes = [(8192, 4096, 4096), (4096, 4096, 25088), (2048, 2048, 64), (1024, 768, 200), (512, 512, 100000), (264, 520, 4096)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
t0 = time.time(); bad = 0; n = 0
for (M, N, K) in shapes:
    K8 = (K + 7) // 8 * 8
    A = (torch.rand(M, K8, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
    B = (torch.rand(N, K8, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
    ref = H.gemm_nt(A, B, out_dtype=torch.float32, tile_hint=256256)
    At, Bt = A.t().contiguous(), B.t().contiguous()
    for rep in range(reps):
        if rep % 3 == 0:
            with torch.cuda.stream(side):
                junk.mul_(1.0001)
        out = H.gemm_nt(A, B, out_dtype=torch.float32, tile_hint=8256256)
        tn = H.gemm_tn(At, Bt)
        bad += int(not torch.equal(out, ref)) + int(not torch.equal(tn, ref)); n += 2
    print((M, N, K), "ok so far, mismatches:", bad, flush=True)
print(f"{n} launches, {bad} mismatches, {time.time() - t0:.0f} s")
