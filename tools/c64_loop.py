import os, sys, torch, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from wsovod_amd.layers import hip_ops
n, H, W = 32, 300, 400
x = (torch.rand(n, H, W, 64, device="cuda") * 2 - 1).to(torch.bfloat16)
w = ((torch.rand(64, 9 * 64, device="cuda") * 2 - 1) * 0.05).to(torch.bfloat16)
bias = torch.randn(64, device="cuda")
geom = dict(n_img=n, H=H, W=W, Cin=64, Ho=H, Wo=W, KH=3, KW=3, stride=1, pad=1, dil=1, pool=0)
out = hip_ops.gemm_nt(x, w, conv=geom, bias=bias, relu=True, out_dtype=torch.bfloat16)
t0 = time.time()
while time.time() - t0 < float(sys.argv[1]):
    for _ in range(200):
        hip_ops.gemm_nt(x, w, conv=geom, bias=bias, relu=True, out=out)
    torch.cuda.synchronize()
