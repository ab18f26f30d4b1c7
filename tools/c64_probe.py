"""The 64-channel 3x3 halo-tile conv alone at the stem / res2 shapes (32 images); WSOVOD_LIB=<path> loads another
build of the library for A/B runs on one box."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import wsovod_amd._lib as _L
if os.environ.get("WSOVOD_LIB"):
    _L.LIB_PATH = os.environ["WSOVOD_LIB"]
from wsovod_amd.layers import hip_ops
n=32
torch.manual_seed(0)
SHAPES = ((300,400,0),(300,400,2),(150,200,0),(150,200,2))
if os.environ.get("C64_ONLY"):  # e.g. C64_ONLY=300,400,0 : one shape (counter passes)
    SHAPES = (tuple(int(v) for v in os.environ["C64_ONLY"].split(",")),)
for (H,W,pool) in SHAPES:
    x = (torch.rand(n, H, W, 64, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand(64, 9*64, device="cuda") * 2 - 1) * 0.05).to(torch.bfloat16)
    bias = torch.randn(64, device="cuda")
    geom = dict(n_img=n, H=H, W=W, Cin=64, Ho=H, Wo=W, KH=3, KW=3, stride=1, pad=1, dil=1, pool=pool)
    out = hip_ops.gemm_nt(x, w, conv=geom, bias=bias, relu=True, out_dtype=torch.bfloat16)
    ts=[]
    for r in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): hip_ops.gemm_nt(x, w, conv=geom, bias=bias, relu=True, out_dtype=torch.bfloat16)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/5)
    med=sorted(ts)[3]; fl=2.0*n*H*W*64*9*64
    chk = int(out.view(torch.int16).to(torch.int64).sum())  # bit-level checksum for old-vs-new comparisons
    print(f"c64 {H}x{W} pool={pool}: {med:.3f} ms {fl/med/1e9:.0f} TF  checksum {chk}", flush=True)
