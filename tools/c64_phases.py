"""Per-phase cycle counts of the register-weights 64-channel conv kernel (an ablation build with C64P_ABL=64:
`bash tools/c64_ablate.sh 64`, then WSOVOD_LIB=.../abl/lib64.so python tools/c64_phases.py)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import wsovod_amd._lib as _L
if os.environ.get("WSOVOD_LIB"):
    _L.LIB_PATH = os.environ["WSOVOD_LIB"]
from wsovod_amd.layers import hip_ops
n = 32
torch.manual_seed(0)
dbg = torch.zeros(256 * 8, device="cuda")
os.environ["WSOVOD_C64_DEBUG_PTR"] = hex(dbg.data_ptr())
for (H, W, pool, res) in ((300, 400, 0, 0), (300, 400, 2, 0), (150, 200, 0, 1), (150, 200, 2, 1)):
    x = (torch.rand(n, H, W, 64, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand(64, 9 * 64, device="cuda") * 2 - 1) * 0.05).to(torch.bfloat16)
    bias = torch.randn(64, device="cuda")
    r = torch.randn(n * H * W, 64, device="cuda").to(torch.bfloat16) if res else None
    geom = dict(n_img=n, H=H, W=W, Cin=64, Ho=H, Wo=W, KH=3, KW=3, stride=1, pad=1, dil=1, pool=pool)
    for _ in range(3):
        hip_ops.gemm_nt(x, w, conv=geom, bias=bias, relu=True, residual=r, out_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    d = dbg.view(256, 8).cpu()
    tiles = n * ((H + 7) // 8) * ((W + 31) // 32) / 256
    m = d.mean(0)
    names = ["mfma loop", "nop+vmcnt wait", "barrier", "first reads+epilogue", "residual+staging issue"]
    print(f"{H}x{W} pool={pool} res={res}: tiles/CU {tiles:.1f}; s_memtime ticks per tile: " +
          ", ".join(f"{nm} {float(m[k]) / tiles:.0f}" for k, nm in enumerate(names)) + f"; total {float(m[:5].sum()) / tiles:.0f}")
