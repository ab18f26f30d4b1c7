"""Per-phase s_memtime ticks of the bf16x2 half-K 64-channel conv kernel (instrumented build:
`ABL_MACRO=C64XH_ABL bash tools/c64_ablate.sh 64`, then WSOVOD_LIB=.../abl/lib64.so python tools/c64x_phases.py)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import wsovod_amd._lib as _L
if os.environ.get("WSOVOD_LIB"):
    _L.LIB_PATH = os.environ["WSOVOD_LIB"]
from wsovod_amd.layers import hip_ops as H
n = 32
torch.manual_seed(0)
dbg = torch.zeros(512 * 8, device="cuda")
os.environ["WSOVOD_C64_DEBUG_PTR"] = hex(dbg.data_ptr())
for (Hh, Ww, pool, res) in ((300, 400, 0, 0), (300, 400, 2, 0), (150, 200, 0, 1), (150, 200, 2, 1)):
    x = H.x2_encode(torch.randn(n * Hh * Ww, 64, device="cuda")).view(n, Hh, Ww, 64)
    w = H.x2_encode(torch.randn(64, 9 * 64, device="cuda") * 0.05)
    b = torch.randn(64, device="cuda")
    r = H.x2_encode(torch.randn(n * Hh * Ww, 64, device="cuda")) if res else None
    geom = dict(n_img=n, H=Hh, W=Ww, Cin=64, Ho=Hh, Wo=Ww, KH=3, KW=3, stride=1, pad=1, dil=1, pool=pool)
    run = lambda: H.gemm_nt(x, w, conv=geom, x2=True, bias=b, relu=True, residual=r, residual_x2=bool(res), out_dtype=H.X2)
    run(); torch.cuda.synchronize(); dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        run()
    e1.record(); torch.cuda.synchronize()
    d = dbg.view(512, 8).cpu()
    cnt = float(d[:, 7].sum())
    m = d[:, :6].sum(0) / cnt
    names = ["prologue", "request issue (DMA + fragment reads)", "MFMA issue", "barrier (DMA + read wait)", "-", "epilogue"]
    print(f"{Hh}x{Ww} pool={pool} res={res}: {e0.elapsed_time(e1) / 3:.3f} ms/launch, {cnt / 3 / 512:.1f} tiles per slot; ticks per tile: " +
          ", ".join(f"{nm} {float(m[k]):.0f}" for k, nm in enumerate(names)) + f"; total {float(m.sum()):.0f}", flush=True)
