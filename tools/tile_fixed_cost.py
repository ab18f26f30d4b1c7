"""Fixed cost of a tile (prologue + epilogue + launch) of the lean 8-wavefront tile: ONE partly filled round of 256x256 tiles
at two K lengths, conv and plain GEMM, bf16x2 operands: T(K) = a + b K -> a.  python tools/tile_fixed_cost.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsovod_amd.layers import hip_ops as H

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timed(f, inner=10, rounds=7):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner * 1000)
    return sorted(ts)[len(ts) // 2]


bias = torch.randn(512, device=dev)
res = {}
for Cin in (128, 256, 512):
    n, Hi, Wi, Cout = 7, 75, 100, 256  # 206 x 1 tiles
    x = H.x2_encode(torch.randn(n * Hi * Wi, Cin, device=dev)).view(n, Hi, Wi, Cin)
    w = H.x2_encode(torch.randn(Cout, 9 * Cin, device=dev) * 0.05)
    geom = dict(n_img=n, H=Hi, W=Wi, Cin=Cin, Ho=Hi, Wo=Wi, KH=3, KW=3, stride=1, pad=2, dil=2)
    res[("conv", 9 * Cin)] = timed(lambda: H.gemm_nt(x, w, conv=geom, x2=True, bias=bias[:Cout], relu=True, out_dtype=H.X2, tile_hint=2256256))
for K in (1152, 2304, 4608):
    a = H.x2_encode(torch.randn(206 * 256, K, device=dev))
    b = H.x2_encode(torch.randn(256, K, device=dev) * 0.01)
    res[("gemm", K)] = timed(lambda: H.gemm_nt(a, b, x2=True, bias=bias[:256], relu=True, out_dtype=H.X2, tile_hint=2256256))
for kind in ("conv", "gemm"):
    t1, t2, t4 = res[(kind, 1152)], res[(kind, 2304)], res[(kind, 4608)]
    b = (t4 - t2) / 2304
    print(f"{kind}: K=1152 {t1:.1f} us, K=2304 {t2:.1f} us, K=4608 {t4:.1f} us -> per 64-value K-step {b * 64:.2f} us, fixed {t2 - b * 2304:.1f} us "
          f"(from 1152/2304: {t1 - (t2 - t1):.1f} us)")
# rounds: 206 tiles per round (7 images); T(rounds) = launch-level cost + rounds x (K-steps + per-tile fixed cost)
for Cin in (128, 256):
    ts = []
    for n in (7, 14, 28):
        Hi, Wi, Cout = 75, 100, 256
        x = H.x2_encode(torch.randn(n * Hi * Wi, Cin, device=dev)).view(n, Hi, Wi, Cin)
        w = H.x2_encode(torch.randn(Cout, 9 * Cin, device=dev) * 0.05)
        geom = dict(n_img=n, H=Hi, W=Wi, Cin=Cin, Ho=Hi, Wo=Wi, KH=3, KW=3, stride=1, pad=2, dil=2)
        ts.append(timed(lambda: H.gemm_nt(x, w, conv=geom, x2=True, bias=bias[:Cout], relu=True, out_dtype=H.X2, tile_hint=2256256)))
    print(f"conv K={9 * Cin}: 1 / 2 / 4 rounds of 206 tiles: {ts[0]:.1f} / {ts[1]:.1f} / {ts[2]:.1f} us -> per added round {(ts[2] - ts[1]) / 2:.1f} us, "
          f"launch-level {ts[0] - (ts[2] - ts[1]) / 2:.1f} us")
