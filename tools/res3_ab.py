"""res3's 128-channel 3x3 convs on bf16x2 maps: the 256x128 tile (the dispatcher's choice until round 5) against the
512x128 tile, interleaved rounds.  python tools/res3_ab.py [images] [tiles]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsovod_amd.layers import hip_ops as H
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
tiles = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 256128, 512128, 256256]
torch.manual_seed(0)
for (Cin, Cout, Hi, Wi, stride, res) in ((64, 128, 150, 200, 2, False), (128, 128, 75, 100, 1, True), (128, 128, 75, 100, 1, False)):
    Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
    x = H.x2_encode(torch.randn(n * Hi * Wi, Cin, device="cuda")).view(n, Hi, Wi, Cin)
    w = H.x2_encode(torch.randn(Cout, 9 * Cin, device="cuda") * 0.05)
    b = torch.randn(Cout, device="cuda")
    r = H.x2_encode(torch.randn(n * Ho * Wo, Cout, device="cuda")) if res else None
    geom = dict(n_img=n, H=Hi, W=Wi, Cin=Cin, Ho=Ho, Wo=Wo, KH=3, KW=3, stride=stride, pad=1, dil=1)
    run = lambda t: H.gemm_nt(x, w, conv=geom, x2=True, bias=b, relu=True, residual=r, residual_x2=bool(res), out_dtype=H.X2, tile_hint=t)
    outs, times = {}, {t: [] for t in tiles}
    for t in tiles:
        outs[t] = run(t)
    torch.cuda.synchronize()
    for rnd in range(7):
        for t in tiles:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run(t)
            e1.record(); torch.cuda.synchronize(); times[t].append(e0.elapsed_time(e1) / 3)
    fl = 6.0 * n * Ho * Wo * Cout * 9 * Cin
    print(f"conv {Cin}->{Cout} s{stride} res={int(res)}: " + "  ".join(
        f"{t}: {sorted(times[t])[3]:.3f} ms {fl / sorted(times[t])[3] / 1e9:.0f} TF eq={torch.equal(outs[t], outs[tiles[0]])}" for t in tiles), flush=True)
