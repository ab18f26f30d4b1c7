"""res4 / res5 implicit-GEMM convs on bf16x2 maps under given tiles, interleaved rounds.  python tools/conv_x2_ab.py [images] [tiles]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wsovod_amd._lib as _L
if os.environ.get("WSOVOD_LIB"):  # another build of the library (ablation builds) for an A/B on one box
    _L.LIB_PATH = os.environ["WSOVOD_LIB"]
from wsovod_amd.layers import hip_ops as H
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
tiles = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [256256, 9256256]
torch.manual_seed(0)
for (Cin, Cout, dil) in ((256, 256, 2), (512, 512, 2), (256, 512, 2)):
    Hi, Wi = 75, 100
    x = H.x2_encode(torch.randn(n * Hi * Wi, Cin, device="cuda")).view(n, Hi, Wi, Cin)
    w = H.x2_encode(torch.randn(Cout, 9 * Cin, device="cuda") * 0.05)
    b = torch.randn(Cout, device="cuda")
    geom = dict(n_img=n, H=Hi, W=Wi, Cin=Cin, Ho=Hi, Wo=Wi, KH=3, KW=3, stride=1, pad=dil, dil=dil)
    run = lambda t: H.gemm_nt(x, w, conv=geom, x2=True, bias=b, relu=True, out_dtype=H.X2, tile_hint=t)
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
    fl = 6.0 * n * Hi * Wi * Cout * 9 * Cin
    print(f"conv {Cin}->{Cout} d{dil}: " + "  ".join(
        f"{t}: {sorted(times[t])[3]:.3f} ms {fl / sorted(times[t])[3] / 1e9:.0f} TF eq={torch.equal(outs[t], outs[tiles[0]])}" for t in tiles), flush=True)
