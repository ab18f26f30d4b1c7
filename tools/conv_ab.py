"""A/B implicit-GEMM conv tiles on the backbone's res3-res5 shapes (16 images, 75x100 maps)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wsovod_amd._lib as _L
if os.environ.get("WSOVOD_LIB"):  # another build of the library (ablation builds) for an A/B on one box
    _L.LIB_PATH = os.environ["WSOVOD_LIB"]
from wsovod_amd.layers import hip_ops
tiles = [int(t) for t in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["256256", "8256256"])]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for (Cin, Cout, H, W, k, dil) in ((128, 128, 75, 100, 3, 1), (256, 256, 75, 100, 3, 2), (512, 512, 75, 100, 3, 2), (128, 256, 75, 100, 3, 2), (256, 512, 75, 100, 1, 1)):
    pad = dil * (k // 2)
    x = (torch.rand(n, H, W, Cin, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand(Cout, k * k * Cin, device="cuda") * 2 - 1) * 0.05).to(torch.bfloat16)
    bias = torch.randn(Cout, device="cuda")
    geom = dict(n_img=n, H=H, W=W, Cin=Cin, Ho=H, Wo=W, KH=k, KW=k, stride=1, pad=pad, dil=dil)
    outs, times = {}, {t: [] for t in tiles}
    for t in list(tiles):
        try:
            outs[t] = hip_ops.gemm_nt(x, w, conv=geom, bias=bias, relu=True, out_dtype=torch.bfloat16, tile_hint=t)
        except RuntimeError as e:  # a tile that does not take this shape
            print(f"conv {Cin}->{Cout} k{k} d{dil}: tile {t} n/a", flush=True)
            outs[t] = None
    for r in range(7):
        for t in tiles:
            if outs[t] is None:
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                hip_ops.gemm_nt(x, w, conv=geom, bias=bias, relu=True, out=outs[t], tile_hint=t)
            e1.record(); torch.cuda.synchronize(); times[t].append(e0.elapsed_time(e1) / 3)
    fl = 2.0 * n * H * W * Cout * k * k * Cin
    for t in tiles:
        if outs[t] is None:
            continue
        med = sorted(times[t])[3]
        diff = float((outs[t].float() - outs[tiles[0]].float()).abs().max())
        print(f"conv {Cin}->{Cout} k{k} d{dil}: tile {t} {med:.3f} ms {fl / med / 1e9:.0f} TF equal={torch.equal(outs[t], outs[tiles[0]])} maxdiff={diff:.3g}", flush=True)
