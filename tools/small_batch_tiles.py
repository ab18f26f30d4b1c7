"""Tile choice at small batches (1 - 4 images): every conv of the backbone and the FC-sized contractions of the heads under
each tile the dispatcher knows, bf16 and bf16x2.  python tools/small_batch_tiles.py [images]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from wsovod_amd.layers import hip_ops as H

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
torch.manual_seed(0)
TILES = [0, 2256256, 256128, 1128128, 1128064, 3128064, 64064, 3064064]


def bench(fns, rounds=5, inner=10):
    times = {k: [] for k in fns}
    for k, f in list(fns.items()):
        try:
            f()
        except RuntimeError as e:
            del fns[k], times[k]
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(inner):
                f()
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / inner)
    return {k: sorted(v)[len(v) // 2] for k, v in times.items()}


def conv(name, Hh, Ww, Cin, Cout, dil, x2):
    if x2:
        x = H.x2_encode(torch.randn(n * Hh * Ww, Cin, device=dev)).view(n, Hh, Ww, Cin)
        w = H.x2_encode(torch.randn(Cout, 9 * Cin, device=dev) * 0.05)
    else:
        x = torch.randn(n, Hh, Ww, Cin, device=dev).to(torch.bfloat16)
        w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.05).to(torch.bfloat16)
    b = torch.randn(Cout, device=dev)
    geom = dict(n_img=n, H=Hh, W=Ww, Cin=Cin, Ho=Hh, Wo=Ww, KH=3, KW=3, stride=1, pad=dil, dil=dil)
    fns = {t: (lambda t=t: H.gemm_nt(x, w, conv=geom, x2=x2, bias=b, relu=True, out_dtype=H.X2 if x2 else torch.bfloat16,
                                     tile_hint=t)) for t in TILES}
    r = bench(fns)
    fl = 2.0 * n * Hh * Ww * Cout * 9 * Cin
    best = min(r, key=r.get)
    print(f"{name:28s} x2={int(x2)}", {t: f"{ms * 1e3:.0f}us" for t, ms in r.items()}, f"auto {r[0] * 1e3:.0f} best {best} {r[best] * 1e3:.0f}us "
          f"({fl / r[best] / 1e9:.0f} TF)", flush=True)


def gemm(name, M, N, K, x2):
    if x2:
        a = H.x2_encode(torch.randn(M, K, device=dev))
        b = H.x2_encode(torch.randn(N, K, device=dev) * 0.01)
    else:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        b = (torch.randn(N, K, device=dev) * 0.01).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    fns = {t: (lambda t=t: H.gemm_nt(a, b, x2=x2, bias=bias, relu=True, out_dtype=H.X2 if x2 else torch.bfloat16, tile_hint=t))
           for t in TILES}
    r = bench(fns)
    best = min(r, key=r.get)
    print(f"{name:28s} x2={int(x2)}", {t: f"{ms * 1e3:.0f}us" for t, ms in r.items()}, f"auto {r[0] * 1e3:.0f} best {best} {r[best] * 1e3:.0f}us "
          f"({2.0 * M * N * K / r[best] / 1e9:.0f} TF)", flush=True)


for x2 in ((True,) if os.environ.get("X2_ONLY") else (False, True)):
    conv("res3 128->128 75x100", 75, 100, 128, 128, 1, x2)
    conv("res4a 128->256 d2", 75, 100, 128, 256, 2, x2)
    conv("res4 256->256 d2", 75, 100, 256, 256, 2, x2)
    conv("res5a 256->512 d2", 75, 100, 256, 512, 2, x2)
    conv("res5 512->512 d2", 75, 100, 512, 512, 2, x2)
    gemm("fc2 fwd (512n x 4096 x 4096)", 512 * n, 4096, 4096, x2)
    gemm("heads (512n x 1088 x 4096)", 512 * n, 1088, 4096, x2)
    gemm("proj2 (512n x 512 x 1024)", 512 * n, 512, 1024, x2)
