"""Where does a split-K implicit-GEMM conv of ONE image spend its time?  (round 6)

    python tools/splitk_probe.py [images]

res4 / res5 3x3 convs on bf16x2 maps as K slices of the lean 8-wavefront 256x256 tile (WSOVOD_CONV_SPLITK=1) for S = 2 .. 8
slices (WSOVOD_SPLITK_S), main kernel and finalize pass timed separately by the library's profiler, against the 128x64 grids."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import wsovod_amd._lib as L
from wsovod_amd.layers import hip_ops as H

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
torch.manual_seed(0)


def run(name, Cin, Cout, dil, tile, env):
    Hh, Ww = 75, 100
    x = H.x2_encode(torch.randn(n * Hh * Ww, Cin, device=dev)).view(n, Hh, Ww, Cin)
    w = H.x2_encode(torch.randn(Cout, 9 * Cin, device=dev) * 0.05)
    b = torch.randn(Cout, device=dev)
    geom = dict(n_img=n, H=Hh, W=Ww, Cin=Cin, Ho=Hh, Wo=Ww, KH=3, KW=3, stride=1, pad=dil, dil=dil)
    # a second, cold set of operands alternated with the first (the step never re-runs a conv on warm operands)
    x2_ = H.x2_encode(torch.randn(n * Hh * Ww, Cin, device=dev)).view(n, Hh, Ww, Cin)
    flush = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    for k, v in env.items():
        os.environ[k] = v
    f = lambda xx: H.gemm_nt(xx, w, conv=geom, x2=True, bias=b, relu=True, out_dtype=H.X2, tile_hint=tile)
    for _ in range(3):
        f(x)
    torch.cuda.synchronize()
    L.profile_reset()
    L.profile_enable(True)
    for i in range(10):
        flush.zero_()  # push the operands out of L2 / the memory-side cache
        f(x if i & 1 else x2_)
    torch.cuda.synchronize()
    t = {e["name"]: e["ms"] / max(e["launches"], 1) * 1e3 for e in L.profile_collect() if e["launches"]}
    L.profile_enable(False)
    for k in env:
        os.environ.pop(k)
    t.pop("memset", None)
    fl = 2.0 * n * Hh * Ww * Cout * 9 * Cin
    tot = sum(v for k, v in t.items())
    print(f"{name:22s} tile {tile:8d} {str(env):58s} " + "  ".join(f"{k} {v:.0f}us" for k, v in t.items()) +
          f"   total {tot:.0f}us = {fl / tot / 1e6:.0f} TF algorithmic", flush=True)


for name, Cin, Cout, dil in (("res4 256->256", 256, 256, 2), ("res5a 256->512", 256, 512, 2), ("res5 512->512", 512, 512, 2)):
    run(name, Cin, Cout, dil, 1128064, {})
    run(name, Cin, Cout, dil, 3128064, {})
    run(name, Cin, Cout, dil, 2256256, {})
    for S in (2, 3, 4, 6, 8):
        run(name, Cin, Cout, dil, 0, {"WSOVOD_CONV_SPLITK": "1", "WSOVOD_SPLITK_S": str(S)})
