"""A/B probe of the bf16x2 (three-MFMA) kernels at the bench shapes: python tools/x2_probe.py [images]
Interleaved rounds in one process, random operands; prints ms and executed TFLOP/s (3 x 2MNK)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import wsovod_amd._lib as _L

if os.environ.get("WSOVOD_LIB"):  # another build of the library (ablation / A-B runs on one box)
    _L.LIB_PATH = os.environ["WSOVOD_LIB"]
from wsovod_amd.layers import hip_ops as H

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
torch.manual_seed(0)


def bench(fns, rounds=6, inner=3):
    times = {k: [] for k in fns}
    for k, f in fns.items():
        f()
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


def conv_case(name, Hh, Ww, Cin, Cout, k, dil, tiles, residual=False, pool=0):
    x = H.x2_encode(torch.randn(n * Hh * Ww, Cin, device=dev)).view(n, Hh, Ww, Cin)
    w = H.x2_encode(torch.randn(Cout, k * k * Cin, device=dev) * 0.05)
    b = torch.randn(Cout, device=dev)
    pad = dil * (k // 2)
    geom = dict(n_img=n, H=Hh, W=Ww, Cin=Cin, Ho=Hh, Wo=Ww, KH=k, KW=k, stride=1, pad=pad, dil=dil, pool=pool)
    res = H.x2_encode(torch.randn(n * Hh * Ww, Cout, device=dev)) if residual else None
    fns = {}
    for t in tiles:
        fns[str(t)] = (lambda t=t: H.gemm_nt(x, w, conv=geom, x2=True, bias=b, relu=True, residual=res, residual_x2=residual,
                                             out_dtype=H.X2, tile_hint=t))
    r = bench(fns)
    fl = 6.0 * n * Hh * Ww * Cout * k * k * Cin
    print(name, {t: f"{ms:.3f} ms {fl / ms / 1e9:.0f} TF" for t, ms in r.items()}, flush=True)


def gemm_case(name, M, N, K, tiles):
    a = H.x2_encode(torch.randn(M, K, device=dev))
    b = H.x2_encode(torch.randn(N, K, device=dev) * 0.01)
    bias = torch.randn(N, device=dev)
    fns = {str(t): (lambda t=t: H.gemm_nt(a, b, x2=True, bias=bias, relu=True, out_dtype=H.X2, tile_hint=t)) for t in tiles}
    r = bench(fns)
    fl = 6.0 * M * N * K
    print(name, {t: f"{ms:.3f} ms {fl / ms / 1e9:.0f} TF" for t, ms in r.items()}, flush=True)


if __name__ == "__main__":
    conv_case("stem 64->64 300x400 (c64 halo vs generic)", 300, 400, 64, 64, 3, 1, [0, 1256064])
    conv_case("res2 64->64 150x200", 150, 200, 64, 64, 3, 1, [0, 1256064], residual=True)
    if os.environ.get("X2_PROBE_C64_ONLY"):  # (WSOVOD_C64X_LEPI=0 / 1 in two processes: the switch is read once)
        conv_case("stem 64->64 300x400 + 2x2 max pool", 300, 400, 64, 64, 3, 1, [0], pool=2)
        conv_case("res2 64->64 150x200, no residual", 150, 200, 64, 64, 3, 1, [0])
        sys.exit(0)
    conv_case("res3 128->128 75x100", 75, 100, 128, 128, 3, 1, [0, 256128, 256256, 8256256])
    conv_case("res4 256->256 d2", 75, 100, 256, 256, 3, 2, [256256, 8256256, 2256256, 256128])
    conv_case("res5 512->512 d2", 75, 100, 512, 512, 3, 2, [256256, 8256256, 2256256])
    if os.environ.get("X2_PROBE_CONV_ONLY"):
        sys.exit(0)
    gemm_case("fc1", n * 512, 4096, 25088, [8256256, 2256256, 256256])
    gemm_case("fc2", n * 512, 4096, 4096, [8256256, 2256256, 256256])
    gemm_case("proj1", n * 512, 1024, 4096, [0, 8256256, 2256256, 256128])
    # plain bf16 forms of the two 8-wavefront schedules (the headline precision)
    A = torch.randn(n * 512, 25088, device=dev).to(torch.bfloat16)
    B = (torch.randn(4096, 25088, device=dev) * 0.01).to(torch.bfloat16)
    r = bench({str(t): (lambda t=t: H.gemm_nt(A, B, out_dtype=torch.bfloat16, tile_hint=t)) for t in (8256256, 2256256, 256256)})
    print("fc1 bf16", {t: f"{ms:.3f} ms {2.0 * n * 512 * 4096 * 25088 / ms / 1e9:.0f} TF" for t, ms in r.items()}, flush=True)
    x = torch.randn(n, 75, 100, 512, device=dev).to(torch.bfloat16)
    w = (torch.randn(512, 9 * 512, device=dev) * 0.02).to(torch.bfloat16)
    geom = dict(n_img=n, H=75, W=100, Cin=512, Ho=75, Wo=100, KH=3, KW=3, stride=1, pad=2, dil=2)
    r = bench({str(t): (lambda t=t: H.gemm_nt(x, w, conv=geom, relu=True, out_dtype=torch.bfloat16, tile_hint=t))
               for t in (256256, 8256256, 2256256)})
    print("res5 bf16", {t: f"{ms:.3f} ms {2.0 * n * 7500 * 512 * 4608 / ms / 1e9:.0f} TF" for t, ms in r.items()}, flush=True)
    # res3 (128 -> 128, K = 1152) and res4 (256 -> 256, K = 2304) in bf16: short-K convs, tile choice
    for name, C, dil, tiles in (("res3 bf16", 128, 1, (256128, 1128128, 128128, 256256, 3128128, 4128128)),
                                ("res4 bf16", 256, 2, (256256, 256128, 1128128, 8256256, 3256128))):
        x = torch.randn(n, 75, 100, C, device=dev).to(torch.bfloat16)
        w = (torch.randn(C, 9 * C, device=dev) * 0.02).to(torch.bfloat16)
        geom = dict(n_img=n, H=75, W=100, Cin=C, Ho=75, Wo=100, KH=3, KW=3, stride=1, pad=dil, dil=dil)
        r = bench({str(t): (lambda t=t: H.gemm_nt(x, w, conv=geom, relu=True, out_dtype=torch.bfloat16, tile_hint=t)) for t in tiles})
        print(name, {t: f"{ms:.3f} ms {2.0 * n * 7500 * C * 9 * C / ms / 1e9:.0f} TF" for t, ms in r.items()}, flush=True)
