"""The f16mx implicit-GEMM conv (wsovod_gemm_f16mx, conv form) against the bf16x2 three-product lean tile on the res4 / res5
shapes of the benchmark (32 images of 75 x 100 after the stride-8 stem).   python tools/mx_conv_ab.py [n_images]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import wsovod_amd._lib as _L

if os.environ.get("WSOVOD_LIB"):  # (timing ablations: an alternative build of the library)
    _L.LIB_PATH = os.environ["WSOVOD_LIB"]
from wsovod_amd.layers import hip_ops as H

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
Hi, Wi = 75, 100


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 3)
    return sorted(ts)[len(ts) // 2]


tot_x2 = tot_mx = 0.0
#            name            Cin  Cout Cin2 count per step
for name, Cin, Cout, Cin2, cnt in (("res4.0.conv1", 128, 256, 0, 1), ("res4.0.conv2+sc", 256, 256, 128, 1), ("res4.1.conv", 256, 256, 0, 2),
                                   ("res5.0.conv1", 256, 512, 0, 1), ("res5.0.conv2+sc", 512, 512, 256, 1), ("res5.1.conv", 512, 512, 0, 2)):
    x = torch.relu(torch.randn(n * Hi * Wi, Cin, device="cuda"))
    w = torch.randn(Cout, 9 * Cin + Cin2, device="cuda") * 0.02
    bias = torch.randn(Cout, device="cuda")
    geom = dict(n_img=n, H=Hi, W=Wi, Cin=Cin, Ho=Hi, Wo=Wi, KH=3, KW=3, stride=1, pad=2, dil=2)
    xx = H.x2_encode(x).view(n, Hi, Wi, Cin)
    xm = H.mx_encode(x, unit=True)[0].view(n, Hi, Wi, Cin)
    wx = H.x2_encode(w)
    wm, sw = H.mx_encode(w)
    a2x = a2m = None
    if Cin2:
        x2 = torch.relu(torch.randn(n * Hi * Wi, Cin2, device="cuda"))
        a2x = H.x2_encode(x2).view(n, Hi, Wi, Cin2)
        a2m = H.mx_encode(x2, unit=True)[0].view(n, Hi, Wi, Cin2)
    o1 = torch.empty(n * Hi * Wi, Cout, device="cuda")
    o2 = torch.empty(n * Hi * Wi, Cout, device="cuda")
    t_x2 = t(lambda: H.gemm_nt(xx, wx, conv=geom, x2=True, bias=bias, relu=True, out=o1, out_dtype=H.X2, A2=a2x))
    os.environ["WSOVOD_MX_TAIL"] = "0"
    t_mx1 = t(lambda: H.gemm_mx(xm, None, wm, sw, conv=geom, A2=a2m, bias=bias, relu=True, out=o2, out_dtype=H.MX))
    os.environ["WSOVOD_MX_TAIL"] = "1"
    t_mx = t(lambda: H.gemm_mx(xm, None, wm, sw, conv=geom, A2=a2m, bias=bias, relu=True, out=o2, out_dtype=H.MX))
    tot_one = globals().get("tot_one", 0.0) + cnt * t_mx1
    if cnt == 2:  # the second conv of the identity blocks adds the block's input (residual) in its epilogue
        rx, rm = H.x2_encode(torch.randn(n * Hi * Wi, Cout, device="cuda")), H.mx_encode(torch.randn(n * Hi * Wi, Cout, device="cuda"), unit=True)[0]
        t_x2r = t(lambda: H.gemm_nt(xx, wx, conv=geom, x2=True, bias=bias, relu=True, out=o1, out_dtype=H.X2, residual=rx, residual_x2=True))
        t_mxr = t(lambda: H.gemm_mx(xm, None, wm, sw, conv=geom, bias=bias, relu=True, out=o2, out_dtype=H.MX, residual=rm, residual_fmt=H.MX))
        print(f"   with the residual: bf16x2 {t_x2r:.3f} ms   f16mx {t_mxr:.3f} ms", flush=True)
        del rx, rm
    fl = 2.0 * n * Hi * Wi * Cout * (9 * Cin + Cin2)
    d1, d2 = H.x2_decode(o1), H.mx_to_f32(o2)
    err = float((d1 - d2).abs().max() / d1.abs().max())
    tot_x2 += cnt * t_x2
    tot_mx += cnt * t_mx
    print(f"{name} ({Cin}->{Cout}, x{cnt}): bf16x2 {t_x2:.3f} ms ({fl / t_x2 / 1e9:.0f} TF algorithmic)   f16mx {t_mx:.3f} ms "
          f"({fl / t_mx / 1e9:.0f} TF; one launch, no split-K tail: {t_mx1:.3f} ms)   x{t_x2 / t_mx:.3f}   max |diff| / max = {err:.2e}", flush=True)
print(f"per step (8 convs): bf16x2 {tot_x2:.3f} ms   f16mx {tot_mx:.3f} ms (single launches: {tot_one:.3f})   x{tot_x2 / tot_mx:.3f}")
