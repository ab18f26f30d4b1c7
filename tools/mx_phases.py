"""Where a K-step of the f16mx two-phase tile (gemm8mx.hip) spends its cycles: instrumented build
`hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -DMX_STAMPS -c wsovod_amd/csrc/gemm8mx.hip -o /tmp/mx_st.o &&
 hipcc --offload-arch=gfx950 -shared -fPIC -o wsovod_amd/lib/abl/libmx.so /tmp/mx_st.o $(ls wsovod_amd/csrc/build/*.o | grep -v "/gemm8mx.o")`,
then `WSOVOD_LIB=$PWD/wsovod_amd/lib/abl/libmx.so python tools/mx_phases.py` on the box."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import wsovod_amd._lib as _L
if os.environ.get("WSOVOD_LIB"):
    _L.LIB_PATH = os.environ["WSOVOD_LIB"]
from wsovod_amd.layers import hip_ops as H
dbg = torch.zeros(32, device="cuda")
os.environ["WSOVOD_MX_DEBUG_PTR"] = hex(dbg.data_ptr())
names = os.environ.get("MX_STAMP_NAMES", "A: reads landed|A: 4 DMA issued|A: lgkm|A: barrier 1|A: 12 MFMA|A: barrier 2|"
                       "B: reads + 4 DMA + vmcnt(8) + lgkm|B: barrier + 12 MFMA + barrier").split("|")
def report(tag, fl, run):
    run(); torch.cuda.synchronize(); dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    d = dbg.view(2, 16).cpu()
    print(f"{tag}: {ms:.3f} ms ({fl / ms / 1e9:.0f} TF algorithmic, instrumented)")
    for g in range(2):
        n = float(d[g, 8])
        if n:
            print(f"  group {g}: ticks per K-step: " + ", ".join(f"{nm} {float(d[g, k]) / n:.0f}" for k, nm in enumerate(names)) +
                  f"; total {float(d[g, :8].sum()) / n:.0f}", flush=True)
            w = float(d[g, 11])
            if w:
                print(f"    per tile: setup + K-steps 0 / 1 requested {float(d[g, 9]) / w:.0f}, their data + barriers {float(d[g, 12]) / w:.0f}, "
                      f"loop {float(d[g, 10]) / w:.0f}, epilogue issue {float(d[g, 13]) / w:.0f}, store drain {float(d[g, 14]) / w:.0f} ticks",
                      flush=True)
for M, N, K in ((16384, 4096, 25088), (16384, 4096, 4096)):
    a, sa = H.mx_encode(torch.randn(M, K, device="cuda")); b, sb = H.mx_encode(torch.randn(N, K, device="cuda") * 0.01)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    report(f"f16mx {M}x{N}x{K}", 2.0 * M * N * K, lambda: H.gemm_mx(a, sa, b, sb, bias=bias, relu=True, out=out, out_dtype=H.X2))
    del a, b
for name, n, Cin, Cout in (("res5 conv 512->512, 16 images", 16, 512, 512), ("res4 conv 256->256, 32 images", 32, 256, 256)):
    Hi, Wi = 75, 100
    x = H.mx_encode(torch.relu(torch.randn(n * Hi * Wi, Cin, device="cuda")), unit=True)[0].view(n, Hi, Wi, Cin)
    w, sw = H.mx_encode(torch.randn(Cout, 9 * Cin, device="cuda") * 0.02)
    bias = torch.randn(Cout, device="cuda")
    geom = dict(n_img=n, H=Hi, W=Wi, Cin=Cin, Ho=Hi, Wo=Wi, KH=3, KW=3, stride=1, pad=2, dil=2)
    out = torch.empty(n * Hi * Wi, Cout, device="cuda")
    os.environ["WSOVOD_MX_TAIL"] = "0"
    report(f"f16mx {name}", 2.0 * n * Hi * Wi * Cout * 9 * Cin, lambda: H.gemm_mx(x, None, w, sw, conv=geom, bias=bias, relu=True, out=out, out_dtype=H.MX))
    del x, w
