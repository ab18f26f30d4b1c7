"""Where a K-step of the 8-wavefront two-phase tile (gemm8.hip, PH = 2) spends its cycles: instrumented build
`hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -DG8_STAMPS=1 -c wsovod_amd/csrc/gemm8.hip -o /tmp/g8_st.o &&
 hipcc --offload-arch=gfx950 -shared -fPIC -o wsovod_amd/lib/abl/libg8.so /tmp/g8_st.o $(ls wsovod_amd/csrc/build/*.o | grep -v "/gemm8.o")`,
then `WSOVOD_LIB=$PWD/wsovod_amd/lib/abl/libg8.so python tools/g8_phases.py` on the box."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import wsovod_amd._lib as _L
if os.environ.get("WSOVOD_LIB"):
    _L.LIB_PATH = os.environ["WSOVOD_LIB"]
from wsovod_amd.layers import hip_ops as H
dbg = torch.zeros(32, device="cuda")
os.environ["WSOVOD_G8_DEBUG_PTR"] = hex(dbg.data_ptr())
names = ["A: 16 reads landed", "A: 6 DMA issued", "A: vmcnt wait", "A: barrier 1", "A: lgkm + 48 MFMA", "A: barrier 2",
         "B: 8 reads + 2 DMA + wait", "B: barrier + 48 MFMA + barrier"]
def report(tag, fl, run):
    run(); torch.cuda.synchronize(); dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    d = dbg.view(2, 16).cpu()
    print(f"{tag}: {ms:.3f} ms ({fl / ms / 1e9:.0f} TF executed, instrumented)")
    for g in range(2):
        n = float(d[g, 8])
        if n and float(d[g, :8].sum()) == 0:  # -DG8_STAMPS=2: tile-level stamps only
            w = float(d[g, 11])
            print(f"  group {g} per tile: setup + K-step 0 requested {float(d[g, 9]) / w:.0f}, its data + barriers {float(d[g, 12]) / w:.0f}, "
                  f"loop {float(d[g, 10]) / w:.0f}, epilogue issue {float(d[g, 13]) / w:.0f}, store drain {float(d[g, 14]) / w:.0f} ticks", flush=True)
        elif n:
            print(f"  group {g}: ticks per K-step: " + ", ".join(f"{nm} {float(d[g, k]) / n:.0f}" for k, nm in enumerate(names)) +
                  f"; total {float(d[g, :8].sum()) / n:.0f}", flush=True)
            w = float(d[g, 11])
            if w:  # per workgroup (the tile's wall time is the caller's: ms x CUs / tiles)
                print(f"    per tile: setup + K-step 0 requested {float(d[g, 9]) / w:.0f}, its data + barriers {float(d[g, 12]) / w:.0f}, "
                      f"loop {float(d[g, 10]) / w:.0f}, epilogue issue {float(d[g, 13]) / w:.0f}, store drain {float(d[g, 14]) / w:.0f} ticks",
                      flush=True)
M, N, K = 8192, 4096, 25088
a = H.x2_encode(torch.randn(M, K, device="cuda")); b = H.x2_encode(torch.randn(N, K, device="cuda") * 0.01)
bias = torch.randn(N, device="cuda")
report("fc1 fwd x2 (two-phase)", 6.0 * M * N * K, lambda: H.gemm_nt(a, b, x2=True, bias=bias, relu=True, out_dtype=H.X2, tile_hint=2256256))
del a, b
a = torch.randn(16384, 4096, device="cuda").to(torch.bfloat16); b = (torch.randn(4096, 4096, device="cuda") * 0.01).to(torch.bfloat16)
report("fc2 dX bf16 (two-phase)", 2.0 * 16384 * 4096 * 4096, lambda: H.gemm_nt(a, b, out_dtype=torch.float32, tile_hint=2256256))
n, Hi, Wi, Cin, Cout = 16, 75, 100, 512, 512
x = H.x2_encode(torch.randn(n * Hi * Wi, Cin, device="cuda")).view(n, Hi, Wi, Cin)
w = H.x2_encode(torch.randn(Cout, 9 * Cin, device="cuda") * 0.05)
geom = dict(n_img=n, H=Hi, W=Wi, Cin=Cin, Ho=Hi, Wo=Wi, KH=3, KW=3, stride=1, pad=2, dil=2)
report("res5 conv x2 (two-phase)", 6.0 * n * Hi * Wi * Cout * 9 * Cin, lambda: H.gemm_nt(x, w, conv=geom, x2=True, bias=bias[:Cout], relu=True, out_dtype=H.X2, tile_hint=2256256))

n, Hi, Wi, Cin, Cout = 32, 75, 100, 256, 256
x = H.x2_encode(torch.randn(n * Hi * Wi, Cin, device="cuda")).view(n, Hi, Wi, Cin)
w = H.x2_encode(torch.randn(Cout, 9 * Cin, device="cuda") * 0.05)
geom = dict(n_img=n, H=Hi, W=Wi, Cin=Cin, Ho=Hi, Wo=Wi, KH=3, KW=3, stride=1, pad=2, dil=2)
report("res4 conv x2, 32 images (938 tiles = 3.66 rounds)", 6.0 * n * Hi * Wi * Cout * 9 * Cin,
       lambda: H.gemm_nt(x, w, conv=geom, x2=True, bias=bias[:Cout], relu=True, out_dtype=H.X2, tile_hint=2256256))
n = 7  # 206 tiles: ONE partly filled round -> ms = the wall time of a tile
x = H.x2_encode(torch.randn(n * Hi * Wi, Cin, device="cuda")).view(n, Hi, Wi, Cin)
geom["n_img"] = n
report("res4 conv x2, 7 images (206 tiles: ms = one tile's wall time)", 6.0 * n * Hi * Wi * Cout * 9 * Cin,
       lambda: H.gemm_nt(x, w, conv=geom, x2=True, bias=bias[:Cout], relu=True, out_dtype=H.X2, tile_hint=2256256))
a = H.x2_encode(torch.randn(256 * 14, 4096, device="cuda")); b = H.x2_encode(torch.randn(4096, 4096, device="cuda") * 0.01)
report("fc2 fwd x2, 224 tiles (ms = one tile's wall time, 64 K-steps)", 6.0 * 256 * 14 * 4096 * 4096,
       lambda: H.gemm_nt(a, b, x2=True, bias=bias, relu=True, out_dtype=H.X2, tile_hint=2256256))
