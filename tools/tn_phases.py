"""Where a phase of the weight-gradient kernel (gemm_tn8.hip) spends its cycles: instrumented build
`mkdir -p wsovod_amd/lib/abl && hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -DTN_STAMPS=1 -c wsovod_amd/csrc/gemm_tn8.hip -o /tmp/tn_st.o
 && hipcc --offload-arch=gfx950 -shared -fPIC -o wsovod_amd/lib/abl/libtn.so /tmp/tn_st.o $(ls wsovod_amd/csrc/build/*.o | grep -v gemm_tn8)`,
then `WSOVOD_LIB=$PWD/wsovod_amd/lib/abl/libtn.so python tools/tn_phases.py` on the box (WSOVOD_TN_LEAN=0 picks the round-4 form)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import wsovod_amd._lib as _L
if os.environ.get("WSOVOD_LIB"):
    _L.LIB_PATH = os.environ["WSOVOD_LIB"]
from wsovod_amd.layers import hip_ops as H
dbg = torch.zeros(16, device="cuda")
os.environ["WSOVOD_TN_DEBUG_PTR"] = hex(dbg.data_ptr())
for (Mred, NI, NJ) in ((8192, 4096, 25088), (16384, 4096, 4096)):
    P = (torch.rand(Mred, NI, device="cuda") * 2 - 1).to(torch.bfloat16)
    Q = (torch.rand(Mred, NJ, device="cuda") * 2 - 1).to(torch.bfloat16)
    out = torch.empty(NI, NJ, device="cuda")
    H.gemm_tn(P, Q, out=out); torch.cuda.synchronize(); dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        H.gemm_tn(P, Q, out=out)
    e1.record(); torch.cuda.synchronize()
    d = dbg.view(2, 8).cpu()
    ms = e0.elapsed_time(e1) / 3
    print(f"{Mred}x{NI}x{NJ}: {ms:.3f} ms ({2.0 * Mred * NI * NJ / ms / 1e9:.0f} TF, instrumented)")
    for g in range(2):
        n = float(d[g, 4])
        names = ["vmcnt wait (LEAN; else reads + DMA issue + wait)", "barrier 1", "lgkm wait + 32 MFMAs", "barrier 2"]
        print(f"  group {g}: ticks per phase: 24 reads landed {float(d[g, 5]) / n:.0f}, DMA issue {float(d[g, 6]) / n:.0f}, " +
              ", ".join(f"{nm} {float(d[g, k]) / n:.0f}" for k, nm in enumerate(names)) +
              f"; total {float(d[g, :4].sum() + d[g, 5:7].sum()) / n:.0f}", flush=True)
