"""What rate would the block-scaled cross-term format's MFMA mix reach in the lean tile?  (round 6; TIMING ONLY)

    build (here):   bash tools/mx_rate_probe.sh         -> wsovod_amd/lib/abl/libmxprobe.so  (gemm8.hip with -DG8_MXPROBE=1)
    run (GPU box):  python tools/mx_rate_probe.py       (runs the product library, then the probe build, same shapes)

The probe build replaces the MFMA block of the PLAIN bf16 lean phase (32 x v_mfma_f32_16x16x32_bf16 = 512 matrix-pipe cycles)
by the mix the format would issue on the same fragments (per 32x32 tile: 2 x v_mfma_f32_32x32x16_f16 + 1 x
v_mfma_scale_f32_32x32x64_f8f6f4 = 12 MFMAs, also 512 cycles): same LDS image, DMA schedule, barriers.  One bf16 K-step (64
slots = 128 B per row) stands for one K-step of the format (32 values), so a GEMM over K values is timed as a bf16 GEMM over 2 K
slots and its ALGORITHMIC rate is 2 M N K / t.  Results of the probe build are meaningless (the bits are whatever the bf16
operands hold); no scale fetch, no encode epilogue: an upper bound of the loop's rate."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    import torch
    import wsovod_amd._lib as L
    if os.environ.get("WSOVOD_LIB"):
        L.LIB_PATH = os.environ["WSOVOD_LIB"]
    from wsovod_amd.layers import hip_ops as H

    def t(fn, n=7):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                fn()
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 3)
        return sorted(ts)[len(ts) // 2]

    tag = "probe" if os.environ.get("WSOVOD_LIB") else "bf16 lean tile"
    for name, M, N, K in (("fc1 16384x4096x25088", 16384, 4096, 25088), ("fc2 16384x4096x4096", 16384, 4096, 4096)):
        A = (torch.rand(M, 2 * K, device="cuda") * 2 - 1).to(torch.bfloat16)
        B = (torch.rand(N, 2 * K, device="cuda") * 2 - 1).to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda")
        ms = t(lambda: H.gemm_nt(A, B, out=out, tile_hint=2256256))
        print(f"{tag:14s} {name:24s} {ms:.3f} ms  = {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s algorithmic for K values", flush=True)
        del A, B, out
    for name, Cin, Cout in (("res5 conv 512->512 d2", 512, 512), ("res4 conv 256->256 d2", 256, 256)):
        n, Hh, Ww = 32, 75, 100
        x = (torch.rand(n, Hh, Ww, 2 * Cin, device="cuda") * 2 - 1).to(torch.bfloat16)
        w = ((torch.rand(Cout, 9 * 2 * Cin, device="cuda") * 2 - 1) * 0.05).to(torch.bfloat16)
        geom = dict(n_img=n, H=Hh, W=Ww, Cin=2 * Cin, Ho=Hh, Wo=Ww, KH=3, KW=3, stride=1, pad=2, dil=2)
        o = H.gemm_nt(x, w, conv=geom, relu=True, out_dtype=torch.bfloat16, tile_hint=2256256)
        ms = t(lambda: H.gemm_nt(x, w, conv=geom, relu=True, out=o, tile_hint=2256256))
        print(f"{tag:14s} {name:24s} {ms:.3f} ms  = {2.0 * n * Hh * Ww * Cout * 9 * Cin / ms / 1e9:.0f} TFLOP/s algorithmic", flush=True)


if __name__ == "__main__":
    if os.environ.get("MX_PROBE_CHILD"):
        child()
    else:
        env = dict(os.environ, MX_PROBE_CHILD="1")
        env.pop("WSOVOD_LIB", None)
        subprocess.run([sys.executable, __file__], env=env, check=False)
        lib = os.path.join(ROOT, "wsovod_amd", "lib", "abl", "libmxprobe.so")
        if os.path.exists(lib):
            subprocess.run([sys.executable, __file__], env=dict(env, WSOVOD_LIB=lib), check=False)
        else:
            print("no probe build (bash tools/mx_rate_probe.sh)")
