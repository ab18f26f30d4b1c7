"""Micro-probe of the hot kernels on one MI355X (not a test, not the bench): prints per-shape rates."""
import sys
import os
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsovod_amd.layers import hip_ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    print(torch.cuda.get_device_name(0))
    shapes = [("fc1_fwd_b1", 512, 4096, 25088), ("fc1_fwd_b8", 4096, 4096, 25088), ("fc2_b8", 4096, 4096, 4096),
              ("fc1_dW_b1", 4096, 25088, 512), ("fc1_dW_b8", 4096, 25088, 4096), ("sq4096", 4096, 4096, 4096)]
    for dt in (torch.bfloat16,):
        for name, M, N, K in shapes:
            if dt == torch.float32 and M * N * K > 4096 * 4096 * 25088 // 2 and "b8" in name and "fc1" in name:
                iters = 3
            else:
                iters = 10
            A = torch.randn(M, K, device=dev).to(dt)
            B = torch.randn(N, K, device=dev).to(dt)
            out = torch.empty(M, N, device=dev, dtype=torch.float32)
            for tile in (0, 256256):
                ms = timeit(lambda: hip_ops.gemm_nt(A, B, out=out, tile_hint=tile), iters=iters)
                print(f"gemm {str(dt)[6:]:9s} {name:12s} tile {tile:6d}  {ms:8.3f} ms  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s")
            del A, B, out
    # conv res5: 75x100x512 -> 512, dil 2
    for dt in (torch.bfloat16, torch.float32):
        for (n, H, W, Cin, Cout, dil, nm) in [(1, 75, 100, 512, 512, 2, "res5_b1"), (8, 75, 100, 512, 512, 2, "res5_b8"),
                                              (8, 300, 400, 64, 64, 1, "stem_b8"), (8, 150, 200, 64, 64, 1, "res2_b8")]:
            x = torch.randn(n, H, W, Cin, device=dev).to(dt)
            w = torch.randn(Cout, 9 * Cin, device=dev).to(dt)
            geom = dict(n_img=n, H=H, W=W, Cin=Cin, Ho=H, Wo=W, KH=3, KW=3, stride=1, pad=dil, dil=dil)
            out = torch.empty(n * H * W, Cout, device=dev, dtype=dt)
            fl = 2.0 * n * H * W * Cout * 9 * Cin
            for tile in (0, 256256):
                ms = timeit(lambda: hip_ops.gemm_nt(x, w, conv=geom, out=out, relu=True, tile_hint=tile), iters=5)
                print(f"conv {str(dt)[6:]:9s} {nm:10s} tile {tile:7d} {ms:8.3f} ms  {fl/ms/1e9:8.1f} TFLOP/s")
    return
    from tests.util import random_rois
    for n, R in [(1, 512), (8, 4096)]:
        feat = torch.randn(n, 75, 100, 512, device=dev).permute(0, 3, 1, 2)
        rois = random_rois(R, n, 600, 800, seed=1, edge_cases=False).to(dev)
        sc = torch.rand(R, device=dev) + 1
        for dt in (torch.float32, torch.bfloat16):
            f = feat.to(dt)
            ms = timeit(lambda: hip_ops.roi_pool_forward(f, rois, 0.125, (7, 7), roi_scale=sc, need_argmax=False))
            ms2 = timeit(lambda: hip_ops.roi_pool_forward(f, rois, 0.125, (7, 7), need_argmax=True))
            esz = 2 if dt == torch.bfloat16 else 4
            byts = n * 75 * 100 * 512 * esz + R * 512 * 49 * esz
            print(f"roi_pool nhwc {str(dt)[6:]:9s} n={n} R={R}: {ms*1e3:8.1f} us ({byts/ms/1e6:7.1f} GB/s alg)  with argmax {ms2*1e3:8.1f} us")
        fn = feat.float().contiguous()
        ms = timeit(lambda: hip_ops.roi_pool_forward(fn, rois, 0.125, (7, 7)), iters=5)
        print(f"roi_pool nchw float32 n={n} R={R}: {ms*1e3:8.1f} us")


if __name__ == "__main__":
    main()
