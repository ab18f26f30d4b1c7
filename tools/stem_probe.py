"""The fused stem conv1 (uint8 canvas -> relu(conv1), bf16 and bf16x2 forms) alone at the headline shape (32 images of
800x600); WSOVOD_LIB=<path> loads another build of the library for an A/B on one box."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import wsovod_amd._lib as _L
if os.environ.get("WSOVOD_LIB"):
    _L.LIB_PATH = os.environ["WSOVOD_LIB"]
from wsovod_amd.layers import hip_ops as H
n, Hh, Ww = 32, 600, 800
torch.manual_seed(0)
img = torch.randint(0, 256, (n, 3, Hh, Ww), dtype=torch.uint8, device="cuda")
sizes = torch.tensor([[Hh, Ww]] * n, dtype=torch.int32, device="cuda")
w32 = (torch.randn(64, 32, device="cuda") * 0.05)
w32[:, 27:] = 0
bias = torch.randn(64, device="cuda")
mean, std = [103.53, 116.28, 123.675], [57.375, 57.12, 58.395]
for name, fn in (("bf16", lambda: H.stem_conv1(img, sizes, mean, std, w32.to(torch.bfloat16).contiguous(), bias)),
                 ("bf16x2", lambda: H.stem_conv1_x2(img, sizes, mean, std, H.x2_encode(w32), bias))):
    out = fn()
    ts = []
    for r in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 5)
    med = sorted(ts)[3]
    chk = int(out.view(torch.int16).to(torch.int64).sum())
    nbytes = img.numel() + out.numel() * out.element_size()
    print(f"stem {name}: {med:.3f} ms  {nbytes / med / 1e6:.0f} GB/s  checksum {chk}", flush=True)
