"""Where the RoIPool launch spends its time (32 x 512 boxes, 512 channels, bf16 NHWC -> bf16): the per-kernel table of the
library's hipEvent profiler (2x2-max pre-pass vs pooling kernel), and the pooling kernel on degenerate boxes -- 2 x 2
cells (no scan to speak of: the fixed cost of roi decode + LDS transpose + 822 MB of stores) and 1-cell boxes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wsovod_amd._lib as _L
from wsovod_amd.data import make_batch
from wsovod_amd.layers import hip_ops as H

N, R, Cc = 32, 512, 512
dev = torch.device("cuda:0")
host = make_batch(N, R, 20, seed=1)
boxes = torch.cat([x["proposals"].proposal_boxes.tensor for x in host]).to(dev)
obj = torch.cat([x["proposals"].objectness_logits for x in host]).to(dev)
seg = torch.tensor([0] + [R * (i + 1) for i in range(N)], dtype=torch.int32, device=dev)
feat = torch.randn(N, 75, 100, Cc, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)
feat32 = torch.randn(N, 75, 100, Cc, device=dev).permute(0, 3, 1, 2)


def table(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    _L.profile_reset(); _L.profile_enable(True)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    t = _L.profile_collect(); _L.profile_enable(False)
    return {e["name"]: e["ms"] / max(e["launches"], 1) for e in t if e["launches"]}


for name, bx in [("bench boxes", boxes),
                 ("2x2-cell boxes", torch.cat([boxes[:, :2], boxes[:, :2] + 12.0], 1)),
                 ("24x24-cell boxes", torch.cat([boxes[:, :2].clamp(max=300), boxes[:, :2].clamp(max=300) + 190.0], 1)),
                 ]:
    rois, scale = H.format_rois(bx.contiguous(), seg, obj)
    for m2 in ("1", "0"):
        os.environ["WSOVOD_ROIPOOL_M2"] = m2
        t = table(lambda: H.roi_pool_forward(feat, rois, 0.125, (7, 7), roi_scale=scale, out_dtype=torch.bfloat16, need_argmax=False))
        print(f"{name:18s} bf16 M2={m2}:", {k: round(v, 3) for k, v in t.items() if "roi" in k})
        t = table(lambda: H.roi_pool_forward(feat32, rois, 0.125, (7, 7), roi_scale=scale, out_dtype=H.X2, need_argmax=False, want_hi=True))
        print(f"{name:18s} fp32->x2+hi M2={m2}:", {k: round(v, 3) for k, v in t.items() if "roi" in k})
# plain write of the same bytes for reference
out = torch.empty(N * R * Cc * 49, dtype=torch.bfloat16, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out.zero_(); torch.cuda.synchronize(); e0.record()
for _ in range(10):
    out.zero_()
e1.record(); torch.cuda.synchronize()
print(f"memset of the 822 MB output: {e0.elapsed_time(e1) / 10:.3f} ms")
