"""Does the ORDER in which rois are pooled matter?    python tools/roi_order_probe.py [N images]

The training pooler (fp32 res5 map -> planar bf16x2, values only, through the 2x2-max map) at the benchmark shape, with the
same boxes handed over (a) in the data loader's order (sorted by objectness: spatially random), (b) sorted by image, then by
the row and column of the window origin in coarse cells (workgroups resident together read overlapping windows), (c) sorted by
window origin on a Z-order curve.  Output rows follow the order given, so (b) / (c) measure what a permuted PROCESSING order
could gain before anything is built for it."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsovod_amd.data import make_batch
from wsovod_amd.layers import hip_ops as H

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
R, Cc, iters = 512, 512, 20
dev = torch.device("cuda:0")
host = make_batch(N, R, 20, seed=1)
boxes = torch.cat([x["proposals"].proposal_boxes.tensor for x in host]).to(dev)
obj = torch.cat([x["proposals"].objectness_logits for x in host]).to(dev)
seg = torch.tensor([0] + [R * (i + 1) for i in range(N)], dtype=torch.int32, device=dev)
rois, scale = H.format_rois(boxes, seg, obj)
feat32 = torch.randn(N, 75, 100, Cc, device=dev).permute(0, 3, 1, 2)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def run(name, order):
    r, s = rois[order].contiguous(), scale[order].contiguous()
    ms = timeit(lambda: H.roi_pool_forward(feat32, r, 0.125, (7, 7), roi_scale=s, out_dtype=H.X2, need_argmax=False, want_hi=True))
    print(f"{name:48s} {ms:.3f} ms")


def zorder(x, y):
    z = torch.zeros_like(x)
    for b in range(7):
        z |= ((x >> b) & 1) << (2 * b) | ((y >> b) & 1) << (2 * b + 1)
    return z


img = rois[:, 0].long()
x0, y0 = (rois[:, 1] / 8).round().long(), (rois[:, 2] / 8).round().long()
ident = torch.arange(rois.size(0), device=dev)
run("loader order (by objectness)", ident)
for cell in (4, 8, 16):
    key = (img * 64 + y0 // cell) * 256 + x0 // cell
    run(f"sorted by image, origin row / col in {cell}-cell tiles", torch.argsort(key, stable=True))
run("sorted by image, Z-order of the origin (4-cell)", torch.argsort(img * (1 << 14) + zorder(x0 // 4, y0 // 4), stable=True))
cx, cy = ((rois[:, 1] + rois[:, 3]) / 16).long(), ((rois[:, 2] + rois[:, 4]) / 16).long()
run("sorted by image, Z-order of the CENTRE (4-cell)", torch.argsort(img * (1 << 14) + zorder(cx // 4, cy // 4), stable=True))
area = ((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
run("sorted by image, then by box area", torch.argsort(img.double() * 1e7 + area.double(), stable=True))
