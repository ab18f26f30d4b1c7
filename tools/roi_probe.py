"""RoIPool / ROIAlign forward alone at the benchmark shape (16 images x 512 boxes, 512 channels, bf16 NHWC):
time per launch and effective gather rate.  python tools/roi_probe.py [iters] [C] [R] [N]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wsovod_amd._lib as _L
if os.environ.get("WSOVOD_LIB"):  # another build of the library for an A/B on one box
    _L.LIB_PATH = os.environ["WSOVOD_LIB"]
from wsovod_amd.data import make_batch
from wsovod_amd.layers import hip_ops as H

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
Cc = int(sys.argv[2]) if len(sys.argv) > 2 else 512
R = int(sys.argv[3]) if len(sys.argv) > 3 else 512
N = int(sys.argv[4]) if len(sys.argv) > 4 else 16
dev = torch.device("cuda:0")
host = make_batch(N, R, 20, seed=1)
boxes = torch.cat([x["proposals"].proposal_boxes.tensor for x in host]).to(dev)
obj = torch.cat([x["proposals"].objectness_logits for x in host]).to(dev)
seg = torch.tensor([0] + [R * (i + 1) for i in range(N)], dtype=torch.int32, device=dev)
rois, scale = H.format_rois(boxes, seg, obj)
feat = torch.randn(N, 75, 100, Cc, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)  # NHWC storage
cells = float(((boxes[:, 2] - boxes[:, 0]) / 8 + 1).mul((boxes[:, 3] - boxes[:, 1]) / 8 + 1).sum())


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


ms = timeit(lambda: H.roi_pool_forward(feat, rois, 0.125, (7, 7), roi_scale=scale, out_dtype=torch.bfloat16, need_argmax=False))
print(f"roi_pool  values only : {ms:.3f} ms  window reads {cells * Cc * 2 / ms / 1e9:.2f} TB/s, out {N*R*Cc*49*2/ms/1e9:.2f} TB/s")
ms = timeit(lambda: H.roi_pool_forward(feat, rois, 0.125, (7, 7), roi_scale=scale, out_dtype=torch.bfloat16, need_argmax=True))
print(f"roi_pool  + argmax    : {ms:.3f} ms")
ms = timeit(lambda: H.roi_align_forward(feat, rois, 0.125, (7, 7), 0, True, roi_scale=scale, out_dtype=torch.bfloat16))
print(f"roi_align aligned     : {ms:.3f} ms")

# the "parity" precision's pooler: fp32 map in, bf16x2 (+ plain bf16 copy) out
feat32 = torch.randn(N, 75, 100, Cc, device=dev).permute(0, 3, 1, 2)
ms = timeit(lambda: H.roi_pool_forward(feat32, rois, 0.125, (7, 7), roi_scale=scale, out_dtype=H.X2, need_argmax=False))
print(f"roi_pool  fp32 -> bf16x2        : {ms:.3f} ms  (WSOVOD_ROIPOOL_F32_CPL={os.environ.get('WSOVOD_ROIPOOL_F32_CPL', '4')})")
ms = timeit(lambda: H.roi_pool_forward(feat32, rois, 0.125, (7, 7), roi_scale=scale, out_dtype=H.X2, need_argmax=False, want_hi=True))
print(f"roi_pool  fp32 -> bf16x2 + bf16 : {ms:.3f} ms")
ms = timeit(lambda: H.roi_pool_forward(feat32, rois, 0.125, (7, 7), roi_scale=scale, out_dtype=torch.float32, need_argmax=False))
print(f"roi_pool  fp32 -> fp32          : {ms:.3f} ms")
ms = timeit(lambda: H.roi_align_forward(feat32, rois, 0.125, (7, 7), 0, True, roi_scale=scale, out_dtype=H.X2, want_hi=True))
print(f"roi_align fp32 -> bf16x2 + bf16 : {ms:.3f} ms")
if os.environ.get("ROI_PROBE_ALIGN_ONLY"):
    sys.exit(0)


# round 4: the 2x2-max-map path against the cell scan (switches are read per call)
print("--- 2x2-max map (WSOVOD_ROIPOOL_M2) / channels per lane / XCD-aware order ---")
def env(**kw):
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)

for m2, cpl, xcd in [(0, None, None), (1, 8, None), (1, 4, None), (1, 4, 1)]:
    env(WSOVOD_ROIPOOL_M2=m2, WSOVOD_ROIPOOL_BF16_CPL=cpl, WSOVOD_ROIPOOL_XCD=xcd)
    ms = timeit(lambda: H.roi_pool_forward(feat, rois, 0.125, (7, 7), roi_scale=scale, out_dtype=torch.bfloat16, need_argmax=False))
    print(f"bf16 -> bf16        M2={m2} CPL={cpl} XCD={xcd}: {ms:.3f} ms  ({N*R*Cc*49*2/ms/1e9 + N*Cc*7500*2/ms/1e9:.2f} TB/s algorithmic)")
for m2, cpl, xcd in [(0, None, None), (1, 4, None), (1, 2, None), (1, 2, 1), (1, 1, 1)]:
    env(WSOVOD_ROIPOOL_M2=m2, WSOVOD_ROIPOOL_F32_CPL=cpl, WSOVOD_ROIPOOL_XCD=xcd)
    ms = timeit(lambda: H.roi_pool_forward(feat32, rois, 0.125, (7, 7), roi_scale=scale, out_dtype=H.X2, need_argmax=False, want_hi=True))
    print(f"fp32 -> bf16x2+bf16 M2={m2} CPL={cpl} XCD={xcd}: {ms:.3f} ms")
env(WSOVOD_ROIPOOL_M2=None, WSOVOD_ROIPOOL_F32_CPL=None, WSOVOD_ROIPOOL_XCD=None, WSOVOD_ROIPOOL_BF16_CPL=None)
