"""-m gpu: proposal-side index kernels (NMS, RPN decode) through the C-ABI against the oracle.  Keep sets are
index work: bit-exact.  Decoded boxes go through expf: relative 1e-6."""
import pytest
import torch

from oracle import roi_ops
from tests.conftest import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


def clustered_boxes(n, seed, extent=800.0, clusters=24):
    g = torch.Generator().manual_seed(seed)
    centres = torch.rand(clusters, 2, generator=g) * extent
    which = torch.randint(0, clusters, (n,), generator=g)
    c = centres[which] + torch.randn(n, 2, generator=g) * 12.0
    wh = 20.0 + torch.rand(n, 2, generator=g) * 120.0
    boxes = torch.cat([c - wh / 2, c + wh / 2], dim=1).clamp(0, extent)
    dup = torch.rand(n, generator=g) < 0.05  # exact duplicates (IoU == 1) and degenerate boxes
    boxes[dup] = boxes[torch.randint(0, n, (int(dup.sum()),), generator=g)] if n else boxes[dup]
    if n > 3:
        boxes[3, 2] = boxes[3, 0]  # zero area
    return boxes


@pytest.mark.parametrize("sizes,thr,max_keep", [
    ([0], 0.5, 0), ([1], 0.5, 0), ([63, 64, 65], 0.7, 0), ([2048] * 4, 0.7, 1024), ([2048, 1500, 0, 7], 0.7, 1024),
    ([5000], 0.3, 0), ([300] * 40 + [0, 1, 2], 0.3, 100), ([4097, 130], 0.5, 0),
])
def test_nms_segments_matches_oracle(gpu, sizes, thr, max_keep):
    from wsovod_amd.layers import hip_ops as H

    offs = [0]
    for s in sizes:
        offs.append(offs[-1] + s)
    boxes = torch.cat([clustered_boxes(s, seed=11 + i) for i, s in enumerate(sizes)]) if offs[-1] else torch.zeros(0, 4)
    ref = roi_ops.nms_segments(boxes, offs, thr, max_keep)
    seg = torch.tensor(offs, dtype=torch.int32, device=gpu)
    keep, count = H.nms_segments(boxes.to(gpu), seg, max(sizes), thr, max_keep)
    count = count.cpu().tolist()
    keep = keep.cpu()
    assert count == [len(r) for r in ref]
    for g, r in enumerate(ref):
        assert torch.equal(keep[offs[g]:offs[g] + count[g]].to(torch.int64), r), g


def test_nms_valid_flags_and_property(gpu):
    """Filtered boxes are never kept and never suppress; kept boxes are mutually below the threshold and every
    dropped valid box overlaps an earlier kept one (the defining property of greedy NMS)."""
    from wsovod_amd.layers import hip_ops as H
    from oracle.wsovod_ref import pairwise_iou

    n = 3000
    boxes = clustered_boxes(n, seed=5)
    valid = torch.rand(n, generator=torch.Generator().manual_seed(6)) > 0.3
    ref = roi_ops.nms_segments(boxes, [0, n], 0.6, 0, valid=valid)[0]
    keep, count = H.nms_segments(boxes.to(gpu), torch.tensor([0, n], dtype=torch.int32, device=gpu), n, 0.6, 0,
                                 valid=valid.to(gpu))
    k = keep[: int(count[0])].cpu().to(torch.int64)
    assert torch.equal(k, ref) and bool(valid[k].all())
    iou = pairwise_iou(boxes[k], boxes[k])
    iou.fill_diagonal_(0)
    assert float(iou.max()) <= 0.6
    dropped = torch.tensor(sorted(set(torch.nonzero(valid).flatten().tolist()) - set(k.tolist())))
    cover = pairwise_iou(boxes[dropped], boxes[k])
    earlier = k[None, :] < dropped[:, None]
    assert bool(((cover > 0.6) & earlier).any(dim=1).all())


def test_nms_rejects_oversized_segment(gpu):
    from wsovod_amd.layers import hip_ops as H

    with pytest.raises(RuntimeError, match="not supported"):
        H.nms_segments(torch.zeros(8, 4, device=gpu), torch.tensor([0, 8], dtype=torch.int32, device=gpu), 20000, 0.5)


@pytest.mark.parametrize("with_index", [False, True])
def test_rpn_decode_matches_oracle(gpu, with_index):
    from wsovod_amd.layers import hip_ops as H
    from oracle import wsovod_ref as R

    g = torch.Generator().manual_seed(3)
    A, B = 5000, 3
    anchors = clustered_boxes(A, seed=9)
    anchors[:, 2:] = torch.maximum(anchors[:, 2:], anchors[:, :2] + 1.0)
    deltas = torch.randn(B, A, 4, generator=g) * 0.5
    deltas[0, 0, 2] = 50.0  # hits the scale clamp
    deltas[1, 5, 0] = float("inf")  # non-finite -> invalid
    sizes = torch.tensor([[600.0, 800.0], [480.0, 640.0], [333.0, 500.0]])
    index = torch.stack([torch.randperm(A, generator=g)[:777] for _ in range(B)]) if with_index else None
    weights, clamp, min_size = (1.0, 1.0, 1.0, 1.0), R.SCALE_CLAMP, 40.0
    boxes, valid = H.rpn_decode(anchors.to(gpu), deltas.to(gpu), index.to(gpu) if with_index else None, sizes.to(gpu),
                                weights, clamp, min_size)
    for b in range(B):
        d = deltas[b] if index is None else deltas[b][index[b]]
        a = anchors if index is None else anchors[index[b]]
        ref, ref_valid = R.rpn_decode_clip(a, d, sizes[b].tolist(), weights, min_size)
        fin = torch.isfinite(ref).all(dim=1)
        torch.testing.assert_close(boxes[b].cpu()[fin], ref[fin], rtol=1e-5, atol=1e-3)
        # the validity flag may legitimately differ only where a side is within rounding of min_size
        side = torch.minimum(ref[:, 2] - ref[:, 0], ref[:, 3] - ref[:, 1])
        clear = fin & ((side - min_size).abs() > 1e-2)
        assert torch.equal(valid[b].cpu()[clear], ref_valid[clear])
        assert not bool(valid[b].cpu()[~fin].any())


def test_batched_eval_tail_groups_of_images_equal_single_image_form(gpu):
    """Large vocabularies: the batched tail bounds its (images, classes, proposals) candidate tensors by working
    through the batch in groups of images.  Same detections as the per-image form, image by image."""
    import torch
    from wsovod_amd.modeling.fast_rcnn_open_vocabulary import (_fast_rcnn_inference_batched, fast_rcnn_inference,
                                                               fast_rcnn_inference_single_image)

    g = torch.Generator().manual_seed(12)
    N, R, K = 3, 1500, 3000  # 4.5 M candidates per image -> groups of one image
    shapes = [(600, 800), (480, 640), (600, 800)]
    boxes, scores = [], []
    for h, w in shapes:
        xy = torch.rand(R, 2, generator=g) * torch.tensor([w - 40.0, h - 40.0])
        wh = 16 + torch.rand(R, 2, generator=g) * torch.tensor([w / 3.0, h / 3.0])
        boxes.append(torch.cat([xy, xy + wh], dim=1).to(gpu))
        logits = torch.randn(R, K + 1, generator=g) * 4
        scores.append(torch.softmax(logits, dim=1).to(gpu))
    res, kept, all_s, all_b = fast_rcnn_inference(boxes, scores, shapes, 1e-3, 0.3, 100)
    assert getattr(res, "packed", None) is None and len(res) == N  # went through in groups
    for i in range(N):
        one, k1, _, _ = fast_rcnn_inference_single_image(boxes[i], scores[i], shapes[i], 1e-3, 0.3, 100)
        assert len(one) == len(res[i]) > 0
        assert torch.equal(one.pred_classes, res[i].pred_classes) and torch.equal(one.pred_inds, res[i].pred_inds)
        assert torch.equal(one.pred_boxes.tensor, res[i].pred_boxes.tensor) and torch.equal(one.scores, res[i].scores)
        assert torch.equal(k1, kept[i])
    two = _fast_rcnn_inference_batched(boxes[:2], scores[:2], shapes[:2], 1e-3, 0.3, 100)[0]
    assert two.packed is not None and torch.equal(two[1].pred_inds, res[1].pred_inds)
