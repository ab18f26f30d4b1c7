"""CPU: independent checks of the pieces whose reference lives in un-vendored dependencies (detectron2 / torchvision;
SURVEY.md 8c) and that tests/golden/make_golden.py therefore plugs into the reference as the repo's own restatements.
Nothing here calls the code under test to produce its own expectation: each expectation is written from the
operator's definition (numpy / python loops, fp64)."""
import math

import numpy as np
import pytest
import torch

from oracle import roi_ops
from oracle import wsovod_ref as R
from tests.util import random_rois
from wsovod_amd.modeling.box_regression import Box2BoxTransform
from wsovod_amd.modeling.matcher import Matcher
from wsovod_amd.modeling.sampling import subsample_labels
from wsovod_amd.structures import Boxes, pairwise_iou


def _tent_roi_align(feat, rois, scale, out_hw, sampling_ratio, aligned):
    """ROIAlign from its definition in fp64.  Box -> continuous map coordinates (x*scale - 0.5 when aligned), each of
    the out_h x out_w bins is sampled on a regular g_h x g_w grid (g = sampling_ratio, or ceil(roi size / out size)
    when 0) and averaged; a sample's value is the bilinear interpolation of the map with coordinates clamped to
    [0, size-1], and 0 when the sample lies more than one cell outside the map.  Bilinear interpolation is written
    as the tent-kernel sum  f(y,x) = sum_ij max(0,1-|y-i|) max(0,1-|x-j|) F[i,j]  -- not as the corner/weight
    bookkeeping of the C oracle or the kernels."""
    feat = feat.double().numpy()
    N, C, H, W = feat.shape
    ph, pw = out_hw
    out = np.zeros((len(rois), C, ph, pw))
    ii, jj = np.arange(H)[:, None], np.arange(W)[None, :]
    for n, roi in enumerate(rois.double().numpy()):
        b = int(roi[0])
        off = 0.5 if aligned else 0.0
        x0, y0, x1, y1 = (roi[1:] * scale - off)
        rw, rh = x1 - x0, y1 - y0
        if not aligned:
            rw, rh = max(rw, 1.0), max(rh, 1.0)
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / ph))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / pw))
        cnt = max(gh * gw, 1)
        for p in range(ph):
            for q in range(pw):
                acc = np.zeros(C)
                for iy in range(gh):
                    y = y0 + (p + (iy + 0.5) / gh) * rh / ph
                    for ix in range(gw):
                        x = x0 + (q + (ix + 0.5) / gw) * rw / pw
                        if y < -1.0 or y > H or x < -1.0 or x > W:
                            continue
                        yc, xc = min(max(y, 0.0), H - 1.0), min(max(x, 0.0), W - 1.0)
                        k = np.maximum(0.0, 1.0 - np.abs(yc - ii)) * np.maximum(0.0, 1.0 - np.abs(xc - jj))
                        acc += (feat[b] * k[None]).sum(axis=(1, 2))
                out[n, :, p, q] = acc / cnt
    return torch.from_numpy(out)


@pytest.mark.parametrize("sampling_ratio,aligned", [(0, True), (2, True), (0, False)])
def test_roi_align_c_oracle_against_fp64_definition(sampling_ratio, aligned):
    """oracle/roi_ops_ref.c:roi_align_forward (restated torchvision roi_align; torchvision itself is absent) against
    the fp64 evaluator above, incl. boxes outside the map, boxes smaller than one cell (aligned: no clamp of the roi
    size to 1), whole-map boxes and the adaptive sampling grid."""
    feat = torch.randn(2, 3, 19, 25, generator=torch.Generator().manual_seed(1))
    rois = random_rois(24, 2, 19 * 8, 25 * 8, seed=4)  # rows 0-7: outside / zero-size / malformed / whole map / sub-cell
    rois = torch.cat([rois, torch.tensor([[0, 40.0, 40.0, 40.5, 40.25], [1, -30.0, 20.0, 260.0, 30.0],
                                          [0, 100.0, -20.0, 101.0, 200.0]])])
    got = roi_ops.roi_align_forward(feat, rois, 0.125, (7, 7), sampling_ratio, aligned)
    want = _tent_roi_align(feat, rois, 0.125, (7, 7), sampling_ratio, aligned)
    assert got.shape == want.shape
    err = (got.double() - want).abs().max()
    assert float(err) < 2e-5, float(err)
    assert float(want.abs().max()) > 0.5  # the comparison is not vacuous
    # rows fully outside the map are exactly zero in both
    assert float(got[0].abs().max()) == 0.0 and float(want[0].abs().max()) == 0.0


def test_roi_align_backward_is_the_adjoint_of_forward():
    """<forward(F), G> == <F, backward(G)>: the C oracle's backward scatters with the same weights the forward
    gathers with (linearity property; no second implementation needed)."""
    g = torch.Generator().manual_seed(2)
    feat = torch.randn(2, 3, 15, 21, generator=g)
    rois = random_rois(16, 2, 120, 168, seed=6)
    out = roi_ops.roi_align_forward(feat, rois, 0.125, (7, 7), 0, True)
    G = torch.randn(out.shape, generator=g)
    gi = roi_ops.roi_align_backward(G, rois, 0.125, 0, True, tuple(feat.shape))
    lhs, rhs = float((out.double() * G.double()).sum()), float((feat.double() * gi.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))


def _iou_scalar(a, b):
    iw = min(a[2], b[2]) - max(a[0], b[0])
    ih = min(a[3], b[3]) - max(a[1], b[1])
    if iw <= 0 or ih <= 0:
        return 0.0
    inter = iw * ih
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)


def test_pairwise_iou_against_brute_force():
    g = torch.Generator().manual_seed(3)
    xy = torch.rand(40, 2, generator=g) * 100
    a = torch.cat([xy, xy + torch.rand(40, 2, generator=g) * 60 + 1], 1)
    xy = torch.rand(17, 2, generator=g) * 100
    b = torch.cat([xy, xy + torch.rand(17, 2, generator=g) * 60 + 1], 1)
    b[0] = a[0]  # identical box -> 1
    b[1] = torch.tensor([500.0, 500.0, 510.0, 510.0])  # disjoint -> 0
    want = torch.tensor([[_iou_scalar(x.tolist(), y.tolist()) for y in b] for x in a], dtype=torch.float64)
    for got in (pairwise_iou(Boxes(a), Boxes(b)), R.pairwise_iou(a, b)):
        torch.testing.assert_close(got.double(), want, rtol=1e-5, atol=1e-6)
    assert float(pairwise_iou(Boxes(a), Boxes(b))[0, 0]) == pytest.approx(1.0) and float(want[:, 1].max()) == 0.0


def test_box2box_transform_round_trip_and_definition():
    g = torch.Generator().manual_seed(4)
    xy = torch.rand(64, 2, generator=g) * 200
    src = torch.cat([xy, xy + torch.rand(64, 2, generator=g) * 150 + 2], 1)
    xy = torch.rand(64, 2, generator=g) * 200
    tgt = torch.cat([xy, xy + torch.rand(64, 2, generator=g) * 150 + 2], 1)
    for impl_get, impl_apply in (
            (Box2BoxTransform((10.0, 10.0, 5.0, 5.0)).get_deltas, Box2BoxTransform((10.0, 10.0, 5.0, 5.0)).apply_deltas),
            (R.box2box_get_deltas, R.box2box_apply_deltas)):
        d = impl_get(src, tgt)
        back = impl_apply(d, src)
        torch.testing.assert_close(back, tgt, rtol=1e-4, atol=1e-3)  # apply_deltas(get_deltas(a, b), a) == b
        # definition, element 0, by hand (R-CNN parameterisation with weights (10,10,5,5))
        sw, sh = float(src[0, 2] - src[0, 0]), float(src[0, 3] - src[0, 1])
        tw, th = float(tgt[0, 2] - tgt[0, 0]), float(tgt[0, 3] - tgt[0, 1])
        want = [10 * ((float(tgt[0, 0]) + tw / 2) - (float(src[0, 0]) + sw / 2)) / sw,
                10 * ((float(tgt[0, 1]) + th / 2) - (float(src[0, 1]) + sh / 2)) / sh,
                5 * math.log(tw / sw), 5 * math.log(th / sh)]
        torch.testing.assert_close(d[0].double(), torch.tensor(want, dtype=torch.float64), rtol=1e-4, atol=1e-5)
        # scale clamp: a huge dw is cut at log(1000/16) before exp
        big = impl_apply(torch.tensor([[0.0, 0.0, 500.0, 0.0]]), torch.tensor([[0.0, 0.0, 16.0, 16.0]]))
        assert float(big[0, 2] - big[0, 0]) == pytest.approx(1000.0, rel=1e-4)


def test_matcher_against_five_line_definition():
    g = torch.Generator().manual_seed(5)
    iou = torch.rand(6, 200, generator=g)
    iou[:, 7] = 0.0
    iou[2, 9] = iou[4, 9] = 0.9  # tie: the first maximum wins
    for thresholds, labels in (([0.5], [0, 1]), ([0.1, 0.5], [-1, 0, 1]), ([0.3, 0.7], [0, -1, 1])):
        matches, lab = Matcher(thresholds, labels, allow_low_quality_matches=False)(iou)
        q = iou.numpy()
        best = q.argmax(axis=0)
        val = q.max(axis=0)
        edges = [-np.inf] + thresholds + [np.inf]
        want = np.array([labels[max(i for i in range(len(labels)) if v >= edges[i])] for v in val])
        assert np.array_equal(matches.numpy(), best) and np.array_equal(lab.numpy(), want)
    m, l = Matcher([0.5], [0, 1])(torch.zeros(0, 9))  # no ground truth: everything background, index 0
    assert m.tolist() == [0] * 9 and l.tolist() == [0] * 9
    # low-quality matches (RPN): every GT's best anchor(s) become positive
    m, l = Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=True)(iou * 0.25)
    for gt in range(6):
        assert bool((l[(iou[gt] == iou[gt].max())] == 1).all())


def test_subsample_labels_quotas_and_membership():
    torch.manual_seed(0)
    K = 20
    lab = torch.full((500,), K, dtype=torch.int64)
    lab[torch.randperm(500)[:60]] = 4
    lab[torch.randperm(500)[:30]] = -1
    pos_set = set(((lab != K) & (lab != -1)).nonzero().flatten().tolist())
    neg_set = set((lab == K).nonzero().flatten().tolist())
    for num, frac in ((128, 0.25), (64, 1.0), (4096, 1.0), (100, 0.0)):
        p, n = subsample_labels(lab, num, frac, K)
        assert len(p) == min(len(pos_set), int(num * frac)) and len(n) == min(len(neg_set), num - len(p))
        assert set(p.tolist()) <= pos_set and set(n.tolist()) <= neg_set
        assert len(set(p.tolist())) == len(p) and len(set(n.tolist())) == len(n)  # without replacement


def _mask_roi_loop_pool(feat, rois, scale, out_hw, ratio=1.8):
    """The reference's 3-output ROILoopPool (layers/ROILoopPool/ROILoopPool_cuda.cu:9-204; its CPU source implements
    only the first output and the .cu file needs CUDA headers, so it cannot be built here) from its definition, with
    window slices and boolean masks instead of the kernel's scalar loops.
      region : max(0, max over the bin)                       (accumulator starts at 0: inputs are post-ReLU)
      frame  : the same bin without the cells STRICTLY inside the roi shrunk by `ratio` about its centre
      context: the bin of the roi GROWN by `ratio`, without the cells strictly inside the roi itself
    argmax = first cell in row-major order that reaches the (strictly positive) maximum, else -1."""
    f = feat.numpy()
    N, C, H, W = f.shape
    ph, pw = out_hw
    R = len(rois)
    out = np.zeros((3 * R, C, ph, pw), dtype=np.float32)
    arg = np.full((3 * R, C, ph, pw), -1, dtype=np.int32)
    f32 = np.float32

    def rnd(v):  # C round(): half away from zero
        return int(np.sign(v) * np.floor(np.abs(np.float32(v)) + np.float32(0.5)))

    def pool(n, b, rect, hole, slot):
        x1, y1, x2, y2 = rect
        sw, sh, ew, eh = rnd(f32(x1) * f32(scale)), rnd(f32(y1) * f32(scale)), rnd(f32(x2) * f32(scale)), rnd(f32(y2) * f32(scale))
        hx1, hy1, hx2, hy2 = (rnd(f32(v) * f32(scale)) for v in hole) if hole is not None else (0, 0, 0, 0)
        rw, rh = max(ew - sw + 1, 1), max(eh - sh + 1, 1)
        bh, bw = f32(rh) / f32(ph), f32(rw) / f32(pw)
        for p in range(ph):
            for q in range(pw):
                hs = min(max(int(np.floor(f32(p) * bh)) + sh, 0), H)
                he = min(max(int(np.ceil(f32(p + 1) * bh)) + sh, 0), H)
                ws = min(max(int(np.floor(f32(q) * bw)) + sw, 0), W)
                we = min(max(int(np.ceil(f32(q + 1) * bw)) + sw, 0), W)
                if he <= hs or we <= ws:
                    continue
                win = f[b, :, hs:he, ws:we].copy()
                if hole is not None:
                    hh, ww = np.arange(hs, he)[:, None], np.arange(ws, we)[None, :]
                    inside = (hh > hy1) & (hh < hy2) & (ww > hx1) & (ww < hx2)
                    win[:, inside] = -np.inf
                flat = win.reshape(C, -1)
                best = flat.max(axis=1)
                first = flat.argmax(axis=1)  # first maximum in row-major order
                hit = best > 0
                out[slot * R + n, hit, p, q] = best[hit]
                idx = (hs + first // (we - ws)) * W + ws + first % (we - ws)
                arg[slot * R + n, hit, p, q] = idx[hit]

    lim_x, lim_y = f32(1.0 * W / scale), f32(1.0 * H / scale)
    for n, roi in enumerate(rois.numpy().astype(np.float32)):
        b, x1, y1, x2, y2 = int(roi[0]), *roi[1:]
        rw, rh = x2 - x1, y2 - y1
        iw, ih = rw - rw / f32(ratio), rh - rh / f32(ratio)
        ow, oh = rw * f32(ratio) - rw, rh * f32(ratio) - rh
        clipx = lambda v: min(max(v, f32(0)), lim_x)
        clipy = lambda v: min(max(v, f32(0)), lim_y)
        inner = (clipx(x1 + iw / 2), clipy(y1 + ih / 2), clipx(x2 - iw / 2), clipy(y2 - ih / 2))
        outer = (clipx(x1 - ow / 2), clipy(y1 - oh / 2), clipx(x2 + ow / 2), clipy(y2 + oh / 2))
        pool(n, b, (x1, y1, x2, y2), None, 0)
        pool(n, b, (x1, y1, x2, y2), inner, 1)
        pool(n, b, outer, (x1, y1, x2, y2), 2)
    return torch.from_numpy(out), torch.from_numpy(arg)


def test_roi_loop_pool_c_oracle_against_mask_definition():
    """oracle/roi_ops_ref.c:roi_loop_pool_forward (frame / context outputs) against the mask-based evaluator above:
    values and argmax exact, on post-ReLU-like features (the op's stated assumption) incl. zero cells and boxes that
    leave the map."""
    g = torch.Generator().manual_seed(8)
    feat = torch.relu(torch.randn(2, 5, 38, 50, generator=g))
    rois = random_rois(40, 2, 38 * 8, 50 * 8, seed=12)
    got, got_arg = roi_ops.roi_loop_pool_forward(feat, rois, 0.125, (7, 7))
    want, want_arg = _mask_roi_loop_pool(feat, rois, 0.125, (7, 7))
    R = len(rois)
    for slot, name in enumerate(("region", "frame", "context")):
        sl = slice(slot * R, (slot + 1) * R)
        assert torch.equal(got[sl], want[sl]), name
        assert torch.equal(got_arg[sl], want_arg[sl]), name
    assert float(want[R:2 * R].sum()) > 0 and float(want[2 * R:].sum()) > 0
    assert not torch.equal(want[:R], want[R:2 * R]) and not torch.equal(want[:R], want[2 * R:])
