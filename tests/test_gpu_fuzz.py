"""GPU parity, randomised sweeps: every case draws shape, layout, dtype and box statistics from a seeded generator, so the
kernels meet sizes no hand-written case names (odd channel counts, 1-cell maps, boxes far outside the image, ragged
segments with empty images, reduction lengths that end inside a tile).  Same bars as the fixed cases: indices and copied
values bit-exact, fp32 arithmetic to the stated tolerance.  The checker is the CPU oracle (test infrastructure)."""
import pytest
import torch

from oracle import roi_ops as O

pytestmark = pytest.mark.gpu


def _rois(g, R, n_img, Hi, Wi):
    """Boxes in image pixels: ordinary ones plus a random third of degenerate forms (outside, inverted, sub-cell,
    huge, exactly on .5 cell boundaries)."""
    x0 = (torch.rand(R, generator=g) * 1.4 - 0.2) * Wi
    y0 = (torch.rand(R, generator=g) * 1.4 - 0.2) * Hi
    w = torch.rand(R, generator=g) ** 2 * Wi
    h = torch.rand(R, generator=g) ** 2 * Hi
    kind = torch.randint(0, 9, (R,), generator=g)
    w = torch.where(kind == 0, torch.zeros_like(w), w)              # zero width
    h = torch.where(kind == 1, -h, h)                               # inverted
    w = torch.where(kind == 2, torch.full_like(w, 3.0), w)          # smaller than a cell at scale 1/8
    x0 = torch.where(kind == 3, (x0 / 4).round() * 4 + 0.5 * 8, x0)  # lands on x.5 after the 1/8 scale
    b = torch.randint(0, n_img, (R,), generator=g).float()
    return torch.stack([b, x0, y0, x0 + w, y0 + h], 1)


@pytest.mark.parametrize("seed", range(10))
def test_roi_pool_fuzz_bit_exact(gpu, seed):
    from wsovod_amd.layers import hip_ops as H

    g = torch.Generator().manual_seed(1000 + seed)
    pick = lambda xs: xs[int(torch.randint(0, len(xs), (1,), generator=g))]
    C = pick([1, 5, 32, 64, 100, 130, 256, 512])
    Hf, Wf = int(torch.randint(1, 48, (1,), generator=g)), int(torch.randint(1, 64, (1,), generator=g))
    n_img, R = pick([1, 2, 5]), pick([1, 3, 64, 257])
    scale = pick([0.125, 0.0625, 0.25, 0.37])
    size = (pick([1, 2, 7, 9]), pick([1, 3, 7, 8]))
    dtype, cl = pick([torch.float32, torch.bfloat16]), pick([False, True])
    feat = torch.randn(n_img, C, Hf, Wf, generator=g).to(dtype)
    feat[torch.rand(feat.shape, generator=g) < 0.05] = float("-inf")  # -inf cells: an all -inf bin must stay -inf
    rois = _rois(g, R, n_img, Hf / scale, Wf / scale)
    ref_out, ref_arg = O.roi_pool_forward(feat.float(), rois, scale, size)
    f = feat.to(gpu)
    if cl:
        f = f.contiguous(memory_format=torch.channels_last)
    out, arg = H.roi_pool_forward(f, rois.to(gpu), scale, size, out_dtype=torch.float32)
    assert torch.equal(arg.cpu(), ref_arg), (C, Hf, Wf, size, scale)
    assert torch.equal(out.cpu(), ref_out)
    # backward: scatter by argmax -- fp32 atomic adds in another order than the oracle's loop
    go = torch.randn(ref_out.shape, generator=g)
    ref_gi = O.roi_pool_backward(go, rois, ref_arg, feat.shape)
    gi = H.roi_pool_backward(go.to(gpu), rois.to(gpu), arg, feat.shape, channels_last=cl)
    torch.testing.assert_close(gi.cpu().contiguous(), ref_gi, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("seed", range(8))
def test_roi_align_fuzz(gpu, seed):
    from wsovod_amd.layers import hip_ops as H

    g = torch.Generator().manual_seed(2000 + seed)
    pick = lambda xs: xs[int(torch.randint(0, len(xs), (1,), generator=g))]
    C = pick([1, 6, 64, 96, 256])
    Hf, Wf = int(torch.randint(1, 40, (1,), generator=g)), int(torch.randint(1, 50, (1,), generator=g))
    n_img, R = pick([1, 3]), pick([1, 2, 65, 200])
    scale, ratio, aligned = pick([0.125, 0.0625, 0.3]), pick([0, 1, 2, 3]), pick([False, True])
    size = (pick([1, 7]), pick([2, 7]))
    cl = pick([False, True])
    feat = torch.randn(n_img, C, Hf, Wf, generator=g)
    rois = _rois(g, R, n_img, Hf / scale, Wf / scale)
    ref = O.roi_align_forward(feat, rois, scale, size, ratio, aligned)
    f = feat.to(gpu)
    if cl:
        f = f.contiguous(memory_format=torch.channels_last)
    out = H.roi_align_forward(f, rois.to(gpu), scale, size, ratio, aligned)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=5e-5)  # fp32 bilinear; FMA contraction differs
    go = torch.randn(ref.shape, generator=g)
    ref_gi = O.roi_align_backward(go, rois, scale, ratio, aligned, feat.shape)
    gi = H.roi_align_backward(go.to(gpu), rois.to(gpu), scale, ratio, aligned, feat.shape, channels_last=cl)
    # adaptive sampling (ratio 0) on a huge box averages thousands of samples per bin: scale the bar with the gradient
    torch.testing.assert_close(gi.cpu().contiguous(), ref_gi, rtol=1e-4, atol=1e-4 * max(1.0, float(ref_gi.abs().max())))


def _greedy_nms(boxes, thr):
    """O(n^2) greedy NMS on score-sorted boxes, written from the definition (keep a box iff no kept box overlaps it by
    more than thr); IoU in fp32 with the reference's area / intersection formulas."""
    keep = []
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    for i in range(len(boxes)):
        ok = True
        for j in keep:
            iw = torch.clamp(torch.min(boxes[i, 2], boxes[j, 2]) - torch.max(boxes[i, 0], boxes[j, 0]), min=0)
            ih = torch.clamp(torch.min(boxes[i, 3], boxes[j, 3]) - torch.max(boxes[i, 1], boxes[j, 1]), min=0)
            inter = iw * ih
            iou = inter / (area[i] + area[j] - inter)
            if float(iou) > thr:
                ok = False
                break
        if ok:
            keep.append(i)
    return keep


@pytest.mark.parametrize("seed", range(6))
def test_nms_fuzz_against_definition(gpu, seed):
    from wsovod_amd.layers import hip_ops as H

    g = torch.Generator().manual_seed(3000 + seed)
    sizes = [int(s) for s in torch.randint(0, 150, (int(torch.randint(1, 6, (1,), generator=g)),), generator=g)]
    thr = [0.3, 0.5, 0.7][seed % 3]
    segs = []
    for s in sizes:
        c = torch.rand(s, 2, generator=g) * 60
        wh = torch.rand(s, 2, generator=g) * 40 + 1
        bx = torch.cat([c, c + wh], 1)
        if s > 4:
            bx[1] = bx[0]            # exact duplicate: IoU = 1
            bx[3, 2:] = bx[3, :2]    # zero area: IoU 0/0 against itself, 0 against the rest
        segs.append(bx)
    boxes = torch.cat(segs) if sum(sizes) else torch.zeros(0, 4)
    offs = [0]
    for s in sizes:
        offs.append(offs[-1] + s)
    keep, count = H.nms_segments(boxes.to(gpu), torch.tensor(offs, dtype=torch.int32, device=gpu), max(sizes + [1]), thr)
    keep, count = keep.cpu().tolist(), count.cpu().tolist()
    ref = O.nms_segments(boxes, offs, thr)
    for k, s in enumerate(sizes):
        got = keep[offs[k]:offs[k] + count[k]]
        assert got == _greedy_nms(segs[k], thr), (k, s)
        assert got == ref[k].tolist()


@pytest.mark.parametrize("seed", range(10))
def test_gemm_nt_fuzz(gpu, seed):
    """Ragged M / N / K (K only whole 16-byte groups, the ABI's rule), every epilogue term on or off at random, bf16 and
    exact-fp32 MFMA paths, against an fp64 contraction."""
    from wsovod_amd.layers import hip_ops as H

    g = torch.Generator().manual_seed(4000 + seed)
    pick = lambda xs: xs[int(torch.randint(0, len(xs), (1,), generator=g))]
    dtype = pick([torch.float32, torch.bfloat16])
    M = pick([1, 15, 64, 257, 700, 1031])
    N = pick([1, 8, 63, 128, 300, 513])
    K = pick([1, 2, 7, 40, 129]) * (4 if dtype == torch.float32 else 8)
    A = (torch.rand(M, K, generator=g) * 2 - 1).to(dtype)
    B = (torch.rand(N, K, generator=g) * 2 - 1).to(dtype)
    bias = torch.randn(N, generator=g) if pick([0, 1]) else None
    res = torch.randn(M, N, generator=g) if pick([0, 1]) else None
    relu, alpha = bool(pick([0, 1])), pick([1.0, 0.5, -2.0])
    ref = alpha * (A.double() @ B.double().t())
    if bias is not None:
        ref = ref + bias.double()
    if res is not None:
        ref = ref + res.double()
    if relu:
        ref = ref.clamp(min=0)
    out = H.gemm_nt(A.to(gpu), B.to(gpu), alpha=alpha, bias=None if bias is None else bias.to(gpu),
                    residual=None if res is None else res.to(gpu), relu=relu, out_dtype=torch.float32)
    # fp32 accumulation of K products of magnitude <= 1 (inputs are exact in both dtypes)
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-5, atol=2e-6 * K ** 0.5 * abs(alpha) + 1e-6)


@pytest.mark.parametrize("seed", range(6))
def test_mil_forward_fuzz_ragged_segments(gpu, seed):
    """softmax over classes x softmax over the proposals of each image (fast_rcnn_open_vocabulary.py:342-354) with
    ragged segments: images with one proposal, with thousands, and logits spread over +-30."""
    from wsovod_amd.layers import hip_ops as H

    g = torch.Generator().manual_seed(5000 + seed)
    K = [3, 2, 20, 80, 300, 1203][seed]
    nums = [int(n) for n in torch.randint(1, 40, (int(torch.randint(1, 7, (1,), generator=g)),), generator=g)]
    nums[0] = 1
    if seed % 2:
        nums.append(2500)
    R = sum(nums)
    logits = torch.randn(R, 2 * K, generator=g) * [1.0, 10.0, 30.0][seed % 3]
    offs = [0]
    for n in nums:
        offs.append(offs[-1] + n)
    seg = torch.tensor(offs, dtype=torch.int32, device=gpu)
    got = H.mil_forward(logits.to(gpu), seg, K)
    scores = got[0] if isinstance(got, (tuple, list)) else got
    c, d = logits[:, :K].double(), logits[:, K:].double()
    ref = torch.cat([torch.softmax(c[a:b], 1) * torch.softmax(d[a:b], 0) for a, b in zip(offs[:-1], offs[1:])])
    torch.testing.assert_close(scores.cpu().double(), ref, rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("seed", range(12))
def test_conv_implicit_gemm_fuzz(gpu, seed):
    """conv2d + bias + (residual) + ReLU as one implicit GEMM on random geometry (kernel 1 / 3, stride 1 / 2, dilation
    1 / 2 / 4, with and without padding, maps a few pixels wide, output channels off every tile size) against fp64."""
    import torch.nn.functional as F
    from wsovod_amd.layers import hip_ops as H

    g = torch.Generator().manual_seed(6000 + seed)
    pick = lambda xs: xs[int(torch.randint(0, len(xs), (1,), generator=g))]
    dtype = pick([torch.float32, torch.bfloat16])
    Cin, Cout = pick([64, 128, 192]), pick([8, 64, 100, 256, 300])
    k, stride, dil = pick([1, 3]), pick([1, 2]), pick([1, 2, 4])
    pad = pick([0, dil * (k // 2)])
    n = pick([1, 2, 3])
    Hh = int(torch.randint(dil * (k - 1) + 1, 46, (1,), generator=g))
    Ww = int(torch.randint(dil * (k - 1) + 1, 46, (1,), generator=g))
    x = (torch.rand(n, Cin, Hh, Ww, generator=g) * 2 - 1).to(dtype)
    w = ((torch.rand(Cout, Cin, k, k, generator=g) * 2 - 1) * 0.1).to(dtype)
    bias = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), bias.double(), stride, pad, dil)
    Ho, Wo = ref.shape[2:]
    res = torch.randn(n, Ho, Wo, Cout, generator=g) if pick([0, 1]) else None
    if res is not None:
        ref = ref + res.permute(0, 3, 1, 2).double()
    ref = ref.clamp(min=0)
    geom = dict(n_img=n, H=Hh, W=Ww, Cin=Cin, Ho=Ho, Wo=Wo, KH=k, KW=k, stride=stride, pad=pad, dil=dil)
    out = H.gemm_nt(x.permute(0, 2, 3, 1).contiguous().to(gpu), w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(gpu),
                    conv=geom, bias=bias.to(gpu), relu=True, out_dtype=torch.float32,
                    residual=None if res is None else res.view(-1, Cout).to(gpu))
    out = out.view(n, Ho, Wo, Cout).permute(0, 3, 1, 2).cpu().double()
    # inputs are exact in both dtypes; fp32 accumulation over k*k*Cin products of magnitude <= 0.1
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=2e-6 * (k * k * Cin) ** 0.5)


@pytest.mark.parametrize("seed", range(8))
def test_refinement_losses_fuzz(gpu, seed):
    """Weighted cross-entropy and the class-specific weighted smooth-L1 box loss (fast_rcnn_open_vocabulary.py:812-905)
    on random row counts, vocabulary sizes, ignore labels (-1), zero weights and beta, forward and backward, against
    the oracle."""
    from oracle import wsovod_ref as R
    from wsovod_amd.layers import functions as Fn

    g = torch.Generator().manual_seed(7000 + seed)
    pick = lambda xs: xs[int(torch.randint(0, len(xs), (1,), generator=g))]
    M, K = pick([1, 2, 63, 64, 257, 1500]), pick([1, 20, 80, 1203])
    weighted, beta = bool(pick([0, 1])), pick([0.0, 0.5, 1.0])
    logits = torch.randn(M, K + 1, generator=g) * pick([1.0, 6.0, 20.0])
    gt = torch.randint(-1, K + 1, (M,), generator=g)
    w = torch.rand(M, generator=g)
    w[torch.rand(M, generator=g) < 0.1] = 0.0
    pb = torch.rand(M, 4, generator=g) * 100
    pb[:, 2:] += pb[:, :2] + 5
    gb = torch.rand(M, 4, generator=g) * 100
    gb[:, 2:] += gb[:, :2] + 5
    pred = torch.randn(M, 4, generator=g)
    lc, pc = logits.clone().requires_grad_(True), pred.clone().requires_grad_(True)
    ref_c, ref_b = R.refinement_losses(lc, pc, gt, w, pb, gb, K, beta=beta, cross_entropy_weighted=weighted,
                                       box_loss_type="smooth_l1_weighted" if weighted else "smooth_l1")
    (ref_c + ref_b).backward()
    lg, pg = logits.clone().to(gpu).requires_grad_(True), pred.clone().to(gpu).requires_grad_(True)
    wk = w.clone()
    wk[gt == -1] = 0
    loss_c = Fn.weighted_cross_entropy(lg, gt.to(gpu), w.to(gpu), weighted)
    loss_b = Fn.weighted_l1_box_loss(pg, pb.to(gpu), gb.to(gpu), gt.to(gpu), wk.to(gpu), K, (10., 10., 5., 5.), beta,
                                     weighted=weighted)
    (loss_c + loss_b).backward()
    torch.testing.assert_close(loss_c.detach().cpu(), ref_c.detach(), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(loss_b.detach().cpu(), ref_b.detach(), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(lg.grad.cpu(), lc.grad, rtol=1e-3, atol=1e-7)
    torch.testing.assert_close(pg.grad.cpu(), pc.grad, rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("seed", range(8))
def test_pgt_mining_and_labelling_fuzz_exact(gpu, seed):
    """Pseudo ground-truth mining + IoU labelling (roi_heads.py:1480-1610, get_pgt_top_k) on random image counts, ragged
    proposal counts (1 .. 700), vocabulary sizes, duplicate boxes, tiny boxes (area <= 20) and tied scores: every index,
    class, box and weight bit-exact against the oracle."""
    from oracle import wsovod_ref as R
    from wsovod_amd.layers import hip_ops as H

    g = torch.Generator().manual_seed(8000 + seed)
    K = [2, 20, 80, 300][seed % 4]
    nums = [int(n) for n in torch.randint(1, 200, (int(torch.randint(1, 6, (1,), generator=g)),), generator=g)]
    if seed % 3 == 0:
        nums.append(700)
    boxes_list, scores_list, gts = [], [], []
    for n in nums:
        x0, y0 = torch.rand(n, generator=g) * 300, torch.rand(n, generator=g) * 200
        b = torch.stack([x0, y0, x0 + 2 + torch.rand(n, generator=g) ** 2 * 200, y0 + 2 + torch.rand(n, generator=g) ** 2 * 150], 1)
        if n > 3:
            b[2] = b[1]  # duplicate boxes: IoU ties
        sc = torch.rand(n, K, generator=g) * 0.05
        if n > 5:
            sc[4] = sc[3]  # tied scores: the first one wins
        boxes_list.append(b)
        scores_list.append(sc)
        gts.append(torch.unique(torch.randint(0, K, (int(torch.randint(1, 4, (1,), generator=g)),), generator=g)))
    img_logits = torch.rand(len(nums), K, generator=g).clamp(1e-6, 1 - 1e-6)
    targets = R.get_pgt_top_k(boxes_list, scores_list, gts, img_logits, K)
    lab = R.label_and_sample_proposals_wsl(boxes_list, targets, K)
    seg = torch.tensor([0] + list(torch.tensor(nums).cumsum(0)), dtype=torch.int32, device=gpu)
    goff = torch.tensor([0] + list(torch.tensor([len(x) for x in gts]).cumsum(0)), dtype=torch.int32, device=gpu)
    o = H.pgt_mine_and_label(torch.cat(scores_list).to(gpu), torch.cat(boxes_list).to(gpu), seg, torch.cat(gts).to(gpu),
                             goff, img_logits.to(gpu), K, 0.5)
    assert o["pgt_count"].cpu().tolist() == [len(t["gt_classes"]) for t in targets]
    for key, ref_key in (("gt_classes", "gt_classes"), ("gt_boxes", "gt_boxes"), ("gt_weights", "gt_weights"),
                         ("gt_scores", "gt_scores")):
        assert torch.equal(o[key].cpu(), torch.cat([l[ref_key] for l in lab])), key
    assert torch.equal(o["matched"].cpu().long(), torch.cat([l["matched_idxs"] for l in lab]))
