"""-m gpu: BASELINE.json's full sizes (800x600 images, 512 proposals, 20 classes, fc1 = 25088 -> 4096).

(i) Against the ORACLE at the configs' own sizes: the CPU restatement (oracle/wsovod_ref.py, pinned to the
    reference's golden vectors) runs a whole fp32 training step of an 800x600 x 512-proposal image in ~1.5 s, so the
    headline config (2 images), WSR_50 x 1024 proposals x K = 80 (1 image) and the ROIAlignV2 pooler are compared with
    it directly: shapes that only exist at full size (M = 1024..16384 rows in fc1, 75x100 maps, 7500-pixel GAP,
    512-row MIL segments, tile tails, split-K) are pinned to the reference this way.
(ii) Size-independent properties (checksums of checksums, shift equivariance, idempotence, order independence) at
    batch sizes beyond what the oracle finishes in seconds."""
import pytest
import torch

from oracle import roi_ops
from tests.conftest import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


def test_fc1_sized_contractions_checksum(gpu):
    """Checksum of checksums at the fc1 shapes: (A B^T) 1 = A (B^T 1) and 1^T (P^T Q) = (P 1)^T Q, evaluated in fp64
    from O(size) vectors, for the 8-phase NT tile (forward) and the transposed-read TN kernel (weight gradient)."""
    from wsovod_amd.layers import hip_ops as H

    g = torch.Generator(device="cuda").manual_seed(0)
    M, N, K = 8192, 4096, 25088
    A = (torch.rand(M, K, device=gpu, generator=g) - 0.5).to(torch.bfloat16)
    B = (torch.rand(N, K, device=gpu, generator=g) - 0.5).to(torch.bfloat16)
    C = H.gemm_nt(A, B, out_dtype=torch.float32)
    rows = C.double().sum(dim=1)
    ref = A.double() @ B.double().sum(dim=0)
    scale = (A.double().abs() @ B.double().abs().sum(dim=0))  # sum of |terms|: the natural error scale
    assert float(((rows - ref).abs() / scale).max()) < 2e-6
    dA = (torch.rand(M, N, device=gpu, generator=g) - 0.5).to(torch.bfloat16)
    dW = H.gemm_tn(dA, A)  # (N, K) = dA^T A
    cols = dW.double().sum(dim=0)
    ref = dA.double().sum(dim=1) @ A.double()
    scale = dA.double().abs().sum(dim=1) @ A.double().abs()
    assert float(((cols - ref).abs() / scale).max()) < 2e-6
    # linearity in the first operand (exact products, fp32 accumulation)
    A2 = (torch.rand(256, K, device=gpu, generator=g) - 0.5).to(torch.bfloat16)
    lhs = H.gemm_nt((A[:256].float() + A2.float()).to(torch.bfloat16), B, out_dtype=torch.float32)
    both = H.gemm_nt(A[:256], B, out_dtype=torch.float32) + H.gemm_nt(A2, B, out_dtype=torch.float32)
    inexact = (A[:256].float() + A2.float()).to(torch.bfloat16).float() - (A[:256].float() + A2.float())
    slack = float(inexact.abs().max()) * K * 0.5 + 1e-2  # the bf16 rounding of A1 + A2 itself
    assert float((lhs - both).abs().max()) <= slack


def test_roi_pool_full_size_properties(gpu):
    """512-channel 75x100 map, 512 boxes per image, 16 images: (i) a sample of boxes against the C oracle bit for bit,
    (ii) shift equivariance pool(f + c) = pool(f) + c where every bin is non-empty, (iii) idempotence on a constant map,
    (iv) every pooled value is a value of the map (max, not a blend)."""
    from tests.util import random_rois
    from wsovod_amd.layers import hip_ops as H

    n, Cc, Hh, Ww = 16, 512, 75, 100
    g = torch.Generator().manual_seed(1)
    feat = torch.randn(n, Hh, Ww, Cc, generator=g).to(torch.bfloat16)  # NHWC storage
    rois = random_rois(n * 512, n, 600, 800, seed=2)
    f_dev = feat.to(gpu).permute(0, 3, 1, 2)  # NCHW view of channels_last memory
    out = H.roi_pool_forward(f_dev, rois.to(gpu), 0.125, (7, 7), need_argmax=False)[0]
    assert out.shape == (n * 512, Cc, 7, 7)
    pick = torch.arange(0, n * 512, 257)
    ref = roi_ops.roi_pool_forward(feat[:, :, :, :64].permute(0, 3, 1, 2).float().contiguous(), rois[pick], 0.125, (7, 7))[0]
    assert torch.equal(out[pick][:, :64].float().cpu(), ref)
    shifted = H.roi_pool_forward((f_dev.float() + 2.0).to(torch.bfloat16), rois.to(gpu), 0.125, (7, 7),
                                 need_argmax=False)[0]
    ref_shift = roi_ops.roi_pool_forward((feat[:, :, :, :64].float() + 2.0).to(torch.bfloat16).permute(0, 3, 1, 2).float()
                                         .contiguous(), rois[pick], 0.125, (7, 7))[0]
    assert torch.equal(shifted[pick][:, :64].float().cpu(), ref_shift)
    const = torch.full((2, Hh, Ww, Cc), 3.0, dtype=torch.bfloat16, device=gpu).permute(0, 3, 1, 2)
    oc = H.roi_pool_forward(const, rois[:1024].to(gpu) * torch.tensor([0, 1, 1, 1, 1.0], device=gpu), 0.125, (7, 7),
                            need_argmax=False)[0]
    assert bool(((oc == 3.0) | (oc == 0.0)).all())
    vals = torch.unique(feat[:, :, :, 5].float())
    got = torch.unique(out[:, 5].float().cpu())
    assert bool(torch.isin(got[got != 0], vals).all())


def test_full_size_training_step_properties(gpu):
    """One bf16 training step at the benchmark size (4 images of 800x600, 512 proposals, K = 20)."""
    from wsovod_amd.data import make_batch
    from wsovod_amd.testing import build_hot_path_model

    cfg, model = build_hot_path_model(seed=0, precision="bf16", device="cuda:0")
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    host = make_batch(4, 512, 20, seed=77)
    batch = [{"image": x["image"].to(gpu), "proposals": x["proposals"].to(gpu), "instances": x["instances"],
              "height": x["height"], "width": x["width"]} for x in host]
    cap = {}
    rh = model.roi_heads
    om, rf = rh.object_miner.forward, rh.box_refinery[0].forward
    rh.object_miner.forward = lambda *a, **k: cap.setdefault("miner", om(*a, **k))
    rh.box_refinery[0].forward = lambda *a, **k: cap.setdefault("refine", rf(*a, **k))
    losses = model(batch)
    rh.object_miner.forward, rh.box_refinery[0].forward = om, rf
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    scores = cap["miner"][0].float()
    assert scores.shape == (4 * 512, 20) and bool((scores >= 0).all())
    per_img = scores.view(4, 512, 20).sum(dim=1)
    assert bool((per_img <= 1.0 + 1e-4).all()) and bool((per_img > 0).all())  # sum_r softmax_r(D) * softmax_k(C) <= 1
    img = rh.pred_class_img_logits
    assert bool((img >= 1e-6).all()) and bool((img <= 1 - 1e-6).all())
    logits = cap["refine"][0].float()
    assert logits.shape == (4 * 512, 21) and bool((logits[:, -1] == 0).all())
    assert float(logits.detach().abs().max()) <= 50.0 * 1.001  # temperature * cosine
    pgt = rh._last_pgt
    gt_cls = [set(torch.unique(x["instances"].gt_classes).tolist()) for x in host]
    labels = pgt["gt_classes"].view(4, 512).cpu()
    for i in range(4):
        assert set(labels[i].tolist()) <= gt_cls[i] | {20}
        assert (labels[i] != 20).any()  # the mined box labels itself (IoU 1)
    assert bool((pgt["gt_weights"] >= 0).all()) and bool((pgt["gt_weights"] <= 1).all())
    for k, v in losses.items():
        assert torch.isfinite(v) and float(v) >= 0, k
    for k, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k
    # image order does not matter: the same losses for the batch in reverse (per-image MIL softmax, per-image mining)
    model.zero_grad(set_to_none=True)
    losses_r = model(batch[::-1])
    for k in losses:
        torch.testing.assert_close(losses_r[k], losses[k], rtol=2e-3, atol=1e-5)


# ---------------------------------------------------------------------------------------------------------------
# the configs' own sizes against the oracle (reference: roi_heads.py:696-907, fast_rcnn_open_vocabulary.py:318-367,726-820)
# ---------------------------------------------------------------------------------------------------------------
_TRAJECTORY = {}
_ORACLE_CACHE = {}  # (n_images, proposals, classes, depth, pooler, seed) -> oracle step: shared by the precisions of one size


def _oracle_vs_hip(gpu, precision, *, n_images, proposals, classes, depth=18, embed_dim=512, pooler="ROIPool", seed=4321,
                   keep_grads_below=2_000_000):
    from oracle import compare as OC
    from wsovod_amd.data import make_batch
    from wsovod_amd.testing import build_hot_path_model, capture_full_step

    host = make_batch(n_images, proposals, classes, seed=seed)
    cfg, model = build_hot_path_model(seed=0, depth=depth, K=classes, D=embed_dim, precision=precision, pooler=pooler,
                                      device="cuda:0")
    model.train()
    for m in model.modules():  # dropout RNG streams cannot match the reference's: off on both sides
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    train_keys = [k for k, p in model.named_parameters() if p.requires_grad]
    batch = [{"image": x["image"].to(gpu), "proposals": x["proposals"].to(gpu), "instances": x["instances"],
              "height": x["height"], "width": x["width"]} for x in host]
    got = capture_full_step(model, batch, keep_grads_below=keep_grads_below)
    del model
    torch.cuda.empty_cache()
    key = (n_images, proposals, classes, depth, embed_dim, pooler, seed)
    if key not in _ORACLE_CACHE:  # (the model is seeded: every precision of a size starts from the same weights)
        if len(_ORACLE_CACHE) >= 1:
            _ORACLE_CACHE.clear()
        _ORACLE_CACHE[key] = OC.oracle_step(sd, host, train_keys, depth=depth, num_classes=classes, pooler_type=pooler)
    want = _ORACLE_CACHE[key]
    rep = OC.compare(got, want)
    print(f"{precision} WSR_{depth} {n_images} x 800x600 x {proposals} proposals, K={classes}, {pooler} vs oracle:", rep)
    return rep


def _assert_parity(rep, grad_tol):
    assert rep["max_abs_logit_err"] < 1e-3, rep  # the north star's bound on the MIL-head logits
    assert rep["max_abs_score_err"] < 1e-3 and rep["max_abs_img_score_err"] < 1e-3, rep
    assert rep["max_abs_delta_err"] < 1e-3, rep
    assert rep["max_rel_loss_err"] < 1e-3, rep
    assert rep["labels_exact"] and rep["label_boxes_exact"] and rep["pgt_exact"], rep  # proposal indexing: bit-exact
    assert rep["max_rel_weight_err"] < 1e-3, rep
    assert rep["max_rel_gradnorm_err"] < grad_tol, rep


@pytest.mark.parametrize("precision,grad_tol", [("fp32", 2e-3), ("bf16x3", 5e-3), ("parity_train", 5e-3)])
def test_headline_config_matches_the_oracle_at_its_own_size(gpu, precision, grad_tol):
    """BASELINE config 2 (WSR_18, 800x600, 512 proposals, K = 20) on 2 images, fp32 and bf16x3, against the oracle's
    step on the same weights: mining scores / refinement logits / deltas < 1e-3, pseudo-GT indices and per-proposal labels
    exact, losses 1e-3, every gradient norm 2e-3 (fp32) / 5e-3 (bf16x3; parity_train = the parity forward with bf16x3's
    backward arithmetic: round 6)."""
    _assert_parity(_oracle_vs_hip(gpu, precision, n_images=2, proposals=512, classes=20), grad_tol)


def test_parity_mode_matches_the_oracle_at_full_size(gpu):
    """MODEL.HIP.PRECISION = "parity" (the fast tolerance-meeting mode: split forward, bf16 backward): forward
    quantities inside the north star's bound against the oracle at the headline size; gradients of the bf16 grade."""
    rep = _oracle_vs_hip(gpu, "parity", n_images=2, proposals=512, classes=20)
    _assert_parity_mode(rep)


@pytest.mark.parametrize("pooler", ["ROIPool", "ROIAlignV2"])
def test_parity_mx_mode_matches_the_oracle_at_full_size(gpu, pooler, monkeypatch):
    """MODEL.HIP.PRECISION = "parity_mx" (round 6): the parity forward with the res4 / res5 convs and fc1 / fc2 on the
    block-scaled f16mx kernels (fp16 hi*hi + e4m3 cross terms; csrc/gemm8mx.hip) -- the same bar as "parity": forward
    quantities inside the north star's bound against the oracle at the headline size, indices exact, gradients of the bf16
    grade (the backward is the parity mode's, on the plain bf16 copies the f16mx producers write).  (The mode hands layers with
    fewer than ~200 tiles to the bf16x2 kernels; the thresholds are lowered here so that two images take the f16mx ones.)"""
    from wsovod_amd.modeling.backbone import ResNet
    from wsovod_amd.modeling.roi_heads import WSOVODROIHeads

    monkeypatch.setattr(ResNet, "MX_MIN_TILES", 1)
    monkeypatch.setattr(WSOVODROIHeads, "MX_MIN_ROWS", 1)
    rep = _oracle_vs_hip(gpu, "parity_mx", n_images=2, proposals=512, classes=20, pooler=pooler)
    _assert_parity_mode(rep)


def test_parity_mx_wsr50_bottleneck_stages_match_the_oracle_at_full_size(gpu, monkeypatch):
    """BASELINE config 3 / 4's backbone under "parity_mx": the BottleneckBlocks of res4 / res5 (1x1 -> 3x3 dilated -> 1x1 + the
    fused projection shortcut, 256 - 2048 channels) and fc1 / fc2 at 1024 proposals on the f16mx kernels, against the oracle
    (thresholds lowered: one image takes the f16mx kernels)."""
    from wsovod_amd.modeling.backbone import ResNet
    from wsovod_amd.modeling.roi_heads import WSOVODROIHeads

    monkeypatch.setattr(ResNet, "MX_MIN_TILES", 1)
    monkeypatch.setattr(WSOVODROIHeads, "MX_MIN_ROWS", 1)
    rep = _oracle_vs_hip(gpu, "parity_mx", n_images=1, proposals=1024, classes=80, depth=50)
    _assert_parity_mode(rep, elem_tol=None)


def _assert_parity_mode(rep, elem_tol=3e-2):
    """The "parity" precision's bar: forward quantities inside the north star's bound, indices exact; the backward runs
    in plain bf16 on the hi halves, so its gradients carry bf16's grade -- every tensor's norm within 1 % of the oracle's
    (measured 2e-3), every element of the small tensors within 3 % of the tensor's largest element (measured 3e-3 at the
    headline size)."""
    assert rep["max_abs_logit_err"] < 1e-3 and rep["max_abs_score_err"] < 1e-3 and rep["max_abs_delta_err"] < 1e-3, rep
    assert rep["max_abs_img_score_err"] < 1e-3 and rep["max_rel_loss_err"] < 1e-3, rep
    assert rep["labels_exact"] and rep["label_boxes_exact"] and rep["pgt_exact"], rep
    assert rep["max_rel_weight_err"] < 1e-3, rep
    assert rep["max_rel_gradnorm_err"] < 1e-2, rep
    assert elem_tol is None or rep["max_rel_grad_elem_err"] < elem_tol, rep


@pytest.mark.parametrize("precision", ["parity", "parity_mx", "fp32"])
def test_benchmarked_batch_of_32_images_matches_the_oracle(gpu, precision):
    """The bench's own step -- 32 x 800x600 images x 512 proposals, one training step -- against the oracle on the same
    weights (~1 min of CPU for the oracle's step, shared by the two precisions): fc1 at M = 16384 rows (64 row tiles,
    split-K tails of the small layers' dW at this batch), 512-row MIL segments x 32, the backbone on 32 images, the
    transposed-read dW over 16384 proposals.  parity / parity_mx: the headline precisions (round 6: the f16mx kernels at their
    full tile counts -- 1876 conv tiles, 1024 FC tiles); fp32: the exact mode."""
    rep = _oracle_vs_hip(gpu, precision, n_images=32, proposals=512, classes=20)
    if precision == "fp32":
        _assert_parity(rep, 2e-3)
        assert rep["max_rel_grad_elem_err"] < 2e-3, rep
    else:
        _assert_parity_mode(rep)


def test_config5_wsr50_1024_proposals_1203_classes_matches_the_oracle_at_full_size(gpu):
    """BASELINE config 5 AT SIZE (reference: meta_arch/rcnn_wsovod_mixed_datasets.py:188-191,237-238): the mixed-dataset
    model on WSR_50 (fc1 100352 -> 4096), one 800x600 image with 1024 proposals from the LVIS-sized source -- its own
    object miner with K = 1203 columns, the (1203, 512) text embeddings handed to the refinement head per call, i.e. the
    region x text GEMM at its only non-trivial size (1024 x 512 x 1204) -- against the oracle, fp32 mode AND the
    headline precision."""
    from oracle import compare as OC
    from wsovod_amd.data import make_batch
    from wsovod_amd.testing import build_mixed_model, capture_full_step

    Ks, source_id = (20, 80, 1203), 2
    host = make_batch(1, 1024, Ks[source_id], seed=555)
    for x in host:
        x["dataset_id"] = source_id
    want = None
    for precision in ("fp32", "parity"):
        cfg, model = build_mixed_model(seed=0, names=("voc_2007_train", "coco_2017_train", "lvis_v1_train"), Ks=Ks, depth=50,
                                       precision=precision, device="cuda:0")
        assert type(model).__name__ == "GeneralizedRCNN_WSOVOD_MixedDatasets"
        model.train()
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.eval()
        model.roi_heads.select_source(source_id)
        prefix = f"roi_heads.object_miners.{source_id}."
        sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
        train_keys = [k for k, p in model.named_parameters()
                      if p.requires_grad and (not k.startswith("roi_heads.object_miners.") or k.startswith(prefix))]
        batch = [{"image": x["image"].to(gpu), "proposals": x["proposals"].to(gpu), "instances": x["instances"],
                  "height": x["height"], "width": x["width"], "dataset_id": source_id} for x in host]
        got = capture_full_step(model, batch, keep_grads_below=2_000_000)
        assert got["mining_scores"].shape == (1024, 1203) and got["refine_logits"].shape == (1024, 1204)
        # the other datasets' miners stay untouched
        assert all(k.startswith(prefix) for k in got["grad_norms"] if k.startswith("roi_heads.object_miners."))
        classifier = model.classifier_train[source_id].detach().float().cpu()
        del model
        torch.cuda.empty_cache()
        if want is None:
            want = OC.oracle_step(sd, host, train_keys, depth=50, num_classes=Ks[source_id], classifier=classifier,
                                  miner_prefix=prefix)
        rep = OC.compare(got, want)
        print(f"config 5 at size, {precision}:", rep)
        if precision == "fp32":
            _assert_parity(rep, 2e-3)
        else:
            # one image, a 1203-way mining softmax at random initialisation: `fc2.bias.grad` is a sum over 1024 rows that
            # cancels to ~1e-3 of its terms (the exact-fp32 mode itself is off by 2.7e-3 of the largest element there,
            # 1000x its usual error), so bf16-grade terms move single elements by up to ~20 % of the largest one while
            # every tensor's norm agrees to 1.3e-3: gated on the norms; the element-wise figure is printed above
            _assert_parity_mode(rep, elem_tol=None)


def test_config4_wsr50_1024_proposals_matches_the_oracle_at_full_size(gpu):
    """BASELINE config 4 shapes at full size: WSR_50 (C5 = 2048, fc1 = 100352 -> 4096), 1024 proposals, K = 80, one
    800x600 image, fp32 mode."""
    _assert_parity(_oracle_vs_hip(gpu, "fp32", n_images=1, proposals=1024, classes=80, depth=50), 2e-3)


def test_roi_align_v2_matches_the_oracle_at_full_size(gpu):
    """The north star's pooler (POOLER_TYPE: ROIAlignV2) at the headline size, fp32 mode."""
    _assert_parity(_oracle_vs_hip(gpu, "fp32", n_images=1, proposals=512, classes=20, pooler="ROIAlignV2"), 2e-3)


def test_roi_align_v2_in_the_headline_precision_matches_the_oracle_at_full_size(gpu):
    """The north star's pooler in the BENCHMARKED precision: ROIAlignV2 (the rows kernel writing bf16x2 + the plain bf16
    copy for dW, poolers.py:176-182) x `parity` on 2 x 800x600 x 512 proposals against the oracle's step -- the
    combination bench.py's ROIAlignV2 side line runs."""
    _assert_parity_mode(_oracle_vs_hip(gpu, "parity", n_images=2, proposals=512, classes=20, pooler="ROIAlignV2"))


@pytest.mark.parametrize("precision", ["fp32", "parity"])
def test_config3_coco_shapes_match_the_oracle_at_full_size(gpu, precision):
    """BASELINE config 3's shapes at size: WSR_18, 512 proposals, K = 80 classes, D = 768 (CLIP ViT-L/14) text
    embeddings -- the 81-column refinement softmax, the 80-column MIL head and the 768-wide projection -- 2 images,
    exact-fp32 mode and the headline precision, against the oracle."""
    rep = _oracle_vs_hip(gpu, precision, n_images=2, proposals=512, classes=80, embed_dim=768)
    if precision == "fp32":
        _assert_parity(rep, 2e-3)
    else:
        _assert_parity_mode(rep)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "parity", "parity_mx", "parity_train"])
def test_five_step_training_trajectory_stays_on_the_oracles(gpu, precision, monkeypatch):
    """Not one step but a TRAJECTORY (reference: engine/trainer.py:57-84 + the SGD of engine/defaults.py:274-318): five
    optimizer steps on five different batches of 2 x 800x600 x 512 proposals, dropout off, through HotPathTrainer +
    HipSGD in the given precision, against the oracle taking the same five SGD steps (momentum, weight decay) from the
    same initial weights.  `parity` runs its backward in plain bf16, so its updates differ from the oracle's in the last
    bits of every step: after five steps the refinement logits and mining scores must STILL be inside the north star's
    1e-3, the labels and pseudo-GT of every step identical, the trained weights within bf16-gradient grade."""
    from oracle import compare as OC
    from oracle import wsovod_ref as R
    from wsovod_amd.data import make_batch
    from wsovod_amd.engine import HotPathTrainer, build_optimizer
    from wsovod_amd.testing import build_hot_path_model, capture_full_step

    steps = 5
    # the learning rates of the reference's own first five iterations: BASE_LR 0.01 reached through a linear warm-up over
    # WARMUP_ITERS = 200 (configs/PascalVOC-Detection/WSOVOD_WSR_18_DC5_1x.yaml:17-25) from detectron2's default
    # WARMUP_FACTOR of 0.001: lr_i = 0.01 * (0.001 * (1 - i / 200) + i / 200) = 1.0e-5, 6.0e-5, 1.1e-4, 1.6e-4, 2.1e-4.
    # (At a constant 1e-3 on the random-init model even the exact-fp32 mode leaves the 1e-3 band after five steps:
    # measured 1.7e-3 -- its 1.6e-4 gradient-summation differences times logits that move by several units.)
    lrs = [0.01 * (0.001 * (1 - i / 200) + i / 200) for i in range(steps)]
    if precision == "parity_mx":  # (two images per step: thresholds lowered so that the f16mx kernels run)
        from wsovod_amd.modeling.backbone import ResNet
        from wsovod_amd.modeling.roi_heads import WSOVODROIHeads

        monkeypatch.setattr(ResNet, "MX_MIN_TILES", 1)
        monkeypatch.setattr(WSOVODROIHeads, "MX_MIN_ROWS", 1)
    cfg, model = build_hot_path_model(seed=0, precision=precision, device="cuda:0")
    cfg.SOLVER.BASE_LR = lrs[0]
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    train_keys = [k for k, p in model.named_parameters() if p.requires_grad]
    hosts = [make_batch(2, 512, 20, seed=900 + s) for s in range(steps + 1)]
    dev = lambda host: [{"image": x["image"].to(gpu), "proposals": x["proposals"].to(gpu), "instances": x["instances"],
                         "height": x["height"], "width": x["width"]} for x in host]
    tr = HotPathTrainer(model, build_optimizer(cfg, model))
    tr.graph_max_batch = 0  # eager launches: `_last_pgt` of every step is read below
    got_labels, got_losses = [], []
    for s in range(steps):
        for grp in tr.optimizer.param_groups:  # the scheduler's job (every group of the hot path has multiplier 1)
            grp["lr"] = lrs[s]
        out = tr.run_step(dev(hosts[s]))
        pgt = model.roi_heads._last_pgt
        got_labels.append((pgt["gt_classes"].cpu().clone(), pgt["pgt_boxes"].cpu().clone(), pgt["pgt_classes"].cpu().clone()))
        got_losses.append({k: float(v) for k, v in out.items()})
    tr.flush()
    probe = capture_full_step(model, dev(hosts[steps]))  # the forward quantities AFTER five updates, on a sixth batch
    trained = {k: v.detach().float().cpu().clone() for k, v in model.named_parameters() if v.requires_grad}
    tr.close()
    del model
    torch.cuda.empty_cache()
    # the oracle's trajectory (computed once: both precisions start from the same seeded weights and batches)
    if "traj" not in _TRAJECTORY:
        params = {k: v.clone() for k, v in sd.items()}
        bufs, per_step = {}, []
        mom, wd = float(cfg.SOLVER.MOMENTUM), float(cfg.SOLVER.WEIGHT_DECAY)
        for s in range(steps):
            w = OC.oracle_step(params, hosts[s], train_keys)
            per_step.append({k: w[k] for k in ("gt_classes", "pgt_boxes", "pgt_classes", "losses")})
            tp = {k: params[k] for k in train_keys}
            R.sgd_step(tp, {k: w["grads"][k] for k in train_keys}, bufs, lrs[s], mom, wd)
            params.update(tp)
        _TRAJECTORY["traj"] = (per_step, params, OC.oracle_step(params, hosts[steps], train_keys))
    per_step, params, want = _TRAJECTORY["traj"]
    for s in range(steps):
        w = per_step[s]
        assert torch.equal(got_labels[s][0], w["gt_classes"]), s          # per-proposal labels of step s
        assert torch.equal(got_labels[s][1], w["pgt_boxes"]) and torch.equal(got_labels[s][2], w["pgt_classes"]), s
        for k, v in w["losses"].items():
            assert abs(got_losses[s][k] - v) <= 1e-3 * max(abs(v), 1e-3), (s, k, got_losses[s][k], v)
    rep = OC.compare(probe, want)
    print(f"{precision}: after {steps} optimizer steps vs the oracle's trajectory:", rep)
    # MEASURED (round 5): logits after five steps 1.1e-4 (fp32) and 6.7e-3 (parity).  `parity` keeps every step's labels
    # and pseudo-GT exact and its losses within 1e-3, but its plain-bf16 backward puts ~2e-3 of gradient error into every
    # update, and five updates move the logits out of the single-step 1e-3 band: the north star bounds the forward pass on
    # identical weights (met, every step: the tests above), not the trained trajectory -- the gate below says so honestly.
    # `parity_train` (round 6) = the parity forward + a backward that keeps the hi/lo split (layers/functions.py:
    # backward_split): it is held to the 1e-3 band after the five updates, like fp32 and bf16x3.
    logit_gate = 2e-2 if precision in ("parity", "parity_mx") else 1e-3  # (parity_mx: the parity mode's plain bf16 backward)
    assert rep["max_abs_logit_err"] < logit_gate and rep["max_abs_score_err"] < 1e-3 and rep["max_abs_delta_err"] < 1e-3, rep
    assert rep["labels_exact"] and rep["label_boxes_exact"] and rep["pgt_exact"] and rep["max_rel_loss_err"] < 2e-3, rep
    tol = 2e-3 if precision == "fp32" else 1e-2
    for k in train_keys:
        step_taken = (params[k] - sd[k]).abs().max()  # how far the oracle moved this tensor in five steps
        err = (trained[k] - params[k]).abs().max()
        assert float(err) <= tol * float(step_taken) + 1e-7 * float(params[k].abs().max()) + 1e-9, (k, float(err), float(step_taken))


def test_bf16_mode_deviation_from_the_oracle_at_full_size(gpu):
    """The timed precision (bf16) against the oracle at the headline size: what it misses the bound by, kept honest."""
    rep = _oracle_vs_hip(gpu, "bf16", n_images=2, proposals=512, classes=20)
    assert rep["max_abs_logit_err"] < 0.15 and rep["max_abs_score_err"] < 2e-3 and rep["max_rel_loss_err"] < 5e-2, rep
    assert rep["max_abs_logit_err"] > 1e-3  # plain bf16 does NOT meet the north star's bound: keep saying so


def test_wsr50_1024_proposals_full_size_step_properties(gpu):
    """Configs 3-5 at size in a test (not only a bench side line): WSR_50 x 1024 proposals x K = 80, 4 images, bf16:
    finite losses / gradients, MIL score sums <= 1, bg logit == 0, labels inside the image-level classes."""
    from wsovod_amd.data import make_batch
    from wsovod_amd.testing import build_hot_path_model, capture_full_step

    cfg, model = build_hot_path_model(seed=0, depth=50, K=80, D=512, precision="bf16", device="cuda:0")
    model.train()
    host = make_batch(4, 1024, 80, seed=99)
    batch = [{"image": x["image"].to(gpu), "proposals": x["proposals"].to(gpu), "instances": x["instances"],
              "height": x["height"], "width": x["width"]} for x in host]
    got = capture_full_step(model, batch)
    sc = got["mining_scores"].view(4, 1024, 80)
    assert bool((sc >= 0).all()) and bool((sc.sum(dim=1) <= 1 + 1e-4).all())
    assert got["refine_logits"].shape == (4096, 81) and bool((got["refine_logits"][:, -1] == 0).all())
    assert float(got["refine_logits"].abs().max()) <= 50.0 * 1.001
    labels = got["gt_classes"].view(4, 1024)
    for i, x in enumerate(host):
        assert set(labels[i].tolist()) <= set(x["instances"].gt_classes.tolist()) | {80}
    for k, v in got["losses"].items():
        assert v == v and 0 <= v < float("inf"), k
    assert all(v == v and v < float("inf") for v in got["grad_norms"].values())
