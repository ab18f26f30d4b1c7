"""CPU: the C-ABI library builds/loads and exports every symbol of include/wsovod_hip.h; host-side
mirror of the reference interface (config loading, registries, containers, matcher, box transform)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from wsovod_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "wsovod_hip.h")).read()
    declared = set(re.findall(r"\b(wsovod_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = _lib.lib()  # raises if the .so is missing or a SIGNATURES symbol is absent
    assert declared == set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(L, name), name
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", hdr, re.S)
        args = m.group(1).strip()
        n = 0 if args in ("void", "") else args.count(",") + 1
        assert n == len(_lib.SIGNATURES[name]), (name, n)
    assert L.wsovod_abi_version() == _lib.ABI_VERSION == 9


def test_error_convention_without_gpu():
    """Argument validation happens before any launch: bad arguments return a status + message."""
    import ctypes as C

    from wsovod_amd import _lib

    L = _lib.lib()
    d = _lib.GemmDesc()
    d.dtype_in, d.M, d.N, d.K = 7, 4, 4, 8
    assert L.wsovod_gemm_nt(C.byref(d), None) == 1
    assert b"dtype" in L.wsovod_last_error()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        from wsovod_amd.layers import hip_ops

        hip_ops.roi_pool_forward(torch.zeros(1, 4, 8, 8), torch.zeros(2, 5), 0.125, (7, 7))


def test_reference_yaml_loads_and_model_keys_match_reference():
    from tests.helpers import golden_shapes
    from wsovod_amd.testing import build_hot_path_model

    cfg, model = build_hot_path_model(device="cpu")
    assert cfg.MODEL.ROI_BOX_HEAD.POOLER_TYPE == "ROIPool" and cfg.MODEL.RESNETS.RES5_DILATION == 2
    ref = golden_shapes()  # state-dict keys/shapes of the REFERENCE model (from make_golden.py)
    mine = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert mine == ref
    n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    assert n_train == 124483313  # 124.5 M trainable, backbone fully frozen
    assert not any(p.requires_grad for p in model.backbone.parameters())
    assert model.backbone.output_shape()["res5"].stride == 8 and model.backbone.output_shape()["res5"].channels == 512


@pytest.mark.skipif(not os.path.exists("/root/reference/configs"), reason="reference checkout absent")
def test_reference_own_config_files_load():
    from wsovod_amd.config import get_cfg

    for rel in ("PascalVOC-Detection/WSOVOD_WSR_18_DC5_1x.yaml", "COCO-Detection/WSOVOD_WSR_50_DC5_1x.yaml"):
        cfg = get_cfg()
        cfg.merge_from_file(os.path.join("/root/reference/configs", rel))
        assert cfg.MODEL.META_ARCHITECTURE == "GeneralizedRCNN_WSOVOD"
        assert cfg.WSOVOD.INSTANCE_REFINEMENT.REFINE_REG == [True]
        assert cfg.MODEL.ROI_BOX_HEAD.OPEN_VOCABULARY.NORM_TEMP == 50.0
        assert cfg.SOLVER.STEPS in ((70000,), (140000,))


def test_out_of_scope_features_fail_loudly():
    from wsovod_amd.modeling import build_model
    from wsovod_amd.testing import hot_path_cfg

    cfg = hot_path_cfg(device="cpu", rpn=True)
    cfg.MODEL.MRRP.MRRP_ON = True
    with pytest.raises(NotImplementedError, match="MRRP"):
        build_model(cfg)
    cfg = hot_path_cfg(device="cpu")
    cfg.WSOVOD.BBOX_REFINE.ENABLE = True
    with pytest.raises(NotImplementedError, match="SAM"):
        build_model(cfg)
    cfg = hot_path_cfg(device="cpu")
    cfg.MODEL.BACKBONE.FREEZE_AT = 0  # a trainable STEM (fused with the uint8 normalisation, forward only) is refused;
    model = build_model(cfg)          # FREEZE_AT 1 - 4 train the residual stages (tests/test_gpu_freeze_at.py)
    with pytest.raises(NotImplementedError, match="FREEZE_AT = 0"):
        model.backbone(torch.zeros(1, 3, 32, 32))
    cfg.MODEL.BACKBONE.FREEZE_AT = 2
    model = build_model(cfg)
    assert model.backbone.has_trainable_stage and not any(p.requires_grad for p in model.backbone.res2.parameters())
    assert all(p.requires_grad for p in model.backbone.res3.parameters())


def test_rpn_host_pieces():
    """Anchor grid (detectron2 DefaultAnchorGenerator restated) against the oracle's, state-dict names of the RPN,
    and the low-quality-match rule of the anchor matcher."""
    from oracle import wsovod_ref as R
    from wsovod_amd.modeling import build_model
    from wsovod_amd.modeling.matcher import Matcher
    from wsovod_amd.testing import hot_path_cfg

    model = build_model(hot_path_cfg(device="cpu", rpn=True))
    pg = model.proposal_generator
    assert type(pg).__name__ == "WSOVODRPN_V2" and model.roi_heads.rpn_on
    keys = {k for k in model.state_dict() if k.startswith("proposal_generator.")}
    assert keys == {f"proposal_generator.rpn_head.{m}.{p}" for m in ("conv", "objectness_logits", "anchor_deltas")
                    for p in ("weight", "bias")}
    assert pg.rpn_head.objectness_logits.weight.shape == (18, 512, 1, 1)
    assert pg.rpn_head.anchor_deltas.weight.shape == (72, 512, 1, 1)
    grid = pg.anchor_generator([torch.zeros(1, 512, 5, 7)])[0].tensor
    assert torch.equal(grid, R.anchor_grid(5, 7))
    assert grid.shape == (5 * 7 * 18, 4)
    torch.testing.assert_close(grid[18:36] - grid[:18], torch.tensor([8.0, 0, 8.0, 0]).expand(18, 4))
    iou = torch.tensor([[0.1, 0.15, 0.7], [0.19, 0.0, 0.1]])
    m, lab = Matcher([0.2, 0.6], [0, -1, 1], allow_low_quality_matches=True)(iou)
    assert lab.tolist() == [1, 0, 1] and m.tolist() == [1, 0, 0]  # anchor 0 is GT 1's best match despite IoU 0.19


def test_structures_and_matcher_semantics():
    from wsovod_amd.modeling.box_regression import Box2BoxTransform
    from wsovod_amd.modeling.matcher import Matcher
    from wsovod_amd.structures import Boxes, ImageList, Instances, pairwise_iou

    a = Boxes(torch.tensor([[0.0, 0.0, 10.0, 10.0], [20.0, 20.0, 30.0, 30.0]]))
    b = Boxes(torch.tensor([[5.0, 5.0, 15.0, 15.0], [0.0, 0.0, 10.0, 10.0], [50.0, 50.0, 60.0, 60.0]]))
    iou = pairwise_iou(a, b)
    torch.testing.assert_close(iou, torch.tensor([[25.0 / 175.0, 1.0, 0.0], [0.0, 0.0, 0.0]]))
    m, l = Matcher([0.5], [0, 1])(iou)
    assert m.tolist() == [0, 0, 0] and l.tolist() == [0, 1, 0]
    m, l = Matcher([0.5], [0, 1])(torch.zeros(0, 3))
    assert m.tolist() == [0, 0, 0] and l.tolist() == [0, 0, 0]
    t = Box2BoxTransform((10.0, 10.0, 5.0, 5.0))
    d = t.get_deltas(a.tensor, b.tensor[:2])
    torch.testing.assert_close(t.apply_deltas(d, a.tensor), b.tensor[:2])
    il = ImageList.from_tensors([torch.ones(3, 4, 5), torch.ones(3, 6, 2)])
    assert il.tensor.shape == (2, 3, 6, 5) and il.tensor[0, :, 4:, :].sum() == 0 and il.image_sizes == [(4, 5), (6, 2)]
    i = Instances((6, 5), proposal_boxes=a, objectness_logits=torch.tensor([0.9, 0.1]))
    assert len(i[torch.tensor([1])]) == 1 and len(Instances.cat([i, i])) == 4


def test_synthetic_batch_format():
    from wsovod_amd.data import make_batch

    b = make_batch(2, 64, 20, seed=1)
    assert b[0]["image"].dtype == torch.uint8 and b[0]["image"].shape == (3, 600, 800)
    p = b[0]["proposals"]
    assert len(p) == 64 and torch.all(p.objectness_logits[:-1] >= p.objectness_logits[1:])
    assert torch.all(p.objectness_logits > 0) and torch.all(p.objectness_logits <= 1)
    bx = p.proposal_boxes.tensor
    assert torch.all(bx[:, 2] <= 800) and torch.all(bx[:, 3] <= 600) and torch.all(bx[:, 2] - bx[:, 0] >= 16)
    assert len(torch.unique(b[0]["instances"].gt_classes)) == 2


def test_multi_dataset_sampler_matches_reference_golden():
    """Index work: the per-rank draw order is bit-identical to the reference's sampler (fixture g11)."""
    import itertools
    import os

    import numpy as np
    from tests.golden import gen
    from wsovod_amd.data import MultiDatasetTrainingSampler

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g11_multi_dataset_sampler.npz"))
    dicts = gen.sampler_dataset_dicts()
    rf = MultiDatasetTrainingSampler.get_repeat_factors(dicts, 3, [1, 1.5, 2], [False] * 3, [False, False, True],
                                                        0.001, 1.0)
    assert np.array_equal(rf.numpy(), g["repeat_factors"])
    streams = []
    for rank in (0, 1):
        s = MultiDatasetTrainingSampler(rf, seed=42, rank=rank, world_size=2)
        streams.append(list(itertools.islice(iter(s), 400)))
        assert streams[-1] == g[f"stream_rank{rank}"].tolist()
    s = MultiDatasetTrainingSampler(rf, shuffle=False, seed=7, rank=0, world_size=1)
    assert list(itertools.islice(iter(s), 300)) == g["stream_noshuffle"].tolist()
    # dataset mix follows the ratios: equalised sizes x [1, 1.5, 2]
    ds = np.array([dicts[i]["dataset_id"] for i in streams[0] + streams[1]])
    share = np.bincount(ds, minlength=3) / len(ds)
    assert abs(share[1] / share[0] - 1.5) < 0.2 and abs(share[2] / share[0] - 2.0) < 0.25


def test_repeat_factor_and_batcher_semantics():
    from wsovod_amd.data import MultiDatasetAspectRatioGroupedDataset, repeat_factors_from_category_frequency

    dicts = [{"annotations": [{"category_id": 0}]}] * 99 + [{"annotations": [{"category_id": 0}, {"category_id": 1}]}]
    rf = repeat_factors_from_category_frequency(dicts, 0.04)
    assert rf[0] == 1.0 and abs(float(rf[-1]) - 2.0) < 1e-6  # f(1) = 0.01 -> sqrt(0.04 / 0.01)
    stream = [{"dataset_id": i % 2, "width": 4 + (i % 3), "height": 5, "i": i} for i in range(40)]
    batches = list(MultiDatasetAspectRatioGroupedDataset(stream, batch_size=[2, 3], num_datasets=2))
    assert batches
    seen = []
    for b in batches:
        ids = {d["dataset_id"] for d in b}
        assert len(ids) == 1 and len(b) == [2, 3][ids.pop()]  # one dataset per batch, its own batch size
        assert len({d["width"] > d["height"] for d in b}) == 1  # one orientation per batch
        seen += [d["i"] for d in b]
    assert len(seen) == len(set(seen))
    assert seen[:2] == [d["i"] for d in batches[0]]


def test_proposal_file_format_matches_reference_golden(tmp_path):
    """n4 input formats: D1-style proposal pickle -> records -> transformed / de-duplicated / top-k Instances, against
    the reference's own load_proposals_into_dataset / unique_boxes / transform_proposals (fixture g13).  Index work:
    exact."""
    import pickle

    import numpy as np
    from tests.golden import gen
    from wsovod_amd.data import proposals as P

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g13_proposal_formats.npz"))
    pk, recs = gen.proposal_pickle()
    path = tmp_path / "props.pkl"
    with open(path, "wb") as f:
        pickle.dump(pk, f)
    recs = P.load_proposals_into_dataset(recs, str(path))
    for i, rec in enumerate(recs):
        assert np.array_equal(rec["proposal_boxes"], g[f"rec{i}/boxes"])
        assert np.array_equal(rec["proposal_objectness_logits"], g[f"rec{i}/logits"])
        assert np.all(np.diff(rec["proposal_objectness_logits"]) <= 0)
        assert np.array_equal(P.unique_boxes(rec["proposal_boxes"]), g[f"rec{i}/unique"])
        h, w = rec["height"], rec["width"]
        tl = P.TransformList([P.ResizeTransform(h, w, h * 2, w * 2)] + ([P.HFlipTransform(w * 2)] if i % 2 else []))
        d = dict(rec)
        P.transform_proposals(d, (h * 2, w * 2), tl, proposal_topk=50, min_box_size=8)
        assert "proposal_boxes" not in d and len(d["proposals"]) <= 50
        assert np.array_equal(d["proposals"].proposal_boxes.tensor.numpy(), g[f"rec{i}/out_boxes"])
        assert np.array_equal(d["proposals"].objectness_logits.numpy(), g[f"rec{i}/out_logits"])
    # directory form: one pickle per image, resolved lazily
    recs2 = P.load_proposals_into_dataset([{"image_id": 7}], str(tmp_path))
    assert recs2[0]["proposal_file"].endswith("/7.pkl")


def test_embedding_and_backbone_pickles(tmp_path):
    import pickle

    import numpy as np
    from wsovod_amd.data import load_class_embeddings, load_d2_pickle_into, make_class_embeddings
    from wsovod_amd.modeling.meta_arch import build_backbone
    from wsovod_amd.testing import hot_path_cfg

    emb = make_class_embeddings(20, 512, seed=3)
    p = tmp_path / "emb.pkl"
    with open(p, "wb") as f:
        pickle.dump(emb, f)
    w = load_class_embeddings(str(p))
    assert w.dtype == torch.float32 and w.shape == (20, 512) and torch.equal(w, emb)
    # detectron2-format backbone pickle: names without the model prefix, numpy arrays, plus a key the model lacks
    bb = build_backbone(hot_path_cfg(device="cpu"))
    sd = bb.state_dict()
    rng = np.random.RandomState(0)
    ckpt = {k: rng.randn(*v.shape).astype(np.float32) for k, v in sd.items()}
    ckpt["fc1000.weight"] = np.zeros((10, 4), dtype=np.float32)
    q = tmp_path / "r18_d2.pkl"
    with open(q, "wb") as f:
        pickle.dump({"model": ckpt, "__author__": "test", "matching_heuristics": True}, f)
    loaded, unused = load_d2_pickle_into(bb, str(q))
    assert unused == ["fc1000.weight"] and set(loaded) == set(sd)
    for k, v in bb.state_dict().items():
        assert np.array_equal(v.numpy(), ckpt[k]), k


def test_cat_rows_returns_views_only_for_consecutive_blocks():
    """hip_ops.cat_rows: a view when the inputs are consecutive row blocks of one buffer, torch.cat otherwise."""
    from wsovod_amd.layers.hip_ops import cat_rows

    base = torch.arange(60, dtype=torch.float32).view(15, 4)
    parts = [base[0:4], base[4:4], base[4:9], base[9:15]]  # an empty block in the middle
    v = cat_rows(parts)
    assert v.data_ptr() == base.data_ptr() and torch.equal(v, base)
    one_d = cat_rows([base.view(-1)[0:10], base.view(-1)[10:60]])
    assert one_d.data_ptr() == base.data_ptr() and one_d.shape == (60,)
    # not consecutive / different buffers / strided / requiring grad: a real concatenation with the right values
    for bad in ([base[0:4], base[5:9]], [base[0:4], base[4:9].clone()], [base[0:4, :2], base[4:9, :2]]):
        c = cat_rows(bad)
        assert torch.equal(c, torch.cat(bad)) and c.data_ptr() != base.data_ptr()
    g = base.clone().requires_grad_(True)
    c = cat_rows([g[0:4], g[4:15]])
    assert c.requires_grad and torch.equal(c, g)
    assert cat_rows([base[2:7]]) is not None and cat_rows([base[2:7]]).data_ptr() == base[2:7].data_ptr()


def test_split_rows_never_adds_a_tile_round():
    """HotPathTrainer.split_rows: both launches together need exactly the tile rounds of the single launch."""
    from wsovod_amd.engine import HotPathTrainer

    def rounds(rows, cols, cus=256):
        return -(-(-(-rows // 256) * -(-cols // 256)) // cus)

    for rows, cols in ((4096, 25088), (4096, 100352), (4096, 4096), (1024, 4096), (300, 700), (8, 8)):
        ra = HotPathTrainer.split_rows(rows, cols)
        if ra == 0:
            continue
        assert 0 < ra < rows and ra % 256 == 0
        assert rounds(ra, cols) + rounds(rows - ra, cols) == rounds(rows, cols), (rows, cols, ra)
    assert HotPathTrainer.split_rows(4096, 25088) == 1280  # fc1 of WSR_18: 2 + 5 = 7 rounds
    assert HotPathTrainer.split_rows(8, 8) == 0


def test_step_graph_row_buckets_and_clip_config():
    """Host logic of round 4 that needs no GPU: the row bucket a step graph is keyed on (padding <= 1/8, at least 64 rows,
    monotone, idempotent) and SOLVER.CLIP_GRADIENTS -> the optimizer's clip setting (engine/defaults.py:292-323)."""
    from wsovod_amd.config import get_cfg
    from wsovod_amd.engine.trainer import HotPathTrainer, gradient_clipping

    prev = 0
    for rows in list(range(1, 700)) + [1000, 1024, 1025, 4000, 4096, 5024, 100000]:
        b = HotPathTrainer.row_bucket(rows)
        assert b >= rows and b >= 64 and b >= prev and HotPathTrainer.row_bucket(b) == b
        assert b - rows < max(64, rows // 8 + 1), (rows, b)
        prev = b
    assert [HotPathTrainer.row_bucket(r) for r in (512, 513, 1024, 4000)] == [512, 576, 1024, 4096]
    cfg = get_cfg()
    assert gradient_clipping(cfg) is None
    cfg.SOLVER.CLIP_GRADIENTS.ENABLED = True
    cfg.SOLVER.CLIP_GRADIENTS.CLIP_TYPE = "full_model"
    cfg.SOLVER.CLIP_GRADIENTS.CLIP_VALUE = 0.0
    assert gradient_clipping(cfg) is None  # the reference enables full-model clipping only for a positive value
    cfg.SOLVER.CLIP_GRADIENTS.CLIP_VALUE = 2.0
    assert gradient_clipping(cfg) == ("full_model", 2.0)
    cfg.SOLVER.CLIP_GRADIENTS.CLIP_TYPE = "value"
    assert gradient_clipping(cfg) == ("value", 2.0)
    cfg.SOLVER.CLIP_GRADIENTS.CLIP_TYPE = "norm"
    assert gradient_clipping(cfg) == ("norm", 2.0)
    cfg.SOLVER.CLIP_GRADIENTS.CLIP_TYPE = "bogus"
    with pytest.raises(NotImplementedError):
        gradient_clipping(cfg)
