"""Helpers shared by bench.py, smoke() and the tests to build the hot-path model on synthetic inputs."""
import os
import pickle
import tempfile

import torch

from .config import get_cfg
from .data import make_class_embeddings

_CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")


def hot_path_cfg(depth=18, K=20, D=512, precision="bf16", pooler="ROIPool", device="cuda", weight_path=None,
                 emb_seed=7, rpn=False, freeze_at=None):
    """WSOVOD_WSR_{18,50}_DC5_1x in proposals-only mode: the keys below are the values of
    /root/reference/configs/PascalVOC-Detection/{Base-RCNN-DilatedC5,WSOVOD_WSR_18_DC5_1x}.yaml
    that the hot path reads, with PROPOSAL_GENERATOR=PrecomputedProposals, BBOX_REFINE off."""
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(_CONFIG_DIR, f"hot_path_wsr{depth}.yaml"))
    if rpn:  # the shipped form: RPN boxes next to the loaded proposals
        assert depth == 18
        cfg.merge_from_file(os.path.join(_CONFIG_DIR, "hot_path_wsr18_rpn.yaml"))
    if weight_path is None:
        weight_path = os.path.join(tempfile.mkdtemp(prefix="wsovod_emb_"), f"emb_{K}x{D}.pkl")
        with open(weight_path, "wb") as f:
            pickle.dump(make_class_embeddings(K, D, seed=emb_seed), f)
    cfg.merge_from_list([
        "MODEL.ROI_HEADS.NUM_CLASSES", K,
        "MODEL.ROI_BOX_HEAD.OPEN_VOCABULARY.WEIGHT_DIM", D,
        "MODEL.ROI_BOX_HEAD.OPEN_VOCABULARY.WEIGHT_PATH_TRAIN", weight_path,
        "MODEL.ROI_BOX_HEAD.OPEN_VOCABULARY.WEIGHT_PATH_TEST", weight_path,
        "MODEL.ROI_BOX_HEAD.POOLER_TYPE", pooler,
        "MODEL.HIP.PRECISION", precision,
        "MODEL.DEVICE", device,
    ])
    if freeze_at is not None:  # (every shipped WSR config: 5 = the whole backbone frozen)
        cfg.MODEL.BACKBONE.FREEZE_AT = int(freeze_at)
    return cfg


def mixed_datasets_cfg(names=("voc_2007_train", "voc_2007_val", "coco_2017_train"), Ks=(20, 20, 80), D=512, **kw):
    """The mixed-dataset variant (BASELINE config 5): one text-embedding file per dataset,
    object miners shared per dataset family."""
    cfg = hot_path_cfg(K=max(Ks), D=D, **kw)
    cfg.merge_from_file(os.path.join(_CONFIG_DIR, "hot_path_wsr18_mixed.yaml"))
    if kw.get("depth", 18) != 18:  # the mixed file is based on the WSR_18 one: put the requested backbone back
        cfg.merge_from_list(["MODEL.RESNETS.DEPTH", kw["depth"], "MODEL.RESNETS.RES2_OUT_CHANNELS", 256])
    tmp = tempfile.mkdtemp(prefix="wsovod_emb_")
    paths = []
    for i, K in enumerate(Ks):
        paths.append(os.path.join(tmp, f"emb{i}_{K}x{D}.pkl"))
        with open(paths[-1], "wb") as f:
            pickle.dump(make_class_embeddings(K, D, seed=100 + K), f)
    cfg.merge_from_list([
        "DATASETS.MIXED_DATASETS.NAMES", list(names),
        "DATASETS.MIXED_DATASETS.NUM_CLASSES", list(Ks),
        "DATASETS.MIXED_DATASETS.WEIGHT_PATH_TRAINS", paths,
        "MODEL.ROI_BOX_HEAD.OPEN_VOCABULARY.WEIGHT_PATH_TEST", paths[-1],
    ])
    return cfg


def build_hot_path_model(seed=0, calibrate_synthetic=True, **kw):
    """Random-init model of the named architecture (reference initialisers).  With
    `calibrate_synthetic` the frozen stem's first FrozenBN scale is set to 1/64 so that the random
    backbone maps raw-scale pixels (PIXEL_STD = 1: inputs of +-128) to O(1) features, as a trained
    checkpoint would; it changes no arithmetic, only keeps synthetic training finite."""
    from .modeling import build_model

    cfg = hot_path_cfg(**kw)
    torch.manual_seed(seed)
    model = build_model(cfg)
    if calibrate_synthetic:
        with torch.no_grad():
            model.backbone.stem.conv1.norm.weight.fill_(1.0 / 64.0)
    return cfg, model


def build_mixed_model(seed=0, calibrate_synthetic=True, **kw):
    """The mixed-dataset model (BASELINE config 5: GeneralizedRCNN_WSOVOD_MixedDatasets, one object miner per dataset
    family, text embeddings handed to the refinement head per call) with the reference initialisers; see
    build_hot_path_model for `calibrate_synthetic`."""
    from .modeling import build_model

    cfg = mixed_datasets_cfg(**kw)
    torch.manual_seed(seed)
    model = build_model(cfg)
    if calibrate_synthetic:
        with torch.no_grad():
            model.backbone.stem.conv1.norm.weight.fill_(1.0 / 64.0)
    return cfg, model


def capture_full_step(model, batched_inputs, keep_grads_below=0):
    """One training forward + backward with everything a parity check reads, moved to the host: losses, mining
    scores, refinement logits / deltas, image-level scores, the mining kernel's labels and pseudo GT, and the L2 norm
    of every trainable tensor's gradient (plus, under "grads", the gradients themselves of the tensors with fewer than
    `keep_grads_below` elements).  Gradients are cleared afterwards."""
    captured = {}
    rh = model.roi_heads
    orig_m, orig_r = rh.object_miner.forward, rh.box_refinery[0].forward

    def cap(name, fn):
        def wrapped(*a, **k):
            o = fn(*a, **k)
            captured[name] = o
            return o
        return wrapped

    rh.object_miner.forward = cap("miner", orig_m)
    rh.box_refinery[0].forward = cap("refine", orig_r)
    try:
        losses = model(batched_inputs)
    finally:
        rh.object_miner.forward, rh.box_refinery[0].forward = orig_m, orig_r
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    pgt = rh._last_pgt
    counts = pgt["pgt_count"].cpu().tolist()
    out = {
        "losses": {k: float(v.detach()) for k, v in losses.items()},
        "mining_scores": captured["miner"][0].detach().float().cpu(),
        "refine_logits": captured["refine"][0].detach().float().cpu(),
        "refine_deltas": captured["refine"][1].detach().float().cpu(),
        "img_scores": rh.pred_class_img_logits.detach().float().cpu(),
        "gt_classes": pgt["gt_classes"].cpu(), "gt_boxes": pgt["gt_boxes"].cpu(), "gt_weights": pgt["gt_weights"].cpu(),
        "pgt_num": counts, "pgt_boxes": pgt["pgt_boxes"].cpu(), "pgt_classes": pgt["pgt_classes"].cpu(),
        # fp64 accumulation: an fp32 norm over fc1's 103 M elements is itself off by ~0.5 %
        "grad_norms": {k: float(p.grad.detach().double().norm()) for k, p in model.named_parameters()
                       if p.requires_grad and p.grad is not None},
        "grads": {k: p.grad.detach().float().cpu() for k, p in model.named_parameters()
                  if p.requires_grad and p.grad is not None and p.numel() < keep_grads_below},
    }
    model.zero_grad(set_to_none=True)
    return out


def capture_step(model, batched_inputs):
    """One training forward + backward; returns (loss dict, object-mining scores (R,K), refinement logits (R,K+1)) --
    the two tensors the north star's "MIL-head logits within 1e-3" bound is about."""
    captured = {}
    rh = model.roi_heads
    orig_m, orig_r = rh.object_miner.forward, rh.box_refinery[0].forward

    def cap(name, fn):
        def wrapped(*a, **k):
            o = fn(*a, **k)
            captured[name] = o
            return o
        return wrapped

    rh.object_miner.forward = cap("miner", orig_m)
    rh.box_refinery[0].forward = cap("refine", orig_r)
    try:
        losses = model(batched_inputs)
    finally:
        rh.object_miner.forward, rh.box_refinery[0].forward = orig_m, orig_r
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    model.zero_grad(set_to_none=True)
    return ({k: v.detach() for k, v in losses.items()}, captured["miner"][0].detach(),
            captured["refine"][0].detach())
