"""Generate the golden vectors in tests/golden/*.npz by running the REFERENCE's own code.

Runs only in the authoring container (needs /root/reference).  detectron2 / fvcore / torchvision /
cv2 / clip are not installed, so the reference's hot-path files are loaded by path with small
stand-in modules registered in sys.modules (semantics: SURVEY.md Appendix A).  The stand-ins for
containers/config/matcher/box-transform are the repo's own compat layer (wsovod_amd.structures,
.config, .modeling.{matcher,box_regression,sampling}); torchvision.ops.RoIPool is the REFERENCE's
own C++ RoIPool compiled where it lies (oracle/_ref).  No reference source is copied: the
fixtures hold only inputs-by-seed and outputs.

    python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import roi_ops  # noqa: E402
from tests.golden import gen  # noqa: E402
from wsovod_amd import config as C  # noqa: E402
from wsovod_amd import structures as S  # noqa: E402
from wsovod_amd.modeling import box_regression, matcher, sampling  # noqa: E402


# --------------------------------------------------------------------------------------
# stand-in modules
# --------------------------------------------------------------------------------------
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave as a package
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


class FrozenBatchNorm2d(nn.Module):
    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def forward(self, x):
        return F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, training=False,
                            eps=self.eps)


def get_norm(norm, out_channels):
    if norm is None or (isinstance(norm, str) and len(norm) == 0):
        return None
    assert norm == "FrozenBN", norm
    return FrozenBatchNorm2d(out_channels)


class D2Conv2d(nn.Conv2d):
    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def forward(self, x):
        x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


class CNNBlockBase(nn.Module):
    def __init__(self, in_channels, out_channels, stride):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        return self


class Backbone(nn.Module):
    @property
    def size_divisibility(self):
        return 0


def c2_msra_fill(module):
    nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


def cat(tensors, dim=0):
    assert isinstance(tensors, (list, tuple))
    if len(tensors) == 1:
        return tensors[0]
    return torch.cat(tensors, dim)


def d2_cross_entropy(input, target, *, reduction="mean", **kwargs):
    if target.numel() == 0 and reduction == "mean":
        return input.sum() * 0.0
    return F.cross_entropy(input, target, reduction=reduction, **kwargs)


def smooth_l1_loss(input, target, beta, reduction="none"):
    if beta < 1e-5:
        loss = torch.abs(input - target)
    else:
        n = torch.abs(input - target)
        loss = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
    if reduction == "mean":
        return loss.mean() if loss.numel() > 0 else 0.0 * loss.sum()
    if reduction == "sum":
        return loss.sum()
    return loss


class RefRoIPool(nn.Module):
    """torchvision.ops.RoIPool stand-in = the reference's own compiled C++ RoIPool (oracle/_ref)."""

    def __init__(self, output_size, spatial_scale):
        super().__init__()
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        self.spatial_scale = spatial_scale

    def forward(self, input, rois):
        if input.requires_grad:  # a trainable backbone stage (G19): forward AND backward of the reference's compiled op
            return _RefRoIPoolFn.apply(input, rois, self.spatial_scale, self.output_size)
        return roi_ops.ref_roi_pool_forward(input, rois, self.spatial_scale, self.output_size)[0]


class _RefRoIPoolFn(torch.autograd.Function):
    """The reference's own autograd front (wsovod/layers/roi_loop_pool.py:9-35) over its compiled CPU op (oracle/_ref)."""

    @staticmethod
    def forward(ctx, input, rois, spatial_scale, output_size):
        out, arg = roi_ops.ref_roi_pool_forward(input, rois, spatial_scale, output_size)
        ctx.save_for_backward(rois, arg)
        ctx.cfg = (spatial_scale, tuple(input.shape))
        return out

    @staticmethod
    def backward(ctx, grad):
        rois, arg = ctx.saved_tensors
        return roi_ops.ref_roi_pool_backward(grad, rois, arg, ctx.cfg[0], ctx.cfg[1]), None, None, None


class OracleROIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio, aligned=True):
        super().__init__()
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        self.spatial_scale, self.sampling_ratio, self.aligned = spatial_scale, sampling_ratio, aligned

    def forward(self, input, rois):
        return roi_ops.roi_align_forward(input, rois, self.spatial_scale, self.output_size, self.sampling_ratio,
                                         self.aligned)


class _Storage:
    iter = 0

    def put_scalar(self, *a, **k):
        pass

    def put_image(self, *a, **k):
        pass


class _Unsupported:
    def __init__(self, *a, **k):
        raise RuntimeError("not available in the golden harness")


def install_shims():
    _mod("cv2")
    _mod("clip")
    _mod("fvcore")
    _mod("fvcore.nn", giou_loss=None, smooth_l1_loss=smooth_l1_loss)
    _mod("fvcore.nn.weight_init", c2_msra_fill=c2_msra_fill, c2_xavier_fill=None)
    _mod("torchvision")
    _mod("torchvision.ops", RoIPool=RefRoIPool)
    _mod("detectron2")
    _mod("detectron2.layers", CNNBlockBase=CNNBlockBase, Conv2d=D2Conv2d, DeformConv=_Unsupported,
         ModulatedDeformConv=_Unsupported, ShapeSpec=S.ShapeSpec, get_norm=get_norm, Linear=nn.Linear, cat=cat,
         nonzero_tuple=sampling.nonzero_tuple, cross_entropy=d2_cross_entropy, batched_nms=None, ciou_loss=None,
         diou_loss=None, ROIAlign=OracleROIAlign, ROIAlignRotated=_Unsupported)
    _mod("detectron2.config", configurable=C.configurable, CfgNode=C.CfgNode)
    _mod("detectron2.structures", Boxes=S.Boxes, Instances=S.Instances, ImageList=S.ImageList,
         pairwise_iou=S.pairwise_iou, PolygonMasks=_Unsupported)
    _mod("detectron2.utils")
    _mod("detectron2.utils.comm", get_world_size=lambda: 1)
    _mod("detectron2.utils.events", get_event_storage=lambda: _Storage())
    _mod("detectron2.utils.visualizer", Visualizer=_Unsupported)
    _mod("detectron2.utils.logger", log_first_n=lambda *a, **k: None)
    _mod("detectron2.data", MetadataCatalog=types.SimpleNamespace(get=lambda name: None), DatasetCatalog=None)
    _mod("detectron2.data.detection_utils", convert_image_to_rgb=None)
    _mod("detectron2.modeling")
    _mod("detectron2.modeling.backbone", Backbone=Backbone,
         build_backbone=lambda cfg: C.BACKBONE_REGISTRY_REF.get(cfg.MODEL.BACKBONE.NAME)(
             cfg, S.ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))))
    _mod("detectron2.modeling.backbone.backbone", Backbone=Backbone)
    C.BACKBONE_REGISTRY_REF = C.Registry("REF_BACKBONE")
    C.ROI_BOX_HEAD_REGISTRY_REF = C.Registry("REF_ROI_BOX_HEAD")
    C.ROI_HEADS_REGISTRY_REF = C.Registry("REF_ROI_HEADS")
    C.META_ARCH_REGISTRY_REF = C.Registry("REF_META_ARCH")
    _mod("detectron2.modeling.backbone.build", BACKBONE_REGISTRY=C.BACKBONE_REGISTRY_REF)
    _mod("detectron2.modeling.box_regression", Box2BoxTransform=box_regression.Box2BoxTransform)
    _mod("detectron2.modeling.matcher", Matcher=matcher.Matcher)
    _mod("detectron2.modeling.sampling", subsample_labels=sampling.subsample_labels)
    _mod("detectron2.modeling.meta_arch")
    _mod("detectron2.modeling.meta_arch.build", META_ARCH_REGISTRY=C.META_ARCH_REGISTRY_REF)
    _mod("detectron2.modeling.proposal_generator", build_proposal_generator=lambda cfg, shape: None)
    _mod("detectron2.modeling.proposal_generator.proposal_utils", add_ground_truth_to_proposals=None)
    _mod("detectron2.modeling.roi_heads", ROI_BOX_HEAD_REGISTRY=C.ROI_BOX_HEAD_REGISTRY_REF,
         ROI_HEADS_REGISTRY=C.ROI_HEADS_REGISTRY_REF)
    _mod("detectron2.modeling.roi_heads.box_head",
         build_box_head=lambda cfg, shape: C.ROI_BOX_HEAD_REGISTRY_REF.get(cfg.MODEL.ROI_BOX_HEAD.NAME)(cfg, shape))
    # the reference package skeleton (real files are loaded into it by path)
    for name in ("wsovod", "wsovod.modeling", "wsovod.modeling.backbone", "wsovod.modeling.roi_heads",
                 "wsovod.modeling.class_heads", "wsovod.modeling.meta_arch"):
        _mod(name)
    _mod("wsovod.layers", ROILoopPool=_Unsupported)
    _mod("wsovod.modeling.proposal_generator", WSOVODRPN_V2=type("WSOVODRPN_V2", (), {}))
    _mod("wsovod.modeling.postprocessing", detector_postprocess=None)


def load_ref(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    parent, _, child = modname.rpartition(".")
    setattr(sys.modules[parent], child, m)
    spec.loader.exec_module(m)
    return m


def load_reference():
    install_shims()
    r = types.SimpleNamespace()
    r.resnet = load_ref("wsovod.modeling.backbone.resnet_wsl", "wsovod/modeling/backbone/resnet_wsl.py")
    r.box_head = load_ref("wsovod.modeling.roi_heads.box_head", "wsovod/modeling/roi_heads/box_head.py")
    r.ovc = load_ref("wsovod.modeling.class_heads.open_vocabulary_classifier",
                     "wsovod/modeling/class_heads/open_vocabulary_classifier.py")
    sys.modules["wsovod.modeling.class_heads"].OpenVocabularyClassifier = r.ovc.OpenVocabularyClassifier
    r.daf = load_ref("wsovod.modeling.class_heads.data_aware_features_head",
                     "wsovod/modeling/class_heads/data_aware_features_head.py")
    r.frcnn = load_ref("wsovod.modeling.roi_heads.fast_rcnn_open_vocabulary",
                       "wsovod/modeling/roi_heads/fast_rcnn_open_vocabulary.py")
    r.poolers = load_ref("wsovod.modeling.poolers", "wsovod/modeling/poolers.py")
    r.roi_heads = load_ref("wsovod.modeling.roi_heads.roi_heads", "wsovod/modeling/roi_heads/roi_heads.py")
    r.meta = load_ref("wsovod.modeling.meta_arch.rcnn_wsovod", "wsovod/modeling/meta_arch/rcnn_wsovod.py")
    return r


# --------------------------------------------------------------------------------------
# fixtures
# --------------------------------------------------------------------------------------
def ref_cfg(depth, K, D, emb_path, pooler="ROIPool"):
    from wsovod_amd.testing import hot_path_cfg

    cfg = hot_path_cfg(depth=depth, K=K, D=D, pooler=pooler, device="cpu", weight_path=emb_path)
    cfg.MODEL.PIXEL_STD = list(gen.PIXEL_STD)
    cfg.MODEL.ROI_HEADS.NAME = "WSOVODROIHeads"
    cfg.DATASETS.TRAIN = ("synthetic",)
    return cfg


def to_inputs(batch):
    out = []
    for b in batch:
        h, w = b["image"].shape[-2:]
        props = S.Instances((h, w), proposal_boxes=S.Boxes(b["boxes"].clone()), objectness_logits=b["objectness"].clone())
        nb = len(b["gt_classes"])
        inst = S.Instances((h, w), gt_boxes=S.Boxes(b["boxes"][:nb].clone()), gt_classes=b["gt_classes"].clone())
        out.append({"image": b["image"], "instances": inst, "proposals": props, "height": h, "width": w})
    return out


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def build_ref_model(r, depth, K, D, seed, freeze_at=None):
    import pickle
    import tempfile

    shapes_probe = None
    emb = os.path.join(tempfile.mkdtemp(prefix="golden_"), "emb.pkl")
    with open(emb, "wb") as f:
        pickle.dump(torch.randn(K, D), f)  # placeholder; class_weight is overwritten from the seeded state
    cfg = ref_cfg(depth, K, D, emb)
    if freeze_at is not None:
        cfg.MODEL.BACKBONE.FREEZE_AT = freeze_at
    model = r.meta.GeneralizedRCNN_WSOVOD(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = gen.seeded_state(shapes, seed)
    missing, unexpected = model.load_state_dict(sd, strict=True), None
    return cfg, model, sd, shapes


def golden_mixed(r):
    """G10: the REFERENCE's mixed-dataset meta-arch + ROI heads (rcnn_wsovod_mixed_datasets.py, roi_heads.py:1860+)
    on two sources with different class counts (voc-like K=20 through the shared voc miner, coco-like K=80)."""
    from wsovod_amd.testing import mixed_datasets_cfg

    r.meta_mixed = load_ref("wsovod.modeling.meta_arch.rcnn_wsovod_mixed_datasets",
                            "wsovod/modeling/meta_arch/rcnn_wsovod_mixed_datasets.py")
    Ks = (20, 20, 80)
    cfg = mixed_datasets_cfg(Ks=Ks, device="cpu")
    cfg.MODEL.PIXEL_STD = list(gen.PIXEL_STD)
    cfg.DATASETS.TRAIN = ("synthetic",)
    torch.manual_seed(0)
    model = r.meta_mixed.GeneralizedRCNN_WSOVOD_MixedDatasets(cfg)
    assert type(model.roi_heads).__name__ == "WSOVODMixedDatasetsROIHeads"
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = gen.mixed_seeded_state(shapes, seed=17)
    model.load_state_dict(sd, strict=True)
    model.train()
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 1e-12
    save("shapes_mixed_r18", keys=np.array(list(shapes.keys())), shapes=np.array([str(v) for v in shapes.values()]))
    arrays = {}
    for source_id in (2, 0):
        K = Ks[source_id]
        batch = gen.seeded_batch(2, 40, K, 256, 352, seed=19 + source_id)
        inputs = to_inputs(batch)
        for d in inputs:
            d["dataset_id"] = source_id
        captured = {}
        rh = model.roi_heads
        miner = rh.object_miners[source_id]
        orig_miner, orig_ref, orig_label = miner.forward, rh.box_refinery[0].forward, rh.label_and_sample_proposals_wsl

        def cap(name, fn):
            def wrapped(*a, **k):
                o = fn(*a, **k)
                captured[name] = o
                return o
            return wrapped

        miner.forward = cap("miner", orig_miner)
        rh.box_refinery[0].forward = cap("refine", orig_ref)
        rh.label_and_sample_proposals_wsl = cap("proposals_k", orig_label)
        model.zero_grad(set_to_none=True)
        loss_dict = model(inputs)
        sum(loss_dict.values()).backward()
        miner.forward, rh.box_refinery[0].forward, rh.label_and_sample_proposals_wsl = orig_miner, orig_ref, orig_label
        p = f"s{source_id}/"
        for k, v in loss_dict.items():
            arrays[p + "loss/" + k] = v
        arrays[p + "mining_scores"] = captured["miner"][0]
        arrays[p + "refine_logits"] = captured["refine"][0]
        arrays[p + "refine_deltas"] = captured["refine"][1]
        arrays[p + "label/gt_classes"] = torch.cat([q.gt_classes for q in captured["proposals_k"]])
        arrays[p + "label/gt_boxes"] = torch.cat([q.gt_boxes.tensor for q in captured["proposals_k"]])
        arrays[p + "label/gt_weights"] = torch.cat([q.gt_weights for q in captured["proposals_k"]])
        for k, q in model.named_parameters():
            if q.requires_grad:
                # float64: a float32 sum of 1e8 squares loses ~1 % on the CPU
                arrays[p + "gradnorm/" + k] = q.grad.double().norm() if q.grad is not None else torch.tensor(-1.0)
    save("g10_mixed_datasets_step", **arrays)


class RefStandardRPNHead(nn.Module):
    """detectron2 StandardRPNHead stand-in (un-vendored) in plain torch."""

    def __init__(self, cfg, input_shape):
        super().__init__()
        from wsovod_amd.modeling.anchor_generator import build_anchor_generator

        in_channels = input_shape[0].channels
        A = build_anchor_generator(cfg, input_shape).num_anchors[0]
        self.conv = D2Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1, activation=nn.ReLU())
        self.objectness_logits = nn.Conv2d(in_channels, A, kernel_size=1, stride=1)
        self.anchor_deltas = nn.Conv2d(in_channels, A * 4, kernel_size=1, stride=1)
        for layer in (self.conv, self.objectness_logits, self.anchor_deltas):
            nn.init.normal_(layer.weight, std=0.01)
            nn.init.constant_(layer.bias, 0)

    def forward(self, features):
        lo, de = [], []
        for x in features:
            t = self.conv(x)
            lo.append(self.objectness_logits(t))
            de.append(self.anchor_deltas(t))
        return lo, de


def _install_rpn(r):
    """Stand-ins the reference's RPN files need (detectron2's anchor generator / matcher / box transform / RPN head are
    the restated ones; batched_nms is the INDEPENDENT brute-force NMS below, not the oracle's) and the reference's own
    proposal_utils.py / rpn.py loaded on top of them.  Returns a builder of the reference model with the RPN branch."""
    from wsovod_amd.modeling import anchor_generator as AG
    from wsovod_amd.testing import hot_path_cfg

    C.PROPOSAL_GENERATOR_REGISTRY_REF = C.Registry("REF_PROPOSAL_GENERATOR")
    C.RPN_HEAD_REGISTRY_REF = C.Registry("REF_RPN_HEAD")
    C.RPN_HEAD_REGISTRY_REF._obj_map["StandardRPNHead"] = RefStandardRPNHead
    L = sys.modules["detectron2.layers"]
    L.CycleBatchNormList = _Unsupported
    L.batched_nms = brute_force_batched_nms
    L.move_device_like = lambda src, dst: src.to(dst.device)
    _mod("detectron2.modeling.anchor_generator", build_anchor_generator=AG.build_anchor_generator,
         DefaultAnchorGenerator=AG.DefaultAnchorGenerator)
    B = sys.modules["detectron2.modeling.box_regression"]
    B.Box2BoxTransformLinear = _Unsupported
    B._dense_box_regression_loss = None
    _mod("detectron2.modeling.proposal_generator.build", PROPOSAL_GENERATOR_REGISTRY=C.PROPOSAL_GENERATOR_REGISTRY_REF)
    _mod("detectron2.modeling.proposal_generator.rpn", RPN_HEAD_REGISTRY=C.RPN_HEAD_REGISTRY_REF,
         build_rpn_head=lambda cfg, shape: C.RPN_HEAD_REGISTRY_REF.get(cfg.MODEL.RPN.HEAD_NAME)(cfg, shape))
    _mod("detectron2.modeling.poolers", convert_boxes_to_pooler_format=None)
    _mod("detectron2.utils.env", TORCH_VERSION=(2, 10))
    _mod("detectron2.utils.memory", retry_if_cuda_oom=lambda f: f)
    sys.modules["detectron2.structures"].pairwise_point_box_distance = None
    sys.modules["fvcore.nn"].giou_loss = None
    sys.modules["wsovod.layers"].csc = None
    pg = _mod("wsovod.modeling.proposal_generator")
    r.putils = load_ref("wsovod.modeling.proposal_generator.proposal_utils",
                        "wsovod/modeling/proposal_generator/proposal_utils.py")
    r.rpn = load_ref("wsovod.modeling.proposal_generator.rpn", "wsovod/modeling/proposal_generator/rpn.py")
    r.rpn.subsample_labels = gen.first_k_subsample
    pg.WSOVODRPN_V2, pg.WSOVODRPN = r.rpn.WSOVODRPN_V2, r.rpn.WSOVODRPN
    r.meta.WSOVODRPN_V2 = r.rpn.WSOVODRPN_V2
    r.meta.build_proposal_generator = lambda cfg, shape: C.PROPOSAL_GENERATOR_REGISTRY_REF.get(
        cfg.MODEL.PROPOSAL_GENERATOR.NAME)(cfg, shape)
    _Storage.iter = 1000

    def build(K=20, D=512):
        import pickle
        import tempfile
        emb = os.path.join(tempfile.mkdtemp(prefix="golden_"), "emb.pkl")
        with open(emb, "wb") as f:
            pickle.dump(torch.randn(K, D), f)
        cfg = hot_path_cfg(depth=18, K=K, D=D, device="cpu", weight_path=emb, rpn=True)
        cfg.MODEL.PIXEL_STD = list(gen.PIXEL_STD)
        cfg.MODEL.ROI_HEADS.NAME = "WSOVODROIHeads"
        cfg.DATASETS.TRAIN = ("synthetic",)
        cfg.SOLVER.MAX_ITER = 4000
        torch.manual_seed(0)
        model = r.meta.GeneralizedRCNN_WSOVOD(cfg)
        assert type(model.proposal_generator).__name__ == "WSOVODRPN_V2" and model.roi_heads.rpn_on
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        sd = gen.seeded_state(shapes, seed=41)
        model.load_state_dict(sd, strict=True)
        return cfg, model, shapes

    return build


def golden_rpn(r):
    """G12: the REFERENCE's WSOVODRPN_V2 + find_top_rpn_proposals + the meta-arch / ROI-heads RPN branches on one
    training step (RPN boxes + loaded boxes, pseudo-GT from the refinement head, RPN losses).  detectron2's
    anchor generator / matcher / box transform / RPN head are the restated stand-ins, batched_nms is the brute-force
    NMS of this file (independent of the oracle and of the HIP kernel); subsample_labels is replaced by the
    deterministic first-k rule of tests/golden/gen.py on both sides."""
    K = 20
    cfg, model, shapes = _install_rpn(r)()
    model.train()
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 1e-12
    save("shapes_rpn_r18", keys=np.array(list(shapes.keys())), shapes=np.array([str(v) for v in shapes.values()]))
    batch = gen.seeded_batch(2, 40, K, 256, 352, seed=43)
    inputs = to_inputs(batch)
    captured = {}
    pgm = model.proposal_generator
    orig = pgm.predict_proposals

    def cap_props(*a, **k):
        o = orig(*a, **k)
        captured["proposals"] = [(p.proposal_boxes.tensor.clone(), p.objectness_logits.clone()) for p in o]
        return o

    pgm.predict_proposals = cap_props
    orig_label = pgm.label_and_sample_anchors

    def cap_label(*a, **k):
        o = orig_label(*a, **k)
        captured["anchor_labels"] = torch.stack(o[0])
        return o

    pgm.label_and_sample_anchors = cap_label
    loss_dict = model(inputs)
    sum(loss_dict.values()).backward()
    arrays = {f"loss/{k}": v for k, v in loss_dict.items()}
    arrays["rpn_logits"] = pgm.pred_objectness_logits[0]
    arrays["rpn_deltas_sample"] = gen.strided_sample(pgm.pred_anchor_deltas[0], 8192)
    for i, (bx, sc) in enumerate(captured["proposals"]):
        arrays[f"prop{i}/boxes"], arrays[f"prop{i}/logits"] = bx, sc
    arrays["anchor_labels"] = captured["anchor_labels"]
    for i, t in enumerate(model.roi_heads.proposal_targets):
        arrays[f"target{i}/gt_boxes"] = t.gt_boxes.tensor
        arrays[f"target{i}/gt_classes"] = t.gt_classes
    for k, q in model.named_parameters():
        if q.requires_grad:
            arrays["gradnorm/" + k] = q.grad.double().norm() if q.grad is not None else torch.tensor(-1.0)
    save("g12_rpn_train_step", **arrays)


def brute_force_batched_nms(boxes, scores, idxs, iou_threshold):
    """Stand-in for detectron2.layers.batched_nms (-> torchvision.ops.batched_nms, un-vendored) used ONLY to generate
    g14/g15: a direct O(n^2) greedy per-class NMS in float32 numpy, written from the definition (visit boxes by
    descending score, ties by index; drop a box whose IoU with an already kept box OF THE SAME CLASS exceeds the
    threshold; return the kept indices in visiting order).  Independent of oracle/det_ops_ref.c and of the HIP kernel."""
    b = boxes.detach().cpu().numpy().astype(np.float32)
    sc = scores.detach().cpu().numpy()
    cl = idxs.detach().cpu().numpy()
    order = np.lexsort((np.arange(len(sc)), -sc.astype(np.float64)))
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    kept = []
    for i in order:
        if kept:
            k = np.asarray(kept)
            k = k[cl[k] == cl[i]]
            if len(k):
                iw = np.maximum(np.minimum(b[k, 2], b[i, 2]) - np.maximum(b[k, 0], b[i, 0]), np.float32(0))
                ih = np.maximum(np.minimum(b[k, 3], b[i, 3]) - np.maximum(b[k, 1], b[i, 1]), np.float32(0))
                inter = iw * ih
                if np.any(inter / (area[k] + area[i] - inter) > np.float32(iou_threshold)):
                    continue
        kept.append(int(i))
    return torch.as_tensor(kept, dtype=torch.int64)


def _eval_reference_model(r, K=20, D=512, seed=1):
    """The reference model in eval mode with the reference's own postprocessing.py and the brute-force NMS stand-in."""
    sys.modules["detectron2.structures"].ROIMasks = _Unsupported
    post = load_ref("wsovod.modeling.postprocessing", "wsovod/modeling/postprocessing.py")
    r.meta.detector_postprocess = post.detector_postprocess
    r.frcnn.batched_nms = brute_force_batched_nms
    cfg, model, sd, shapes = build_ref_model(r, 18, K, D, seed=seed)
    model.eval()
    return cfg, model, post


def golden_eval_tail(r):
    """G14: the REFERENCE's inference path -- GeneralizedRCNN_WSOVOD.inference (rcnn_wsovod.py:236-319) ->
    WSOVODROIHeads eval forward -> predict_probs_K / predict_boxes_K (fast_rcnn_open_vocabulary.py:987-1058) ->
    fast_rcnn_inference_single_image (:149-217) -> detector_postprocess (postprocessing.py:8-82) on three images whose
    requested output size differs from the network input.  Stored: the raw per-proposal scores / boxes the tail
    consumes, the detections before and after post-processing."""
    cfg, model, post = _eval_reference_model(r)
    batch = gen.seeded_batch(3, 200, 20, 256, 352, seed=15)
    # a few non-finite rows and exact score ties exercise the valid-mask and tie-order branches of the tail
    inputs = to_inputs(batch)
    for i, x in enumerate(inputs):
        x["height"], x["width"] = 300 + 40 * i, 500 - 30 * i
    clf = torch.randn(20, 512, generator=torch.Generator().manual_seed(77))
    arrays = {"classifier": clf}
    with torch.no_grad():
        raw, all_scores, all_boxes = model.inference(inputs, do_postprocess=False, classifier=clf)
        full = model.inference(inputs, do_postprocess=True, classifier=clf)
    for i, (res, sc, bx, out) in enumerate(zip(raw, all_scores, all_boxes, full)):
        arrays[f"img{i}/all_scores"], arrays[f"img{i}/all_boxes"] = sc[0], bx[0]
        arrays[f"img{i}/raw_boxes"], arrays[f"img{i}/raw_scores"] = res.pred_boxes.tensor, res.scores
        arrays[f"img{i}/raw_classes"], arrays[f"img{i}/raw_inds"] = res.pred_classes, res.pred_inds
        o = out["instances"]
        arrays[f"img{i}/out_size"] = np.array(o.image_size)
        arrays[f"img{i}/out_boxes"], arrays[f"img{i}/out_scores"] = o.pred_boxes.tensor, o.scores
        arrays[f"img{i}/out_classes"], arrays[f"img{i}/out_inds"] = o.pred_classes, o.pred_inds
    # the tail on hand-made inputs: non-finite rows, exact score ties, class-specific (R, K*4) boxes, top-k cut
    g = torch.Generator().manual_seed(78)
    R_, K_ = 150, 6
    xy = torch.rand(R_, 2, generator=g) * torch.tensor([300.0, 200.0])
    wh = torch.rand(R_, 2, generator=g) * 120 + 4
    base = torch.cat([xy, xy + wh], dim=1)
    boxes_cs = (base[:, None, :] + torch.randn(R_, K_, 4, generator=g) * 3).reshape(R_, K_ * 4)
    sc = torch.softmax(torch.randn(R_, K_ + 1, generator=g) * 2, dim=1)
    sc[10] = sc[11]  # exact ties between two proposals
    sc[20, 2] = float("nan")
    boxes_cs[30, 5] = float("inf")
    for name, bx, topk in (("agnostic", base, 40), ("specific", boxes_cs, -1)):
        res, kept, _, _ = r.frcnn.fast_rcnn_inference_single_image(bx, sc, (210, 330), 0.05, 0.3, topk)
        arrays[f"tail_{name}/boxes_in"], arrays[f"tail_{name}/scores_in"] = bx, sc
        arrays[f"tail_{name}/boxes"], arrays[f"tail_{name}/scores"] = res.pred_boxes.tensor, res.scores
        arrays[f"tail_{name}/classes"], arrays[f"tail_{name}/inds"] = res.pred_classes, res.pred_inds[:, 0] if res.pred_inds.dim() > 1 else res.pred_inds
        arrays[f"tail_{name}/kept_rows"] = kept
    save("g14_eval_tail", **arrays)


def _install_tta(r):
    """Stand-ins the reference's TTA files import: detectron2's ResizeShortestEdge / RandomFlip / apply_augmentations /
    fvcore transforms are un-vendored and wrap the restatements in wsovod_amd/data/proposals.py (image resampling = PIL
    bilinear); detectron2's own `fast_rcnn_inference_single_image` (un-vendored) is the reference's superset of it cut to
    its 2-tuple return."""
    from wsovod_amd.data import proposals as P
    from wsovod_amd.modeling import test_time_augmentation as T

    class TL(T._InvertibleList):
        def __add__(self, other):
            return TL(list(self.transforms) + list(getattr(other, "transforms", [other])))

        def __radd__(self, other):
            return TL(list(getattr(other, "transforms", [other])) + list(self.transforms))

    class NoOp(TL):
        def __init__(self):
            super().__init__([])

    class ResizeShortestEdge:
        def __init__(self, short, max_size):
            self.short, self.max_size = short, max_size

    class RandomFlip:
        def __init__(self, prob=1.0):
            assert prob == 1.0

    def apply_augmentations(augs, image):
        tf = []
        for a in augs:
            if isinstance(a, ResizeShortestEdge):
                h, w = image.shape[:2]
                nh, nw = T._shortest_edge_size(h, w, a.short, a.max_size)
                image = T._resize_image(image, nh, nw)
                tf.append(P.ResizeTransform(h, w, nh, nw))
            else:
                tf.append(P.HFlipTransform(image.shape[1]))
                image = image[:, ::-1]
        return image, TL(tf)

    sys.modules["detectron2.data.detection_utils"].read_image = None  # (extend the shim module; later generators read others)
    _mod("detectron2.data.transforms", RandomFlip=RandomFlip, ResizeShortestEdge=ResizeShortestEdge,
         ResizeTransform=lambda h, w, nh, nw: TL([P.ResizeTransform(h, w, nh, nw)]),
         apply_augmentations=apply_augmentations)
    sys.modules["detectron2.modeling.meta_arch"].GeneralizedRCNN = type("GeneralizedRCNN", (), {})
    _mod("detectron2.modeling.roi_heads.fast_rcnn",
         fast_rcnn_inference_single_image=lambda *a: r.frcnn.fast_rcnn_inference_single_image(*a)[:2])
    _mod("fvcore.transforms", HFlipTransform=P.HFlipTransform, NoOpTransform=NoOp)


def golden_tta(r):
    """G15: the REFERENCE's DatasetMapperTTAAVG + GeneralizedRCNNWithTTAAVG (test_time_augmentation_avg.py:67-334)
    around the reference model: 2 short-edge sizes x flip = 4 views of one image, per-view inference, boxes mapped back
    through the inverse transforms, mean over views, one tail pass (stand-ins: _install_tta)."""
    cfg, model, post = _eval_reference_model(r)
    _install_tta(r)
    tta_mod = load_ref("wsovod.modeling.test_time_augmentation_avg", "wsovod/modeling/test_time_augmentation_avg.py")
    model.classifier = torch.randn(20, 512, generator=torch.Generator().manual_seed(77))
    cfg.TEST.DETECTIONS_PER_IMAGE = 100
    cfg.MODEL.KEYPOINT_ON = False
    mapper = tta_mod.DatasetMapperTTAAVG([192, 256], 4000, True, 0)
    tta = tta_mod.GeneralizedRCNNWithTTAAVG(cfg, model, mapper)
    inp = to_inputs(gen.seeded_batch(1, 60, 20, 256, 352, seed=21))[0]
    # the reference mapper leaves already-mapped `proposals` untouched when proposal_topk == 0; views of other sizes
    # need their boxes moved with the image, so the harness hands each view its transformed boxes (what
    # transform_proposals does for proposal_topk > 0, minus the top-k cut) through the mapper hook below
    orig_call = mapper.__class__.__call__

    def mapped(self, d):
        views = orig_call(self, d)
        for v in views:
            p = v["proposals"]
            q = S.Instances(tuple(v["image"].shape[1:]), **p.get_fields())
            b = S.Boxes(torch.from_numpy(v["transforms"].apply_box(p.proposal_boxes.tensor.numpy())).float())
            b.clip(q.image_size)
            q.proposal_boxes = b
            v["proposals"] = q
        return views

    mapper.__class__.__call__ = mapped
    captured = {}
    orig_get = tta._get_augmented_boxes

    def cap(aug, tfms):
        captured["views"] = [(a["image"].clone(), a["proposals"].proposal_boxes.tensor.clone()) for a in aug]
        o = orig_get(aug, tfms)
        captured["avg"] = o
        return o

    tta._get_augmented_boxes = cap
    with torch.no_grad():
        out = tta([inp])[0]["instances"]
    arrays = {"classifier": model.classifier, "avg_boxes": captured["avg"][0], "avg_scores": captured["avg"][1],
              "boxes": out.pred_boxes.tensor, "scores": out.scores, "classes": out.pred_classes}
    for i, (img, bx) in enumerate(captured["views"]):
        arrays[f"view{i}/shape"] = np.array(img.shape)
        arrays[f"view{i}/image_checksum"] = img.double().sum()
        arrays[f"view{i}/proposal_boxes"] = bx
    save("g15_tta_avg", **arrays)


def golden_tta_union(r):
    """G17: the REFERENCE's DatasetMapperTTAUNION + GeneralizedRCNNWithTTAUNION (test_time_augmentation_union.py:66-330;
    the wrapper the reference picks when the model has an RPN) around the reference model WITH its RPN branch in eval
    mode: 2 short-edge sizes x flip = 4 views, the mapper's own `transform_proposals` (proposal_topk > 0) moves the loaded
    boxes with each view, per-view inference (RPN boxes + loaded boxes -> heads -> detection tail), the views' detections
    mapped back through the inverse transforms and pooled, one more tail pass at threshold 1e-8.  Stored: every view's
    image checksum / proposals / detections, the pooled boxes and the merged result."""
    cfg, model, shapes = _install_rpn(r)()
    _eval_patch = _eval_reference_model  # (its stand-ins: postprocessing.py + brute-force NMS in the detection tail)
    sys.modules["detectron2.structures"].ROIMasks = _Unsupported
    post = load_ref("wsovod.modeling.postprocessing", "wsovod/modeling/postprocessing.py")
    r.meta.detector_postprocess = post.detector_postprocess
    r.frcnn.batched_nms = brute_force_batched_nms
    _install_tta(r)
    tta_mod = load_ref("wsovod.modeling.test_time_augmentation_union", "wsovod/modeling/test_time_augmentation_union.py")
    model.eval()
    model.classifier = torch.randn(20, 512, generator=torch.Generator().manual_seed(79))
    cfg.TEST.DETECTIONS_PER_IMAGE = 100
    cfg.MODEL.KEYPOINT_ON = False
    cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST = 0.3
    mapper = tta_mod.DatasetMapperTTAUNION([192, 256], 4000, True, 4000)
    tta = tta_mod.GeneralizedRCNNWithTTAUNION(cfg, model, mapper)
    inp = to_inputs(gen.seeded_batch(1, 60, 20, 256, 352, seed=23))[0]
    captured = {}
    orig_batch = tta._batch_inference

    def cap_batch(aug, det=None):
        captured["views"] = [(a["image"].clone(), a["proposals"].proposal_boxes.tensor.clone(),
                              a["proposals"].objectness_logits.clone()) for a in aug]
        outs = orig_batch(aug, det)
        captured["dets"] = [(o.pred_boxes.tensor.clone(), o.scores.clone(), o.pred_classes.clone()) for o in outs]
        return outs

    tta._batch_inference = cap_batch
    orig_get = tta._get_augmented_boxes

    def cap_get(aug, tfms):
        o = orig_get(aug, tfms)
        captured["pooled"] = (o[0].clone(), torch.stack(list(o[1])), torch.stack(list(o[2])))
        return o

    tta._get_augmented_boxes = cap_get
    rpn_props = []
    orig_pred = model.proposal_generator.predict_proposals

    def cap_props(*a, **k):
        o = orig_pred(*a, **k)
        rpn_props.append([(p.proposal_boxes.tensor.clone(), p.objectness_logits.clone()) for p in o])
        return o

    model.proposal_generator.predict_proposals = cap_props
    with torch.no_grad():
        out = tta([inp])[0]["instances"]
    arrays = {"classifier": model.classifier, "pooled_boxes": captured["pooled"][0], "pooled_scores": captured["pooled"][1],
              "pooled_classes": captured["pooled"][2], "boxes": out.pred_boxes.tensor, "scores": out.scores,
              "classes": out.pred_classes}
    for i, ((img, bx, ol), (db, ds, dc)) in enumerate(zip(captured["views"], captured["dets"])):
        arrays[f"view{i}/shape"] = np.array(img.shape)
        arrays[f"view{i}/image_checksum"] = img.double().sum()
        arrays[f"view{i}/proposal_boxes"], arrays[f"view{i}/objectness"] = bx, ol
        arrays[f"view{i}/det_boxes"], arrays[f"view{i}/det_scores"], arrays[f"view{i}/det_classes"] = db, ds, dc
        arrays[f"view{i}/rpn_boxes"], arrays[f"view{i}/rpn_logits"] = rpn_props[i][0]
    save("g17_tta_union", **arrays)


def golden_subsample(r):
    """G16: the REFERENCE's WSOVODROIHeads._sample_proposals_wsl (roi_heads.py:1566-1603) on label vectors longer
    than BATCH_SIZE_PER_IMAGE (the shipped RPN form: 4000 loaded + 1024 RPN boxes against 4096) and with
    POSITIVE_FRACTION < 1.  detectron2's random subsample_labels (un-vendored) is replaced by the deterministic
    first-k rule of tests/golden/gen.py (randperm cannot be reproduced across implementations)."""
    orig_subsample = r.roi_heads.subsample_labels
    r.roi_heads.subsample_labels = gen.first_k_subsample
    cfg, model, sd, shapes = build_ref_model(r, 18, 20, 512, seed=1)
    rh = model.roi_heads
    arrays = {}
    g = torch.Generator().manual_seed(91)
    cases = [(5024, 4096, 1.0, 0.02), (5024, 4096, 0.25, 0.4), (3000, 512, 0.25, 0.05), (300, 4096, 0.5, 0.9),
             (4097, 4096, 1.0, 0.0)]
    for i, (R_, num, frac, p_fg) in enumerate(cases):
        rh.batch_size_per_images[0], rh.positive_sample_fractions[0] = num, frac
        G_ = 3
        matched_idxs = torch.randint(0, G_, (R_,), generator=g)
        u = torch.rand(R_, generator=g)
        matched_labels = (u < p_fg).to(torch.int8)  # Matcher([0.5],[0,1]): 1 = foreground, 0 = background
        gt_classes = torch.tensor([3, 7, 11])
        idx, lab = rh._sample_proposals_wsl(0, matched_idxs, matched_labels, gt_classes)
        assert torch.equal(idx, torch.arange(R_))
        full = gt_classes[matched_idxs].clone()
        full[matched_labels == 0] = rh.num_classes
        arrays[f"case{i}/params"] = np.array([R_, num, frac, rh.num_classes], dtype=np.float64)
        arrays[f"case{i}/labels_in"], arrays[f"case{i}/labels_out"] = full, lab
    r.roi_heads.subsample_labels = orig_subsample
    save("g16_subsample", **arrays)


def golden_formats():
    """G13: the REFERENCE's load_proposals_into_dataset (data/build.py:112-173) and unique_boxes /
    transform_proposals (data/detection_utils.py:206-265) on a synthetic D1-style proposal pickle (`indexes` /
    `scores` aliases, shuffled ids, duplicates, tiny boxes).  detectron2's BoxMode / TransformList are the
    restated stand-ins (XYXY proposals: identity conversion)."""
    import enum
    import pickle
    import tempfile
    from wsovod_amd.data import proposals as P

    class BoxMode(enum.IntEnum):
        XYXY_ABS = 0
        XYWH_ABS = 1

        @staticmethod
        def convert(box, from_mode, to_mode):
            assert int(from_mode) == int(to_mode) == 0
            return box

    class _Any:
        def __getattr__(self, k):
            return None

    for name in ("detectron2", "detectron2.data", "detectron2.utils", "wsovod", "wsovod.data"):
        if name not in sys.modules:
            _mod(name)
    _mod("pycocotools")
    _mod("pycocotools.mask")
    _mod("termcolor", colored=lambda s, *a, **k: s)
    _mod("detectron2.config", CfgNode=C.CfgNode,
         configurable=lambda *a, **k: a[0] if a and callable(a[0]) else (lambda f: f))
    _mod("detectron2.data.transforms")
    _mod("detectron2.data.catalog", MetadataCatalog=None, DatasetCatalog=None)
    _mod("detectron2.data.common", AspectRatioGroupedDataset=None, DatasetFromList=None, MapDataset=None,
         ToIterableDataset=None)
    _mod("detectron2.data.detection_utils", check_metadata_consistency=None)
    _mod("detectron2.data.samplers", InferenceSampler=None, RepeatFactorTrainingSampler=None, TrainingSampler=None)
    _mod("detectron2.structures", Boxes=S.Boxes, Instances=S.Instances, BoxMode=BoxMode, BitMasks=None, Keypoints=None,
         PolygonMasks=None, RotatedBoxes=None, polygons_to_bitmask=None)
    _mod("detectron2.utils.comm", get_world_size=lambda: 1)
    _mod("detectron2.utils.env", seed_all_rng=None)
    _mod("detectron2.utils.file_io", PathManager=types.SimpleNamespace(open=open))
    _mod("detectron2.utils.logger", _log_api_usage=None, log_first_n=None)
    _mod("wsovod.data.common", ClassAspectRatioGroupedDataset=None)
    _mod("wsovod.data.dataset_mapper", DatasetMapper=None)
    du = load_ref("wsovod.data.detection_utils", "wsovod/data/detection_utils.py")
    bd = load_ref("wsovod.data.build", "wsovod/data/build.py")

    pk, recs = gen.proposal_pickle()
    path = os.path.join(tempfile.mkdtemp(prefix="golden_"), "props.pkl")
    with open(path, "wb") as f:
        pickle.dump(pk, f)
    recs = bd.load_proposals_into_dataset(recs, path)
    arrays = {}
    for i, rec in enumerate(recs):
        arrays[f"rec{i}/boxes"], arrays[f"rec{i}/logits"] = rec["proposal_boxes"], rec["proposal_objectness_logits"]
        d = dict(rec)
        h, w = rec["height"], rec["width"]
        tl = P.TransformList([P.ResizeTransform(h, w, h * 2, w * 2)] + ([P.HFlipTransform(w * 2)] if i % 2 else []))
        du.transform_proposals(d, (h * 2, w * 2), tl, proposal_topk=50, min_box_size=8)
        arrays[f"rec{i}/out_boxes"] = d["proposals"].proposal_boxes.tensor
        arrays[f"rec{i}/out_logits"] = d["proposals"].objectness_logits
        arrays[f"rec{i}/unique"] = du.unique_boxes(S.Boxes(torch.as_tensor(rec["proposal_boxes"])))
    save("g13_proposal_formats", **arrays)


def golden_sampler():
    """G11: the REFERENCE's MultiDatasetTrainingSampler (repeat factors with class-aware sampling on one dataset,
    and the per-rank index streams of a 2-rank job)."""
    import itertools

    rank_box = {"rank": 0, "world": 2}
    _mod("detectron2")
    _mod("detectron2.data")
    _mod("detectron2.data.samplers", RepeatFactorTrainingSampler=_Unsupported)
    _mod("detectron2.utils")
    _mod("detectron2.utils.comm", get_world_size=lambda: rank_box["world"], get_rank=lambda: rank_box["rank"],
         shared_random_seed=lambda: 0)
    for name in ("wsovod", "wsovod.data", "wsovod.data.samplers"):
        if name not in sys.modules:
            _mod(name)
    m = load_ref("wsovod.data.samplers.distributed_sampler_multi_dataset",
                 "wsovod/data/samplers/distributed_sampler_multi_dataset.py")
    dicts = gen.sampler_dataset_dicts()
    rf = m.MultiDatasetTrainingSampler.get_repeat_factors(dicts, 3, [1, 1.5, 2], [False] * 3, [False, False, True],
                                                          0.001, 1.0)
    arrays = {"repeat_factors": rf}
    for rank in (0, 1):
        rank_box["rank"] = rank
        arrays[f"stream_rank{rank}"] = np.array(list(itertools.islice(iter(
            m.MultiDatasetTrainingSampler(rf, seed=42)), 400)))
    rank_box.update(rank=0, world=1)
    arrays["stream_noshuffle"] = np.array(list(itertools.islice(iter(
        m.MultiDatasetTrainingSampler(rf, shuffle=False, seed=7)), 300)))
    save("g11_multi_dataset_sampler", **arrays)


def golden_edges(r):
    """G18: the branches of the reference that the happy-path vectors never reach, each run through the reference's own
    code: (a) the MIL head at num_classes == 1 (fast_rcnn_open_vocabulary.py:338-357), directly and as a whole step;
    (b) get_pgt_top_k on images whose candidate boxes are all <= 20 px^2 (roi_heads.py:1090-1111,1181-1207) followed by
    the labelling, directly and as a whole step with gradients; (c) the refinement losses with -1 ignores, zero weights
    and an all-background batch (fast_rcnn_open_vocabulary.py:813-820,864-878); (d) OpenVocabularyClassifier with
    use_bias != 0, at K = 80 / D = 768, with and without norm_weight (open_vocabulary_classifier.py:35-37,79-105)."""
    import pickle
    import tempfile

    arrays = {}
    K, D = 20, 512
    cfg, model, sd, shapes = build_ref_model(r, 18, K, D, seed=1)  # the state of g8 (shapes_r18_k20, seed 1)
    model.train()
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 1e-12
    rh = model.roi_heads

    # ---- (b1) whole step on the edge batch ----
    batch = gen.edge_batch(K)
    captured = {}

    def cap(name, fn):
        def wrapped(*a, **k):
            o = fn(*a, **k)
            captured[name] = o
            return o
        return wrapped

    orig = (rh.object_miner.forward, rh.box_refinery[0].forward, rh.label_and_sample_proposals_wsl, rh.get_pgt_top_k)
    rh.object_miner.forward = cap("miner", orig[0])
    rh.box_refinery[0].forward = cap("refine", orig[1])
    rh.label_and_sample_proposals_wsl = cap("proposals_k", orig[2])
    rh.get_pgt_top_k = cap("targets", orig[3])
    loss_dict = model(to_inputs(batch))
    sum(loss_dict.values()).backward()
    rh.object_miner.forward, rh.box_refinery[0].forward, rh.label_and_sample_proposals_wsl, rh.get_pgt_top_k = orig
    p = "step/"
    for k, v in loss_dict.items():
        arrays[p + "loss/" + k] = v
    arrays[p + "mining_scores"] = captured["miner"][0]
    arrays[p + "refine_logits"] = captured["refine"][0]
    arrays[p + "refine_deltas"] = captured["refine"][1]
    arrays[p + "pred_class_img_logits"] = rh.pred_class_img_logits
    for f in ("gt_classes", "gt_weights", "gt_scores"):
        arrays[p + "label/" + f] = torch.cat([getattr(q, f) for q in captured["proposals_k"]])
    arrays[p + "label/gt_boxes"] = torch.cat([q.gt_boxes.tensor for q in captured["proposals_k"]])
    arrays[p + "pgt/num"] = np.array([len(t) for t in captured["targets"]])
    arrays[p + "pgt/gt_boxes"] = torch.cat([t.gt_boxes.tensor for t in captured["targets"]])
    for f in ("gt_classes", "gt_weights", "gt_scores"):
        arrays[p + "pgt/" + f] = torch.cat([getattr(t, f) for t in captured["targets"]])
    for k, q in model.named_parameters():
        if q.requires_grad and q.grad is not None:
            arrays[p + "gradnorm/" + k] = q.grad.norm()
            arrays[p + "gradsample/" + k] = gen.strided_sample(q.grad, 1024)
    model.zero_grad(set_to_none=True)
    assert arrays[p + "pgt/num"].tolist()[1] == 1 and float(arrays[p + "pgt/gt_boxes"][arrays[p + "pgt/num"][0]][0]) == -10000.0

    # ---- (b2) get_pgt_top_k + labelling called directly on crafted candidates ----
    g = torch.Generator().manual_seed(41)
    sizes = [(120, 160), (120, 160), (96, 128), (96, 128)]
    nums = [12, 9, 7, 5]
    boxes, scores = [], []
    for i, (n, (h, w)) in enumerate(zip(nums, sizes)):
        x0 = torch.rand(n, generator=g) * (w - 40)
        y0 = torch.rand(n, generator=g) * (h - 40)
        wh = 8 + torch.rand(n, 2, generator=g) * 30
        boxes.append(torch.stack([x0, y0, x0 + wh[:, 0], y0 + wh[:, 1]], 1))
        scores.append(torch.rand(n, K, generator=g) / n)
    boxes[1][:, 2:] = boxes[1][:, :2] + torch.tensor([5.0, 4.0])      # image 1: every box has area exactly 20 (not > 20)
    gt_int = [torch.tensor([3, 11]), torch.tensor([0, 7, 19]), torch.tensor([5]), torch.tensor([2, 4])]
    top = int(scores[2][:, 5].argmax())                               # image 2: its best box for class 5 is filtered
    boxes[2][top, 2:] = boxes[2][top, :2] + torch.tensor([4.0, 4.0])
    boxes[3][:, 2:] = boxes[3][:, :2] + torch.tensor([3.0, 3.0])      # image 3: all filtered but one
    boxes[3][4] = torch.tensor([10.0, 10.0, 60.0, 50.0])
    img_logits = torch.rand(4, K, generator=g).clamp(1e-6, 1 - 1e-6)
    props = [S.Instances(sz, proposal_boxes=S.Boxes(b.clone()), objectness_logits=torch.zeros(len(b)))
             for sz, b in zip(sizes, boxes)]
    rh.gt_classes_img_int = gt_int
    rh.pred_class_img_logits = img_logits
    rh.images = [torch.zeros(3, *sz) for sz in sizes]
    targets = rh.get_pgt_top_k([b.clone() for b in boxes], [s_.clone() for s_ in scores], props)
    labelled = rh.label_and_sample_proposals_wsl(0, props, targets)
    p = "direct/"
    arrays[p + "nums"] = np.array(nums)
    arrays[p + "boxes"] = torch.cat(boxes)
    arrays[p + "scores"] = torch.cat(scores)
    arrays[p + "gt_int"] = torch.cat(gt_int)
    arrays[p + "gt_int_num"] = np.array([len(t) for t in gt_int])
    arrays[p + "img_logits"] = img_logits
    arrays[p + "pgt/num"] = np.array([len(t) for t in targets])
    arrays[p + "pgt/gt_boxes"] = torch.cat([t.gt_boxes.tensor for t in targets])
    for f in ("gt_classes", "gt_weights", "gt_scores"):
        arrays[p + "pgt/" + f] = torch.cat([getattr(t, f) for t in targets])
        arrays[p + "label/" + f] = torch.cat([getattr(q, f) for q in labelled])
    arrays[p + "label/gt_boxes"] = torch.cat([q.gt_boxes.tensor for q in labelled])
    assert arrays[p + "pgt/num"].tolist() == [2, 1, 1, 2], arrays[p + "pgt/num"]

    # ---- (c) refinement losses on crafted labels ----
    ref0 = rh.box_refinery[0]
    n = 40
    logits = torch.randn(n, K + 1, generator=g) * 3
    deltas = torch.randn(n, 4, generator=g) * 0.1
    pb = torch.cat([boxes[0], boxes[0], boxes[0], boxes[0][:4]])[:n].clone()
    gb = pb + torch.randn(n, 4, generator=g) * 2
    gb[:, 2:] = torch.maximum(gb[:, 2:], gb[:, :2] + 1)
    wts = torch.rand(n, generator=g)
    cases = {
        "ignores": torch.randint(0, K + 1, (n,), generator=g),
        "all_background": torch.full((n,), K, dtype=torch.int64),
        "zero_weights": torch.randint(0, K + 1, (n,), generator=g),
        "one_foreground": torch.full((n,), K, dtype=torch.int64),
    }
    cases["ignores"][::3] = -1
    cases["one_foreground"][17] = 4
    arrays["loss/logits"], arrays["loss/deltas"], arrays["loss/proposal_boxes"], arrays["loss/gt_boxes"] = logits, deltas, pb, gb
    for name, gc in cases.items():
        w = wts.clone()
        if name == "zero_weights":
            w[1::2] = 0.0
            w[4] = 1e-13  # below the 1e-12 validity threshold: contributes to the sum, not to the count
        q = S.Instances((120, 160), proposal_boxes=S.Boxes(pb.clone()), gt_boxes=S.Boxes(gb.clone()), gt_classes=gc.clone(),
                        gt_weights=w.clone())
        lg, dl = logits.clone().requires_grad_(True), deltas.clone().requires_grad_(True)
        out = ref0.losses((lg, dl), [q])
        tot = out["loss_cls_r0"] + out["loss_box_reg_r0"]
        if tot.requires_grad:
            tot.backward()
        arrays[f"loss/{name}/gt_classes"], arrays[f"loss/{name}/gt_weights"] = gc, w
        arrays[f"loss/{name}/loss_cls"], arrays[f"loss/{name}/loss_box"] = out["loss_cls_r0"], out["loss_box_reg_r0"]
        arrays[f"loss/{name}/dlogits"] = lg.grad if lg.grad is not None else torch.zeros_like(logits)
        arrays[f"loss/{name}/ddeltas"] = dl.grad if dl.grad is not None else torch.zeros_like(deltas)

    # ---- (a) MIL head at num_classes == 1 ----
    cfg1, model1, sd1, shapes1 = build_ref_model(r, 18, 1, D, seed=5)
    model1.train()
    for m in model1.modules():
        if isinstance(m, nn.Dropout):
            m.p = 1e-12
    om = model1.roi_heads.object_miner
    x = gen.edge_features("k1", 40)
    arrays["k1/cls_w"], arrays["k1/cls_b"] = om.cls.weight, om.cls.bias
    arrays["k1/det_w"], arrays["k1/det_b"] = om.det.weight, om.det.bias
    pl = [S.Instances((96, 128), proposal_boxes=S.Boxes(torch.zeros(k_, 4))) for k_ in (25, 15)]
    with torch.no_grad():
        s2, _ = om(x, pl)
        s1, _ = om(x[:25], pl[:1])
        s0, _ = om(x, None)
    arrays["k1/scores_two_images"], arrays["k1/scores_one_image"], arrays["k1/scores_no_proposals"] = s2, s1, s0
    oh = torch.tensor([[1.0], [0.0]])
    xs = x.clone().requires_grad_(True)
    pred = om(xs, pl)
    lm = om.losses(pred, pl, oh)["loss_cls_object_mining"]
    lm.backward()
    arrays["k1/gt_oh"], arrays["k1/loss"], arrays["k1/dx_sample"] = oh, lm, gen.strided_sample(xs.grad, 2048)
    arrays["k1/dcls_w"], arrays["k1/ddet_w"] = om.cls.weight.grad, om.det.weight.grad
    model1.zero_grad(set_to_none=True)
    # the whole step at K = 1 (refinement over 2 columns: class 0 + background)
    save("shapes_r18_k1", keys=np.array(list(shapes1.keys())), shapes=np.array([str(v) for v in shapes1.values()]))
    b1 = gen.seeded_batch(2, 20, 1, 128, 160, seed=9)
    ld = model1(to_inputs(b1))
    sum(ld.values()).backward()
    for k, v in ld.items():
        arrays["k1/step/loss/" + k] = v
    arrays["k1/step/pred_class_img_logits"] = model1.roi_heads.pred_class_img_logits
    for k, q in model1.named_parameters():
        if q.requires_grad and q.grad is not None:
            arrays["k1/step/gradnorm/" + k] = q.grad.norm()

    # ---- (d) OpenVocabularyClassifier: use_bias != 0, K = 80 / D = 768, norm_weight on / off ----
    Kc, Dc = 80, 768
    tmp = tempfile.mkdtemp(prefix="golden_")
    emb_path = os.path.join(tmp, "emb.pkl")
    emb, clsf = gen.edge_embeddings(Kc, Dc)
    with open(emb_path, "wb") as f:
        pickle.dump(emb.numpy(), f)
    xin = gen.edge_features("ovc", 33)
    for tag, kw in (("bias", dict(use_bias=-2.0, norm_weight=True)), ("nobias", dict(use_bias=0.0, norm_weight=True)),
                    ("bias_nonorm", dict(use_bias=0.75, norm_weight=False))):
        head = r.ovc.OpenVocabularyClassifier(S.ShapeSpec(channels=4096), num_classes=Kc, weight_path=emb_path,
                                              weight_dim=Dc, norm_temperature=50.0, **kw)
        shp = {"cls.projection." + k: tuple(v.shape) for k, v in head.projection.state_dict().items()}
        st = gen.seeded_state(shp, 23)  # the tests rebuild the projection from (shapes, seed 23)
        head.projection.load_state_dict({k[len("cls.projection."):]: v for k, v in st.items()})
        xg = xin.clone().requires_grad_(True)
        out = head(xg, None, append_background=True)
        out.square().mean().backward()
        arrays[f"ovc/{tag}/logits_bg"] = out
        arrays[f"ovc/{tag}/dx_sample"] = gen.strided_sample(xg.grad, 2048)
        if head.use_bias:
            arrays[f"ovc/{tag}/dcls_bias"] = head.cls_bias.grad
        with torch.no_grad():
            arrays[f"ovc/{tag}/logits_nobg"] = head(xin, None, append_background=False)
            arrays[f"ovc/{tag}/logits_classifier"] = head(xin, clsf, append_background=True)
    save("g18_edge_branches", **arrays)



def golden_trainable_stage(r):
    """G19: MODEL.BACKBONE.FREEZE_AT = 4 (resnet_wsl.py:530-552: res5 trainable, the stem and res2 - res4 frozen) -- one
    whole training step of the reference with the gradient running through the reference's own RoIPool backward into
    res5: losses, logits, labels and the gradient of EVERY trainable tensor, the eight res5 conv weights included."""
    K, D = 20, 512
    cfg, model, sd, shapes = build_ref_model(r, 18, K, D, seed=1, freeze_at=4)  # the state of g8 (shapes_r18_k20, seed 1)
    model.train()
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 1e-12
    train = [k for k, p_ in model.named_parameters() if p_.requires_grad]
    assert any(k.startswith("backbone.res5.") for k in train) and not any(k.startswith("backbone.res4.") for k in train)
    batch = gen.seeded_batch(2, 24, K, 160, 208, seed=11)
    rh = model.roi_heads
    captured = {}

    def cap(name, fn):
        def wrapped(*a, **k):
            o = fn(*a, **k)
            captured[name] = o
            return o
        return wrapped

    rh.object_miner.forward = cap("miner", rh.object_miner.forward)
    rh.box_refinery[0].forward = cap("refine", rh.box_refinery[0].forward)
    rh.label_and_sample_proposals_wsl = cap("proposals_k", rh.label_and_sample_proposals_wsl)
    loss_dict = model(to_inputs(batch))
    sum(loss_dict.values()).backward()
    arrays = {f"loss/{k}": v for k, v in loss_dict.items()}
    arrays["mining_scores"] = captured["miner"][0]
    arrays["refine_logits"] = captured["refine"][0]
    arrays["label/gt_classes"] = torch.cat([q.gt_classes for q in captured["proposals_k"]])
    arrays["train_keys"] = np.array(train)
    for k, q in model.named_parameters():
        if q.requires_grad:
            assert q.grad is not None, k
            arrays[f"gradnorm/{k}"] = q.grad.norm()
            arrays[f"gradsample/{k}"] = gen.strided_sample(q.grad, 1024)
    save("g19_freeze_at_4", **arrays)


def main():
    if "--only-sampler" in sys.argv:
        return golden_sampler()
    if "--only-formats" in sys.argv:
        return golden_formats()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    r = load_reference()
    if "--only-mixed" in sys.argv:
        return golden_mixed(r)
    if "--only-rpn" in sys.argv:
        return golden_rpn(r)
    if "--only-union" in sys.argv:
        return golden_tta_union(r)
    if "--only-edges" in sys.argv:
        return golden_edges(r)
    if "--only-trainable" in sys.argv:
        return golden_trainable_stage(r)
    if "--only-eval" in sys.argv:
        golden_eval_tail(r)
        golden_subsample(r)
        return golden_tta(r)

    # ---------------- G2: RoIPool / ROIAlign (reference C++ op) ----------------
    from tests.util import random_rois

    feat = torch.randn(2, 8, 75, 100, generator=torch.Generator().manual_seed(11))
    rois = random_rois(64, 2, 600, 800, seed=3)
    out, arg = roi_ops.ref_roi_pool_forward(feat, rois, 0.125, (7, 7))
    g = torch.randn(out.shape, generator=torch.Generator().manual_seed(12))
    gi = roi_ops.ref_roi_pool_backward(g, rois, arg, 0.125, feat.shape)
    save("g2_roi_pool", feat=feat, rois=rois, out=out, argmax=arg, grad_out=g, grad_in=gi)

    # ---------------- G1/G3-G9: whole model at the plumbing scale ----------------
    K, D = 20, 512
    cfg, model, sd, shapes = build_ref_model(r, 18, K, D, seed=1)
    model.train()
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 1e-12  # dropout RNG cannot be reproduced: golden vectors are taken with p = 1e-12 ~ off;
            # (kept in training mode so it still returns a fresh tensor for the reference's in-place `+=`)
    save("shapes_r18_k20", keys=np.array(list(shapes.keys())), shapes=np.array([str(v) for v in shapes.values()]))

    # G1 backbone on a small image pair
    x = torch.randn(2, 3, 96, 128, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        feats = model.backbone(x)
    save("g1_backbone_small", x=x, res5=feats["res5"])

    # G8 whole training step: 4 ragged images ~ 320x416, R = 64 (last image 57), K = 20
    batch = gen.seeded_batch(4, 64, K, 320, 416, seed=2)
    inputs = to_inputs(batch)
    captured = {}
    rh = model.roi_heads
    orig_miner, orig_ref, orig_head = rh.object_miner.forward, rh.box_refinery[0].forward, rh.box_head.forward

    def cap(name, fn):
        def wrapped(*a, **k):
            o = fn(*a, **k)
            captured[name] = o
            return o
        return wrapped

    rh.object_miner.forward = cap("miner", orig_miner)
    rh.box_refinery[0].forward = cap("refine", orig_ref)
    rh.box_head.forward = cap("neck", orig_head)
    orig_label = rh.label_and_sample_proposals_wsl
    rh.label_and_sample_proposals_wsl = cap("proposals_k", orig_label)
    orig_pgt = rh.get_pgt_top_k
    rh.get_pgt_top_k = cap("targets", orig_pgt)
    loss_dict = model(inputs)
    total = sum(loss_dict.values())
    total.backward()
    grads = {k: p.grad for k, p in model.named_parameters() if p.requires_grad and p.grad is not None}
    arrays = {f"loss/{k}": v for k, v in loss_dict.items()}
    arrays["mining_scores"] = captured["miner"][0]
    arrays["refine_logits"] = captured["refine"][0]
    arrays["refine_deltas"] = captured["refine"][1]
    arrays["neck_out_sample"] = gen.strided_sample(captured["neck"], 8192)
    arrays["pred_class_img_logits"] = rh.pred_class_img_logits
    arrays["label/gt_classes"] = torch.cat([p.gt_classes for p in captured["proposals_k"]])
    arrays["label/gt_boxes"] = torch.cat([p.gt_boxes.tensor for p in captured["proposals_k"]])
    arrays["label/gt_weights"] = torch.cat([p.gt_weights for p in captured["proposals_k"]])
    arrays["label/gt_scores"] = torch.cat([p.gt_scores for p in captured["proposals_k"]])
    arrays["pgt/num"] = np.array([len(t) for t in captured["targets"]])
    arrays["pgt/gt_boxes"] = torch.cat([t.gt_boxes.tensor for t in captured["targets"]])
    arrays["pgt/gt_classes"] = torch.cat([t.gt_classes for t in captured["targets"]])
    arrays["pgt/gt_weights"] = torch.cat([t.gt_weights for t in captured["targets"]])
    for k, gth in grads.items():
        arrays[f"gradnorm/{k}"] = gth.norm()
        arrays[f"gradsample/{k}"] = gen.strided_sample(gth, 2048)
    save("g8_train_step_r18_k20", **arrays)

    # G1b: R50 (BottleneckBlock) backbone on a small image pair, random init with seeded FrozenBN stats
    cfg50, model50, sd50, shapes50 = build_ref_model(r, 50, 20, 512, seed=3)
    bshapes = {k: v for k, v in shapes50.items() if k.startswith("backbone.")}
    save("shapes_r50_backbone", keys=np.array(list(bshapes.keys())), shapes=np.array([str(v) for v in bshapes.values()]))
    x50 = torch.randn(2, 3, 64, 96, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        f50 = model50.backbone(x50)
    save("g1_backbone_r50_small", x=x50, res5=f50["res5"])

    # G5: OV classifier variants on a fixed feature matrix
    feat_x = torch.randn(48, 4096, generator=torch.Generator().manual_seed(21)) * 0.5
    head = rh.box_refinery[0].cls
    with torch.no_grad():
        l_default = head(feat_x, None, append_background=True)
        l_nobg = head(feat_x, None, append_background=False)
        clsf = torch.randn(33, D, generator=torch.Generator().manual_seed(22))
        l_call = head(feat_x, clsf, append_background=True)
    save("g5_ov_classifier", x=feat_x, classifier=clsf, logits_default=l_default, logits_nobg=l_nobg,
         logits_classifier=l_call)

    # G9: data-aware head on a fixed feature map
    fm = torch.randn(3, 512, 9, 13, generator=torch.Generator().manual_seed(31))
    props = [S.Instances((72, 104), proposal_boxes=S.Boxes(torch.zeros(n, 4))) for n in (3, 1, 2)]
    with torch.no_grad():
        daf = model.data_aware_head({"res5": fm}, props)
    save("g9_data_aware", res5=fm, daf=daf, nums=np.array([3, 1, 2]))

    golden_eval_tail(r)
    golden_subsample(r)
    golden_tta(r)
    golden_mixed(r)
    golden_sampler()
    golden_rpn(r)
    golden_formats()
    golden_edges(r)
    golden_trainable_stage(r)


if __name__ == "__main__":
    main()
