"""Config + registry plumbing with Detectron2's surface (yacs/detectron2 are not installable here).

Provides `CfgNode` (attribute dict with `merge_from_file` honouring `_BASE_`, and
`merge_from_list`), `get_cfg()` = the subset of detectron2's defaults the hot path reads
(last row) + the reference's `add_wsovod_config`
(/root/reference/wsovod/config/defaults.py:7-96) so that the reference's own YAML files
(configs/*/WSOVOD_WSR_{18,50}_DC5_1x.yaml) load unchanged, plus `configurable` and `Registry`.
"""
import copy
import functools
import inspect
import os

import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def clone(self):
        return copy.deepcopy(self)

    def _merge(self, other, path=""):
        for k, v in other.items():
            if k == "_BASE_":
                continue
            if isinstance(v, dict):
                if k not in self or not isinstance(self[k], CfgNode):
                    self[k] = CfgNode()
                self[k]._merge(v, path + k + ".")
            else:
                if isinstance(v, str) and v.startswith("(") and v.endswith(")"):
                    try:
                        v = tuple(yaml.safe_load("[" + v[1:-1] + "]"))
                    except Exception:
                        pass
                self[k] = v

    def merge_from_file(self, path):
        with open(path) as f:
            data = yaml.safe_load(f) or {}
        base = data.get("_BASE_")
        if base:
            if not os.path.isabs(base):
                base = os.path.join(os.path.dirname(path), base)
            self.merge_from_file(base)
        self._merge(data)

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            if isinstance(val, str):
                try:
                    val = yaml.safe_load(val)
                except Exception:
                    pass
            node[parts[-1]] = val

    def freeze(self):
        return self


def _d2_defaults():
    C = CfgNode
    _C = C()
    _C.VERSION = 2
    _C.MODEL = C()
    _C.MODEL.LOAD_PROPOSALS = False
    _C.MODEL.DEVICE = "cuda"
    _C.MODEL.META_ARCHITECTURE = "GeneralizedRCNN"
    _C.MODEL.WEIGHTS = ""
    _C.MODEL.PIXEL_MEAN = [103.530, 116.280, 123.675]
    _C.MODEL.PIXEL_STD = [1.0, 1.0, 1.0]
    _C.INPUT = C()
    _C.INPUT.FORMAT = "BGR"
    _C.INPUT.MIN_SIZE_TRAIN = (800,)
    _C.INPUT.MAX_SIZE_TRAIN = 1333
    _C.INPUT.MIN_SIZE_TEST = 800
    _C.INPUT.MAX_SIZE_TEST = 1333
    _C.INPUT.CROP = C({"ENABLED": False})
    _C.DATASETS = C()
    _C.DATASETS.TRAIN = ()
    _C.DATASETS.TEST = ()
    _C.DATASETS.PROPOSAL_FILES_TRAIN = ()
    _C.DATASETS.PROPOSAL_FILES_TEST = ()
    _C.DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TRAIN = 2000
    _C.DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TEST = 1000
    _C.DATALOADER = C({"NUM_WORKERS": 4, "FILTER_EMPTY_ANNOTATIONS": True})
    _C.MODEL.BACKBONE = C({"NAME": "build_resnet_backbone", "FREEZE_AT": 2})
    _C.MODEL.PROPOSAL_GENERATOR = C({"NAME": "RPN", "MIN_SIZE": 0})
    _C.MODEL.ANCHOR_GENERATOR = C()
    _C.MODEL.ANCHOR_GENERATOR.NAME = "DefaultAnchorGenerator"
    _C.MODEL.ANCHOR_GENERATOR.SIZES = [[32, 64, 128, 256, 512]]
    _C.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS = [[0.5, 1.0, 2.0]]
    _C.MODEL.ANCHOR_GENERATOR.OFFSET = 0.0
    _C.MODEL.RPN = C()
    _C.MODEL.RPN.HEAD_NAME = "StandardRPNHead"
    _C.MODEL.RPN.IN_FEATURES = ["res4"]
    _C.MODEL.RPN.BOUNDARY_THRESH = -1
    _C.MODEL.RPN.IOU_THRESHOLDS = [0.3, 0.7]
    _C.MODEL.RPN.IOU_LABELS = [0, -1, 1]
    _C.MODEL.RPN.BATCH_SIZE_PER_IMAGE = 256
    _C.MODEL.RPN.POSITIVE_FRACTION = 0.5
    _C.MODEL.RPN.BBOX_REG_LOSS_TYPE = "smooth_l1"
    _C.MODEL.RPN.BBOX_REG_LOSS_WEIGHT = 1.0
    _C.MODEL.RPN.BBOX_REG_WEIGHTS = (1.0, 1.0, 1.0, 1.0)
    _C.MODEL.RPN.SMOOTH_L1_BETA = 0.0
    _C.MODEL.RPN.LOSS_WEIGHT = 1.0
    _C.MODEL.RPN.PRE_NMS_TOPK_TRAIN = 12000
    _C.MODEL.RPN.PRE_NMS_TOPK_TEST = 6000
    _C.MODEL.RPN.POST_NMS_TOPK_TRAIN = 2000
    _C.MODEL.RPN.POST_NMS_TOPK_TEST = 1000
    _C.MODEL.RPN.NMS_THRESH = 0.7
    _C.MODEL.RPN.CONV_DIMS = [-1]
    _C.MODEL.ROI_HEADS = C()
    _C.MODEL.ROI_HEADS.NAME = "Res5ROIHeads"
    _C.MODEL.ROI_HEADS.NUM_CLASSES = 80
    _C.MODEL.ROI_HEADS.IN_FEATURES = ["res4"]
    _C.MODEL.ROI_HEADS.IOU_THRESHOLDS = [0.5]
    _C.MODEL.ROI_HEADS.IOU_LABELS = [0, 1]
    _C.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 512
    _C.MODEL.ROI_HEADS.POSITIVE_FRACTION = 0.25
    _C.MODEL.ROI_HEADS.SCORE_THRESH_TEST = 0.05
    _C.MODEL.ROI_HEADS.NMS_THRESH_TEST = 0.5
    _C.MODEL.ROI_HEADS.PROPOSAL_APPEND_GT = True
    _C.MODEL.ROI_BOX_HEAD = C()
    _C.MODEL.ROI_BOX_HEAD.NAME = ""
    _C.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE = "smooth_l1"
    _C.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_WEIGHT = 1.0
    _C.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS = (10.0, 10.0, 5.0, 5.0)
    _C.MODEL.ROI_BOX_HEAD.SMOOTH_L1_BETA = 0.0
    _C.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION = 14
    _C.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO = 0
    _C.MODEL.ROI_BOX_HEAD.POOLER_TYPE = "ROIAlignV2"
    _C.MODEL.ROI_BOX_HEAD.NUM_FC = 0
    _C.MODEL.ROI_BOX_HEAD.FC_DIM = 1024
    _C.MODEL.ROI_BOX_HEAD.NUM_CONV = 0
    _C.MODEL.ROI_BOX_HEAD.CONV_DIM = 256
    _C.MODEL.ROI_BOX_HEAD.NORM = ""
    _C.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = False
    _C.MODEL.ROI_BOX_HEAD.TRAIN_ON_PRED_BOXES = False
    _C.MODEL.RESNETS = C()
    _C.MODEL.RESNETS.DEPTH = 50
    _C.MODEL.RESNETS.OUT_FEATURES = ["res4"]
    _C.MODEL.RESNETS.NUM_GROUPS = 1
    _C.MODEL.RESNETS.NORM = "FrozenBN"
    _C.MODEL.RESNETS.WIDTH_PER_GROUP = 64
    _C.MODEL.RESNETS.STRIDE_IN_1X1 = True
    _C.MODEL.RESNETS.RES5_DILATION = 1
    _C.MODEL.RESNETS.RES2_OUT_CHANNELS = 256
    _C.MODEL.RESNETS.STEM_OUT_CHANNELS = 64
    _C.MODEL.RESNETS.DEFORM_ON_PER_STAGE = [False, False, False, False]
    _C.MODEL.RESNETS.DEFORM_MODULATED = False
    _C.MODEL.RESNETS.DEFORM_NUM_GROUPS = 1
    _C.SOLVER = C()
    _C.SOLVER.MAX_ITER = 40000
    _C.SOLVER.BASE_LR = 0.001
    _C.SOLVER.MOMENTUM = 0.9
    _C.SOLVER.NESTEROV = False
    _C.SOLVER.WEIGHT_DECAY = 0.0001
    _C.SOLVER.WEIGHT_DECAY_BIAS = None
    _C.SOLVER.BIAS_LR_FACTOR = 1.0
    _C.SOLVER.STEPS = (30000,)
    _C.SOLVER.WARMUP_ITERS = 1000
    _C.SOLVER.IMS_PER_BATCH = 16
    _C.SOLVER.REFERENCE_WORLD_SIZE = 0
    _C.SOLVER.CLIP_GRADIENTS = C({"ENABLED": False, "CLIP_TYPE": "value", "CLIP_VALUE": 1.0, "NORM_TYPE": 2.0})
    _C.TEST = C()
    _C.TEST.DETECTIONS_PER_IMAGE = 100
    _C.TEST.EVAL_PERIOD = 0
    _C.TEST.AUG = C({"ENABLED": False, "MIN_SIZES": (400, 500, 600, 700, 800, 900, 1000, 1100, 1200), "MAX_SIZE": 4000,
                     "FLIP": True})
    _C.OUTPUT_DIR = "./output"
    _C.VIS_PERIOD = 0
    _C.SEED = -1  # detectron2 default: negative = not fixed
    return _C


def add_wsovod_config(cfg):
    """Keys of /root/reference/wsovod/config/defaults.py:7-96 that the hot path reads."""
    C = CfgNode
    _C = cfg
    _C.WSOVOD = C()
    _C.WSOVOD.ITER_SIZE = 1
    _C.WSOVOD.CLS_AGNOSTIC_BBOX_KNOWN = False
    _C.WSOVOD.SAMPLING = C()
    _C.WSOVOD.SAMPLING.SAMPLING_ON = False
    _C.WSOVOD.SAMPLING.IOU_THRESHOLDS = [[0.5], [0.5], [0.5], [0.5]]
    _C.WSOVOD.SAMPLING.IOU_LABELS = [[0, 1], [0, 1], [0, 1], [0, 1]]
    _C.WSOVOD.SAMPLING.BATCH_SIZE_PER_IMAGE = [4096, 4096, 4096, 4096]
    _C.WSOVOD.SAMPLING.POSITIVE_FRACTION = [1.0, 1.0, 1.0, 1.0]
    _C.WSOVOD.OBJECT_MINING = C({"WEIGHT": 1.0, "MEAN_LOSS": True})
    _C.WSOVOD.INSTANCE_REFINEMENT = C()
    _C.WSOVOD.INSTANCE_REFINEMENT.WEIGHT = 1.0
    _C.WSOVOD.INSTANCE_REFINEMENT.REFINE_NUM = 3
    _C.WSOVOD.INSTANCE_REFINEMENT.REFINE_REG = [False, False, False]
    _C.WSOVOD.INSTANCE_REFINEMENT.REFINE_MIST = False
    _C.WSOVOD.INSTANCE_REFINEMENT.CROSS_ENTROPY_WEIGHTED = True
    _C.WSOVOD.BBOX_REFINE = C({"ENABLE": False, "MODEL_TYPE": "vit_b", "MODEL_CHECKPOINT": ""})
    _C.MODEL.ROI_BOX_HEAD.DAN_DIM = [4096, 4096]
    _C.MODEL.ROI_BOX_HEAD.OPEN_VOCABULARY = C()
    _C.MODEL.ROI_BOX_HEAD.OPEN_VOCABULARY.WEIGHT_PATH_TRAIN = ""
    _C.MODEL.ROI_BOX_HEAD.OPEN_VOCABULARY.WEIGHT_PATH_TEST = ""
    _C.MODEL.ROI_BOX_HEAD.OPEN_VOCABULARY.WEIGHT_DIM = 512
    _C.MODEL.ROI_BOX_HEAD.OPEN_VOCABULARY.USE_BIAS = 0.0
    _C.MODEL.ROI_BOX_HEAD.OPEN_VOCABULARY.NORM_WEIGHT = True
    _C.MODEL.ROI_BOX_HEAD.OPEN_VOCABULARY.NORM_TEMP = 100.0
    _C.MODEL.ROI_BOX_HEAD.OPEN_VOCABULARY.DATA_AWARE = False
    _C.MODEL.ROI_BOX_HEAD.OPEN_VOCABULARY.PROTOTYPE_NUM = 5
    _C.MODEL.MRRP = C({"MRRP_ON": False, "NUM_BRANCH": 3, "BRANCH_DILATIONS": [1, 2, 3],
                       "MRRP_STAGE": "res4", "TEST_BRANCH_IDX": 1})
    _C.DATASETS.MIXED_DATASETS = C()
    _C.DATASETS.MIXED_DATASETS.NAMES = ["coco_2017_train"]
    _C.DATASETS.MIXED_DATASETS.WEIGHT_PATH_TRAINS = ["models/coco_text_embedding_single_prompt.pkl"]
    _C.DATASETS.MIXED_DATASETS.NUM_CLASSES = [80]
    _C.DATASETS.MIXED_DATASETS.PROPOSAL_FILES = [""]
    _C.DATASETS.MIXED_DATASETS.RATIOS = [1]
    _C.TEST.EVAL_TRAIN = False
    _C.VIS_TEST = False
    _C.SOLVER.OPTIMIZER = "SGD"
    _C.SOLVER.BACKBONE_MULTIPLIER = 1.0
    # hot-path extensions of this implementation (not in the reference)
    _C.MODEL.HIP = C()
    # "bf16" (bf16 MFMA, fp32 accumulate) | "fp32" (exact-fp32 MFMA) | "bf16x3" (fp32 tensors, bf16 MFMA on hi/lo-split
    # operands: fp32-grade products at a third of the bf16 rate) | "bf16x3f" (the split in the forward pass only, plain
    # bf16 backward: fp32-grade logits, bf16-grade gradients) | "parity" (the fast tolerance-meeting mode: the forward
    # split of "bf16x3f" on the activation format / kernels built for it -- DESIGN.md section 3)
    _C.MODEL.HIP.PRECISION = "bf16"
    return _C


def get_cfg():
    return add_wsovod_config(_d2_defaults())


def configurable(init_func=None, *, from_config=None):
    """detectron2.config.configurable: `Cls(cfg, *args)` -> `Cls(**Cls.from_config(cfg, *args))`."""
    assert init_func is not None and from_config is None
    assert inspect.isfunction(init_func) and init_func.__name__ == "__init__"

    @functools.wraps(init_func)
    def wrapped(self, *args, **kwargs):
        fc = type(self).from_config
        if _called_with_cfg(*args, **kwargs):
            explicit = fc(*args, **kwargs)
            init_func(self, **explicit)
        else:
            init_func(self, *args, **kwargs)

    return wrapped


def _called_with_cfg(*args, **kwargs):
    if len(args) and isinstance(args[0], CfgNode):
        return True
    # a lone `cfg=` keyword; explicit construction (a subclass forwarding from_config's dict) may carry cfg too
    if len(kwargs) == 1 and isinstance(kwargs.get("cfg"), CfgNode):
        return True
    return False


class Registry:
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def register(self, obj=None):
        if obj is None:
            def deco(func_or_class):
                self._obj_map[func_or_class.__name__] = func_or_class
                return func_or_class
            return deco
        self._obj_map[obj.__name__] = obj
        return obj

    def get(self, name):
        if name not in self._obj_map:
            raise KeyError("No object named '{}' found in '{}' registry!".format(name, self._name))
        return self._obj_map[name]

    def __contains__(self, name):
        return name in self._obj_map


BACKBONE_REGISTRY = Registry("BACKBONE")
ROI_HEADS_REGISTRY = Registry("ROI_HEADS")
ROI_BOX_HEAD_REGISTRY = Registry("ROI_BOX_HEAD")
META_ARCH_REGISTRY = Registry("META_ARCH")
PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")
