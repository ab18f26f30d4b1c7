import os

import numpy as np
import torch

from tests.golden import gen
from wsovod_amd.structures import Boxes, Instances

G = os.path.join(os.path.dirname(__file__), "golden")


def load_golden(name):
    d = np.load(os.path.join(G, name + ".npz"), allow_pickle=False)
    return {k: torch.from_numpy(d[k]) if d[k].dtype.kind in "fiub" else d[k] for k in d.files}


def golden_shapes():
    d = np.load(os.path.join(G, "shapes_r18_k20.npz"))
    return {str(k): eval(str(s)) for k, s in zip(d["keys"], d["shapes"])}


def seeded_sd(seed=1):
    return gen.seeded_state(golden_shapes(), seed)


def to_inputs(batch):
    """oracle plain-tensor batch -> DatasetMapper-format dicts (reference: data/dataset_mapper.py:144-191)."""
    out = []
    for b in batch:
        h, w = b["image"].shape[-2:]
        props = Instances((h, w), proposal_boxes=Boxes(b["boxes"].clone()), objectness_logits=b["objectness"].clone())
        nb = len(b["gt_classes"])
        inst = Instances((h, w), gt_boxes=Boxes(b["boxes"][:nb].clone()), gt_classes=b["gt_classes"].clone())
        out.append({"image": b["image"], "instances": inst, "proposals": props, "height": h, "width": w})
        if "dataset_id" in b:  # mixed-dataset batches carry their source (reference: data/build_multi_dataset.py:270-272)
            out[-1]["dataset_id"] = b["dataset_id"]
    return out


def build_seeded_hip_model(precision, seed=1, pooler="ROIPool", dropout=False):
    from wsovod_amd.testing import build_hot_path_model

    cfg, model = build_hot_path_model(seed=0, precision=precision, pooler=pooler, device="cuda:0",
                                      calibrate_synthetic=False)
    cfg.MODEL.PIXEL_STD = list(gen.PIXEL_STD)
    model._std = [float(v) for v in gen.PIXEL_STD]
    sd = seeded_sd(seed)
    model.load_state_dict(sd, strict=True)
    model.train()
    if not dropout:
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.eval()
    return cfg, model, sd
