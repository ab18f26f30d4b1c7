"""On-disk input formats of the path, host side.

* SAM proposal pickle -> per-image `Instances{proposal_boxes, objectness_logits}`: mirrors
  /root/reference/wsovod/data/build.py:112-173 (`load_proposals_into_dataset`: D1 key aliases
  `indexes`/`scores`, join on str(image_id), sort by descending score) and
  /root/reference/wsovod/data/detection_utils.py:206-265 (`unique_boxes`, `transform_proposals`: geometric
  transform, clip, de-duplicate on the rounded coordinates, drop boxes with a side <= min_box_size, top-k).
* class text-embedding pickle: torch fp32 (K, D), written by tools/generate_class_text_embedding_cuda.py:117-126
  and read by open_vocabulary_classifier.py:52-56 (np.load(..., allow_pickle=True)).
* detectron2 backbone pickle ({"model": {name: ndarray}, "matching_heuristics": True}) -> state dict by
  longest-suffix matching, what DetectionCheckpointer (un-vendored) does for `MODEL.WEIGHTS: *_d2.pkl`.

Index work is bit-exact against fixtures generated from the reference's own functions
(tests/golden/g13_proposal_formats.npz).
"""
import pickle
from pathlib import Path

import numpy as np
import torch

from ..structures import Boxes, Instances

__all__ = ["load_proposals_into_dataset", "unique_boxes", "transform_proposals", "load_class_embeddings",
           "load_d2_pickle_into", "ResizeTransform", "HFlipTransform", "TransformList", "XYXY_ABS", "XYWH_ABS"]

XYXY_ABS, XYWH_ABS = 0, 1  # detectron2.structures.BoxMode values


def _to_xyxy(boxes, mode):
    mode = int(getattr(mode, "value", mode))
    if mode == XYXY_ABS:
        return boxes
    if mode == XYWH_ABS:
        out = np.array(boxes, dtype=boxes.dtype, copy=True)
        out[:, 2] += out[:, 0]
        out[:, 3] += out[:, 1]
        return out
    raise NotImplementedError(f"proposal bbox_mode {mode}: only XYXY_ABS / XYWH_ABS proposal files exist for WSOVOD")


def load_proposals_into_dataset(dataset_dicts, proposal_file):
    """Attach `proposal_boxes` / `proposal_objectness_logits` / `proposal_bbox_mode` to every record."""
    if proposal_file == "":
        return dataset_dicts
    if Path(proposal_file).is_dir():  # one pickle per image, read lazily by the mapper
        for record in dataset_dicts:
            record["proposal_file"] = proposal_file + "/" + str(record["image_id"]) + ".pkl"
        return dataset_dicts
    with open(proposal_file, "rb") as f:
        proposals = pickle.load(f, encoding="latin1")
    for old, new in (("indexes", "ids"), ("scores", "objectness_logits")):
        if old in proposals:
            proposals[new] = proposals.pop(old)
    wanted = {str(record["image_id"]) for record in dataset_dicts}
    id_to_index = {str(i): n for n, i in enumerate(proposals["ids"]) if str(i) in wanted}
    bbox_mode = proposals.get("bbox_mode", XYXY_ABS)
    for record in dataset_dicts:
        n = id_to_index[str(record["image_id"])]
        boxes, logits = proposals["boxes"][n], proposals["objectness_logits"][n]
        inds = logits.argsort()[::-1]
        record["proposal_boxes"] = boxes[inds]
        record["proposal_objectness_logits"] = logits[inds]
        record["proposal_bbox_mode"] = bbox_mode
    return dataset_dicts


def unique_boxes(boxes, scale=1.0):
    """Indices (ascending) of the first occurrence of every distinct rounded box."""
    arr = boxes.tensor.data.numpy() if isinstance(boxes, Boxes) else np.asarray(boxes)
    v = np.array([1, 1e3, 1e6, 1e9])
    hashes = np.round(arr * scale).dot(v).astype(int)
    _, index = np.unique(hashes, return_index=True)
    return np.sort(index)


class ResizeTransform:
    """detectron2 ResizeTransform on boxes: scale x by new_w / w and y by new_h / h."""

    def __init__(self, h, w, new_h, new_w):
        self.h, self.w, self.new_h, self.new_w = h, w, new_h, new_w

    def apply_coords(self, coords):
        coords = coords.copy()
        coords[:, 0] = coords[:, 0] * (self.new_w * 1.0 / self.w)
        coords[:, 1] = coords[:, 1] * (self.new_h * 1.0 / self.h)
        return coords


class HFlipTransform:
    def __init__(self, width):
        self.width = width

    def apply_coords(self, coords):
        coords = coords.copy()
        coords[:, 0] = self.width - coords[:, 0]
        return coords


class TransformList:
    """detectron2 TransformList.apply_box: transform the four corners, take their bounding box."""

    def __init__(self, transforms=()):
        self.transforms = list(transforms)

    def apply_coords(self, coords):
        for t in self.transforms:
            coords = t.apply_coords(coords)
        return coords

    def apply_box(self, box):
        idxs = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
        coords = np.asarray(box).reshape(-1, 4)[:, idxs].reshape(-1, 2)
        coords = self.apply_coords(coords).reshape((-1, 4, 2))
        minxy, maxxy = coords.min(axis=1), coords.max(axis=1)
        return np.concatenate((minxy, maxxy), axis=1)


def transform_proposals(dataset_dict, image_shape, transforms, *, proposal_topk, min_box_size=0):
    """In place: the three proposal_* keys are replaced by `proposals` (Instances)."""
    if "proposal_boxes" not in dataset_dict:
        return
    boxes = transforms.apply_box(_to_xyxy(dataset_dict.pop("proposal_boxes"), dataset_dict.pop("proposal_bbox_mode")))
    boxes = Boxes(boxes)
    objectness_logits = torch.as_tensor(dataset_dict.pop("proposal_objectness_logits").astype("float32"))
    boxes.clip(image_shape)
    keep = unique_boxes(boxes)
    boxes, objectness_logits = boxes[keep], objectness_logits[keep]
    keep = boxes.nonempty(threshold=min_box_size)
    boxes, objectness_logits = boxes[keep], objectness_logits[keep]
    proposals = Instances(image_shape)
    proposals.proposal_boxes = boxes[:proposal_topk]
    proposals.objectness_logits = objectness_logits[:proposal_topk]
    dataset_dict["proposals"] = proposals


def load_class_embeddings(path, device=None):
    """(K, D) fp32 CLIP text embeddings of a class vocabulary."""
    w = torch.as_tensor(np.load(path, encoding="bytes", allow_pickle=True)).to(torch.float32).contiguous()
    assert w.dim() == 2, w.shape
    return w.to(device) if device is not None else w


def load_d2_pickle_into(module, path, prefix=""):
    """Load a detectron2-format pickle (`MODEL.WEIGHTS: models/DRN-WSOD/resnet18_ws_model_120_d2.pkl`) into
    `module`: every checkpoint tensor goes to the model key that has it as its longest suffix
    (detectron2's align_and_update_state_dicts, un-vendored).  Returns (loaded model keys, unused checkpoint keys)."""
    with open(path, "rb") as f:
        data = pickle.load(f, encoding="latin1")
    ckpt = data["model"] if "model" in data else data
    sd = module.state_dict()
    model_keys = sorted(sd.keys())
    loaded, unused = {}, []
    for ck, val in ckpt.items():
        name = prefix + ck
        cands = [mk for mk in model_keys if mk == name or mk.endswith("." + name)]
        if not cands:
            unused.append(ck)
            continue
        mk = max(cands, key=len) if name not in cands else name
        t = torch.as_tensor(np.asarray(val))
        if tuple(t.shape) != tuple(sd[mk].shape):
            raise ValueError(f"{ck} -> {mk}: shape {tuple(t.shape)} != {tuple(sd[mk].shape)}")
        loaded[mk] = t.to(sd[mk].dtype)
    module.load_state_dict(loaded, strict=False)
    return sorted(loaded), unused
