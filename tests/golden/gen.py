"""Seeded inputs/parameters shared by tests/golden/make_golden.py (which feeds them to the
REFERENCE) and by the tests (which feed them to the oracle and to the HIP path).  Everything is a
pure function of (key, seed) on torch's CPU generator, so no weights need to be committed."""
import hashlib
import math

import torch

PIXEL_MEAN = [102.9801, 115.9465, 122.7717]
PIXEL_STD = [57.375, 57.12, 58.395]  # exercises the /std path; keeps activations O(1)


def _gen(key, seed):
    h = int(hashlib.sha256(f"{seed}:{key}".encode()).hexdigest()[:12], 16)
    return torch.Generator().manual_seed(h)


def seeded_state(shapes, seed=0):
    """shapes: {state_dict key: shape} of a model built from the reference key names.  Initialisers follow the
    reference's (c2_msra_fill, N(0,.005)/0.1, xavier, N(0,.001), nn.Linear default, U(-.01,.01), N(0,1)) but with
    non-trivial FrozenBN statistics so that BN folding is exercised."""
    sd = {}
    for k, shp in shapes.items():
        g = _gen(k, seed)
        shp = tuple(shp)
        if k.endswith("norm.weight") or k.endswith("norm.running_var"):
            t = 0.5 + torch.rand(shp, generator=g)
        elif k.endswith("norm.bias") or k.endswith("norm.running_mean"):
            t = 0.1 * torch.randn(shp, generator=g)
        elif k.startswith("proposal_generator."):
            t = torch.randn(shp, generator=g) * (0.02 if k.endswith("weight") else 0.05)
        elif k.startswith("backbone.") and k.endswith("weight"):
            fan_out = shp[0] * shp[2] * shp[3]
            t = torch.randn(shp, generator=g) * math.sqrt(2.0 / fan_out)
        elif ".box_head.fc" in k:
            t = torch.randn(shp, generator=g) * 0.005 if k.endswith("weight") else torch.full(shp, 0.1)
        elif ".object_miner." in k or ".object_miners." in k:
            if k.endswith("weight"):
                a = math.sqrt(6.0 / (shp[0] + shp[1]))
                t = (torch.rand(shp, generator=g) * 2 - 1) * a
            else:
                t = 0.01 * torch.randn(shp, generator=g)
        elif k.endswith("bbox_pred.weight"):
            t = torch.randn(shp, generator=g) * 0.001
        elif k.endswith("bbox_pred.bias"):
            t = 0.01 * torch.randn(shp, generator=g)
        elif ".projection." in k:
            fan_in = shp[1] if len(shp) == 2 else None
            bound = 1.0 / math.sqrt(fan_in) if fan_in else 0.02
            t = (torch.rand(shp, generator=g) * 2 - 1) * bound
        elif k.endswith("class_weight"):
            t = torch.nn.functional.normalize(torch.randn(shp, generator=g), p=2, dim=0)  # (D,K), unit columns
        elif k.startswith("data_aware_head.datasets_feat"):
            t = torch.randn(shp, generator=g)
        elif k.startswith("data_aware_head."):
            t = (torch.rand(shp, generator=g) * 2 - 1) * (0.3 if k.endswith("weight") else 0.05)
        elif k.endswith("cls_bias"):
            t = torch.full(shp, 0.5)
        else:
            raise KeyError(f"no initialiser for {k} {shp}")
        sd[k] = t.float()
    return sd


def mixed_seeded_state(shapes, seed):
    """seeded_state for the mixed-dataset model: miners shared per dataset family appear under every index of
    `object_miners` in the state dict and must hold the same tensors."""
    sd = seeded_state(shapes, seed)
    for k in list(sd):
        if k.startswith("roi_heads.object_miners.1."):
            sd[k] = sd[k.replace("object_miners.1.", "object_miners.0.")]
    return sd


def sampler_dataset_dicts(sizes=(7, 5, 30), num_categories=6, seed=3):
    """Concatenated dataset dicts (id order) with random category sets, as the multi-dataset sampler sees them."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for ds, n in enumerate(sizes):
        for i in range(n):
            k = int(torch.randint(1, 4, (1,), generator=g))
            cats = torch.randint(0, num_categories, (k,), generator=g).tolist()
            out.append({"dataset_id": ds, "annotations": [{"category_id": c} for c in cats]})
    return out


def first_k_subsample(labels, num_samples, positive_fraction, bg_label):
    """Deterministic stand-in for detectron2's subsample_labels (random there): the FIRST num_pos positives and
    the FIRST num_neg negatives in index order.  Injected on both sides of the RPN parity tests."""
    positive = ((labels != -1) & (labels != bg_label)).nonzero().flatten()
    negative = (labels == bg_label).nonzero().flatten()
    num_pos = min(positive.numel(), int(num_samples * positive_fraction))
    num_neg = min(negative.numel(), num_samples - num_pos)
    return positive[:num_pos], negative[:num_neg]


def proposal_pickle(n_images=4, seed=5):
    """A D1-style proposal file ({indexes, scores, boxes}) + dataset records that reference a shuffled subset of
    it; boxes include exact duplicates, near-duplicates that collide after rounding, and boxes below the size cut."""
    import numpy as np

    rng = np.random.RandomState(seed)
    ids = [str(2007000 + i) for i in range(n_images + 2)]
    boxes, scores = [], []
    for i in range(len(ids)):
        n = 120 + 7 * i
        h, w = 200 + 10 * i, 300 + 20 * i
        xy = rng.rand(n, 2) * [w - 40, h - 40]
        wh = 2 + rng.rand(n, 2) * [w / 2, h / 2]
        b = np.concatenate([xy, np.minimum(xy + wh, [w, h])], axis=1).astype(np.float32)
        b[10:20] = b[0:10]                      # exact duplicates
        b[20:25] = b[0:5] + 0.2                 # collide after rounding
        b[25:30, 2:] = b[25:30, :2] + 3.0       # tiny
        boxes.append(b)
        scores.append(rng.rand(n).astype(np.float32))
    pk = {"indexes": ids, "boxes": boxes, "scores": scores}
    order = [3, 0, 4, 1][:n_images]
    recs = [{"image_id": int(ids[j]) if j % 2 else ids[j], "height": 200 + 10 * j, "width": 300 + 20 * j} for j in order]
    return pk, recs


def seeded_batch(n_images, R, K, H, W, seed=0, edge_cases=True):
    """Plain-tensor batch (oracle.wsovod_ref.train_forward format)."""
    batch = []
    for i in range(n_images):
        g = _gen(f"img{i}", seed)
        h, w = (H, W) if i % 2 == 0 else (H - 8 * (i % 3), W - 16)  # ragged sizes -> exercises padding
        image = torch.randint(0, 256, (3, h, w), generator=g, dtype=torch.uint8)
        r = R if i != n_images - 1 else max(R - 7, 1)  # ragged proposal counts
        x0 = torch.rand(r, generator=g) * (w - 17)
        y0 = torch.rand(r, generator=g) * (h - 17)
        bw = 16 + torch.rand(r, generator=g) * (torch.clamp(w - x0, max=400.0) - 16)
        bh = 16 + torch.rand(r, generator=g) * (torch.clamp(h - y0, max=300.0) - 16)
        boxes = torch.stack([x0, y0, x0 + bw, y0 + bh], dim=1)
        if edge_cases and r >= 4:
            boxes[0] = torch.tensor([3.0, 3.0, 6.0, 7.0])  # area 12 <= 20: filtered from pseudo-GT mining
            boxes[1] = torch.tensor([0.0, 0.0, float(w), float(h)])  # whole image
        obj = (1.0 - torch.rand(r, generator=g)).sort(descending=True).values
        ncls = 1 if i % 3 == 2 else 2
        gt_classes = torch.randperm(K, generator=g)[: min(ncls, K)].to(torch.int64)
        gt_classes = torch.cat([gt_classes, gt_classes[:1]])  # a duplicate: exercises torch.unique
        batch.append(dict(image=image, boxes=boxes, objectness=obj, gt_classes=gt_classes))
    return batch


def strided_sample(t, n=4096):
    """Deterministic sub-sample of a big tensor for compact fixtures."""
    flat = t.detach().reshape(-1)
    if flat.numel() <= n:
        return flat.clone()
    idx = (torch.arange(n, dtype=torch.int64) * (flat.numel() - 1)) // (n - 1)
    return flat[idx].clone()


def edge_batch(K=20, seed=7):
    """The batch of G18's whole step (shared with the tests): 3 ragged images ~160x208, 24 proposals each; EVERY proposal
    of image 1 has area <= 20 px^2 (the pseudo-GT miner filters them all: roi_heads.py:1090-1111 -> the empty-result
    fallbacks :1181-1207), image 2's best-scoring candidates include filtered boxes next to normal ones."""
    batch = seeded_batch(3, 24, K, 160, 208, seed=seed)
    g = torch.Generator().manual_seed(seed + 1000)
    b = batch[1]
    r = len(b["boxes"])
    h, w = b["image"].shape[-2:]
    x0 = torch.rand(r, generator=g) * (w - 8)
    y0 = torch.rand(r, generator=g) * (h - 8)
    wh = 2.0 + 2.4 * torch.rand(r, 2, generator=g)  # area <= 19.4
    b["boxes"] = torch.stack([x0, y0, x0 + wh[:, 0], y0 + wh[:, 1]], dim=1)
    b2 = batch[2]
    b2["boxes"][2::3, 2:] = b2["boxes"][2::3, :2] + torch.tensor([4.0, 4.5])  # a third of image 2's boxes: area 18
    return batch


def edge_features(tag, rows, dim=4096):
    """Box-feature matrix of G18's direct head calls (a function of the tag only: nothing to commit)."""
    return torch.randn(rows, dim, generator=_gen("edge_features_" + tag, 0)) * 0.5


def edge_embeddings(K=80, D=768):
    """(class text embeddings (K,D) as the weight file holds them, a per-call classifier (17,D)) of G18's (d)."""
    return torch.randn(K, D, generator=_gen("edge_emb", 0)), torch.randn(17, D, generator=_gen("edge_clsf", 0))
