"""Synthetic inputs of the benchmark / parity configs: uint8 BGR images, random
proposal boxes with descending objectness in (0,1], two image-level labels, N(0,1) class
embeddings.  Format = what the reference's DatasetMapper emits
(/root/reference/wsovod/data/dataset_mapper.py:144-191): dicts with `image`, `instances`,
`proposals`, `height`, `width`."""
import torch

from ..structures import Boxes, Instances


def make_class_embeddings(K, D, seed=7):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(K, D, generator=g)


def make_batch(n_images, R, K, H=600, W=800, seed=1234):
    g = torch.Generator().manual_seed(seed)
    batch = []
    for _ in range(n_images):
        image = torch.randint(0, 256, (3, H, W), generator=g, dtype=torch.uint8)
        x0 = torch.rand(R, generator=g) * (W - 17)
        y0 = torch.rand(R, generator=g) * (H - 17)
        w = 16 + torch.rand(R, generator=g) * (torch.clamp(W - x0, max=400.0) - 16)
        h = 16 + torch.rand(R, generator=g) * (torch.clamp(H - y0, max=300.0) - 16)
        boxes = torch.stack([x0, y0, x0 + w, y0 + h], dim=1)
        logits = (1.0 - torch.rand(R, generator=g)).sort(descending=True).values  # (0,1]
        props = Instances((H, W), proposal_boxes=Boxes(boxes), objectness_logits=logits)
        cls = torch.randperm(K, generator=g)[: min(2, K)].to(torch.int64)
        gt_idx = torch.randint(0, R, (len(cls),), generator=g)
        inst = Instances((H, W), gt_boxes=Boxes(boxes[gt_idx]), gt_classes=cls)
        batch.append({"image": image, "instances": inst, "proposals": props, "height": H, "width": W})
    return batch
