"""Open-vocabulary classifier + data-aware feature head on HIP kernels.

Mirrors /root/reference/wsovod/modeling/class_heads/open_vocabulary_classifier.py:14-105 and
data_aware_features_head.py:19-131 (same constructor arguments, parameter/buffer names and
forward signatures).
"""
import logging
from math import fabs
from typing import List

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from ..config import configurable
from ..layers import functions as Fn
from ..layers import hip_ops as H
from ..structures import ShapeSpec

logger = logging.getLogger(__name__)


class OpenVocabularyClassifier(nn.Module):
    """x -> ReLU(L2(ReLU(L1 x))) -> T * x/||x|| -> @ L2-normalised class text embeddings (+ zero
    background column).  Projection = two fused Linear+ReLU launches; the cosine-similarity GEMM
    runs on MFMA with the T/||x|| row scale folded into its epilogue."""

    @configurable
    def __init__(self, input_shape: ShapeSpec, *, num_classes: int, weight_path: str, weight_dim: int = 512,
                 use_bias: float = 0.0, norm_weight: bool = True, norm_temperature: float = 50.0):
        super().__init__()
        if isinstance(input_shape, int):
            input_shape = ShapeSpec(channels=input_shape)
        input_size = input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
        self.norm_weight = norm_weight
        self.weight_dim = weight_dim
        self.norm_temperature = norm_temperature
        self.use_bias = fabs(use_bias) > 1e-9
        if self.use_bias:
            self.cls_bias = nn.Parameter(torch.ones(1) * use_bias)
        self.projection = nn.Sequential(nn.Linear(input_size, 1024), nn.ReLU(), nn.Linear(1024, weight_dim), nn.ReLU())
        if weight_path == "rand":
            class_weight = torch.randn((weight_dim, num_classes))
            nn.init.normal_(class_weight, std=0.01)
        else:
            logger.info("Loading " + weight_path)
            class_weight = (torch.as_tensor(np.load(weight_path, encoding="bytes", allow_pickle=True))
                            .to(torch.float32).permute(1, 0).contiguous())  # D x C
        if self.norm_weight:
            class_weight = F.normalize(class_weight, p=2, dim=0)
        # "rand" makes it a Parameter in the reference (a debugging mode); the HIP path treats the
        # embeddings as constants either way and refuses to silently drop their gradient.
        if weight_path == "rand":
            self.class_weight = nn.Parameter(class_weight, requires_grad=False)
        else:
            self.register_buffer("class_weight", class_weight)
        self._cache = None

    @classmethod
    def from_config(cls, cfg, input_shape, weight_path=None, use_bias=None, norm_weight=None, norm_temperature=None):
        ov = cfg.MODEL.ROI_BOX_HEAD.OPEN_VOCABULARY
        return {"input_shape": input_shape, "num_classes": cfg.MODEL.ROI_HEADS.NUM_CLASSES,
                "weight_path": weight_path if weight_path is not None else ov.WEIGHT_PATH_TRAIN,
                "weight_dim": ov.WEIGHT_DIM, "use_bias": use_bias if use_bias is not None else ov.USE_BIAS,
                "norm_weight": norm_weight if norm_weight is not None else ov.NORM_WEIGHT,
                "norm_temperature": norm_temperature if norm_temperature is not None else ov.NORM_TEMP}

    def _class_matrix(self, classifier, append_background, dtype):
        """(Wn (K1,D), WnT (D,K1 padded to 8)) in the compute dtype, L2-normalised, bg row zero."""
        src = classifier if classifier is not None else self.class_weight
        key = (src.data_ptr(), src._version, classifier is not None, append_background, dtype)
        hit = self._cache.get(key) if self._cache else None
        if hit is not None:
            return hit
        with torch.no_grad():
            rows = (classifier if classifier is not None else self.class_weight.t()).to(torch.float32).contiguous()
            C, D = rows.shape  # (C', D)
            K1 = C + (1 if append_background else 0)
            wn = torch.zeros((K1, D), dtype=dtype, device=rows.device)
            if classifier is not None and self.norm_weight:
                rs = H.row_l2norm_scale(rows, 1.0)  # F.normalize(classifier.T, dim=0)
            else:
                rs = torch.ones((C,), dtype=torch.float32, device=rows.device)
            H.scale_rows(rows, rs, wn)
            wnT = H.transpose_cast(wn, dtype, ld_dst=(K1 + 7) // 8 * 8)  # (D, K1p)
            # transpose_cast(src (K1,D)) -> (D, ld): rows of wnT are embedding dims
        if self._cache is None or len(self._cache) >= 8:  # one entry per dataset in mixed-dataset mode
            self._cache = {}
        self._cache[key] = (wn, wnT)
        return wn, wnT

    def forward(self, x, classifier=None, append_background=False, hidden=None):
        """x: (B, D_in) in the compute dtype; classifier: optional (C', D) raw embeddings; hidden: optional
        precomputed relu(projection[0](x)) (the ROI heads batch that layer with the other heads on x)."""
        l1, l2 = self.projection[0], self.projection[2]
        x2 = H.x3_active() == "x2"  # "parity": x / hidden are bf16x2, the second layer's output z is real fp32
        x = hidden if hidden is not None else Fn.linear(x, l1.weight, l1.bias, relu=True, out_dtype=H.X2 if x2 else None)
        x = Fn.linear(x, l2.weight, l2.bias, relu=True)
        wn, wnT = self._class_matrix(classifier, append_background, x.dtype)
        bias_vec = self.cls_bias.expand(wn.size(0)).contiguous() if self.use_bias else None
        return Fn.cosine_logits(x, wn, wnT, self.norm_temperature, self.norm_weight, bias_vec)


class DataAwareFeaturesHead(nn.Module):
    """GAP(res5) -> Linear+ReLU -> Linear+tanh -> @ prototype embedding (5 x F); one row per image.
    `forward` returns the per-image features; the per-proposal repeat of the reference
    (data_aware_features_head.py:117-121) is folded into the broadcast add downstream."""

    @configurable
    def __init__(self, input_shape, *, datasets_prototype_num: int = 5, features_dim: int = 512,
                 cls_in_features: List[str], mrrp_on: bool = False, mrrp_num_branch: int = 3):
        super().__init__()
        if mrrp_on:
            raise NotImplementedError("MRRP is off in every WSR config (out of hot-path scope)")
        self.in_features = self.cls_in_features = cls_in_features
        self.features_dim = features_dim
        in_channels = [input_shape[f].channels for f in self.in_features]
        assert len(set(in_channels)) == 1, in_channels
        in_channels = in_channels[0]
        self.datasets_prototype_num = datasets_prototype_num
        self.datasets_feat = nn.Embedding(self.datasets_prototype_num, self.features_dim)
        self.linear1 = nn.Linear(in_channels, in_channels // 16)
        self.linear_relu1 = nn.ReLU(inplace=True)
        self.linear2 = nn.Linear(in_channels // 16, self.datasets_prototype_num)
        self.linear_tanh2 = nn.Tanh()
        for layer in (self.linear1, self.linear2):
            nn.init.uniform_(layer.weight, -0.01, 0.01)
            torch.nn.init.constant_(layer.bias, 0)

    @classmethod
    def from_config(cls, cfg, input_shape):
        return {"cls_in_features": cfg.MODEL.ROI_HEADS.IN_FEATURES, "input_shape": input_shape,
                "datasets_prototype_num": cfg.MODEL.ROI_BOX_HEAD.OPEN_VOCABULARY.PROTOTYPE_NUM,
                "features_dim": cfg.MODEL.ROI_BOX_HEAD.DAN_DIM[-1], "mrrp_on": cfg.MODEL.MRRP.MRRP_ON,
                "mrrp_num_branch": cfg.MODEL.MRRP.NUM_BRANCH}

    def pooled_stats(self, features):
        """Parameter-free part: GAP of every input feature map -> list of (N,C) fp32."""
        gaps = []
        for f in [features[k] for k in self.cls_in_features]:
            if f.is_contiguous(memory_format=torch.channels_last):
                nhwc = f.permute(0, 2, 3, 1)
            else:
                nhwc = f.permute(0, 2, 3, 1).contiguous()
            gaps.append(Fn.global_avgpool_nhwc(nhwc))  # (differentiable when a backbone stage is trainable)
        return gaps

    def from_stats(self, gaps):
        outs = [Fn.data_aware_features(g, self.linear1.weight, self.linear1.bias, self.linear2.weight,
                                       self.linear2.bias, self.datasets_feat.weight) for g in gaps]
        return outs[0] if len(outs) == 1 else torch.stack(outs).mean(0)

    def forward_per_image(self, features):
        return self.from_stats(self.pooled_stats(features))

    def forward(self, features, proposals):
        """Reference signature: one feature row per proposal (repeat materialised)."""
        per_image = self.forward_per_image(features)
        return torch.cat([per_image[i].repeat(len(proposals[i]), 1) for i in range(len(proposals))])
