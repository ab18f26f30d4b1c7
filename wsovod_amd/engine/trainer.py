"""Training-step plumbing of the hot path: optimizer, DDP wrap, run_step.

Mirrors /root/reference/wsovod/engine/trainer.py:37-84 (`run_step`: forward, sum of the loss dict,
backward, optimizer step every ITER_SIZE) and engine/defaults.py:135-153,274-318
(`wrap_model_with_ddp`, `build_optimizer`: one param group per tensor, SGD momentum).  The
parameter update itself is the fused HIP kernel `wsovod_sgd_momentum`; gradient exchange is
torch DistributedDataParallel over RCCL ("nccl" backend on ROCm) -- the only collective on the path.
"""
import gc as _gc
import os
import warnings
import weakref

import torch
import torch.distributed as dist

from ..layers import hip_ops as H


class HipSGD(torch.optim.Optimizer):
    """torch.optim.SGD semantics (momentum, weight decay, dampening 0) on wsovod_sgd_momentum."""

    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0, clip=None):
        """clip: None or (kind, value) -- "full_model" (the reference's FullModelGradientClippingOptimizer,
        engine/defaults.py:292-318: clip_grad_norm_ over every parameter at once), "norm" / "value" (detectron2's
        per-parameter clip_grad_norm_ / clip_grad_value_, maybe_add_gradient_clipping).  The norm is taken of the gradient
        the optimizer sees (after the data-parallel average); it stays on the device."""
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self.grad_scale = 1.0  # 1/world_size when gradients arrive as a SUM over ranks
        if clip is not None and (clip[0] not in ("full_model", "norm", "value") or not clip[1] > 0.0):
            raise ValueError(f"HipSGD: clip must be (full_model | norm | value, positive value), got {clip!r}")
        self.clip = clip
        self.last_clip_coef = None  # device tensor of the coefficients applied by the last step (norm kinds; tests, logging)
        self.lr_device = None  # {param: 1-element fp32 device tensor}: set while a step graph is captured (see _StepGraph)

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        by_momentum = {}
        for group in self.param_groups:
            for p in group["params"]:
                wire = getattr(p, "_wire_grad", None)  # bf16 slice of the reduced wire buffer (HotPathTrainer)
                if p.grad is None and wire is None:
                    continue
                state = self.state[p]
                if "momentum_buffer" not in state:
                    state["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                g = wire if wire is not None else (p.grad if p.grad.is_contiguous() else p.grad.contiguous())
                sh = getattr(p, "_hip_shadow", None)  # bf16 copy used by the MFMA kernels: refreshed in the same pass
                shadow = sh[0] if (sh is not None and sh[1] == p._version and sh[0].dtype == torch.bfloat16) else None
                xe = getattr(p, "_x2_enc", None)  # "parity" precision: the bf16x2 operand (hip_ops.x2_cached) instead
                if xe is not None and xe[0] == (p._version, p.data_ptr(), None) and p.numel() % 32 == 0:
                    shadow = xe[1]
                shadow = _mx_shadow(p) or shadow  # "parity_mx": the f16mx operand (hip_ops.mx_cached, one scale per tensor)
                lr = group["lr"] if self.lr_device is None else self.lr_device[p]
                by_momentum.setdefault(group["momentum"], []).append(
                    (p.data, g, state["momentum_buffer"], shadow, lr, group["weight_decay"],
                     getattr(p, "_used_flag", None), p))
        if self.clip is not None and self.clip[0] == "full_model" and len(by_momentum) > 1:
            raise NotImplementedError("HipSGD: full-model clipping needs one momentum value for all parameter groups")
        for mu, entries in by_momentum.items():  # every tensor of the model in one launch
            self.last_clip_coef = H.sgd_momentum_multi([e[:7] for e in entries], mu, grad_scale=self.grad_scale,
                                                       clip=self.clip)
            for e in entries:
                # the kernel wrote through raw pointers: advance the version counter so that caches keyed on it (folded
                # conv weights, class matrices) are rebuilt, and re-stamp the bf16 shadow the kernel refreshed itself
                p, shadow = e[7], e[3]
                torch.autograd.graph.increment_version(p)
                _restamp_shadow(p, shadow)


def _mx_shadow(p):
    """(f16mx carrier, per-tensor E8M0 byte) of a parameter whose CURRENT f16mx operand was encoded with one scale for the
    tensor (hip_ops.mx_cached(tensor_scale=True): the FC weights under "parity_mx"), else None: the update kernels re-encode
    it element-wise in their pass."""
    me = getattr(p, "_mx_enc", None)
    if me is None or len(me) < 3 or not me[2] or me[0] != (p._version, p.data_ptr(), None) or p.numel() % 32:
        return None
    byte = getattr(p, "_mx_scale", None)
    return (me[1][0], byte) if byte is not None else None


def _restamp_shadow(p, shadow):
    """After an update kernel refreshed `shadow` through raw pointers: the operand cache of the parameter carries its new
    version (a stale second format, if the parameter has one, stays behind and is re-encoded at its next use)."""
    if isinstance(shadow, tuple):
        p._mx_enc = ((p._version, p.data_ptr(), None), p._mx_enc[1], True)
    elif shadow is not None and shadow.dtype == torch.float32:
        p._x2_enc = ((p._version, p.data_ptr(), None), shadow)
    elif shadow is not None:
        p._hip_shadow = (shadow, p._version)


class _FusedUpdate:
    """The optimizer step of ONE large weight applied from inside its weight-gradient kernel (wsovod_gemm_tn_sgd; installed
    by HotPathTrainer on the weight as `_fused_update`, called by layers/functions.py:_Linear.backward).  At the reference's
    own batch (1 - 2 images per GPU) the step is dominated by the bytes of fc1: its gradient (411 MB fp32 for WSR_18) was
    written by one kernel and read back by the optimizer's; fused, an element costs 20 bytes instead of 32.  The arithmetic
    is HipSGD's (same formula, same order); what changes is WHEN this tensor's update lands: inside backward instead of
    with the other tensors -- nothing reads the weight in between (the input gradient of the layer is taken first).
    Only without gradient exchange, accumulation or clipping, and only up to `max_rows` reduction rows: the fused kernel
    keeps a partly filled last round of tiles whole, which costs more than the saved bytes at large batches."""

    def __init__(self, optimizer, param, max_rows):
        self.opt = weakref.proxy(optimizer)
        self.param = weakref.ref(param)
        self.max_rows = int(max_rows)
        self.group = next(g for g in optimizer.param_groups if any(q is param for q in g["params"]))
        self.calls = 0
        self.armed = False  # only inside the trainer's OWN backward: a caller's loss.backward() on the model gets gradients

    def wants(self, rows):
        p = self.param()
        return self.armed and p is not None and rows <= self.max_rows and p.grad is None \
            and getattr(p, "_wire_grad", None) is None \
            and self.opt.clip is None and getattr(p, "_used_flag", None) is None

    def __call__(self, dA, x, q_x2):
        p, opt, g = self.param(), self.opt, self.group
        state = opt.state[p]
        if "momentum_buffer" not in state:
            state["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        sh = getattr(p, "_hip_shadow", None)
        shadow = sh[0] if (sh is not None and sh[1] == p._version and sh[0].dtype == torch.bfloat16) else None
        xe = getattr(p, "_x2_enc", None)
        if xe is not None and xe[0] == (p._version, p.data_ptr(), None) and p.numel() % 32 == 0:
            shadow = xe[1]
        shadow = _mx_shadow(p) or shadow
        lr = g["lr"] if opt.lr_device is None else opt.lr_device[p]
        H.gemm_tn_sgd(dA, x, p.data, state["momentum_buffer"], shadow, lr, g["weight_decay"], g["momentum"],
                      grad_scale=opt.grad_scale, q_x2=q_x2)
        torch.autograd.graph.increment_version(p)  # (as HipSGD.step: caches keyed on the version are rebuilt ...
        _restamp_shadow(p, shadow)                 # ... and the refreshed operand copy is re-stamped)
        self.calls += 1
        return True


def build_optimizer(cfg, model):
    """engine/defaults.py:274-318: every trainable tensor is its own group with BASE_LR / WEIGHT_DECAY."""
    params, memo = [], set()
    for key, value in model.named_parameters(recurse=True):
        if not value.requires_grad or value in memo:
            continue
        memo.add(value)
        lr = cfg.SOLVER.BASE_LR * (cfg.SOLVER.BACKBONE_MULTIPLIER if "backbone" in key else 1.0)
        params.append({"params": [value], "lr": lr, "weight_decay": cfg.SOLVER.WEIGHT_DECAY})
    if cfg.SOLVER.OPTIMIZER != "SGD":
        raise NotImplementedError("hot path optimizer is SGD (every WSR config)")
    return HipSGD(params, cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM, clip=gradient_clipping(cfg))


def gradient_clipping(cfg):
    """SOLVER.CLIP_GRADIENTS -> HipSGD's `clip` (engine/defaults.py:292-323).  "full_model" is the reference's own
    FullModelGradientClippingOptimizer (L2 norm over all parameters, enabled only for CLIP_VALUE > 0); every other
    CLIP_TYPE goes to detectron2's maybe_add_gradient_clipping: "value" / "norm" per parameter (NORM_TYPE 2 only)."""
    cg = cfg.SOLVER.CLIP_GRADIENTS
    if not cg.ENABLED:
        return None
    kind, value = str(cg.CLIP_TYPE).lower(), float(cg.CLIP_VALUE)
    if kind == "full_model":
        return ("full_model", value) if value > 0.0 else None
    if kind == "value":
        return ("value", value)
    if kind == "norm":
        if float(cg.NORM_TYPE) != 2.0:
            raise NotImplementedError(f"SOLVER.CLIP_GRADIENTS.NORM_TYPE {cg.NORM_TYPE}: the HIP optimizer clips L2 norms")
        return ("norm", value)
    raise NotImplementedError(f"SOLVER.CLIP_GRADIENTS.CLIP_TYPE {cg.CLIP_TYPE!r} (full_model | value | norm)")


def wrap_model_with_ddp(model, local_rank, find_unused_parameters=False, bucket_cap_mb=128):
    """engine/defaults.py:135-153.  Single-dataset mode touches every parameter each step, so the
    unused-parameter graph walk of the reference is unnecessary."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return model
    from torch.nn.parallel import DistributedDataParallel

    kw = dict(broadcast_buffers=False, find_unused_parameters=find_unused_parameters, bucket_cap_mb=bucket_cap_mb,
              gradient_as_bucket_view=True)
    if next(model.parameters()).is_cuda:
        kw["device_ids"] = [local_rank]
    return DistributedDataParallel(model, **kw)


def run_step(model, optimizer, data, iter_size=1, it=0):
    """One iteration of DefaultTrainer_WSOVOD.run_step.  Returns the loss dict (device tensors)."""
    loss_dict = model(data)
    losses = sum(loss_dict.values())
    if iter_size > 1:
        losses = losses / iter_size
    losses.backward()
    if it % iter_size == 0:
        optimizer.step()
        optimizer.zero_grad(set_to_none=True)
    return loss_dict


class _StepMeta:
    """The per-step index data of a captured training step, as STATIC device buffers refilled before every replay:
    per-image segment offsets of the proposals, the row -> image map, and the image-level labels of
    get_image_level_gt (roi_heads.py:158-174) -- concatenated sorted-unique classes, their offsets, the one-hot
    matrix.  One pinned staging buffer and one asynchronous H2D copy per step."""

    MAX_CLASSES_PER_IMAGE = 128  # == kMaxPgt of the mining kernel

    def __init__(self, n_img, rows, num_classes, device, n_lr=0):
        self.n, self.rows, self.K = n_img, rows, num_classes
        self.t_cap = n_img * min(num_classes, self.MAX_CLASSES_PER_IMAGE)
        lay, off = {}, 0
        self.n_lr = int(n_lr)
        for name, count, dt in (("gt_cat", self.t_cap, torch.int64), ("onehot", n_img * num_classes, torch.float32),
                                ("seg", n_img + 1, torch.int32), ("gt_off", n_img + 1, torch.int32),
                                ("lr", max(self.n_lr, 1), torch.float32), ("row_group", rows, torch.int32)):
            lay[name] = (off, count, dt)
            off += (count * torch.empty((), dtype=dt).element_size() + 15) // 16 * 16
        self.nbytes = off
        self.dev = torch.zeros((off,), dtype=torch.uint8, device=device)
        self.host = [torch.zeros((off,), dtype=torch.uint8).pin_memory() for _ in range(4)]
        self.events = [None] * len(self.host)
        self.k = 0
        self.lay = lay
        self.seg, self.gt_off = self._view(self.dev, "seg"), self._view(self.dev, "gt_off")
        self.row_group, self.gt_cat = self._view(self.dev, "row_group"), self._view(self.dev, "gt_cat")
        self.onehot = self._view(self.dev, "onehot").view(n_img, num_classes)
        self.lr = self._view(self.dev, "lr")  # one learning rate per parameter group, refilled before every replay
        self.nums = None

    def _view(self, buf, name):
        off, count, dt = self.lay[name]
        return buf[off:off + count * torch.empty((), dtype=dt).element_size()].view(dt)

    def image_level_gt(self):
        return self.gt_cat, self.gt_off, self.onehot

    def overrides(self, nums):
        """{(values, dtype): static tensor} for hip_ops.const_override while the step is captured.  `nums`: the per-image
        row counts the model code sees during the capture (the last image carries the padding rows up to the bucket)."""
        offs = [0]
        for n in nums:
            offs.append(offs[-1] + n)
        rg = tuple(i for i, n in enumerate(nums) for _ in range(n))
        return {(tuple(offs), torch.int32): self.seg, (rg, torch.int32): self.row_group}

    @property
    def rows_true(self):
        """1-element int32 view of seg[G] = the step's REAL row count (<= the bucket the graph runs on)."""
        return self.seg[self.n:self.n + 1]

    def fill(self, batched_inputs, lrs=()):
        """Host side of one step: returns False when this batch does not fit the captured layout.  lrs: the optimizer's
        current learning rates, one per parameter group (a scheduler may have moved them since the last step)."""
        nums = [len(x["proposals"]) for x in batched_inputs]
        if len(nums) != self.n or sum(nums) > self.rows or min(nums) <= 0:
            return False
        cls = []
        for x in batched_inputs:
            gc = x["instances"].gt_classes
            if gc.is_cuda:
                return False
            cls.append(torch.unique(gc, sorted=True).to(torch.int64))
        if any(len(u) > self.MAX_CLASSES_PER_IMAGE for u in cls) or sum(len(u) for u in cls) > self.t_cap:
            return False
        k = self.k
        self.k = (k + 1) % len(self.host)
        if self.events[k] is not None:
            self.events[k].synchronize()  # (the copy issued four steps ago)
        h = self.host[k]
        h.zero_()
        seg, goff = self._view(h, "seg"), self._view(h, "gt_off")
        rg, cat = self._view(h, "row_group"), self._view(h, "gt_cat")
        oh = self._view(h, "onehot").view(self.n, self.K)
        r = t = 0
        for i, (n, u) in enumerate(zip(nums, cls)):
            rg[r:r + n] = i
            r += n
            seg[i + 1] = r
            cat[t:t + len(u)] = u
            t += len(u)
            goff[i + 1] = t
            oh[i, u] = 1.0
        if self.n_lr:
            self._view(h, "lr")[:self.n_lr] = torch.tensor([float(v) for v in lrs], dtype=torch.float32)
        self.dev.copy_(h, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev
        self.nums = nums
        return True


class _StepGraph:
    """One training step of the hot path as captured HIP graph(s) for one input layout (images per step, canvas size,
    bucket of the total proposal count -- `HotPathTrainer.row_bucket`): at the reference's own per-GPU batch (1 - 2 images, Base-RCNN-DilatedC5.yaml:61,65) the step is
    bound by the host issuing ~150 launches, not by the device.

    world == 1: ONE graph = frozen forward + heads forward + backward + fused SGD update.
    world  > 1: TWO graphs split at the exchange wait -- G1 = the frozen forward (replayed while the previous step's
    gradients are on the wire), then the host waits for the exchange and launches the SGD update on the reduced wire
    slices eagerly (one kernel; `flush()` between steps keeps its meaning), G2 = heads forward + backward + gradient
    pack; the collectives themselves are issued eagerly after G2 (never captured).

    Static inputs, refilled before a replay: the uint8 image canvas, the concatenated proposal boxes / objectness, and the
    index data of _StepMeta.  The dropout mask advances through a device-resident step term (box_head.py), the mining
    kernel reads its segment and label ranges from memory, SGD is one launch: nothing in the step needs a host value."""

    def __init__(self, trainer, batched_inputs, key):
        self.key = key
        self.tr = weakref.proxy(trainer)
        model = trainer.model
        dev = model.device
        imgs = [x["image"] for x in batched_inputs]
        n = len(imgs)
        self.canvas = torch.empty((n,) + tuple(imgs[0].shape), dtype=torch.uint8, device=dev)
        nums = [len(x["proposals"]) for x in batched_inputs]
        rows = trainer.row_bucket(sum(nums))  # the graph runs on this many rows; the steps' own totals may be anything below
        self.rows = rows
        self.source = key[-2]
        if self.source is not None:  # mixed-dataset model: the class count of THIS graph's source sizes the label buffers
            model.roi_heads.select_source(self.source)
        self.boxes = torch.zeros((rows, 4), dtype=torch.float32, device=dev)
        self.objectness = torch.zeros((rows,), dtype=torch.float32, device=dev)
        groups = trainer.optimizer.param_groups
        self.meta = _StepMeta(n, rows, model.roi_heads.num_classes, dev, n_lr=len(groups))
        self.hyper = [(g["weight_decay"], g["momentum"]) for g in groups]  # baked into the graph: a change drops it
        self.losses = None
        self.graphs = []
        self.fresh = False
        self._capture(batched_inputs, nums)

    # ---- inputs ----
    def _load(self, batched_inputs):
        from ..modeling.meta_arch import GeneralizedRCNN_WSOVOD as M

        groups = self.tr.optimizer.param_groups
        if [(g["weight_decay"], g["momentum"]) for g in groups] != self.hyper:
            return False
        if not self.meta.fill(batched_inputs, [g["lr"] for g in groups]):
            return False
        imgs = [x["image"] for x in batched_inputs]
        adj = M._adjacent(imgs) if imgs[0].is_cuda else None
        if adj is not None:
            self.canvas.copy_(adj, non_blocking=True)
        else:
            for i, im in enumerate(imgs):
                self.canvas[i].copy_(im, non_blocking=True)
        r = 0
        for x in batched_inputs:
            p = x["proposals"]
            m = len(p)
            self.boxes[r:r + m].copy_(p.proposal_boxes.tensor, non_blocking=True)
            self.objectness[r:r + m].copy_(p.objectness_logits, non_blocking=True)
            r += m
        if r < self.rows:  # padding rows: an empty box of the last image, objectness 0 (finite features, label -1)
            self.boxes[r:].zero_()
            self.objectness[r:].zero_()
        return True

    def _static_batch(self, batched_inputs, nums):
        from ..structures import Boxes, Instances

        out, r = [], 0
        nums = list(nums)
        nums[-1] += self.rows - sum(nums)  # the model code of the capture sees the padding rows as the last image's
        self.capture_nums = nums
        for i, (x, m) in enumerate(zip(batched_inputs, nums)):
            props = Instances(x["proposals"].image_size, proposal_boxes=Boxes(self.boxes[r:r + m]),
                              objectness_logits=self.objectness[r:r + m])
            d = {"image": self.canvas[i], "proposals": props, "instances": x["instances"]}
            for k in ("height", "width", "dataset_id"):
                if k in x:
                    d[k] = x[k]
            out.append(d)
            r += m
        return out

    # ---- capture ----
    def _capture(self, batched_inputs, nums):
        tr, model = self.tr, self.tr.model
        split = tr.exchange
        tr._finish_pending()  # the previous (eager) step's update lands first: the capture starts from clean state
        if not self._load(batched_inputs):
            raise RuntimeError("the batch does not fit the static input layout")
        static = self._static_batch(batched_inputs, nums)
        for m in model.modules():  # the device-resident step term must exist BEFORE the capture (its fill is not replayed)
            if hasattr(m, "_step_term"):
                m._step_term(model.device)
        dw = [(p, getattr(p, "_dw_split", None)) for p in tr.params]
        for p, _ in dw:  # the early weight-gradient block launches a collective from inside backward: not under capture
            p._dw_split = None
        model._step_meta = self.meta
        # the learning rates are read from memory by the captured SGD launch (an LR scheduler moves them between replays)
        tr.optimizer.lr_device = {p: self.meta.lr[i:i + 1] for i, g in enumerate(tr.optimizer.param_groups)
                                  for p in g["params"]}
        # no cyclic garbage collection while a stream is capturing: a collector run in the middle of the capture may
        # finalise an unreachable CUDAGraph / private pool of an earlier trainer -- HIP refuses that under capture and the
        # process aborts (seen once in ~10 full test runs).  torch.cuda.graph() collects BEFORE it starts capturing.
        gc_was_on = _gc.isenabled()
        _gc.disable()
        try:
            with H.const_override(self.meta.overrides(self.capture_nums)), H.tail_rows(self.meta.rows_true):
                if split:
                    g1 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                        st = model.forward_frozen(static)
                    g2 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g2, pool=g1.pool(), capture_error_mode="thread_local"):
                        loss_dict = model.forward_trainable(st)
                        tr._backward(loss_dict, capturing=True)
                        self.blocks = tr._exchange_bf16(pack_only=True)
                    self.wire = [(p, p._wire_grad) for p in tr.params]  # re-attached after every replay
                    self.graphs = [g1, g2]
                else:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        st = model.forward_frozen(static)
                        loss_dict = model.forward_trainable(st)
                        tr._backward(loss_dict, capturing=True)
                        tr.optimizer.step()
                        tr.optimizer.zero_grad(set_to_none=True)
                    self.graphs = [g]
        finally:
            if gc_was_on:
                _gc.enable()
            model._step_meta = None
            tr.optimizer.lr_device = None
            for p, v in dw:
                p._dw_split = v
        self.losses = {k: v.detach() for k, v in loss_dict.items()}
        self.last_pgt = getattr(model.roi_heads, "_last_pgt", None)  # this graph's static label / pseudo-GT buffers
        del st, loss_dict
        self.fresh = True  # the Python side of this step (counters, version stamps) ran during the capture

    # ---- one step ----
    def step(self, batched_inputs):
        tr = self.tr
        if self.fresh:
            self.fresh = False  # inputs were loaded for the capture; the capture itself executed nothing
        else:
            if not self._load(batched_inputs):
                return None
            if len(self.graphs) == 1:
                # an eager step (another layout) may have left its update deferred: it lands before this replay reads
                # the weights and runs its own SGD launch (no-op when nothing is pending)
                tr._finish_pending()
            tr._graph_bookkeeping()
        if len(self.graphs) == 1:
            if tr._stats is not None:
                tr._stats["steps"] += 1
            self.graphs[0].replay()
        else:
            done = tr._bracket("frozen")
            self.graphs[0].replay()  # frozen forward, while the previous step's gradients are still on the wire
            if done is not None:
                done.record()
            if tr._stats is not None:
                tr._stats["steps"] += 1
            tr._finish_pending()     # wait for the exchange, SGD on the reduced wire slices (one eager launch)
            self.graphs[1].replay()
            for p, sl in self.wire:
                p._wire_grad = sl
            tr._pending = [tr._reduce_block(lo, hi) for lo, hi in self.blocks]
            tr._stamp_hyper()
            if not tr.overlap:
                tr._wait_pending()
        # the Python-side state an eager step would have left: the heads' source (mixed-dataset model) and the step's
        # label / pseudo-GT record (with several graphs in rotation each has its own static buffers)
        rh = tr.model.roi_heads
        if self.source is not None:
            rh.select_source(self.source)
        if self.last_pgt is not None:
            rh._last_pgt = self.last_pgt
        # fresh tensors, as the eager path returns: the static loss buffers are overwritten by the next replay
        return {k: v.clone() for k, v in self.losses.items()}


class HotPathTrainer:
    """Data-parallel training loop of the hot path with the gradient exchange hidden behind compute.

    Same arithmetic as `run_step` + DistributedDataParallel (gradients averaged over ranks, then the
    SGD update; reference: engine/trainer.py:37-84, engine/defaults.py:143-148), different schedule:

        step t:  frozen forward(t)  ||  all-reduce of step t-1's gradients (RCCL, its own stream)
                 wait -> SGD update(t-1) (grad_scale = 1/world)  -> heads forward/backward(t)
                 -> launch all-reduce(t) asynchronously

    The frozen part (backbone, GAP, RoI pooling) reads no trainable parameter, so running it before the
    previous update lands changes nothing numerically; the update still precedes every use of the weights.
    Gradients are exchanged tensor-by-tensor (`all_reduce(async_op=True)`), so there is no bucket copy and
    `fc1.weight`'s 411 MB -- produced last in backward -- overlaps the next step's backbone instead of
    stalling the step.  Call `flush()` after the last step.
    """

    split_on_cpu = False  # tests: let the early-exchange logic run on a CPU stand-in model

    def __init__(self, model, optimizer, overlap=True, reduce_unused=False, grad_wire="fp32", iter_size=1,
                 exchange="ring", start_iter=0):
        """reduce_unused: parameters that received no gradient this step (mixed-dataset mode: the other
        datasets' object miners) still take part in the exchange with zeros, so that every rank issues the same
        collectives -- the job `find_unused_parameters=True` does in the reference (engine/defaults.py:146-148).

        grad_wire: "fp32" exchanges the fp32 gradients in place, tensor by tensor (DDP's arithmetic).  "bf16" is the
        counterpart of the reference's fp16 compression hook (engine/defaults.py:149-152): after backward ONE kernel
        rounds every gradient into its slice of a flat bf16 buffer, ONE all-reduce moves half the bytes (249 MB
        instead of 498 MB on R18 -- what matters at 2 and 4 ranks, where a ring has one / three xGMI links per GPU
        to work with), and the SGD kernel reads the reduced bf16 slices; master weights, momentum and the update
        stay fp32.

        exchange (bf16 wire only): "ring" = one RCCL all-reduce per block of the wire buffer, the library picks the
        algorithm and the running sum is rounded to bf16 at every hop (world - 1 roundings).  "direct" spells the
        reduce-scatter / all-gather out over the point-to-point xGMI mesh: an all-to-all hands every rank its shard of
        every other rank's buffer (each pair talks over its own link, all seven at once, one hop), a kernel sums the
        `world` copies in fp32 and rounds ONCE, an all-gather distributes the reduced shards -- the same bytes per GPU
        as a ring, one rounding of the sum whatever the world size.  "auto" = direct from 3 ranks up (at 2 ranks a
        ring's single addition rounds once as well).  The chain runs on a side stream, so the next step's frozen
        forward overlaps all three stages.

        iter_size: WSOVOD.ITER_SIZE (engine/trainer.py:72-84 of the reference): losses are divided by iter_size,
        gradients accumulate locally and the exchange + update happen on the iterations with `iter % iter_size == 0`
        (the reference's rule, first iteration included).

        The update of step t is applied lazily (inside the next run_step, behind the frozen forward).  Anything that
        reads the weights between steps -- `model.state_dict()`, `optimizer.state_dict()`, a checkpoint / eval / TTA
        hook -- must see them after optimizer.step() as in the reference (hooks run after run_step there): state-dict
        pre-hooks on the model and the optimizer call `synchronize()`; code that reads parameters directly calls
        `synchronize()` (= `flush()`) itself."""
        if grad_wire not in ("fp32", "bf16"):
            raise ValueError(f"grad_wire must be 'fp32' or 'bf16', got {grad_wire!r}")
        if grad_wire == "bf16" and not isinstance(optimizer, HipSGD):
            raise ValueError("grad_wire='bf16' needs the HIP optimizer (it reads the bf16 slices)")
        if exchange not in ("ring", "direct", "auto"):
            raise ValueError(f"exchange must be 'ring', 'direct' or 'auto', got {exchange!r}")
        if exchange == "direct" and grad_wire != "bf16":
            raise ValueError("exchange='direct' is defined on the bf16 wire buffer (grad_wire='bf16')")
        self.grad_wire = grad_wire
        self._wire = None
        self._direct = None  # (recv buffer, reduced-shard buffer) of the direct exchange
        self._side = None
        self._split = None  # (param, rows, elements) of the weight whose gradient is exchanged in two pieces
        self._early = None
        self.reduce_unused = reduce_unused
        self.model = model
        self.optimizer = optimizer
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.exchange = dist.is_initialized()  # a 1-rank group still goes through RCCL (exercised by the GPU tests)
        self.overlap = overlap
        self._pending = None  # list of (work, param) of the in-flight exchange
        self._pending_hyper = None  # (lr, weight_decay) per group at the time the pending step's gradients were produced
        self._stats = None  # stats_enable(): hipEvent pairs around the exchange wait and the frozen forward
        self._used = None  # per-tensor "some rank has a gradient" flags of the in-flight exchange (reduce_unused)
        self.params = [p for p in model.parameters() if p.requires_grad]
        if exchange == "auto":
            exchange = "direct" if grad_wire == "bf16" and self.world > 2 else "ring"
        self.exchange_algo = exchange if self.exchange else "none"
        self.iter_size = int(iter_size)
        self._seed = None
        # small batches are host-bound: the frozen backbone runs as one captured HIP graph per input shape
        # (modeling/backbone.py; WSOVOD_BACKBONE_GRAPH=0 keeps the eager launches)
        bb = getattr(model, "backbone", None)
        if bb is not None and hasattr(bb, "graph_max_batch") and os.environ.get("WSOVOD_BACKBONE_GRAPH", "1") != "0":
            bb.graph_max_batch = 8
        # the WHOLE step as captured HIP graph(s) at small batches (WSOVOD_STEP_GRAPH=0: eager launches + backbone graph)
        self.graph_max_batch = 8 if os.environ.get("WSOVOD_STEP_GRAPH", "1") != "0" else 0
        self._graphs, self._graph_seen = {}, {}
        self._graph_evicted, self._graph_recaptures = set(), 0
        self.iter = int(start_iter)  # the reference's global iteration (engine/trainer.py:72-84): pass it when resuming
        if self.iter_size < 1:
            raise ValueError(f"iter_size must be >= 1, got {iter_size}")
        if self.iter:  # resume: the neck's dropout stream is a function of the iteration (box_head.set_step)
            for m in model.modules():
                if hasattr(m, "set_step"):
                    m.set_step(self.iter)
        for p in self.params:
            p._wire_grad = None
        if self.exchange and grad_wire == "bf16" and self.params and self.iter_size == 1:
            self._setup_early_exchange()  # (with accumulation the early block would miss the earlier micro-steps)
        # round 6: without a gradient exchange, the optimizer step of the largest weight rides in its weight-gradient kernel
        # at small batches (WSOVOD_FUSED_SGD=0 switches it off; WSOVOD_FUSED_SGD_ROWS = the largest reduction it takes)
        self._fused = []  # the large 2-D weights (fc1, fc2): largest first
        if (not self.exchange and self.iter_size == 1 and isinstance(optimizer, HipSGD) and optimizer.clip is None
                and self.params and os.environ.get("WSOVOD_FUSED_SGD", "1") != "0"):
            for big in sorted(self.params, key=lambda q: -q.numel()):
                if big.dim() == 2 and big.is_cuda and big.is_contiguous() and big.numel() >= (1 << 24) and big.shape[1] % 32 == 0:
                    self._fused.append(big)
                    # (measured, WSR_18 x 512 proposals: 1 image 2.71 -> 2.59 ms per step, 8 images 7.55 -> 7.46; beyond 8
                    # images the longer epilogue costs more than the saved bytes: profiles/HISTORY.md, round 6)
                    big._fused_update = _FusedUpdate(optimizer, big, int(os.environ.get("WSOVOD_FUSED_SGD_ROWS", "4096")))
        self._hooks = []
        if hasattr(model, "register_state_dict_pre_hook"):
            self._hooks.append(model.register_state_dict_pre_hook(lambda *_a, **_k: self.synchronize()))
        if hasattr(optimizer, "register_state_dict_pre_hook"):
            self._hooks.append(optimizer.register_state_dict_pre_hook(lambda *_a, **_k: self.synchronize()))
        if isinstance(optimizer, HipSGD):
            optimizer.grad_scale = 1.0 / self.world
        # model.inference() between steps (EvalHook, TTA wrappers call it directly, past any forward hook) applies the
        # pending update first
        # (a weak reference: a deepcopy / pickle / torch.save of the model must not drag the trainer, the optimizer and the
        # process-group handles along)
        ref = weakref.ref(self)

        def _pre_inference():
            t = ref()
            if t is not None:
                t.synchronize()

        try:
            model._pre_inference = _pre_inference
            self._pre_inference_hook = _pre_inference
        except Exception:  # noqa: BLE001 -- a stand-in model without attribute assignment
            self._pre_inference_hook = None

    def close(self):
        """Apply the pending update and detach from the model / optimizer (state-dict hooks, inference hook)."""
        self.flush()
        for h in self._hooks:
            h.remove()
        self._hooks = []
        bb = getattr(self.model, "backbone", None)
        if bb is not None and hasattr(bb, "graph_max_batch"):
            bb.graph_max_batch = 0
            bb.__dict__.pop("_graphs", None)
        self._graphs.clear()
        for p in getattr(self, "_fused", None) or []:
            p.__dict__.pop("_fused_update", None)
        self._fused = []
        if getattr(self.model, "_pre_inference", None) is getattr(self, "_pre_inference_hook", None):
            self.model._pre_inference = None

    def broadcast_parameters(self, src=0):
        if self.exchange:
            for t in list(self.model.parameters()) + list(self.model.buffers()):
                dist.broadcast(t.data, src)
                # the broadcast wrote through `.data`: bump the version counter so that caches keyed on it (bf16 / bf16x2
                # shadows, folded conv weights, class matrices) are rebuilt from the received values
                torch.autograd.graph.increment_version(t)

    def set_exchange(self, algo):
        """Switch the bf16-wire exchange algorithm ("ring" / "direct") between steps, in this process: applies the pending
        update first, then rebuilds the wire buffer (the direct form pads it to whole shards) and the early block."""
        if algo not in ("ring", "direct"):
            raise ValueError(f"exchange must be 'ring' or 'direct', got {algo!r}")
        if algo == "direct" and self.grad_wire != "bf16":
            raise ValueError("exchange='direct' is defined on the bf16 wire buffer (grad_wire='bf16')")
        self.flush()
        if not self.exchange or algo == self.exchange_algo:
            return
        self.exchange_algo = algo
        self._reset_wire()

    def _reset_wire(self):
        self._wire = self._direct = self._early = None
        if self._split is not None:
            self._split[0]._dw_split = None
            self._split = None
        for p in self.params:
            p._wire_grad = None
        if self.exchange and self.grad_wire == "bf16" and self.params and self.iter_size == 1:
            self._setup_early_exchange()

    def abort_pending(self):
        """Drop an exchange that failed half way (an exception out of a collective): the step's gradients are discarded,
        no update is applied, the wire state is rebuilt.  Every rank must call it for the same step."""
        self._pending = self._used = None
        for p in self.params:
            p.grad = None
            p._wire_grad = None
            p._used_flag = None
        self._reset_wire()

    # ---- what a data-parallel step spent where (bench.py --gpus N puts it in the line, per rank) ----
    def stats_enable(self, on=True):
        """Bracket, from now on, (a) the wait for the previous step's gradient exchange and (b) the frozen forward it is
        meant to hide behind with hipEvents on the compute stream (two event records per step each: no synchronisation,
        no host read until stats())."""
        self._stats = {"wait": [], "frozen": [], "wait_ms": 0.0, "frozen_ms": 0.0, "steps": 0} if on else None

    def _bracket(self, kind):
        st = self._stats
        if st is None or not self.params or not self.params[0].is_cuda:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pairs = st[kind]
        while len(pairs) >= 256 and pairs[0][1].query():  # a long run keeps running sums, not one event pair per step
            a, b = pairs.pop(0)
            st[kind + "_ms"] += a.elapsed_time(b)
        pairs.append((e0, e1))
        return e1

    def stats(self):
        """-> dict: per-step means (ms, on this rank's compute stream) of `exchange_wait_ms` -- how long the stream stood
        still for the collectives of the previous step, i.e. the EXPOSED communication -- and `overlap_window_ms` -- the
        frozen forward the exchange runs behind; `wire_bytes` per step and rank; the exchange form.  Synchronises."""
        st = self._stats
        if st is None:
            return None
        if self.params and self.params[0].is_cuda:
            torch.cuda.synchronize(self.params[0].device)
        n = max(st["steps"], 1)
        wire = 0
        if self.exchange:
            if self.grad_wire == "bf16" and self._wire is not None:
                wire = self._wire[0].numel() * 2
            else:
                wire = sum(p.numel() for p in self.params) * 4
        return {"steps": st["steps"],
                "exchange_wait_ms": (st["wait_ms"] + sum(a.elapsed_time(b) for a, b in st["wait"])) / n,
                "overlap_window_ms": (st["frozen_ms"] + sum(a.elapsed_time(b) for a, b in st["frozen"])) / n,
                "wire_bytes_per_step": wire, "wire": self.grad_wire if self.exchange else None,
                "exchange": self.exchange_algo, "world": self.world,
                # a ring moves 2 (N-1)/N of the buffer over each GPU's links, the direct form the same bytes one hop
                "bytes_sent_per_rank_per_step": int(2 * (self.world - 1) / max(self.world, 1) * wire)}

    def _wait_pending(self):
        if self._pending is not None:
            done = self._bracket("wait") if any(w is not None for w in self._pending) else None
            for work in self._pending:
                if work is not None:
                    work.wait()
            if done is not None:
                done.record()
            self._pending = [None] * len(self._pending)  # waited for, update not applied yet

    def _finish_pending(self):
        if self._pending is None:
            return
        self._wait_pending()
        self._apply_update()

    def _stamp_hyper(self):
        """The update of a step is applied lazily (inside the NEXT run_step); an LR scheduler stepping in between -- the
        reference's LRScheduler hook runs after run_step, i.e. after optimizer.step() there (engine/trainer.py:57-84) --
        must not change the rate that step is applied with: the groups' (lr, weight_decay) are recorded with the
        pending gradients and put in place for the deferred optimizer.step()."""
        self._pending_hyper = [(g["lr"], g["weight_decay"]) for g in self.optimizer.param_groups]

    def _apply_update(self):
        """The optimizer step on the exchanged gradients (everything _finish_pending does after the waits)."""
        hyper, self._pending_hyper = self._pending_hyper, None
        groups = self.optimizer.param_groups
        now = None
        if hyper is not None and len(hyper) == len(groups):
            now = [(g["lr"], g["weight_decay"]) for g in groups]
            if now == hyper:
                now = None
            else:
                for g, (lr, wd) in zip(groups, hyper):
                    g["lr"], g["weight_decay"] = lr, wd
        try:
            self._apply_update_now()
        finally:
            if now is not None:
                for g, (lr, wd) in zip(groups, now):
                    g["lr"], g["weight_decay"] = lr, wd

    def _apply_update_now(self):
        if self._used is not None and not isinstance(self.optimizer, HipSGD):
            # a torch optimizer skips `grad is None`: restore that for tensors no rank touched (host read of a few flags;
            # the HIP optimizer reads the flags on the device instead)
            for p, u in zip(self.params, self._used.tolist()):
                if u == 0:
                    p.grad = None
        if self.world > 1 and not isinstance(self.optimizer, HipSGD):
            for p in self.params:
                if p.grad is not None:
                    p.grad.div_(self.world)
        self.optimizer.step()
        self.optimizer.zero_grad(set_to_none=True)
        for p in self.params:
            p._wire_grad = None
            p._used_flag = None
        self._used = None
        self._pending = None

    def _exchange_used_flags(self):
        """reduce_unused mode: one flag per tensor (this rank produced a gradient), summed over ranks.  Tensors unused on
        EVERY rank keep parameter and momentum untouched, exactly as SGD skips `grad is None` under the reference's
        DDP(find_unused_parameters=True) (engine/defaults.py:146-148) -- sending zeros alone would still apply weight
        decay and momentum to them.  Must be called before the gradients are packed / released."""
        host = torch.tensor([0.0 if p.grad is None else 1.0 for p in self.params], dtype=torch.float32)
        dev = self.params[0].device
        self._used = H.h2d_small(host, dev) if dev.type == "cuda" else host
        for i, p in enumerate(self.params):
            p._used_flag = self._used[i:i + 1]
        return dist.all_reduce(self._used, op=dist.ReduceOp.SUM, async_op=True)

    @staticmethod
    def split_rows(n_rows, n_cols, cus=256, tile=256):
        """Rows of the first block when an (n_rows, n_cols) weight gradient is computed as two launches of 256x256
        tiles on `cus` CUs: the split must not add a round of tiles (ceil(a*tj/cus) + ceil((ti-a)*tj/cus) equal to
        the unsplit count) and the first block should carry ~30 % of the rows, so that its all-reduce has the second
        block's contraction to hide behind.  0 = no admissible split."""
        ti, tj = -(-n_rows // tile), -(-n_cols // tile)
        whole = -(-ti * tj // cus)
        ok = [a for a in range(1, ti) if -(-a * tj // cus) + -(-(ti - a) * tj // cus) == whole and a * tile < n_rows]
        if not ok:
            return 0
        return min(ok, key=lambda a: abs(a - 0.3 * ti)) * tile

    def _setup_early_exchange(self):
        """The largest weight (fc1: 83 % of the gradient bytes) is the LAST gradient backward produces.  Its
        contraction is cut into two row blocks (layers/functions.py:_Linear.backward); the first block's slice of
        the wire buffer is packed and its all-reduce launched while the second block is still being computed."""
        p = max(self.params, key=lambda q: q.numel())
        if p.dim() != 2 or not (p.is_cuda or self.split_on_cpu):
            return
        cus = torch.cuda.get_device_properties(p.device).multi_processor_count if p.is_cuda else 256
        ra = self.split_rows(p.shape[0], p.shape[1], cus)
        if ra and self.exchange_algo == "direct" and (ra * p.shape[1]) % (8 * self.world):
            ra = 0  # the early block must be whole 16-byte groups per shard (fc1: 25088 columns = 64 * 392, always is)
        if ra:
            self._split = (p, ra, ra * p.shape[1])
            p._dw_split = (ra, self._early_block)

    def _early_block(self, dw_rows):
        """Called from inside backward with the first row block of the split weight's gradient (fp32, contiguous)."""
        flat, _ = self._wire_slices()
        n = self._split[2]
        assert dw_rows.numel() == n
        H.pack_bf16_multi([(dw_rows.reshape(-1), flat[:n])])
        self._early = self._reduce_block(0, n)

    def _wire_slices(self):
        """One flat bf16 buffer; every tensor owns a slice whose offset is rounded up to 8 elements (16-byte aligned).
        The split weight comes first, so that its first row block is the head [0, rows * cols) of the buffer."""
        if self._wire is None:
            order = list(self.params)
            if self._split is not None:
                order = [self._split[0]] + [p for p in order if p is not self._split[0]]
            offs, total = {}, 0
            for p in order:
                offs[id(p)] = total
                total += (p.numel() + 7) // 8 * 8
            if self.exchange_algo == "direct":  # every block of the buffer splits into `world` shards of 16-byte groups
                g = 8 * self.world
                total = (total + g - 1) // g * g
            flat = torch.zeros(total, dtype=torch.bfloat16, device=self.params[0].device)
            self._wire = (flat, [flat[offs[id(p)]:offs[id(p)] + p.numel()] for p in self.params])
        return self._wire

    def _reduce_block(self, lo, hi):
        """Sum flat[lo:hi] over the ranks, asynchronously; returns the work whose wait() makes the result visible."""
        flat, _ = self._wire_slices()
        if self.exchange_algo != "direct":
            return dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True)
        n = self.world
        assert (hi - lo) % (8 * n) == 0 and lo % (8 * n) == 0, (lo, hi, n)
        if self._direct is None:
            self._direct = (torch.empty_like(flat), torch.empty(flat.numel() // n, dtype=flat.dtype, device=flat.device))
            if flat.is_cuda:
                self._side = torch.cuda.Stream(device=flat.device)
        recv, mine = self._direct
        shard = (hi - lo) // n
        m = mine[lo // n:lo // n + shard]
        if flat.is_cuda:
            # a stream of its own: the three stages are ordered among themselves and behind the gradient pack, and the
            # main stream goes on with the next step's frozen forward (the buffers are persistent: no allocator hazard)
            self._side.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self._side):
                dist.all_to_all_single(recv[lo:hi], flat[lo:hi], async_op=True).wait()
                H.sum_shards_bf16(recv[lo:hi], n, m)
                return dist.all_gather_into_tensor(flat[lo:hi], m, async_op=True)
        dist.all_to_all_single(recv[lo:hi], flat[lo:hi], async_op=True).wait()
        H.sum_shards_bf16(recv[lo:hi], n, m)
        return dist.all_gather_into_tensor(flat[lo:hi], m, async_op=True)

    def _exchange_bf16(self, pack_only=False):
        """pack_only (a captured step graph): only the gradient pack is enqueued; returns the [(lo, hi)] blocks of the
        wire buffer whose collectives the caller issues (eagerly, after the graph's replay)."""
        flat, slices = self._wire_slices()
        pairs = []
        early, head = self._early, 0
        self._early = None
        if self._split is not None and early is None and self._split[0].grad is None and not self.reduce_unused:
            raise RuntimeError("the split weight received no gradient: every rank must issue the same collectives")
        for p, sl in zip(self.params, slices):
            if p.grad is None:
                if not self.reduce_unused:
                    continue
                sl.zero_()  # this rank contributes nothing to a tensor another rank may have touched
            else:
                g = (p.grad if p.grad.is_contiguous() else p.grad.contiguous()).reshape(-1)
                if early is not None and p is self._split[0]:  # its head is already packed and on the wire
                    head = self._split[2]
                    pairs.append((g[head:], sl[head:]))
                else:
                    pairs.append((g, sl))
            p._wire_grad = sl
        H.pack_bf16_multi(pairs)
        for p in self.params:  # the fp32 gradients are dead once packed (stream-ordered free): the update reads the slices
            p.grad = None
        if pack_only:
            assert early is None
            head = self._split[2] if self._split is not None else 0
            return [(0, head), (head, flat.numel())] if head else [(0, flat.numel())]
        if self._split is not None and early is None:
            # the early block did not come (e.g. a non-TN contraction): keep the collective sequence of the other ranks
            head = self._split[2]
            early = self._reduce_block(0, head)
        return [early, self._reduce_block(head, flat.numel())] if head else [self._reduce_block(0, flat.numel())]

    def _backward(self, loss_dict, capturing=False):
        """d(sum of the loss dict / iter_size): one backward pass from the loss tensors themselves with a cached seed --
        the same gradients as `sum(loss_dict.values()).backward()` without the adds, the division and the ones_like.
        capturing (a step graph is being recorded): the gradients are taken with autograd.grad and assigned, not
        accumulated -- an AccumulateGrad node kept alive from an EARLIER iteration (somebody still holds that step's loss
        tensors) belongs to the default stream and would pull work out of the capture."""
        roots = [v for v in loss_dict.values() if v.requires_grad]  # (a constant term has nothing to differentiate)
        if not roots:
            raise RuntimeError("HotPathTrainer: no term of the loss dict requires grad (is every parameter frozen, or was "
                               "the forward run under no_grad?): nothing to differentiate")
        seed = self._seed
        if seed is None or seed.device != roots[0].device or seed.dtype != roots[0].dtype:
            seed = self._seed = torch.full((), 1.0 / self.iter_size, dtype=roots[0].dtype, device=roots[0].device)
        fused = [p._fused_update for p in getattr(self, "_fused", None) or []]
        for f in fused:
            f.armed = True  # (this backward is followed by this trainer's optimizer step: the fused update IS that step)
        try:
            if capturing:
                grads = torch.autograd.grad(roots, self.params, [seed.expand_as(r) for r in roots], allow_unused=True)
                for p, g in zip(self.params, grads):
                    p.grad = g
                return
            torch.autograd.backward(roots, [seed.expand_as(r) for r in roots])
        finally:
            for f in fused:
                f.armed = False

    # ---- whole-step HIP graphs (small batches) ----
    GRAPH_AFTER = 3  # eager steps with a layout before it is captured (the first ones fill caches and allocator pools)
    GRAPH_CACHE = 4  # captured layouts kept (least recently replayed goes first)
    GRAPH_RECAPTURES = 8  # captures of layouts that had been captured before and evicted: beyond this, no new captures

    def _graph_key(self, data):
        m = self.model
        # (reduce_unused only matters where gradients are exchanged: at world = 1 the mixed-dataset model's unused miners
        # simply have no gradient and the captured SGD launch lists the tensors that do)
        if not (self.graph_max_batch and 0 < len(data) <= self.graph_max_batch and self.iter_size == 1
                and not (self.reduce_unused and self.exchange) and isinstance(self.optimizer, HipSGD)
                and getattr(m, "proposal_generator", True) is None and hasattr(m, "forward_frozen")
                and (not self.exchange or self.grad_wire == "bf16")):
            return None
        x0 = data[0]
        if "proposals" not in x0 or "instances" not in x0 or not torch.is_tensor(x0.get("image")):
            return None
        source = None
        if hasattr(m, "classifier_train"):
            # mixed-dataset model (rcnn_wsovod_mixed_datasets.py:188-191,237-238): the batch's source picks the object miner,
            # the class count and the text embeddings handed to the refinement head -- all three are STATIC per source, so
            # the source is part of the layout: one graph per (source, layout), replayed when that source comes round again
            ids = {int(x.get("dataset_id", 0)) for x in data}
            if len(ids) != 1:
                return None
            source = ids.pop()
            if not 0 <= source < len(m.classifier_train) or not hasattr(m.roi_heads, "num_classes_list"):
                return None
        if getattr(getattr(m, "backbone", None), "has_trainable_stage", False):
            return None  # FREEZE_AT < 5: the stage's backward is torch autograd over MIOpen convs, not capturable launches
        shape = tuple(x0["image"].shape)
        if any(tuple(x["image"].shape) != shape or x["image"].dtype != torch.uint8 for x in data):
            return None
        if not m.device.type == "cuda" or H.x3_active() or torch.cuda.is_current_stream_capturing():
            return None
        from .._lib import PROFILING

        if PROFILING[0]:  # the per-launch profiler needs the launches
            return None
        rh = m.roi_heads
        rows = self.row_bucket(sum(len(x["proposals"]) for x in data))
        # every row is kept by the refinement sampling (no random sub-sample inside the step): the shipped hot-path setting
        if not getattr(rh, "sampling_on", False) or rows > min(rh.batch_size_per_images[:rh.refine_K] or [0]) \
                or any(f < 1.0 for f in rh.positive_sample_fractions[:rh.refine_K]):
            return None
        K = rh.num_classes if source is None else rh.num_classes_list[source]
        return (len(data), shape, rows, K, self.exchange_algo, source, m.training)

    @staticmethod
    def row_bucket(rows):
        """The row count a step graph runs on: the total proposal count rounded up to a bucket (1/8 of the power of two
        below it, at least 64: <= 12 % padding rows, which ride along as an empty box with label -1)."""
        rows = max(int(rows), 1)
        step = max(64, (1 << (rows.bit_length() - 1)) // 8)
        return (rows + step - 1) // step * step

    def _graph_for(self, data):
        key = self._graph_key(data)
        if key is None or not key[-1]:
            return None
        g = self._graphs.get(key)
        if g is not None:
            self._graphs[key] = self._graphs.pop(key)  # least recently USED order: a replayed layout moves to the back
        else:
            if self._graph_recaptures > self.GRAPH_RECAPTURES:
                return None  # more hot layouts than the cache holds (multi-scale input): captures would thrash, stay eager
            seen = self._graph_seen
            if key not in seen and len(seen) >= 256:  # bounded: forget the layout sighted longest ago
                seen.pop(next(iter(seen)))
            seen[key] = seen.pop(key, 0) + 1
            if seen[key] < self.GRAPH_AFTER:
                return None
            if key in self._graph_evicted:
                self._graph_recaptures += 1
                if self._graph_recaptures > self.GRAPH_RECAPTURES:
                    warnings.warn("wsovod_amd: more training-step layouts in rotation than the HIP-graph cache holds "
                                  f"({self.GRAPH_CACHE}); further layouts keep the eager launches")
                    return None
            if len(self._graphs) >= self.GRAPH_CACHE:
                old = next(iter(self._graphs))
                self._graphs.pop(old)
                # an evicted layout starts over (and counts double: it has to show it is hotter than what replaced it)
                seen[old] = -self.GRAPH_AFTER
                self._graph_evicted.add(old)
            snap = self._counters()
            try:
                g = _StepGraph(self, data, key)
            except Exception as e:  # noqa: BLE001 -- out of memory in the graph's pool, an API call refused under capture
                warnings.warn(f"wsovod_amd: HIP graph capture of the training step failed for layout {key} "
                              f"({type(e).__name__}: {e}); this layout keeps the eager launches")
                self._restore_counters(snap)
                for p in self.params:
                    p.grad = None
                g = False
            self._graphs[key] = g
        return g or None

    def _counters(self):
        mods = [m for m in self.model.modules() if hasattr(m, "_step_term")]
        return ([(m, m._step) for m in mods], self.model.roi_heads.iter)

    def _restore_counters(self, snap):
        for m, v in snap[0]:
            m._step = v
        self.model.roi_heads.iter = snap[1]

    def _graph_bookkeeping(self):
        """The host-side state one eager step advances, mirrored for a replay (no Python model code runs then): the
        dropout step counters (their device terms advance inside the graph), the heads' iteration, the parameters'
        version counters (caches keyed on them) and the stamps of the shadows the SGD kernel refreshed."""
        rh = self.model.roi_heads
        rh.iter += 1
        for m in self._counters()[0]:
            m[0]._step += 1
        for p in self.params:
            sh = getattr(p, "_hip_shadow", None)
            xe = getattr(p, "_x2_enc", None)
            stamped = sh is not None and sh[1] == p._version
            stamped_x = xe is not None and xe[0] == (p._version, p.data_ptr(), None)
            torch.autograd.graph.increment_version(p)
            if stamped:
                p._hip_shadow = (sh[0], p._version)
            if stamped_x:
                p._x2_enc = ((p._version, p.data_ptr(), None), xe[1])

    def run_step(self, data):
        g = self._graph_for(data)
        if g is not None:
            out = g.step(data)
            if out is not None:
                self.iter += 1
                return out
        done = self._bracket("frozen")
        st = self.model.forward_frozen(data)
        if done is not None:
            done.record()
        if self._stats is not None:
            self._stats["steps"] += 1
        self._finish_pending()
        loss_dict = self.model.forward_trainable(st)
        self._backward(loss_dict)
        step_now = self.iter % self.iter_size == 0
        self.iter += 1
        if not step_now:  # gradients keep accumulating in p.grad; nothing goes on the wire
            return {k: v.detach() for k, v in loss_dict.items()}
        works = []
        if self.exchange and self.reduce_unused:
            works.append(self._exchange_used_flags())
        if self.exchange and self.grad_wire == "bf16":
            works += self._exchange_bf16()
        elif self.exchange:
            for p in self.params:
                if p.grad is None and self.reduce_unused:
                    p.grad = torch.zeros_like(p)
                if p.grad is not None:
                    works.append(dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, async_op=True))
        self._pending = works
        self._stamp_hyper()
        if not self.overlap:
            self._finish_pending()
        # (detached: the step's autograd graph dies here whatever the caller keeps)
        return {k: v.detach() for k, v in loss_dict.items()}

    def flush(self):
        """Apply the update of the last run_step (waits for its gradient exchange)."""
        self._finish_pending()

    synchronize = flush
