"""torch.autograd fronts of the HIP kernels -- the counterpart of the reference's
`wsovod/layers/roi_loop_pool.py:9-35` (autograd.Function over the native op, `save_for_backward`,
`once_differentiable`) extended to every op on the hot path.  Forward AND backward run on the
C-ABI kernels; torch only carries tensors, streams and the autograd graph.
"""
import contextlib
import os
import threading

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import hip_ops as H


_USE_TN = os.environ.get("WSOVOD_DISABLE_TN", "0") != "1"  # A/B switch for the transposed-read dW kernel


def _pad(n, m):
    return (n + m - 1) // m * m


def scale_losses(losses, loss_weight):
    """{name: loss * loss_weight.get(name, 1)}; a weight of exactly 1 (every shipped config) launches nothing."""
    out = {}
    for k, v in losses.items():
        w = loss_weight.get(k, 1.0) if isinstance(loss_weight, dict) else loss_weight
        out[k] = v if w == 1.0 else v * w
    return out


def _contig2d(t):
    return t if (t.dim() == 2 and t.stride(1) == 1) else t.contiguous()


def weight_shadow(weight, cd):
    """Compute-dtype copy of an fp32 master weight.  For leaf parameters the bf16 copy is cached on the
    parameter (`_hip_shadow` = (tensor, version)); HipSGD refreshes it inside the fused update kernel, so in
    steady state no separate cast pass over the weights runs.  Any other in-place change of the parameter
    bumps `_version` and invalidates the cache."""
    if cd == torch.float32:
        return weight
    sh = getattr(weight, "_hip_shadow", None)
    if sh is not None and sh[1] == weight._version and sh[0].dtype == cd:
        return sh[0]
    t = H.cast(weight.detach(), cd)
    if weight.is_leaf:
        weight._hip_shadow = (t, weight._version)
    return t


def _x2_mode():
    """MODEL.HIP.PRECISION = "parity": the activations that reach the Linear fronts are bf16x2 tensors (hip_ops.X2)."""
    return H.x3_active() == "x2"


class _BwdSplitState(threading.local):
    on = False


_BWD_SPLIT = _BwdSplitState()


@contextlib.contextmanager
def backward_split(on=True):
    """MODEL.HIP.PRECISION = "parity_train": Functions created inside keep the hi/lo split in their BACKWARD contractions
    too (three bf16 MFMA products on fp32 gradients, decoded bf16x2 activations and fp32 master weights -- the arithmetic
    of the "bf16x3" mode's backward) instead of plain bf16 products on the hi halves.  The "parity" forward is unchanged;
    what changes is the trained trajectory (tests/test_gpu_full_size.py: five optimizer steps against the oracle).
    Which contractions keep it (WSOVOD_PT_SPLIT, default "dx"): the ablation of round 6 (tools/parity_train_ablation.py,
    profiles/r06_parity_train_ablation.json) shows that the trajectory error of the plain-bf16 backward comes from the
    INPUT-gradient contractions dX = dA W -- their rounding is inherited by every layer further back -- and not from the
    weight gradients dW = dA^T X, whose rounding is independent noise per element: after five optimizer steps the logits are
    6.7e-3 from the oracle's with neither, 7.1e-3 with the split in dW only, 4.0e-4 with the split in dX only, 4.2e-4 with
    both.  "dw,dx" keeps both (the "bf16x3" mode's backward, ~+35 % step time instead of ~+11 %)."""
    prev = _BWD_SPLIT.on
    _BWD_SPLIT.on = bool(on)
    try:
        yield
    finally:
        _BWD_SPLIT.on = prev


def _bwd_split():
    """-> frozenset of {"dw", "dx"}: which backward contractions of a Function created now keep the split."""
    if not _BWD_SPLIT.on:
        return frozenset()
    which = os.environ.get("WSOVOD_PT_SPLIT", "dx")
    return frozenset(w for w in which.split(",") if w in ("dw", "dx"))


def _no_split(x3):
    """The x3 state a backward pass runs under: the forward-only modes ("fwd", "x2") contract in plain bf16."""
    return x3 if x3 not in ("fwd", "x2") else False


class _Linear(Function):
    """y = dropout(relu(x @ W^T + b)); x (M,K) in the compute dtype, W fp32 master (N,K).
    "parity" precision: x is bf16x2, the products are three-MFMA sums on the bf16x2 weight, y is bf16x2 (out_dtype =
    hip_ops.X2) or fp32; the backward is plain bf16 on the hi halves (mask from y's hi halves, dW by the transposed-read
    kernel straight from the bf16x2 x)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, dropout_p, seed, out_dtype, seed_add=None, grad_on=True):
        ctx.x2 = _x2_mode()
        ctx.y_x2 = out_dtype == H.X2
        x_hi = y_mask = None
        if ctx.x2 and H.mx_of(x):
            # "parity_mx": x is a unit-scale f16mx carrier (the pooler's, or the previous FC layer's); the products are
            # fp16 hi*hi + block-scaled e4m3 cross terms on the f16mx weight (one re-encode per optimizer step).  An f16mx
            # output comes with its plain bf16 rounding: the mask source of this layer's backward and the operand of the
            # NEXT layer's weight gradient
            wq, wscale = H.mx_cached(weight, tensor_scale=weight.is_leaf and weight.requires_grad)
            y_bf16 = None
            # (grad_on: the caller's grad mode -- it is off inside Function.forward, and needs_input_grad ignores no_grad)
            if out_dtype == H.MX and grad_on and any(ctx.needs_input_grad):
                y_bf16 = torch.empty((x.shape[0], weight.shape[0]), dtype=torch.bfloat16, device=x.device)
            y = H.gemm_mx(x, None, wq, wscale, bias=bias, relu=relu, dropout_p=dropout_p, dropout_seed=seed,
                          dropout_seed_add=seed_add, out_dtype=out_dtype, out_bf16=y_bf16)
            if out_dtype == H.MX:
                y._mx = True
                if y_bf16 is not None:
                    y._x2_hi = y_mask = y_bf16
            x_hi = H.x2_hi_pop(x)
            if x_hi is None and grad_on and ctx.needs_input_grad[1]:
                raise RuntimeError("wsovod_hip linear: an f16mx input needs its plain bf16 copy for the weight gradient "
                                   "(the pooler / the previous layer writes it in training mode)")
        elif ctx.x2:
            y = H.gemm_nt(x, H.x2_cached(weight), x2=True, bias=bias, relu=relu, dropout_p=dropout_p, dropout_seed=seed,
                          dropout_seed_add=seed_add, out_dtype=out_dtype, a_planar=H.x2_planar_of(x))
            x_hi = H.x2_hi_pop(x)  # the pooler's plain bf16 copy of x, if it wrote one: the operand of dW
        else:
            wq = weight_shadow(weight, x.dtype)
            y = H.gemm_nt(x, wq, bias=bias, relu=relu, dropout_p=dropout_p, dropout_seed=seed, dropout_seed_add=seed_add,
                          out_dtype=out_dtype)
        ctx.relu, ctx.dropout_p = relu, dropout_p
        ctx.bwd = _bwd_split() if ctx.x2 else frozenset()
        if ctx.bwd and H.mx_of(x):
            raise NotImplementedError('"parity_train" (a backward that keeps the hi/lo split) is not combined with f16mx activations')
        if "dw" in ctx.bwd:
            x_hi = None  # the split weight gradient reads hi AND lo: the carrier itself is kept (either layout)
        if y_mask is not None:
            ctx.y_x2 = False  # (the mask is read from the plain bf16 rounding of the f16mx output)
        ctx.save_for_backward(x if x_hi is None else x_hi.view(x.shape), weight,
                              (y if y_mask is None else y_mask) if (relu or dropout_p > 0) else None)
        ctx.x_is_hi = x_hi is not None
        ctx.has_bias = bias is not None
        # (rows, callback) set by a data-parallel trainer on ONE large weight: its gradient is produced in two row
        # blocks and the callback sees the first as soon as it is enqueued (engine/trainer.py: early exchange)
        ctx.dw_split = getattr(weight, "_dw_split", None)
        ctx.x3 = H.x3_active()
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        with H.x3_mode(_no_split(ctx.x3)):
            return _Linear._backward(ctx, dy)

    @staticmethod
    def _backward_split(ctx, dy):
        """"parity_train": dW = dA^T X and / or dX = dA W with the hi/lo split kept -- fp32 dA (mask applied, not rounded),
        the saved bf16x2 input decoded to fp32, the fp32 master weight; the contractions are the "bf16x3" mode's (three bf16
        MFMA products per value pair, operands split on the fly).  A contraction not named in ctx.bwd runs as in "parity"."""
        x, weight, y = ctx.saved_tensors
        M, K = x.shape
        N = weight.size(0)
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_db = ctx.has_bias and ctx.needs_input_grad[2]
        s_dw, s_dx = need_dw and "dw" in ctx.bwd, need_dx and "dx" in ctx.bwd
        dy = _contig2d(dy)
        scale = 1.0 / (1.0 - ctx.dropout_p) if ctx.dropout_p > 0 else 1.0
        Mp, Np = _pad(M, 64), _pad(N, 8)
        y_x2 = ctx.y_x2
        db = torch.zeros((N,), dtype=torch.float32, device=dy.device) if need_db else None
        dA32 = dAt32 = dA16 = None
        if s_dw or s_dx:
            dA32, dAt32 = H.mask_transpose(dy, y, scale, torch.float32, want_plain=s_dx, want_t=s_dw, ld_t=Mp, ld_plain=Np,
                                           colsum=db, y_x2=y_x2)
        if (need_dw and not s_dw) or (need_dx and not s_dx) or (db is not None and dA32 is None and dAt32 is None):
            dA16, _ = H.mask_transpose(dy, y, scale, torch.bfloat16, want_plain=True, want_t=False, ld_plain=Np,
                                       colsum=db if (dA32 is None and dAt32 is None) else None, y_x2=y_x2)
        dx = dw = None
        if s_dw:
            xt = H.transpose_cast(H.x2_to_f32(x), torch.float32, ld_dst=Mp)  # (K, Mp)
            with H.x3_mode("full"):
                dw = H.gemm_nt(dAt32, xt, out_dtype=torch.float32)  # (N, K)
            del xt
        elif need_dw:
            # (x is the pooler's plain bf16 copy when it wrote one -- ctx.x_is_hi --, else the interleaved bf16x2 carrier)
            dw = H.gemm_tn(dA16, x, q_x2=not ctx.x_is_hi)
            if Np != N:
                dw = dw[:N]
        if dw is not None and ctx.dw_split is not None and 0 < ctx.dw_split[0] < N and ctx.dw_split[0] % 8 == 0:
            ctx.dw_split[1](dw[:ctx.dw_split[0]])  # (a data-parallel trainer's early block: handed over late, still first)
        if s_dx:
            wt = H.transpose_cast(weight, torch.float32, ld_dst=Np)  # (K, Np)
            with H.x3_mode("full"):
                dx = H.gemm_nt(dA32, wt, out_dtype=torch.float32)  # (M, K)
        elif need_dx:
            dx = H.gemm_nt(dA16, H.transpose_cast(weight, torch.bfloat16, ld_dst=Np), out_dtype=torch.float32)
        return dx, dw, db, None, None, None, None, None, None

    @staticmethod
    def _backward(ctx, dy):
        if ctx.bwd:
            return _Linear._backward_split(ctx, dy)
        x, weight, y = ctx.saved_tensors
        in_dtype = torch.float32 if ctx.x_is_hi else x.dtype
        if ctx.x3 == "fwd":  # bf16x3f: plain bf16 backward on a cast of the saved fp32 input
            x = H.cast(x, torch.bfloat16)
        cd = torch.bfloat16 if ctx.x2 else x.dtype
        q_x2 = ctx.x2 and not ctx.x_is_hi  # x saved as bf16x2: the transposed-read kernel takes its hi halves
        M, K = x.shape
        N = weight.size(0)
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_db = ctx.has_bias and ctx.needs_input_grad[2]
        dy = _contig2d(dy)
        scale = 1.0 / (1.0 - ctx.dropout_p) if ctx.dropout_p > 0 else 1.0
        Mp, Np = _pad(M, 64), _pad(N, 8)
        # dW = dA^T X contracts over the slow (proposal) index of both row-major operands.  Large bf16 layers use the
        # transposed-read kernel on them as they are; the rest goes through explicit transposes + the NT kernel.
        tn = ctx.x2 or (_USE_TN and need_dw and cd == torch.bfloat16 and K % 8 == 0 and x.stride(1) == 1
                        and x.stride(0) % 8 == 0 and ((Np + 255) // 256) * ((K + 255) // 256) >= 128)
        db = torch.zeros((N,), dtype=torch.float32, device=dy.device) if need_db else None  # summed in the same pass
        dA, dAt = H.mask_transpose(dy, y, scale, cd, want_plain=need_dx or tn, want_t=need_dw and not tn,
                                   ld_t=Mp, ld_plain=Np, colsum=db, y_x2=ctx.x2 and ctx.y_x2)
        dx = dw = None
        # round 6: the trainer may have put a fused update on ONE large weight (engine/trainer.py:_FusedUpdate): dW and the
        # optimizer step of that tensor are then one kernel and no gradient is returned (the optimizer skips `grad is None`).
        # The input gradient reads the weights BEFORE the update, so it is taken first.
        fused = getattr(weight, "_fused_update", None) if (tn and need_dw and Np == N and ctx.dw_split is None) else None
        if fused is not None and fused.wants(M):
            if need_dx:
                dx = H.gemm_nt(dA, H.transpose_cast(weight, cd, ld_dst=Np), out_dtype=in_dtype)
            if fused(dA, x, q_x2):
                return dx, None, db, None, None, None, None, None, None
            need_dx = need_dx and dx is None
        if tn:
            split = ctx.dw_split
            if split is not None and Np == N and 0 < split[0] < N and split[0] % 8 == 0:
                ra = split[0]  # two launches over row blocks of dW; the first block is handed over before the second runs
                dw = torch.empty((N, K), dtype=torch.float32, device=x.device)
                H.gemm_tn(dA[:, :ra], x, out=dw[:ra], q_x2=q_x2)
                split[1](dw[:ra])
                H.gemm_tn(dA[:, ra:], x, out=dw[ra:], q_x2=q_x2)
            else:
                dw = H.gemm_tn(dA, x, q_x2=q_x2) if need_dw else None  # (Np, K)
                if dw is not None and Np != N:
                    dw = dw[:N]
        elif need_dw:
            xt = H.transpose_cast(x, cd, ld_dst=Mp)  # (K, Mp)
            dw = H.gemm_nt(dAt, xt, out_dtype=torch.float32)  # (N,K) = dA^T X, reduction over proposals
        if need_dx and dx is None:
            wt = H.transpose_cast(weight, cd, ld_dst=Np)  # (K, Np) shadow of W^T
            dx = H.gemm_nt(dA, wt, out_dtype=in_dtype)  # (M,K)
        return dx, dw, db, None, None, None, None, None, None


def linear(x, weight, bias=None, relu=False, dropout_p=0.0, seed=0, out_dtype=None, seed_add=None):
    """out_dtype: a torch dtype, or hip_ops.X2 for a bf16x2 output ("parity" precision; the default there is fp32).
    seed_add: optional 1-element int64 DEVICE tensor added to `seed` inside the kernel (the per-step term of the dropout
    seed kept in memory, so that a captured HIP graph draws a fresh mask at every replay)."""
    return _Linear.apply(x, weight, bias, relu, float(dropout_p), int(seed), out_dtype or x.dtype, seed_add,
                         torch.is_grad_enabled())


class _LinearGroup(Function):
    """Several Linear(+ReLU) heads on ONE input (object mining, box regression and the classifier projection all
    read the same box features).  A head's weight may be given in row blocks (object mining: [cls | det] are two
    modules, one (2K, F) contraction) and consecutive heads with the same epilogue can be JOINED: one forward GEMM on the
    stacked rows (the input is streamed once), the heads' outputs are column blocks of its result.  Backward: the heads'
    masked output gradients are written side by side into one (M, sum N_h) matrix, so that the input gradient is ONE GEMM
    against the stacked weights (instead of one skinny GEMM per head plus adds), the weight gradients ONE contraction
    (one pass over x, no per-head x^T), the bias gradients one column sum."""

    @staticmethod
    def forward(ctx, x, meta, *wb):
        cd = x.dtype
        heads, joins = meta  # heads[h] = (relu, out_dtype, number of row blocks); joins = runs of head indices
        ws, bs = wb[0::2], wb[1::2]
        first = [0]
        for h in heads:
            first.append(first[-1] + h[2])
        ctx.x2 = _x2_mode()

        def operand(w):
            return H.x2_cached(w) if ctx.x2 else weight_shadow(w, cd)

        ys = [None] * len(heads)
        for run in joins:
            relu, out_dtype, _ = heads[run[0]]
            blocks = [i for h in run for i in range(first[h], first[h + 1])]
            wq = operand(ws[blocks[0]]) if len(blocks) == 1 else torch.cat([operand(ws[i]) for i in blocks])
            bias = bs[blocks[0]] if len(blocks) == 1 else \
                (torch.cat([bs[i] for i in blocks]) if all(bs[i] is not None for i in blocks) else None)
            if bias is None and any(bs[i] is not None for i in blocks):
                raise RuntimeError("linear_group: the row blocks of a joined contraction must all have a bias or none")
            y = H.gemm_nt(x, wq, x2=ctx.x2, bias=bias, relu=relu, out_dtype=out_dtype or cd)
            col = 0
            for h in run:
                n = sum(ws[i].size(0) for i in range(first[h], first[h + 1]))
                ys[h] = y if len(run) == 1 else y[:, col:col + n]
                col += n
        ctx.heads, ctx.first = heads, first
        ctx.bwd = _bwd_split() if ctx.x2 else frozenset()
        ctx.save_for_backward(x, *ws, *[y if heads[h][0] else None for h, y in enumerate(ys)])
        ctx.has_bias = [b is not None for b in bs]
        ctx.x3 = H.x3_active()
        return tuple(ys)

    @staticmethod
    @once_differentiable
    def backward(ctx, *dys):
        with H.x3_mode(_no_split(ctx.x3)):
            return _LinearGroup._backward(ctx, *dys)

    @staticmethod
    def _backward(ctx, *dys):
        heads, first = ctx.heads, ctx.first
        nh, nb = len(heads), first[-1]
        saved = ctx.saved_tensors
        x, ws, ys = saved[0], saved[1:1 + nb], saved[1 + nb:]
        in_dtype = x.dtype
        if ctx.x3 == "fwd":  # bf16x3f: plain bf16 backward on a cast of the saved fp32 input
            x = H.cast(x, torch.bfloat16)
        cd = torch.bfloat16 if ctx.x2 else x.dtype
        # "parity_train": fp32 masked gradients, decoded input, fp32 weights -- the generic path below under the bf16x3 split
        # (WSOVOD_PT_SPLIT names one of the two contractions only: the whole group then follows the weight gradient's choice)
        split = ctx.x2 and bool(ctx.bwd)
        if split:
            cd = torch.float32
        M, K = x.shape
        Ns = [sum(ws[i].size(0) for i in range(first[h], first[h + 1])) for h in range(nh)]
        offs = [0]
        for n in Ns:
            offs.append(offs[-1] + _pad(n, 8))
        # (the stacked width in whole 64-column K-steps: the input-gradient GEMM contracts over it and takes the lean form
        # of the 8-wavefront tile; the columns behind the last head stay zero)
        Nt = _pad(offs[-1], 64)
        dA = torch.zeros((M, Nt), dtype=cd, device=x.device)
        want_db = any(ctx.has_bias[i] and ctx.needs_input_grad[3 + 2 * i] for i in range(nb))
        dbcat = torch.zeros((Nt,), dtype=torch.float32, device=x.device) if want_db else None
        for h in range(nh):
            if dys[h] is not None:
                H.mask_transpose(_contig2d(dys[h]), ys[h], 1.0, cd, want_t=False,
                                 out_plain=dA[:, offs[h]:offs[h] + Ns[h]],
                                 colsum=dbcat[offs[h]:offs[h] + Ns[h]] if want_db else None,
                                 y_x2=ctx.x2 and heads[h][1] == H.X2)
        xs = H.x3_mode("full") if split else contextlib.nullcontext()
        need_dx = ctx.needs_input_grad[0]
        need_dw = any(ctx.needs_input_grad[2 + 2 * i] for i in range(nb))
        rows = []  # (row block i, first row in the stacked layout)
        for h in range(nh):
            r = offs[h]
            for i in range(first[h], first[h + 1]):
                rows.append((i, r))
                r += ws[i].size(0)
        dx = None
        if need_dx:
            wcat = torch.zeros((Nt, K), dtype=cd, device=x.device)
            for i, r in rows:
                wcat[r:r + ws[i].size(0)] = ws[i]
            with xs:
                dx = H.gemm_nt(dA, H.transpose_cast(wcat, cd), out_dtype=in_dtype)  # (M,K) = dA_cat @ W_cat
        grads = [None] * (2 * nb)
        if need_dw:
            if split:
                Mp = _pad(M, 64)
                with xs:
                    dwcat = H.gemm_nt(H.transpose_cast(dA, cd, ld_dst=Mp), H.transpose_cast(H.x2_to_f32(x), cd, ld_dst=Mp),
                                      out_dtype=torch.float32)
            elif ctx.x2:
                dwcat = H.gemm_tn(dA, x, q_x2=True)
            elif _USE_TN and cd == torch.bfloat16 and K % 8 == 0 and x.stride(1) == 1 and x.stride(0) % 8 == 0 \
                    and ((Nt + 255) // 256) * ((K + 255) // 256) >= 64:
                dwcat = H.gemm_tn(dA, x)
            else:
                Mp = _pad(M, 64)
                dwcat = H.gemm_nt(H.transpose_cast(dA, cd, ld_dst=Mp), H.transpose_cast(x, cd, ld_dst=Mp),
                                  out_dtype=torch.float32)
            for i, r in rows:
                if ctx.needs_input_grad[2 + 2 * i]:
                    grads[2 * i] = dwcat[r:r + ws[i].size(0)]
        if want_db:
            for i, r in rows:
                if ctx.has_bias[i] and ctx.needs_input_grad[3 + 2 * i]:
                    grads[2 * i + 1] = dbcat[r:r + ws[i].size(0)]
        return (dx, None, *grads)


def linear_group(x, heads, joins=None):
    """heads: list of (weight (N_h,K) fp32 master -- or a list of row blocks --, bias (list of biases) or None, relu,
    out_dtype or None); joins: optional list of runs of consecutive head indices computed by ONE forward GEMM each (same
    relu / out_dtype; no bf16x2 output) -- heads not named are contractions of their own.  -> tuple of outputs, one per
    head (the outputs of a joined run are column blocks of one matrix)."""
    meta_heads, wb = [], []
    for h in heads:
        blocks = list(h[0]) if isinstance(h[0], (list, tuple)) else [h[0]]
        biases = list(h[1]) if isinstance(h[1], (list, tuple)) else [h[1]] * len(blocks)
        if len(biases) != len(blocks) or (len(blocks) > 1 and h[1] is not None and not isinstance(h[1], (list, tuple))):
            raise RuntimeError("linear_group: one bias per row block expected")
        meta_heads.append((bool(h[2]), h[3], len(blocks)))
        for w, b in zip(blocks, biases):
            wb += [w, b]
    named = set()
    runs = []
    for run in (joins or []):
        run = list(run)
        if run != list(range(run[0], run[0] + len(run))) or named & set(run) or \
                any(meta_heads[h][:2] != meta_heads[run[0]][:2] for h in run) or \
                (len(run) > 1 and (meta_heads[run[0]][0] or meta_heads[run[0]][1] == H.X2)):
            raise RuntimeError("linear_group: a join is a run of consecutive heads with the same (ReLU-free) epilogue")
        named |= set(run)
        runs.append(tuple(run))
    runs += [(h,) for h in range(len(heads)) if h not in named]
    runs.sort()
    return _LinearGroup.apply(x, (tuple(meta_heads), tuple(runs)), *wb)


class _WantHi(threading.local):
    """Per-thread (as the x3 state): set by the ROI heads while training under the "parity" precision, when the pooled
    tensor feeds a weight-gradient contraction."""
    on = False


_WANT_HI = _WantHi()


class _RoIPool(Function):
    @staticmethod
    def forward(ctx, feat, rois, output_size, spatial_scale, roi_scale, out_dtype):
        need_grad = feat.requires_grad
        out, argmax = H.roi_pool_forward(feat, rois, spatial_scale, output_size, roi_scale=roi_scale,
                                         out_dtype=out_dtype, need_argmax=need_grad,
                                         want_hi=out_dtype in (H.X2, H.MX) and _WANT_HI.on)
        ctx.shape = tuple(feat.shape)
        ctx.cl = not feat.is_contiguous()
        ctx.in_dtype = feat.dtype
        if need_grad:
            ctx.save_for_backward(rois, argmax, roi_scale)
            ctx.mark_non_differentiable(argmax)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        rois, argmax, roi_scale = ctx.saved_tensors
        g = H.roi_pool_backward(grad_output, rois, argmax, ctx.shape, channels_last=ctx.cl, roi_scale=roi_scale)
        return g.to(ctx.in_dtype), None, None, None, None, None


class _RoILoopPool(Function):
    """The reference's own autograd front (wsovod/layers/roi_loop_pool.py:9-35) on the HIP op: (3R, C, ph, pw)."""

    @staticmethod
    def forward(ctx, feat, rois, output_size, spatial_scale):
        out, argmax = H.roi_loop_pool_forward(feat, rois, spatial_scale, output_size)
        ctx.shape = tuple(feat.shape)
        ctx.cl = not feat.is_contiguous()
        ctx.in_dtype = feat.dtype
        ctx.save_for_backward(rois, argmax)
        return out.to(feat.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        rois, argmax = ctx.saved_tensors
        g = H.roi_pool_backward(grad_output, rois.repeat(3, 1), argmax, ctx.shape, channels_last=ctx.cl)
        return g.to(ctx.in_dtype), None, None, None


def roi_loop_pool(feat, rois, output_size, spatial_scale):
    return _RoILoopPool.apply(feat, rois, tuple(output_size), float(spatial_scale))


def roi_pool(feat, rois, output_size, spatial_scale, roi_scale=None, out_dtype=None):
    return _RoIPool.apply(feat, rois, tuple(output_size), float(spatial_scale), roi_scale, out_dtype or feat.dtype)


class _RoIAlign(Function):
    @staticmethod
    def forward(ctx, feat, rois, output_size, spatial_scale, sampling_ratio, aligned, roi_scale, out_dtype):
        out = H.roi_align_forward(feat, rois, spatial_scale, output_size, sampling_ratio, aligned,
                                  roi_scale=roi_scale, out_dtype=out_dtype, want_hi=out_dtype in (H.X2, H.MX) and _WANT_HI.on)
        ctx.cfg = (tuple(feat.shape), not feat.is_contiguous(), spatial_scale, sampling_ratio, aligned, feat.dtype)
        ctx.save_for_backward(rois, roi_scale)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        rois, roi_scale = ctx.saved_tensors
        shape, cl, scale, sr, aligned, dt = ctx.cfg
        g = H.roi_align_backward(grad_output, rois, scale, sr, aligned, shape, channels_last=cl, roi_scale=roi_scale)
        return g.to(dt), None, None, None, None, None, None, None


def roi_align(feat, rois, output_size, spatial_scale, sampling_ratio, aligned, roi_scale=None, out_dtype=None):
    return _RoIAlign.apply(feat, rois, tuple(output_size), float(spatial_scale), int(sampling_ratio), bool(aligned),
                           roi_scale, out_dtype or feat.dtype)


class _CosineLogits(Function):
    """logits = (T * z/||z||) @ Wn^T (+ bias); Wn (K1,D) already L2-normalised with its zero
    background row, WnT (D, K1 padded) its transpose (open_vocabulary_classifier.py:91-104)."""

    @staticmethod
    def forward(ctx, z, wn, wnT, temperature, normalize, bias_vec):
        rs = H.row_l2norm_scale(z, temperature) if normalize else None
        if _x2_mode():  # z is a real fp32 tensor here (the norm reads it): encode it and the class matrix, 3-MFMA products
            logits = H.gemm_nt(H.x2_encode(z), H.x2_cached(wn), x2=True, row_scale=rs, bias=bias_vec, out_dtype=torch.float32)
        else:
            logits = H.gemm_nt(z, wn, row_scale=rs, bias=bias_vec, out_dtype=torch.float32)
        ctx.save_for_backward(z, wnT)
        ctx.cfg = (temperature, normalize, bias_vec is not None)
        ctx.x3 = H.x3_active()
        ctx.bwd = _bwd_split() if ctx.x3 == "x2" else frozenset()
        return logits

    @staticmethod
    @once_differentiable
    def backward(ctx, dl):
        with H.x3_mode(_no_split(ctx.x3)):
            return _CosineLogits._backward(ctx, dl)

    @staticmethod
    def _backward(ctx, dl):
        z, wnT = ctx.saved_tensors
        temperature, normalize, has_bias = ctx.cfg
        cd = z.dtype
        split = "dx" in ctx.bwd and wnT.dtype == torch.float32  # "parity_train": the input gradient keeps the split
        if ctx.x3 in ("fwd", "x2") and wnT.dtype == torch.float32 and not split:  # forward-only split: the backward class matrix in bf16
            wnT = H.cast(wnT, torch.bfloat16)
            cd = torch.bfloat16
        dl = _contig2d(dl)
        M, K1 = dl.shape
        dA, _ = H.mask_transpose(dl, None, 1.0, cd, want_plain=True, want_t=False, ld_plain=wnT.size(1))
        with (H.x3_mode("full") if split else contextlib.nullcontext()):
            u = H.gemm_nt(dA, wnT, out_dtype=torch.float32)  # (M,D) = dL/d(zn)
        dz = H.row_l2norm_backward(z, u, temperature, relu_mask=False) if normalize else u
        if z.dtype != torch.float32:
            dz = H.cast(dz, z.dtype)
        db = None
        if has_bias and ctx.needs_input_grad[5]:
            seg = H.const_tensor((0, M), torch.int32, z.device)
            db = H.segment_colsum(dl, seg).view(K1)
        return dz, None, None, None, None, db


def cosine_logits(z, wn, wnT, temperature, normalize=True, bias_vec=None):
    return _CosineLogits.apply(z, wn, wnT, float(temperature), bool(normalize), bias_vec)


class _MilScores(Function):
    @staticmethod
    def forward(ctx, logits2k, seg_offsets, K):
        scores, P, Q = H.mil_forward(logits2k, seg_offsets, K)
        ctx.save_for_backward(P, Q, seg_offsets)
        ctx.K = K
        ctx.tail = H.tail_rows_active() is not None  # (the backward runs on autograd's thread: no thread-local there)
        return scores

    @staticmethod
    @once_differentiable
    def backward(ctx, dscores):
        P, Q, seg = ctx.saved_tensors
        return H.mil_backward(dscores, P, Q, seg, ctx.K, tail=ctx.tail), None, None


def mil_scores(logits2k, seg_offsets, K):
    """softmax(C, dim=1) * softmax(D, dim=0) per image segment on logits (M,2K) = [C | D]."""
    return _MilScores.apply(_contig2d(logits2k), seg_offsets, int(K))


class _ImageBCE(Function):
    @staticmethod
    def forward(ctx, scores, seg_offsets, labels_onehot, norm):
        loss, img, dS = H.image_bce_forward(scores, seg_offsets, labels_onehot, norm)
        ctx.save_for_backward(dS, seg_offsets)
        ctx.M = scores.size(0)
        ctx.tail = H.tail_rows_active() is not None
        ctx.mark_non_differentiable(img)
        return loss.view(()), img

    @staticmethod
    @once_differentiable
    def backward(ctx, gout, _gimg):
        dS, seg = ctx.saved_tensors
        return H.image_bce_backward(dS, seg, ctx.M, gout.reshape(1).contiguous(), tail=ctx.tail), None, None, None


def image_bce(scores, seg_offsets, labels_onehot, norm):
    """-> (loss scalar, clamped image-level scores (N,K))."""
    # (the kernel reads a DENSE (M,K) matrix: the one-class head hands in a column view of its two-column scores)
    return _ImageBCE.apply(scores.contiguous(), seg_offsets, labels_onehot.to(torch.float32).contiguous(), float(norm))


class _WeightedCE(Function):
    @staticmethod
    def forward(ctx, logits, gt_classes, weights, weighted):
        loss, dl, accum = H.weighted_ce_forward(logits, gt_classes, weights, weighted)
        ctx.save_for_backward(dl, accum)
        return loss.view(())

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        dl, accum = ctx.saved_tensors
        H.scale_by_device_scalar(dl, gout.reshape(1).contiguous(), accum[1:2])
        return dl, None, None, None


def weighted_cross_entropy(logits, gt_classes, weights=None, weighted=True):
    return _WeightedCE.apply(_contig2d(logits), gt_classes.contiguous(),
                             None if weights is None else weights.to(torch.float32).contiguous(), bool(weighted))


class _WeightedL1Box(Function):
    @staticmethod
    def forward(ctx, pred, proposal_boxes, gt_boxes, gt_classes, weights, K, bbox_weights, beta, weighted):
        loss, dp = H.weighted_l1_box_forward(pred, proposal_boxes, gt_boxes, gt_classes, weights, K, bbox_weights,
                                             beta, weighted)
        ctx.save_for_backward(dp)
        return loss.view(())

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        (dp,) = ctx.saved_tensors
        H.scale_by_device_scalar(dp, gout.reshape(1).contiguous(), None)
        return dp, None, None, None, None, None, None, None, None


def weighted_l1_box_loss(pred, proposal_boxes, gt_boxes, gt_classes, weights, K, bbox_weights, beta, weighted=True):
    return _WeightedL1Box.apply(_contig2d(pred), proposal_boxes.to(torch.float32).contiguous(),
                                gt_boxes.to(torch.float32).contiguous(), gt_classes.contiguous(),
                                None if weights is None else weights.to(torch.float32).contiguous(), int(K),
                                tuple(bbox_weights), float(beta), bool(weighted))


class _DataAware(Function):
    @staticmethod
    def forward(ctx, gap, W1, b1, W2, b2, E):
        daf, h1, h2 = H.data_aware_forward(gap, W1, b1, W2, b2, E)
        ctx.save_for_backward(gap, W2, E, h1, h2)
        return daf

    @staticmethod
    @once_differentiable
    def backward(ctx, ddaf):
        gap, W2, E, h1, h2 = ctx.saved_tensors
        dW1, db1, dW2, db2, dE = H.data_aware_backward(ddaf, gap, W2, E, h1, h2)
        return None, dW1, db1, dW2, db2, dE


def data_aware_features(gap, W1, b1, W2, b2, E):
    return _DataAware.apply(gap, W1, b1, W2, b2, E)


class _GapNHWC(Function):
    """Global average pool of an NHWC map -> (N, C) fp32.  The backward (the mean's gradient spread over the pixels) only
    runs when a backbone stage is trainable (MODEL.BACKBONE.FREEZE_AT < 5): a broadcast, left to torch."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape, ctx.dtype = tuple(x.shape), x.dtype
        return H.global_avgpool_nhwc(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        n, h, w, c = ctx.shape
        return (dy.to(torch.float32) / float(h * w)).view(n, 1, 1, c).expand(n, h, w, c).to(ctx.dtype)


def global_avgpool_nhwc(x):
    return _GapNHWC.apply(x) if x.requires_grad else H.global_avgpool_nhwc(x)


class _AddGroupRows(Function):
    @staticmethod
    def forward(ctx, x, add, row_group, seg_offsets):
        ctx.save_for_backward(seg_offsets)
        return H.add_group_rows(x, row_group, add, x2=_x2_mode())  # "parity": x and the result are bf16x2

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (seg,) = ctx.saved_tensors
        dadd = H.segment_colsum(_contig2d(dy), seg) if ctx.needs_input_grad[1] else None
        return dy, dadd, None, None


def add_group_rows(x, add, row_group, seg_offsets):
    """x[m] + add[row_group[m]] (box_features += data_aware_features without the per-proposal repeat)."""
    return _AddGroupRows.apply(x, add, row_group, seg_offsets)
