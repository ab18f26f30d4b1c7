"""Raw (non-autograd) Python fronts of the C-ABI kernels: tensors in, tensors out.

Every function enqueues on torch's current HIP stream and never synchronises.  There is no
CPU fallback: CPU tensors or a missing library raise RuntimeError.
"""
import ctypes as C
import contextlib
import os
import threading

import torch

from .. import _lib
from .._lib import BF16, BF16X2, BF16X2P, F16MX, F32, NCHW, NHWC, GemmDesc, check, dtype_code, lib, ptr, require_gpu, stream


_CONST_CACHE = {}


class _ConstOverride(threading.local):
    """While a training step is being CAPTURED as a HIP graph (engine/trainer.py:_StepGraph), the index tensors that
    change from step to step (per-image segment offsets, row -> image map) must be the graph's static input buffers, not
    value-keyed constants: the capture registers {(values, dtype): static tensor} here for its own thread."""
    table = None


_OVERRIDE = _ConstOverride()


class const_override:
    def __init__(self, table):
        self.table = table

    def __enter__(self):
        self.prev, _OVERRIDE.table = _OVERRIDE.table, self.table
        return self

    def __exit__(self, *exc):
        _OVERRIDE.table = self.prev
        return False


class _TailRows(threading.local):
    """A captured step graph runs on a BUCKETED row count: behind the last image's proposals sit padding rows that belong
    to no segment.  While this is set (to the 1-element int32 device tensor holding the step's REAL row count), the fronts
    below allocate what the segment kernels leave untouched as zeros / ignore labels, so that a padding row carries a zero
    gradient and label -1, and row-count normalisers read the real count from memory."""
    rows_true = None


_TAIL = _TailRows()


class tail_rows:
    def __init__(self, rows_true):
        self.rows_true = rows_true

    def __enter__(self):
        self.prev, _TAIL.rows_true = _TAIL.rows_true, self.rows_true
        return self

    def __exit__(self, *exc):
        _TAIL.rows_true = self.prev
        return False


def tail_rows_active():
    return _TAIL.rows_true


def _alloc(shape, dtype, device, tail=None):
    """torch.empty, or zeros when padding rows must read as zero."""
    return (torch.zeros if (tail if tail is not None else _TAIL.rows_true is not None) else torch.empty)(
        shape, dtype=dtype, device=device)


def const_tensor(values, dtype, device):
    """Small constant index tensors (segment offsets, sizes, row->image maps) keyed by value: built and
    copied to the device once, so steady-state steps issue no tiny blocking H2D copies."""
    if _OVERRIDE.table is not None:
        t = _OVERRIDE.table.get((tuple(values), dtype))
        if t is not None:
            return t
    key = (tuple(values), dtype, str(device))
    t = _CONST_CACHE.get(key)
    if t is None:
        if len(_CONST_CACHE) > 4096:
            _CONST_CACHE.clear()
        t = torch.tensor(list(values), dtype=dtype, device=device)
        _CONST_CACHE[key] = t
    return t


class _PinnedRing:
    """Small page-locked staging ring for per-step host data (image-level labels ...): a pageable
    `tensor.to(device)` blocks the host until the stream drains, which would serialise host and GPU once per
    step; copies from pinned memory are truly asynchronous.  A slot is reused only after its copy event."""

    def __init__(self, slots=16, nbytes=1 << 16):
        self.bufs = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self.events = [None] * slots
        self.i = 0

    def stage(self, t_cpu, device):
        nb = t_cpu.numel() * t_cpu.element_size()
        if nb == 0 or nb > self.bufs[0].numel():
            return t_cpu.to(device)
        k = self.i
        self.i = (self.i + 1) % len(self.bufs)
        if self.events[k] is not None:
            self.events[k].synchronize()
        view = self.bufs[k][:nb].view(t_cpu.dtype).view(t_cpu.shape)
        view.copy_(t_cpu)
        out = view.to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev
        return out


_RING = None


def h2d_small(t_cpu, device):
    """Asynchronous host->device copy of a small CPU tensor through the pinned staging ring."""
    global _RING
    if not torch.device(device).type == "cuda":
        return t_cpu.to(device)
    if _RING is None:
        _RING = _PinnedRing()
    return _RING.stage(t_cpu.contiguous(), device)


def feature_layout(feat):
    """(layout code, N, C, H, W) of a logical-NCHW feature tensor; NHWC == torch.channels_last."""
    if feat.dim() != 4:
        raise RuntimeError(f"wsovod_hip: expected a 4-D (N,C,H,W) feature map, got {tuple(feat.shape)}")
    N, Cc, H, W = feat.shape
    if feat.is_contiguous():
        return NCHW, N, Cc, H, W
    if feat.is_contiguous(memory_format=torch.channels_last):
        return NHWC, N, Cc, H, W
    raise RuntimeError("wsovod_hip: feature map must be contiguous (NCHW) or channels_last (NHWC)")


def _rois_f32(rois):
    if rois.dim() != 2 or rois.size(1) != 5:
        raise RuntimeError(f"wsovod_hip: rois must be (R,5), got {tuple(rois.shape)}")
    return rois.to(torch.float32).contiguous()


def cat_rows(tensors):
    """torch.cat(tensors, 0) without the copy when the inputs already are consecutive row blocks of one buffer (the
    per-image slices of a kernel's packed output): returns a view spanning them."""
    tensors = list(tensors)
    t0 = tensors[0]
    if len(tensors) == 1:
        return t0
    if t0.is_contiguous() and not t0.requires_grad and t0.dim() >= 1:
        base, inner, per_row = t0.untyped_storage().data_ptr(), t0.shape[1:], 1
        for d in inner:
            per_row *= d
        off, rows, ok = t0.storage_offset(), 0, True
        for t in tensors:
            if (t.requires_grad or t.dtype != t0.dtype or t.shape[1:] != inner or not t.is_contiguous()
                    or t.untyped_storage().data_ptr() != base or t.storage_offset() != off + rows * per_row):
                ok = False
                break
            rows += t.shape[0]
        if ok and per_row > 0:
            return t0.as_strided((rows,) + tuple(inner), t0.stride(), off)
    return torch.cat(tensors, dim=0)


def format_rois(boxes, seg_offsets, objectness=None):
    """(M,4) concatenated boxes + (G+1) int32 offsets -> ((M,5) pooler-format rois, objectness + 1 or None)."""
    require_gpu(boxes, seg_offsets, objectness)
    boxes = boxes.detach().to(torch.float32).contiguous()
    M, G = boxes.size(0), seg_offsets.numel() - 1
    rois = torch.empty((M, 5), dtype=torch.float32, device=boxes.device)
    scale = None
    if objectness is not None:
        objectness = objectness.detach().to(torch.float32).contiguous()
        scale = torch.empty((M,), dtype=torch.float32, device=boxes.device)
    check(lib().wsovod_format_rois(ptr(boxes), ptr(seg_offsets), G, M, ptr(objectness), ptr(rois), ptr(scale), stream()),
          "format_rois")
    return rois, scale


def x2_hi_pop(x):
    """The plain bf16 rounding the pooler wrote next to the bf16x2 tensor `x` (or None): the first FC layer keeps it for
    its weight-gradient contraction, which would otherwise fetch half lines out of the bf16x2 rows.  The copy rides on
    the pooled tensor OBJECT (and on the views the box head makes of it: `x2_hi_of`), never on a pointer-keyed table."""
    hi = x2_hi_of(x)
    return hi if (hi is not None and hi.numel() == x.numel()) else None


def x2_hi_of(x):
    """The bf16 copy attached to `x` or to the tensor `x` is a view of (flatten / view keep `_base`)."""
    hi = getattr(x, "_x2_hi", None)
    if hi is None and getattr(x, "_base", None) is not None and x._base.data_ptr() == x.data_ptr() \
            and x._base.numel() == x.numel():
        hi = getattr(x._base, "_x2_hi", None)
    return hi


def _x2_hi_alloc(out):
    hi = torch.empty(out.shape, dtype=torch.bfloat16, device=out.device)
    out._x2_hi = hi
    return hi


def _x2_planar_ok(R, row_values):
    """The planar form is addressed through one buffer resource per 256-row tile: hi plane + 256 rows must stay below 2 GiB
    (R18: ~83 images x 512 proposals per step; R50 / 1024 proposals: 10 images).  Larger steps keep the interleaved layout +
    the plain bf16 copy (round 4's form)."""
    return X2_PLANAR and (R * row_values) % 8 == 0 and (R + 256) * row_values * 2 < (1 << 31)


def _x2_planar_tag(out):
    """`out` (an fp32-typed carrier) was written as PLANAR bf16x2 (include/wsovod_hip.h: WSOVOD_BF16X2P): its first half is
    the bf16 matrix of hi values -- attached as the tensor's plain bf16 rounding (`x2_hi_of`), no copy -- its second half
    the lo values.  The first FC layer's forward reads both planes, its weight gradient the hi plane."""
    n = out.numel()
    out._x2_hi = out.view(-1).view(torch.bfloat16)[:n].view(out.shape)
    out._x2_planar = True
    return out


def x2_planar_of(x):
    """True when `x` (or the tensor it is a whole view of) is a planar bf16x2 carrier."""
    if x is None or x.dtype != torch.float32:
        return False  # (a carrier is float32-typed: the bf16 view of its hi plane is a plain bf16 matrix)
    if getattr(x, "_x2_planar", False):
        return True
    b = getattr(x, "_base", None)
    return bool(b is not None and b.data_ptr() == x.data_ptr() and b.numel() == x.numel() and getattr(b, "_x2_planar", False))


def _refuse_undeclared_planar(who, *tensors, declared=False):
    """The planar layout changes what the BYTES of a carrier mean and is recorded on the tensor object only: a consumer that
    reads interleaved bf16x2 must refuse a planar carrier instead of contracting garbage."""
    if declared:
        return
    for t in tensors:
        if t is not None and x2_planar_of(t):
            raise RuntimeError(f"wsovod_hip {who}: got a PLANAR bf16x2 carrier where the interleaved layout is read "
                               "(only the first FC layer's forward takes a_planar=True; x2_to_f32 decodes either layout)")


def x2_to_f32(x):
    """fp32 values of a bf16x2 tensor in either layout (tests, debugging)."""
    if x2_planar_of(x):
        n = x.numel()
        flat = x.reshape(-1).view(torch.bfloat16)
        return (flat[:n].float() + flat[n:].float()).view(x.shape)
    return x2_decode(x.reshape(x.shape[0], -1)).view(x.shape)  # (groups of 32 run along the flattened row)


# round 5: with `want_hi` (training, "parity") the poolers write PLANAR bf16x2 instead of the interleaved layout plus a plain
# bf16 copy (WSOVOD_X2_PLANAR=0: the round-4 form, for A/B runs)
# (only the LEAN two-phase tile reads the planar form: WSOVOD_G8_LEAN=0 therefore also selects the round-4 layout)
MX = "f16mx"  # format tag of the block-scaled parity format (include/wsovod_hip.h: WSOVOD_F16MX); carriers are float32-typed


def mx_of(x):
    """True when `x` (or the tensor it is a whole view of) was written as a unit-scale f16mx carrier."""
    if x is None or x.dtype != torch.float32:
        return False
    if getattr(x, "_mx", False):
        return True
    b = getattr(x, "_base", None)
    return bool(b is not None and b.data_ptr() == x.data_ptr() and b.numel() == x.numel() and getattr(b, "_mx", False))


X2_PLANAR = os.environ.get("WSOVOD_X2_PLANAR", "1") != "0" and os.environ.get("WSOVOD_G8_LEAN", "1") != "0"
POISON_OUTPUTS = os.environ.get("WSOVOD_POISON_OUTPUTS", "0") == "1"
POISON_BYTE = 0x7F


def _out_empty(shape, dtype, device):
    """Output buffer a kernel must cover completely (pool values / argmax indices are allocated uninitialised).
    POISON_OUTPUTS (tests, `WSOVOD_POISON_OUTPUTS=1`): every byte is 0x7f first, so an element a kernel path skipped
    reads as 0x7f7f7f7f (3.4e38 / 2139062143) instead of whatever the allocator left there."""
    t = torch.empty(shape, dtype=dtype, device=device)
    if POISON_OUTPUTS and t.numel():
        t.view(torch.uint8).fill_(POISON_BYTE)
    return t


def roi_pool_forward(feat, rois, spatial_scale, output_size, roi_scale=None, out_dtype=None, need_argmax=True,
                     want_hi=False):
    """RoI max pool -> (out (R,C,ph,pw), argmax int32 or None).  want_hi (bf16x2 output only): also write the plain
    bf16 rounding, fetched later with x2_hi_pop(out)."""
    require_gpu(feat, rois, roi_scale)
    layout, N, Cc, H, W = feature_layout(feat)
    rois = _rois_f32(rois)
    ph, pw = output_size
    R = rois.size(0)
    out_dtype = out_dtype or feat.dtype
    out = _out_empty((R, Cc, ph, pw), storage_dtype(out_dtype), feat.device)
    argmax = _out_empty((R, Cc, ph, pw), torch.int32, feat.device) if need_argmax else None
    if roi_scale is not None:
        roi_scale = roi_scale.to(torch.float32).contiguous()
    planar = bool(want_hi and out_dtype == X2 and R > 0 and _x2_planar_ok(R, Cc * ph * pw))
    hi = _x2_hi_alloc(out) if (want_hi and out_dtype in (X2, MX) and R > 0 and not planar) else None
    if out_dtype == MX:
        out._mx = True  # a unit-scale f16mx carrier (with, in training, its plain bf16 rounding for fc1's weight gradient)
    # scratch for the map's 2x2 maxima (0 bytes: this shape keeps the cell scan; include/wsovod_hip.h)
    ws_bytes = int(lib().wsovod_roi_pool_workspace_bytes(dtype_code(feat.dtype), layout, R, N, Cc, H, W, ph, pw,
                                                         int(need_argmax))) if R > 0 else 0
    m2 = _m2_take(feat, ws_bytes) if ws_bytes > 0 else None  # the map's 2x2 maxima written with the GAP of the same map
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=feat.device) if ws_bytes > 0 and m2 is None else None
    entry = lib().wsovod_roi_pool_forward_ws if m2 is None else lib().wsovod_roi_pool_forward_m2
    check(entry(
        ptr(feat), dtype_code(feat.dtype), layout, ptr(rois), ptr(roi_scale), R, N, Cc, H, W, ph, pw,
        C.c_float(spatial_scale), ptr(out), BF16X2P if planar else fmt_code(out_dtype), ptr(argmax), ptr(hi),
        ptr(ws if m2 is None else m2), ws_bytes, stream()), "roi_pool_forward")
    if planar:
        _x2_planar_tag(out)
    return out, argmax


def roi_loop_pool_forward(feat, rois, spatial_scale, output_size, context_ratio=1.8):
    """The reference's 3-output ROILoopPool: -> (out (3R,C,ph,pw) fp32 = [region | frame | context], argmax int32)."""
    require_gpu(feat, rois)
    if feat.dim() == 4 and not feat.is_contiguous(memory_format=torch.channels_last):
        # the kernel runs channels-per-lane on NHWC maps (the HIP backbone's layout); the reference's NCHW tensors are
        # brought into it first, as its own op makes its input `.contiguous()` (ROILoopPool_cuda.cu:293)
        feat = feat.contiguous(memory_format=torch.channels_last)
    layout, N, Cc, Hh, Ww = feature_layout(feat)
    if layout != NHWC:  # (a 1 x 1 map or one channel is both NCHW- and NHWC-contiguous)
        layout = NHWC
    rois = _rois_f32(rois)
    ph, pw = output_size
    R = rois.shape[0]
    out = _out_empty((3 * R, Cc, ph, pw), torch.float32, feat.device)
    arg = _out_empty((3 * R, Cc, ph, pw), torch.int32, feat.device)
    check(lib().wsovod_roi_loop_pool_forward(ptr(feat), dtype_code(feat.dtype), layout, ptr(rois), R, N, Cc, Hh, Ww, ph,
                                             pw, C.c_float(spatial_scale), C.c_float(context_ratio), ptr(out), ptr(arg),
                                             stream()), "roi_loop_pool_forward")
    return out, arg


def roi_pool_backward(grad_out, rois, argmax, input_shape, channels_last=False, roi_scale=None):
    require_gpu(grad_out, rois, argmax)
    N, Cc, H, W = input_shape
    rois = _rois_f32(rois)
    grad_out = grad_out.to(torch.float32).contiguous()
    R, _, ph, pw = grad_out.shape
    mf = torch.channels_last if channels_last else torch.contiguous_format
    grad_in = torch.zeros((N, Cc, H, W), dtype=torch.float32, device=grad_out.device).contiguous(memory_format=mf)
    check(lib().wsovod_roi_pool_backward(
        ptr(grad_out), ptr(rois), ptr(roi_scale), ptr(argmax), R, N, Cc, H, W, ph, pw,
        NHWC if channels_last else NCHW, ptr(grad_in), stream()), "roi_pool_backward")
    return grad_in


def roi_align_forward(feat, rois, spatial_scale, output_size, sampling_ratio, aligned, roi_scale=None,
                      out_dtype=None, want_hi=False):
    require_gpu(feat, rois, roi_scale)
    layout, N, Cc, H, W = feature_layout(feat)
    rois = _rois_f32(rois)
    ph, pw = output_size
    R = rois.size(0)
    out_dtype = out_dtype or feat.dtype
    out = _out_empty((R, Cc, ph, pw), storage_dtype(out_dtype), feat.device)
    if roi_scale is not None:
        roi_scale = roi_scale.to(torch.float32).contiguous()
    planar = bool(want_hi and out_dtype == X2 and R > 0 and _x2_planar_ok(R, Cc * ph * pw))
    hi = _x2_hi_alloc(out) if (want_hi and out_dtype in (X2, MX) and R > 0 and not planar) else None
    if out_dtype == MX:
        out._mx = True
    check(lib().wsovod_roi_align_forward_x2hi(
        ptr(feat), dtype_code(feat.dtype), layout, ptr(rois), ptr(roi_scale), R, N, Cc, H, W, ph, pw,
        C.c_float(spatial_scale), int(sampling_ratio), int(bool(aligned)), ptr(out),
        BF16X2P if planar else fmt_code(out_dtype), ptr(hi), stream()), "roi_align_forward")
    if planar:
        _x2_planar_tag(out)
    return out


def roi_align_backward(grad_out, rois, spatial_scale, sampling_ratio, aligned, input_shape, channels_last=False,
                       roi_scale=None):
    require_gpu(grad_out, rois)
    N, Cc, H, W = input_shape
    rois = _rois_f32(rois)
    grad_out = grad_out.to(torch.float32).contiguous()
    R, _, ph, pw = grad_out.shape
    mf = torch.channels_last if channels_last else torch.contiguous_format
    grad_in = torch.zeros((N, Cc, H, W), dtype=torch.float32, device=grad_out.device).contiguous(memory_format=mf)
    check(lib().wsovod_roi_align_backward(
        ptr(grad_out), ptr(rois), ptr(roi_scale), R, N, Cc, H, W, ph, pw, C.c_float(spatial_scale),
        int(sampling_ratio), int(bool(aligned)), NHWC if channels_last else NCHW, ptr(grad_in), stream()),
        "roi_align_backward")
    return grad_in


# ---------------------------------------------------------------------------------------
# bf16x2 activations (MODEL.HIP.PRECISION = "parity", include/wsovod_hip.h: WSOVOD_BF16X2).  A bf16x2 tensor is carried
# as a torch.float32 tensor of the LOGICAL shape whose 4-byte slots hold (hi, lo) bf16 pairs in 32-value groups -- same
# shapes, strides and byte sizes as fp32, so views / flatten / autograd's shape checks all work, but its numbers are
# meaningless to torch ops: only the kernels that take the `X2` format tag may read it.
# ---------------------------------------------------------------------------------------
X2 = "bf16x2"  # format tag accepted wherever a kernel front takes an `out_dtype` / `x2=` argument


def storage_dtype(fmt):
    return torch.float32 if fmt in (X2, "f16mx") else fmt


def fmt_code(fmt):
    return BF16X2 if fmt == X2 else F16MX if fmt == "f16mx" else dtype_code(fmt)


def x2_encode(src, out=None):
    """fp32 (rows, cols) with a contiguous last dim, cols % 32 == 0 -> bf16x2 tensor of the same shape."""
    require_gpu(src, out)
    if src.dtype != torch.float32 or src.dim() != 2 or src.stride(1) != 1 or src.size(1) % 32:
        raise RuntimeError("x2_encode: source must be a 2-D fp32 matrix with a contiguous last dim of a multiple of 32 columns")
    rows, cols = src.shape
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.float32, device=src.device)
    check(lib().wsovod_bf16x2_encode(ptr(src), src.stride(0), rows, cols, ptr(out), out.stride(0), stream()), "bf16x2_encode")
    return out


def x2_decode(src):
    """bf16x2 (rows, cols) -> the fp32 values hi + lo (tests, debugging)."""
    require_gpu(src)
    _refuse_undeclared_planar("x2_decode", src)  # (x2_to_f32 reads either layout)
    rows, cols = src.shape
    out = torch.empty((rows, cols), dtype=torch.float32, device=src.device)
    check(lib().wsovod_bf16x2_decode(ptr(src), src.stride(0), rows, cols, ptr(out), out.stride(0), stream()), "bf16x2_decode")
    return out


def mx_encode(src, nseg=1, unit=False, tensor_byte=None):
    """fp32 (rows, cols), cols % (32 nseg) == 0 -> (f16mx carrier (rows, cols) float32-typed, scales (rows, nseg) uint8: one
    tied E8M0 scale per row segment): include/wsovod_hip.h, wsovod_f16mx_encode.  unit: the activations' form, no scales
    (-> (carrier, None)).  tensor_byte: a 1-element uint8 DEVICE tensor = ONE scale for the tensor (wsovod_f16mx_encode_with)."""
    require_gpu(src, tensor_byte)
    src = src.contiguous()
    rows, cols = src.shape
    out = torch.empty((rows, cols), dtype=torch.float32, device=src.device)
    if tensor_byte is not None:
        scales = torch.empty((rows, 1), dtype=torch.uint8, device=src.device)
        check(lib().wsovod_f16mx_encode_with(ptr(src), src.stride(0), rows, cols, ptr(out), out.stride(0), ptr(scales),
                                             ptr(tensor_byte), stream()), "f16mx_encode_with")
        return out, scales
    scales = None if unit else torch.empty((rows, nseg), dtype=torch.uint8, device=src.device)
    check(lib().wsovod_f16mx_encode(ptr(src), src.stride(0), rows, cols, int(nseg), ptr(out), out.stride(0), ptr(scales),
                                    stream()), "f16mx_encode")
    return out, scales


def mx_decode(carrier, scales=None):
    """(tests, tools) the three planes of an f16mx carrier as fp32: hi, q * 2^s, ql * 2^(s - 11); scales None = unit scale."""
    shape = carrier.shape
    carrier = carrier.reshape(-1, shape[-1])
    rows, cols = carrier.shape
    raw = carrier.contiguous().view(torch.uint8).view(rows, cols // 32, 128)
    hi = raw[:, :, :64].contiguous().view(torch.float16).float().view(rows, cols)
    nseg = 1 if scales is None else scales.shape[1]
    q = raw[:, :, 64:96].contiguous().view(torch.float8_e4m3fn).float().view(rows, nseg, -1)
    ql = raw[:, :, 96:].contiguous().view(torch.float8_e4m3fn).float().view(rows, nseg, -1)
    s = 1.0 if scales is None else torch.exp2(scales.float() - 127.0).unsqueeze(-1)
    return hi.view(shape), (q * s).view(shape), (ql * s * 2.0 ** -11).view(shape)


def mx_to_f32(carrier):
    """The values a unit-scale f16mx tensor stands for: hi + ql 2^-11 (tests, debugging)."""
    hi, _, ql = mx_decode(carrier)
    return hi + ql


def mx_from_x2(src):
    """interleaved bf16x2 tensor -> unit-scale f16mx tensor of the same shape (wsovod_f16mx_from_bf16x2)."""
    require_gpu(src)
    _refuse_undeclared_planar("mx_from_x2", src)
    assert src.dtype == torch.float32 and src.is_contiguous() and src.shape[-1] % 32 == 0
    out = torch.empty_like(src)
    check(lib().wsovod_f16mx_from_bf16x2(ptr(src), ptr(out), src.numel(), stream()), "f16mx_from_bf16x2")
    return out


def gemm_mx(A, a_scale, B, b_scale, *, bias=None, relu=False, dropout_p=0.0, dropout_seed=0, dropout_seed_add=None,
            alpha=1.0, out=None, out_dtype=torch.float32, out_bf16=None, residual=None, residual_fmt=None, conv=None, A2=None):
    """C = epilogue(A B^T) on f16mx operands (mx_encode): wsovod_gemm_f16mx.  a_scale None: A is a unit-scale (activation)
    carrier.  out_dtype: float32, bfloat16, X2 or MX (unit-scale f16mx); out_bf16: an optional (M, N) bfloat16 tensor that
    receives the plain bf16 copy of C.  residual (+ residual_fmt = X2 / MX for carriers).  conv: the geometry dict of gemm_nt
    on an NHWC unit-scale f16mx map (A2: the fused projection shortcut's input)."""
    require_gpu(A, a_scale, B, b_scale, bias, out, out_bf16, residual, A2)
    assert b_scale.dtype == torch.uint8 and b_scale.is_contiguous() and (a_scale is None or (a_scale.dtype == torch.uint8 and a_scale.is_contiguous()))
    d = GemmDesc()
    d.dtype_in = F16MX
    d.N, d.K = B.size(0), B.size(1)
    d.B, d.ldb = B.data_ptr(), _ld(B)
    if conv is not None:
        g = d.geom
        for k, v in conv.items():
            setattr(g, k, int(v))
        d.conv = 1
        d.M = g.n_img * g.Ho * g.Wo
        d.A, d.lda = A.data_ptr(), g.Cin
        if not A.is_contiguous():
            raise RuntimeError("wsovod_hip gemm_mx: the conv input must be NHWC-contiguous")
        if A2 is not None:
            if not A2.is_contiguous() or A2.numel() != d.M * A2.shape[-1]:
                raise RuntimeError("wsovod_hip gemm_mx: the fused shortcut input must be NHWC-contiguous (n_img, Ho, Wo, Cin2)")
            d.A2, d.Cin2 = A2.data_ptr(), int(A2.shape[-1])
    else:
        d.M = A.size(0)
        d.A, d.lda = A.data_ptr(), _ld(A)
    if out is None:
        out = torch.empty((d.M, d.N), dtype=storage_dtype(out_dtype), device=A.device)
    d.C, d.ldc = out.data_ptr(), _ld(out)
    d.dtype_c = fmt_code(out_dtype)
    d.alpha = alpha
    if bias is not None:
        d.bias = bias.data_ptr()
    if residual is not None:
        d.residual, d.ldr = residual.data_ptr(), _ld(residual)
        d.dtype_r = fmt_code(residual_fmt) if residual_fmt is not None else dtype_code(residual.dtype)
    d.relu = int(bool(relu))
    d.dropout_p = float(dropout_p)
    d.dropout_seed = int(dropout_seed)
    if dropout_seed_add is not None:
        require_gpu(dropout_seed_add)
        d.dropout_seed_add = dropout_seed_add.data_ptr()
    if out_bf16 is not None:
        assert out_bf16.dtype == torch.bfloat16 and out_bf16.shape == (d.M, d.N)
    check(lib().wsovod_gemm_f16mx(C.byref(d), ptr(a_scale), 1 if a_scale is None else int(a_scale.shape[1]), ptr(b_scale),
                                  int(b_scale.shape[1]), ptr(out_bf16), 0 if out_bf16 is None else _ld(out_bf16), stream()),
          "gemm_f16mx")
    return out


MX_WEIGHT_HEADROOM = 1  # binades between a trained weight's largest magnitude at its first encode and the q plane's 256


def mx_tensor_scale(t):
    """The per-tensor E8M0 byte (1-element uint8 DEVICE tensor) of a TRAINED weight's f16mx operand, derived from the tensor's
    largest magnitude (+ MX_WEIGHT_HEADROOM binades) without a host read at every FULL encode of the weight (the first use, a
    loaded checkpoint) and written IN PLACE into the tensor kept on the parameter (`_mx_scale`: captured step graphs and the
    optimizer's table keep its address).  Between full encodes the scale is fixed and the optimizer kernels re-encode the
    operand element-wise inside their update pass; e4m3's own exponent carries the rows and the growth of the weights
    (beyond 448 / 256 x 2^headroom of that maximum the cross terms saturate: their accuracy goes, not the product's)."""
    amax = t.detach().abs().amax().float().clamp_(min=2.0 ** -14)
    byte = (torch.floor(torch.log2(amax)) - 7 + MX_WEIGHT_HEADROOM + 127).clamp_(1, 254).to(torch.uint8).reshape(1)
    s = getattr(t, "_mx_scale", None)
    if s is None:
        s = byte
        try:
            t._mx_scale = s
        except AttributeError:
            pass
    else:
        s.copy_(byte)
    return s


def mx_cached(t, view_rows_cols=None, tensor_scale=False):
    """f16mx encoding (carrier, per-row scales) of a weight, cached on the tensor object and keyed by its version counter (frozen
    backbone weights: once, with their own row scales).  tensor_scale (trained weights): ONE scale for the tensor
    (mx_tensor_scale) -- the optimizer then refreshes the carrier in its update pass and re-stamps the cache, so that in steady
    state no encode pass runs."""
    key = (t._version, t.data_ptr(), view_rows_cols)
    c = getattr(t, "_mx_enc", None)
    if c is not None and c[0] == key:
        return c[1]
    src = t.detach()
    if tensor_scale and view_rows_cols is None and src.dim() == 2 and src.is_contiguous() and src.shape[1] % 32 == 0:
        byte = mx_tensor_scale(t)
        if c is not None and len(c) > 2 and c[2] and c[1][0].shape == src.shape:
            car, scales = c[1]  # (re-encoded in place: a captured graph / the optimizer's table keep the addresses)
        else:
            car = torch.empty_like(src)
            scales = torch.empty((src.shape[0], 1), dtype=torch.uint8, device=src.device)
        check(lib().wsovod_f16mx_encode_with(ptr(src), src.stride(0), src.shape[0], src.shape[1], ptr(car), car.stride(0),
                                             ptr(scales), ptr(byte), stream()), "f16mx_encode_with")
        out = (car, scales)
        try:
            t._mx_enc = (key, out, True)  # (True: ONE scale for the tensor -- what the optimizer kernels can refresh)
        except AttributeError:
            pass
        return out
    out = mx_encode(src.reshape(view_rows_cols) if view_rows_cols is not None else src)
    try:
        t._mx_enc = (key, out, False)
    except AttributeError:
        pass
    return out


def x2_cached(t, view_rows_cols=None):
    """bf16x2 encoding of a weight, cached on the tensor object and keyed by its version counter (one re-encode per
    optimizer step at most; frozen backbone weights once)."""
    key = (t._version, t.data_ptr(), view_rows_cols)
    c = getattr(t, "_x2_enc", None)
    if c is not None and c[0] == key:
        return c[1]
    src = t.detach()
    out = x2_encode(src.reshape(view_rows_cols) if view_rows_cols is not None else src)
    try:
        t._x2_enc = (key, out)
    except AttributeError:
        pass
    return out


def stem_conv1_x2(images_u8, sizes, mean, std, w32_x2, bias):
    """uint8 (N,3,Hp,Wp) images -> relu(conv1 3x3/s2 (folded BN)) as (N,Ho,Wo,64) bf16x2 NHWC (three-MFMA products)."""
    require_gpu(images_u8, sizes, w32_x2, bias)
    assert w32_x2.dtype == torch.float32 and tuple(w32_x2.shape) == (64, 32) and w32_x2.is_contiguous()
    N, _, Hp, Wp = images_u8.shape
    Ho, Wo = (Hp - 1) // 2 + 1, (Wp - 1) // 2 + 1
    out = torch.empty((N, Ho, Wo, 64), dtype=torch.float32, device=images_u8.device)
    check(lib().wsovod_stem_conv1_x2(ptr(images_u8), ptr(sizes), _f3(mean), _f3(std), N, Hp, Wp, ptr(w32_x2), ptr(bias),
                                     ptr(out), stream()), "stem_conv1_x2")
    return out


# ---------------------------------------------------------------------------------------
# bf16x3 mode (MODEL.HIP.PRECISION = "bf16x3"): fp32 tensors everywhere, every contraction evaluated on the bf16 MFMA
# kernels as sum ah*bh + ah*bl + al*bh over operands split by wsovod_split3_bf16 (include/wsovod_hip.h).
# ---------------------------------------------------------------------------------------
import threading


class _X3StateT(threading.local):
    """Per-thread: a TTA / data-loader thread must neither see nor clobber the mode of the autograd thread."""
    active = False


_X3State = _X3StateT()


class x3_mode:
    """Context manager: fp32 x fp32 contractions issued inside go through the bf16x3 split.  Model entry points enter
    it when their precision is "bf16x3" (mode "full"), "parity" (mode "x2": bf16x2 activations, see above; fp32 x fp32
    contractions issued without the x2 tag -- the RPN head on the fp32 res5 map -- still take the split below) or "bf16x3f"
    (mode "fwd": the split only in the forward pass;
    the backward contractions then run as plain bf16 on casts of the saved fp32 tensors -- forward logits of fp32
    grade, gradients of the bf16 mode's grade).  Autograd Functions capture `x3_active()` in forward and act on it in
    backward (the autograd engine runs backward outside the forward's context)."""

    def __init__(self, on=True):
        self.on = ("full" if on is True else on) if on else False

    def __enter__(self):
        self.prev = _X3State.active
        _X3State.active = self.on
        return self

    def __exit__(self, *exc):
        _X3State.active = self.prev
        return False


class _MxStateT(threading.local):
    on = False


_MxState = _MxStateT()


class mx_mode:
    """Context manager (MODEL.HIP.PRECISION = "parity_mx"): inside the "x2" (parity) mode, the big forward contractions -- the
    res4 / res5 convs and the box head's FC layers -- take the block-scaled f16mx kernels (wsovod_gemm_f16mx) and the tensors
    between them travel as unit-scale f16mx carriers instead of bf16x2."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.prev = _MxState.on
        _MxState.on = self.on
        return self

    def __exit__(self, *exc):
        _MxState.on = self.prev
        return False


def mx_active():
    return bool(_MxState.on) and x3_active() == "x2"


def x3_active():
    return _X3State.active


def split3_bf16(src, side, stack_rows=False, rows_pad=None, out=None):
    """(rows, cols) fp32 (row stride >= cols) -> bf16 blocks [hi | hi | lo] (side 0, A operand) or [hi | lo | hi]
    (side 1, B operand): side by side along the columns -> (rows, 3 cols), or with stack_rows stacked along the rows
    -> (3 rows_pad, cols) (rows_pad >= rows, padding rows zero)."""
    require_gpu(src, out)
    if src.dtype != torch.float32 or src.dim() != 2 or src.stride(1) != 1:
        raise RuntimeError("split3_bf16: source must be a 2-D fp32 matrix with a contiguous last dim")
    rows, cols = src.shape
    if stack_rows:
        rp = rows if rows_pad is None else int(rows_pad)
        if out is None:
            out = (torch.zeros if rp != rows else torch.empty)((3 * rp, cols), dtype=torch.bfloat16, device=src.device)
        ld_dst, block = out.stride(0), rp * out.stride(0)
    else:
        cp = (cols + 7) // 8 * 8  # every block starts 16-byte aligned; padding columns are zero (contribute nothing)
        if out is None:
            out = (torch.zeros if cp != cols else torch.empty)((rows, 3 * cp), dtype=torch.bfloat16, device=src.device)
        ld_dst, block = out.stride(0), cp
    check(lib().wsovod_split3_bf16(ptr(src), src.stride(0), rows, cols, ptr(out), ld_dst, block, int(side), stream()),
          "split3_bf16")
    return out


def _split3_cached(t, side, view_rows_cols=None):
    """B-side operands are usually weights: the split is cached on the tensor object, keyed by its version counter
    (HipSGD bumps it after every update), so steady-state forwards re-split a weight once per step at most."""
    key = (t._version, t.data_ptr(), side, view_rows_cols)
    c = getattr(t, "_x3_split", None)
    if c is not None and c[0] == key:
        return c[1]
    src = t.detach()
    out = split3_bf16(src.reshape(view_rows_cols) if view_rows_cols is not None else src, side)
    try:
        t._x3_split = (key, out)
    except AttributeError:
        pass
    return out


def _gemm_nt_x3(A, B, conv=None, **kw):
    if conv is not None:
        cin = int(conv["Cin"])
        a3 = split3_bf16(A.reshape(-1, cin), 0)
        b3 = _split3_cached(B, 1, (B.numel() // cin, cin)).view(B.size(0), -1)
        conv = dict(conv, Cin=3 * cin)
        if conv.get("pool"):
            raise RuntimeError("bf16x3: the fused-pool 64-channel conv is a bf16-only kernel")
        return gemm_nt(a3, b3, conv=conv, **kw)
    return gemm_nt(split3_bf16(A, 0), _split3_cached(B, 1), **kw)


def _ld(t):
    if t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError("wsovod_hip gemm: operands must be 2-D with a contiguous last dim")
    return t.stride(0)


def gemm_nt(A, B, *, out=None, out_dtype=None, out_t=None, alpha=1.0, row_scale=None, bias=None, residual=None,
            relu=False, dropout_p=0.0, dropout_seed=0, row_group=None, group_add=None, mask_src=None,
            mask_scale=1.0, accumulate=False, M=None, N=None, K=None, conv=None, tile_hint=0, want_c=True, A2=None,
            x2=False, residual_x2=False, dropout_seed_add=None, a_planar=False):
    """C[M][N] = epilogue(sum_k A[m][k]*B[n][k]); see include/wsovod_hip.h for the epilogue order.
    a_planar (x2 only, plain GEMM): A is a PLANAR bf16x2 carrier (the poolers' training output: hi matrix, then lo matrix).

    A: (M,K) or, with `conv` (a dict of geometry), the NHWC input tensor.  B: (N,K).
    x2: A, B (and A2) are bf16x2 tensors (fp32-typed carriers, see above): three-MFMA products.  out_dtype = X2 asks
    for a bf16x2 output (N a multiple of 32); residual_x2: the residual is bf16x2.
    `out_t` is an optional (N, >=M) tensor that receives the transposed copy.
    `A2` (conv only): a second NHWC input (n_img, Ho, Wo, Cin2) contracted 1x1 in the same accumulation (the block's
    projection shortcut); B rows are then [W | Wshortcut].
    Returns `out` (or None if want_c is False).
    """
    require_gpu(A, B, out, out_t, row_scale, bias, residual, row_group, group_add, mask_src)
    _refuse_undeclared_planar("gemm_nt", A, declared=a_planar)
    _refuse_undeclared_planar("gemm_nt", B, A2)
    if x2 and (A.dtype != torch.float32 or B.dtype != torch.float32):
        raise RuntimeError("wsovod_hip gemm: bf16x2 operands travel as float32-typed tensors")
    if not x2 and _X3State.active and A.dtype == torch.float32 and B.dtype == torch.float32:
        if M is not None or N is not None or K is not None or A2 is not None:
            raise RuntimeError("bf16x3: explicit M/N/K overrides / a fused shortcut input are not supported")
        return _gemm_nt_x3(A, B, conv=conv, out=out, out_dtype=out_dtype or torch.float32, out_t=out_t, alpha=alpha,
                           row_scale=row_scale, bias=bias, residual=residual, relu=relu, dropout_p=dropout_p,
                           dropout_seed=dropout_seed, dropout_seed_add=dropout_seed_add, row_group=row_group,
                           group_add=group_add, mask_src=mask_src,
                           mask_scale=mask_scale, accumulate=accumulate, tile_hint=tile_hint, want_c=want_c)
    d = GemmDesc()
    d.dtype_in = BF16X2 if x2 else dtype_code(B.dtype)
    if A.dtype != B.dtype:
        raise RuntimeError(f"wsovod_hip gemm: A is {A.dtype} but B is {B.dtype}")
    d.N = B.size(0) if N is None else N
    d.K = B.size(1) if K is None else K
    d.B, d.ldb = B.data_ptr(), _ld(B)
    if conv is not None:
        g = d.geom
        for k, v in conv.items():
            setattr(g, k, int(v))
        d.conv = 1
        d.M = g.n_img * g.Ho * g.Wo
        d.A, d.lda = A.data_ptr(), g.Cin
        if A2 is not None:
            require_gpu(A2)
            if A2.dtype != A.dtype or not A2.is_contiguous() or A2.numel() != d.M * A2.shape[-1]:
                raise RuntimeError("wsovod_hip gemm: the fused shortcut input must be NHWC-contiguous (n_img, Ho, Wo, Cin2) "
                                   "in the dtype of A")
            d.A2, d.Cin2 = A2.data_ptr(), int(A2.shape[-1])
    else:
        d.M = A.size(0) if M is None else M
        d.A, d.lda = A.data_ptr(), _ld(A)
        if a_planar:
            if not x2 or not A.is_contiguous() or M is not None:
                raise RuntimeError("wsovod_hip gemm: a planar bf16x2 A is a whole contiguous bf16x2 carrier")
            d.a_plane_bytes = A.numel() * 2
    if want_c:
        if out is None:
            rows = d.M if not (conv is not None and d.geom.pool) else d.geom.n_img * (d.geom.Ho // 2) * (d.geom.Wo // 2)
            out = torch.empty((rows, d.N), dtype=storage_dtype(out_dtype or (X2 if x2 else A.dtype)), device=B.device)
        d.C, d.ldc = out.data_ptr(), _ld(out)
        d.dtype_c = BF16X2 if (out_dtype == X2 or (out_dtype is None and x2)) else dtype_code(out.dtype)
    if out_t is not None:
        d.Ct, d.ldct, d.dtype_ct = out_t.data_ptr(), _ld(out_t), dtype_code(out_t.dtype)
    d.alpha = alpha
    if row_scale is not None:
        d.row_scale = row_scale.data_ptr()
    if bias is not None:
        if bias.dtype != torch.float32:
            raise RuntimeError("wsovod_hip gemm: bias must be fp32")
        d.bias = bias.data_ptr()
    if residual is not None:
        d.residual, d.ldr = residual.data_ptr(), _ld(residual)
        d.dtype_r = BF16X2 if residual_x2 else dtype_code(residual.dtype)
    d.relu = int(bool(relu))
    d.dropout_p, d.dropout_seed = float(dropout_p), int(dropout_seed)
    if dropout_seed_add is not None:  # a 1-element int64 device tensor: the step term of the seed (graph-replayable)
        require_gpu(dropout_seed_add)
        assert dropout_seed_add.dtype == torch.int64 and dropout_seed_add.numel() == 1
        d.dropout_seed_add = dropout_seed_add.data_ptr()
    if group_add is not None:
        d.row_group, d.group_add, d.ld_ga = row_group.data_ptr(), group_add.data_ptr(), _ld(group_add)
    if mask_src is not None:
        d.mask_src, d.ldm, d.dtype_m = mask_src.data_ptr(), _ld(mask_src), dtype_code(mask_src.dtype)
    d.mask_scale = float(mask_scale)
    d.accumulate = int(bool(accumulate))
    d.tile_hint = int(tile_hint)
    check(lib().wsovod_gemm_nt(C.byref(d), stream()), "gemm_nt")
    return out


# ---------------------------------------------------------------------------------------
# elementwise / reduction kernels
# ---------------------------------------------------------------------------------------
def _f3(vals):
    return (C.c_float * 3)(*[float(v) for v in vals])


def preprocess_image(images_u8, sizes, mean, std):
    """(N,3,Hp,Wp) uint8 canvas + (N,2) int32 sizes -> normalised fp32 NCHW (zero outside each image)."""
    require_gpu(images_u8, sizes)
    N, _, Hp, Wp = images_u8.shape
    out = torch.empty((N, 3, Hp, Wp), dtype=torch.float32, device=images_u8.device)
    check(lib().wsovod_preprocess_image(ptr(images_u8), ptr(sizes), _f3(mean), _f3(std), N, Hp, Wp, ptr(out),
                                        stream()), "preprocess_image")
    return out


def stem_im2col(images_u8, sizes, mean, std, dtype):
    """-> ((N*Ho*Wo, 32) operand of the stem conv1 GEMM, Ho, Wo)."""
    require_gpu(images_u8, sizes)
    N, _, Hp, Wp = images_u8.shape
    Ho, Wo = (Hp - 1) // 2 + 1, (Wp - 1) // 2 + 1
    out = torch.empty((N * Ho * Wo, 32), dtype=dtype, device=images_u8.device)
    check(lib().wsovod_stem_im2col(ptr(images_u8), ptr(sizes), _f3(mean), _f3(std), N, Hp, Wp, ptr(out),
                                   dtype_code(dtype), stream()), "stem_im2col")
    return out, Ho, Wo


def stem_conv1(images_u8, sizes, mean, std, w32, bias):
    """uint8 (N,3,Hp,Wp) images -> relu(conv1 3x3/s2 (folded BN)) as (N,Ho,Wo,64) bf16 NHWC, no im2col operand."""
    require_gpu(images_u8, sizes, w32, bias)
    assert w32.dtype == torch.bfloat16 and tuple(w32.shape) == (64, 32) and w32.is_contiguous()
    N, _, Hp, Wp = images_u8.shape
    Ho, Wo = (Hp - 1) // 2 + 1, (Wp - 1) // 2 + 1
    out = torch.empty((N, Ho, Wo, 64), dtype=torch.bfloat16, device=images_u8.device)
    check(lib().wsovod_stem_conv1(ptr(images_u8), ptr(sizes), _f3(mean), _f3(std), N, Hp, Wp, ptr(w32), ptr(bias),
                                  ptr(out), stream()), "stem_conv1")
    return out


def maxpool2x2_nhwc(x, stride, zero_pad_br=False, x2=False):
    """x: (N,H,W,C) contiguous -> (N,Ho,Wo,C).  x2: x is a bf16x2 map (C a multiple of 32)."""
    require_gpu(x)
    N, H, W, Cc = x.shape
    Hin, Win = H + int(zero_pad_br), W + int(zero_pad_br)
    Ho, Wo = (Hin - 2) // stride + 1, (Win - 2) // stride + 1
    out = torch.empty((N, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
    check(lib().wsovod_maxpool2x2_nhwc(ptr(x), BF16X2 if x2 else dtype_code(x.dtype), N, H, W, Cc, stride, int(zero_pad_br),
                                       ptr(out), stream()), "maxpool2x2_nhwc")
    return out


def maxpool2x2_nhwc_backward(x, dout, stride, zero_pad_br=False, x2=False):
    """x: the pool's input (N,H,W,C) as the forward saw it; dout (N,Ho,Wo,C) fp32 -> din (N,H,W,C) fp32 (first-maximum
    routing, torch's max_pool2d rule)."""
    require_gpu(x, dout)
    N, H, W, Cc = x.shape
    dout = dout.contiguous()
    assert dout.dtype == torch.float32 and x.is_contiguous()
    din = torch.empty((N, H, W, Cc), dtype=torch.float32, device=x.device)
    check(lib().wsovod_maxpool2x2_nhwc_backward(ptr(x), BF16X2 if x2 else dtype_code(x.dtype), N, H, W, Cc, stride,
                                                int(zero_pad_br), ptr(dout), ptr(din), stream()), "maxpool2x2_nhwc_backward")
    return din


def _colsum_workspace(G, M, N, device):
    """fp32 scratch of the two-stage column sums (one partial per 128-row chunk and column): allocated per call from
    torch's caching allocator, so it belongs to the calling stream like any other temporary."""
    return torch.empty((max(1, int(lib().wsovod_colsum_workspace_floats(G, M, N))),), dtype=torch.float32, device=device)


# ---- the global average pool FUSED with the RoIPool pre-pass (round 5) -------------------------------------------------
# Inside `gap_with_pool_prepass(num_rois, pooled_size)` (the frozen part of the training forward, where the data-aware head's
# GAP and the RoI max pool read the same res5 map) global_avgpool_nhwc writes the map's stride-1 2x2 maxima in the same pass
# (wsovod_max2x2_gap_nhwc) and parks them; the next roi_pool_forward on THAT map (same storage, version and shape) takes
# them instead of running its own pre-pass.  WSOVOD_GAP_FUSE=0 switches it off (A/B runs).
GAP_FUSE = os.environ.get("WSOVOD_GAP_FUSE", "1") != "0"


class _M2State(threading.local):  # per thread, as the x3 mode and the want-hi flag: an eval / TTA thread calling the GAP or
    on = None                     # the pooler while the training thread is inside the context must not see (or steal) its map
    map = None


_M2 = _M2State()


@contextlib.contextmanager
def gap_with_pool_prepass(num_rois, pooled_size=(7, 7)):
    prev = _M2.on
    _M2.on = (int(num_rois), tuple(pooled_size)) if GAP_FUSE else None
    try:
        yield
    finally:
        _M2.on, _M2.map = prev, None


def _m2_key(x_nhwc_or_feat):
    t = x_nhwc_or_feat
    return (t.data_ptr(), t._version, t.dtype, t.numel())


def _m2_take(feat, ws_bytes):
    got, _M2.map = _M2.map, None
    if got is None or got[0] != _m2_key(feat) or got[1].numel() * got[1].element_size() < ws_bytes:
        return None
    return got[1]


def global_avgpool_nhwc(x):
    """x: (N,H,W,C) contiguous -> (N,C) fp32."""
    require_gpu(x)
    N, H, W, Cc = x.shape
    out = torch.empty((N, Cc), dtype=torch.float32, device=x.device)
    if _M2.on is not None and x.dtype in (torch.float32, torch.bfloat16) and x.is_contiguous() and N > 0:
        R, (ph, pw) = _M2.on
        code = dtype_code(x.dtype)
        # (the pooler's own rule: enough rois to re-read the map many times over, 7 bins wide, ...)
        if int(lib().wsovod_roi_pool_workspace_bytes(code, NHWC, R, N, Cc, H, W, ph, pw, 0)) > 0:
            nfl = int(lib().wsovod_max2x2_gap_workspace_floats(code, N, Cc, H, W))
            if nfl > 0:
                m2 = torch.empty_like(x)
                ws = torch.empty((nfl,), dtype=torch.float32, device=x.device)
                check(lib().wsovod_max2x2_gap_nhwc(ptr(x), code, N, Cc, H, W, ptr(m2), ptr(out), ptr(ws), stream()),
                      "max2x2_gap")
                _M2.map = (_m2_key(x), m2)
                return out
    ws = _colsum_workspace(N, N * H * W, Cc, x.device)
    check(lib().wsovod_global_avgpool_nhwc(ptr(x), dtype_code(x.dtype), N, H * W, Cc, ptr(out), ptr(ws), stream()), "gap")
    return out


def transpose_cast(src, dst_dtype, ld_dst=None, out=None):
    """(R,C) -> (C, ld_dst>=R) transposed copy in dst_dtype; padding columns are zero."""
    require_gpu(src)
    R, Cc = src.shape
    if out is None:
        ld = ld_dst or R
        out = torch.zeros((Cc, ld), dtype=dst_dtype, device=src.device) if ld != R else torch.empty(
            (Cc, R), dtype=dst_dtype, device=src.device)
    check(lib().wsovod_transpose_cast(ptr(src), dtype_code(src.dtype), _ld(src), R, Cc, ptr(out),
                                      dtype_code(out.dtype), _ld(out), stream()), "transpose_cast")
    return out


def cast(src, dst_dtype, out=None):
    require_gpu(src)
    src = src.contiguous()
    if out is None:
        out = torch.empty(src.shape, dtype=dst_dtype, device=src.device)
    check(lib().wsovod_cast(ptr(src), dtype_code(src.dtype), ptr(out), dtype_code(out.dtype), src.numel(), stream()),
          "cast")
    return out


def row_l2norm_scale(x, temperature, eps=1e-12):
    require_gpu(x)
    M, D = x.shape
    out = torch.empty((M,), dtype=torch.float32, device=x.device)
    check(lib().wsovod_row_l2norm_scale(ptr(x), dtype_code(x.dtype), _ld(x), M, D, C.c_float(temperature),
                                        C.c_float(eps), ptr(out), stream()), "row_l2norm_scale")
    return out


def row_l2norm_backward(z, u, temperature, eps=1e-12, relu_mask=True):
    require_gpu(z, u)
    M, D = z.shape
    dz = torch.empty((M, D), dtype=torch.float32, device=z.device)
    check(lib().wsovod_row_l2norm_backward(ptr(z), dtype_code(z.dtype), _ld(z), ptr(u), _ld(u), M, D,
                                           C.c_float(temperature), C.c_float(eps), int(relu_mask), ptr(dz), _ld(dz),
                                           stream()), "row_l2norm_backward")
    return dz


def segment_colsum(x, seg_offsets, out=None, accumulate=False):
    """x (M,N), seg_offsets int32 (G+1) -> (G,N) fp32 column sums per segment."""
    require_gpu(x, seg_offsets)
    M, N = x.shape
    G = seg_offsets.numel() - 1
    if out is None:
        out = torch.empty((G, N), dtype=torch.float32, device=x.device)
    ws = _colsum_workspace(G, M, N, x.device)
    check(lib().wsovod_segment_colsum(ptr(x), dtype_code(x.dtype), _ld(x), ptr(seg_offsets), G, M, N, ptr(out),
                                      _ld(out), int(accumulate), ptr(ws), stream()), "segment_colsum")
    return out


def scale_by_device_scalar(x, num=None, den=None):
    require_gpu(x, num, den)
    assert x.dtype == torch.float32 and x.is_contiguous()
    check(lib().wsovod_scale_by_device_scalar(ptr(x), x.numel(), ptr(num), ptr(den), stream()), "scale_by_scalar")
    return x


def sgd_momentum(param, grad, buf, lr, momentum, weight_decay, grad_scale=1.0, bf16_shadow=None):
    require_gpu(param, grad, buf, bf16_shadow)
    assert param.is_contiguous() and grad.is_contiguous() and buf.is_contiguous()
    check(lib().wsovod_sgd_momentum(ptr(param), ptr(grad), ptr(buf), param.numel(), C.c_float(lr),
                                    C.c_float(momentum), C.c_float(weight_decay), C.c_float(grad_scale),
                                    ptr(bf16_shadow), stream()), "sgd_momentum")


# ---------------------------------------------------------------------------------------
# MIL head kernels
# ---------------------------------------------------------------------------------------
def mil_forward(logits, seg_offsets, K):
    """logits (M,2K) fp32 [cls|det] -> (scores, P, Q) each (M,K)."""
    require_gpu(logits, seg_offsets)
    M = logits.size(0)
    G = seg_offsets.numel() - 1
    scores = _alloc((M, K), torch.float32, logits.device)
    P = _alloc((M, K), torch.float32, logits.device)
    Q = _alloc((M, K), torch.float32, logits.device)
    check(lib().wsovod_mil_forward(ptr(logits), _ld(logits), ptr(seg_offsets), G, K, ptr(scores), ptr(P), ptr(Q), M,
                                   stream()), "mil_forward")
    return scores, P, Q


def mil_backward(dscores, P, Q, seg_offsets, K, tail=False):
    require_gpu(dscores, P, Q, seg_offsets)
    M = P.size(0)
    G = seg_offsets.numel() - 1
    dlogits = _alloc((M, 2 * K), torch.float32, P.device, tail)
    check(lib().wsovod_mil_backward(ptr(dscores.contiguous()), ptr(P), ptr(Q), ptr(seg_offsets), G, K, ptr(dlogits),
                                    _ld(dlogits), M, stream()), "mil_backward")
    return dlogits


def image_bce_forward(scores, seg_offsets, labels_onehot, norm):
    require_gpu(scores, seg_offsets, labels_onehot)
    G, K = labels_onehot.shape
    if scores.dim() != 2 or scores.size(1) != K or not scores.is_contiguous() or scores.dtype != torch.float32:
        raise RuntimeError(f"image_bce_forward: scores must be a dense fp32 (M,{K}) matrix, got {tuple(scores.shape)} "
                           f"strides {tuple(scores.stride())} {scores.dtype}")
    img = torch.empty((G, K), dtype=torch.float32, device=scores.device)
    dS = torch.empty_like(img)
    loss = torch.empty((1,), dtype=torch.float32, device=scores.device)
    check(lib().wsovod_image_bce_forward(ptr(scores), ptr(seg_offsets), G, K, ptr(labels_onehot), C.c_float(norm),
                                         ptr(img), ptr(dS), ptr(loss), stream()), "image_bce_forward")
    return loss, img, dS


def image_bce_backward(dS_img, seg_offsets, M, grad_out, tail=False):
    require_gpu(dS_img, seg_offsets, grad_out)
    G, K = dS_img.shape
    d = _alloc((M, K), torch.float32, dS_img.device, tail)
    check(lib().wsovod_image_bce_backward(ptr(dS_img), ptr(seg_offsets), G, K, ptr(grad_out), ptr(d), stream()),
          "image_bce_backward")
    return d


def weighted_ce_forward(logits, gt_classes, weights, weighted=True):
    """-> (loss (1), dlogits un-normalised (M,K1), accum (2): [sum, count])."""
    require_gpu(logits, gt_classes, weights)
    M, K1 = logits.shape
    dl = torch.empty((M, K1), dtype=torch.float32, device=logits.device)
    accum = torch.empty((2,), dtype=torch.float32, device=logits.device)
    loss = torch.empty((1,), dtype=torch.float32, device=logits.device)
    check(lib().wsovod_weighted_ce_forward(ptr(logits), _ld(logits), M, K1, ptr(gt_classes), ptr(weights),
                                           int(weighted), ptr(dl), _ld(dl), ptr(accum), ptr(loss), stream()),
          "weighted_ce_forward")
    return loss, dl, accum


def weighted_l1_box_forward(pred, proposal_boxes, gt_boxes, gt_classes, weights, K, bbox_weights, beta, weighted=True):
    """-> (loss (1), dpred (M,4) already divided by max(M,1))."""
    require_gpu(pred, proposal_boxes, gt_boxes, gt_classes, weights)
    M = pred.size(0)
    dp = torch.empty((M, 4), dtype=torch.float32, device=pred.device)
    accum = torch.empty((2,), dtype=torch.float32, device=pred.device)
    loss = torch.empty((1,), dtype=torch.float32, device=pred.device)
    bw = (C.c_float * 4)(*[float(v) for v in bbox_weights])
    check(lib().wsovod_weighted_l1_box_forward(ptr(pred), _ld(pred), ptr(proposal_boxes), ptr(gt_boxes),
                                               ptr(gt_classes), ptr(weights), M, K, bw, C.c_float(beta),
                                               int(weighted), ptr(dp), ptr(accum), ptr(loss), ptr(_TAIL.rows_true), stream()),
          "weighted_l1_box_forward")
    return loss, dp


def pgt_mine_and_label(scores, boxes, seg_offsets, gt_classes_img, gt_offsets, img_scores, K, iou_threshold):
    """See include/wsovod_hip.h.  Returns a dict of device tensors."""
    require_gpu(scores, boxes, seg_offsets, gt_classes_img, gt_offsets, img_scores)
    dev = scores.device
    M = scores.size(0)
    G = seg_offsets.numel() - 1
    T = gt_classes_img.numel()
    zero = torch.zeros((32 * T + 4 * G,), dtype=torch.uint8, device=dev)  # one fill for the five zero-initialised outputs
    o = dict(
        pgt_boxes=zero[:16 * T].view(torch.float32).view(T, 4),
        pgt_classes=zero[16 * T:24 * T].view(torch.int64),
        pgt_scores=zero[24 * T:28 * T].view(torch.float32),
        pgt_weights=zero[28 * T:32 * T].view(torch.float32),
        pgt_index=torch.full((T,), -1, dtype=torch.int32, device=dev),
        pgt_count=zero[32 * T:].view(torch.int32),
        # (padding rows behind the last segment -- a bucketed step graph -- must read as ignored: label -1, weight 0)
        gt_classes=torch.full((M,), -1, dtype=torch.int64, device=dev) if _TAIL.rows_true is not None
        else torch.empty((M,), dtype=torch.int64, device=dev),
        gt_boxes=_alloc((M, 4), torch.float32, dev),
        gt_scores=_alloc((M,), torch.float32, dev),
        gt_weights=_alloc((M,), torch.float32, dev),
        matched=_alloc((M,), torch.int32, dev),
    )
    boxes = boxes.to(torch.float32).contiguous()
    check(lib().wsovod_pgt_mine_and_label(
        ptr(scores), _ld(scores), ptr(boxes), ptr(seg_offsets), G, ptr(gt_classes_img), ptr(gt_offsets),
        ptr(img_scores), K, C.c_float(iou_threshold), ptr(o["pgt_boxes"]), ptr(o["pgt_classes"]),
        ptr(o["pgt_scores"]), ptr(o["pgt_weights"]), ptr(o["pgt_index"]), ptr(o["pgt_count"]), ptr(o["gt_classes"]),
        ptr(o["gt_boxes"]), ptr(o["gt_scores"]), ptr(o["gt_weights"]), ptr(o["matched"]), stream()),
        "pgt_mine_and_label")
    return o


def subsample_labels(labels, keys, seg_offsets, max_rows, num_samples, positive_fraction, bg_label):
    """detectron2 subsample_labels on the device for every image at once (include/wsovod_hip.h): rows that are not
    sampled come back as -1.  `keys` (float32, one per row) order the rows inside their group."""
    require_gpu(labels, keys, seg_offsets)
    labels = labels.to(torch.int64).contiguous()
    keys = keys.to(torch.float32).contiguous()
    out = torch.empty_like(labels)
    check(lib().wsovod_subsample_labels(ptr(labels), ptr(keys), ptr(seg_offsets), seg_offsets.numel() - 1, int(max_rows),
                                        int(num_samples), int(num_samples * positive_fraction), int(bg_label), ptr(out),
                                        stream()), "subsample_labels")
    return out


def mask_transpose(dy, y, scale, out_dtype, want_plain=True, want_t=True, ld_t=None, ld_plain=None, out_plain=None,
                   colsum=None, y_x2=False):
    """dA = dy * [y>0] * scale -> (dA (M, ld_plain>=N) or None, dAt (N, ld_t>=M) or None), zero padded.
    out_plain: optional (M, N) view (contiguous last dim) of a wider buffer that receives dA in place.
    colsum: optional zero-filled fp32 (N,) that receives the column sums of dA (the bias gradient) in the same pass."""
    require_gpu(dy, y, colsum)
    M, N = dy.shape
    ldp = ld_plain or N
    dA = None
    if out_plain is not None:
        assert out_plain.shape == (M, N) and out_plain.stride(1) == 1 and out_plain.dtype == out_dtype
        dA, ldp, want_plain = out_plain, out_plain.stride(0), True
    elif want_plain:
        dA = (torch.zeros if ldp != N else torch.empty)((M, ldp), dtype=out_dtype, device=dy.device)
    dAt = None
    if want_t:
        ld = ld_t or M
        dAt = (torch.zeros if ld != M else torch.empty)((N, ld), dtype=out_dtype, device=dy.device)
    if colsum is not None:
        assert colsum.dtype == torch.float32 and colsum.numel() == N and colsum.is_contiguous()
    if y is not None and (y_x2 or y.dtype != dy.dtype):
        # the mask source is the layer's bf16x2 output (its hi halves carry the sign) or -- f16mx outputs -- its plain bf16 rounding
        check(lib().wsovod_mask_transpose_ex(
            ptr(dy), _ld(dy), dtype_code(dy.dtype), ptr(y), _ld(y), BF16X2 if y_x2 else dtype_code(y.dtype), M, N, C.c_float(scale), ptr(dA), ldp, ptr(dAt),
            _ld(dAt) if want_t else 0, dtype_code(out_dtype), ptr(colsum), stream()), "mask_transpose_ex")
        return dA, dAt
    if colsum is not None:
        check(lib().wsovod_mask_transpose_colsum(
            ptr(dy), _ld(dy), ptr(y), _ld(y) if y is not None else 0, dtype_code(dy.dtype), M, N, C.c_float(scale),
            ptr(dA), ldp, ptr(dAt), _ld(dAt) if want_t else 0, dtype_code(out_dtype), ptr(colsum), stream()),
            "mask_transpose_colsum")
        return dA, dAt
    check(lib().wsovod_mask_transpose(ptr(dy), _ld(dy), ptr(y), _ld(y) if y is not None else 0, dtype_code(dy.dtype),
                                      M, N, C.c_float(scale), ptr(dA), ldp, ptr(dAt), _ld(dAt) if want_t else 0,
                                      dtype_code(out_dtype), stream()), "mask_transpose")
    return dA, dAt


def add_group_rows(x, row_group, add, x2=False):
    require_gpu(x, row_group, add)
    M, N = x.shape
    out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    check(lib().wsovod_add_group_rows(ptr(x), _ld(x), BF16X2 if x2 else dtype_code(x.dtype), ptr(row_group), ptr(add),
                                      _ld(add), M, N, ptr(out), N, stream()), "add_group_rows")
    return out


def scale_rows(x, row_scale, out):
    """out[:R, :C] = x * row_scale[:, None] (out may be wider/taller and of another dtype)."""
    require_gpu(x, row_scale, out)
    R, Cc = x.shape
    check(lib().wsovod_scale_rows(ptr(x), _ld(x), ptr(row_scale), R, Cc, ptr(out), _ld(out), dtype_code(out.dtype),
                                  stream()), "scale_rows")
    return out


def data_aware_forward(gap, W1, b1, W2, b2, E):
    require_gpu(gap, W1, b1, W2, b2, E)
    N, Cc = gap.shape
    Hd, P, F = W1.size(0), W2.size(0), E.size(1)
    dev = gap.device
    h1 = torch.empty((N, Hd), dtype=torch.float32, device=dev)
    h2 = torch.empty((N, P), dtype=torch.float32, device=dev)
    daf = torch.empty((N, F), dtype=torch.float32, device=dev)
    check(lib().wsovod_data_aware_forward(ptr(gap), N, Cc, ptr(W1), ptr(b1), Hd, ptr(W2), ptr(b2), P, ptr(E), F,
                                          ptr(h1), ptr(h2), ptr(daf), stream()), "data_aware_forward")
    return daf, h1, h2


def data_aware_backward(ddaf, gap, W2, E, h1, h2):
    require_gpu(ddaf, gap, W2, E, h1, h2)
    N, Cc = gap.shape
    Hd, P, F = h1.size(1), h2.size(1), E.size(1)
    dev = gap.device
    dW1 = torch.empty((Hd, Cc), dtype=torch.float32, device=dev)
    db1 = torch.empty((Hd,), dtype=torch.float32, device=dev)
    dW2 = torch.empty((P, Hd), dtype=torch.float32, device=dev)
    db2 = torch.empty((P,), dtype=torch.float32, device=dev)
    dE = torch.empty((P, F), dtype=torch.float32, device=dev)
    scratch = torch.empty((N * (P + Hd),), dtype=torch.float32, device=dev)
    check(lib().wsovod_data_aware_backward(ptr(ddaf.contiguous()), N, ptr(gap), Cc, ptr(W2), ptr(E), F, ptr(h1), Hd,
                                           ptr(h2), P, ptr(dW1), ptr(db1), ptr(dW2), ptr(db2), ptr(dE), ptr(scratch), stream()),
          "data_aware_backward")
    return dW1, db1, dW2, db2, dE


def nms_segments(boxes, seg_offsets, max_seg_len, iou_threshold, max_keep=0, valid=None):
    """Greedy NMS per segment over boxes sorted by descending score inside each segment.
    boxes (N,4) f32; seg_offsets (G+1) int32 on the device; max_seg_len: host int.  Returns
    (keep_idx (N) int32 -- segment g's kept positions, relative to its start, at [seg_offsets[g], ...),
     keep_count (G) int32)."""
    require_gpu(boxes, seg_offsets)
    boxes = boxes.contiguous()
    assert boxes.dtype == torch.float32 and seg_offsets.dtype == torch.int32
    N, G = boxes.shape[0], seg_offsets.numel() - 1
    W = max(1, (int(max_seg_len) + 63) // 64)
    ws = torch.empty((max(N, 1) * W,), dtype=torch.int64, device=boxes.device)
    keep_idx = torch.empty((max(N, 1),), dtype=torch.int32, device=boxes.device)
    keep_count = torch.zeros((max(G, 1),), dtype=torch.int32, device=boxes.device)
    if valid is not None:
        valid = valid.contiguous().view(torch.uint8)
    check(lib().wsovod_nms_segments(ptr(boxes), ptr(seg_offsets), ptr(valid), G, N, int(max_seg_len),
                                    C.c_float(iou_threshold), int(max_keep), ptr(ws), ptr(keep_idx), ptr(keep_count),
                                    stream()), "nms_segments")
    return keep_idx[:N], keep_count[:G]


def rpn_decode(anchors, deltas, index, image_sizes, weights, scale_clamp, min_size):
    """anchors (A,4), deltas (B,A,4), index (B,k) int64 or None, image_sizes (B,2) f32 device tensor (h,w).
    Returns (boxes (B,k,4) clipped, valid (B,k) bool)."""
    require_gpu(anchors, deltas, image_sizes)
    anchors, deltas = anchors.contiguous(), deltas.contiguous()
    B, A = deltas.shape[0], deltas.shape[1]
    k = index.shape[1] if index is not None else A
    if index is not None:
        index = index.contiguous()
        assert index.dtype == torch.int64
    boxes = torch.empty((B, k, 4), dtype=torch.float32, device=deltas.device)
    valid = torch.empty((B, k), dtype=torch.uint8, device=deltas.device)
    w = (C.c_float * 4)(*[float(v) for v in weights])
    check(lib().wsovod_rpn_decode(ptr(anchors), ptr(deltas), ptr(index), B, k, A, ptr(image_sizes), w,
                                  C.c_float(scale_clamp), C.c_float(min_size), ptr(boxes), ptr(valid), stream()),
          "rpn_decode")
    return boxes, valid.bool()


def im2col_rows(x_nhwc, rows, kernel_size, stride=1, padding=0, dilation=1):
    """x (N,H,W,C) NHWC contiguous; rows (n) int64 flat output-pixel ids (negative -> zero row).
    Returns (n, k*k*C) patch rows, tap-major then channel."""
    require_gpu(x_nhwc, rows)
    assert x_nhwc.is_contiguous() and rows.dtype == torch.int64
    N, Hh, Ww, Cc = x_nhwc.shape
    k = kernel_size
    Ho = (Hh + 2 * padding - dilation * (k - 1) - 1) // stride + 1
    Wo = (Ww + 2 * padding - dilation * (k - 1) - 1) // stride + 1
    rows = rows.contiguous()
    out = torch.empty((rows.numel(), k * k * Cc), dtype=x_nhwc.dtype, device=x_nhwc.device)
    check(lib().wsovod_im2col_rows(ptr(x_nhwc), dtype_code(x_nhwc.dtype), ptr(rows), rows.numel(), Hh, Ww, Cc, Ho, Wo,
                                   k, k, stride, padding, dilation, ptr(out), stream()), "im2col_rows")
    return out


def rpn_label_anchors(anchors, gt_boxes, gt_start, gt_count, thr_lo, thr_hi):
    """gt_start / gt_count: (B) int32 device arrays.
    -> (labels (B,A) int8 {1,0,-1}, best_gt (B,A) int32 row of gt_boxes or -1, best_iou (B,A) f32)."""
    require_gpu(anchors, gt_boxes, gt_start, gt_count)
    num_images = gt_start.numel()
    anchors, gt_boxes = anchors.contiguous(), gt_boxes.to(torch.float32).contiguous()
    A, T, dev = anchors.shape[0], gt_boxes.shape[0], anchors.device
    labels = torch.empty((num_images, A), dtype=torch.int8, device=dev)
    best_gt = torch.empty((num_images, A), dtype=torch.int32, device=dev)
    best_iou = torch.empty((num_images, A), dtype=torch.float32, device=dev)
    ws = torch.empty((max(T, 1),), dtype=torch.int32, device=dev)
    check(lib().wsovod_rpn_label_anchors(ptr(anchors), A, ptr(gt_boxes), ptr(gt_start), ptr(gt_count), num_images, T,
                                         C.c_float(thr_lo), C.c_float(thr_hi), ptr(best_iou), ptr(best_gt), ptr(ws),
                                         ptr(labels), stream()), "rpn_label_anchors")
    return labels, best_gt, best_iou


GEMM_TN_MAX_OPERAND_BYTES = (1 << 31) - 1  # one buffer resource per operand (tests lower it to exercise the row blocks)
# WSOVOD_DETERMINISTIC=1: weight-gradient tails / small grids are NOT cut along the reduction (their slices would meet by
# fp32 atomic adds in no fixed order): every dW is then run-to-run bit-identical, at ~3 % of the step.  The default keeps
# the split (wsovod_gemm_tn, include/wsovod_hip.h); the bias-gradient and loss sums still meet by a few float atomics.
DETERMINISTIC = __import__("os").environ.get("WSOVOD_DETERMINISTIC", "0") == "1"


def gemm_tn(P, Q, out=None, alpha=1.0, accumulate=False, split_tail=True, q_x2=False):
    """out (NI, NJ) fp32 (+)= alpha * P^T @ Q for row-major bf16 P (Mred, NI) and Q (Mred, NJ): the weight-gradient
    contraction over the operands' slow index, no transposed copies (transposed LDS reads).
    split_tail=False keeps a partial last round of tiles unsplit (fixed summation order, bit-reproducible)."""
    require_gpu(P, Q, out)
    _refuse_undeclared_planar("gemm_tn", P, Q)  # (the hi plane of a planar carrier is passed as its own bf16 view)
    split_tail = split_tail and not DETERMINISTIC
    # q_x2: Q is a bf16x2 matrix (fp32-typed carrier); the kernel reads the hi halves = Q rounded to bf16
    assert P.dtype == torch.bfloat16 and Q.dtype == (torch.float32 if q_x2 else torch.bfloat16) and P.shape[0] == Q.shape[0]
    Mred, NI, NJ = P.shape[0], P.shape[1], Q.shape[1]
    if out is None:
        out = torch.empty((NI, NJ), dtype=torch.float32, device=P.device)
    # the kernel addresses an operand through one buffer resource (< 2 GiB): longer reductions (e.g. 96 images x 512
    # proposals x 25088 pooled features = 2.5 GB) are cut into row blocks that accumulate into `out`
    limit = GEMM_TN_MAX_OPERAND_BYTES // (2 * max(_ld(P), _ld(Q) * (2 if q_x2 else 1), 1))
    step = max(64, limit // 64 * 64) if Mred > limit else max(Mred, 1)
    for r0 in range(0, max(Mred, 1), step):
        r1 = min(Mred, r0 + step)
        check(lib().wsovod_gemm_tn_ex(ptr(P[r0:r1]), _ld(P), ptr(Q[r0:r1]), _ld(Q), BF16X2 if q_x2 else BF16, r1 - r0, NI,
                                      NJ, ptr(out), _ld(out), C.c_float(alpha),
                                      int(bool(accumulate) or r0 > 0) | (0 if split_tail else 2), stream()), "gemm_tn")
    return out


def gemm_tn_sgd(P, Q, param, momentum_buf, shadow, lr, weight_decay, momentum, grad_scale=1.0, alpha=1.0, q_x2=False):
    """The weight-gradient contraction alpha * P^T @ Q FUSED with the momentum-SGD update of `param` (NI, NJ) it is the
    gradient of (wsovod_gemm_tn_sgd): the gradient never goes to memory.  shadow: None, the bf16 copy or the bf16x2
    (float32-typed) copy of param, refreshed in the same pass; lr: a float or a 1-element fp32 DEVICE tensor."""
    from .._lib import TnSgd

    mx_byte = None
    if isinstance(shadow, tuple):  # (f16mx carrier, per-tensor E8M0 byte): the "parity_mx" weight operand
        shadow, mx_byte = shadow
    require_gpu(P, Q, param, momentum_buf, shadow, mx_byte)
    _refuse_undeclared_planar("gemm_tn_sgd", P, Q)
    assert P.dtype == torch.bfloat16 and Q.dtype == (torch.float32 if q_x2 else torch.bfloat16) and P.shape[0] == Q.shape[0]
    Mred, NI, NJ = P.shape[0], P.shape[1], Q.shape[1]
    if tuple(param.shape) != (NI, NJ) or not param.is_contiguous() or param.dtype != torch.float32 \
            or momentum_buf.shape != param.shape or not momentum_buf.is_contiguous() or momentum_buf.dtype != torch.float32:
        raise RuntimeError("wsovod_hip gemm_tn_sgd: param / momentum buffer must be contiguous fp32 (NI, NJ)")
    if 2 * Mred * max(_ld(P), _ld(Q) * (2 if q_x2 else 1)) >= GEMM_TN_MAX_OPERAND_BYTES:
        raise RuntimeError("wsovod_hip gemm_tn_sgd: the reduction does not fit one launch (the fused update needs the "
                           "finished sum of a tile)")
    u = TnSgd()
    u.param, u.momentum_buf = param.data_ptr(), momentum_buf.data_ptr()
    u.shadow = shadow.data_ptr() if shadow is not None else None
    u.shadow_is_bf16x2 = 2 if mx_byte is not None else int(shadow is not None and shadow.dtype == torch.float32)
    u.mx_scale = mx_byte.data_ptr() if mx_byte is not None else None
    if shadow is not None and (shadow.numel() != param.numel() or not shadow.is_contiguous()):
        raise RuntimeError("wsovod_hip gemm_tn_sgd: the shadow must be a contiguous copy of the parameter")
    if torch.is_tensor(lr):
        require_gpu(lr)
        u.lr, u.lr_dev = 0.0, lr.data_ptr()
    else:
        u.lr, u.lr_dev = float(lr), None
    u.weight_decay, u.momentum, u.grad_scale = float(weight_decay), float(momentum), float(grad_scale)
    check(lib().wsovod_gemm_tn_sgd(ptr(P), _ld(P), ptr(Q), _ld(Q), BF16X2 if q_x2 else BF16, Mred, NI, NJ, C.c_float(alpha),
                                   C.byref(u), stream()), "gemm_tn_sgd")


def sgd_momentum_multi(entries, momentum, grad_scale=1.0, clip=None):
    """entries: list of (param, grad fp32 or bf16, momentum_buf, bf16_shadow or None, lr, weight_decay[, used_flag]);
    one launch per 32.  used_flag: optional 1-element fp32 device tensor, 0 = leave the tensor untouched.
    A float32-typed shadow is a bf16x2 copy of the parameter (X2 carrier): refreshed as (hi, lo) pairs.
    clip: None, or (kind, value) with kind "full_model" (one L2 norm over all entries, engine/defaults.py:292-318),
    "norm" (detectron2's per-parameter clip_grad_norm_) or "value" (per-parameter clip_grad_value_): the norms and the
    coefficients min(1, value / (norm + 1e-6)) are computed on the device, nothing is read back."""
    from .._lib import SgdTensor

    if not entries:
        return
    arr = (SgdTensor * len(entries))()
    coef = None
    for d, e in zip(arr, entries):
        p, g, b, sh, lr, wd = e[:6]
        used = e[6] if len(e) > 6 else None
        mx_byte = None
        if isinstance(sh, tuple):  # (f16mx carrier, per-tensor E8M0 byte)
            sh, mx_byte = sh
        require_gpu(p, g, b, sh, used, mx_byte)
        if torch.is_tensor(lr):  # a 1-element fp32 DEVICE tensor: the rate is read from memory when the kernel runs
            require_gpu(lr)
            assert lr.dtype == torch.float32 and lr.numel() == 1
            d.lr_dev, lr = lr.data_ptr(), 0.0
        d.used_flag = used.data_ptr() if used is not None else None
        if g.dtype not in (torch.float32, torch.bfloat16) or g.numel() != p.numel():
            raise RuntimeError("sgd_momentum_multi: gradient must be fp32 or bf16 with the parameter's element count")
        d.param, d.grad, d.momentum_buf = p.data_ptr(), g.data_ptr(), b.data_ptr()
        d.bf16_shadow = sh.data_ptr() if sh is not None else None
        d.shadow_is_bf16x2 = 2 if mx_byte is not None else (1 if (sh is not None and sh.dtype == torch.float32) else 0)
        d.mx_scale = mx_byte.data_ptr() if mx_byte is not None else None
        d.numel, d.lr, d.weight_decay = p.numel(), lr, wd
        d.grad_is_bf16 = 1 if g.dtype == torch.bfloat16 else 0
    if clip is not None:
        kind, value = clip
        if kind == "value":
            for d in arr:
                d.clip_value = float(value)
        elif kind in ("full_model", "norm"):
            dev = entries[0][0].device
            ws = torch.empty((int(lib().wsovod_grad_clip_workspace_floats(arr, len(entries))),), dtype=torch.float32, device=dev)
            coef = torch.empty((len(entries),), dtype=torch.float32, device=dev)
            check(lib().wsovod_grad_clip_coef(arr, len(entries), C.c_float(grad_scale), C.c_float(value),
                                              1 if kind == "norm" else 0, ptr(ws), ptr(coef), stream()), "grad_clip_coef")
            for k, d in enumerate(arr):
                d.grad_coef = coef.data_ptr() + 4 * k
        else:
            raise ValueError(f"sgd_momentum_multi: unknown clip kind {kind!r}")
    check(lib().wsovod_sgd_momentum_multi(arr, len(entries), C.c_float(momentum), C.c_float(grad_scale), stream()),
          "sgd_momentum_multi")
    return coef


def pack_bf16_multi(pairs):
    """pairs: list of (src fp32 contiguous, dst bf16 contiguous view, same numel): the gradient wire format, one
    launch per 32 tensors."""
    from .._lib import PackTensor

    if not pairs:
        return
    arr = (PackTensor * len(pairs))()
    for d, (src, dst) in zip(arr, pairs):
        require_gpu(src, dst)
        if src.dtype != torch.float32 or dst.dtype != torch.bfloat16 or src.numel() != dst.numel() or not (
                src.is_contiguous() and dst.is_contiguous()):
            raise RuntimeError("pack_bf16_multi: (fp32, bf16) contiguous pairs of equal size expected")
        d.src, d.dst, d.numel = src.data_ptr(), dst.data_ptr(), src.numel()
    check(lib().wsovod_pack_bf16_multi(arr, len(pairs), stream()), "pack_bf16_multi")


def sum_shards_bf16(src, n_shards, dst):
    """dst[e] = bf16(sum_j float(src[j * len(dst) + e])): the local reduction of the direct gradient exchange
    (all-to-all -> this -> all-gather); fp32 accumulation, one rounding."""
    require_gpu(src, dst)
    if src.dtype != torch.bfloat16 or dst.dtype != torch.bfloat16 or not (src.is_contiguous() and dst.is_contiguous()) \
            or src.numel() != n_shards * dst.numel():
        raise RuntimeError("sum_shards_bf16: contiguous bf16 src of n_shards * dst.numel() elements expected")
    check(lib().wsovod_sum_shards_bf16(src.data_ptr(), n_shards, dst.numel(), dst.data_ptr(), stream()), "sum_shards_bf16")
    return dst
