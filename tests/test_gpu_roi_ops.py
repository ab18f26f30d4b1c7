"""GPU parity: HIP RoIPool / ROIAlign through the C-ABI vs the C oracle (bit-exact indices)."""
import pytest
import torch

from oracle import roi_ops as O
from tests.util import random_rois

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [8, 64, 100])
def test_roi_pool_forward_bit_exact(gpu, channels_last, dtype, C):
    from wsovod_amd.layers import hip_ops

    torch.manual_seed(1)
    feat = torch.randn(2, C, 75, 100).to(dtype)
    rois = random_rois(96, 2, 600, 800, seed=3)
    ref_out, ref_arg = O.roi_pool_forward(feat.float(), rois, 0.125, (7, 7))
    f = feat.to(gpu)
    if channels_last:
        f = f.contiguous(memory_format=torch.channels_last)
    out, arg = hip_ops.roi_pool_forward(f, rois.to(gpu), 0.125, (7, 7), out_dtype=torch.float32)
    torch.cuda.synchronize()
    assert torch.equal(arg.cpu(), ref_arg)  # int32 argmax: bit-exact
    assert torch.equal(out.cpu(), ref_out)  # values are copies: bit-exact


@pytest.mark.parametrize("size", [(3, 5), (14, 14), (1, 1)])
def test_roi_pool_generic_output_size(gpu, size):
    from wsovod_amd.layers import hip_ops

    torch.manual_seed(5)
    feat = torch.randn(2, 70, 30, 40)
    rois = random_rois(32, 2, 240, 320, seed=11)
    ref_out, ref_arg = O.roi_pool_forward(feat, rois, 0.125, size)
    for cl in (False, True):
        f = feat.to(gpu)
        if cl:
            f = f.contiguous(memory_format=torch.channels_last)
        out, arg = hip_ops.roi_pool_forward(f, rois.to(gpu), 0.125, size)
        assert torch.equal(arg.cpu(), ref_arg) and torch.equal(out.cpu(), ref_out)


def test_roi_pool_scale_and_no_argmax(gpu):
    from wsovod_amd.layers import hip_ops

    torch.manual_seed(2)
    feat = torch.randn(1, 128, 75, 100)
    rois = random_rois(64, 1, 600, 800, seed=5)
    sc = torch.rand(64) + 1.0
    ref_out, _ = O.roi_pool_forward(feat, rois, 0.125, (7, 7))
    ref_out = ref_out * sc.view(-1, 1, 1, 1)
    f = feat.to(gpu).contiguous(memory_format=torch.channels_last)
    out, arg = hip_ops.roi_pool_forward(f, rois.to(gpu), 0.125, (7, 7), roi_scale=sc.to(gpu), need_argmax=False)
    assert arg is None
    assert torch.equal(out.cpu(), ref_out)
    out_bf, _ = hip_ops.roi_pool_forward(f, rois.to(gpu), 0.125, (7, 7), roi_scale=sc.to(gpu),
                                         out_dtype=torch.bfloat16, need_argmax=False)
    assert torch.equal(out_bf.cpu(), ref_out.to(torch.bfloat16))


@pytest.mark.parametrize("channels_last", [False, True])
def test_roi_pool_backward(gpu, channels_last):
    from wsovod_amd.layers import hip_ops

    torch.manual_seed(3)
    feat = torch.randn(2, 16, 40, 50)
    rois = random_rois(40, 2, 320, 400, seed=7)
    _, arg = O.roi_pool_forward(feat, rois, 0.125, (7, 7))
    g = torch.randn(40, 16, 7, 7)
    ref = O.roi_pool_backward(g, rois, arg, feat.shape)
    gi = hip_ops.roi_pool_backward(g.to(gpu), rois.to(gpu), arg.to(gpu), feat.shape, channels_last=channels_last)
    # atomic scatter-add: order-dependent fp32 sums -> tolerance, not bit-exact
    torch.testing.assert_close(gi.cpu().contiguous(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("aligned", [True, False])
@pytest.mark.parametrize("sampling_ratio", [0, 2])
def test_roi_align_forward_backward(gpu, channels_last, aligned, sampling_ratio):
    from wsovod_amd.layers import hip_ops

    torch.manual_seed(4)
    feat = torch.randn(2, 24, 40, 50)
    rois = random_rois(48, 2, 320, 400, seed=9)
    ref = O.roi_align_forward(feat, rois, 0.125, (7, 7), sampling_ratio, aligned)
    f = feat.to(gpu)
    if channels_last:
        f = f.contiguous(memory_format=torch.channels_last)
    out = hip_ops.roi_align_forward(f, rois.to(gpu), 0.125, (7, 7), sampling_ratio, aligned)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=5e-5)  # fp32 bilinear; FMA contraction differs
    g = torch.randn_like(ref)
    ref_gi = O.roi_align_backward(g, rois, 0.125, sampling_ratio, aligned, feat.shape)
    gi = hip_ops.roi_align_backward(g.to(gpu), rois.to(gpu), 0.125, sampling_ratio, aligned, feat.shape,
                                    channels_last=channels_last)
    torch.testing.assert_close(gi.cpu().contiguous(), ref_gi, rtol=1e-4, atol=1e-4)


def test_roi_pool_empty_and_errors(gpu):
    from wsovod_amd.layers import hip_ops

    feat = torch.randn(1, 8, 10, 10, device=gpu)
    out, arg = hip_ops.roi_pool_forward(feat, torch.zeros(0, 5, device=gpu), 0.125, (7, 7))
    assert out.shape == (0, 8, 7, 7) and arg.shape == (0, 8, 7, 7)
    with pytest.raises(RuntimeError):
        hip_ops.roi_pool_forward(feat, torch.zeros(3, 4, device=gpu), 0.125, (7, 7))
    with pytest.raises(RuntimeError):
        hip_ops.roi_pool_forward(feat.cpu(), torch.zeros(3, 5), 0.125, (7, 7))


@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_roi_loop_pool_three_outputs_bit_exact(gpu, channels_last, dtype):
    """The reference's native op in its CUDA form (region / frame / context) against the C restatement: values and
    argmax bit for bit; region == plain RoIPool on the non-negative map; frame <= region; the backward through the
    3R outputs is the RoIPool scatter with the rois repeated."""
    from wsovod_amd.layers import hip_ops as H

    g = torch.Generator().manual_seed(11)
    feat = torch.relu(torch.randn(2, 16, 38, 50, generator=g)).to(dtype)
    rois = random_rois(96, 2, 300, 400, seed=12)
    ref, ref_arg = O.roi_loop_pool_forward(feat.float(), rois, 0.125, (7, 7))
    f = feat.to(gpu)
    if channels_last:
        f = f.contiguous(memory_format=torch.channels_last)
    out, arg = H.roi_loop_pool_forward(f, rois.to(gpu), 0.125, (7, 7))
    assert out.shape == (3 * 96, 16, 7, 7)
    assert torch.equal(out.cpu(), ref) and torch.equal(arg.cpu(), ref_arg)
    plain = O.roi_pool_forward(feat.float(), rois, 0.125, (7, 7))[0]
    assert torch.equal(ref[:96], plain) and bool((ref[96:192] <= ref[:96]).all())
    assert bool((ref[192:] != ref[:96]).any())  # the context ring is a different window
    grad = torch.randn(3 * 96, 16, 7, 7, generator=g)
    gi = H.roi_pool_backward(grad.to(gpu), rois.repeat(3, 1).to(gpu), arg, (2, 16, 38, 50))
    ref_gi = O.roi_pool_backward(grad, rois.repeat(3, 1), ref_arg, (2, 16, 38, 50))
    torch.testing.assert_close(gi.cpu(), ref_gi, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("C", [512, 1024])
@pytest.mark.parametrize("aligned,sampling_ratio", [(True, 0), (False, 2)])
def test_roi_align_512_channel_bf16_form(gpu, C, aligned, sampling_ratio):
    """bf16 maps with a multiple of 512 channels and bf16 output (res5 of both depths: the bench's ROIAlignV2 line)
    take the 16-bytes-per-lane form of the row kernel with a bf16 transpose tile.  Per channel it issues the same
    FMAs in the same order as the 4-channel form: bit-identical to that form's fp32 output rounded to bf16, and within
    bf16 rounding of the oracle; edge-case rois and the objectness scale included."""
    from wsovod_amd.layers import hip_ops

    torch.manual_seed(8)
    feat = torch.randn(2, C, 30, 41).to(torch.bfloat16)
    rois = random_rois(70, 2, 240, 328, seed=21)
    sc = torch.rand(70) + 1.0
    f = feat.to(gpu).contiguous(memory_format=torch.channels_last)
    wide = hip_ops.roi_align_forward(f, rois.to(gpu), 0.125, (7, 7), sampling_ratio, aligned, roi_scale=sc.to(gpu),
                                     out_dtype=torch.bfloat16)
    narrow = hip_ops.roi_align_forward(f, rois.to(gpu), 0.125, (7, 7), sampling_ratio, aligned, roi_scale=sc.to(gpu),
                                       out_dtype=torch.float32)
    assert wide.dtype == torch.bfloat16 and torch.equal(wide, narrow.to(torch.bfloat16))
    ref = O.roi_align_forward(feat.float(), rois, 0.125, (7, 7), sampling_ratio, aligned) * sc.view(-1, 1, 1, 1)
    torch.testing.assert_close(wide.float().cpu(), ref, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("C", [512, 2048])
def test_roi_pool_512_channel_bf16_form(gpu, C):
    """The training step's pooling of the frozen backbone's res5 map: bf16 map of 512 (R18) / 2048 (R50) channels, bf16
    output, no argmax, objectness scale applied.  Values are copies of map cells times the scale: bit-identical to the
    oracle's fp32 result rounded to bf16."""
    from wsovod_amd.layers import hip_ops

    torch.manual_seed(12)
    feat = torch.randn(2, C, 38, 50).to(torch.bfloat16)
    rois = random_rois(90, 2, 304, 400, seed=31)
    sc = torch.rand(90) + 1.0
    ref, _ = O.roi_pool_forward(feat.float(), rois, 0.125, (7, 7))
    f = feat.to(gpu).contiguous(memory_format=torch.channels_last)
    out, arg = hip_ops.roi_pool_forward(f, rois.to(gpu), 0.125, (7, 7), roi_scale=sc.to(gpu), out_dtype=torch.bfloat16,
                                        need_argmax=False)
    assert arg is None and torch.equal(out.cpu(), (ref * sc.view(-1, 1, 1, 1)).to(torch.bfloat16))
    plain, _ = hip_ops.roi_pool_forward(f, rois.to(gpu), 0.125, (7, 7), out_dtype=torch.bfloat16, need_argmax=False)
    assert torch.equal(plain.cpu(), ref.to(torch.bfloat16))


@pytest.mark.parametrize("dtype,C,out_fmt", [(torch.bfloat16, 512, "bf16"), (torch.bfloat16, 1024, "bf16"),
                                             (torch.bfloat16, 264, "bf16"), (torch.bfloat16, 512, "f32"),
                                             (torch.float32, 256, "x2"), (torch.float32, 512, "x2hi"),
                                             (torch.float32, 260, "f32")])
def test_roi_pool_through_the_2x2_max_map_is_bit_identical(gpu, monkeypatch, dtype, C, out_fmt):
    """Values-only pooling of an NHWC map with many rois goes through the map's stride-1 2x2 maxima
    (`wsovod_roi_pool_forward_ws`): a quarter of the gather's requests.  max() is order-free and the windows cover exactly
    the bin's cells, so the result equals the cell scan (`WSOVOD_ROIPOOL_M2=0`) and the oracle bit for bit -- with NaN and
    +-Inf cells (the reference's `v > maxval` skips NaN and never lets -Inf replace -FLT_MAX), bins one cell wide or
    high (those rows of bins keep the cell scan), boxes outside the map, every output format."""
    from wsovod_amd._lib import lib
    from wsovod_amd.layers import hip_ops as H

    g = torch.Generator().manual_seed(77)
    feat = torch.randn(3, C, 38, 50, generator=g).to(dtype)
    feat[torch.rand(feat.shape, generator=g) < 0.02] = float("nan")
    feat[torch.rand(feat.shape, generator=g) < 0.02] = float("-inf")
    feat[torch.rand(feat.shape, generator=g) < 0.01] = float("inf")
    feat[0, :, 5:9, 7:12] = float("nan")   # whole bins of NaN
    feat[1, :, 20:30, 10:30] = float("-inf")
    rois = random_rois(400, 3, 304, 400, seed=41)
    small = random_rois(60, 3, 304, 400, seed=42, edge_cases=False)
    small[:, 3] = small[:, 1] + torch.rand(60, generator=g) * 70  # 1 .. 9 cells wide: bins of one and two columns mixed
    small[:, 4] = small[:, 2] + torch.rand(60, generator=g) * 70
    rois = torch.cat([rois, small])
    sc = torch.rand(len(rois), generator=g) + 1.0
    f = feat.to(gpu).contiguous(memory_format=torch.channels_last)
    need = lib().wsovod_roi_pool_workspace_bytes(1 if dtype == torch.bfloat16 else 0, 1, len(rois), 3, C, 38, 50, 7, 7, 0)
    assert need == 3 * 38 * 50 * C * (2 if dtype == torch.bfloat16 else 4)  # this shape takes the 2x2-max path
    assert lib().wsovod_roi_pool_workspace_bytes(1, 1, len(rois), 3, C, 38, 50, 7, 7, 1) == 0  # argmax: the cell scan
    od = {"bf16": torch.bfloat16, "f32": torch.float32, "x2": H.X2, "x2hi": H.X2}[out_fmt]

    def run():
        out, arg = H.roi_pool_forward(f, rois.to(gpu), 0.125, (7, 7), roi_scale=sc.to(gpu), out_dtype=od,
                                      need_argmax=False, want_hi=out_fmt == "x2hi")
        assert arg is None
        hi = H.x2_hi_pop(out) if out_fmt == "x2hi" else None
        if out_fmt == "x2hi":  # round 5: the training output is PLANAR bf16x2 -- the bf16 rounding IS its first plane
            assert H.x2_planar_of(out) and hi.data_ptr() == out.data_ptr() and hi.dtype == torch.bfloat16
            assert H.x2_planar_of(torch.flatten(out, start_dim=1)) and H.x2_hi_pop(torch.flatten(out, start_dim=1)) is not None
        if od == H.X2:
            out = H.x2_to_f32(out)
        return out, hi

    got, hi = run()
    monkeypatch.setenv("WSOVOD_ROIPOOL_M2", "0")
    want, hi0 = run()
    monkeypatch.delenv("WSOVOD_ROIPOOL_M2")
    eq = lambda a, b: torch.equal(torch.nan_to_num(a.float(), nan=12345.0), torch.nan_to_num(b.float(), nan=12345.0))
    assert eq(got, want)
    if hi is not None:
        assert eq(hi, hi0)
    ref, _ = O.roi_pool_forward(feat.float(), rois, 0.125, (7, 7))
    ref = ref * sc.view(-1, 1, 1, 1)
    if out_fmt == "bf16":
        ref = ref.to(torch.bfloat16)
    if od == H.X2:  # hi + lo of an fp32 value: exact up to 2^-17
        torch.testing.assert_close(torch.nan_to_num(got.cpu(), nan=0.0, posinf=1e30, neginf=-1e30),
                                   torch.nan_to_num(ref, nan=0.0, posinf=1e30, neginf=-1e30), rtol=2e-5, atol=0)
    else:
        assert eq(got.cpu(), ref)


@pytest.mark.parametrize("size,C", [((3, 5), 70), ((14, 14), 16), ((1, 1), 130), ((7, 7), 256), ((9, 8), 64)])
def test_roi_loop_pool_generic_sizes_bit_exact(gpu, size, C):
    """The wavefront-per-pooled-row NHWC form of the 3-output op at pooled sizes other than 7 x 7 (columns in chunks of
    7, pooled rows strided over the wavefronts; 14 x 14: one LDS tile, region and frame from two scans), odd channel
    counts, two channels per lane (C = 256), bf16 and fp32 maps: values and argmax equal to the C restatement."""
    from wsovod_amd.layers import hip_ops as H

    g = torch.Generator().manual_seed(5)
    for dtype in (torch.float32, torch.bfloat16):
        feat = torch.relu(torch.randn(2, C, 30, 41, generator=g)).to(dtype)
        rois = random_rois(50, 2, 240, 328, seed=9)
        ref, ref_arg = O.roi_loop_pool_forward(feat.float(), rois, 0.125, size)
        out, arg = H.roi_loop_pool_forward(feat.to(gpu), rois.to(gpu), 0.125, size)
        assert out.shape == (150, C) + size
        assert torch.equal(out.cpu(), ref) and torch.equal(arg.cpu(), ref_arg), (size, C, dtype)


@pytest.mark.parametrize("C,size", [(16, (7, 7)), (70, (3, 5)), (256, (7, 7)), (512, (7, 7))])
def test_pool_kernels_cover_every_output_element(gpu, monkeypatch, C, size):
    """The pool outputs are allocated uninitialised; the backward scatters through argmax.  With every output byte
    poisoned (0x7f) before the launch, no path of the kernels -- boxes off the map (empty bins: value 0, argmax -1),
    zero-area boxes, an roi count that leaves a partly filled last workgroup, odd channel counts, generic pooled sizes --
    may leave an element unwritten: outputs equal the C oracle's everywhere, argmax stays inside [-1, H*W)."""
    from wsovod_amd.layers import hip_ops as H

    monkeypatch.setattr(H, "POISON_OUTPUTS", True)
    g = torch.Generator().manual_seed(21)
    Hh, Ww = 30, 41
    feat = torch.relu(torch.randn(2, C, Hh, Ww, generator=g))
    rois = random_rois(37, 2, 240, 328, seed=22)  # 37: a ragged tail for every rois-per-workgroup form
    rois[3, 1:] = torch.tensor([500.0, 500.0, 600.0, 600.0])   # wholly off the map
    rois[4, 1:] = torch.tensor([-90.0, -80.0, -20.0, -10.0])   # wholly negative
    rois[5, 1:] = torch.tensor([100.0, 100.0, 100.0, 100.0])   # zero area
    rois[6, 1:] = torch.tensor([0.0, 0.0, 327.0, 239.0])       # the whole map
    poison_i = int.from_bytes(bytes([H.POISON_BYTE]) * 4, "little")
    f = feat.to(gpu).contiguous(memory_format=torch.channels_last)
    ref3, ref3_arg = O.roi_loop_pool_forward(feat, rois, 0.125, size)
    out3, arg3 = H.roi_loop_pool_forward(f, rois.to(gpu), 0.125, size)
    assert not bool((arg3 == poison_i).any()) and not bool((out3.view(torch.int32) == poison_i).any())
    assert int(arg3.min()) >= -1 and int(arg3.max()) < Hh * Ww
    assert torch.equal(out3.cpu(), ref3) and torch.equal(arg3.cpu(), ref3_arg)
    for layout_f in (f, feat.to(gpu)):
        ref, ref_arg = O.roi_pool_forward(feat, rois, 0.125, size)
        out, arg = H.roi_pool_forward(layout_f, rois.to(gpu), 0.125, size)
        assert not bool((arg == poison_i).any()) and not bool((out.view(torch.int32) == poison_i).any())
        assert torch.equal(out.cpu(), ref) and torch.equal(arg.cpu(), ref_arg)
        out_v, none = H.roi_pool_forward(layout_f, rois.to(gpu), 0.125, size, need_argmax=False)
        assert none is None and torch.equal(out_v.cpu(), ref)
    if size == (7, 7):
        al = H.roi_align_forward(f, rois.to(gpu), 0.125, size, 0, True)
        assert not bool((al.view(torch.int32) == poison_i).any()) and bool(torch.isfinite(al).all())
    # a corrupted index is never scattered (bounds check in roi_pool_bwd): only the valid entries reach grad_in
    grad = torch.randn(ref.shape, generator=g)
    bad = ref_arg.clone()
    bad[0, 0, 0, 0] = Hh * Ww + 12345
    gi = H.roi_pool_backward(grad.to(gpu), rois.to(gpu), bad.to(gpu), (2, C, Hh, Ww))
    fixed = ref_arg.clone()
    fixed[0, 0, 0, 0] = -1
    torch.testing.assert_close(gi.cpu(), O.roi_pool_backward(grad, rois, fixed, (2, C, Hh, Ww)), rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,C,Hh,Ww", [(torch.float32, 512, 38, 50), (torch.float32, 2048, 9, 13), (torch.bfloat16, 512, 19, 25),
                                         (torch.float32, 256, 8, 3)])
def test_gap_fused_with_the_pool_prepass(gpu, dtype, C, Hh, Ww):
    """Round 5 (ABI 7): inside hip_ops.gap_with_pool_prepass the global average pool writes the map's stride-1 2x2 maxima in
    the same pass and the next RoI max pool on that map takes them (wsovod_max2x2_gap_nhwc + wsovod_roi_pool_forward_m2).
    Pooled values: bit-identical to the unfused path (NaN / Inf cells included); the means: fp64 reference to 1e-6, run-to-run
    bit-identical; another map, or a map modified in between, falls back to the pooler's own pre-pass."""
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(21)
    N = 3
    feat = torch.randn(N, C, Hh, Ww, device=gpu).to(dtype).contiguous(memory_format=torch.channels_last)
    feat[0, 5, 2, 1] = float("nan")
    feat[1, 7, 3, 2] = float("inf")
    rois = random_rois(700, N, Hh * 8, Ww * 8, seed=22).to(gpu)
    plain, _ = H.roi_pool_forward(feat, rois, 0.125, (7, 7), need_argmax=False)
    nhwc = feat.permute(0, 2, 3, 1)
    gap_plain = H.global_avgpool_nhwc(nhwc)
    with H.gap_with_pool_prepass(rois.size(0), (7, 7)):
        gap = H.global_avgpool_nhwc(nhwc)
        assert H._M2.map is not None  # (700 rois on this map: the pooler's rule asks for the 2x2-max map)
        fused, _ = H.roi_pool_forward(feat, rois, 0.125, (7, 7), need_argmax=False)
        assert H._M2.map is None      # taken
        gap2 = H.global_avgpool_nhwc(nhwc)
        H._M2.map = None
    assert torch.equal(plain.view(torch.int16 if dtype == torch.bfloat16 else torch.int32),
                       fused.view(torch.int16 if dtype == torch.bfloat16 else torch.int32))
    assert torch.equal(gap.view(torch.int32), gap2.view(torch.int32))
    ok = torch.isfinite(gap_plain)
    ref = feat.permute(0, 2, 3, 1).double().mean(dim=(1, 2)).float()
    torch.testing.assert_close(gap[ok], ref[ok], rtol=2e-6 if dtype == torch.float32 else 1e-5, atol=1e-6)
    torch.testing.assert_close(gap[ok], gap_plain[ok], rtol=2e-6 if dtype == torch.float32 else 1e-5, atol=1e-6)
    assert torch.equal(torch.isnan(gap), torch.isnan(gap_plain)) and torch.equal(torch.isinf(gap), torch.isinf(gap_plain))
    # a map changed in place after the GAP: the parked maxima are stale and must not be used
    with H.gap_with_pool_prepass(rois.size(0), (7, 7)):
        H.global_avgpool_nhwc(nhwc)
        feat.mul_(1.5)
        stale, _ = H.roi_pool_forward(feat, rois, 0.125, (7, 7), need_argmax=False)
    fresh, _ = H.roi_pool_forward(feat, rois, 0.125, (7, 7), need_argmax=False)
    assert torch.equal(stale.view(torch.int16 if dtype == torch.bfloat16 else torch.int32),
                       fresh.view(torch.int16 if dtype == torch.bfloat16 else torch.int32))
