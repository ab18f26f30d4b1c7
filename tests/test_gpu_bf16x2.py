"""-m gpu: the bf16x2 activation format (include/wsovod_hip.h: WSOVOD_BF16X2; MODEL.HIP.PRECISION = "parity") kernel
by kernel against fp64 / the C oracle.  A bf16x2 value is hi + lo with hi = bf16(x), lo = bf16(x - hi): ~16 significant
bits, and a contraction over two such operands is three bf16 MFMA products with fp32 accumulation -- expected error
~2^-16 per product, i.e. ~1e-5 relative to sum |a||b|, against bf16's ~4e-3."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _enc_ref(x):
    """host model of the encoding: (hi, lo) as float tensors."""
    hi = x.to(torch.bfloat16).float()
    lo = (x - hi).to(torch.bfloat16).float()
    return hi, lo


def test_encode_decode_round_trip_and_layout(gpu):
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(0)
    x = torch.randn(37, 96) * torch.logspace(-6, 6, 96)[None]
    x[3, 5], x[4, 6], x[5, 7] = float("inf"), float("-inf"), 0.0
    enc = H.x2_encode(x.to(gpu))
    dec = H.x2_decode(enc).cpu()
    hi, lo = _enc_ref(x)
    fin = torch.isfinite(x)
    assert torch.equal(dec[fin], (hi + lo)[fin])  # decode = hi + lo exactly
    assert torch.equal(dec[~fin], x[~fin])  # an infinite value stays infinite
    assert float(((dec - x).abs() / x.abs().clamp(min=1e-30))[fin].max()) < 2.0 ** -15
    # layout: groups of 32 values as [32 hi | 32 lo] bf16 slots
    raw = enc.view(torch.bfloat16).view(37, 3, 2, 32).float().cpu()
    assert torch.equal(raw[:, :, 0, :].reshape(37, 96)[fin], hi[fin])
    assert torch.equal(raw[:, :, 1, :].reshape(37, 96)[fin], lo[fin])
    nan = torch.full((1, 32), float("nan"))
    assert torch.isnan(H.x2_decode(H.x2_encode(nan.to(gpu)))).all()


@pytest.mark.parametrize("tile", [0, 8256256, 2256256, 256256, 256128, 1256064, 1128064, 1128128, 64064])
@pytest.mark.parametrize("shape", [(512, 512, 1024), (300, 40, 256), (333, 129, 192), (64, 4, 4096), (1024, 4096, 512)])
def test_gemm_x2_against_fp64(gpu, tile, shape):
    """fp32-grade products on the bf16 MFMA pipe: error ~1e-5 of sum |a||b| (plain bf16: ~4e-3)."""
    from wsovod_amd.layers import hip_ops as H

    M, N, K = shape
    torch.manual_seed(1)
    A, B = torch.randn(M, K), torch.randn(N, K)
    out = H.gemm_nt(H.x2_encode(A.to(gpu)), H.x2_encode(B.to(gpu)), x2=True, out_dtype=torch.float32, tile_hint=tile)
    ref = A.double() @ B.double().t()
    scale = A.double().abs() @ B.double().abs().t()
    err = float(((out.cpu().double() - ref).abs() / scale).max())
    assert err < 3e-5, err
    # the value computed is exactly the three-product form on the encoded operands (fp32 accumulation)
    (ah, al), (bh, bl) = _enc_ref(A), _enc_ref(B)
    model = (ah.double() @ bh.double().t()) + (ah.double() @ bl.double().t()) + (al.double() @ bh.double().t())
    assert float(((out.cpu().double() - model).abs() / scale).max()) < 2e-6


def test_gemm_x2_epilogue_and_x2_output(gpu):
    from wsovod_amd.layers import hip_ops as H

    M, N, K = 520, 256, 320
    torch.manual_seed(2)
    A, B, bias, res = torch.randn(M, K), torch.randn(N, K) * 0.1, torch.randn(N), torch.randn(M, N)
    grp = torch.randint(0, 3, (M,), dtype=torch.int32)
    ga = torch.randn(3, N)
    ref = torch.relu((A.double() @ B.double().t()).float() + bias[None] + res)
    a2, b2 = H.x2_encode(A.to(gpu)), H.x2_encode(B.to(gpu))
    for tile in (0, 8256256, 2256256, 128128):
        out = H.gemm_nt(a2, b2, x2=True, out_dtype=H.X2, bias=bias.to(gpu), residual=H.x2_encode(res.to(gpu)),
                        residual_x2=True, relu=True, tile_hint=tile)
        got = H.x2_decode(out).cpu()
        torch.testing.assert_close(got, ref, rtol=3e-5, atol=3e-4)
        # slow epilogue path (group_add) with a bf16x2 output, fp32 residual
        out = H.gemm_nt(a2, b2, x2=True, out_dtype=H.X2, bias=bias.to(gpu), residual=res.to(gpu), relu=True,
                        row_group=grp.to(gpu), group_add=ga.to(gpu), tile_hint=tile)
        torch.testing.assert_close(H.x2_decode(out).cpu(), ref + ga[grp.long()], rtol=3e-5, atol=3e-4)
    # dropout: the same counter-based mask as the bf16 / fp32 kernels
    y0 = H.gemm_nt(A.to(gpu), B.to(gpu), out_dtype=torch.float32, dropout_p=0.5, dropout_seed=77)
    y1 = H.x2_decode(H.gemm_nt(a2, b2, x2=True, out_dtype=H.X2, dropout_p=0.5, dropout_seed=77))
    assert torch.equal(y0 == 0, y1 == 0)
    with pytest.raises(RuntimeError, match="multiples of 32"):
        H.gemm_nt(a2, H.x2_encode(B[:40].to(gpu)), x2=True, out_dtype=H.X2)


@pytest.mark.parametrize("geom", [
    dict(n=2, H=20, W=28, Cin=64, Cout=64, k=3, s=1, p=1, d=1),
    dict(n=2, H=19, W=25, Cin=128, Cout=256, k=3, s=1, p=2, d=2),
    dict(n=1, H=16, W=16, Cin=256, Cout=512, k=1, s=1, p=0, d=1),
    dict(n=3, H=33, W=21, Cin=64, Cout=128, k=3, s=1, p=1, d=1),
])
@pytest.mark.parametrize("tile", [0, 256256, 8256256, 2256256, 1256064, 512128])
def test_conv_x2_against_fp64(gpu, geom, tile):
    from wsovod_amd.layers import hip_ops as H

    g = geom
    if tile == 1256064 and g["Cout"] > 64:
        pytest.skip("the tall 64-column tile is for 64 output channels")
    if tile == 512128 and g["Cout"] != 128:
        pytest.skip("the 512x128 tile serves the 128-channel convs of res3")
    torch.manual_seed(3)
    x = torch.randn(g["n"], g["Cin"], g["H"], g["W"])
    w = torch.randn(g["Cout"], g["Cin"], g["k"], g["k"]) * 0.05
    b = torch.randn(g["Cout"])
    ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), g["s"], g["p"], g["d"])).float()
    Ho, Wo = ref.shape[-2:]
    xn = H.x2_encode(x.permute(0, 2, 3, 1).reshape(-1, g["Cin"]).contiguous().to(gpu)).view(g["n"], g["H"], g["W"], g["Cin"])
    wq = H.x2_encode(w.permute(0, 2, 3, 1).reshape(g["Cout"], -1).contiguous().to(gpu))
    conv = dict(n_img=g["n"], H=g["H"], W=g["W"], Cin=g["Cin"], Ho=Ho, Wo=Wo, KH=g["k"], KW=g["k"], stride=g["s"],
                pad=g["p"], dil=g["d"])
    out = H.gemm_nt(xn, wq, conv=conv, x2=True, bias=b.to(gpu), relu=True, out_dtype=H.X2, tile_hint=tile)
    got = H.x2_decode(out).view(g["n"], Ho, Wo, g["Cout"]).permute(0, 3, 1, 2).cpu()
    scale = float(F.conv2d(x.abs().double(), w.abs().double(), None, g["s"], g["p"], g["d"]).max())
    assert float((got - ref).abs().max()) < 3e-5 * scale
    out32 = H.gemm_nt(xn, wq, conv=conv, x2=True, bias=b.to(gpu), relu=True, out_dtype=torch.float32, tile_hint=tile)
    assert float((out32.view(g["n"], Ho, Wo, g["Cout"]).permute(0, 3, 1, 2).cpu() - ref).abs().max()) < 3e-5 * scale


def test_conv_x2_with_fused_projection_shortcut(gpu):
    """out = relu(conv3x3(h) + shortcut1x1(x)) as ONE implicit GEMM on bf16x2 operands (wsovod_gemm_desc.A2)."""
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(4)
    n, Hh, Ww, C1, C2, Co = 2, 18, 22, 128, 64, 128
    h, x = torch.randn(n, C1, Hh, Ww), torch.randn(n, C2, Hh, Ww)
    w, ws, b = torch.randn(Co, C1, 3, 3) * 0.05, torch.randn(Co, C2, 1, 1) * 0.1, torch.randn(Co)
    ref = torch.relu(F.conv2d(h.double(), w.double(), b.double(), 1, 2, 2) + F.conv2d(x.double(), ws.double())).float()
    enc = lambda t, c: H.x2_encode(t.permute(0, 2, 3, 1).reshape(-1, c).contiguous().to(gpu)).view(n, Hh, Ww, c)
    wcat = torch.cat([w.permute(0, 2, 3, 1).reshape(Co, -1), ws.reshape(Co, C2)], dim=1).contiguous()
    conv = dict(n_img=n, H=Hh, W=Ww, Cin=C1, Ho=Hh, Wo=Ww, KH=3, KW=3, stride=1, pad=2, dil=2)
    for tile in (0, 256256, 8256256, 2256256):  # 16-wavefront tile, four-phase and two-phase 8-wavefront tiles
        out = H.gemm_nt(enc(h, C1), H.x2_encode(wcat.to(gpu)), conv=conv, x2=True, bias=b.to(gpu), relu=True,
                        out_dtype=H.X2, A2=enc(x, C2), tile_hint=tile)
        got = H.x2_decode(out).view(n, Hh, Ww, Co).permute(0, 3, 1, 2).cpu()
        assert float((got - ref).abs().max()) < 3e-4, tile


def test_maxpool_add_group_rows_and_mask_on_x2(gpu):
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(5)
    x = torch.randn(2, 9, 11, 64)
    xe = H.x2_encode(x.view(-1, 64).to(gpu)).view(2, 9, 11, 64)
    xv = H.x2_decode(xe.view(-1, 64)).view(2, 9, 11, 64).cpu()  # the values the map really holds
    for stride, pad in ((2, False), (1, True)):
        got = H.maxpool2x2_nhwc(xe, stride, zero_pad_br=pad, x2=True)
        src = xv.permute(0, 3, 1, 2)
        if pad:
            src = F.pad(src, (0, 1, 0, 1))
        want = F.max_pool2d(src, 2, stride).permute(0, 2, 3, 1)
        assert torch.equal(H.x2_decode(got.view(-1, 64)).view(want.shape).cpu(), want)
    M, N = 77, 96
    a = torch.randn(M, N)
    grp = torch.randint(0, 4, (M,), dtype=torch.int32)
    add = torch.randn(4, N)
    ae = H.x2_encode(a.to(gpu))
    got = H.x2_decode(H.add_group_rows(ae, grp.to(gpu), add.to(gpu), x2=True)).cpu()
    torch.testing.assert_close(got, H.x2_decode(ae).cpu() + add[grp.long()], rtol=2e-5, atol=1e-6)
    # backward mask from a bf16x2 layer output: identical to the mask of its fp32 values
    y = torch.relu(torch.randn(M, N))
    dy = torch.randn(M, N)
    cs = torch.zeros(N, device=gpu)
    dA, dAt = H.mask_transpose(dy.to(gpu), H.x2_encode(y.to(gpu)), 2.0, torch.bfloat16, want_plain=True, want_t=True, ld_t=128,
                               colsum=cs, y_x2=True)
    want = torch.where(y > 0, dy * 2.0, torch.zeros(()))
    assert torch.equal(dA.float().cpu(), want.to(torch.bfloat16).float())
    assert torch.equal(dAt.float().cpu()[:, :M], want.to(torch.bfloat16).float().t())
    torch.testing.assert_close(cs.cpu(), want.sum(0), rtol=1e-4, atol=1e-4)


def test_gemm_tn_reads_the_hi_halves_of_a_bf16x2_operand(gpu):
    """dW = dY^T X with X as the parity forward left it (bf16x2): bit-identical to the contraction on bf16(X)."""
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(6)
    Mred, NI, NJ = 1000, 256, 512
    P = torch.randn(Mred, NI).to(torch.bfloat16).to(gpu)
    X = torch.randn(Mred, NJ).to(gpu)
    want = H.gemm_tn(P, X.to(torch.bfloat16), split_tail=False)
    got = H.gemm_tn(P, H.x2_encode(X), split_tail=False, q_x2=True)
    assert torch.equal(got, want)


def test_roi_pool_and_align_emit_bf16x2(gpu):
    from tests.util import random_rois
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(7)
    n, Cc, Hh, Ww = 2, 512, 38, 50
    feat = torch.randn(n, Hh, Ww, Cc).to(gpu).permute(0, 3, 1, 2)  # fp32 NHWC storage
    rois = random_rois(96, n, 300, 400, seed=8).to(gpu)
    scale = (torch.rand(96) + 1).to(gpu)
    want = H.roi_pool_forward(feat, rois, 0.125, (7, 7), roi_scale=scale, need_argmax=False)[0]
    got = H.roi_pool_forward(feat, rois, 0.125, (7, 7), roi_scale=scale, out_dtype=H.X2, need_argmax=False)[0]
    dec = H.x2_decode(got.view(96, -1)).view_as(want)
    assert float(((dec - want).abs() / want.abs().clamp(min=1e-20)).max()) < 2.0 ** -15  # the fp32 pool, re-encoded
    want = H.roi_align_forward(feat, rois, 0.125, (7, 7), 0, True, roi_scale=scale)
    got = H.roi_align_forward(feat, rois, 0.125, (7, 7), 0, True, roi_scale=scale, out_dtype=H.X2)
    dec = H.x2_decode(got.view(96, -1)).view_as(want)
    assert float(((dec - want).abs() / want.abs().clamp(min=1e-20)).max()) < 2.0 ** -15


def test_stem_conv1_x2_against_fp64(gpu):
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(9)
    N, Hp, Wp = 2, 37, 45
    img = torch.randint(0, 256, (N, 3, Hp, Wp), dtype=torch.uint8)
    sizes = torch.tensor([[37, 45], [30, 41]], dtype=torch.int32)
    mean, std = [102.9801, 115.9465, 122.7717], [1.0, 1.0, 1.0]
    w = torch.randn(64, 3, 3, 3) * 0.02
    b = torch.randn(64)
    x = img.double()
    for i in range(N):  # zero AFTER normalisation outside each image's own size
        x[i] = (x[i] - torch.tensor(mean).view(3, 1, 1).double())
        x[i, :, sizes[i, 0]:, :] = 0
        x[i, :, :, sizes[i, 1]:] = 0
    ref = torch.relu(F.conv2d(x, w.double(), b.double(), stride=2, padding=1)).float()
    w32 = torch.zeros(64, 32)
    w32[:, :27] = w.permute(0, 2, 3, 1).reshape(64, 27)  # k = (r*3+q)*3 + c
    out = H.stem_conv1_x2(img.to(gpu), sizes.to(gpu), mean, std, H.x2_encode(w32.to(gpu)), b.to(gpu))
    got = H.x2_decode(out.view(-1, 64)).view(N, ref.shape[2], ref.shape[3], 64).permute(0, 3, 1, 2).cpu()
    scale = float(F.conv2d(x.abs(), w.abs().double(), None, stride=2, padding=1).max())
    assert float((got - ref).abs().max()) < 3e-5 * scale


@pytest.mark.parametrize("form", [None, "16", "32"])
@pytest.mark.parametrize("pool", [False, True])
@pytest.mark.parametrize("with_res", [False, True])
def test_conv3x3_c64_halo_kernel_on_bf16x2(gpu, pool, with_res, form, monkeypatch):
    """The 64 -> 64 channel 3x3 halo-tile kernel in its bf16x2 form (stem conv2 / conv3, res2): against fp64, and against
    the generic implicit-GEMM tile on the same operands (same products, another summation order); ragged tile edges.
    form: None = the shipped half-K 8 x 32 tile, "16" / "32" = the whole-K tiles kept for A/B runs."""
    from wsovod_amd.layers import hip_ops as H

    if form is None:
        monkeypatch.delenv("WSOVOD_C64X_TW", raising=False)
    else:
        monkeypatch.setenv("WSOVOD_C64X_TW", form)
    torch.manual_seed(11)
    n, Hh, Ww = 2, 22, 70  # not multiples of the 8 x 32 tile
    x = torch.randn(n, 64, Hh, Ww)
    w = torch.randn(64, 64, 3, 3) * 0.05
    b = torch.randn(64)
    res = torch.randn(n, 64, Hh, Ww)
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1, 1)
    if with_res:
        ref = ref + res.double()
    ref = torch.relu(ref)
    if pool:
        ref = F.max_pool2d(ref, 2, 2)
    ref = ref.float()
    enc = lambda t: H.x2_encode(t.permute(0, 2, 3, 1).reshape(-1, 64).contiguous().to(gpu)).view(n, Hh, Ww, 64)
    wq = H.x2_encode(w.permute(0, 2, 3, 1).reshape(64, -1).contiguous().to(gpu))
    conv = dict(n_img=n, H=Hh, W=Ww, Cin=64, Ho=Hh, Wo=Ww, KH=3, KW=3, stride=1, pad=1, dil=1)
    kw = dict(x2=True, bias=b.to(gpu), relu=True, out_dtype=H.X2)
    if with_res:
        kw.update(residual=enc(res).view(-1, 64), residual_x2=True)
    out = H.gemm_nt(enc(x), wq, conv=dict(conv, pool=2) if pool else conv, **kw)
    Ho, Wo = ref.shape[-2:]
    got = H.x2_decode(out).view(n, Ho, Wo, 64).permute(0, 3, 1, 2).cpu()
    scale = float(F.conv2d(x.abs().double(), w.abs().double(), None, 1, 1, 1).max()) + 4.0
    assert float((got - ref).abs().max()) < 3e-5 * scale
    gen = H.gemm_nt(enc(x), wq, conv=conv, tile_hint=1256064, **kw)  # the generic tile, unpooled
    gen = H.x2_decode(gen).view(n, Hh, Ww, 64).permute(0, 3, 1, 2)
    if pool:
        gen = F.max_pool2d(gen, 2, 2)
    assert float((got - gen.cpu()).abs().max()) < 1e-5 * scale


def test_sgd_refreshes_the_bf16x2_weight_operand(gpu):
    """HipSGD keeps the cached bf16x2 encoding of a weight current inside the fused update (no re-encode pass)."""
    from wsovod_amd.engine.trainer import HipSGD
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(12)
    p = torch.nn.Parameter(torch.randn(96, 160, device=gpu))
    q = torch.nn.Parameter(torch.randn(7, 5, device=gpu))  # numel not a multiple of 32: no bf16x2 shadow
    opt = HipSGD([{"params": [p], "lr": 0.1, "weight_decay": 1e-3}, {"params": [q], "lr": 0.1, "weight_decay": 0.0}], 0.1,
                 momentum=0.9)
    enc0 = H.x2_cached(p)
    for _ in range(3):
        p.grad, q.grad = torch.randn_like(p), torch.randn_like(q)
        opt.step()
        enc = H.x2_cached(p)
        assert enc.data_ptr() == enc0.data_ptr()  # the same buffer, refreshed by the update kernel
        assert torch.equal(H.x2_decode(enc), H.x2_decode(H.x2_encode(p.detach())))
    ref = torch.nn.Parameter(p.detach().clone())  # and the update itself is torch.optim.SGD's
    ref_opt = torch.optim.SGD([ref], lr=0.1, momentum=0.9, weight_decay=1e-3)
    p2 = torch.nn.Parameter(ref.detach().clone())
    opt2 = HipSGD([{"params": [p2], "lr": 0.1, "weight_decay": 1e-3}], 0.1, momentum=0.9)
    H.x2_cached(p2)
    for _ in range(3):
        g = torch.randn_like(ref)
        ref.grad, p2.grad = g.clone(), g.clone()
        ref_opt.step()
        opt2.step()
    torch.testing.assert_close(p2.detach(), ref.detach(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("case", ["res4_residual", "res5_shortcut_dilated"])
def test_conv_x2_split_k_at_few_tiles(gpu, case, monkeypatch):
    """The implicit-GEMM conv as split-K slices of the 8-wavefront tile + the finalize pass (bias, residual, ReLU, bf16x2
    encode), reachable with WSOVOD_CONV_SPLITK=1 (measured in round 5 at 1 - 4 images per step: equal to the 128x64 grid at
    best -- res5 of one image 124 vs 124 us -- so the dispatcher does not choose it by itself).  Same result as the unsplit
    tiles up to the summation order; slices start inside the filter taps, inside the channel chunks and inside the fused
    1x1 shortcut's K range."""
    from wsovod_amd.layers import hip_ops as H

    monkeypatch.setenv("WSOVOD_CONV_SPLITK", "1")
    torch.manual_seed(6)
    if case == "res4_residual":
        n, Hh, Ww, C1, Co, dil, C2 = 1, 38, 50, 256, 256, 1, 0
    else:
        n, Hh, Ww, C1, Co, dil, C2 = 1, 30, 41, 512, 512, 2, 256
    h = torch.randn(n, C1, Hh, Ww) * 0.5
    w, b = torch.randn(Co, C1, 3, 3) * 0.03, torch.randn(Co)
    res = torch.randn(n, Co, Hh, Ww) if not C2 else None
    x, ws = (torch.randn(n, C2, Hh, Ww) * 0.5, torch.randn(Co, C2, 1, 1) * 0.05) if C2 else (None, None)
    ref = F.conv2d(h.double(), w.double(), b.double(), 1, dil, dil)
    if C2:
        ref = ref + F.conv2d(x.double(), ws.double())
    else:
        ref = ref + res.double()
    ref = torch.relu(ref).float()
    enc = lambda t, c: H.x2_encode(t.permute(0, 2, 3, 1).reshape(-1, c).contiguous().to(gpu)).view(n, Hh, Ww, c)
    wrow = w.permute(0, 2, 3, 1).reshape(Co, -1)
    wcat = (torch.cat([wrow, ws.reshape(Co, C2)], dim=1) if C2 else wrow).contiguous()
    conv = dict(n_img=n, H=Hh, W=Ww, Cin=C1, Ho=Hh, Wo=Ww, KH=3, KW=3, stride=1, pad=dil, dil=dil)
    kw = dict(conv=conv, x2=True, bias=b.to(gpu), relu=True, out_dtype=H.X2)
    if C2:
        kw["A2"] = enc(x, C2)
    else:
        kw.update(residual=H.x2_encode(res.permute(0, 2, 3, 1).reshape(-1, Co).contiguous().to(gpu)), residual_x2=True)
    M = n * Hh * Ww
    assert -(-M // 256) * -(-Co // 256) <= 128  # few tiles: the automatic choice is the split form
    outs = {}
    for tile in (0, 2256256, 1128128):
        out = H.gemm_nt(enc(h, C1), H.x2_encode(wcat.to(gpu)), tile_hint=tile, **kw)
        outs[tile] = H.x2_decode(out).view(n, Hh, Ww, Co).permute(0, 3, 1, 2).cpu()
    scale = float(F.conv2d(h.abs().double(), w.abs().double(), None, 1, dil, dil).max())
    for tile, got in outs.items():
        assert float((got - ref).abs().max()) < 3e-5 * scale, tile
    monkeypatch.delenv("WSOVOD_CONV_SPLITK")
    assert float((outs[0] - outs[2256256]).abs().max()) < 1e-5 * scale


@pytest.mark.parametrize("case", ["res3", "res4_residual", "res5_shortcut_dilated", "fc2"])
def test_deep_dma_small_tiles_equal_the_two_stage_tiles(gpu, case):
    """Round 6: at 1 - 2 images per step the 128x64 / 64x64 grids run behind a deep LDS-DMA pipeline (three / four stages in
    flight: tiles 3128064 / 3064064) instead of one DMA round trip per K-step.  Same products in the same order: bit-identical
    to the two-stage tiles 1128064 / 64064 -- conv with residual, conv with the fused 1x1 shortcut under dilation (K range
    ends inside the shortcut), ragged rows / columns, a plain GEMM with bias + ReLU.  (The dispatcher picks the deep forms
    wherever it used to pick the two-stage ones; WSOVOD_X2_DEEP=0 restores those.)"""
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(8)
    if case == "fc2":
        M, N, K = 520, 1000 - 8, 4096
        A, B = H.x2_encode(torch.randn(M, K, device=gpu)), H.x2_encode(torch.randn(N, K, device=gpu) * 0.02)
        kw = dict(x2=True, bias=torch.randn(N, device=gpu), relu=True, out_dtype=torch.float32)
        run = lambda tile: H.gemm_nt(A, B, tile_hint=tile, **kw)
        pairs = [(3128064, 1128064)]
    else:
        if case == "res3":
            n, Hh, Ww, C1, Co, dil, C2, res_on = 1, 75, 100, 128, 128, 1, 0, True
        elif case == "res4_residual":
            n, Hh, Ww, C1, Co, dil, C2, res_on = 1, 38, 50, 256, 256, 2, 0, True
        else:
            n, Hh, Ww, C1, Co, dil, C2, res_on = 1, 30, 41, 512, 512, 2, 256, False
        enc = lambda t, c: H.x2_encode(t.reshape(-1, c).contiguous().to(gpu)).view(n, Hh, Ww, c)
        h = enc(torch.randn(n, Hh, Ww, C1) * 0.5, C1)
        wcat = H.x2_encode((torch.randn(Co, 9 * C1 + C2) * 0.03).to(gpu))
        conv = dict(n_img=n, H=Hh, W=Ww, Cin=C1, Ho=Hh, Wo=Ww, KH=3, KW=3, stride=1, pad=dil, dil=dil)
        kw = dict(conv=conv, x2=True, bias=torch.randn(Co, device=gpu), relu=True, out_dtype=H.X2)
        if C2:
            kw["A2"] = enc(torch.randn(n, Hh, Ww, C2) * 0.5, C2)
        if res_on:
            kw.update(residual=H.x2_encode(torch.randn(n * Hh * Ww, Co, device=gpu)), residual_x2=True)
        run = lambda tile: H.gemm_nt(h, wcat, tile_hint=tile, **kw)
        pairs = [(3128064, 1128064), (3064064, 64064)]
    for deep, base in pairs:
        a, b = run(deep), run(base)
        torch.cuda.synchronize()
        assert torch.equal(a, b), (case, deep)
        assert bool(torch.isfinite(H.x2_decode(a) if a.dtype == torch.float32 and kw.get("out_dtype") == H.X2 else a).all())


def test_res3_convs_on_the_512x128_tile_equal_the_256x128_tile(gpu):
    """Round 5: from ~14 images per step the 128-channel convs of res3 run on 512 x 128 tiles (16 wavefronts as 8 x 2, 160 KiB
    of LDS).  Same products in the same order per output element as the 256 x 128 tile: bit-identical outputs -- stride 2
    with the fused 1x1 projection shortcut, residual, ragged last tile (M not a multiple of 512)."""
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(8)
    n, Hi, Wi = 3, 46, 62
    enc = lambda t, c: H.x2_encode(t.reshape(-1, c).contiguous().to(gpu))
    x64 = torch.randn(n, Hi, Wi, 64)
    Ho, Wo = (Hi - 1) // 2 + 1, (Wi - 1) // 2 + 1
    w = torch.randn(128, 9 * 64) * 0.05
    b = torch.randn(128)
    geom = dict(n_img=n, H=Hi, W=Wi, Cin=64, Ho=Ho, Wo=Wo, KH=3, KW=3, stride=2, pad=1, dil=1)
    outs = [H.gemm_nt(enc(x64, 64).view(n, Hi, Wi, 64), enc(w, 9 * 64), conv=geom, x2=True, bias=b.to(gpu), relu=True,
                      out_dtype=H.X2, tile_hint=t) for t in (256128, 512128)]
    assert (n * Ho * Wo) % 512 != 0 and torch.equal(outs[0], outs[1])
    h = torch.randn(n, Ho, Wo, 128)
    xs = torch.randn(n, Ho, Wo, 64)  # the block's input at the output resolution: the fused projection shortcut's operand
    wcat = torch.cat([torch.randn(128, 9 * 128) * 0.03, torch.randn(128, 64) * 0.1], dim=1)
    geom = dict(n_img=n, H=Ho, W=Wo, Cin=128, Ho=Ho, Wo=Wo, KH=3, KW=3, stride=1, pad=1, dil=1)
    outs = [H.gemm_nt(enc(h, 128).view(n, Ho, Wo, 128), enc(wcat, wcat.shape[1]), conv=geom, x2=True, bias=b.to(gpu), relu=True,
                      out_dtype=H.X2, A2=enc(xs, 64).view(n, Ho, Wo, 64), tile_hint=t) for t in (256128, 512128)]
    assert torch.equal(outs[0], outs[1])
    res = enc(torch.randn(n * Ho * Wo, 128), 128)
    outs = [H.gemm_nt(enc(h, 128).view(n, Ho, Wo, 128), enc(wcat[:, :9 * 128].contiguous(), 9 * 128), conv=geom, x2=True,
                      bias=b.to(gpu), relu=True, residual=res, residual_x2=True, out_dtype=H.X2, tile_hint=t)
            for t in (256128, 512128)]
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("M,N,K", [(1024, 512, 2048), (300, 4096, 25088), (16384, 1024, 1024)])
def test_planar_bf16x2_operand_equals_the_interleaved_one(gpu, M, N, K):
    """Round 5: the poolers' training output is PLANAR bf16x2 (all hi values, then all lo values: the hi plane doubles as the
    plain bf16 operand of fc1's weight gradient).  The lean two-phase tile reads it as the A operand -- 64 bytes of each
    plane per K-step into the SAME LDS image as the interleaved layout: identical products in identical order, so the
    result equals the interleaved operand's bit for bit, split-K (M = 300, K = 25088) included."""
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(13)
    x = torch.randn(M, K, device=gpu)
    w = H.x2_encode(torch.randn(N, K, device=gpu) * 0.02)
    b = torch.randn(N, device=gpu)
    inter = H.x2_encode(x)
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    planar = torch.cat([hi.reshape(-1), lo.reshape(-1)]).view(torch.float32).view(M, K)
    assert torch.equal(H.x2_decode(inter), hi.float() + lo.float())  # the same (hi, lo) pairs in the two layouts
    # (no tile hint on either side: both take the lean two-phase tile, with the same split along K where the grid is small)
    want = H.gemm_nt(inter, w, x2=True, bias=b, relu=True, out_dtype=H.X2)
    got = H.gemm_nt(planar, w, x2=True, bias=b, relu=True, out_dtype=H.X2, a_planar=True)
    assert torch.equal(got, want)
    if M >= 512 and N >= 512:  # and against the unsplit tile named explicitly: the same values up to the summation order
        ref = H.x2_decode(H.gemm_nt(inter, w, x2=True, bias=b, relu=True, out_dtype=H.X2, tile_hint=2256256))
        assert float((H.x2_decode(got) - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    with pytest.raises(RuntimeError, match="planar"):
        H.gemm_nt(planar, w, x2=True, out_dtype=H.X2, a_planar=True, tile_hint=256256)


def test_pooled_planar_output_feeds_fc1_forward_and_weight_gradient(gpu, monkeypatch):
    """RoIPool / ROIAlign with `want_hi` (training, "parity") write planar bf16x2; fc1's forward and dW on it equal the
    round-4 form (interleaved bf16x2 + a plain bf16 copy, `X2_PLANAR = False`) bit for bit."""
    from tests.util import random_rois
    from wsovod_amd.layers import functions as Fn
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(14)
    feat = torch.relu(torch.randn(2, 256, 30, 41, device=gpu)).contiguous(memory_format=torch.channels_last)
    rois = random_rois(512, 2, 240, 328, seed=15).to(gpu)  # (one image's worth of rows: both forms take the same tile)
    w = (torch.randn(512, 256 * 49, device=gpu) * 0.01).requires_grad_(True)
    b = torch.zeros(512, device=gpu, requires_grad=True)
    dy = torch.randn(512, 512, device=gpu)
    res = {}
    for planar in (True, False):
        monkeypatch.setattr(H, "X2_PLANAR", planar)
        for pooler in ("pool", "align"):
            Fn._WANT_HI.on = True
            try:
                with H.x3_mode("x2"):
                    pooled = Fn.roi_pool(feat, rois, (7, 7), 0.125, out_dtype=H.X2) if pooler == "pool" else \
                        Fn.roi_align(feat, rois, (7, 7), 0.125, 0, True, out_dtype=H.X2)
                    assert H.x2_planar_of(pooled) == planar
                    y = Fn.linear(torch.flatten(pooled, start_dim=1), w, b, relu=True, out_dtype=H.X2)
                    gw, gb = torch.autograd.grad(y, [w, b], dy)
            finally:
                Fn._WANT_HI.on = False
            res[(planar, pooler)] = (H.x2_to_f32(pooled).clone(), y.detach().clone(), gw.clone(), gb.clone())
    monkeypatch.setattr(H, "X2_PLANAR", True)
    assert H._x2_planar_ok(32 * 512, 25088) and H._x2_planar_ok(8 * 1024, 100352)      # the bench's step, config 3 / 5 shapes
    assert not H._x2_planar_ok(96 * 512, 25088) and not H._x2_planar_ok(16 * 1024, 100352)  # beyond one resource: round-4 form
    for pooler in ("pool", "align"):
        for name, a, c in zip(("pooled", "y", "dW", "db"), res[(True, pooler)], res[(False, pooler)]):
            if name == "db":  # (the bias gradient's column sums meet by float atomics: equal to the last bits, run to run)
                torch.testing.assert_close(a, c, rtol=1e-5, atol=1e-4)
            else:
                assert torch.equal(a, c), (pooler, name, float((a.float() - c.float()).abs().max()))


@pytest.mark.parametrize("M,N", [(4096, 40), (512, 64), (777, 21)])
def test_skinny_head_gemm_takes_split_k_slices_of_the_wide_tile(gpu, M, N):
    """Round 5: N <= 64 columns on K >= 2048 (the MIL [cls | det] rows on the box features) run as split-K slices of the lean
    256x256 tile (B rows past N are range-checked away) instead of a 64x64 grid: 104 -> 69 us at 32 images, 74 -> 39 at 8
    (tools/skinny_heads_ab.py).  Same values as the 64x64 grid up to the summation order, bias + fp32 output, ragged M."""
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(11)
    K = 4096
    a, b, bias = torch.randn(M, K, device=gpu), torch.randn(N, K, device=gpu) * 0.02, torch.randn(N, device=gpu)
    xa, xb = H.x2_encode(a), H.x2_encode(b)
    auto = H.gemm_nt(xa, xb, x2=True, bias=bias, out_dtype=torch.float32)
    grid = H.gemm_nt(xa, xb, x2=True, bias=bias, out_dtype=torch.float32, tile_hint=64064)
    ref = (H.x2_decode(xa).double() @ H.x2_decode(xb).double().t() + bias.double()).float()
    scale = float(ref.abs().max())
    assert float((auto - grid).abs().max()) < 1e-5 * scale
    assert float((auto - ref).abs().max()) < 3e-5 * scale
