"""Round 6, opt-in: the block-scaled forward format "f16mx" (include/wsovod_hip.h: wsovod_f16mx_encode, wsovod_gemm_f16mx) --
fp16 hi*hi on v_mfma_f32_32x32x16_f16, both cross terms as ONE v_mfma_scale_f32_32x32x64_f8f6f4 on MX-e4m3 planes.  Replaces
(opt-in) the first FC layer's forward F.linear (roi_heads/box_head.py:60-75) in the "parity_mx" precision."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nseg", [8, 2, 1])
def test_encoder_writes_the_documented_format(gpu, nseg):
    """hi = fp16 rounding; one E8M0 byte per row segment from the segment's largest |hi| exponent; q / ql = OCP e4m3 roundings
    of x / 2^s and (x - hi) / 2^(s - 11): against torch's own float16 / float8_e4m3fn casts; zeros, tiny and large segments."""
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(1)
    x = torch.randn(37, 256, device=gpu) * torch.logspace(-6, 3, 37, device=gpu).unsqueeze(1)
    x[3, 64:96] = 0.0            # an all-zero group
    x[5, :32] = 3.0e-7           # below the fp16 normal range
    x[7, 5] = 60000.0            # near the top of the fp16 range
    car, sc = H.mx_encode(x, nseg)
    G = 256 // nseg
    hi, q, ql = H.mx_decode(car, sc)
    want_hi = x.half().float()
    assert torch.equal(hi, want_hi)
    amax = want_hi.abs().view(37, nseg, G).amax(-1).clamp(min=2.0 ** -14)
    s = torch.floor(torch.log2(amax)) - 7
    assert torch.equal(sc.float() - 127.0, s)
    sq = torch.exp2(s).unsqueeze(-1)
    want_q = ((x.view(37, nseg, G) / sq).to(torch.float8_e4m3fn).float() * sq).view(37, 256)
    want_ql = (((x - want_hi).view(37, nseg, G) / (sq * 2.0 ** -11)).to(torch.float8_e4m3fn).float() * sq * 2.0 ** -11).view(37, 256)
    assert torch.equal(q, want_q) and torch.equal(ql, want_ql)
    # what the format keeps of a value: hi + ql within 2^-15 of the group's maximum
    assert float(((hi + ql - x).abs().view(37, nseg, G) / amax.unsqueeze(-1)).max()) < 2.0 ** -15


@pytest.mark.parametrize("M,N,K,nseg_a,nseg_b", [(256, 256, 128, 1, 1), (300, 520, 1024, 1, 1), (300, 520, 1536, 8, 4),
                                                 (512, 4096, 25088 // 2, 1, 1), (40, 260, 384, 2, 1), (70, 64, 32, 1, 1),
                                                 (70, 64, 64, 1, 1), (70, 64, 96, 1, 1), (70, 64, 160, 1, 1),
                                                 (70, 64, 224, 1, 1), (520, 300, 352, 1, 1)])
def test_gemm_equals_the_three_plane_contraction(gpu, M, N, K, nseg_a, nseg_b):
    """The kernel against fp64 contractions of the decoded planes: hi_a hi_b + q_a ql_b + ql_a q_b -- the instruction mix, the
    lane / K mapping of the scaled MFMA, the tied block scales (changing every K-step along a row) and the B-row permutation
    all have to be right for this to hold to fp32 accumulation error; ragged edges in M and N; and against the exact product
    within the format's precision."""
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(2)
    a = torch.randn(M, K, device=gpu) * torch.exp2(torch.randint(-3, 4, (M, K // 32), device=gpu).float()).repeat_interleave(32, 1)
    b = torch.randn(N, K, device=gpu) * 0.05
    A, sa = H.mx_encode(a, nseg_a)
    B, sb = H.mx_encode(b, nseg_b)
    got = H.gemm_mx(A, sa, B, sb)
    ha, qa, la = (t.double() for t in H.mx_decode(A, sa))
    hb, qb, lb = (t.double() for t in H.mx_decode(B, sb))
    want = ha @ hb.t() + qa @ lb.t() + la @ qb.t()
    scale = float((a.abs().double() @ b.abs().double().t()).max())
    assert float((got.double() - want).abs().max()) < 2e-6 * scale
    exact = a.double() @ b.double().t()
    assert float((got.double() - exact).abs().max()) < 2.0 ** -14 * scale


def test_gemm_epilogue_bias_relu_dropout_and_bf16x2_output(gpu):
    """alpha, bias, ReLU and the counter dropout (the SAME mask as wsovod_gemm_nt draws for this seed: the step graph and the
    backward's mask pass rely on it), outputs in fp32, bf16 and bf16x2."""
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(3)
    M, N, K = 260, 512, 512
    a, b, bias = torch.randn(M, K, device=gpu), torch.randn(N, K, device=gpu) * 0.05, torch.randn(N, device=gpu)
    A, sa = H.mx_encode(a)
    B, sb = H.mx_encode(b)
    plain = H.gemm_mx(A, sa, B, sb)
    want = torch.relu(0.5 * plain + bias)
    got = H.gemm_mx(A, sa, B, sb, bias=bias, relu=True, alpha=0.5)
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)
    g16 = H.gemm_mx(A, sa, B, sb, bias=bias, relu=True, alpha=0.5, out_dtype=torch.bfloat16)
    assert torch.equal(g16, want.to(torch.bfloat16))
    gx2 = H.gemm_mx(A, sa, B, sb, bias=bias, relu=True, alpha=0.5, out_dtype=H.X2)
    assert torch.equal(gx2, H.x2_encode(want))
    drop = H.gemm_mx(A, sa, B, sb, bias=bias, relu=True, alpha=0.5, dropout_p=0.5, dropout_seed=77)
    ref = H.gemm_nt(H.x2_encode(a), H.x2_encode(b), x2=True, bias=bias, relu=True, alpha=0.5, dropout_p=0.5, dropout_seed=77,
                    out_dtype=torch.float32)
    kept = drop != 0
    assert torch.equal(kept | (want == 0), (ref != 0) | (want == 0))  # the same keep mask
    torch.testing.assert_close(drop[kept], 2.0 * want[kept], rtol=1e-6, atol=1e-6)


def test_unit_scale_encoder_and_the_bf16x2_conversion(gpu):
    """Activations carry no scale: q = e4m3(x), ql = e4m3((x - hi) 2^11), saturating at +-448 (also far beyond the e4m3 range,
    where hi alone still carries the value); the bf16x2 -> f16mx conversion writes the same bytes as encoding the decoded
    values."""
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(4)
    x = torch.randn(33, 128, device=gpu) * torch.logspace(-5, 2, 33, device=gpu).unsqueeze(1)
    x[2, 7] = 1000.0
    x[4, 9] = -3.0e4
    car, sc = H.mx_encode(x, unit=True)
    assert sc is None
    hi, q, ql = H.mx_decode(car)
    want_hi = x.half().float()
    assert torch.equal(hi, want_hi)
    assert torch.equal(q, x.clamp(-448, 448).to(torch.float8_e4m3fn).float())
    assert torch.equal(ql, ((x - want_hi) * 2048).clamp(-448, 448).to(torch.float8_e4m3fn).float() / 2048)
    x2 = H.x2_encode(x)
    assert torch.equal(H.mx_from_x2(x2).view(torch.int32), H.mx_encode(H.x2_decode(x2), unit=True)[0].view(torch.int32))
    assert float((H.mx_to_f32(car) - x).abs().max() / x.abs().max()) < 2.0 ** -11  # (the 3e4 entry: fp16 alone)
    small = x.abs() < 400
    assert float(((H.mx_to_f32(car) - x).abs() / x.abs().clamp(min=2.0 ** -6))[small].max()) < 2.0 ** -14


@pytest.mark.parametrize("M,N,K", [(300, 544, 1024), (256, 4096, 4096)])
def test_gemm_with_unit_scale_activations_f16mx_output_and_residual(gpu, M, N, K):
    """What an FC layer / a conv of the "parity_mx" chain does: unit-scale A, row-scaled B, residual in f16mx, the output as
    unit-scale f16mx AND its plain bf16 copy: against fp64 contractions of the decoded planes, the encoder's own bytes."""
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(5)
    a = torch.relu(torch.randn(M, K, device=gpu)) * 3.0
    b = torch.randn(N, K, device=gpu) * 0.02
    bias = torch.randn(N, device=gpu)
    res = torch.randn(M, N, device=gpu)
    A, _ = H.mx_encode(a, unit=True)
    B, sb = H.mx_encode(b)
    R, _ = H.mx_encode(res, unit=True)
    ha, qa, la = (t.double() for t in H.mx_decode(A))
    hb, qb, lb = (t.double() for t in H.mx_decode(B, sb))
    want = torch.relu(ha @ hb.t() + qa @ lb.t() + la @ qb.t() + bias.double() + H.mx_to_f32(R).double()).float()
    got32 = H.gemm_mx(A, None, B, sb, bias=bias, relu=True, residual=R, residual_fmt=H.MX)
    scale = float((a.abs().double() @ b.abs().double().t()).max())
    assert float((got32 - want).abs().max()) < 2e-6 * scale
    cb = torch.empty(M, N, dtype=torch.bfloat16, device=gpu)
    gmx = H.gemm_mx(A, None, B, sb, bias=bias, relu=True, residual=R, residual_fmt=H.MX, out_dtype=H.MX, out_bf16=cb)
    assert torch.equal(gmx.view(torch.int32), H.mx_encode(got32, unit=True)[0].view(torch.int32))
    assert torch.equal(cb, got32.to(torch.bfloat16))
    gx2 = H.gemm_mx(A, None, B, sb, bias=bias, relu=True, residual=H.x2_encode(res), residual_fmt=H.X2, out_dtype=H.X2)
    want2 = torch.relu(ha @ hb.t() + qa @ lb.t() + la @ qb.t() + bias.double() + H.x2_decode(H.x2_encode(res)).double()).float()
    assert float((H.x2_decode(gx2) - want2).abs().max()) < 2e-6 * scale


@pytest.mark.parametrize("n,Hh,Ww,Cin,Cout,k,dil,Cin2", [(2, 19, 23, 64, 256, 3, 2, 0), (3, 20, 17, 128, 256, 3, 2, 64),
                                                          (1, 38, 50, 256, 512, 3, 2, 128), (2, 9, 11, 32, 48, 1, 1, 0),
                                                          (2, 12, 10, 64, 64, 3, 1, 32)])
def test_conv_equals_the_three_plane_convolution(gpu, n, Hh, Ww, Cin, Cout, k, dil, Cin2):
    """The implicit-GEMM form (res4 / res5: 3x3, dilation 2, padding 2, the fused 1x1 projection shortcut as extra K-steps)
    against torch convolutions of the decoded planes in fp64: borders, ragged M / N tiles, every tap x channel-chunk K-step,
    ring / buffer positions of K-step counts with every remainder mod 6."""
    import torch.nn.functional as F
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(6)
    pad = dil * (k - 1) // 2
    x = torch.relu(torch.randn(n, Hh, Ww, Cin, device=gpu)) * 2.0
    w = torch.randn(Cout, k, k, Cin, device=gpu) * 0.05
    bias = torch.randn(Cout, device=gpu)
    X, _ = H.mx_encode(x.view(-1, Cin), unit=True)
    X = X.view(n, Hh, Ww, Cin)
    rows = w.reshape(Cout, -1)
    x2 = w2 = X2 = None
    if Cin2:
        x2 = torch.randn(n, Hh, Ww, Cin2, device=gpu)
        w2 = torch.randn(Cout, Cin2, device=gpu) * 0.05
        X2 = H.mx_encode(x2.view(-1, Cin2), unit=True)[0].view(n, Hh, Ww, Cin2)
        rows = torch.cat([rows, w2], dim=1)
    W, sw = H.mx_encode(rows.contiguous())
    geom = dict(n_img=n, H=Hh, W=Ww, Cin=Cin, Ho=Hh, Wo=Ww, KH=k, KW=k, stride=1, pad=pad, dil=dil)
    got = H.gemm_mx(X, None, W, sw, conv=geom, A2=X2, bias=bias, relu=True).view(n, Hh, Ww, Cout)

    def conv(xp, wp):  # NHWC planes -> NCHW fp64 conv -> NHWC
        return F.conv2d(xp.double().permute(0, 3, 1, 2), wp.double(), None, 1, pad, dil).permute(0, 2, 3, 1)

    hx, qx, lx = H.mx_decode(X)
    hw, qw, lw = (t[:, :k * k * Cin].reshape(Cout, k, k, Cin).permute(0, 3, 1, 2) for t in H.mx_decode(W, sw))
    want = conv(hx, hw) + conv(qx, lw) + conv(lx, qw)
    if Cin2:
        h2, q2, l2 = (t.double().view(-1, Cin2) for t in H.mx_decode(X2))
        hv, qv, lv = (t[:, k * k * Cin:].double() for t in H.mx_decode(W, sw))
        want = want + (h2 @ hv.t() + q2 @ lv.t() + l2 @ qv.t()).view(n, Hh, Ww, Cout)
    want = torch.relu(want + bias.double())
    scale = float(want.abs().max())
    assert float((got.double() - want).abs().max()) < 1e-5 * scale


@pytest.mark.parametrize("form", ["gemm", "conv", "conv_shortcut_f16mx_out"])
def test_last_round_of_tiles_as_split_k_slices(gpu, form, monkeypatch):
    """More tiles than CUs with a partly filled last round: the rows of that round go to a second launch of the kernel as S
    copies of the grid over slices of K + a finalize pass (fp32 workspace, the same epilogue chain and output formats).
    Against the single launch (WSOVOD_MX_TAIL=0) to fp32 summation-order error, output bytes of the f16mx form included
    up to that error."""
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(7)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    if form == "gemm":
        M, N, K = (cus + 18) * 256 - 100, 256, 1536
        a = torch.relu(torch.randn(M, K, device=gpu))
        b = torch.randn(N, K, device=gpu) * 0.03
        bias = torch.randn(N, device=gpu)
        A, _ = H.mx_encode(a, unit=True)
        B, sb = H.mx_encode(b)
        run = lambda: H.gemm_mx(A, None, B, sb, bias=bias, relu=True, dropout_p=0.5, dropout_seed=11)
    else:
        n, Hh, Ww, Cin, Cout = (cus * 256) // 7500 + 4, 75, 100, 128, 256
        x = torch.relu(torch.randn(n * Hh * Ww, Cin, device=gpu))
        Cin2 = 64 if form != "conv" else 0
        w = torch.randn(Cout, 9 * Cin + Cin2, device=gpu) * 0.03
        bias = torch.randn(Cout, device=gpu)
        X = H.mx_encode(x, unit=True)[0].view(n, Hh, Ww, Cin)
        X2 = H.mx_encode(torch.randn(n * Hh * Ww, Cin2, device=gpu), unit=True)[0].view(n, Hh, Ww, Cin2) if Cin2 else None
        res = None if Cin2 else H.mx_encode(torch.randn(n * Hh * Ww, Cout, device=gpu), unit=True)[0]
        W, sw = H.mx_encode(w)
        geom = dict(n_img=n, H=Hh, W=Ww, Cin=Cin, Ho=Hh, Wo=Ww, KH=3, KW=3, stride=1, pad=2, dil=2)
        fmt = H.MX if Cin2 else torch.float32
        run = lambda: H.gemm_mx(X, None, W, sw, conv=geom, A2=X2, bias=bias, relu=True, residual=res,
                                residual_fmt=H.MX if res is not None else None, out_dtype=fmt)
    got = run()
    monkeypatch.setenv("WSOVOD_MX_TAIL", "0")
    want = run()
    if form == "conv_shortcut_f16mx_out":
        got, want = H.mx_to_f32(got), H.mx_to_f32(want)
        tol = 2.0 ** -13  # (a last-bit difference of the fp32 sum can move an e4m3 rounding of the lo plane)
    else:
        tol = 2e-6
    assert not torch.equal(got[: got.shape[0] // 2], torch.zeros_like(got[: got.shape[0] // 2]))
    assert float((got - want).abs().max()) <= tol * float(want.abs().max())
    assert torch.equal(got == 0, want == 0) or form != "gemm"  # the same ReLU / dropout pattern


def test_optimizer_kernels_refresh_the_f16mx_weight_operand(gpu):
    """A trained weight's f16mx operand has ONE scale for the tensor (chosen at its first encode, kept as a byte in device
    memory): the multi-tensor SGD kernel and the fused weight-gradient + update kernel re-encode it element-wise in their pass
    -- the bytes a fresh encode of the updated parameter with that scale writes -- and the cache is re-stamped (no encode pass
    in steady state)."""
    from wsovod_amd.engine.trainer import _mx_shadow, _restamp_shadow
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(8)
    NI, NJ, M = 512, 1024, 256
    w = (torch.randn(NI, NJ, device=gpu) * 0.01).requires_grad_(True)
    car, sc = H.mx_cached(w, tensor_scale=True)
    byte = w._mx_scale
    assert int(byte) == int(torch.floor(torch.log2(w.detach().abs().max()))) - 7 + H.MX_WEIGHT_HEADROOM + 127
    assert torch.equal(sc, byte.expand(NI, 1)) and torch.equal(car.view(torch.int32), H.mx_encode(w.detach(), tensor_byte=byte)[0].view(torch.int32))
    hi, q, ql = H.mx_decode(car, sc)
    assert float((hi + ql - w.detach()).abs().max() / w.detach().abs().max()) < 2.0 ** -14
    g, buf = torch.randn(NI, NJ, device=gpu) * 1e-3, torch.zeros(NI, NJ, device=gpu)
    sh = _mx_shadow(w)
    assert sh is not None and sh[0] is car
    with torch.no_grad():
        H.sgd_momentum_multi([(w.data, g, buf, sh, 0.5, 1e-4)], 0.9)
        torch.autograd.graph.increment_version(w)
        _restamp_shadow(w, sh)
    assert torch.equal(car.view(torch.int32), H.mx_encode(w.detach(), tensor_byte=byte)[0].view(torch.int32))
    assert H.mx_cached(w, tensor_scale=True)[0] is car  # the cache is current: no encode
    dA = (torch.randn(M, NI, device=gpu) * 0.1).to(torch.bfloat16)
    x = torch.randn(M, NJ, device=gpu).to(torch.bfloat16)
    with torch.no_grad():
        H.gemm_tn_sgd(dA, x, w.data, buf, _mx_shadow(w), 0.01, 1e-4, 0.9)
    assert torch.equal(car.view(torch.int32), H.mx_encode(w.detach(), tensor_byte=byte)[0].view(torch.int32))
    with torch.no_grad():  # a change behind the optimizer's back (a loaded checkpoint): full encode, scale re-derived IN PLACE
        w.mul_(8.0)
    car2, _ = H.mx_cached(w, tensor_scale=True)
    assert car2 is car and w._mx_scale is byte and int(byte) == int(torch.floor(torch.log2(w.detach().abs().max()))) - 7 + H.MX_WEIGHT_HEADROOM + 127
