"""GPU parity: MFMA contraction kernel (plain + implicit-GEMM conv) vs a torch fp32/fp64 CPU reference."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref_mm(A, B):
    return (A.double() @ B.double().t()).float()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("tile", [0, 128128, 128064, 64128, 64064, 256256, 256128, 1128128, 1128064, 1256064, 3256128, 4128128, 3128128])
@pytest.mark.parametrize("shape", [(512, 4096, 1024), (100, 40, 256), (333, 129, 192), (64, 4, 4096)])
def test_gemm_nt_plain(gpu, dtype, tile, shape):
    from wsovod_amd.layers import hip_ops

    M, N, K = shape
    torch.manual_seed(0)
    A = torch.randn(M, K).to(dtype)
    B = torch.randn(N, K).to(dtype)
    ref = _ref_mm(A, B)
    out = hip_ops.gemm_nt(A.to(gpu), B.to(gpu), out_dtype=torch.float32, tile_hint=tile)
    # fp32 path: exact-fp32 MFMA (fmaf chain) -> 1e-5 relative to sum|a||b|; bf16 inputs are exact
    # products accumulated in fp32 -> same bound
    tol = 2e-6 * K ** 0.5 * 4
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=tol * 10)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogue(gpu, dtype):
    from wsovod_amd.layers import hip_ops

    M, N, K = 200, 96, 320
    torch.manual_seed(1)
    A = torch.randn(M, K).to(dtype)
    B = torch.randn(N, K).to(dtype)
    bias = torch.randn(N)
    rs = torch.rand(M) + 0.5
    res = torch.randn(M, N)
    grp = torch.randint(0, 3, (M,), dtype=torch.int32)
    ga = torch.randn(3, N)
    ref = torch.relu(_ref_mm(A, B) * 0.5 * rs[:, None] + bias[None] + res) + ga[grp.long()]
    out_t = torch.zeros(N, 256, device=gpu)
    out = hip_ops.gemm_nt(A.to(gpu), B.to(gpu), out_dtype=torch.float32, alpha=0.5, row_scale=rs.to(gpu),
                          bias=bias.to(gpu), residual=res.to(gpu), relu=True, row_group=grp.to(gpu),
                          group_add=ga.to(gpu), out_t=out_t)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(out_t.cpu()[:, :M], ref.t(), rtol=1e-4, atol=1e-3)
    assert torch.all(out_t.cpu()[:, M:] == 0)
    # backward-style mask + accumulate
    mask = torch.randn(M, N)
    acc0 = torch.randn(M, N)
    out2 = acc0.clone().to(gpu)
    hip_ops.gemm_nt(A.to(gpu), B.to(gpu), out=out2, mask_src=mask.to(gpu), mask_scale=2.0, accumulate=True)
    ref2 = acc0 + torch.where(mask > 0, _ref_mm(A, B) * 2.0, torch.zeros(()))
    torch.testing.assert_close(out2.cpu(), ref2, rtol=1e-4, atol=1e-3)
    # bf16 output rounding
    out3 = hip_ops.gemm_nt(A.to(gpu), B.to(gpu), out_dtype=torch.bfloat16)
    torch.testing.assert_close(out3.cpu().float(), _ref_mm(A, B), rtol=1e-2, atol=1e-1)


@pytest.mark.parametrize("shape", [(512, 4096, 1024), (100, 40, 256), (333, 129, 192), (700, 300, 64), (256, 512, 8192)])
def test_gemm_8phase_tile(gpu, shape):
    """The 8-wavefront staggered 256x256 tile (bf16 only): plain, ragged edges, short and long K."""
    from wsovod_amd.layers import hip_ops

    M, N, K = shape
    torch.manual_seed(0)
    A = torch.randn(M, K).to(torch.bfloat16)
    B = torch.randn(N, K).to(torch.bfloat16)
    ref = _ref_mm(A, B)
    out = hip_ops.gemm_nt(A.to(gpu), B.to(gpu), out_dtype=torch.float32, tile_hint=8256256)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=2e-6 * K ** 0.5 * 40)
    # same summation order as the 16-wavefront tile: bit-identical
    out16 = hip_ops.gemm_nt(A.to(gpu), B.to(gpu), out_dtype=torch.float32, tile_hint=256256)
    assert torch.equal(out, out16)
    # the two-phase form of the 8-wavefront tile: the same products in the same order
    assert torch.equal(out, hip_ops.gemm_nt(A.to(gpu), B.to(gpu), out_dtype=torch.float32, tile_hint=2256256))
    with pytest.raises(RuntimeError, match="bf16 only"):
        hip_ops.gemm_nt(A.float().to(gpu), B.float().to(gpu), tile_hint=8256256)


@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("variant", ["fast", "fast_dropout", "generic"])
def test_gemm_8phase_epilogues(gpu, out_dtype, variant):
    """Vectorised fast epilogue (bias / residual / ReLU / dropout) and the generic one (row scale, group add, mask,
    transposed copy, accumulate) of the 8-phase tile against the 16-wavefront tile's results."""
    from wsovod_amd.layers import hip_ops

    M, N, K = 600, 512, 320
    torch.manual_seed(3)
    A = torch.randn(M, K, device=gpu).to(torch.bfloat16)
    B = torch.randn(N, K, device=gpu).to(torch.bfloat16)
    bias = torch.randn(N, device=gpu)
    res = torch.randn(M, N, device=gpu).to(out_dtype)
    kw = dict(alpha=0.5, bias=bias, residual=res, relu=True, out_dtype=out_dtype)
    if variant == "fast_dropout":
        kw.update(dropout_p=0.5, dropout_seed=77)
    if variant == "generic":
        kw.update(row_scale=torch.rand(M, device=gpu) + 0.5, row_group=torch.randint(0, 3, (M,), dtype=torch.int32, device=gpu),
                  group_add=torch.randn(3, N, device=gpu), mask_src=torch.randn(M, N, device=gpu), mask_scale=2.0)
    outs = {}
    for tile in (256256, 8256256):
        out_t = torch.zeros(N, 640, device=gpu) if variant == "generic" else None
        outs[tile] = (hip_ops.gemm_nt(A, B, tile_hint=tile, out_t=out_t, **kw), out_t)
    assert torch.equal(outs[256256][0], outs[8256256][0])
    if variant == "generic":
        assert torch.equal(outs[256256][1], outs[8256256][1])
    ref = torch.relu(_ref_mm(A.cpu(), B.cpu()) * 0.5 + bias.cpu()[None] + res.cpu().float())
    if variant == "fast":
        torch.testing.assert_close(outs[8256256][0].cpu().float(), ref, rtol=2e-2 if out_dtype == torch.bfloat16 else 1e-4,
                                   atol=0.2 if out_dtype == torch.bfloat16 else 1e-3)
    if variant == "generic" and out_dtype == torch.float32:
        acc0 = torch.randn(M, N, device=gpu)
        o = {}
        for tile in (256256, 8256256):
            o[tile] = acc0.clone()
            hip_ops.gemm_nt(A, B, out=o[tile], accumulate=True, tile_hint=tile)
        assert torch.equal(o[256256], o[8256256])


@pytest.mark.parametrize("shape", [(8192, 512, 768), (249, 256, 512), (1000, 136, 264), (64, 8, 8), (5000, 1024, 256)])
def test_gemm_tn_transposed_reads(gpu, shape):
    """dW-style contraction over the slow index of two row-major bf16 operands (ds_read_b64_tr_b16 fragments):
    ragged reduction length, ragged tile edges, strided operands, accumulate; and bit-identity with the NT tile fed
    with explicit transposes (same products, same order)."""
    from wsovod_amd.layers import hip_ops

    Mred, NI, NJ = shape
    torch.manual_seed(4)
    P = torch.randn(Mred, NI + 8).to(torch.bfloat16)[:, :NI]  # row stride > NI
    Q = torch.randn(Mred, NJ).to(torch.bfloat16)
    ref = (P.double().t() @ Q.double()).float()
    Pg, Qg = P.to(gpu), Q.to(gpu)
    torch.testing.assert_close(hip_ops.gemm_tn(Pg, Qg).cpu(), ref, rtol=1e-4, atol=2e-6 * Mred ** 0.5 * 40)
    out = hip_ops.gemm_tn(Pg, Qg, split_tail=False)  # fixed summation order (no K slices meeting by atomics)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=2e-6 * Mred ** 0.5 * 40)
    Mp = (Mred + 63) // 64 * 64
    nt = hip_ops.gemm_nt(hip_ops.transpose_cast(Pg.contiguous(), torch.bfloat16, ld_dst=Mp),
                         hip_ops.transpose_cast(Qg, torch.bfloat16, ld_dst=Mp), out_dtype=torch.float32, tile_hint=8256256)
    assert torch.equal(out, nt)
    acc = torch.randn(NI, NJ, device=gpu)
    out2 = hip_ops.gemm_tn(Pg, Qg, out=acc.clone(), alpha=0.5, accumulate=True)
    torch.testing.assert_close(out2.cpu(), acc.cpu() + 0.5 * ref, rtol=1e-4, atol=2e-6 * Mred ** 0.5 * 40)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_linear_group_equals_separate_linears(gpu, dtype):
    """Heads sharing one input: grouped autograd node (one dX GEMM, one dW contraction) vs one Linear per head; and the
    joined form the ROI heads use -- [cls | det] given as two row blocks of one head, joined with the box regression into
    ONE forward GEMM (N = 44), outputs = column blocks of its result."""
    from wsovod_amd.layers import functions as Fn

    torch.manual_seed(5)
    M, K = 1000, 512
    x0 = torch.randn(M, K, device=gpu).to(dtype)
    specs = [(40, False, torch.float32), (4, False, torch.float32), (1024, True, None)]
    params = [(torch.randn(n, K, device=gpu) * 0.05, torch.randn(n, device=gpu) * 0.1) for n, _, _ in specs]
    gys = None
    res = {}
    for mode in ("separate", "group", "joined"):
        x = x0.clone().requires_grad_(True)
        ws = [(w.clone().requires_grad_(True), b.clone().requires_grad_(True)) for w, b in params]
        if mode == "separate":
            ys = [Fn.linear(x, w, b, relu=r, out_dtype=od) for (w, b), (_, r, od) in zip(ws, specs)]
        elif mode == "group":
            ys = Fn.linear_group(x, [(w, b, r, od) for (w, b), (_, r, od) in zip(ws, specs)])
        else:  # the 40-row head as two leaf blocks of 20 rows
            w0 = [params[0][0][:20].clone().requires_grad_(True), params[0][0][20:].clone().requires_grad_(True)]
            b0 = [params[0][1][:20].clone().requires_grad_(True), params[0][1][20:].clone().requires_grad_(True)]
            heads = [(w0, b0, False, torch.float32)] + [(w, b, r, od) for (w, b), (_, r, od) in zip(ws[1:], specs[1:])]
            ys = Fn.linear_group(x, heads, joins=[[0, 1]])
            assert ys[0].data_ptr() + 40 * 4 == ys[1].data_ptr() and ys[0].stride(0) == 44  # one (M, 44) result
        if gys is None:
            gys = [torch.randn_like(y) for y in ys]
        torch.autograd.backward(list(ys), gys)
        if mode == "joined":
            ws[0] = (torch.cat([w.grad for w in w0]), torch.cat([b.grad for b in b0]))
            res[mode] = ([y.detach() for y in ys], x.grad, [ws[0]] + [(w.grad, b.grad) for w, b in ws[1:]])
        else:
            res[mode] = ([y.detach() for y in ys], x.grad, [(w.grad, b.grad) for w, b in ws])
    for a, b in zip(res["separate"][0], res["group"][0]):
        assert torch.equal(a, b)  # same forward GEMMs
    tol = dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)
    for a, b in zip(res["separate"][0], res["joined"][0]):  # same products per output, another tile shape
        torch.testing.assert_close(b.float(), a.float(), rtol=1e-5, atol=1e-5)
    for mode in ("group", "joined"):
        # dx: one K = sum N_h contraction vs a bf16-rounded sum of per-head products
        torch.testing.assert_close(res[mode][1].float(), res["separate"][1].float(), **tol)
        for (wa, ba), (wb, bb) in zip(res["separate"][2], res[mode][2]):
            torch.testing.assert_close(wb, wa, rtol=1e-4, atol=1e-4)
            torch.testing.assert_close(bb, ba, rtol=1e-4, atol=1e-4)
    with pytest.raises(RuntimeError, match="join"):
        Fn.linear_group(x0, [(params[0][0], params[0][1], False, torch.float32),
                             (params[2][0], params[2][1], True, None)], joins=[[0, 1]])


def test_fused_stem_conv1_is_bit_identical_to_im2col_gemm(gpu):
    """uint8 images -> normalise -> 3x3/s2 conv -> ReLU in one kernel vs the im2col operand + GEMM (ragged sizes)."""
    from wsovod_amd.layers import hip_ops as H

    g = torch.Generator().manual_seed(9)
    N, Hp, Wp = 3, 77, 130
    img = torch.randint(0, 256, (N, 3, Hp, Wp), dtype=torch.uint8, generator=g).to(gpu)
    sizes = torch.tensor([[77, 130], [64, 101], [33, 130]], dtype=torch.int32, device=gpu)
    mean, std = (102.98, 115.95, 122.77), (57.4, 57.1, 58.4)
    w32 = torch.zeros(64, 32)
    w32[:, :27] = torch.randn(64, 27, generator=g) * 0.1
    w32 = w32.to(torch.bfloat16).to(gpu)
    bias = torch.randn(64, generator=g).to(gpu)
    a, ho, wo = H.stem_im2col(img, sizes, mean, std, torch.bfloat16)
    ref = H.gemm_nt(a, w32, bias=bias, relu=True, out_dtype=torch.bfloat16).view(N, ho, wo, 64)
    out = H.stem_conv1(img, sizes, mean, std, w32, bias)
    assert out.shape == ref.shape and torch.equal(out, ref)


def test_staggered_tiles_race_screen(gpu):
    """The 8-phase NT tile and the transposed-read TN kernel order their LDS-DMA prefetch by counted waits and raw
    barriers only.  Screen for races: many shapes (short / long / ragged K, ragged edges), repeated launches, with a
    second stream hammering HBM to perturb DMA timing; every result must be bit-identical to the 16-wavefront tile
    (whose LDS hand-off is a plain __syncthreads) fed with the same operands."""
    from wsovod_amd.layers import hip_ops as H

    g = torch.Generator(device="cuda").manual_seed(123)
    side = torch.cuda.Stream()
    junk = torch.empty(64 * 1024 * 1024, device=gpu)
    shapes = [(256, 256, 64), (512, 768, 128), (300, 520, 192), (1024, 1024, 1000), (2048, 512, 4096), (777, 1333, 2048),
              (4096, 4096, 512), (256, 4096, 8192), (1536, 256, 320)]
    for (M, N, K) in shapes:
        K8 = (K + 7) // 8 * 8
        A = (torch.rand(M, K8, device=gpu, generator=g) - 0.5).to(torch.bfloat16)
        B = (torch.rand(N, K8, device=gpu, generator=g) - 0.5).to(torch.bfloat16)
        ref = H.gemm_nt(A, B, out_dtype=torch.float32, tile_hint=256256)
        At, Bt = A.t().contiguous(), B.t().contiguous()  # (K, M), (K, N): operands of the TN form
        for rep in range(6):
            with torch.cuda.stream(side):
                junk.mul_(1.0001)
            out = H.gemm_nt(A, B, out_dtype=torch.float32, tile_hint=8256256)
            assert torch.equal(out, ref), (M, N, K, rep, "8-phase")
            out = H.gemm_nt(A, B, out_dtype=torch.float32, tile_hint=2256256)  # its two-phase form (round 3)
            assert torch.equal(out, ref), (M, N, K, rep, "2-phase")
            if M % 8 == 0 and N % 8 == 0:
                tn = H.gemm_tn(At, Bt, split_tail=False)  # (fixed summation order: no K slices meeting by atomics)
                assert torch.equal(tn, ref), (M, N, K, rep, "tn")
    # the bf16x2 (three-MFMA) forms of the two schedules: same products in the same order
    for (M, N, K) in [(512, 768, 128), (777, 1333, 2048), (2048, 512, 4096), (256, 4096, 8192)]:
        K32 = (K + 31) // 32 * 32
        A = H.x2_encode(torch.rand(M, K32, device=gpu, generator=g) - 0.5)
        B = H.x2_encode(torch.rand(N, K32, device=gpu, generator=g) - 0.5)
        ref = H.gemm_nt(A, B, x2=True, out_dtype=torch.float32, tile_hint=256256)  # 16 wavefronts, plain __syncthreads
        for rep in range(6):
            with torch.cuda.stream(side):
                junk.mul_(1.0001)
            assert torch.equal(H.gemm_nt(A, B, x2=True, out_dtype=torch.float32, tile_hint=2256256), ref), (M, N, K, rep, "x3 2-phase")
            assert torch.equal(H.gemm_nt(A, B, x2=True, out_dtype=torch.float32, tile_hint=8256256), ref), (M, N, K, rep, "x3 8-phase")
    torch.cuda.synchronize()


def test_gemm_dropout_statistics(gpu):
    from wsovod_amd.layers import hip_ops

    M, N, K = 512, 1024, 64
    A = torch.ones(M, K, device=gpu)
    B = torch.ones(N, K, device=gpu)
    out = hip_ops.gemm_nt(A, B, dropout_p=0.5, dropout_seed=1234)
    kept = (out > 0).float().mean().item()
    assert abs(kept - 0.5) < 0.01  # 524288 Bernoulli(0.5) draws: sigma = 7e-4
    assert torch.all((out == 0) | (out == 2.0 * K))
    out_b = hip_ops.gemm_nt(A, B, dropout_p=0.5, dropout_seed=1234)
    assert torch.equal(out, out_b)  # counter-based: reproducible
    out_c = hip_ops.gemm_nt(A, B, dropout_p=0.5, dropout_seed=99)
    assert not torch.equal(out, out_c)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [
    dict(Cin=64, Cout=64, H=38, W=50, k=3, stride=1, pad=1, dil=1),
    dict(Cin=128, Cout=256, H=19, W=25, k=3, stride=1, pad=2, dil=2),
    dict(Cin=64, Cout=128, H=19, W=25, k=1, stride=1, pad=0, dil=1),
    dict(Cin=64, Cout=96, H=20, W=26, k=3, stride=2, pad=1, dil=1),
    dict(Cin=64, Cout=64, H=37, W=71, k=3, stride=1, pad=1, dil=1),  # halo-tile kernel with ragged tiles
])
@pytest.mark.parametrize("tile", [0, 256256, 8256256, 256128, 1128128, 1128064, 1256064, 3256128, 4128128])
def test_conv_implicit_gemm(gpu, dtype, cfg, tile):
    from wsovod_amd.layers import hip_ops

    if tile == 8256256 and dtype != torch.bfloat16:
        pytest.skip("the 8-phase tile is bf16 only")

    torch.manual_seed(2)
    n = 2
    x = torch.randn(n, cfg["Cin"], cfg["H"], cfg["W"]).to(dtype)
    w = (torch.randn(cfg["Cout"], cfg["Cin"], cfg["k"], cfg["k"]) * 0.05).to(dtype)
    bias = torch.randn(cfg["Cout"])
    ref = F.relu(F.conv2d(x.double(), w.double(), bias.double(), cfg["stride"], cfg["pad"], cfg["dil"])).float()
    Ho, Wo = ref.shape[2:]
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(gpu)
    w_k = w.permute(0, 2, 3, 1).reshape(cfg["Cout"], -1).contiguous().to(gpu)
    geom = dict(n_img=n, H=cfg["H"], W=cfg["W"], Cin=cfg["Cin"], Ho=Ho, Wo=Wo, KH=cfg["k"], KW=cfg["k"],
                stride=cfg["stride"], pad=cfg["pad"], dil=cfg["dil"])
    out = hip_ops.gemm_nt(x_nhwc, w_k, conv=geom, bias=bias.to(gpu), relu=True, out_dtype=torch.float32, tile_hint=tile)
    out = out.view(n, Ho, Wo, cfg["Cout"]).permute(0, 3, 1, 2).cpu()
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=2e-3)


def test_gemm_argument_errors(gpu):
    from wsovod_amd.layers import hip_ops

    A = torch.randn(8, 30, device=gpu)
    B = torch.randn(8, 30, device=gpu)
    with pytest.raises(RuntimeError, match="multiple"):
        hip_ops.gemm_nt(A, B)  # K=30 is not a multiple of 4 fp32 elements (16-B chunks)
    with pytest.raises(RuntimeError):
        hip_ops.gemm_nt(A.cpu(), B.cpu())


def test_conv3x3_c64_halo_kernel_with_residual(gpu):
    """The Cin=Cout=64 halo-tile kernel (auto dispatch, bf16) incl. residual add, vs fp64 conv."""
    from wsovod_amd.layers import hip_ops

    torch.manual_seed(7)
    n, H, W = 3, 41, 67
    x = torch.randn(n, 64, H, W).to(torch.bfloat16)
    w = (torch.randn(64, 64, 3, 3) * 0.05).to(torch.bfloat16)
    bias = torch.randn(64)
    res = torch.randn(n, H, W, 64).to(torch.bfloat16)
    ref = F.relu(F.conv2d(x.double(), w.double(), bias.double(), 1, 1) + res.double().permute(0, 3, 1, 2)).float()
    geom = dict(n_img=n, H=H, W=W, Cin=64, Ho=H, Wo=W, KH=3, KW=3, stride=1, pad=1, dil=1)
    out = hip_ops.gemm_nt(x.permute(0, 2, 3, 1).contiguous().to(gpu), w.permute(0, 2, 3, 1).reshape(64, -1).contiguous().to(gpu),
                          conv=geom, bias=bias.to(gpu), relu=True, residual=res.view(-1, 64).to(gpu),
                          out_dtype=torch.float32)
    torch.testing.assert_close(out.view(n, H, W, 64).permute(0, 3, 1, 2).cpu(), ref, rtol=1e-3, atol=5e-3)
    from wsovod_amd import _lib
    _lib.profile_reset(); _lib.profile_enable(True)
    hip_ops.gemm_nt(x.permute(0, 2, 3, 1).contiguous().to(gpu), w.permute(0, 2, 3, 1).reshape(64, -1).contiguous().to(gpu),
                    conv=geom, out_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    names = [e["name"] for e in _lib.profile_collect() if e["launches"] > 0]
    _lib.profile_enable(False)
    assert "conv3x3_c64_halo_bf16" in names  # the dedicated kernel is the one that ran


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 40, 64), (3, 41, 67), (1, 300, 400)])
@pytest.mark.parametrize("with_res", [False, True])
def test_conv3x3_c64_fused_maxpool_equals_conv_then_pool(gpu, shape, with_res):
    """geom.pool = 2: the 64-channel kernel's epilogue applies MaxPool2d(2, 2) (stem tail / res2 block tail).  Same bits
    as the unfused conv followed by the pool kernel (rounding to bf16 is monotonic), odd sizes drop the last row/column
    as the pool does; any other conv shape with pool set is refused."""
    from wsovod_amd.layers import hip_ops

    torch.manual_seed(11)
    n, H, W = shape
    x = torch.randn(n, H, W, 64, device=gpu).to(torch.bfloat16)
    w = (torch.randn(64, 9 * 64, device=gpu) * 0.05).to(torch.bfloat16)
    bias = torch.randn(64, device=gpu)
    res = torch.randn(n * H * W, 64, device=gpu).to(torch.bfloat16) if with_res else None
    geom = dict(n_img=n, H=H, W=W, Cin=64, Ho=H, Wo=W, KH=3, KW=3, stride=1, pad=1, dil=1)
    full = hip_ops.gemm_nt(x, w, conv=geom, bias=bias, relu=True, residual=res, out_dtype=torch.bfloat16)
    want = hip_ops.maxpool2x2_nhwc(full.view(n, H, W, 64), 2)
    got = hip_ops.gemm_nt(x, w, conv=dict(geom, pool=2), bias=bias, relu=True, residual=res, out_dtype=torch.bfloat16)
    assert got.shape == (n * (H // 2) * (W // 2), 64)
    assert torch.equal(got.view(n, H // 2, W // 2, 64), want)
    x128 = torch.randn(1, 16, 16, 128, device=gpu).to(torch.bfloat16)
    w128 = torch.randn(64, 9 * 128, device=gpu).to(torch.bfloat16)
    with pytest.raises(RuntimeError, match="pool"):
        hip_ops.gemm_nt(x128, w128, conv=dict(n_img=1, H=16, W=16, Cin=128, Ho=16, Wo=16, KH=3, KW=3, stride=1, pad=1,
                                              dil=1, pool=2), out_dtype=torch.bfloat16)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 300, 400), (8, 121, 163), (8, 128, 128), (24, 150, 200), (2, 300, 400, "lds")])
@pytest.mark.parametrize("with_res", [False, True])
@pytest.mark.parametrize("pool", [0, 2])
def test_conv3x3_c64_persistent_kernel_equals_one_tile_kernel(gpu, shape, with_res, pool, monkeypatch):
    """From 512 tiles up the 64-channel 3x3 conv runs as one persistent workgroup per CU: the weight fragments in
    registers (AGPR operands of hand-issued MFMAs), three halo-patch buffers whose requests are issued inside the MFMA
    loop two tiles ahead, residual rows fetched ahead of the MFMAs, buffer-addressed stores.  Same products in the same
    order as the one-tile kernel: the outputs are bit-identical -- ragged image edges, residual and fused pool included,
    at exactly two tiles per workgroup (8 x 128 x 128: the look-ahead runs past the last tile from the start), at three
    and at seventeen; the round-2 form (weights read from LDS at every step, "lds") as well -- and all agree with an fp64
    convolution."""
    from wsovod_amd.layers import hip_ops

    torch.manual_seed(13)
    monkeypatch.setenv("WSOVOD_C64_WREG", "0" if len(shape) == 4 else "1")
    n, H, W = shape[:3]
    x = torch.randn(n, H, W, 64, device=gpu).to(torch.bfloat16)
    w = (torch.randn(64, 9 * 64, device=gpu) * 0.05).to(torch.bfloat16)
    bias = torch.randn(64, device=gpu)
    res = torch.randn(n * H * W, 64, device=gpu).to(torch.bfloat16) if with_res else None
    geom = dict(n_img=n, H=H, W=W, Cin=64, Ho=H, Wo=W, KH=3, KW=3, stride=1, pad=1, dil=1, pool=pool)
    monkeypatch.setenv("WSOVOD_C64_PERSIST", "0")
    one = hip_ops.gemm_nt(x, w, conv=geom, bias=bias, relu=True, residual=res, out_dtype=torch.bfloat16)
    monkeypatch.setenv("WSOVOD_C64_PERSIST", "1")
    per = hip_ops.gemm_nt(x, w, conv=geom, bias=bias, relu=True, residual=res, out_dtype=torch.bfloat16)
    assert torch.equal(one, per)
    for _ in range(8):  # counted LDS / vector-memory waits: a stale fragment or patch would differ from run to run
        again = hip_ops.gemm_nt(x, w, conv=geom, bias=bias, relu=True, residual=res, out_dtype=torch.bfloat16)
        assert torch.equal(again, per)
    if not pool:
        img = 1
        ref = F.conv2d(x[img].permute(2, 0, 1)[None].double(), w.view(64, 3, 3, 64).permute(0, 3, 1, 2).double(),
                       bias.double(), 1, 1)[0]
        if with_res:
            ref = ref + res.view(n, H, W, 64)[img].permute(2, 0, 1).double()
        torch.testing.assert_close(per.view(n, H, W, 64)[img].permute(2, 0, 1).double(), F.relu(ref), rtol=1e-2, atol=2e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("pool", [0, 2])
def test_conv3x3_c64_persistent_kernel_epilogue_options(gpu, pool, monkeypatch):
    """The register-weights form with the epilogue options the backbone does not use: no bias, no ReLU, alpha != 1, output
    rows inside a wider buffer, residual rows with another pitch; an fp32 output goes to round 2's form (still
    bit-identical to the one-tile kernel)."""
    from wsovod_amd.layers import hip_ops

    torch.manual_seed(21)
    n, H, W = 6, 122, 130
    x = torch.randn(n, H, W, 64, device=gpu).to(torch.bfloat16)
    w = (torch.randn(64, 9 * 64, device=gpu) * 0.05).to(torch.bfloat16)
    geom = dict(n_img=n, H=H, W=W, Cin=64, Ho=H, Wo=W, KH=3, KW=3, stride=1, pad=1, dil=1, pool=pool)
    rows = n * (H // 2) * (W // 2) if pool else n * H * W
    res_wide = torch.randn(n * H * W, 96, device=gpu).to(torch.bfloat16)
    for kw in (dict(), dict(alpha=0.5), dict(residual=res_wide[:, 16:80])):
        outs = []
        for persist in ("0", "1"):
            monkeypatch.setenv("WSOVOD_C64_PERSIST", persist)
            buf = torch.full((rows, 128), 7.0, device=gpu, dtype=torch.bfloat16)
            hip_ops.gemm_nt(x, w, conv=geom, relu=False, out=buf[:, 32:96], **kw)
            outs.append(buf)
        assert torch.equal(outs[0], outs[1])
        assert bool((outs[1][:, :32] == 7).all()) and bool((outs[1][:, 96:] == 7).all())  # nothing outside the view
        assert float(outs[1][:, 32:96].float().min()) < 0  # no ReLU
    if not pool:
        monkeypatch.setenv("WSOVOD_C64_PERSIST", "0")
        a = hip_ops.gemm_nt(x, w, conv=geom, out_dtype=torch.float32)
        monkeypatch.setenv("WSOVOD_C64_PERSIST", "1")
        assert torch.equal(a, hip_ops.gemm_nt(x, w, conv=geom, out_dtype=torch.float32))


def test_gemm_tn_reduction_longer_than_one_buffer_resource(gpu, monkeypatch):
    """Operands past the 2 GiB a buffer resource addresses (96 images x 512 proposals x 25088 features) are reduced in
    row blocks that accumulate: same result as the single launch up to fp32 summation order, `accumulate` and `alpha`
    respected on the first block only / on every block."""
    from wsovod_amd.layers import hip_ops

    torch.manual_seed(9)
    P = torch.randn(1000, 64, device=gpu).to(torch.bfloat16)
    Q = torch.randn(1000, 136, device=gpu).to(torch.bfloat16)
    base = torch.randn(64, 136, device=gpu)
    want = hip_ops.gemm_tn(P, Q, out=base.clone(), alpha=0.5, accumulate=True, split_tail=False)
    monkeypatch.setattr(hip_ops, "GEMM_TN_MAX_OPERAND_BYTES", 200 * 136 * 2)  # 200 rows of Q -> blocks of 192 rows
    got = hip_ops.gemm_tn(P, Q, out=base.clone(), alpha=0.5, accumulate=True, split_tail=False)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-4)
    ref = base.double() + 0.5 * (P.double().t() @ Q.double())
    torch.testing.assert_close(got.double(), ref, rtol=1e-5, atol=1e-4)
    fresh = hip_ops.gemm_tn(P, Q, alpha=1.0, split_tail=False)
    torch.testing.assert_close(fresh.double(), P.double().t() @ Q.double(), rtol=1e-5, atol=1e-4)


def test_gemm_tn_tail_split_matches_unsplit(gpu):
    """More than one round of 256x256 tiles with a small last round: the tail tiles are reduced in K slices that meet
    by atomic adds.  Same result as the unsplit launch up to fp32 summation order; accumulate keeps the old contents;
    tiles of the full rounds are bit-identical."""
    from wsovod_amd.layers import hip_ops

    torch.manual_seed(5)
    Mred, NI, NJ = 1024, 2048 + 256, 256 * 33  # 9 x 33 = 297 tiles: one full round + 41 tail tiles
    P = (torch.rand(Mred, NI, device=gpu) - 0.5).to(torch.bfloat16)
    Q = (torch.rand(Mred, NJ, device=gpu) - 0.5).to(torch.bfloat16)
    fixed = hip_ops.gemm_tn(P, Q, split_tail=False)
    assert torch.equal(fixed, hip_ops.gemm_tn(P, Q, split_tail=False))
    split = hip_ops.gemm_tn(P, Q)
    torch.testing.assert_close(split, fixed, rtol=0, atol=1e-4)  # |terms| <= 0.25, 1024 of them
    same = (split == fixed).view(9, 256, 33, 256).all(dim=3).all(dim=1)  # per tile
    assert int(same.sum()) >= 256, int(same.sum())  # every tile of the full round
    ref = (P.double().t() @ Q.double()).float()
    torch.testing.assert_close(split, ref, rtol=1e-4, atol=1e-3)
    acc = torch.randn(NI, NJ, device=gpu)
    out = hip_ops.gemm_tn(P, Q, out=acc.clone(), alpha=0.5, accumulate=True)
    torch.testing.assert_close(out, acc + 0.5 * ref, rtol=1e-4, atol=1e-3)


def test_gemm_split_k_matches_unsplit_tile(gpu):
    """Few output tiles and a long K: the dispatcher takes the 256x256 8-phase tile with split-K (slices of K into a
    workspace, then a finalize kernel with the whole epilogue).  Against the same tile named explicitly (never split):
    equal up to fp32 summation order, dropout mask identical, accumulate / transposed copy / residual honoured."""
    from wsovod_amd.layers import hip_ops

    torch.manual_seed(3)
    M, N, K = 520, 1032, 8192 + 64  # ragged edges, 3 x 5 tiles
    A = (torch.rand(M, K, device=gpu) - 0.5).to(torch.bfloat16)
    B = (torch.rand(N, K, device=gpu) - 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=gpu)
    res = torch.randn(M, N, device=gpu).to(torch.bfloat16)
    kw = dict(bias=bias, relu=True, residual=res, alpha=0.5, dropout_p=0.5, dropout_seed=77)
    want = hip_ops.gemm_nt(A, B, out_dtype=torch.float32, tile_hint=8256256, **kw)
    got = hip_ops.gemm_nt(A, B, out_dtype=torch.float32, **kw)
    assert torch.equal(got == 0, want == 0)  # same ReLU / dropout pattern
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-3)
    acc = torch.randn(M, N, device=gpu)
    ct = torch.zeros(N, M + 8, device=gpu)
    o1 = hip_ops.gemm_nt(A, B, out=acc.clone(), accumulate=True, out_t=ct)
    torch.testing.assert_close(o1, acc + A.float() @ B.float().t(), rtol=1e-4, atol=2e-2)
    assert torch.equal(ct[:, :M], o1.t())  # the transposed copy carries the stored value (after the accumulate)
    b16 = hip_ops.gemm_nt(A, B, bias=bias, relu=True, out_dtype=torch.bfloat16)
    torch.testing.assert_close(b16.float(), torch.relu(A.float() @ B.float().t() + bias), rtol=2e-2, atol=0.2)


_WS_SCRIPT = r"""
import sys, torch
sys.path.insert(0, {root!r})
from wsovod_amd.layers import hip_ops
gpu = torch.device("cuda", 0)
torch.manual_seed(5)
# 2 x 2 output tiles of 256 x 256 and a long K => split-K slices; a FRESH process: the workspace starts empty
A1 = (torch.rand(300, 16384 + 128, device=gpu) - 0.5).to(torch.bfloat16)
B1 = (torch.rand(260, 16384 + 128, device=gpu) - 0.5).to(torch.bfloat16)
want = hip_ops.gemm_nt(A1, B1, out_dtype=torch.float32)  # eager: the workspace now fits this shape (2.5 MB)
out = torch.empty_like(want)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    hip_ops.gemm_nt(A1, B1, out=out)
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    hip_ops.gemm_nt(A1, B1, out=out)
out.zero_(); g.replay()
assert torch.equal(out, want), "replay != eager"
# a larger split-K shape (128 tiles x 2 slices = 67 MB of partial sums): growing UNDER capture is refused ...
A2 = (torch.rand(4096, 8192, device=gpu) - 0.5).to(torch.bfloat16)
B2 = (torch.rand(2048, 8192, device=gpu) - 0.5).to(torch.bfloat16)
out2 = torch.empty(4096, 2048, device=gpu)
g2 = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g2, capture_error_mode="thread_local"):
        hip_ops.gemm_nt(A2, B2, out=out2)
    raise SystemExit("growing the workspace under capture was not refused")
except RuntimeError as e:
    assert "split-K workspace" in str(e), str(e)
torch.cuda.synchronize()
# ... eagerly it grows: the block the first graph points to must stay allocated (retired, never freed)
big = hip_ops.gemm_nt(A2, B2, out=out2)
torch.testing.assert_close(big, A2.float() @ B2.float().t(), rtol=1e-4, atol=5e-2)
junk = [torch.full((1 << 20,), float("nan"), device=gpu) for _ in range(16)]  # would land in a freed block
out.zero_(); g.replay(); torch.cuda.synchronize()
assert torch.equal(out, want), "replay after the growth != eager"
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3):  # the grown workspace serves the capture now
    hip_ops.gemm_nt(A2, B2, out=out2)
out2.zero_(); g3.replay(); g.replay(); torch.cuda.synchronize()
assert torch.equal(out2, big) and torch.equal(out, want)
print("WS_OK")
"""


def test_split_k_workspace_outlives_a_captured_graph(gpu, second_gpu_process):
    """the split-K workspace grows with the shape.  A HIP graph captured on a SMALL split-K launch
    keeps the workspace pointer it was captured with; a later, larger split-K launch must retire that block (keep it
    allocated), never free it, and growing UNDER capture must be refused instead of calling hipMalloc inside the capture.
    Runs in a fresh process (the workspace is process-wide and other tests may already have grown it to its maximum)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _WS_SCRIPT.format(root=root)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "WS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("k,dil", [(3, 2), (1, 1)])
def test_conv_with_fused_projection_shortcut(gpu, dtype, tol, k, dil):
    """wsovod_gemm_desc.A2: out = relu(conv(h) + conv1x1(x) + biases) as ONE contraction (K = k*k*C + Cin2) against the
    two-launch form (shortcut conv, then conv with the residual epilogue) and against torch's conv2d in fp32."""
    import torch.nn.functional as F
    from wsovod_amd.layers import hip_ops as H

    g = torch.Generator().manual_seed(11)
    n, Hh, Ww, C, Cin2 = 2, 19, 27, 128, 64
    h = torch.randn(n, Hh, Ww, C, generator=g)
    x = torch.randn(n, Hh, Ww, Cin2, generator=g)
    w = torch.randn(C, k, k, C, generator=g) * 0.05
    wsc = torch.randn(C, Cin2, generator=g) * 0.1
    b, bsc = torch.randn(C, generator=g), torch.randn(C, generator=g)
    pad = dil * (k // 2)
    want = F.relu(F.conv2d(h.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), b, padding=pad, dilation=dil)
                  + F.conv2d(x.permute(0, 3, 1, 2), wsc.view(C, Cin2, 1, 1), bsc)).permute(0, 2, 3, 1)
    hd, xd = h.to(gpu).to(dtype).contiguous(), x.to(gpu).to(dtype).contiguous()
    wcat = torch.cat([w.reshape(C, -1), wsc], dim=1).to(gpu).to(dtype).contiguous()
    geom = dict(n_img=n, H=Hh, W=Ww, Cin=C, Ho=Hh, Wo=Ww, KH=k, KW=k, stride=1, pad=pad, dil=dil)
    fused = H.gemm_nt(hd, wcat, conv=geom, bias=(b + bsc).to(gpu), relu=True, out_dtype=torch.float32, A2=xd)
    geom1 = dict(n_img=n, H=Hh, W=Ww, Cin=Cin2, Ho=Hh, Wo=Ww, KH=1, KW=1, stride=1, pad=0, dil=1)
    sc = H.gemm_nt(xd, wsc.to(gpu).to(dtype).contiguous(), conv=geom1, bias=bsc.to(gpu), out_dtype=torch.float32)
    two = H.gemm_nt(hd, w.reshape(C, -1).to(gpu).to(dtype).contiguous(), conv=geom, bias=b.to(gpu), relu=True,
                    residual=sc, out_dtype=torch.float32)
    scale = float(want.abs().max())
    assert float((fused.view(n, Hh, Ww, C).cpu() - want).abs().max()) <= tol * scale
    assert float((fused - two).abs().max()) <= (1e-5 if dtype == torch.float32 else 1e-3) * scale  # same products, other order
    for hint in (256128, 128128, 64064):  # the other tile shapes of the same kernel
        alt = H.gemm_nt(hd, wcat, conv=geom, bias=(b + bsc).to(gpu), relu=True, out_dtype=torch.float32, A2=xd, tile_hint=hint)
        assert float((alt - fused).abs().max()) <= 1e-5 * scale + (0 if dtype == torch.float32 else 1e-3 * scale)
    # the 8-wavefront tile (four- and two-phase form) carries the fused shortcut too (round 3): same products, same order
    b16 = lambda t: t.to(torch.bfloat16)
    ref16 = H.gemm_nt(b16(hd), b16(wcat), conv=geom, A2=b16(xd), bias=(b + bsc).to(gpu), relu=True, tile_hint=256256,
                      out_dtype=torch.float32)
    for hint in (8256256, 2256256):
        alt = H.gemm_nt(b16(hd), b16(wcat), conv=geom, A2=b16(xd), bias=(b + bsc).to(gpu), relu=True, tile_hint=hint,
                        out_dtype=torch.float32)
        assert torch.equal(alt, ref16), hint


@pytest.mark.parametrize("mode", ["bf16", "x2", "x2_shortcut"])
def test_conv_tile_round_tail_launch_is_bit_identical(gpu, mode, monkeypatch):
    """A conv whose 256x256 tile grid ends in a partly filled round of 256 tiles sends the rows behind the last full round
    through a second launch with 256x128 tiles (gemm.hip: `m_base`): same bits as the one-launch form (WSOVOD_CONV_TAIL=0)
    and as the 256x128 tile everywhere; residual / fused shortcut rows are addressed absolutely."""
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(5)
    n, Hh, Ww, Cin, Cout = 5, 75, 100, 64, 512  # 147 x 2 tiles = 1 full round + 38 tiles
    x = torch.randn(n * Hh * Ww, Cin, device=gpu)
    w = torch.randn(Cout, 9 * Cin, device=gpu) * 0.05
    b = torch.randn(Cout, device=gpu)
    res = torch.randn(n * Hh * Ww, Cout, device=gpu)
    geom = dict(n_img=n, H=Hh, W=Ww, Cin=Cin, Ho=Hh, Wo=Ww, KH=3, KW=3, stride=1, pad=2, dil=2)
    if mode == "bf16":
        xa, wa, kw = x.to(torch.bfloat16).view(n, Hh, Ww, Cin), w.to(torch.bfloat16), dict(out_dtype=torch.bfloat16, residual=res.to(torch.bfloat16))
    elif mode == "x2":
        xa, wa = H.x2_encode(x).view(n, Hh, Ww, Cin), H.x2_encode(w)
        kw = dict(x2=True, out_dtype=H.X2, residual=H.x2_encode(res), residual_x2=True)
    else:  # fused 1x1 projection shortcut: second input at the output pixel
        x2in = torch.randn(n * Hh * Ww, 64, device=gpu)
        w = torch.cat([w, torch.randn(Cout, 64, device=gpu) * 0.05], 1)
        xa, wa = H.x2_encode(x).view(n, Hh, Ww, Cin), H.x2_encode(w)
        kw = dict(x2=True, out_dtype=H.X2, A2=H.x2_encode(x2in).view(n, Hh, Ww, 64))

    def run(tile=0):
        return H.gemm_nt(xa, wa, conv=geom, bias=b, relu=True, tile_hint=tile, **kw)

    monkeypatch.setenv("WSOVOD_CONV_TAIL", "0")
    whole = run()
    monkeypatch.delenv("WSOVOD_CONV_TAIL")
    split = run()
    torch.cuda.synchronize()
    assert torch.equal(whole.view(torch.int32) if whole.dtype == torch.float32 else whole.view(torch.int16),
                       split.view(torch.int32) if split.dtype == torch.float32 else split.view(torch.int16))
    narrow = run(256128)
    assert torch.equal(narrow.view(torch.int16) if narrow.dtype != torch.float32 else narrow.view(torch.int32),
                       split.view(torch.int16) if split.dtype != torch.float32 else split.view(torch.int32))


@pytest.mark.parametrize("shortcut", [False, True])
def test_conv_tile_round_tail_as_split_k_slices(gpu, shortcut, monkeypatch):
    """Round 5: a tile-round tail of few tiles and a long K (res5 at 32 images: 84 tiles of 72 K-steps) runs as split-K slices
    of the lean 8-wavefront tile + the finalize pass (gemm8.hip: `m_base`, workspace rows counted from it).  The rows of the
    full rounds keep their bits; the tail rows equal the unsplit form up to the summation order; residual / fused-shortcut
    rows are addressed absolutely."""
    from wsovod_amd.layers import hip_ops as H

    torch.manual_seed(9)
    n, Hh, Ww, Cin, Cout = 5, 75, 100, 256, 512  # 147 x 2 tiles = 1 full round + 38 tiles; K = 2304 (+ 64): two slices
    x = torch.randn(n * Hh * Ww, Cin, device=gpu) * 0.5
    w = torch.randn(Cout, 9 * Cin, device=gpu) * 0.03
    b = torch.randn(Cout, device=gpu)
    geom = dict(n_img=n, H=Hh, W=Ww, Cin=Cin, Ho=Hh, Wo=Ww, KH=3, KW=3, stride=1, pad=2, dil=2)
    kw = dict(x2=True, out_dtype=H.X2, bias=b, relu=True)
    if shortcut:
        x2in = torch.randn(n * Hh * Ww, 64, device=gpu)
        w = torch.cat([w, torch.randn(Cout, 64, device=gpu) * 0.05], 1)
        kw["A2"] = H.x2_encode(x2in).view(n, Hh, Ww, 64)
    else:
        kw.update(residual=H.x2_encode(torch.randn(n * Hh * Ww, Cout, device=gpu)), residual_x2=True)
    xa, wa = H.x2_encode(x).view(n, Hh, Ww, Cin), H.x2_encode(w)
    monkeypatch.setenv("WSOVOD_CONV_TAIL_SPLITK", "0")
    plain = H.x2_decode(H.gemm_nt(xa, wa, conv=geom, **kw))
    monkeypatch.delenv("WSOVOD_CONV_TAIL_SPLITK")
    split = H.x2_decode(H.gemm_nt(xa, wa, conv=geom, **kw))
    torch.cuda.synchronize()
    main_rows = 128 * 256
    assert torch.equal(plain[:main_rows], split[:main_rows])
    assert not torch.equal(plain[main_rows:], split[main_rows:])  # (the tail did take the split form)
    scale = float(plain.abs().max())
    assert float((plain[main_rows:] - split[main_rows:]).abs().max()) < 1e-5 * scale
