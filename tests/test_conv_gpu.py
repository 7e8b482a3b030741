"""GPU: implicit-GEMM conv kernels (fwd / dgrad / wgrad, all gather modes, prologues, epilogues)
against plain PyTorch CPU fp32 references of the same op (tolerance: fp32 accumulation order,
rtol 1e-4 of the output scale)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def close(a, b, tol=2e-4):
    a, b = a.double().cpu(), b.double().cpu()
    scale = b.abs().max().item() + 1e-12
    err = (a - b).abs().max().item()
    assert err <= tol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


def ref_pro(x, mode, scale, shift):
    if mode in (2, 4):
        x = x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if mode in (1, 2):
        x = F.relu(x)
    if mode == 3:
        x = F.leaky_relu(x, 0.2)
    return x


CASES = [
    # kind, B, H, W, Ci, Co, R, stride, pad
    ("conv", 4, 16, 16, 32, 64, 3, 1, 1),
    ("conv", 2, 8, 8, 128, 128, 3, 1, 1),
    ("conv", 3, 5, 7, 4, 64, 3, 1, 1),          # ragged M, padded-RGB input
    ("conv", 2, 16, 16, 64, 48, 3, 1, 1),        # Co not a tile multiple
    ("conv", 2, 8, 8, 64, 128, 1, 1, 0),         # 1x1 shortcut conv
    ("conv", 2, 16, 16, 16, 32, 3, 2, 1),        # DCGAN D strided conv
    ("conv", 2, 16, 16, 48, 20, 3, 1, 1),        # K not a multiple of 32
    ("convT", 2, 8, 8, 96, 48, 4, 2, 1),         # DCGAN G
    ("convT", 4, 1, 1, 384, 192, 4, 1, 0),       # DCGAN G first layer
    ("conv", 64, 4, 4, 256, 256, 3, 1, 1),
    ("conv", 2, 16, 16, 64, 32, 3, 2, 1),        # strided conv on the uniform-tap loader path (Ci % 32 == 0)
    ("conv", 3, 8, 8, 32, 64, 4, 2, 1),          # 4x4 / stride 2 (DCGAN D shape), uniform-tap path, odd batch
]


def make(kind, B, H, W, Ci, Co, R, stride, pad, seed=0):
    from diagan.ops import conv as C
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Ci, H, W, generator=g)
    if kind == "conv":
        w = torch.randn(Co, Ci, R, R, generator=g) / (Ci * R * R) ** 0.5
    else:
        w = torch.randn(Ci, Co, R, R, generator=g) / (Ci * R * R) ** 0.5
    geom = C.Geom(kind, Ci, Co, R, R, stride, pad)
    wp = (C.pack_oihw(w, geom.Kp) if kind == "conv" else C.pack_iohw(w, geom.Kp)).cuda()
    return geom, x, w, wp


def ref_fwd(kind, x, w, bias, stride, pad):
    if kind == "conv":
        return F.conv2d(x, w, bias, stride=stride, padding=pad)
    return F.conv_transpose2d(x, w, bias, stride=stride, padding=pad)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("pro", [0, 1, 2, 3])
def test_fwd(case, pro):
    from diagan.ops import conv as C
    kind, B, H, W, Ci, Co, R, stride, pad = case
    geom, x, w, wp = make(*case)
    g = torch.Generator().manual_seed(1)
    bias = torch.randn(Co, generator=g)
    scale = torch.rand(Ci, generator=g) + 0.5
    shift = torch.randn(Ci, generator=g) * 0.3
    ref = ref_fwd(kind, ref_pro(x, pro, scale, shift), w, bias, stride, pad)
    res = torch.randn(ref.shape, generator=g)
    y = C.conv_fwd(geom, nhwc(x).cuda(), wp, bias=bias.cuda(), residual=nhwc(res).cuda(),
                   pro=(pro, scale.cuda(), shift.cuda()))
    close(nchw(y), ref + res)


@pytest.mark.parametrize("case", [
    ("conv", 2, 8, 8, 128, 128, 3, 1, 1),        # 36 K-steps: 18 + 18 (the D-32 8x8 blocks' shape)
    ("conv", 2, 8, 8, 40, 72, 3, 1, 1),          # 12 K-steps of the general loader path (Ci % 32 != 0), Co no tile multiple
    ("conv", 3, 5, 7, 96, 64, 3, 1, 1),          # 27 K-steps: 14 + 13, ragged M
    ("conv", 2, 4, 4, 32, 64, 1, 1, 0),          # ONE K-step: the second group has nothing to do
    ("conv", 2, 8, 8, 64, 64, 1, 1, 0),          # two K-steps: one each
    ("convT", 2, 8, 8, 96, 48, 4, 2, 1),         # transposed gather
    ("conv", 2, 16, 16, 64, 32, 3, 2, 1),        # strided
])
@pytest.mark.parametrize("pro", [0, 2])
@pytest.mark.parametrize("cfg", [14])
def test_two_k_groups_per_workgroup(case, pro, cfg):
    """tile_cfg 14: two K-groups of four waves per workgroup, joined through LDS -- every K-split (even, odd,
    groups with nothing to do), both loader paths, prologue, bias + residual epilogue; and the same launch against tile_cfg 7 bit for bit where
    the split leaves the summation order of a K-group's steps unchanged only up to the final add (so: tolerance, not equality)."""
    from diagan.ops import conv as C
    kind, B, H, W, Ci, Co, R, stride, pad = case
    geom, x, w, wp = make(*case)
    g = torch.Generator().manual_seed(1)
    scale, shift = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.1
    bias = torch.randn(Co, generator=g)
    ref = ref_fwd(kind, ref_pro(x, pro, scale, shift), w, bias, stride, pad)
    res = torch.randn(ref.shape, generator=g)
    y = C.conv_fwd(geom, nhwc(x).cuda(), wp, bias=bias.cuda(), residual=nhwc(res).cuda(),
                   pro=(pro, scale.cuda(), shift.cuda()), tile_cfg=cfg)
    close(nchw(y), ref + res)
    y7 = C.conv_fwd(geom, nhwc(x).cuda(), wp, bias=bias.cuda(), residual=nhwc(res).cuda(),
                    pro=(pro, scale.cuda(), shift.cuda()), tile_cfg=7)
    close(nchw(y), nchw(y7), tol=2e-6)


def test_two_k_groups_chosen_for_lone_tiles():
    """The automatic choice takes tile_cfg 14 for at most one 64x64 tile per CU with 16 <= K-steps < 64 (the 8x8 blocks of
    SNGAN-32's discriminator at batch 128: M = 8192, N = 128, K = 1152), not for short or long K loops or larger launches."""
    from diagan import _native as nat
    from diagan.ops import conv as C  # noqa: F401
    pick = nat.fn("diagan_conv_gemm_pick_cfg")
    assert pick(8192, 128, 1152, 1) == 14
    assert pick(8192, 128, 256, 1) == 7            # short K loop
    assert pick(32768, 128, 1152, 1) != 14         # 1024 tiles
    assert pick(2048, 512, 4608, 1) != 14 or nat.fn("diagan_conv_gemm_pick_ksplit")(2048, 512, 4608, 7) == 1


@pytest.mark.parametrize("tile_cfg", [1, 3, 5, 7, 8, 14])
def test_fwd_all_tile_configs(tile_cfg):
    from diagan.ops import conv as C
    case = ("conv", 3, 9, 11, 64, 96, 3, 1, 1)
    geom, x, w, wp = make(*case)
    y = C.conv_fwd(geom, nhwc(x).cuda(), wp, tile_cfg=tile_cfg)
    close(nchw(y), ref_fwd("conv", x, w, None, 1, 1))


@pytest.mark.parametrize("case", CASES)
def test_dgrad_and_wgrad(case):
    from diagan.ops import conv as C
    kind, B, H, W, Ci, Co, R, stride, pad = case
    geom, x, w, wp = make(*case)
    g = torch.Generator().manual_seed(2)
    scale = torch.rand(Ci, generator=g) + 0.5
    shift = torch.randn(Ci, generator=g) * 0.3
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    a = ref_pro(xr, 2, scale, shift)
    a.retain_grad()
    y = ref_fwd(kind, a, wr, None, stride, pad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    # dgrad: gradient w.r.t. the conv input a (after prologue)
    wd = torch.zeros((Ci, geom.Kd), device="cuda")
    C.pack_weights(wp, Co, Ci, R * R, geom.Kp, geom.Kd, Wd=wd)
    dyd = nhwc(dy).cuda()
    da = C.conv_dgrad(geom, dyd, wd, (H, W))
    close(nchw(da), a.grad)
    # with the ReLU-backward mask + residual epilogue
    msrc = torch.randn(B, Ci, H, W, generator=g)
    res = torch.randn(B, Ci, H, W, generator=g)
    da2 = C.conv_dgrad(geom, dyd, wd, (H, W), residual=nhwc(res).cuda(), mask_src=nhwc(msrc).cuda())
    close(nchw(da2), (a.grad + res) * (msrc > 0))
    # wgrad (prologue recomputed from raw x), accumulate on top of an existing gradient
    grad = torch.full((Co, geom.Kp), 0.5, device="cuda")
    C.conv_wgrad(geom, dyd, nhwc(x).cuda(), grad, accumulate=True, pro=(2, scale.cuda(), shift.cuda()))
    gw = (C.unpack_oihw(grad, Co, Ci, R, R) if kind == "conv" else C.unpack_iohw(grad, Ci, Co, R, R)) - 0.5
    close(gw, wr.grad, tol=5e-4)
    if geom.Kp > R * R * Ci:   # padded columns receive exactly zero gradient
        assert torch.all(grad[:, R * R * Ci:] == 0.5)


def test_random_geometries():
    """Seeded fuzz over geometry (odd sizes, channel counts that are not tile multiples, 1x1 / 3x3 / 4x4 filters,
    stride 1 and 2, conv and transposed conv), prologue mode and epilogue options: forward, data gradient and
    weight gradient against torch CPU autograd.  Catches gather / padding / tile-edge mistakes the fixed cases miss."""
    from diagan.ops import conv as C
    rng = torch.Generator().manual_seed(20240)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=rng))
    for trial in range(36):
        kind = "convT" if trial % 4 == 3 else "conv"
        R = (1, 3, 3, 4)[ri(0, 3)] if kind == "conv" else 4
        stride = ri(1, 2) if R > 1 else 1
        pad = {1: 0, 3: 1, 4: 1}[R] if not (kind == "convT" and stride == 1) else 0
        B, H, W = ri(1, 5), ri(R, 19), ri(R, 21)
        if kind == "conv" and stride == 2:
            H, W = H + (H + 2 * pad - R) % 2 * 0, W          # any size: output is floor((H + 2p - R)/2) + 1
        Ci, Co = 4 * ri(1, 24), 4 * ri(1, 40)
        pro = ri(0, 4)
        case = (kind, B, H, W, Ci, Co, R, stride, pad)
        geom, x, w, wp = make(*case, seed=100 + trial)
        scale = torch.rand(Ci, generator=rng) + 0.5
        shift = torch.randn(Ci, generator=rng) * 0.3
        bias = torch.randn(Co, generator=rng)
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        a = ref_pro(xr, pro, scale, shift)
        a.retain_grad()
        y = ref_fwd(kind, a, wr, bias, stride, pad)
        res = torch.randn(y.shape, generator=rng)
        dy = torch.randn(y.shape, generator=rng)
        y.backward(dy)
        tag = f"trial {trial}: {case} pro={pro}"
        pro_t = (pro, scale.cuda(), shift.cuda())
        try:
            got = C.conv_fwd(geom, nhwc(x).cuda(), wp, bias=bias.cuda(), residual=nhwc(res).cuda(), pro=pro_t)
            close(nchw(got), y.detach() + res)
            wd = torch.zeros((Ci, geom.Kd), device="cuda")
            C.pack_weights(wp, Co, Ci, R * R, geom.Kp, geom.Kd, Wd=wd)
            da = C.conv_dgrad(geom, nhwc(dy).cuda(), wd, (H, W))
            close(nchw(da), a.grad)
            grad = torch.zeros((Co, geom.Kp), device="cuda")
            C.conv_wgrad(geom, nhwc(dy).cuda(), nhwc(x).cuda(), grad, accumulate=False, pro=pro_t)
            gw = C.unpack_oihw(grad, Co, Ci, R, R) if kind == "conv" else C.unpack_iohw(grad, Ci, Co, R, R)
            close(gw, wr.grad, tol=5e-4)
        except AssertionError as e:
            raise AssertionError(f"{tag}: {e}") from None


@pytest.mark.parametrize("Ci,H,W,B", [(256, 32, 32, 4), (128, 16, 16, 3), (64, 12, 20, 2), (64, 64, 64, 2)])
@pytest.mark.parametrize("pro", [0, 2])
def test_small_co_kernel_fwd_and_dgrad(Ci, H, W, B, pro):
    """conv3x3_co4 (4 output channels, lane-parallel): the generator's last conv and D's first-layer dgrad."""
    from diagan.ops import conv as C
    g = torch.Generator().manual_seed(Ci + pro)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(3, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5
    bias = torch.randn(3, generator=g)
    scale, shift = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.3
    geom = C.Geom("conv", Ci, 4, 3, 3, 1, 1)
    wp = torch.zeros(4, geom.Kp)
    wp[:3] = C.pack_oihw(w, geom.Kp)
    b4 = torch.zeros(4)
    b4[:3] = bias
    ref = F.conv2d(ref_pro(x, pro, scale, shift), w, bias, padding=1)
    res = torch.randn(B, 4, H, W, generator=g)
    y = C.conv_fwd(geom, nhwc(x).cuda(), wp.cuda(), bias=b4.cuda(), residual=nhwc(res).cuda(),
                   pro=(pro, scale.cuda(), shift.cuda()))
    close(nchw(y)[:, :3], ref + res[:, :3])
    assert torch.equal(nchw(y)[:, 3].cpu(), res[:, 3])            # padded channel: zero weights, zero bias
    if pro == 0:
        # data-gradient of a conv FROM 4 (RGB+pad) channels TO Ci: output has 4 channels
        geom2 = C.Geom("conv", 4, Ci, 3, 3, 1, 1)
        w2 = torch.randn(Ci, 3, 3, 3, generator=g) / 27 ** 0.5
        xi = torch.randn(B, 3, H, W, generator=g, requires_grad=True)
        yo = F.conv2d(xi, w2, None, padding=1)
        dy = torch.randn(yo.shape, generator=g)
        yo.backward(dy)
        wp2 = C.pack_oihw(w2, geom2.Kp, ci_pad=4).cuda()
        wd = torch.zeros((4, geom2.Kd), device="cuda")
        C.pack_weights(wp2, Ci, 4, 9, geom2.Kp, geom2.Kd, Wd=wd)
        dx = C.conv_dgrad(geom2, nhwc(dy).cuda(), wd, (H, W))
        close(nchw(dx)[:, :3], xi.grad)


@pytest.mark.parametrize("B,H,W,Co", [(3, 5, 7, 64), (1, 4, 4, 128), (6, 32, 32, 128), (4, 64, 64, 64), (2, 9, 33, 128)])
def test_first_layer_kernel_from_four_channels(B, H, W, Co):
    """conv3x3_ci4 (round 5): the discriminators' first convolution (RGB + pad -> 64 / 128 channels) on its own kernel -- weights
    in registers, 32 pixels x all channels per wave, no LDS.  Against float64 F.conv2d (ragged pixel counts, one image, borders),
    with the bias, without it, with the two spectral-norm scales of a paired pass; and against the implicit GEMM (tile_cfg 7), which
    adds the same products in the same order (the last bit of scale * sum + bias may differ: fused or not)."""
    from diagan.ops import conv as C
    g = torch.Generator().manual_seed(B * 1000 + H + Co)
    x = torch.randn(B, 4, H, W, generator=g)
    w = torch.randn(Co, 4, 3, 3, generator=g) / 6.0
    bias = torch.randn(Co, generator=g)
    geom = C.Geom("conv", 4, Co, 3, 3, 1, 1)
    wp = C.pack_oihw(w, geom.Kp).cuda()
    xc = nhwc(x).cuda()
    assert C.nat.fn("diagan_conv3x3_ci4_supported")(4, Co, 3, 3, 1, 1, -1, 1) == 1
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    y = C.conv_fwd(geom, xc, wp, bias=bias.cuda())
    close(nchw(y), ref + bias.double().view(1, -1, 1, 1), tol=2e-6)
    close(y, C.conv_fwd(geom, xc, wp, bias=bias.cuda(), tile_cfg=7), tol=1e-6)
    close(nchw(C.conv_fwd(geom, xc, wp)), ref, tol=2e-6)
    close(nchw(C.conv_fwd(geom, xc, wp, bias=bias.cuda(), out_scale=0.37)), 0.37 * ref + bias.double().view(1, -1, 1, 1), tol=2e-6)
    if B % 2 == 0:
        s0, s1 = torch.tensor([0.7]).cuda(), torch.tensor([1.9]).cuda()
        y2 = C.conv_fwd(geom, xc, wp, bias=bias.cuda(), row_scale=(s0, s1))
        r2 = ref.clone()
        r2[:B // 2] *= 0.7
        r2[B // 2:] *= 1.9
        close(nchw(y2), r2 + bias.double().view(1, -1, 1, 1), tol=2e-6)
        close(y2, C.conv_fwd(geom, xc, wp, bias=bias.cuda(), row_scale=(s0, s1), tile_cfg=7), tol=1e-6)


@pytest.mark.parametrize("case", [(128, 8, 8, 128, 128), (64, 8, 8, 128, 128), (64, 4, 4, 256, 256), (3, 6, 10, 64, 24), (2, 5, 7, 64, 200)])
def test_split_operand_implicit_gemm_on_the_bf16_pipe(case):
    """conv_gemm_x3.hip (round 5, tile_cfg 16): the lone-tile 3x3 launches with every fp32 operand split exactly into three bf16
    pieces, six piece products accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  fp32-grade: against float64 F.conv2d /
    conv_transpose2d at the implicit GEMM's own tolerance, forward (ReLU prologue, bias, ReLU-ed residual, the two scales of a
    paired pass) and data gradient (residual + backward mask), ragged pixel counts and channel counts that fill no tile; and the
    automatic choice takes it exactly when the switch is on."""
    from diagan.ops import conv as C
    B, H, W, Ci, Co = case
    g = torch.Generator().manual_seed(B + Ci + Co)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5
    bias, res = torch.randn(Co, generator=g), torch.randn(B, Co, H, W, generator=g)
    geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
    wp = C.pack_oihw(w, geom.Kp).cuda()
    xc, rc = nhwc(x).cuda(), nhwc(res).cuda()
    relu = (C.PRO_RELU, None, None)
    ref = F.conv2d(F.relu(x.double()), w.double(), bias.double(), padding=1) + F.relu(res.double())
    y = C.conv_fwd(geom, xc, wp, bias=bias.cuda(), residual=rc, res_relu=True, pro=relu, tile_cfg=16)
    close(nchw(y), ref, tol=2e-5)
    close(y, C.conv_fwd(geom, xc, wp, bias=bias.cuda(), residual=rc, res_relu=True, pro=relu, tile_cfg=7), tol=2e-5)
    close(nchw(C.conv_fwd(geom, xc, wp, tile_cfg=16)), F.conv2d(x.double(), w.double(), None, padding=1), tol=2e-5)
    if B % 2 == 0:
        s0, s1 = torch.tensor([0.7]).cuda(), torch.tensor([1.9]).cuda()
        r2 = F.conv2d(x.double(), w.double(), None, padding=1)
        r2[:B // 2] *= 0.7
        r2[B // 2:] *= 1.9
        close(nchw(C.conv_fwd(geom, xc, wp, bias=bias.cuda(), row_scale=(s0, s1), tile_cfg=16)), r2 + bias.double().view(1, -1, 1, 1), tol=2e-5)
    # data gradient: gathers from dy [B,H,W,Co] with the flipped taps of the packed data-gradient operand
    wd = torch.zeros(Ci, geom.Kd, device="cuda")
    C.pack_weights(wp, Co, Ci, 9, geom.Kp, geom.Kd, Wd=wd)
    gy = torch.randn(B, Co, H, W, generator=g)
    msk, r0 = torch.randn(B, Ci, H, W, generator=g), torch.randn(B, Ci, H, W, generator=g)
    if Co % 32 == 0 and (9 * Co // 32) % 2 == 0:
        dx = C.conv_dgrad(geom, nhwc(gy).cuda(), wd, (H, W), residual=nhwc(r0).cuda(), mask_src=nhwc(msk).cuda(), tile_cfg=16)
        refd = (F.conv_transpose2d(gy.double(), w.double(), padding=1) + r0.double()) * (msk.double() > 0)
        close(nchw(dx), refd, tol=2e-5)
    # the automatic choice: the lone-tile configuration (14) becomes this kernel with the switch on, and only then
    if (B, H, Ci, Co) == (128, 8, 128, 128):
        timer = C.KernelTimer()
        try:
            C.TIMER = timer
            for on in (True, False):
                C.set_gemm_x3(on)
                y2 = C.conv_fwd(geom, xc, wp, bias=bias.cuda(), residual=rc, res_relu=True, pro=relu)
                close(nchw(y2), ref, tol=2e-5)
        finally:
            C.TIMER = None
            C.set_gemm_x3(None)
        names = [r[0] for r in timer.records]
        assert names[0] == "conv_gemm_x3_kernel<1>" and names[1].startswith("conv_gemm_kernel<64,64"), names


@pytest.mark.parametrize("case", [(3, 37, 41, 64, 192, 3, 3, 2, 0), (2, 40, 40, 128, 64, 2, 2, 1, 1), (5, 30, 30, 32, 128, 1, 1, 1, 0),
                                  (2, 50, 34, 96, 320, 2, 1, 1, 1), (8, 32, 32, 256, 256, 1, 2, 1, 1)])
@pytest.mark.parametrize("form", [1, 2])
def test_split_operand_implicit_gemm_128_tiles(case, form):
    """conv_gemm_x3b.hip (round 6, tile_cfg 17): the large implicit GEMMs of StyleGAN2 (3x3 / stride 2, the 2x2 / 2x1 / 1x2 / 1x1 parity
    classes of the stride-2 transposed gathers, 1x1) on the bf16 pipe with exactly split operands, both forms (1: 128 x 128 tiles, two
    workgroups per CU; 2: 256 x 128 tiles, MFMA waves + loader waves).  fp32-grade: against float64 F.conv2d at the implicit GEMM's own
    tolerance, with prologue, bias and residual, ragged pixel / channel counts; the data gradient of a stride-1 layer; an output map
    (a parity class written straight into the interleaved tensor, border trimmed); repeated launches bit-identical (the loader waves'
    counted waits: a race would show here)."""
    from diagan import _native as nat
    from diagan.ops import conv as C
    B, H, W, Ci, Co, R, S, st, pd = case
    g = torch.Generator().manual_seed(B + Ci + Co + R)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, R, S, generator=g) / (R * S * Ci) ** 0.5
    geom = C.Geom("conv", Ci, Co, R, S, st, pd)
    Ho, Wo = geom.out_hw(H, W)
    bias, res = torch.randn(Co, generator=g), torch.randn(B, Co, Ho, Wo, generator=g)
    wp = C.pack_oihw(w, geom.Kp).cuda()
    xc, rc = nhwc(x).cuda(), nhwc(res).cuda()
    try:
        nat.call("diagan_conv_gemm_x3b_force_form", form)
        ref = F.conv2d(F.leaky_relu(x.double(), 0.2), w.double(), bias.double(), stride=st, padding=pd) + res.double()
        y = C.conv_fwd(geom, xc, wp, bias=bias.cuda(), residual=rc, pro=(C.PRO_LRELU, None, None), tile_cfg=17)
        close(nchw(y), ref, tol=2e-5)
        plain = C.conv_fwd(geom, xc, wp, tile_cfg=17)
        close(nchw(plain), F.conv2d(x.double(), w.double(), None, stride=st, padding=pd), tol=2e-5)
        close(plain, C.conv_fwd(geom, xc, wp, tile_cfg=1, wino=False), tol=2e-5)
        for _ in range(10):
            assert torch.equal(C.conv_fwd(geom, xc, wp, tile_cfg=17), plain)
        # an output map: rows [1, Ho - 1) of the launch's grid to the odd pixels of a larger tensor
        big = torch.full((B, 2 * Ho + 1, 2 * Wo + 1, Co), float("nan"), device="cuda")
        C.conv_fwd(geom, xc, wp, tile_cfg=17, out=big, out_map=(2, 1, 1, 1, Ho - 1, 0, Wo))
        assert torch.equal(big[:, 1:2 * (Ho - 2):2, 1:2 * Wo + 1:2], plain[:, 1:Ho - 1])
        keep = torch.zeros_like(big, dtype=torch.bool)
        keep[:, 1:2 * (Ho - 2):2, 1:2 * Wo + 1:2] = True
        assert bool(torch.isnan(big[~keep]).all()) and not bool(torch.isnan(big[keep]).any())
        if st == 1 and Co % 32 == 0:          # data gradient (dr = -1 gather from dy)
            wd = torch.zeros(Ci, geom.Kd, device="cuda")
            C.pack_weights(wp, Co, Ci, R * S, geom.Kp, geom.Kd, Wd=wd)
            gy = torch.randn(B, Co, Ho, Wo, generator=g)
            dx = C.conv_dgrad(geom, nhwc(gy).cuda(), wd, (H, W), tile_cfg=17)
            close(nchw(dx), F.conv_transpose2d(gy.double(), w.double(), stride=1, padding=pd), tol=2e-5)
    finally:
        nat.call("diagan_conv_gemm_x3b_force_form", 0)


def test_split_operand_256_tiles_repeatable_over_several_rounds_of_workgroups():
    """The loader waves of form 2 wait for their in-flight loads by COUNT (inline-assembly loads, `vmcnt(N)`); a first build let the
    compiler copy the still-in-flight registers in front of the wait and was wrong once in a few launches, only where the launch ran
    more than one round of workgroups (profiles/r06_x3b.md).  The shape it showed on, 40 launches, each bit-identical to form 1."""
    from diagan import _native as nat
    from diagan.ops import conv as C
    g = torch.Generator(device="cuda").manual_seed(3)
    geom = C.Geom("conv", 512, 512, 2, 2, 1, 1)
    x = torch.randn(32, 32, 32, 512, device="cuda", generator=g)
    wp = torch.randn(512, geom.Kp, device="cuda", generator=g) * (4 * 512) ** -0.5
    try:
        nat.call("diagan_conv_gemm_x3b_force_form", 1)
        ref = C.conv_fwd(geom, x, wp, tile_cfg=17)
        nat.call("diagan_conv_gemm_x3b_force_form", 2)
        for _ in range(40):
            assert torch.equal(C.conv_fwd(geom, x, wp, tile_cfg=17), ref)
    finally:
        nat.call("diagan_conv_gemm_x3b_force_form", 0)


def test_split_operand_128_tiles_automatic_choice():
    """the automatic choice upgrades a large implicit-GEMM pick to tile_cfg 17 exactly when the switch is on, and reports it
    (diagan_conv_gemm_final_cfg: what the kernel timer names and what out_map_ok answers)"""
    from diagan.ops import conv as C
    geom = C.Geom("conv", 128, 256, 3, 3, 2, 0)
    x = torch.randn(16, 129, 129, 128, device="cuda")
    wp = torch.randn(256, geom.Kp, device="cuda") * (9 * 128) ** -0.5
    timer = C.KernelTimer()
    try:
        C.TIMER = timer
        for on in (True, False):
            C.nat.call("diagan_conv_gemm_set_x3b", 1 if on else 0)
            assert C.out_map_ok(geom, 16, 129, 129) == on
            y = C.conv_fwd(geom, x, wp)
            if on:
                y_on = y
        close(y, y_on, tol=2e-5)
    finally:
        C.TIMER = None
        C.nat.call("diagan_conv_gemm_set_x3b", -1)
    names = [r[0] for r in timer.records]
    assert names[0].startswith("conv_gemm_x3b_kernel<0,false>") and names[1].startswith("conv_gemm_kernel<128,128"), names


@pytest.mark.parametrize("case", [(2, 33, 33, 128, 128, 3, 3, 2, 0), (3, 20, 24, 64, 128, 2, 2, 1, 1), (2, 16, 16, 128, 256, 1, 1, 1, 0),
                                  (2, 19, 21, 64, 128, 2, 1, 1, 1), (2, 19, 21, 64, 128, 1, 2, 1, 1), (1, 5, 5, 128, 128, 1, 1, 1, 0),
                                  (5, 3, 3, 32, 128, 2, 2, 1, 1), (2, 31, 17, 128, 128, 3, 3, 2, 1), (3, 9, 40, 32, 256, 2, 2, 1, 0),
                                  (40, 16, 16, 128, 128, 1, 1, 1, 0)])
def test_split_operand_weight_gradient(case):
    """conv_wgrad_x3_kernel (both operands split exactly in three bf16 pieces in the loader, six piece products on the bf16 matrix
    pipe): the weight gradient of strided / padded / 1x1 / 2x1 / 1x2 geometries -- pixel counts that are no multiple of the 32-pixel
    K-step, rows narrower than a loader's 4 pixels, several splits -- against float64 autograd; its error must be fp32-grade, i.e.
    within 4x the fp32 kernel's on the same launch."""
    from diagan.ops import conv as C
    B, H, W, Ci, Co, R, S, st, pd = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Ci, H, W, generator=g, dtype=torch.float64)
    w = torch.zeros(Co, Ci, R, S, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, w, stride=st, padding=pd)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    ref = w.grad.permute(0, 2, 3, 1).reshape(Co, R * S * Ci)
    geom = C.Geom("conv", Ci, Co, R, S, st, pd)
    xd = x.float().permute(0, 2, 3, 1).contiguous().cuda()
    dyd = dy.float().permute(0, 2, 3, 1).contiguous().cuda()
    errs = []
    try:
        for on in (0, 2):
            C.set_wgrad_x3(on)
            assert C.wgrad_uses_x3(geom, B, H, W, dyd.shape[1], dyd.shape[2]) == bool(on)
            grad = torch.full((Co, geom.Kp), 3.0, device="cuda")
            C.conv_wgrad(geom, dyd, xd, grad, False)
            errs.append((grad.double().cpu() - ref).abs().max().item() / ref.abs().max().item())
            if on:       # accumulating form, and the same bits on a second launch (fixed summation order)
                again = torch.zeros_like(grad)
                C.conv_wgrad(geom, dyd, xd, again, False)
                assert torch.equal(again, grad)
                C.conv_wgrad(geom, dyd, xd, again, True)
                close(again, 2 * grad, tol=1e-6)
    finally:
        C.set_wgrad_x3(None)
    assert errs[1] < 2e-6 and errs[1] < 4 * max(errs[0], 1e-7), errs


def test_two_piece_mode_is_opt_in_and_priced():
    """round 6: DIAGAN_X3_PIECES=2 / set_x3_pieces(2) drops the third piece pair of the large split-operand kernels (forward / data gradient,
    weight gradient): operands at ~2^-16.  The default is three pieces (fp32-grade); the opt-in mode's error against float64 is the
    class of the F(4x4) Winograd layers (1e-5 of the output scale), far from bf16's 4e-3."""
    from diagan.ops import conv as C
    assert C.nat.fn("diagan_conv_gemm_get_x3_pieces")() == 3
    g = torch.Generator().manual_seed(7)
    B, H, Ci, Co = 4, 33, 128, 128
    x = torch.randn(B, Ci, H, H, generator=g, dtype=torch.float64)
    w = (torch.randn(Co, Ci, 3, 3, generator=g, dtype=torch.float64) * (9 * Ci) ** -0.5).requires_grad_(True)
    y = F.conv2d(x, w, stride=2)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    geom = C.Geom("conv", Ci, Co, 3, 3, 2, 0)
    xd = x.float().permute(0, 2, 3, 1).contiguous().cuda()
    wp = C.pack_oihw(w.detach().float().cuda(), geom.Kp)
    dyd = dy.float().permute(0, 2, 3, 1).contiguous().cuda()
    yref = y.detach().permute(0, 2, 3, 1)
    gref = w.grad.permute(0, 2, 3, 1).reshape(Co, 9 * Ci)
    errs = {}
    try:
        C.set_wgrad_x3(2)
        for n in (3, 2):
            C.set_x3_pieces(n)
            assert C.nat.fn("diagan_conv_gemm_get_x3_pieces")() == n
            out = C.conv_fwd(geom, xd, wp, tile_cfg=17)
            grad = torch.zeros(Co, geom.Kp, device="cuda")
            C.conv_wgrad(geom, dyd, xd, grad, False)
            errs[n] = ((out.double().cpu() - yref).abs().max().item() / yref.abs().max().item(),
                       (grad.double().cpu() - gref).abs().max().item() / gref.abs().max().item())
    finally:
        C.set_x3_pieces(None)
        C.set_wgrad_x3(None)
    assert errs[3][0] < 2e-6 and errs[3][1] < 2e-6, errs
    assert 2e-6 < errs[2][0] < 6e-5 and 2e-6 < errs[2][1] < 6e-5, errs


def test_split_operand_weight_gradient_follows_the_exact_fp32_switch():
    """set_gemm_x3(False) -- the exact-fp32 mode of the parity tools -- keeps the weight gradient on the fp32 pipe as well; launches
    with a prologue or a bias column stay on the fp32 kernel"""
    from diagan.ops import conv as C
    geom = C.Geom("conv", 128, 256, 3, 3, 2, 0)
    try:
        C.set_wgrad_x3(2)
        assert C.wgrad_uses_x3(geom, 4, 65, 65, 32, 32)
        assert not C.wgrad_uses_x3(geom, 4, 65, 65, 32, 32, mode=1)
        assert not C.wgrad_uses_x3(geom, 4, 65, 65, 32, 32, bias_off=256 * geom.Kp)
        assert not C.wgrad_uses_x3(C.Geom("conv", 128, 192, 3, 3, 2, 0), 4, 65, 65, 32, 32)       # Co % 128
        assert not C.wgrad_uses_x3(C.Geom("conv", 128, 256, 3, 3, 1, 1), 4, 64, 64, 64, 64)       # the Winograd kernel's
        C.set_gemm_x3(False)
        assert not C.wgrad_uses_x3(geom, 4, 65, 65, 32, 32)
    finally:
        C.set_gemm_x3(None)
        C.set_wgrad_x3(None)


@pytest.mark.parametrize("Ci,H,W,B", [(256, 32, 32, 4), (128, 16, 16, 3), (64, 12, 20, 2), (64, 64, 64, 2)])
@pytest.mark.parametrize("pro", [0, 1, 2])
def test_small_co_kernel_wgrad(Ci, H, W, B, pro):
    """conv3x3_co4_wgrad: weight + bias gradient of a 3x3 conv to 4 (RGB+pad) channels, partial slabs summed
    over splits, against autograd of F.conv2d on the prologue-transformed input."""
    from diagan.ops import conv as C
    g = torch.Generator().manual_seed(Ci + pro)
    x = torch.randn(B, Ci, H, W, generator=g)
    scale, shift = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.3
    w = (torch.randn(4, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5).requires_grad_(True)
    bias = torch.zeros(4, requires_grad=True)
    y = F.conv2d(ref_pro(x, pro, scale, shift), w, bias, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    geom = C.Geom("conv", Ci, 4, 3, 3, 1, 1)
    assert C.small_co_wgrad(geom)
    splits = C.small_co_wgrad_splits(B, H)
    stride = 4 * geom.Kp + 4
    slab = torch.full((splits, stride), float('nan'), device="cuda")
    C.conv_wgrad_into(geom, nhwc(dy).cuda(), nhwc(x).cuda(), slab, splits, stride, 4 * geom.Kp,
                      pro=(pro, scale.cuda(), shift.cuda()))
    tot = slab.double().sum(0).float().cpu()
    assert torch.isfinite(tot).all()
    close(C.unpack_oihw(tot[:4 * geom.Kp].view(4, geom.Kp), 4, Ci, 3, 3), w.grad)
    close(tot[4 * geom.Kp:], bias.grad)


def test_full_size_sngan32_g_block4():
    """BASELINE configs[1] dominant GEMM: M=65536, N=256, K=2304 (SNGAN G-32 block4.c1)."""
    from diagan.ops import conv as C
    case = ("conv", 64, 32, 32, 256, 256, 3, 1, 1)
    geom, x, w, wp = make(*case)
    y = C.conv_fwd(geom, nhwc(x).cuda(), wp)
    close(nchw(y), ref_fwd("conv", x, w, None, 1, 1))


@pytest.mark.parametrize("winograd", [True, False])
@pytest.mark.parametrize("B,H,Ci,Co,pro", [
    (128, 32, 128, 128, 1), (64, 32, 256, 256, 2),             # D-32 block1.c2 (real+fake pass), G-32 block4.c2
    (64, 64, 64, 64, 1),                                       # SNGAN-64 D block1.c2 / G block5.c2: (M, N, K) = (262144, 64, 576)
    (64, 8, 1024, 512, 2),                                     # SNGAN-64 G block2.c1: (4096, 512, 9216)
    (64, 4, 512, 1024, 1)])                                    # SNGAN-64 D block5.c2: (1024, 1024, 4608)
def test_full_size_backward_sampled(B, H, Ci, Co, pro, winograd):
    """BASELINE-size data- and weight-gradients of the SNGAN-32 and SNGAN-64 block shapes at batch 64, with the Winograd
    kernels (the default where they qualify) and with the implicit GEMM only, checked on SAMPLED output elements against
    float64 sums taken straight from the definition, plus two size-independent properties: linearity of the weight
    gradient in dy, and <dy, conv(x)> == <dW, W>."""
    from diagan.ops import conv as C
    C.set_winograd(winograd)
    try:
        _full_size_backward_sampled(B, H, Ci, Co, pro)
    finally:
        C.set_winograd(None)


def _full_size_backward_sampled(B, H, Ci, Co, pro):
    from diagan.ops import conv as C
    g = torch.Generator().manual_seed(B + Ci)
    x = torch.randn(B, H, H, Ci, generator=g)
    dy = torch.randn(B, H, H, Co, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5
    scale, shift = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.3
    geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
    wp = C.pack_oihw(w, geom.Kp).cuda()
    xa = ref_pro(x.permute(0, 3, 1, 2), pro, scale, shift).permute(0, 2, 3, 1).contiguous()      # activated input, NHWC
    xd, dyd = x.cuda(), dy.cuda()
    pro_t = (pro, scale.cuda(), shift.cuda())
    # weight gradient through the slab path the networks use
    splits = C.wgrad_splits_geom(geom, B, H, H, H, H)
    stride = Co * geom.Kp + Co
    slab = torch.empty(splits * stride, device="cuda")
    C.conv_wgrad_into(geom, dyd, xd, slab, splits, stride, Co * geom.Kp, pro=pro_t)
    tot = slab.view(splits, stride).double().sum(0)
    dW = tot[: Co * geom.Kp].view(Co, 3, 3, Ci)                     # packed k = (r*3+s)*Ci + c
    db = tot[Co * geom.Kp:]
    idx = torch.randint(0, Co * 9 * Ci, (48,), generator=g)
    xa64, dy64 = xa.double(), dy.double()
    xpad = torch.nn.functional.pad(xa64, (0, 0, 1, 1, 1, 1))        # zero padding AFTER the prologue
    for i in idx.tolist():
        n, r, s_, c = i // (9 * Ci), (i // (3 * Ci)) % 3, (i // Ci) % 3, i % Ci
        ref = (dy64[:, :, :, n] * xpad[:, r:r + H, s_:s_ + H, c]).sum().item()
        got = dW[n, r, s_, c].item()
        assert abs(got - ref) <= 2e-4 * (abs(ref) + 30.0), (n, r, s_, c, got, ref)
    np_db = dy64.sum(dim=(0, 1, 2))
    assert (db.cpu() - np_db).abs().max() <= 1e-3 * np_db.abs().max() + 1e-2
    # data gradient (a full GEMM launch), sampled
    wd = torch.zeros((Ci, geom.Kd), device="cuda")
    C.pack_weights(wp, Co, Ci, 9, geom.Kp, geom.Kd, Wd=wd)
    dx = C.conv_dgrad(geom, dyd, wd, (H, H)).cpu().double()
    dypad = torch.nn.functional.pad(dy64, (0, 0, 1, 1, 1, 1))
    w64 = w.double()
    for i in torch.randint(0, B * H * H * Ci, (48,), generator=g).tolist():
        c, ix, iy, b = i % Ci, (i // Ci) % H, (i // (Ci * H)) % H, i // (Ci * H * H)
        # dx[b,iy,ix,c] = sum_{n,r,s} dy[b, iy-r+1, ix-s+1, n] * w[n,c,r,s]
        patch = dypad[b, iy:iy + 3, ix:ix + 3, :].flip(0, 1)        # [r, s, n] -> dy[b, iy+1-r, ix+1-s, n]
        ref = (patch * w64[:, c].permute(1, 2, 0)).sum().item()
        assert abs(dx[b, iy, ix, c].item() - ref) <= 2e-4 * (abs(ref) + 1.0), (b, iy, ix, c)
    # properties at full size: linearity in dy, and the adjoint identity <dy, conv(a)> == <dW, W>
    slab2 = torch.empty_like(slab)
    C.conv_wgrad_into(geom, (2.5 * dyd).contiguous(), xd, slab2, splits, stride, Co * geom.Kp, pro=pro_t)
    tot2 = slab2.view(splits, stride).double().sum(0)
    assert (tot2 - 2.5 * tot).abs().max() <= 1e-4 * tot.abs().max()
    y = C.conv_fwd(geom, xd, wp, pro=pro_t).double()
    lhs = (y * dyd.double()).sum().item()
    rhs = (dW.cuda() * wp.double().view(Co, 3, 3, Ci)).sum().item()
    assert abs(lhs - rhs) <= 2e-5 * (abs(lhs) + abs(rhs) + 1e3), (lhs, rhs)


def test_spectral_norm_forward_backward():
    """torch_mimicry SpectralNorm semantics: one power iteration, sigma = u W v^T, W/sigma used by
    the conv, gradient through sigma with u, v constant."""
    from diagan.ops import conv as C
    Co, Ci, R = 64, 32, 3
    g = torch.Generator().manual_seed(3)
    w = torch.randn(Co, Ci, R, R, generator=g) * 0.1
    u0 = torch.randn(1, Co, generator=g)
    geom = C.Geom("conv", Ci, Co, R, R, 1, 1)
    wp = C.pack_oihw(w, geom.Kp).cuda()
    # reference (written from SURVEY §8 a8)
    wr = w.clone().requires_grad_(True)
    Wm = wr.view(Co, -1)
    with torch.no_grad():
        v = F.normalize(torch.matmul(u0, Wm), eps=1e-12)
        u1 = F.normalize(torch.matmul(v, Wm.t()), eps=1e-12)
    sigma = torch.mm(u1, torch.mm(Wm, v.t()))
    w_sn = wr / sigma
    x = torch.randn(2, Ci, 8, 8, generator=g)
    y = F.conv2d(x, w_sn, None, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    # device
    u_buf = u0.view(-1).clone().cuda()
    s_buf = torch.ones(1, device="cuda")
    u, vv, state = C.sn_power_iter(wp, u_buf, s_buf, training=True)
    close(u.view(1, -1), u1, 1e-5)
    close(u_buf.view(1, -1), u1, 1e-5)
    assert abs(state[0].item() - sigma.item()) < 1e-5 * abs(sigma.item())
    assert abs(s_buf.item() - sigma.item()) < 1e-5 * abs(sigma.item())
    wf = torch.zeros_like(wp)
    wd = torch.zeros((Ci, geom.Kd), device="cuda")
    C.pack_weights(wp, Co, Ci, R * R, geom.Kp, geom.Kd, inv_sigma=state[1:], Wf=wf, Wd=wd)
    yd = C.conv_fwd(geom, nhwc(x).cuda(), wf)
    close(nchw(yd), y.detach(), 1e-4)
    grad = torch.zeros_like(wp)
    C.conv_wgrad(geom, nhwc(dy).cuda(), nhwc(x).cuda(), grad, accumulate=False, sn=(wp, u, vv, state))
    close(C.unpack_oihw(grad, Co, Ci, R, R), wr.grad, 5e-4)
    # eval mode must not touch the buffers
    before = u_buf.clone()
    C.sn_power_iter(wp, u_buf, s_buf, training=False)
    assert torch.equal(before, u_buf)


_HALF_CHILD = r"""
import sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
from diagan.ops import conv as C
out = {}
for (B, H, Ci, Co, pro) in ((16, 32, 128, 128, 1), (8, 16, 256, 256, 2), (6, 12, 128, 256, 0)):
    g = torch.Generator().manual_seed(B + H)
    x, dy = torch.randn(B, H, H, Ci, generator=g).cuda(), torch.randn(B, H, H, Co, generator=g).cuda()
    sc, sh = (torch.rand(Ci, generator=g) + 0.5).cuda(), (torch.randn(Ci, generator=g) * 0.3).cuda()
    geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
    grad = torch.zeros(Co, geom.Kp, device="cuda")
    C.conv_wgrad(geom, dy, x, grad, accumulate=False, pro=(pro, sc, sh))
    out[(B, H, Ci, Co, pro)] = grad.cpu()
torch.save(out, sys.argv[3])
"""