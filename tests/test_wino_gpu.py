"""GPU: the Winograd F(2x2,3x3) kernel (tile_cfg 9, csrc/conv_wino.hip) against float64 PyTorch references of the same
convolution (F.conv2d and its input gradient, as used by mimicry's GBlock / DBlock: predefined_models.py:19-21,38-40),
with every fused prologue and epilogue, and against the implicit-GEMM kernel.  Tolerance: 2e-5 of the output scale
(the implicit GEMM itself sits at ~1e-6; Winograd's transforms add a factor of a few)."""
import pytest
import torch
import torch.nn.functional as F

from test_conv_gpu import close, nchw, nhwc, ref_pro

pytestmark = pytest.mark.gpu

# B, H, W, Ci, Co: square / non-square images, ragged tile counts, Co not a multiple of 64, the SNGAN block shapes
CASES = [(4, 8, 8, 64, 64), (3, 6, 10, 16, 24), (5, 16, 16, 128, 72), (2, 32, 32, 256, 256), (8, 64, 64, 64, 64),
         (16, 4, 4, 512, 256), (1, 2, 2, 8, 4)]


def make(B, H, W, Ci, Co, seed=0):
    from diagan.ops import conv as C
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
    geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
    return geom, x, w, C.pack_oihw(w, geom.Kp).cuda()


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("pro", [0, 1, 2, 3, 4])
def test_forward_all_prologues_with_bias_and_residual(case, pro):
    from diagan.ops import conv as C
    B, H, W, Ci, Co = case
    geom, x, w, wp = make(*case)
    g = torch.Generator().manual_seed(1)
    bias, scale, shift = torch.randn(Co, generator=g), torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.3
    ref = F.conv2d(ref_pro(x.double(), pro, scale.double(), shift.double()), w.double(), bias.double(), padding=1)
    res = torch.randn(ref.shape, generator=g)
    y = C.conv_fwd(geom, nhwc(x).cuda(), wp, bias=bias.cuda(), residual=nhwc(res).cuda(),
                   pro=(pro, scale.cuda(), shift.cuda()), tile_cfg=9)
    close(nchw(y), ref + res.double(), tol=2e-5)


@pytest.mark.parametrize("case", CASES[:6])
def test_data_gradient_with_mask_and_residual(case):
    """dx = conv^T(dy) (+ residual) * relu'(mask): the backward of DBlock's c1 / c2 (Winograd with the taps reversed)"""
    from diagan.ops import conv as C
    B, H, W, Ci, Co = case
    geom, x, w, wp = make(*case)
    g = torch.Generator().manual_seed(2)
    dy = torch.randn(B, Co, H, W, generator=g)
    xr = x.double().requires_grad_(True)
    F.conv2d(xr, w.double(), padding=1).backward(dy.double())
    msk, res = torch.randn(B, Ci, H, W, generator=g), torch.randn(B, Ci, H, W, generator=g)
    wd = torch.zeros(Ci, geom.Kd, device="cuda")
    C.pack_weights(wp, Co, Ci, 9, geom.Kp, geom.Kd, Wd=wd)
    dx = C.conv_dgrad(geom, nhwc(dy).cuda(), wd, (H, W), residual=nhwc(res).cuda(), mask_src=nhwc(msk).cuda(), tile_cfg=9)
    close(nchw(dx), (xr.grad + res.double()) * (msk > 0).double(), tol=2e-5)
    dx = C.conv_dgrad(geom, nhwc(dy).cuda(), wd, (H, W), mask_src=nhwc(msk).cuda(), mask_slope=0.2, tile_cfg=9)
    close(nchw(dx), torch.where(msk > 0, xr.grad, 0.2 * xr.grad), tol=2e-5)


def test_pair_scales_res_relu_statistics_and_groups():
    """the remaining epilogue / prologue modes of diagan_conv_gemm on the Winograd path: per-half 1/sigma (D(real) and
    D(fake) as one pass), max(residual, 0) (DBlock's aliased shortcut), BatchNorm statistics from the epilogue, and the
    per-group affine prologue of the stacked generator forward"""
    from diagan.ops import conv as C
    B, H, W, Ci, Co = 8, 16, 16, 64, 96
    geom, x, w, wp = make(B, H, W, Ci, Co, seed=3)
    g = torch.Generator().manual_seed(4)
    xg = nhwc(x).cuda()
    ref = F.conv2d(F.relu(x.double()), w.double(), padding=1)
    s0, s1 = torch.tensor([0.7]), torch.tensor([1.3])
    y = C.conv_fwd(geom, xg, wp, pro=(C.PRO_RELU, None, None), row_scale=(s0.cuda(), s1.cuda()), tile_cfg=9)
    half = torch.cat([ref[: B // 2] * 0.7, ref[B // 2:] * 1.3])
    close(nchw(y), half, tol=2e-5)
    res = torch.randn(ref.shape, generator=g)
    y = C.conv_fwd(geom, xg, wp, pro=(C.PRO_RELU, None, None), residual=nhwc(res).cuda(), res_relu=True, tile_cfg=9)
    close(nchw(y), ref + F.relu(res.double()), tol=2e-5)
    bias = torch.randn(Co, generator=g)
    y, st = C.conv_fwd(geom, xg, wp, bias=bias.cuda(), tile_cfg=9, want_stats=True)
    full = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
    close(st[0][:, 0].sum(0), full.sum((0, 2, 3)), tol=2e-5)
    close(st[0][:, 1].sum(0), (full * full).sum((0, 2, 3)), tol=2e-5)
    groups = 4                                                   # 2 images per group, 2 * 8 * 8 = 128 tiles... rows 512 = 2 x 256
    sc, sh = torch.rand(groups, Ci, generator=g) + 0.5, torch.randn(groups, Ci, generator=g) * 0.3
    y = C.conv_fwd(geom, xg, wp, pro=(C.PRO_AFFINE_RELU, sc.cuda(), sh.cuda(), B // groups), tile_cfg=9)
    parts = [F.conv2d(F.relu(x[2 * k: 2 * k + 2].double() * sc[k].double().view(1, -1, 1, 1) + sh[k].double().view(1, -1, 1, 1)),
                      w.double(), padding=1) for k in range(groups)]
    close(nchw(y), torch.cat(parts), tol=2e-5)


def test_auto_selection_and_agreement_with_the_implicit_gemm():
    """tile_cfg 0 picks Winograd for a qualifying layer with enough workgroups and the implicit GEMM otherwise; both
    kernels agree to rounding on the same inputs"""
    from diagan import _native as nat
    from diagan.ops import conv as C
    pick_any = nat.fn("diagan_conv_gemm_pick_cfg_geom")

    def pick(*a):                    # the choice between THIS kernel and the implicit GEMM (F(4x4), tile_cfg 13, set aside)
        C.set_winograd4(False)
        try:
            return pick_any(*a)
        finally:
            C.set_winograd4(None)
    ws = 16 << 20
    assert pick(128, 32, 32, 128, 32, 32, 128, 3, 3, 1, 1, -1, 1, 1152, 1, ws) == 9          # D32 block1.c2 (pair pass)
    assert pick(128, 32, 32, 128, 32, 32, 128, 3, 3, 1, -1, 1, 1, 1152, 1, ws) == 9          # its data-gradient
    assert pick(64, 4, 4, 512, 4, 4, 512, 3, 3, 1, 1, -1, 1, 4608, 1, ws) == 9               # few tiles, long channel loop: split-K
    assert pick(64, 4, 4, 512, 4, 4, 512, 3, 3, 1, 1, -1, 1, 4608, 0, ws) != 9               # ... not without a slab
    assert pick(128, 8, 8, 128, 8, 8, 128, 3, 3, 1, 1, -1, 1, 1152, 1, ws) != 9              # too few workgroups AND channels
    assert pick(64, 16, 16, 128, 8, 8, 128, 3, 3, 2, 1, -1, 1, 1152, 1, ws) != 9             # stride 2
    assert pick(64, 32, 32, 4, 32, 32, 128, 3, 3, 1, 1, -1, 1, 64, 1, ws) != 9               # RGB input
    assert pick(128, 32, 32, 128, 32, 32, 128, 3, 3, 1, 1, -1, 1, 1152, 1, 1024) != 9        # no room for the transformed weights
    geom, x, w, wp = make(16, 16, 16, 128, 128, seed=5)
    a = C.conv_fwd(geom, nhwc(x).cuda(), wp, tile_cfg=9)
    b = C.conv_fwd(geom, nhwc(x).cuda(), wp, tile_cfg=7)
    close(a, b, tol=1e-5)
    # split-K of the Winograd kernel (raw partial outputs + the shared second stage) on a few-tile, many-channel layer
    geom, x, w, wp = make(64, 4, 4, 512, 512, seed=6)
    g = torch.Generator().manual_seed(7)
    bias, res = torch.randn(512, generator=g), torch.randn(64, 512, 4, 4, generator=g)
    y = C.conv_fwd(geom, nhwc(x).cuda(), wp, bias=bias.cuda(), residual=nhwc(res).cuda(), pro=(C.PRO_RELU, None, None))
    ref = F.conv2d(F.relu(x.double()), w.double(), bias.double(), padding=1) + res.double()
    close(nchw(y), ref, tol=2e-5)
    with pytest.raises(RuntimeError, match="Winograd"):
        g2 = C.Geom("conv", 512, 512, 3, 3, 2, 1)
        C.conv_fwd(g2, nhwc(x).cuda(), wp, tile_cfg=9)


@pytest.mark.parametrize("case", [(128, 32, 32, 128, 128), (64, 16, 16, 128, 256), (128, 8, 8, 256, 512), (64, 4, 4, 512, 512),
                                  (3, 6, 10, 16, 24)])
def test_data_gradient_adds_an_unpooled_half_resolution_residual(case):
    """Round 5: DBlock's shortcut is pooled before its 1x1 convolution, so its gradient arrives at half resolution; the data
    gradient of c1 adds avg_pool2d_backward of it (a quarter of the value at (y / 2, x / 2)) in its own epilogue (res_relu bit 2;
    F(4x4), F(2x2) and the split-K second stage) instead of reading a tensor diagan_avgpool2_bwd wrote: the same products, so the
    same bits as the two-launch path; and against float64."""
    from diagan.ops import conv as C, eltwise as E
    B, H, W, Ci, Co = case
    geom, x, w, wp = make(*case, seed=31)
    wd = torch.zeros(Ci, geom.Kd, device="cuda")
    C.pack_weights(wp, Co, Ci, 9, geom.Kp, geom.Kd, Wd=wd)
    g = torch.Generator(device="cuda").manual_seed(32)
    gy = torch.randn(B, H, W, Co, device="cuda", generator=g)
    lo = torch.randn(B, H // 2, W // 2, Ci, device="cuda", generator=g)
    msk = torch.randn(B, H, W, Ci, device="cuda", generator=g)
    if not C.res_unpool_fused(geom, B, H, W):
        with pytest.raises(RuntimeError, match="Winograd"):
            C.conv_dgrad(geom, gy, wd, (H, W), residual=lo, mask_src=msk, res_unpool=True)
        cfgs = (9,)
    else:
        cfgs = (0, 9)
    for cfg in cfgs:
        a = C.conv_dgrad(geom, gy, wd, (H, W), residual=lo, mask_src=msk, res_unpool=True, tile_cfg=cfg)
        b = C.conv_dgrad(geom, gy, wd, (H, W), residual=E.avgpool2_bwd(lo), mask_src=msk, tile_cfg=cfg)
        assert torch.equal(a, b)
        a = C.conv_dgrad(geom, gy, wd, (H, W), residual=lo, res_unpool=True, tile_cfg=cfg)
        assert torch.equal(a, C.conv_dgrad(geom, gy, wd, (H, W), residual=E.avgpool2_bwd(lo), tile_cfg=cfg))
    ref = F.conv_transpose2d(nchw(gy).double().cpu(), w.double(), padding=1) + \
        0.25 * F.interpolate(nchw(lo).double().cpu(), scale_factor=2, mode="nearest")
    close(nchw(a), ref, tol=1e-4)


@pytest.mark.parametrize("case", [(64, 4, 4, 512, 512), (64, 8, 8, 256, 512), (128, 4, 4, 1024, 1024), (64, 8, 8, 512, 256)])
def test_split_k_combine_by_the_last_arriving_workgroup(case):
    """Round 5: a split-K launch's partial sums are added, in slab order, by the tile's LAST workgroup to deliver, which then
    runs the epilogue itself (ticket counter per tile; no second launch).  Same inputs through diagan_conv_gemm_set_splitk_fused(0)
    (the splitk_epilogue_kernel launch): the same sums in the same order -- forward with bias + residual, forward with a
    half-resolution residual, data gradient with a mask -- equal up to the fused multiply-add of scale and bias; repeated launches
    (the counters return to zero) give the same bits; and against float64."""
    from diagan.ops import conv as C
    B, H, W, Ci, Co = case
    geom, x, w, wp = make(*case, seed=21)
    g = torch.Generator().manual_seed(22)
    bias, res = torch.randn(Co, generator=g), torch.randn(B, Co, H, W, generator=g)
    low = torch.randn(B, Co, H // 2, W // 2, generator=g)
    xc, rc, lc = nhwc(x).cuda(), nhwc(res).cuda(), nhwc(low).cuda()
    relu = (C.PRO_RELU, None, None)
    wd = torch.zeros(Ci, geom.Kd, device="cuda")
    C.pack_weights(wp, Co, Ci, 9, geom.Kp, geom.Kd, Wd=wd)
    gy = torch.randn(B, H, W, Co, device="cuda", generator=torch.Generator(device="cuda").manual_seed(23))
    msk = torch.randn(B, H, W, Ci, device="cuda", generator=torch.Generator(device="cuda").manual_seed(24))
    out = {}
    C.set_winograd4(False)                   # these shapes: the F(2x2) kernel, split 2 - 4 ways by the automatic choice
    try:
        for fused in (True, False, True):
            C.set_splitk_fused(fused)
            cur = (C.conv_fwd(geom, xc, wp, bias=bias.cuda(), residual=rc, pro=relu),
                   C.conv_fwd(geom, xc, wp, bias=bias.cuda(), residual=lc, res_up=True),
                   C.conv_dgrad(geom, gy, wd, (H, W), mask_src=msk))
            if fused in out:
                for a, b in zip(out[fused], cur):
                    assert torch.equal(a, b)
            out[fused] = cur
    finally:
        C.set_splitk_fused(None)
        C.set_winograd4(None)
    for a, b in zip(out[True], out[False]):
        close(a, b, tol=1e-6)
    ref = F.conv2d(F.relu(x.double()), w.double(), bias.double(), padding=1) + res.double()
    close(nchw(out[True][0]), ref, tol=2e-5)


@pytest.mark.parametrize("case", [(2, 32, 32, 256, 256), (3, 6, 10, 16, 24), (5, 16, 16, 128, 72), (64, 4, 4, 512, 512)])
def test_half_resolution_residual_is_upsampled_in_the_epilogue(case):
    """mimicry GBlock's shortcut is c_sc(interpolate(x, scale_factor=2, mode='bilinear')) (= interpolate(c_sc(x)), a 1x1
    convolution): the Winograd epilogue and its split-K second stage blend the x2 up-sampling of the half-resolution
    residual themselves (res_relu bit 1).  Against F.interpolate in float64, and against the same launch given the
    tensor diagan_upsample2x materialises (same blend arithmetic: equal to an fp32 rounding of the blend)."""
    from diagan.ops import conv as C, eltwise as E
    B, H, W, Ci, Co = case
    geom, x, w, wp = make(*case, seed=11)
    g = torch.Generator().manual_seed(12)
    bias, low = torch.randn(Co, generator=g), torch.randn(B, Co, H // 2, W // 2, generator=g)
    ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1) + \
        F.interpolate(low.double(), scale_factor=2, mode="bilinear", align_corners=False)
    lo = nhwc(low).cuda()
    for cfg in (9, 0):                                     # forced, and the automatic choice (split-K on the 4x4 case)
        if cfg == 0 and not C.res_up_fused(geom, B, H, W):
            with pytest.raises(RuntimeError, match="Winograd"):
                C.conv_fwd(geom, nhwc(x).cuda(), wp, bias=bias.cuda(), residual=lo, res_up=True)
            continue
        y = C.conv_fwd(geom, nhwc(x).cuda(), wp, bias=bias.cuda(), residual=lo, res_up=True, tile_cfg=cfg)
        close(nchw(y), ref, tol=2e-5)
        z = C.conv_fwd(geom, nhwc(x).cuda(), wp, bias=bias.cuda(), residual=E.upsample2x(lo), tile_cfg=cfg)
        assert (y - z).abs().max().item() <= 1e-6 * max(1.0, z.abs().max().item())
    # fused BatchNorm statistics see the blended residual
    y, st = C.conv_fwd(geom, nhwc(x).cuda(), wp, bias=bias.cuda(), residual=lo, res_up=True, tile_cfg=9, want_stats=True)
    close(st[0][:, 0].sum(0), y.double().sum((0, 1, 2)), tol=1e-5)
    with pytest.raises(RuntimeError, match="Winograd"):
        C.conv_fwd(geom, nhwc(x).cuda(), wp, residual=lo, res_up=True, tile_cfg=7)
    with pytest.raises(RuntimeError, match="residual"):
        C.conv_fwd(geom, nhwc(x).cuda(), wp, residual=E.upsample2x(lo), res_up=True, tile_cfg=9)


def test_gblock_shortcut_fused_equals_materialised_upsampling():
    """GBlock forward with the shortcut's up-sampling blended into c2's epilogue (Winograd launches) against the same
    block with Winograd off, where ConvLayer falls back to diagan_upsample2x."""
    from diagan.models.sngan import GBlock
    from diagan.ops import conv as C
    torch.manual_seed(3)
    blk = GBlock(256, 256, upsample=True).cuda()
    x = torch.randn(16, 16, 16, 256, device="cuda")
    outs = []
    for on in (None, False):
        C.set_winograd(on)
        try:
            out, ctx, _ = blk.forward(x, True)
            outs.append(out)
        finally:
            C.set_winograd(None)
    assert C.res_up_fused(blk.c2.geom, 16, 32, 32)
    close(outs[0], outs[1], tol=2e-5)


POOL_CASES = [(4, 32, 32, 128, 128), (6, 16, 16, 64, 256), (3, 6, 10, 16, 128), (16, 8, 8, 256, 128), (2, 4, 4, 256, 128),
              (4, 32, 32, 64, 64), (3, 6, 10, 16, 192), (5, 16, 16, 128, 64)]      # ... 64 / 192 columns: 128-tile x 64-column workgroups


@pytest.mark.parametrize("case", POOL_CASES)
@pytest.mark.parametrize("pro", [0, 1])
def test_convolution_plus_average_pool_in_one_launch(case, pro):
    """tile_cfg 11 (csrc/conv_wino_pool.hip): F.avg_pool2d(conv3x3(pro(x)) + bias, 2) + residual -- the end of mimicry's
    DBlock / DBlockOptimized with downsample=True -- from 9 of the 16 Winograd products, against float64 PyTorch and
    against the two-launch path (Winograd convolution + diagan_avgpool2); with the pair pass's two 1/sigma scalars."""
    from diagan.ops import conv as C, eltwise as E
    B, H, W, Ci, Co = case
    geom, x, w, wp = make(*case, seed=21)
    g = torch.Generator().manual_seed(22)
    bias, res = torch.randn(Co, generator=g), torch.randn(B, Co, H // 2, W // 2, generator=g)
    ref = F.avg_pool2d(F.conv2d(ref_pro(x.double(), pro, None, None), w.double(), bias.double(), padding=1), 2) + res.double()
    y = C.conv_fwd(geom, nhwc(x).cuda(), wp, bias=bias.cuda(), residual=nhwc(res).cuda(), pro=(pro, None, None), pool=True)
    assert tuple(y.shape) == (B, H // 2, W // 2, Co)
    close(nchw(y), ref, tol=2e-5)
    two = E.avgpool2(C.conv_fwd(geom, nhwc(x).cuda(), wp, bias=bias.cuda(), pro=(pro, None, None), tile_cfg=9),
                     residual=nhwc(res).cuda())
    close(y, two, tol=5e-6)
    if B % 2 == 0:
        s0, s1 = torch.tensor([0.7]).cuda(), torch.tensor([1.9]).cuda()
        y = C.conv_fwd(geom, nhwc(x).cuda(), wp, pro=(pro, None, None), row_scale=(s0, s1), pool=True)
        ref = F.avg_pool2d(F.conv2d(ref_pro(x.double(), pro, None, None), w.double(), padding=1), 2)
        ref[:B // 2] *= 0.7
        ref[B // 2:] *= 1.9
        close(nchw(y), ref, tol=2e-5)


def test_pooled_launch_selection_and_refusals():
    from diagan import _native as nat
    from diagan.ops import conv as C
    ok = nat.fn("diagan_conv_wino_pool_supported")
    ws = 64 << 20
    assert ok(128, 32, 32, 128, 32, 32, 128, 3, 3, 1, 1, -1, 1, 1, ws) == 1          # D32 block1.c2, pair pass
    assert ok(128, 16, 16, 128, 16, 16, 128, 3, 3, 1, 1, -1, 1, 1, ws) == 1          # D32 block2.c2: split-K
    assert ok(128, 32, 32, 128, 32, 32, 64, 3, 3, 1, 1, -1, 1, 1, ws) == 1           # 64 output channels: 128-tile workgroups
    assert ok(128, 32, 32, 128, 32, 32, 32, 3, 3, 1, 1, -1, 1, 1, ws) == 0           # 32 output channels
    assert ok(128, 32, 32, 128, 32, 32, 128, 3, 3, 1, -1, 1, 1, 1, ws) == 0          # a data-gradient
    assert ok(128, 32, 32, 128, 32, 32, 128, 3, 3, 1, 1, -1, 1, 2, ws) == 0          # BatchNorm prologue
    assert ok(2, 4, 4, 256, 4, 4, 128, 3, 3, 1, 1, -1, 1, 1, ws) == 0                # too small to be worth it
    assert ok(128, 32, 32, 128, 32, 32, 128, 3, 3, 1, 1, -1, 1, 1, 1024) == 0        # no room for the weights
    geom, x, w, wp = make(4, 8, 8, 64, 32, seed=23)
    with pytest.raises(RuntimeError, match="tile_cfg 11"):
        C.conv_fwd(geom, nhwc(x).cuda(), wp, pool=True)                              # Co = 32
    C.set_winograd(False)
    try:
        assert ok(128, 32, 32, 128, 32, 32, 128, 3, 3, 1, 1, -1, 1, 1, ws) == 0
    finally:
        C.set_winograd(None)


@pytest.mark.parametrize("case", POOL_CASES)
def test_data_gradient_through_the_average_pool_in_one_launch(case):
    """tile_cfg 12: dx = conv^T(avg_pool2d_backward(g)) * relu'(mask) (+ residual) from the pooled gradient -- the backward
    of the down-sampling DBlocks' c2 -- against float64 autograd of avg_pool2d(conv2d(x)) and against the two-launch path
    (diagan_avgpool2_bwd + Winograd data-gradient); with the pair pass's two 1/sigma scalars."""
    from diagan.ops import conv as C, eltwise as E
    B, H, W, Co, Ci = case                                  # the data-gradient's 128-multiple is the layer's INPUT channels
    geom, x, w, wp = make(B, H, W, Ci, Co, seed=31)
    wd = torch.zeros(Ci, geom.Kd, device="cuda")
    C.pack_weights(wp, Co, Ci, 9, geom.Kp, geom.Kd, Wd=wd)
    g = torch.Generator().manual_seed(32)
    gp = torch.randn(B, Co, H // 2, W // 2, generator=g)
    mask, res = torch.randn(B, Ci, H, W, generator=g), torch.randn(B, Ci, H, W, generator=g)
    xx = x.double().requires_grad_(True)
    F.avg_pool2d(F.conv2d(xx, w.double(), padding=1), 2).backward(gp.double())
    ref = (xx.grad + res.double()) * (mask.double() > 0)
    dx = C.conv_dgrad(geom, nhwc(gp).cuda(), wd, (H, W), residual=nhwc(res).cuda(), mask_src=nhwc(mask).cuda(), unpool=True)
    close(nchw(dx), ref, tol=2e-5)
    two = C.conv_dgrad(geom, E.avgpool2_bwd(nhwc(gp).cuda()), wd, (H, W), residual=nhwc(res).cuda(), mask_src=nhwc(mask).cuda(),
                       tile_cfg=9)
    close(dx, two, tol=5e-6)
    if B % 2 == 0:
        s0, s1 = torch.tensor([0.7]).cuda(), torch.tensor([1.9]).cuda()
        dx = C.conv_dgrad(geom, nhwc(gp).cuda(), wd, (H, W), row_scale=(s0, s1), unpool=True)
        ref = xx.grad.clone()
        ref[:B // 2] *= 0.7
        ref[B // 2:] *= 1.9
        close(nchw(dx), ref, tol=2e-5)


@pytest.mark.parametrize("case", [(4, 32, 32, 128, 128), (6, 16, 16, 64, 256), (2, 6, 10, 16, 24)])
def test_weight_gradient_through_the_average_pool_as_a_strided_convolution(case):
    """d/dW of avg_pool2d(conv3x3(relu(x)), 2) against the pooled gradient g: the weight gradient of a 3x3 / stride 2 / pad 0
    convolution over diagan_boxsum2(x) ((H+1) x (W+1) box sums, x 1/4) against g -- what ConvLayer.wgrad_pooled launches --
    equals float64 autograd and the ordinary path (diagan_avgpool2_bwd + the layer's own weight gradient)."""
    from diagan.ops import conv as C, eltwise as E
    B, H, W, Ci, Co = case
    geom, x, w, wp = make(*case, seed=41)
    g2 = C.Geom("conv", Ci, Co, 3, 3, 2, 0)
    assert g2.Kp == geom.Kp
    gp = torch.randn(B, Co, H // 2, W // 2, generator=torch.Generator().manual_seed(42))
    ww = w.double().requires_grad_(True)
    F.avg_pool2d(F.conv2d(F.relu(x.double()), ww, padding=1), 2).backward(gp.double())
    ref = C.pack_oihw(ww.grad.float(), geom.Kp)
    ga, gb = torch.zeros(Co, geom.Kp, device="cuda"), torch.zeros(Co, geom.Kp, device="cuda")
    C.conv_wgrad(g2, nhwc(gp).cuda(), E.boxsum2(nhwc(x).cuda(), relu_in=True), ga, accumulate=False)
    close(ga, ref, tol=2e-5)
    C.conv_wgrad(geom, E.avgpool2_bwd(nhwc(gp).cuda()), nhwc(x).cuda(), gb, accumulate=False, pro=(C.PRO_RELU, None, None))
    close(ga, gb, tol=1e-5)
    # round 5: the box sums taken by the loader itself (prologue modes 5 / 6 of diagan_conv_wgrad: four loads from x per gathered
    # piece, added in diagan_boxsum2's order) -- the same operands, so the same gradient bit for bit
    gc = torch.zeros(Co, geom.Kp, device="cuda")
    C.conv_wgrad(g2, nhwc(gp).cuda(), nhwc(x).cuda(), gc, accumulate=False, pro=(C.PRO_BOX_RELU, None, None))
    assert torch.equal(gc, ga)
    C.conv_wgrad(g2, nhwc(gp).cuda(), E.boxsum2(nhwc(x).cuda(), relu_in=False), ga, accumulate=False)
    C.conv_wgrad(g2, nhwc(gp).cuda(), nhwc(x).cuda(), gc, accumulate=False, pro=(C.PRO_BOX, None, None))
    assert torch.equal(gc, ga)


@pytest.mark.parametrize("case", [(128, 32, 32, 128, 128), (64, 64, 64, 64, 64), (128, 16, 16, 128, 256)])
def test_pooled_launches_at_the_discriminators_full_sizes(case):
    """The three fused launches at the SNGAN-32 / SNGAN-64 discriminator's own sizes (pair pass of B = 128, 64 x 64 maps,
    split-K), against the two-launch path on the same inputs: a size-independent property (both paths are held to float64
    on the small cases above)."""
    from diagan.ops import conv as C, eltwise as E
    B, H, W, Ci, Co = case
    g = torch.Generator(device="cuda").manual_seed(51)
    geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
    x = torch.randn(B, H, W, Ci, device="cuda", generator=g)
    wp = torch.randn(Co, geom.Kp, device="cuda", generator=g) * (9 * Ci) ** -0.5
    wd = torch.zeros(Ci, geom.Kd, device="cuda")
    C.pack_weights(wp, Co, Ci, 9, geom.Kp, geom.Kd, Wd=wd)
    bias = torch.randn(Co, device="cuda", generator=g)
    sc = torch.randn(B, H // 2, W // 2, Co, device="cuda", generator=g)
    gp = torch.randn(B, H // 2, W // 2, Co, device="cuda", generator=g)
    relu = (C.PRO_RELU, None, None)
    assert C.pool_fused(geom, B, H, W, relu) and C.unpool_fused(geom, B, H, W)
    from diagan import _native as nat
    ws = C._splitk_ws(torch.device('cuda', 0)).numel()
    # at these sizes the pooled launches run on the F(4x4) kernel (25 products per 4x4 tile, rounding ~1e-5 of scale); with that
    # kernel set aside they run on conv_wino_pool.hip's F(2x2) kernels (9 products per 2x2 tile, ~1e-6): both against the
    # two-launch path
    for w4, tol in ((None, 1e-4), (False, 5e-6)):
        C.set_winograd4(w4)
        try:
            assert bool(nat.fn("diagan_conv_wino4_pool_used")(B, H, W, Ci, Co, ws)) == (w4 is None)
            close(C.conv_fwd(geom, x, wp, bias=bias, pro=relu, residual=sc, pool=True),
                  E.avgpool2(C.conv_fwd(geom, x, wp, bias=bias, pro=relu, tile_cfg=9), residual=sc), tol=tol)
            close(C.conv_dgrad(geom, gp, wd, (H, W), mask_src=x, unpool=True),
                  C.conv_dgrad(geom, E.avgpool2_bwd(gp), wd, (H, W), mask_src=x, tile_cfg=9), tol=tol)
        finally:
            C.set_winograd4(None)
    ga, gb = torch.zeros(Co, geom.Kp, device="cuda"), torch.zeros(Co, geom.Kp, device="cuda")
    C.set_wgrad_x3(False)       # (the plain strided launch of the first case is large enough for the split-operand kernel of round 6;
    try:                        #  the loader comparison below is between two launches of the SAME fp32 kernel)
        C.conv_wgrad(C.Geom("conv", Ci, Co, 3, 3, 2, 0), gp, E.boxsum2(x, relu_in=True), ga, accumulate=False)
        C.conv_wgrad(geom, E.avgpool2_bwd(gp), x, gb, accumulate=False, pro=relu)
        close(ga, gb, tol=2e-5)
        gc = torch.zeros(Co, geom.Kp, device="cuda")
        C.conv_wgrad(C.Geom("conv", Ci, Co, 3, 3, 2, 0), gp, x, gc, accumulate=False, pro=(C.PRO_BOX_RELU, None, None))
        assert torch.equal(gc, ga)               # the loader's box sums: bit-identical to the boxsum2 pass + gather
    finally:
        C.set_wgrad_x3(None)
