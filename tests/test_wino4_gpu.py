"""GPU: the Winograd F(4x4,3x3) kernel (tile_cfg 13, csrc/conv_wino4.hip) against float64 PyTorch references of the same
convolution (F.conv2d and its input gradient, as used by mimicry's GBlock / DBlock: predefined_models.py:19-21,38-40,
57-59,76-78) with every fused prologue and epilogue, against the F(2x2,3x3) kernel, and the automatic selection.
Tolerance: 1e-4 of the output scale -- the 6x6 transforms (interpolation points 0, +-1, +-2, inf) put the kernel at
~1e-5 (tools/micro/wino_f4_error.py: fp32 F(4x4) against float64 on CPU gives the same figure), against ~6e-7 for F(2x2)
and ~1e-6 for the implicit GEMM; north_star's bar is 1e-3 on losses."""
import pytest
import torch
import torch.nn.functional as F

from test_conv_gpu import close, nchw, nhwc, ref_pro
from test_wino_gpu import make

pytestmark = pytest.mark.gpu

TOL = 1e-4
# B, H, W, Ci, Co: ragged tile counts (tiles % 32 != 0), non-square images, Co not a multiple of 64, SNGAN block shapes
CASES = [(4, 8, 8, 64, 64), (3, 4, 12, 16, 24), (5, 16, 16, 128, 72), (2, 32, 32, 256, 256), (8, 64, 64, 64, 64),
         (16, 4, 4, 512, 256), (1, 4, 4, 8, 4)]


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
                   pro=(pro, scale.cuda(), shift.cuda()), tile_cfg=13)
    close(nchw(y), ref + res.double(), tol=TOL)


@pytest.mark.parametrize("case", CASES[:6])
def test_data_gradient_with_mask_and_residual(case):
    """dx = conv^T(dy) (+ residual) * relu'(mask): the backward of DBlock's / GBlock's c1, c2 (taps reversed in U)"""
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
    dx = C.conv_dgrad(geom, nhwc(dy).cuda(), wd, (H, W), residual=nhwc(res).cuda(), mask_src=nhwc(msk).cuda(), tile_cfg=13)
    close(nchw(dx), (xr.grad + res.double()) * (msk > 0).double(), tol=TOL)
    dx = C.conv_dgrad(geom, nhwc(dy).cuda(), wd, (H, W), mask_src=nhwc(msk).cuda(), mask_slope=0.2, tile_cfg=13)
    close(nchw(dx), torch.where(msk > 0, xr.grad, 0.2 * xr.grad), tol=TOL)


def test_pair_scales_res_relu_statistics_and_groups():
    """the epilogue / prologue variants the networks use: per-half 1/sigma of a paired D(real) | D(fake) pass, DBlock's
    relu(shortcut), BatchNorm statistics from the epilogue, one BatchNorm row per stacked batch (grouped prologue)"""
    from diagan.ops import conv as C
    B, H, W, Ci, Co = 8, 16, 16, 64, 128
    geom, x, w, wp = make(B, H, W, Ci, Co, seed=3)
    g = torch.Generator().manual_seed(4)
    xg = nhwc(x).cuda()
    ref = F.conv2d(F.relu(x.double()), w.double(), padding=1)
    s0, s1 = torch.tensor([0.7]), torch.tensor([1.9])
    y = C.conv_fwd(geom, xg, wp, pro=(C.PRO_RELU, None, None), row_scale=(s0.cuda(), s1.cuda()), tile_cfg=13)
    half = torch.cat([ref[: B // 2] * 0.7, ref[B // 2:] * 1.9])
    close(nchw(y), half, tol=TOL)
    res = torch.randn(ref.shape, generator=g)
    y = C.conv_fwd(geom, xg, wp, pro=(C.PRO_RELU, None, None), residual=nhwc(res).cuda(), res_relu=True, tile_cfg=13)
    close(nchw(y), ref + F.relu(res.double()), tol=TOL)
    bias = torch.randn(Co, generator=g)
    y, st = C.conv_fwd(geom, xg, wp, bias=bias.cuda(), tile_cfg=13, want_stats=True)
    full = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
    assert st[1] == B * H * W // 512 and st[0].shape == (st[1], 2, Co)
    close(st[0][:, 0].sum(0), full.sum((0, 2, 3)), tol=TOL)
    close(st[0][:, 1].sum(0), (full * full).sum((0, 2, 3)), tol=TOL)
    close(st[0][:, 0].sum(0), y.double().sum((0, 1, 2)), tol=1e-5)       # the sums are of the kernel's own outputs
    groups = 4
    sc, sh = torch.rand(groups, Ci, generator=g) + 0.5, torch.randn(groups, Ci, generator=g) * 0.3
    y = C.conv_fwd(geom, xg, wp, pro=(C.PRO_AFFINE_RELU, sc.cuda(), sh.cuda(), B // groups), tile_cfg=13)
    parts = [F.conv2d(ref_pro(x[i * 2: i * 2 + 2].double(), 2, sc[i].double(), sh[i].double()), w.double(), padding=1)
             for i in range(groups)]
    close(nchw(y), torch.cat(parts), tol=TOL)


@pytest.mark.parametrize("case", [(2, 32, 32, 256, 256), (3, 4, 12, 16, 24), (5, 16, 16, 128, 72)])
def test_half_resolution_residual_is_upsampled_in_the_epilogue(case):
    """GBlock: c2's output + bilinear x2 of the low-resolution shortcut conv (mimicry GBlock._upsample_conv on the
    shortcut, align_corners=False), blended in the epilogue from the half-resolution tensor"""
    from diagan.ops import conv as C
    B, H, W, Ci, Co = case
    geom, x, w, wp = make(*case, seed=5)
    g = torch.Generator().manual_seed(6)
    bias = torch.randn(Co, generator=g)
    lo = torch.randn(B, Co, H // 2, W // 2, generator=g)
    ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1) + \
        F.interpolate(lo.double(), scale_factor=2, mode='bilinear', align_corners=False)
    y = C.conv_fwd(geom, nhwc(x).cuda(), wp, bias=bias.cuda(), residual=nhwc(lo).cuda(), res_up=True, tile_cfg=13)
    close(nchw(y), ref, tol=TOL)


def test_auto_selection_split_k_and_refusals():
    """tile_cfg 0 takes the F(4x4) kernel for a qualifying launch of >= 512 workgroups (diagan_conv_gemm_pick_cfg_geom), F(2x2)
    or the implicit GEMM otherwise; DIAGAN_WINO4 / set_winograd4(False) keeps it out; a forced split over the input channels
    goes through the shared second stage; geometries the kernel cannot take are refused with the reason"""
    from diagan import _native as nat
    from diagan.ops import conv as C
    pick = nat.fn("diagan_conv_gemm_pick_cfg_geom")
    ws = C._splitk_ws(torch.device('cuda', 0)).numel()
    assert pick(64, 32, 32, 256, 32, 32, 256, 3, 3, 1, 1, -1, 1, 2304, 1, ws) == 13
    assert pick(64, 16, 16, 256, 16, 16, 256, 3, 3, 1, 1, -1, 1, 2304, 1, ws) == 13        # 128 workgroups, split two ways
    assert pick(64, 16, 16, 128, 16, 16, 128, 3, 3, 1, 1, -1, 1, 1152, 1, ws) == 9
    C.set_winograd4(False)
    try:
        assert pick(64, 32, 32, 256, 32, 32, 256, 3, 3, 1, 1, -1, 1, 2304, 1, ws) == 9
    finally:
        C.set_winograd4(None)
    geom, x, w, wp = make(64, 32, 32, 64, 128, seed=7)
    ref = F.conv2d(x[:4].double(), w.double(), padding=1)
    y = C.conv_fwd(geom, nhwc(x).cuda(), wp)                     # automatic: 256 workgroups of F(4x4)
    close(nchw(y[:4]), ref, tol=TOL)
    a = C.conv_fwd(geom, nhwc(x).cuda(), wp, tile_cfg=13)
    assert torch.equal(a, y)
    b = C.conv_fwd(geom, nhwc(x).cuda(), wp, tile_cfg=9)
    close(a, b, tol=TOL)
    nat.call("diagan_conv_gemm_tune", 2, -1, 0)                 # forced split-K (tuning entry point)
    try:
        s = C.conv_fwd(geom, nhwc(x).cuda(), wp, tile_cfg=13)
    finally:
        nat.call("diagan_conv_gemm_tune", 0, -1, 0)
    close(s, a, tol=1e-5)
    # a launch the policy splits two ways over the input channels (128 workgroups, 32 K-steps): same result as unsplit
    geom, x, w, wp = make(64, 16, 16, 256, 256, seed=8)
    auto = C.conv_fwd(geom, nhwc(x).cuda(), wp, bias=torch.ones(256, device='cuda'))
    close(auto, C.conv_fwd(geom, nhwc(x).cuda(), wp, bias=torch.ones(256, device='cuda'), tile_cfg=13), tol=TOL / 2)
    # 1.5 rounds of workgroups with a long K loop (SNGAN-64's stacked generator forward at 8x8: 384 workgroups, Ci = 1024):
    # F(4x4) split two ways rather than the F(2x2) kernel; same values as the unsplit launch
    assert pick(384, 8, 8, 1024, 8, 8, 512, 3, 3, 1, 1, -1, 1, 9216, 1, ws) == 13
    assert pick(384, 8, 8, 1024, 8, 8, 512, 3, 3, 1, 1, -1, 1, 9216, 0, ws) == 9           # no workspace for a slab: F(2x2)
    geom, x, w, wp = make(96, 8, 8, 512, 512, seed=9)            # 96 images x 4 tiles = 12 blocks x 8 column blocks... x 4
    x4 = torch.cat([x, x, x, x])                                 # 384 images: 384 workgroups
    auto = C.conv_fwd(geom, nhwc(x4).cuda(), wp)
    close(auto, C.conv_fwd(geom, nhwc(x4).cuda(), wp, tile_cfg=13), tol=TOL / 2)
    close(nchw(auto[:2]), F.conv2d(x[:2].double(), w.double(), padding=1), tol=TOL)
    g2, x2, w2, wp2 = make(2, 6, 10, 16, 24)                     # H, W not multiples of 4
    with pytest.raises(RuntimeError, match="tile_cfg 13"):
        C.conv_fwd(g2, nhwc(x2).cuda(), wp2, tile_cfg=13)


# ---- the pooled launches on this kernel (MODE 1 / 2 of conv_wino4.hip): 25 of the 36 products per 4x4 tile -----------------------
# B, H, W, Ci, Co with H, W multiples of 4 and Co a multiple of 64 (what pool_fused admits): ragged tile counts, 4x4 maps
POOL_CASES = [(4, 32, 32, 128, 128), (6, 16, 16, 64, 256), (3, 4, 12, 32, 192), (16, 8, 8, 256, 128), (2, 4, 4, 256, 128),
              (4, 32, 32, 64, 64), (5, 16, 16, 128, 64)]


@pytest.fixture
def force_pool():
    from diagan.ops import conv as C
    C.set_winograd4('force-pool')            # the pooled launches take the F(4x4) kernel at any launch size
    yield
    C.set_winograd4(None)


@pytest.mark.parametrize("case", POOL_CASES)
@pytest.mark.parametrize("pro", [0, 1])
def test_convolution_plus_average_pool_in_25_products(case, pro, force_pool):
    """F.avg_pool2d(conv3x3(pro(x)) + bias, 2) + residual -- the end of mimicry's DBlock / DBlockOptimized with downsample=True
    (predefined_models.py:38-40,76-78) -- in one launch, against float64 PyTorch; with the pair pass's two 1/sigma scalars."""
    from diagan import _native as nat
    from diagan.ops import conv as C
    B, H, W, Ci, Co = case
    geom, x, w, wp = make(*case, seed=21)
    assert nat.fn("diagan_conv_wino4_pool_used")(B, H, W, Ci, Co, C._splitk_ws(torch.device('cuda', 0)).numel())
    g = torch.Generator().manual_seed(22)
    bias, res = torch.randn(Co, generator=g), torch.randn(B, Co, H // 2, W // 2, generator=g)
    ref = F.avg_pool2d(F.conv2d(ref_pro(x.double(), pro, None, None), w.double(), bias.double(), padding=1), 2) + res.double()
    y = C.conv_fwd(geom, nhwc(x).cuda(), wp, bias=bias.cuda(), residual=nhwc(res).cuda(), pro=(pro, None, None), pool=True)
    assert tuple(y.shape) == (B, H // 2, W // 2, Co)
    close(nchw(y), ref, tol=TOL)
    if B % 2 == 0:
        s0, s1 = torch.tensor([0.7]).cuda(), torch.tensor([1.9]).cuda()
        y = C.conv_fwd(geom, nhwc(x).cuda(), wp, pro=(pro, None, None), row_scale=(s0, s1), pool=True)
        ref = F.avg_pool2d(F.conv2d(ref_pro(x.double(), pro, None, None), w.double(), padding=1), 2)
        ref[:B // 2] *= 0.7
        ref[B // 2:] *= 1.9
        close(nchw(y), ref, tol=TOL)


@pytest.mark.parametrize("case", POOL_CASES)
def test_data_gradient_through_the_average_pool_from_the_pooled_gradient(case, force_pool):
    """dx = conv^T(avg_pool2d_backward(g)) * relu'(mask) (+ residual) from the POOLED gradient -- the backward of the
    down-sampling DBlocks' c2 -- against float64 autograd of avg_pool2d(conv2d(x)); with the pair pass's two 1/sigma scalars."""
    from diagan.ops import conv as C
    B, H, W, Co, Ci = case                                  # (the data gradient's output channels are the layer's INPUT channels)
    if Ci % 8:
        pytest.skip("the data gradient gathers the layer's output channels: a multiple of 8")
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
    close(nchw(dx), ref, tol=TOL)
    if B % 2 == 0:
        s0, s1 = torch.tensor([0.7]).cuda(), torch.tensor([1.9]).cuda()
        dx = C.conv_dgrad(geom, nhwc(gp).cuda(), wd, (H, W), row_scale=(s0, s1), unpool=True)
        ref = xx.grad.clone()
        ref[:B // 2] *= 0.7
        ref[B // 2:] *= 1.9
        close(nchw(dx), ref, tol=TOL)


# ---- the convolution of an up-sampled input on this kernel (MODE 3 of conv_wino4.hip, tile_cfg 15) -----------------------------------
# B, Hl, Wl, Ci, Co: HALF-resolution input sizes; one tile per axis (first AND last tile: 2x2 -> 4x4), ragged tile counts,
# non-square maps, Co not a multiple of 64, the SNGAN generator's block shapes
UPIN_CASES = [(3, 2, 2, 16, 24), (2, 2, 6, 8, 4), (4, 4, 4, 64, 64), (5, 8, 8, 128, 72), (2, 16, 16, 256, 256),
              (2, 32, 32, 128, 64), (16, 4, 4, 512, 256)]


@pytest.fixture
def force_w4():
    from diagan.ops import conv as C
    C.set_winograd4('force-pool')            # the F(4x4) variants with a shape of their own take any launch size
    yield
    C.set_winograd4(None)


@pytest.mark.parametrize("case", UPIN_CASES)
@pytest.mark.parametrize("pro", [0, 1, 2, 3, 4])
def test_convolution_of_the_bilinear_upsampling_from_the_half_resolution_input(case, pro, force_w4):
    """conv3x3(F.interpolate(pro(x), scale_factor=2, mode='bilinear', align_corners=False)) + bias + residual -- the start of
    mimicry's GBlock residual branch, BN -> ReLU -> up-sampling -> c1 (GBlock._upsample_conv, predefined_models.py:19,57) -- as
    one launch on the low-resolution input, against float64 PyTorch: interior and border tiles (the interpolation clamps at the
    image edge, the convolution pads with zeros at the HIGH resolution), every prologue (applied BEFORE the interpolation)."""
    from diagan.ops import conv as C
    B, Hl, Wl, Ci, Co = case
    geom, x, w, wp = make(*case, seed=41)
    assert C.upin_fused(geom, B, Hl, Wl)
    g = torch.Generator().manual_seed(42)
    bias, scale, shift = torch.randn(Co, generator=g), torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.3
    up = F.interpolate(ref_pro(x.double(), pro, scale.double(), shift.double()), scale_factor=2, mode='bilinear',
                       align_corners=False)
    ref = F.conv2d(up, w.double(), bias.double(), padding=1)
    res = torch.randn(ref.shape, generator=g)
    y = C.conv_fwd(geom, nhwc(x).cuda(), wp, bias=bias.cuda(), residual=nhwc(res).cuda(),
                   pro=(pro, scale.cuda(), shift.cuda()), up_in=True)
    assert tuple(y.shape) == (B, 2 * Hl, 2 * Wl, Co)
    close(nchw(y), ref + res.double(), tol=TOL)


def test_upsampled_input_statistics_groups_and_the_two_launch_form(force_w4):
    """what GBlock.c1 uses in the stacked generator forward: BatchNorm statistics of the output from the epilogue and one
    BatchNorm row per stacked batch in the prologue; and the result against the two-launch form it replaces
    (diagan_upsample2x with the same prologue, then the F(4x4) kernel)"""
    from diagan.ops import conv as C
    from diagan.ops import eltwise as E
    B, Hl, Wl, Ci, Co = 8, 8, 8, 64, 128
    geom, x, w, wp = make(B, Hl, Wl, Ci, Co, seed=43)
    g = torch.Generator().manual_seed(44)
    xg = nhwc(x).cuda()
    groups = 4                                   # 2 images x 16 x 16 = 512 output rows per group: one workgroup tile
    sc, sh = torch.rand(groups, Ci, generator=g) + 0.5, torch.randn(groups, Ci, generator=g) * 0.3
    bias = torch.randn(Co, generator=g)
    pro = (C.PRO_AFFINE_RELU, sc.cuda(), sh.cuda(), B // groups)
    assert C.upin_fused(geom, B, Hl, Wl, group_imgs=B // groups)
    y, st = C.conv_fwd(geom, xg, wp, bias=bias.cuda(), pro=pro, up_in=True, want_stats=True)
    parts = [F.conv2d(F.interpolate(ref_pro(x[i * 2: i * 2 + 2].double(), 2, sc[i].double(), sh[i].double()), scale_factor=2,
                                    mode='bilinear', align_corners=False), w.double(), bias.double(), padding=1)
             for i in range(groups)]
    full = torch.cat(parts)
    close(nchw(y), full, tol=TOL)
    assert st[1] == B * 4 * Hl * Wl // 512 and st[0].shape == (st[1], 2, Co)
    close(st[0][:, 0].sum(0), full.sum((0, 2, 3)), tol=TOL)
    close(st[0][:, 1].sum(0), (full * full).sum((0, 2, 3)), tol=TOL)
    close(st[0][:, 0].sum(0), y.double().sum((0, 1, 2)), tol=1e-5)       # the sums are of the kernel's own outputs
    two = C.conv_fwd(geom, E.upsample2x(xg, pro=pro), wp, bias=bias.cuda(), tile_cfg=13)
    close(y, two, tol=TOL / 2)
    assert not C.upin_fused(geom, B, Hl, Wl, group_imgs=1)              # 256-row groups: a 512-row tile would straddle two


def test_upsampled_input_selection_and_refusals():
    """the automatic policy (no force): the stacked SNGAN generator launches qualify, small launches and the data-gradient
    geometry do not; DIAGAN_WINO4 off keeps it out; a backward mask is refused with the reason"""
    from diagan.ops import conv as C
    g256 = C.Geom("conv", 256, 256, 3, 3, 1, 1)
    assert C.upin_fused(g256, 384, 16, 16, group_imgs=64) and C.upin_fused(g256, 384, 4, 4, group_imgs=64)
    assert C.upin_fused(C.Geom("conv", 128, 64, 3, 3, 1, 1), 384, 32, 32, group_imgs=64)       # SNGAN-64 block5
    assert not C.upin_fused(g256, 64, 4, 4)                                                    # 32 workgroups
    assert not C.upin_fused(C.Geom("conv", 12, 64, 3, 3, 1, 1), 384, 16, 16)                   # Ci % 8
    C.set_winograd4(False)
    try:
        assert not C.upin_fused(g256, 384, 16, 16)
    finally:
        C.set_winograd4(None)
    geom, x, w, wp = make(2, 4, 4, 16, 16)
    with pytest.raises(RuntimeError, match="backward mask"):
        C._gemm(nhwc(x).cuda(), wp, torch.empty(2, 8, 8, 16, device='cuda'), geom.fwd_params(), 3, 3, geom.Kp, None, None,
                torch.zeros(2, 8, 8, 16, device='cuda'), 0.0, None, 1.0, 0, up_in=True)


# ---- X3 (round 5): the same kernel with its frequency GEMMs on the bf16 matrix pipe, operands split exactly in three ----------
# Off by default (profiles/r05_bf16x6_wino.md: correct, not faster); kept behind diagan_conv_gemm_set_wino4x / DIAGAN_WINO4_X3
# and held to float64 here so that the switch stays usable.  Only launches with Ci % 32 == 0 take it.
@pytest.fixture
def x3():
    from diagan.ops import conv as C
    C.set_winograd4x(True)
    yield
    C.set_winograd4x(None)


@pytest.mark.parametrize("case", [(4, 8, 8, 64, 64), (5, 16, 16, 128, 72), (2, 32, 32, 256, 256), (16, 4, 4, 512, 256)])
@pytest.mark.parametrize("pro", [0, 2, 3])
def test_x3_forward_and_data_gradient_against_float64(case, pro, x3):
    from diagan.ops import conv as C
    B, H, W, Ci, Co = case
    geom, x, w, wp = make(*case)
    g = torch.Generator().manual_seed(1)
    bias, scale, shift = torch.randn(Co, generator=g), torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.3
    ref = F.conv2d(ref_pro(x.double(), pro, scale.double(), shift.double()), w.double(), bias.double(), padding=1)
    res = torch.randn(ref.shape, generator=g)
    args = dict(bias=bias.cuda(), residual=nhwc(res).cuda(), pro=(pro, scale.cuda(), shift.cuda()), tile_cfg=13)
    y = C.conv_fwd(geom, nhwc(x).cuda(), wp, **args)
    close(nchw(y), ref + res.double(), tol=TOL)
    C.set_winograd4x(False)
    y32 = C.conv_fwd(geom, nhwc(x).cuda(), wp, **args)
    C.set_winograd4x(True)
    assert not torch.equal(y, y32), "the X3 switch changed nothing: the bf16 kernel did not run"
    if pro == 0 and Co % 32 == 0:            # the data gradient's K loop runs over Co
        dy = torch.randn(B, Co, H, W, generator=g)
        xr = x.double().requires_grad_(True)
        F.conv2d(xr, w.double(), padding=1).backward(dy.double())
        msk = torch.randn(B, Ci, H, W, generator=g)
        wd = torch.zeros(Ci, geom.Kd, device="cuda")
        C.pack_weights(wp, Co, Ci, 9, geom.Kp, geom.Kd, Wd=wd)
        dx = C.conv_dgrad(geom, nhwc(dy).cuda(), wd, (H, W), mask_src=nhwc(msk).cuda(), tile_cfg=13)
        close(nchw(dx), xr.grad * (msk > 0).double(), tol=TOL)


@pytest.mark.parametrize("case", [(4, 4, 4, 64, 64), (5, 8, 8, 128, 72), (2, 16, 16, 256, 256)])
def test_x3_upsampled_input_against_float64(case, x3, force_w4):
    from diagan.ops import conv as C
    B, Hl, Wl, Ci, Co = case
    geom, x, w, wp = make(*case, seed=41)
    g = torch.Generator().manual_seed(42)
    scale, shift = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.3
    up = F.interpolate(ref_pro(x.double(), 2, scale.double(), shift.double()), scale_factor=2, mode='bilinear', align_corners=False)
    ref = F.conv2d(up, w.double(), padding=1)
    y = C.conv_fwd(geom, nhwc(x).cuda(), wp, pro=(2, scale.cuda(), shift.cuda()), up_in=True)
    close(nchw(y), ref, tol=TOL)
