"""GPU: the experimental bf16x6 mode of the GEMM kernels (diagan_set_mfma_mode(1)) -- every fp32 operand split exactly
into three bf16 pieces, six exact piece products per element accumulated in fp32 on the bf16 matrix pipe.  Its error
against float64 must be at (or below) the level of the exact-fp32 MFMA kernels, for the forward / data-gradient GEMM
and for the weight gradient, including the fused prologues; a training step in this mode meets the same parity bar."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture
def modes():
    from diagan.ops import conv as C
    start = C.get_mfma_mode()
    yield C.set_mfma_mode
    C.set_mfma_mode(start)


def rel(a, ref):
    return float((a.cpu().double() - ref).abs().mean() / ref.abs().mean())


@pytest.mark.parametrize("case", [
    # B, H, Ci, Co, k, stride, tile, prologue
    (4, 16, 64, 128, 3, 1, 1, None), (4, 16, 64, 128, 3, 1, 3, None), (2, 8, 256, 256, 3, 1, 1, 'bn_relu'),
    (8, 8, 32, 64, 1, 1, 3, 'relu'), (2, 17, 32, 48, 3, 2, 0, None), (3, 9, 4, 32, 3, 1, 0, None)])
def test_forward_gemm_error_at_fp32_level(modes, case):
    from diagan.ops import conv as C
    B, H, Ci, Co, k, stride, cfg, pro = case
    g = torch.Generator().manual_seed(B * H + Ci)
    x = torch.randn(B, H, H, Ci, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    xa, prot = x, None
    if pro == 'bn_relu':
        sc, sh = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.1
        xa, prot = torch.relu(x * sc + sh), (C.PRO_AFFINE_RELU, sc.cuda(), sh.cuda())
    elif pro == 'relu':
        xa, prot = torch.relu(x), (C.PRO_RELU, None, None)
    ref = F.conv2d(xa.permute(0, 3, 1, 2).double(), w.double(), stride=stride, padding=k // 2).permute(0, 2, 3, 1)
    geom = C.Geom('conv', Ci, Co, k, k, stride, k // 2)
    wp = C.pack_oihw(w, geom.Kp).cuda()
    err = []
    for m in (0, 1):
        modes(m)
        err.append(rel(C.conv_fwd(geom, x.cuda(), wp, pro=prot, tile_cfg=cfg), ref))
    assert err[1] < 2e-6 and err[1] < 1.5 * err[0] + 1e-7, err


@pytest.mark.parametrize("case", [(4, 16, 128, 128, 3, None), (2, 8, 256, 256, 3, 'bn_relu'), (3, 12, 128, 256, 3, None),
                                  (4, 16, 128, 128, 1, 'relu'), (2, 11, 128, 160, 3, None),
                                  (4, 16, 64, 64, 3, 'bn_relu'), (2, 8, 128, 48, 3, None), (3, 10, 32, 64, 3, 'relu')])
def test_weight_gradient_error_at_fp32_level(modes, case):
    """the transposing-LDS-read kernels (dy tiles of 128 and of 64 columns); 11x11 / 10x10 images take the
    non-power-of-two coordinate path; 48 output channels leave dy columns past Co masked"""
    from diagan.ops import conv as C
    B, H, Ci, Co, k, pro = case
    g = torch.Generator().manual_seed(H + Co)
    x, dy = torch.randn(B, H, H, Ci, generator=g), torch.randn(B, H, H, Co, generator=g)
    geom = C.Geom('conv', Ci, Co, k, k, 1, k // 2)
    xa, prot = x, None
    if pro == 'bn_relu':
        sc, sh = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.1
        xa, prot = torch.relu(x * sc + sh), (C.PRO_AFFINE_RELU, sc.cuda(), sh.cuda())
    elif pro == 'relu':
        xa, prot = torch.relu(x), (C.PRO_RELU, None, None)
    wref = torch.zeros(Co, Ci, k, k, dtype=torch.float64, requires_grad=True)
    (F.conv2d(xa.permute(0, 3, 1, 2).double(), wref, padding=k // 2) * dy.permute(0, 3, 1, 2).double()).sum().backward()
    ref = C.pack_oihw(wref.grad, geom.Kp)
    err = []
    for m in (0, 1):
        modes(m)
        grad = torch.zeros(Co, geom.Kp, device="cuda")
        C.conv_wgrad(geom, dy.cuda(), x.cuda(), grad, accumulate=False, pro=prot)
        err.append(rel(grad, ref))
    assert err[1] < 2e-6 and err[1] < 1.5 * err[0] + 1e-7, err


def test_training_step_parity_in_bf16x6_mode(modes):
    """one D and one G update of SNGAN-32 in this mode against the fp32 mode from the same state: losses to 1e-5,
    updated parameters to a relative L2 of 1e-4 (Adam's first step is a sign step: tiny gradients may flip)"""
    import bench

    from diagan.ops import conv as C

    def run(mode):
        modes(mode)
        C.set_winograd(False)        # like with like: the fp32 leg on the implicit GEMM, as this mode's kernels are
        dev = torch.device("cuda", 0)
        netG, netD, _, optG, optD, _ = bench.build_models('cifar10', 'ns', 1, dev)
        gen = torch.Generator().manual_seed(5)
        x = (torch.rand(16, 3, 32, 32, generator=gen) * 2 - 1).to(dev)
        z = torch.randn(16, 128, generator=gen).to(dev)

        class Log:
            def __init__(self): self.m = {}
            def add_metric(self, k, v, **kw): self.m[k] = float(v)
        log = Log()
        netD.train_step(real_batch=(x, None), netG=netG, optD=optD, log_data=log, device=dev, noise=z)
        netG.train_step(real_batch=(x, None), netD=netD, optG=optG, log_data=log, device=dev, noise=z)
        torch.cuda.synchronize()
        return log.m, netD.flat_params.clone(), netG.flat_params.clone()

    try:
        m0, d0, g0 = run(0)
        m1, d1, g1 = run(1)
    finally:
        C.set_winograd(None)
    for k in ('errD', 'errG'):
        assert abs(m0[k] - m1[k]) < 1e-5 * max(1.0, abs(m0[k])), (k, m0[k], m1[k])
    for a, b in ((d0, d1), (g0, g1)):
        assert float((a - b).double().norm() / a.double().norm()) < 1e-4
