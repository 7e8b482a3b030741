"""Resampling kernels of csrc/elementwise.hip against plain PyTorch fp32 on the CPU: bilinear x2 (mimicry GBlock's
`F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)`, torch_mimicry/modules/resblocks.py:64-71) with the
fused BatchNorm + ReLU prologue (also per group of a stacked batch), its adjoint, and the 2x2 average pool of DBlock
(resblocks.py:161-168) with its adjoint."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("B,H,W,C", [(2, 1, 1, 4), (3, 4, 4, 8), (2, 5, 7, 12), (6, 8, 8, 64), (2, 16, 16, 256)])
@pytest.mark.parametrize("pro", ["none", "relu", "bn_relu", "bn_relu_groups"])
def test_upsample2x_forward(B, H, W, C, pro):
    from diagan.ops import eltwise as E
    from diagan.ops.conv import PRO_AFFINE_RELU, PRO_RELU
    g = torch.Generator().manual_seed(B * 100 + H + C)
    x = torch.randn(B, C, H, W, generator=g)
    if pro == "none":
        want, arg = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False), None
    elif pro == "relu":
        want, arg = F.interpolate(F.relu(x), scale_factor=2, mode='bilinear', align_corners=False), (PRO_RELU, None, None)
    else:
        groups = 1 if pro == "bn_relu" else (3 if B % 3 == 0 else 2)
        if B % groups:
            pytest.skip("batch does not split into the groups")
        scale, shift = torch.randn(groups, C, generator=g), torch.randn(groups, C, generator=g)
        per = B // groups
        idx = torch.arange(B) // per
        h = F.relu(x * scale[idx][:, :, None, None] + shift[idx][:, :, None, None])
        want = F.interpolate(h, scale_factor=2, mode='bilinear', align_corners=False)
        arg = (PRO_AFFINE_RELU, scale.reshape(-1).cuda(), shift.reshape(-1).cuda(), per if groups > 1 else 0)
    got = nchw(E.upsample2x(nhwc(x).cuda(), pro=arg).cpu())
    assert got.shape == want.shape
    assert (got - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("B,H,W,C", [(2, 1, 1, 4), (3, 4, 4, 8), (2, 5, 7, 12), (4, 16, 16, 128)])
def test_upsample2x_adjoint(B, H, W, C):
    from diagan.ops import eltwise as E
    g = torch.Generator().manual_seed(7 + H)
    x = torch.randn(B, C, H, W, generator=g, requires_grad=True)
    gy = torch.randn(B, C, 2 * H, 2 * W, generator=g)
    res = torch.randn(B, C, H, W, generator=g)
    F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False).backward(gy)
    got = nchw(E.upsample2x_bwd(nhwc(gy).cuda(), residual=nhwc(res).cuda()).cpu())
    want = x.grad + res
    assert (got - want).abs().max().item() <= 4e-6 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("B,H,W,C", [(2, 2, 2, 4), (3, 6, 10, 12), (4, 32, 32, 128)])
def test_avgpool2_and_adjoint(B, H, W, C):
    from diagan.ops import eltwise as E
    g = torch.Generator().manual_seed(11 + H)
    x = torch.randn(B, C, H, W, generator=g, requires_grad=True)
    res = torch.randn(B, C, H // 2, W // 2, generator=g)
    y = F.avg_pool2d(F.relu(x), 2)
    got = nchw(E.avgpool2(nhwc(x.detach()).cuda(), residual=nhwc(res).cuda(), relu_in=True).cpu())
    assert (got - (y + res)).abs().max().item() <= 2e-6 * max(1.0, y.abs().max().item())
    gy = torch.randn(B, C, H // 2, W // 2, generator=g)
    x2 = torch.randn(B, C, H, W, generator=g, requires_grad=True)
    F.avg_pool2d(x2, 2).backward(gy)
    got = nchw(E.avgpool2_bwd(nhwc(gy).cuda()).cpu())
    assert (got - x2.grad).abs().max().item() <= 1e-6 * max(1.0, x2.grad.abs().max().item())


@pytest.mark.parametrize("M,C", [(7, 4), (300, 12), (4096, 64), (65536, 256), (1000, 260)])
@pytest.mark.parametrize("relu", [False, True])
def test_batchnorm_backward_column_sums(M, C, relu):
    """dgamma / dbeta / dx of BatchNorm (+ the ReLU behind it) against torch autograd in float64 -- the column reduction
    `colred_kernel<1>` and `bn_bwd_apply_kernel` (mimicry GBlock: `self.activation(self.b1(h))`, resblocks.py:73-78)."""
    from diagan.ops import eltwise as E
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g, dtype=torch.float64, requires_grad=True)
    gamma = (torch.rand(C, generator=g, dtype=torch.float64) + 0.5).requires_grad_()
    beta = torch.randn(C, generator=g, dtype=torch.float64).requires_grad_()
    gy = torch.randn(M, C, generator=g, dtype=torch.float64)
    y = F.batch_norm(x, None, None, gamma, beta, True, 0.1, 1e-5)
    (F.relu(y) if relu else y).backward(gy)
    xc = x.detach().float().cuda().reshape(1, M, 1, C)
    rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    ctx = E.bn_stats(xc, gamma.detach().float().cuda(), beta.detach().float().cuda(), rm, rv, True)
    dgamma, dbeta = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    gx = E.bn_bwd(gy.float().cuda().reshape(1, M, 1, C), xc, ctx, relu, dgamma, dbeta, False)
    tol = 3e-5 if M > 8 else 1e-3                 # (M = 7: variance of seven samples, conditioning of 1/std)
    if relu:                                      # a pre-activation within fp32 rounding of zero may take the other branch
        tol = max(tol, 2e-4)
    for got, want in ((dgamma, gamma.grad), (dbeta, beta.grad)):
        assert (got.cpu() - want.float()).abs().max().item() <= tol * max(1.0, want.abs().max().item())
    far = (y.detach().abs() > 1e-4) if relu else torch.ones(M, C, dtype=torch.bool)
    err = ((gx.reshape(M, C).cpu() - x.grad.float()).abs() * far).max().item()
    assert err <= tol * max(1.0, x.grad.abs().max().item())


@pytest.mark.parametrize("B,H,C,groups", [(384, 4, 256, 6), (12, 3, 20, 3), (128, 8, 128, 2), (12, 1, 4, 6)])
def test_grouped_batchnorm_statistics_in_one_launch(B, H, C, groups):
    """Round 5: the statistics of `groups` stacked batches (the stacked generator forward's first BatchNorm reads the latent
    linear layer's output: no convolution epilogue to take them from) come from ONE column-reduction launch and ONE finalisation
    (diagan_bn_stats_grouped) instead of a launch pair per group: every group equals its own torch batch norm in float64, the
    running statistics equal `groups` successive forwards, and a single-group call gives the same statistics."""
    from diagan.ops import eltwise as E
    g = torch.Generator().manual_seed(B + C)
    x = torch.randn(B, H, H, C, generator=g) * 1.7 + 0.3
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    ctx = E.bn_stats(x.cuda(), gamma.cuda(), beta.cuda(), rm, rv, True, groups=groups)
    rm64, rv64 = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    per = B // groups
    for k in range(groups):
        xg = x[k * per:(k + 1) * per].double().reshape(-1, C)
        F.batch_norm(xg, rm64, rv64, None, None, True, 0.1, 1e-5)
        mean, var = xg.mean(0), xg.var(0, unbiased=False)
        assert (ctx.mean[k].cpu().double() - mean).abs().max().item() <= 1e-5 * max(1.0, mean.abs().max().item())
        inv = 1.0 / torch.sqrt(var + 1e-5)
        assert ((ctx.invstd[k].cpu().double() - inv).abs() / inv).max().item() <= 1e-5
        sc = gamma.double() * inv
        assert ((ctx.scale[k].cpu().double() - sc).abs() / sc.abs()).max().item() <= 2e-5
        one = E.bn_stats(x[k * per:(k + 1) * per].cuda().contiguous(), gamma.cuda(), beta.cuda(), torch.zeros(C, device='cuda'),
                         torch.ones(C, device='cuda'), True)
        assert (one.mean.reshape(-1) - ctx.mean[k]).abs().max().item() <= 1e-6 * max(1.0, mean.abs().max().item())
    assert (rm.cpu().double() - rm64).abs().max().item() <= 1e-5 * max(1.0, rm64.abs().max().item())
    assert ((rv.cpu().double() - rv64).abs() / rv64).max().item() <= 1e-5
