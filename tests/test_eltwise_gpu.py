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
