"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's two StyleGAN2 native ops.

Follows the reference's own CPU statements: `upfirdn2d_native`
(/root/reference/diagan-pkg/diagan/models/op/upfirdn2d.py:159-200) and the kernel's act/grad table
(op/fused_bias_act_kernel.cu:35-46).  PINNED: tests/golden/stylegan_ops.npz was produced by the
reference functions themselves (tools/gen_goldens_stylegan_ops.py) and this file reproduces it."""
import torch
import torch.nn.functional as F


def upfirdn2d(x, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """x [B,C,H,W]: zero-insertion upsample, pad/crop, true convolution with `kernel`, decimate."""
    B, C, H, W = x.shape
    kh, kw = kernel.shape
    u = x.new_zeros(B * C, 1, H * up_y, W * up_x)
    u[:, :, ::up_y, ::up_x] = x.reshape(B * C, 1, H, W)                       # upfirdn2d.py:169-171
    u = F.pad(u, [max(pad_x0, 0), max(pad_x1, 0), max(pad_y0, 0), max(pad_y1, 0)])
    u = u[:, :, max(-pad_y0, 0): u.shape[2] - max(-pad_y1, 0), max(-pad_x0, 0): u.shape[3] - max(-pad_x1, 0)]
    w = torch.flip(kernel, [0, 1]).view(1, 1, kh, kw)                        # :186
    y = F.conv2d(u, w)[:, :, ::down_y, ::down_x]                             # :187-195
    return y.reshape(B, C, y.shape[2], y.shape[3])


def fused_bias_act(x, bias, ref, act, grad, alpha, scale):
    """fused_bias_act_kernel.cu:24-48"""
    if bias is not None and bias.numel():
        x = x + bias.view(1, -1, *([1] * (x.dim() - 2)))
    mode = act * 10 + grad
    if mode in (12, 32):
        y = torch.zeros_like(x)
    elif mode == 30:
        y = torch.where(x > 0, x, x * alpha)
    elif mode == 31:
        y = torch.where(ref > 0, x, x * alpha)
    else:
        y = x
    return y * scale


def fused_leaky_relu(x, bias=None, negative_slope=0.2, scale=2 ** 0.5):
    return fused_bias_act(x, bias, None, 3, 0, negative_slope, scale)
