"""upsample / FIR filter / downsample (reference: diagan-pkg/diagan/models/op/upfirdn2d.py:19-156).

`upfirdn2d_op(input[major,H,W,minor], kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)`
is the reference's pybind entry (upfirdn2d.cpp:4-22) on HIP.

Autograd: upfirdn2d is a LINEAR map of its input, and its adjoint is again an upfirdn2d -- flipped kernel, up and
down exchanged, pads (kw - px0 - 1, in_w*up - out_w*down + px0 - up + 1, same in y).  So one autograd Function
parameterised by a `_Plan` serves every order of differentiation: its backward applies the Function with the dual
plan (R1 / path-length regularisation of the reference need the second order)."""
import ctypes

import torch
from torch.autograd import Function

from diagan import _native as nat

P, I = nat.c_void_p, nat.c_int
nat.register("diagan_upfirdn2d", [P, P, P] + [I] * 14 + [P, P, P])


def upfirdn2d_op(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    if not input.is_cuda:
        raise RuntimeError("upfirdn2d: input must be a CUDA tensor")              # CHECK_CUDA of the reference
    x = input.contiguous().float()
    k = kernel.contiguous().float().to(x.device)
    major, in_h, in_w, minor = x.shape
    kh, kw = k.shape
    oh, ow = ctypes.c_int(0), ctypes.c_int(0)
    args = [major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1]
    nat.call("diagan_upfirdn2d", None, None, None, *args, ctypes.byref(oh), ctypes.byref(ow), None)   # size query
    out = torch.empty((major, oh.value, ow.value, minor), dtype=torch.float32, device=x.device)
    nat.call("diagan_upfirdn2d", nat.ptr(x), nat.ptr(k), nat.ptr(out), *args, None, None, nat.current_stream())
    return out.to(input.dtype)


_FLIPS = {}


def _flipped(kernel):
    """torch.flip(kernel, [0, 1]) of a FIR kernel, made once per kernel tensor and content version (every FIR call builds its plan and
    the plan's adjoint: the flip was a launch per call, ~65 per StyleGAN2 iteration); kernels that require grad are never cached"""
    if kernel.requires_grad:
        return torch.flip(kernel, [0, 1])
    key = (kernel.data_ptr(), kernel._version, tuple(kernel.shape), kernel.device)
    f = _FLIPS.get(key)
    if f is None:
        if len(_FLIPS) > 64:
            _FLIPS.clear()
        f = _FLIPS[key] = (torch.flip(kernel, [0, 1]), kernel)          # (the source is kept alive: its data_ptr cannot be recycled)
    return f[0]


class _Plan:
    """Geometry of one upfirdn2d application on [batch, channel, in_h, in_w] images, linked to its adjoint."""

    def __init__(self, kernel, up, down, pad, in_hw, out_hw, dual=None, channels_last=False):
        self.kernel, self.up, self.down, self.pad = kernel, tuple(up), tuple(down), tuple(pad)
        self.in_hw, self.out_hw = tuple(in_hw), tuple(out_hw)
        self.dual = dual
        self.channels_last = channels_last      # images are [batch, in_h, in_w, channel]: the native op's own layout

    @classmethod
    def forward_plan(cls, kernel, up, down, pad, in_hw, channels_last=False):
        (ux, uy), (dx, dy), (px0, px1, py0, py1) = up, down, pad
        kh, kw = kernel.shape
        in_h, in_w = in_hw
        out_h = (in_h * uy + py0 + py1 - kh) // dy + 1
        out_w = (in_w * ux + px0 + px1 - kw) // dx + 1
        plan = cls(kernel, up, down, pad, in_hw, (out_h, out_w), channels_last=channels_last)
        adjoint_pad = (kw - px0 - 1, in_w * ux - out_w * dx + px0 - ux + 1,
                       kh - py0 - 1, in_h * uy - out_h * dy + py0 - uy + 1)
        plan.dual = cls(_flipped(kernel), down, up, adjoint_pad, (out_h, out_w), in_hw, dual=plan,
                        channels_last=channels_last)
        return plan

    def run(self, images):
        if self.channels_last:
            return upfirdn2d_op(images, self.kernel, self.up[0], self.up[1], self.down[0], self.down[1], *self.pad)
        b, c = images.shape[:2]
        flat = images.reshape(-1, self.in_hw[0], self.in_hw[1], 1)
        out = upfirdn2d_op(flat, self.kernel, self.up[0], self.up[1], self.down[0], self.down[1], *self.pad)
        return out.view(b, c, self.out_hw[0], self.out_hw[1])


class _LinearFIR(Function):
    @staticmethod
    def forward(ctx, images, plan):
        ctx.plan = plan
        return plan.run(images)

    @staticmethod
    def backward(ctx, grad):
        return _LinearFIR.apply(grad.contiguous(), ctx.plan.dual), None


class UpFirDn2d:
    """Call-compatible with the reference's autograd Function: `UpFirDn2d.apply(input, kernel, up, down, pad)` with
    up = (up_x, up_y), down = (down_x, down_y), pad = (pad_x0, pad_x1, pad_y0, pad_y1)."""

    @staticmethod
    def apply(input, kernel, up, down, pad):
        return _LinearFIR.apply(input, _Plan.forward_plan(kernel, up, down, pad, input.shape[2:]))


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    return UpFirDn2d.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))


def upfirdn2d_nhwc(input, kernel, up=1, down=1, pad=(0, 0)):
    """Same filter on [batch, H, W, channel] activations (the layout of this engine's convolutions)."""
    plan = _Plan.forward_plan(kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]), input.shape[1:3],
                              channels_last=True)
    return _LinearFIR.apply(input, plan)
