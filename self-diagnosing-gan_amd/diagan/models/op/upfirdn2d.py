"""upsample / FIR filter / downsample (reference: diagan-pkg/diagan/models/op/upfirdn2d.py:19-156).

`upfirdn2d_op(input[major,H,W,minor], kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)`
is the reference's pybind entry (upfirdn2d.cpp:4-22) on HIP; the autograd wrapper mirrors the
reference's: backward = the same op with the flipped kernel and up <-> down swapped, and it is itself
differentiable (R1 / path-length regularisation need the second order)."""
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


class UpFirDn2dBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, kernel, grad_kernel, up, down, pad, g_pad, in_size, out_size):
        up_x, up_y = up
        down_x, down_y = down
        g_pad_x0, g_pad_x1, g_pad_y0, g_pad_y1 = g_pad
        grad_output = grad_output.reshape(-1, out_size[0], out_size[1], 1)
        grad_input = upfirdn2d_op(grad_output, grad_kernel, down_x, down_y, up_x, up_y, g_pad_x0, g_pad_x1, g_pad_y0,
                                  g_pad_y1)
        grad_input = grad_input.view(in_size[0], in_size[1], in_size[2], in_size[3])
        ctx.save_for_backward(kernel)
        ctx.up, ctx.down, ctx.pad, ctx.in_size, ctx.out_size = up, down, pad, in_size, out_size
        return grad_input

    @staticmethod
    def backward(ctx, gradgrad_input):
        kernel, = ctx.saved_tensors
        gradgrad_input = gradgrad_input.reshape(-1, ctx.in_size[2], ctx.in_size[3], 1)
        out = upfirdn2d_op(gradgrad_input, kernel, ctx.up[0], ctx.up[1], ctx.down[0], ctx.down[1], *ctx.pad)
        out = out.view(ctx.in_size[0], ctx.in_size[1], ctx.out_size[0], ctx.out_size[1])
        return out, None, None, None, None, None, None, None, None


class UpFirDn2d(Function):
    @staticmethod
    def forward(ctx, input, kernel, up, down, pad):
        up_x, up_y = up
        down_x, down_y = down
        pad_x0, pad_x1, pad_y0, pad_y1 = pad
        kernel_h, kernel_w = kernel.shape
        batch, channel, in_h, in_w = input.shape
        ctx.in_size = input.shape
        input = input.reshape(-1, in_h, in_w, 1)
        ctx.save_for_backward(kernel, torch.flip(kernel, [0, 1]))
        out_h = (in_h * up_y + pad_y0 + pad_y1 - kernel_h) // down_y + 1
        out_w = (in_w * up_x + pad_x0 + pad_x1 - kernel_w) // down_x + 1
        ctx.out_size = (out_h, out_w)
        ctx.up, ctx.down, ctx.pad = (up_x, up_y), (down_x, down_y), (pad_x0, pad_x1, pad_y0, pad_y1)
        ctx.g_pad = (kernel_w - pad_x0 - 1, in_w * up_x - out_w * down_x + pad_x0 - up_x + 1,
                     kernel_h - pad_y0 - 1, in_h * up_y - out_h * down_y + pad_y0 - up_y + 1)
        out = upfirdn2d_op(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
        return out.view(-1, channel, out_h, out_w)

    @staticmethod
    def backward(ctx, grad_output):
        kernel, grad_kernel = ctx.saved_tensors
        grad_input = UpFirDn2dBackward.apply(grad_output, kernel, grad_kernel, ctx.up, ctx.down, ctx.pad, ctx.g_pad,
                                             ctx.in_size, ctx.out_size)
        return grad_input, None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    return UpFirDn2d.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
