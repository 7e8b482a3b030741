"""fused bias + leaky ReLU + gain (reference: diagan-pkg/diagan/models/op/fused_act.py:22-118).

Same operator contract as the reference's pybind `fused.fused_bias_act(input, bias, refer, act, grad,
alpha, scale)` (fused_bias_act.cpp:4-20): bias is broadcast along dim 1, act 3 = leaky ReLU, grad 1 =
derivative gated by `refer > 0`.  Unlike the reference's CPU fallback (which hard-codes slope 0.2,
fused_act.py:106-118) the device op honours `negative_slope`."""
import torch
from torch import nn
from torch.autograd import Function

from diagan import _native as nat

P, I, F32, I64 = nat.c_void_p, nat.c_int, nat.c_f32, nat.c_i64
nat.register("diagan_fused_bias_act", [P, P, P, P, I64, I64, I, I, I, F32, F32, P])


def fused_bias_act(input, bias, refer, act, grad, alpha, scale):
    """The reference's native entry point, on HIP.  Empty `bias` / `refer` tensors mean "absent"."""
    if not input.is_cuda:
        raise RuntimeError("fused_bias_act: input must be a CUDA tensor")        # CHECK_CUDA of the reference
    x = input.contiguous().float()
    b = bias.contiguous().float() if bias is not None and bias.numel() else None
    r = refer.contiguous().float() if refer is not None and refer.numel() else None
    if b is not None and not b.is_cuda:
        raise RuntimeError("fused_bias_act: bias must be a CUDA tensor")
    step_b = 1
    for i in range(2, x.dim()):
        step_b *= x.size(i)
    out = torch.empty_like(x)
    nat.call("diagan_fused_bias_act", nat.ptr(x), nat.ptr(b), nat.ptr(r), nat.ptr(out), x.numel(), step_b,
             b.numel() if b is not None else 1, act, grad, float(alpha), float(scale), nat.current_stream())
    return out.to(input.dtype)


class FusedLeakyReLUFunctionBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, out, bias, negative_slope, scale):
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        empty = grad_output.new_empty(0)
        grad_input = fused_bias_act(grad_output, empty, out, 3, 1, negative_slope, scale)
        dim = [0] + list(range(2, grad_input.ndim))
        grad_bias = grad_input.sum(dim).detach() if bias else empty
        return grad_input, grad_bias

    @staticmethod
    def backward(ctx, gradgrad_input, gradgrad_bias):
        out, = ctx.saved_tensors
        gradgrad_out = fused_bias_act(gradgrad_input, gradgrad_bias, out, 3, 1, ctx.negative_slope, ctx.scale)
        return gradgrad_out, None, None, None, None


class FusedLeakyReLUFunction(Function):
    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        empty = input.new_empty(0)
        ctx.bias = bias is not None
        out = fused_bias_act(input, bias if bias is not None else empty, empty, 3, 0, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        return out

    @staticmethod
    def backward(ctx, grad_output):
        out, = ctx.saved_tensors
        grad_input, grad_bias = FusedLeakyReLUFunctionBackward.apply(grad_output, out, ctx.bias, ctx.negative_slope,
                                                                     ctx.scale)
        return grad_input, (grad_bias if ctx.bias else None), None, None


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, bias=True, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel)) if bias else None
        self.negative_slope, self.scale = negative_slope, scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


def fused_leaky_relu(input, bias=None, negative_slope=0.2, scale=2 ** 0.5):
    return FusedLeakyReLUFunction.apply(input, bias, negative_slope, scale)
