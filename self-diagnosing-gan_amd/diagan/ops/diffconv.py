"""Convolution as a twice-(any-order-)differentiable torch op over the HIP implicit-GEMM kernels.

The StyleGAN2 row differentiates THROUGH a gradient: R1 takes d/d(theta_D) of
|d D(x)/d x|^2 and the path-length penalty d/d(theta_G) of |d G(w)/d w| (stylegan2/train_ffhq.py:74-102).  The
reference gets this from cuDNN's conv2d / conv_transpose2d autograd; here the three bilinear maps of a convolution

    C(x, w) = conv(x, w)             forward                 diagan_conv_gemm, forward gather
    D(g, w) = C^T_x(g, w)            data gradient           diagan_conv_gemm, adjoint gather
    G(g, x) = C^T_w(g, x)            weight gradient         diagan_conv_wgrad

are closed under differentiation (each one's backward is two of the others), so three autograd Functions whose
backwards call each other give every order.  Activations are NHWC fp32, weights the packed operand Wp[Co][Kp]
(diagan/ops/conv.py); `pack` below turns a reference-shaped OIHW parameter into Wp differentiably."""
import torch
import torch.nn.functional as F
from torch.autograd import Function

from diagan.ops import conv as K


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _wd(geom, wp):
    """data-gradient operand Wd[Ci][Kd] of packed forward weights"""
    wd = torch.zeros((geom.Ci, geom.Kd), dtype=torch.float32, device=wp.device)
    K.pack_weights(_c(wp), geom.Co, geom.Ci, geom.R * geom.S, geom.Kp, geom.Kd, Wd=wd)
    return wd


class _Conv(Function):
    @staticmethod
    def forward(ctx, x, wp, geom):
        ctx.geom = geom
        ctx.save_for_backward(x, wp)
        return K.conv_fwd(geom, _c(x), _c(wp))

    @staticmethod
    def backward(ctx, gy):
        x, wp = ctx.saved_tensors
        gx = _DataGrad.apply(gy, wp, ctx.geom, tuple(x.shape[1:3])) if ctx.needs_input_grad[0] else None
        gw = _WeightGrad.apply(gy, x, ctx.geom) if ctx.needs_input_grad[1] else None
        return gx, gw, None


class _DataGrad(Function):
    @staticmethod
    def forward(ctx, g, wp, geom, in_hw):
        ctx.geom, ctx.in_hw = geom, in_hw
        ctx.save_for_backward(g, wp)
        return K.conv_dgrad(geom, _c(g), _wd(geom, wp), in_hw)

    @staticmethod
    def backward(ctx, ggx):
        g, wp = ctx.saved_tensors
        gg = _Conv.apply(ggx, wp, ctx.geom) if ctx.needs_input_grad[0] else None
        gw = _WeightGrad.apply(g, ggx, ctx.geom) if ctx.needs_input_grad[1] else None
        return gg, gw, None, None


class _WeightGrad(Function):
    @staticmethod
    def forward(ctx, g, x, geom):
        ctx.geom = geom
        ctx.save_for_backward(g, x)
        grad = torch.empty((geom.Co, geom.Kp), dtype=torch.float32, device=g.device)
        return K.conv_wgrad(geom, _c(g), _c(x), grad, accumulate=False)

    @staticmethod
    def backward(ctx, ggw):
        g, x = ctx.saved_tensors
        gg = _Conv.apply(x, ggw, ctx.geom) if ctx.needs_input_grad[0] else None
        gx = _DataGrad.apply(g, ggw, ctx.geom, tuple(x.shape[1:3])) if ctx.needs_input_grad[1] else None
        return gg, gx, None


def pack(w_oihw, geom):
    """[Co', Ci', R, S] parameter (Co' <= geom.Co, Ci' <= geom.Ci: zero-padded) -> Wp[Co][Kp]; differentiable."""
    co, ci, r, s = w_oihw.shape
    w = w_oihw.permute(0, 2, 3, 1)
    if ci != geom.Ci or co != geom.Co:
        w = F.pad(w, (0, geom.Ci - ci, 0, 0, 0, 0, 0, geom.Co - co))
    w = w.reshape(geom.Co, r * s * geom.Ci)
    if w.shape[1] != geom.Kp:
        w = F.pad(w, (0, geom.Kp - w.shape[1]))
    return w.contiguous()


def conv2d(x, w_oihw, stride=1, padding=0):
    """F.conv2d on NHWC activations: x [B,H,W,Ci] (Ci % 4 == 0), w [Co', Ci', R, S] -> [B,Ho,Wo,roundup(Co',4)]"""
    geom = K.Geom('conv', x.shape[3], K.round_up(w_oihw.shape[0], 4), w_oihw.shape[2], w_oihw.shape[3], stride, padding)
    return _Conv.apply(x, pack(w_oihw, geom), geom)


def conv_transpose2d(x, w_oihw, stride=2, padding=0):
    """F.conv_transpose2d(x, w.transpose(0, 1)) on NHWC activations: w is given output-channel-major like conv2d's
    (the modulated convolution of the reference transposes it itself, stylegan2.py:243-248)."""
    geom = K.Geom('convT', x.shape[3], K.round_up(w_oihw.shape[0], 4), w_oihw.shape[2], w_oihw.shape[3], stride, padding)
    return _Conv.apply(x, pack(w_oihw, geom), geom)


def linear(x, weight):
    """F.linear(x, weight) (no bias) as a 1x1 convolution over [B,1,1,Ci]"""
    b, ci = x.shape
    cp = K.round_up(ci, 4)
    if cp != ci:
        x = F.pad(x, (0, cp - ci))
    co = weight.shape[0]
    y = conv2d(x.view(b, 1, 1, cp), weight.view(co, ci, 1, 1))
    return y.view(b, -1)[:, :co]
