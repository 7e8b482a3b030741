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
nat.register("diagan_rowdot_chunks", [I, I])
nat.register("diagan_rowdot", [P, P, P, P, I, I, I, P])
nat.register("diagan_styled_bias_act", [P, P, P, P, P, P, I, I, I, I, F32, F32, P])
nat.register("diagan_styled_bias_act_bwd", [P, P, P, P, P, P, P, P, P, I, I, I, I, F32, F32, P])
nat.register("diagan_styled_bias_act_bwd_finish", [P, P, P, P, P, P, I, I, I, P])

import os as _os
FUSED_BWD = _os.environ.get("DIAGAN_SG2_FUSED_BWD", "1") != "0"


def _fused_bwd_ok(gy):
    """the one-pass first-order backward applies: no graph is being recorded for this backward (R1 / path-length penalties
    differentiate THROUGH the backward and take the differentiable composition), channels-last 4-D gradient, C a power of
    two in [4, 1024]"""
    c = gy.shape[-1]
    return (FUSED_BWD and not torch.is_grad_enabled() and gy.dim() == 4 and gy.is_cuda and gy.dtype == torch.float32
            and 4 <= c <= 1024 and not (c & (c - 1)))


def _fused_bwd(gy, y, x, demod, noise, slope, scale, need_gx=True):
    """(gx, d(demod) [B,C] or None, d(bias) [C], d(strength) [1] or None) from one pass over gy (diagan_styled_bias_act_bwd)"""
    b, h, w, c = gy.shape
    gy, y = gy.contiguous(), y.contiguous()
    chunks = nat.fn("diagan_rowdot_chunks")(b, h * w)
    f32 = dict(dtype=torch.float32, device=gy.device)
    gx = torch.empty_like(gy) if need_gx else None
    wd = torch.empty((b, chunks, c), **f32) if x is not None else None
    wb = torch.empty((b * chunks, c), **f32)
    per_image = noise is not None and noise.shape[0] == b and b > 1
    ws = torch.empty(b * chunks, **f32) if noise is not None else None
    nat.call("diagan_styled_bias_act_bwd", nat.ptr(gy), nat.ptr(y), nat.ptr(x.contiguous()) if x is not None else None,
             nat.ptr(demod.contiguous()) if demod is not None else None,
             nat.ptr(noise.contiguous()) if noise is not None else None, nat.ptr(gx), nat.ptr(wd), nat.ptr(wb), nat.ptr(ws),
             b, h * w, c, 1 if per_image else 0, float(slope), float(scale), nat.current_stream())
    # the partial sums (rows of a few dozen blocks) to their results, in double, ONE launch (was nine torch launches per call)
    gd = torch.empty((b, c), **f32) if wd is not None else None
    gb = torch.empty(c, **f32)
    gs = torch.empty(1, **f32) if ws is not None else None
    nat.call("diagan_styled_bias_act_bwd_finish", nat.ptr(wd), nat.ptr(wb), nat.ptr(ws), nat.ptr(gd), nat.ptr(gb), nat.ptr(gs),
             b, h * w, c, nat.current_stream())
    return gx, gd, gb, gs



def fused_bias_act(input, bias, refer, act, grad, alpha, scale, bias_dim=1):
    """The reference's native entry point, on HIP.  Empty `bias` / `refer` tensors mean "absent".
    `bias_dim` (extension): the dimension the bias runs along; the reference fixes 1, channels-last callers pass -1."""
    if not input.is_cuda:
        raise RuntimeError("fused_bias_act: input must be a CUDA tensor")        # CHECK_CUDA of the reference
    x = input.contiguous().float()
    b = bias.contiguous().float() if bias is not None and bias.numel() else None
    r = refer.contiguous().float() if refer is not None and refer.numel() else None
    if b is not None and not b.is_cuda:
        raise RuntimeError("fused_bias_act: bias must be a CUDA tensor")
    step_b = 1
    for i in range(bias_dim % x.dim() + 1, x.dim()):
        step_b *= x.size(i)
    out = torch.empty_like(x)
    nat.call("diagan_fused_bias_act", nat.ptr(x), nat.ptr(b), nat.ptr(r), nat.ptr(out), x.numel(), step_b,
             b.numel() if b is not None else 1, act, grad, float(alpha), float(scale), nat.current_stream())
    return out.to(input.dtype)


class _LeakyGate(Function):
    """g -> g * scale * (1 where ref > 0 else slope): the derivative of the fused activation with respect to its
    pre-activation, as ONE launch of the native op (act 3, grad 1, gated by `ref`).  The map is linear in g, so its own
    backward is the same map: the class differentiates itself to any order (the reference needs second order for
    the R1 / path-length penalties, stylegan2/train_ffhq.py:74-102)."""

    @staticmethod
    def forward(ctx, g, ref, slope, scale):
        ctx.save_for_backward(ref)
        ctx.hyper = (slope, scale)
        return fused_bias_act(g, None, ref, 3, 1, slope, scale)

    @staticmethod
    def backward(ctx, gg):
        ref, = ctx.saved_tensors
        return _LeakyGate.apply(gg, ref, *ctx.hyper), None, None, None


class _BiasLeakyReLU(Function):
    """y = leaky_relu(x + bias[channel]) * scale in one launch; sign(y) == sign(x + bias), so y itself gates the
    backward and the input is not kept."""

    @staticmethod
    def forward(ctx, x, bias, slope, scale, bias_dim=1):
        y = fused_bias_act(x, bias, None, 3, 0, slope, scale, bias_dim)
        ctx.save_for_backward(y)
        ctx.hyper = (slope, scale)
        ctx.has_bias, ctx.bias_dim = bias is not None, bias_dim % x.dim()
        return y

    @staticmethod
    def backward(ctx, gy):
        y, = ctx.saved_tensors
        if ctx.has_bias and ctx.bias_dim == gy.dim() - 1 and _fused_bwd_ok(gy):
            gx, _, gb, _ = _fused_bwd(gy, y, None, None, None, *ctx.hyper)      # gate + bias gradient in one pass
            return gx, gb, None, None, None
        gx = _LeakyGate.apply(gy, y, *ctx.hyper)
        gb = gx.sum([d for d in range(gx.dim()) if d != ctx.bias_dim]) if ctx.has_bias else None
        return gx, gb, None, None, None


class _RowDot(Function):
    """R(a, b)[n, c] = sum over pixels of a[n, h, w, c] * b[n, h, w, c]: one pass over both tensors.  Together with
    S(x, s) = x * s[:, None, None, :] it is closed under differentiation (dR/da = S(b, g), dS/ds = R(g, x))."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        n, h, w, c = a.shape
        if c < 4 or c > 1024 or c & (c - 1):
            return (a * b).sum((1, 2))
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty((n, c), dtype=torch.float32, device=a.device)
        work = torch.empty(n * nat.fn("diagan_rowdot_chunks")(n, h * w) * c, dtype=torch.float32, device=a.device)
        nat.call("diagan_rowdot", nat.ptr(a), nat.ptr(b), nat.ptr(out), nat.ptr(work), n, h * w, c, nat.current_stream())
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return (scale_rows(b, g) if ctx.needs_input_grad[0] else None,
                scale_rows(a, g) if ctx.needs_input_grad[1] else None)


class _ScaleRows(Function):
    @staticmethod
    def forward(ctx, x, s):
        ctx.save_for_backward(x, s)
        return x * s[:, None, None, :]

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        return (scale_rows(g, s) if ctx.needs_input_grad[0] else None,
                rowdot(g, x) if ctx.needs_input_grad[1] else None)


def rowdot(a, b):
    return _RowDot.apply(a, b)


def scale_rows(x, s):
    """x[n, h, w, c] * s[n, c] whose gradient with respect to s is the one-pass `rowdot`"""
    return _ScaleRows.apply(x, s)


class _StyledAct(Function):
    """y = leaky_relu(x * demod[b, c] + strength * noise[b, h, w] + bias[c]) * scale on [B, H, W, C] in ONE launch
    (the reference spends three passes: weight demodulation aside, NoiseInjection and FusedLeakyReLU,
    stylegan2.py:268-329).  sign(y) = sign(pre-activation), so y gates the backward.  The backward is written with
    differentiable pieces (the self-differentiating gate + torch products / sums), which gives every higher order."""

    @staticmethod
    def forward(ctx, x, demod, noise, strength, bias, slope, scale):
        b, h, w, c = x.shape
        x = x.contiguous()
        y = torch.empty_like(x)
        per_image = noise is not None and noise.shape[0] == b and b > 1
        nat.call("diagan_styled_bias_act", nat.ptr(x), nat.ptr(demod.contiguous()) if demod is not None else None,
                 nat.ptr(noise.contiguous()) if noise is not None else None,
                 nat.ptr(strength) if noise is not None else None, nat.ptr(bias) if bias is not None else None,
                 nat.ptr(y), b, h * w, c, 1 if per_image else 0, float(slope), float(scale), nat.current_stream())
        ctx.save_for_backward(x, demod, noise, y)
        ctx.hyper = (slope, scale)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, demod, noise, y = ctx.saved_tensors
        need = ctx.needs_input_grad
        if _fused_bwd_ok(gy):
            # first-order backward: gate, demodulated gradient, d(demod), d(bias), d(noise strength) from ONE pass over gy
            gx, gd, gb, gs = _fused_bwd(gy, y, x if (demod is not None and need[1]) else None, demod,
                                        noise if need[3] else None, *ctx.hyper, need_gx=need[0])
            return gx, gd, None, gs, (gb if need[4] else None), None, None
        gpre = _LeakyGate.apply(gy, y, *ctx.hyper)
        gx = (scale_rows(gpre, demod) if demod is not None else gpre) if need[0] else None
        gd = rowdot(gpre, x) if demod is not None and need[1] else None
        gs = (gpre * noise).sum().reshape(1) if noise is not None and need[3] else None
        gb = gpre.sum((0, 1, 2)) if need[4] else None
        return gx, gd, None, gs, gb, None, None


nat.register("diagan_styled_bias_act_mod", [P, P, P, P, P, P, P, P, I, I, I, I, F32, F32, P])
nat.register("diagan_styled_bias_act_mod_bwd", [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, F32, F32, P])


class _StyledActMod(Function):
    """(y, y * post[b, c]) with y the StyledConv tail of x (see _StyledAct) in ONE launch: the next layer's modulated input leaves in
    the pass that makes y (round 6).  First-order backward: ONE pass over gy (None when y has no other consumer) + g_mod * post --
    gate, demodulated gradient, d(demod), d(bias), d(strength), d(post); otherwise the differentiable composition."""

    @staticmethod
    def forward(ctx, x, demod, noise, strength, bias, post, slope, scale):
        b, h, w, c = x.shape
        x, post = x.contiguous(), post.contiguous()
        y, ym = torch.empty_like(x), torch.empty_like(x)
        per_image = noise is not None and noise.shape[0] == b and b > 1
        nat.call("diagan_styled_bias_act_mod", nat.ptr(x), nat.ptr(demod.contiguous()) if demod is not None else None,
                 nat.ptr(noise.contiguous()) if noise is not None else None, nat.ptr(strength) if noise is not None else None,
                 nat.ptr(bias) if bias is not None else None, nat.ptr(post), nat.ptr(y), nat.ptr(ym), b, h * w, c, 1 if per_image else 0,
                 float(slope), float(scale), nat.current_stream())
        ctx.save_for_backward(x, demod, noise, y, post)
        ctx.hyper = (slope, scale)
        ctx.set_materialize_grads(False)
        return y, ym

    @staticmethod
    def backward(ctx, gy, gm):
        x, demod, noise, y, post = ctx.saved_tensors
        need = ctx.needs_input_grad
        slope, scale = ctx.hyper
        if gm is None:                                     # only y was used: the plain tail's backward
            if gy is None:
                return (None,) * 8
            gm_total, gp = gy, None
        elif _fused_bwd_ok(gm):
            b, h, w, c = gm.shape
            gm = gm.contiguous()
            gy = gy.contiguous() if gy is not None else None
            chunks = nat.fn("diagan_rowdot_chunks")(b, h * w)
            f32 = dict(dtype=torch.float32, device=gm.device)
            want_d, want_s = demod is not None and need[1], noise is not None and need[3]
            gx = torch.empty_like(gm) if need[0] else None
            wd = torch.empty((b, chunks, c), **f32) if want_d else None
            wb, wp = torch.empty((b * chunks, c), **f32), torch.empty((b, chunks, c), **f32)
            per_image = noise is not None and noise.shape[0] == b and b > 1
            ws = torch.empty(b * chunks, **f32) if want_s else None
            st = nat.current_stream()
            nat.call("diagan_styled_bias_act_mod_bwd", nat.ptr(gy), nat.ptr(gm), nat.ptr(post), nat.ptr(y),
                     nat.ptr(x) if want_d else None, nat.ptr(demod.contiguous()) if demod is not None else None,
                     nat.ptr(noise.contiguous()) if want_s else None, nat.ptr(gx), nat.ptr(wd), nat.ptr(wb), nat.ptr(ws), nat.ptr(wp),
                     b, h * w, c, 1 if per_image else 0, float(slope), float(scale), st)
            gd = torch.empty((b, c), **f32) if want_d else None
            gb = torch.empty(c, **f32)
            gs = torch.empty(1, **f32) if want_s else None
            gp = torch.empty((b, c), **f32)
            nat.call("diagan_styled_bias_act_bwd_finish", nat.ptr(wd), nat.ptr(wb), nat.ptr(ws), nat.ptr(gd), nat.ptr(gb), nat.ptr(gs),
                     b, h * w, c, st)
            nat.call("diagan_styled_bias_act_bwd_finish", nat.ptr(wp), None, None, nat.ptr(gp), None, None, b, h * w, c, st)
            return gx, gd, None, gs, (gb if need[4] else None), (gp if need[5] else None), None, None
        else:
            gp = rowdot(gm, y) if need[5] else None
            gm_total = scale_rows(gm, post)
            if gy is not None:
                gm_total = gm_total + gy
        if _fused_bwd_ok(gm_total):
            gx, gd, gb, gs = _fused_bwd(gm_total, y, x if (demod is not None and need[1]) else None, demod,
                                        noise if need[3] else None, slope, scale, need_gx=need[0])
            return gx, gd, None, gs, (gb if need[4] else None), gp, None, None
        gpre = _LeakyGate.apply(gm_total, y, slope, scale)
        gx = (scale_rows(gpre, demod) if demod is not None else gpre) if need[0] else None
        gd = rowdot(gpre, x) if demod is not None and need[1] else None
        gs = (gpre * noise).sum().reshape(1) if noise is not None and need[3] else None
        gb = gpre.sum((0, 1, 2)) if need[4] else None
        return gx, gd, None, gs, gb, gp, None, None


def styled_bias_act_mod(x, demod, noise, strength, bias, post, negative_slope=0.2, scale=2 ** 0.5):
    """(y, y * post[:, None, None, :]) with y = styled_bias_act(x, demod, noise, strength, bias); x [B,H,W,C] with C a multiple of 4"""
    return _StyledActMod.apply(x, demod, noise, strength, bias, post, negative_slope, scale)


def styled_bias_act(x, demod=None, noise=None, strength=None, bias=None, negative_slope=0.2, scale=2 ** 0.5):
    """x [B,H,W,C]; demod [B,C]; noise [B or 1, H, W, 1] with its scalar `strength` [1]; bias [C]"""
    return _StyledAct.apply(x, demod, noise, strength, bias, negative_slope, scale)


def fused_leaky_relu(input, bias=None, negative_slope=0.2, scale=2 ** 0.5, bias_dim=1):
    return _BiasLeakyReLU.apply(input, bias, negative_slope, scale, bias_dim)


class FusedLeakyReLU(nn.Module):
    """Module form (reference fused_act.py:90-103): owns the per-channel bias."""

    def __init__(self, channel, bias=True, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel)) if bias else None
        self.negative_slope, self.scale = negative_slope, scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
