"""Activation passes folded into the pass next to them (round 6; csrc/stylegan_ops.hip: fir_cl4_kernel's fused forms).

The reference's discriminator block runs conv1 -> FusedLeakyReLU -> Blur -> conv2 (stride 2) -> FusedLeakyReLU, + skip, / sqrt 2
(diagan-pkg/diagan/models/stylegan2.py:553-614), its generator's up-sampling StyledConv conv_transpose -> Blur -> NoiseInjection ->
FusedLeakyReLU (:268-329).  Activation and blur are one pass each over a full-resolution tensor there; here

    bias_act_blur(z, bias, kernel, pad)        = blur(leaky_relu(z + bias) * scale)           one pass (the blur's)
    bias_act_add(z, bias, r)                   = leaky_relu(z + bias) * scale + r             one pass instead of two
    blur_styled_act(x, kernel, pad, ...)       = the StyledConv tail of blur(x) [* next style] one pass (no graph recorded only:
                                                 the backward of the tail needs the blurred tensor itself)

each bit-identical to the launches it replaces.  Autograd: the first-order backward (no graph being recorded) is the blur's adjoint
followed by ONE gate pass that takes the sign of z + bias and the bias gradient's partial sums (diagan_bias_act_gate_bwd); when
the backward is itself differentiated (R1 / path-length penalties, stylegan2/train_ffhq.py:74-102) it is the composition of the
self-differentiating pieces of fused_act.py / upfirdn2d.py, with the activated tensor recomputed as the gate's reference."""
import os

import torch
from torch.autograd import Function

from diagan import _native as nat
from diagan.models.op import fused_act as FA
from diagan.models.op.upfirdn2d import _LinearFIR, _Plan, upfirdn2d_nhwc

P, I, F32, I64 = nat.c_void_p, nat.c_int, nat.c_f32, nat.c_i64
nat.register("diagan_bias_act_fir", [P, P, P, P] + [I] * 10 + [F32, F32, P])
nat.register("diagan_bias_act_gate_bwd", [P, P, P, P, P, I, I, I, F32, F32, P])
nat.register("diagan_bias_act_add", [P, P, P, P, I64, I, F32, F32, P])
nat.register("diagan_fir_gate_bwd", [P, P, P, P, P, P] + [I] * 10 + [F32, F32, P])
nat.register("diagan_fir_styled_act", [P, P, P] + [I] * 10 + [P, P, P, P, P, I, F32, F32, P])

FUSED_TAILS = os.environ.get("DIAGAN_SG2_FUSED_TAILS", "1") != "0"
FUSED_GATE = os.environ.get("DIAGAN_SG2_FUSED_GATE", "1") != "0"      # the blur's adjoint + the activation's gate in one pass


def _fir_ok(x, kernel):
    return (FUSED_TAILS and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.shape[3] % 4 == 0 and kernel.dim() == 2
            and kernel.shape[1] == 4 and kernel.shape[0] <= 16)


def _gate_from_pre(gy, z, bias, slope, scale):
    """(gz, d(bias)) of leaky_relu(z + bias) * scale from the incoming gradient, ONE pass: the gate is the sign of z + bias"""
    b, h, w, c = gy.shape
    gy = gy.contiguous()
    chunks = nat.fn("diagan_rowdot_chunks")(b, h * w)
    gz = torch.empty_like(gy)
    wb = torch.empty((b * chunks, c), dtype=torch.float32, device=gy.device)
    st = nat.current_stream()
    nat.call("diagan_bias_act_gate_bwd", nat.ptr(gy), nat.ptr(z), nat.ptr(bias), nat.ptr(gz), nat.ptr(wb), b, h * w, c, float(slope),
             float(scale), st)
    gb = torch.empty(c, dtype=torch.float32, device=gy.device)
    nat.call("diagan_styled_bias_act_bwd_finish", None, nat.ptr(wb), None, None, nat.ptr(gb), None, b, h * w, c, st)
    return gz, gb


def _blur_adjoint_gate(gy, dual, z, bias, slope, scale):
    """(gz, d(bias)) of blur(leaky_relu(z + bias) * scale): the blur's adjoint and the gate in ONE pass (diagan_fir_gate_bwd)"""
    b, h, w, c = gy.shape
    if FUSED_GATE and dual.kernel.shape[1] == 4 and dual.kernel.shape[0] <= 16 and tuple(dual.out_hw) == tuple(z.shape[1:3]):
        kh, kw = dual.kernel.shape
        gz = torch.empty_like(z)
        chunks = nat.fn("diagan_rowdot_chunks")(b, z.shape[1] * z.shape[2])
        wb = torch.empty((b * chunks, c), dtype=torch.float32, device=gy.device)
        st = nat.current_stream()
        nat.call("diagan_fir_gate_bwd", nat.ptr(gy), nat.ptr(dual.kernel.contiguous()), nat.ptr(z), nat.ptr(bias), nat.ptr(gz), nat.ptr(wb),
                 b, h, w, c, kh, kw, *dual.pad, float(slope), float(scale), st)
        gb = torch.empty(c, dtype=torch.float32, device=gy.device)
        nat.call("diagan_styled_bias_act_bwd_finish", None, nat.ptr(wb), None, None, nat.ptr(gb), None, b, z.shape[1] * z.shape[2], c, st)
        return gz, gb
    return _gate_from_pre(dual.run(gy), z, bias, slope, scale)


def _gate_any_order(g, z, bias, slope, scale):
    """the same gate as differentiable pieces (fused_act._LeakyGate differentiates itself); the activated tensor is recomputed as its
    reference -- only the penalties' double backward comes here"""
    with torch.no_grad():
        ref = FA.fused_bias_act(z, bias, None, 3, 0, slope, scale, -1)
    gz = FA._LeakyGate.apply(g, ref, slope, scale)
    return gz, gz.sum((0, 1, 2))


class _BiasActBlur(Function):
    @staticmethod
    def forward(ctx, z, bias, slope, scale, plan):
        z, bias = z.contiguous(), bias.contiguous()
        b, h, w, c = z.shape
        kh, kw = plan.kernel.shape
        out = torch.empty((b, plan.out_hw[0], plan.out_hw[1], c), dtype=torch.float32, device=z.device)
        nat.call("diagan_bias_act_fir", nat.ptr(z), nat.ptr(bias), nat.ptr(plan.kernel.contiguous()), nat.ptr(out), b, h, w, c, kh, kw,
                 *plan.pad, float(slope), float(scale), nat.current_stream())
        ctx.save_for_backward(z, bias)
        ctx.hyper, ctx.plan = (slope, scale), plan
        return out

    @staticmethod
    def backward(ctx, gy):
        z, bias = ctx.saved_tensors
        if FA._fused_bwd_ok(gy):
            gz, gb = _blur_adjoint_gate(gy.contiguous(), ctx.plan.dual, z, bias, *ctx.hyper)
        else:
            gz, gb = _gate_any_order(_LinearFIR.apply(gy.contiguous(), ctx.plan.dual), z, bias, *ctx.hyper)
        return gz, (gb if ctx.needs_input_grad[1] else None), None, None, None


def bias_act_blur(x, bias, kernel, pad, negative_slope=0.2, scale=2 ** 0.5):
    """upfirdn2d_nhwc(fused_leaky_relu(x, bias, bias_dim=-1), kernel, pad=pad) on [B,H,W,C] in one pass over x"""
    if bias is None or not _fir_ok(x, kernel) or kernel.requires_grad:
        return upfirdn2d_nhwc(FA.fused_leaky_relu(x, bias, negative_slope, scale, bias_dim=-1), kernel, pad=pad)
    plan = _Plan.forward_plan(kernel, (1, 1), (1, 1), (pad[0], pad[1], pad[0], pad[1]), x.shape[1:3], channels_last=True)
    return _BiasActBlur.apply(x, bias, negative_slope, scale, plan)


class _BiasActAdd(Function):
    @staticmethod
    def forward(ctx, z, bias, r, slope, scale):
        z, bias, r = z.contiguous(), bias.contiguous(), r.contiguous()
        out = torch.empty_like(z)
        nat.call("diagan_bias_act_add", nat.ptr(z), nat.ptr(bias), nat.ptr(r), nat.ptr(out), z.numel(), z.shape[-1], float(slope),
                 float(scale), nat.current_stream())
        ctx.save_for_backward(z, bias)
        ctx.hyper = (slope, scale)
        return out

    @staticmethod
    def backward(ctx, gy):
        z, bias = ctx.saved_tensors
        need = ctx.needs_input_grad
        gz = gb = None
        if need[0] or need[1]:
            gz, gb = _gate_from_pre(gy, z, bias, *ctx.hyper) if FA._fused_bwd_ok(gy) else _gate_any_order(gy, z, bias, *ctx.hyper)
        return gz, (gb if need[1] else None), (gy if need[2] else None), None, None


def bias_act_add(x, bias, r, negative_slope=0.2, scale=2 ** 0.5):
    """fused_leaky_relu(x, bias, bias_dim=-1) + r on [B,H,W,C] in one pass"""
    if (not FUSED_TAILS or bias is None or not x.is_cuda or x.dim() != 4 or x.dtype != torch.float32 or x.shape[3] % 4
            or r.shape != x.shape or r.dtype != torch.float32):
        return FA.fused_leaky_relu(x, bias, negative_slope, scale, bias_dim=-1) + r
    return _BiasActAdd.apply(x, bias, r, negative_slope, scale)


def blur_styled_act_ok(x, kernel):
    """the one-pass blur + StyledConv tail applies: nothing records a graph (its backward needs the blurred tensor)"""
    return not torch.is_grad_enabled() and _fir_ok(x, kernel)


def blur_styled_act(x, kernel, pad, demod=None, noise=None, strength=None, bias=None, negative_slope=0.2, scale=2 ** 0.5, post=None):
    """[post[b, c] *] styled_bias_act(upfirdn2d_nhwc(x, kernel, pad=pad), demod, noise, strength, bias) in ONE pass; no autograd
    (callers check blur_styled_act_ok)"""
    x = x.contiguous()
    b, h, w, c = x.shape
    kh, kw = kernel.shape
    oh, ow = h + pad[0] + pad[1] - kh + 1, w + pad[0] + pad[1] - kw + 1
    out = torch.empty((b, oh, ow, c), dtype=torch.float32, device=x.device)
    per_image = noise is not None and noise.shape[0] == b and b > 1
    c_ = lambda t: nat.ptr(t.contiguous()) if t is not None else None           # noqa: E731
    nat.call("diagan_fir_styled_act", nat.ptr(x), nat.ptr(kernel.contiguous()), nat.ptr(out), b, h, w, c, kh, kw, pad[0], pad[1], pad[0],
             pad[1], c_(demod), c_(noise), nat.ptr(strength) if noise is not None else None, c_(bias), c_(post), 1 if per_image else 0,
             float(negative_slope), float(scale), nat.current_stream())
    return out


# ---- the generator's ToRGB in one pass over its input ---------------------------------------------------------------------------------
nat.register("diagan_torgb_fwd", [P, P, P, P, P, I, I, I, P])
nat.register("diagan_torgb_bwd", [P, P, P, P, P, P, P, P, I, I, I, P])


def torgb_ok(x):
    c = x.shape[-1] if x.dim() == 4 else 0
    return FUSED_TAILS and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and 4 <= c <= 1024 and not (c & (c - 1))


class _ToRGB(Function):
    """out[b,h,w,:3] = sum_c (weight * scale)[o, c] * (x * s)[b,h,w,c] + bias[o] (4th plane zero): the reference's ToRGB
    (stylegan2.py:332-351) in one read of x.  First-order backward: one read of x + one write of gx (diagan_torgb_bwd); when the
    backward is itself differentiated, the composition scale_rows -> 1x1 convolution of ops/diffconv.py (any order)."""

    @staticmethod
    def forward(ctx, x, s, weight, bias, scale):
        b, h, w, c = x.shape
        x, s = x.contiguous(), s.contiguous()
        ws = (weight * scale).contiguous()                                   # [3, C]
        out = torch.empty((b, h, w, 4), dtype=torch.float32, device=x.device)
        nat.call("diagan_torgb_fwd", nat.ptr(x), nat.ptr(s), nat.ptr(ws), nat.ptr(bias.contiguous()) if bias is not None else None,
                 nat.ptr(out), b, h * w, c, nat.current_stream())
        ctx.save_for_backward(x, s, weight, bias)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, gy):
        x, s, weight, bias = ctx.saved_tensors
        need = ctx.needs_input_grad
        b, h, w, c = x.shape
        if not torch.is_grad_enabled():
            gy = gy.contiguous()
            ws = (weight * ctx.scale).contiguous()
            f32 = dict(dtype=torch.float32, device=x.device)
            gx = torch.empty_like(x) if need[0] else None
            chunks = nat.fn("diagan_rowdot_chunks")(b, h * w)
            work = torch.empty(b * (chunks + 1) * 3 * c, **f32)
            gs, gw = torch.empty((b, c), **f32), torch.empty((3, c), **f32)
            nat.call("diagan_torgb_bwd", nat.ptr(gy), nat.ptr(x), nat.ptr(s), nat.ptr(ws), nat.ptr(gx), nat.ptr(work), nat.ptr(gs),
                     nat.ptr(gw), b, h * w, c, nat.current_stream())
            gb = gy.sum((0, 1, 2))[:3] if (bias is not None and need[3]) else None
            return gx, (gs if need[1] else None), (gw * ctx.scale if need[2] else None), gb, None
        # differentiable composition on the saved (graph-connected) tensors
        from diagan.ops import diffconv as dc
        from diagan.ops import conv as K
        geom = K.Geom('conv', c, 4, 1, 1, 1, 0)
        wp = dc.pack_scaled(weight.view(3, c, 1, 1), ctx.scale, geom)
        xm = FA.scale_rows(x, s)
        gxm = dc._DataGrad.apply(gy, wp, geom, (h, w)) if (need[0] or need[1]) else None
        gx = FA.scale_rows(gxm, s) if need[0] else None
        gs = FA.rowdot(gxm, x) if need[1] else None
        gw = None
        if need[2]:
            gwp = dc._WeightGrad.apply(gy, xm, geom)
            gw = dc._UnpackScaled.apply(gwp, ctx.scale, geom, (3, c, 1, 1)).view(3, c)
        gb = gy.sum((0, 1, 2))[:3] if (bias is not None and need[3]) else None
        return gx, gs, gw, gb, None


def torgb(x, s, weight, bias, scale):
    """x [B,H,W,C], s [B,C], weight [3,C], bias [3] or None -> [B,H,W,4]"""
    return _ToRGB.apply(x, s, weight, bias, scale)


# ---- a tensor that feeds a convolution AND a resampling filter: the filter's adjoint adds the other gradient on its way out -------------
nat.register("diagan_upfirdn2d_add", [P, P, P, P] + [I] * 14 + [P])


class _ForkFIR(Function):
    """(x, upfirdn2d(x)): autograd hands this node BOTH consumers' gradients at once, so the plain backward is ONE pass --
    adjoint filter + the other branch's gradient (diagan_upfirdn2d_add) -- instead of the adjoint and a separate accumulation over a
    full-resolution tensor.  Differentiated backward: the sum of the differentiable pieces."""

    @staticmethod
    def forward(ctx, x, plan):
        ctx.plan = plan
        return x.view_as(x), plan.run(x)

    @staticmethod
    def backward(ctx, ga, gf):
        dual = ctx.plan.dual
        if gf is None:
            return ga, None
        if ga is None:
            return _LinearFIR.apply(gf.contiguous(), dual), None
        if torch.is_grad_enabled() or not (gf.is_cuda and gf.shape[3] % 4 == 0 and dual.channels_last):
            return ga + _LinearFIR.apply(gf.contiguous(), dual), None
        gf, ga = gf.contiguous(), ga.contiguous()
        b, h, w, c = gf.shape
        kh, kw = dual.kernel.shape
        out = torch.empty_like(ga)
        nat.call("diagan_upfirdn2d_add", nat.ptr(gf), nat.ptr(dual.kernel.contiguous()), nat.ptr(ga), nat.ptr(out), b, h, w, c, kh, kw,
                 dual.up[0], dual.up[1], dual.down[0], dual.down[1], *dual.pad, nat.current_stream())
        return out, None


def fork_fir(x, kernel, up=1, down=1, pad=(0, 0)):
    """(x, upfirdn2d_nhwc(x, kernel, up, down, pad)) for a tensor with one more consumer besides the filter"""
    if not (FUSED_TAILS and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.shape[3] % 4 == 0) or kernel.requires_grad:
        return x, upfirdn2d_nhwc(x, kernel, up=up, down=down, pad=pad)
    plan = _Plan.forward_plan(kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]), x.shape[1:3], channels_last=True)
    return _ForkFIR.apply(x, plan)


# ---- the discriminator's first layer (1x1 convolution from RGB + bias + leaky ReLU) in one write of its output --------------------------
nat.register("diagan_fromrgb_fwd", [P, P, P, P, I, I, I, F32, F32, F32, P])
nat.register("diagan_fromrgb_bwd", [P, P, P, P, P, P, I, I, I, F32, F32, F32, P])


def fromrgb_ok(x, weight, bias):
    c = weight.shape[0]
    return (FUSED_TAILS and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.shape[3] == 4 and bias is not None
            and tuple(weight.shape[1:]) == (3, 1, 1) and 4 <= c <= 256 and not (c & (c - 1)))


class _FromRGB(Function):
    """y = leaky_relu(conv1x1(x[..., :3], weight * wscale) + bias) * scale on x [B,H,W,4] (RGB + zero plane): the reference's first
    ConvLayer (stylegan2.py:553-595) in one write of y.  Plain backward: ONE read of gy and y for the gate, d(weight), d(bias) and -- when
    the images need it -- d(x); differentiated backward: gate + convolution ops of ops/diffconv.py (any order)."""

    @staticmethod
    def forward(ctx, x, weight, bias, wscale, slope, scale):
        b, h, w, _ = x.shape
        c = weight.shape[0]
        x, weight, bias = x.contiguous(), weight.contiguous(), bias.contiguous()
        y = torch.empty((b, h, w, c), dtype=torch.float32, device=x.device)
        nat.call("diagan_fromrgb_fwd", nat.ptr(x), nat.ptr(weight), nat.ptr(bias), nat.ptr(y), b, h * w, c, float(wscale), float(slope),
                 float(scale), nat.current_stream())
        ctx.save_for_backward(x, weight, y)
        ctx.hyper = (wscale, slope, scale)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        wscale, slope, scale = ctx.hyper
        need = ctx.needs_input_grad
        b, h, w, c = gy.shape
        if not torch.is_grad_enabled():
            gy = gy.contiguous()
            f32 = dict(dtype=torch.float32, device=gy.device)
            chunks = nat.fn("diagan_rowdot_chunks")(b, h * w)
            work = torch.empty((b * chunks, 4 * c), **f32)
            gx = torch.empty_like(x) if need[0] else None
            st = nat.current_stream()
            nat.call("diagan_fromrgb_bwd", nat.ptr(gy), nat.ptr(y), nat.ptr(x), nat.ptr(weight), nat.ptr(gx), nat.ptr(work), b, h * w, c,
                     float(wscale), float(slope), float(scale), st)
            sums = torch.empty(4 * c, **f32)
            nat.call("diagan_styled_bias_act_bwd_finish", None, nat.ptr(work), None, None, nat.ptr(sums), None, b, h * w, 4 * c, st)
            sums = sums.view(4, c)
            gw = (sums[:3].t() * wscale).reshape(c, 3, 1, 1) if need[1] else None
            return gx, gw, (sums[3] if need[2] else None), None, None, None
        from diagan.ops import diffconv as dc
        from diagan.ops import conv as K
        geom = K.Geom('conv', 4, c, 1, 1, 1, 0)
        gz = FA._LeakyGate.apply(gy, y, slope, scale)
        wp = dc.pack_scaled(weight, wscale, geom)
        gx = dc._DataGrad.apply(gz, wp, geom, (h, w)) if need[0] else None
        gw = dc._UnpackScaled.apply(dc._WeightGrad.apply(gz, x, geom), wscale, geom, tuple(weight.shape)) if need[1] else None
        gb = gz.sum((0, 1, 2)) if need[2] else None
        return gx, gw, gb, None, None, None


def fromrgb(x, weight, bias, wscale, negative_slope=0.2, scale=2 ** 0.5):
    """x [B,H,W,4], weight [C,3,1,1], bias [C] -> [B,H,W,C]"""
    return _FromRGB.apply(x, weight, bias, wscale, negative_slope, scale)


# ---- the small dense pieces of the modulated convolution as single launches (csrc/stylegan_dense.hip) ----------------------------------
nat.register("diagan_small_linear_fwd", [P, P, P, P, I, I, I, F32, F32, P])
nat.register("diagan_small_linear_bwd", [P, P, P, P, P, P, I, I, I, F32, F32, P])
nat.register("diagan_demod_fwd", [P, P, P, P, I, I, I, I, F32, F32, P])
nat.register("diagan_demod_bwd", [P, P, P, P, P, P, P, I, I, I, I, F32, P])

FUSED_DENSE = os.environ.get("DIAGAN_SG2_FUSED_DENSE", "1") != "0"


def _vjp_any_order(fn, inputs, cotangent, need):
    """gradients of fn(*inputs) for a backward that is itself being differentiated: fn is recomputed from the saved (graph-connected)
    tensors with differentiable ops and differentiated with create_graph"""
    with torch.enable_grad():
        out = fn(*inputs)
        wanted = [t for t, n in zip(inputs, need) if n and t is not None and t.requires_grad]
        grads = iter(torch.autograd.grad(out, wanted, cotangent, create_graph=True, allow_unused=True)) if wanted else iter(())
    return [next(grads) if (n and t is not None and t.requires_grad) else None for t, n in zip(inputs, need)]


class _ModLinear(Function):
    """EqualLinear without activation (reference stylegan2.py:132-166): x @ (W * scale).T + bias * lr_mul in one launch; plain backward
    in one launch (d(W), d(bias), d(x))"""

    @staticmethod
    def forward(ctx, x, W, bias, scale, lr_mul):
        x, W = x.contiguous(), W.contiguous()
        b, k = x.shape
        c = W.shape[0]
        out = torch.empty((b, c), dtype=torch.float32, device=x.device)
        nat.call("diagan_small_linear_fwd", nat.ptr(x), nat.ptr(W), nat.ptr(bias.contiguous()) if bias is not None else None, nat.ptr(out),
                 b, k, c, float(scale), float(lr_mul), nat.current_stream())
        ctx.save_for_backward(x, W, bias)
        ctx.hyper = (scale, lr_mul)
        return out

    @staticmethod
    def backward(ctx, g):
        x, W, bias = ctx.saved_tensors
        scale, lr_mul = ctx.hyper
        need = ctx.needs_input_grad
        if torch.is_grad_enabled():
            from diagan.ops import diffconv as dc
            fn = lambda x_, W_, b_: dc.linear(x_, W_, scale=scale) + (b_ * lr_mul if b_ is not None else 0.0)       # noqa: E731
            return tuple(_vjp_any_order(fn, (x, W, bias), g, need[:3])) + (None, None)
        b, k = x.shape
        c = W.shape[0]
        g = g.contiguous()
        f32 = dict(dtype=torch.float32, device=g.device)
        gW = torch.empty_like(W)
        gb = torch.empty(c, **f32) if (bias is not None and need[2]) else None
        gx = torch.empty_like(x) if need[0] else None
        nat.call("diagan_small_linear_bwd", nat.ptr(g), nat.ptr(x), nat.ptr(W), nat.ptr(gW), nat.ptr(gb), nat.ptr(gx), b, k, c, float(scale),
                 float(lr_mul), nat.current_stream())
        return gx, (gW if need[1] else None), gb, None, None


def mod_linear_ok(x, W):
    return FUSED_DENSE and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[1] % 4 == 0 and W.dim() == 2


def mod_linear(x, W, bias, scale, lr_mul=1.0):
    return _ModLinear.apply(x, W, bias, scale, lr_mul)


class _Demod(Function):
    """d[b, co] = rsqrt(scale^2 * sum_ci s[b, ci]^2 * sum_taps w[co, ci, :, :]^2 + eps) (reference stylegan2.py:236-238, as a [B,Ci] x [Ci,Co]
    product) in one launch; plain backward (d(s), d(w)) in one launch"""

    @staticmethod
    def forward(ctx, s, w, scale2, eps):
        s, w = s.contiguous(), w.contiguous()
        b, ci = s.shape
        co, taps = w.shape[0], w.shape[2] * w.shape[3]
        f32 = dict(dtype=torch.float32, device=s.device)
        d, wsq = torch.empty((b, co), **f32), torch.empty((co, ci), **f32)
        nat.call("diagan_demod_fwd", nat.ptr(s), nat.ptr(w), nat.ptr(d), nat.ptr(wsq), b, ci, co, taps, float(scale2), float(eps),
                 nat.current_stream())
        ctx.save_for_backward(s, w, d, wsq)
        ctx.hyper = (scale2, eps)
        return d

    @staticmethod
    def backward(ctx, gd):
        s, w, d, wsq = ctx.saved_tensors
        scale2, eps = ctx.hyper
        need = ctx.needs_input_grad
        if torch.is_grad_enabled() or s.shape[0] > 64:
            from diagan.ops import diffconv as dc
            fn = lambda s_, w_: torch.rsqrt(dc.linear(s_.square(), w_.square().sum((2, 3)), scale=scale2) + eps)       # noqa: E731
            return tuple(_vjp_any_order(fn, (s, w), gd, need[:2])) + (None, None)
        b, ci = s.shape
        co, taps = w.shape[0], w.shape[2] * w.shape[3]
        gw = torch.empty_like(w)
        gs = torch.empty_like(s) if need[0] else None
        nat.call("diagan_demod_bwd", nat.ptr(gd.contiguous()), nat.ptr(d), nat.ptr(s), nat.ptr(w), nat.ptr(wsq), nat.ptr(gw), nat.ptr(gs),
                 b, ci, co, taps, float(scale2), nat.current_stream())
        return gs, (gw if need[1] else None), None, None


def demod_ok(s, w):
    return FUSED_DENSE and s.is_cuda and s.dtype == torch.float32 and w.dim() == 4 and s.dim() == 2 and w.shape[1] == s.shape[1]


def demod(s, w, scale2, eps):
    return _Demod.apply(s, w, scale2, eps)
