"""HBM-bound layer ops, loss heads and Adam over the C ABI (csrc/elementwise.hip, train_ops.hip)."""
import ctypes

import torch

from diagan import _native as nat

P, I, F, I64 = nat.c_void_p, nat.c_int, nat.c_f32, nat.c_i64
nat.register("diagan_nchw_to_nhwc", [P, P, I, I, I, I, I, P])
nat.register("diagan_nhwc_to_nchw", [P, P, I, I, I, I, I, P])
nat.register("diagan_tanh_fwd", [P, P, I64, P])
nat.register("diagan_tanh_bwd", [P, P, P, I64, P])
nat.register("diagan_colred_workspace", [I64, I])
nat.register("diagan_bn_stats", [P, I64, I, P, P, F, F, P, P, I, P, P, P, P, P, P])
nat.register("diagan_bn_stats_fused", [P, I, I64, I, P, P, F, F, P, P, P, P, P, P, I, P, I64, P])
nat.register("diagan_bn_stats_fused_splits", [I, I, I])
nat.register("diagan_bn_bwd", [P, P, I64, I, P, P, P, P, I, I, F, P, F, P, P, I, P, P, P, P, P])
nat.register("diagan_act_fwd", [P, P, P, F, P, F, P, I64, I, P])
nat.register("diagan_act_bwd", [P, P, F, P, F, P, I64, P])
nat.register("diagan_linear1_bwd_input", [P, P, P, I, I, P])
nat.register("diagan_linear1_fwd", [P, P, P, P, I, I, P])
nat.register("diagan_linear1_wgrad", [P, P, P, P, I, I, P])
nat.register("diagan_bn_stats_grouped", [P, I64, I, I, P, P, F, F, P, P, P, P, P, P, P, P])
nat.register("diagan_colsum", [P, I64, I, P, I, P, P])
nat.register("diagan_upsample2x", [P, P, I, I, I, I, I, P, P, I, P])
nat.register("diagan_upsample2x_bwd", [P, P, I, I, I, I, P, P])
nat.register("diagan_avgpool2", [P, P, I, I, I, I, P, I, P])
nat.register("diagan_avgpool2_bwd", [P, P, I, I, I, I, P, P])
nat.register("diagan_boxsum2", [P, P, I, I, I, I, I, P])
nat.register("diagan_head_fwd", [P, P, P, P, I, P, P, P, I, I, I, P])
nat.register("diagan_head_bwd", [P, P, P, P, I, P, P, P, P, P, P, I, I, I, I, P])
nat.register("diagan_add", [P, P, P, I64, P])
nat.register("diagan_loss_dis", [P, I, P, I, I, I, P, P, P, P])
nat.register("diagan_loss_gen", [P, I, I, I, P, P, P])
nat.register("diagan_adam_step", [P, P, P, P, I64, F, F, F, F, F, F, F, P])
nat.register("diagan_adam_step_dev", [P, P, P, P, I64, P, P])

LOSS_TYPES = {'gan': 0, 'ns': 1, 'hinge': 2, 'wasserstein': 3}
ptr, st = nat.ptr, nat.current_stream


def _f32(shape, dev):
    return torch.empty(shape, dtype=torch.float32, device=dev)


_ws = {}
_ws_retired = []      # outgrown scratch buffers stay allocated: a captured hipGraph (utils/graph.py) may still point at them


def _workspace(dev, nbytes):
    w = _ws.get(dev.index)
    if w is None or w.numel() < nbytes:
        if w is not None:
            _ws_retired.append(w)
        w = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=dev)
        _ws[dev.index] = w
    return w


def _colred_ws(dev, M, C, groups=1):
    fn = nat.lib().diagan_colred_workspace
    fn.restype = ctypes.c_int64
    fn.argtypes = [ctypes.c_int64, ctypes.c_int]
    return _workspace(dev, fn(M, C) * groups)


def nchw_to_nhwc(x, Cp, out=None):
    B, C, H, W = x.shape
    if out is None:
        out = _f32((B, H, W, Cp), x.device)
    nat.call("diagan_nchw_to_nhwc", ptr(x.contiguous()), ptr(out), B, C, H, W, Cp, st())
    return out


def nhwc_to_nchw(x, C):
    B, H, W, Cp = x.shape
    out = _f32((B, C, H, W), x.device)
    nat.call("diagan_nhwc_to_nchw", ptr(x), ptr(out), B, C, H, W, Cp, st())
    return out


def tanh_fwd(x, out=None):
    y = torch.empty_like(x) if out is None else out
    nat.call("diagan_tanh_fwd", ptr(x), ptr(y), x.numel(), st())
    return y


def tanh_bwd(y, g):
    gx = torch.empty_like(y)
    nat.call("diagan_tanh_bwd", ptr(y), ptr(g), ptr(gx), y.numel(), st())
    return gx


class BNCtx:
    """mean / invstd / scale / shift of one BatchNorm application: [C], or [G, C] when the batch consists of G groups
    that are normalised independently (`group_imgs` images each, 0 = one group)."""
    __slots__ = ("mean", "invstd", "scale", "shift", "M", "C", "training", "group_imgs")


def _bn_ctx(C, M, training, groups, group_imgs, dev):
    buf = _f32((4, groups, C) if groups > 1 else (4, C), dev)
    ctx = BNCtx()
    ctx.mean, ctx.invstd, ctx.scale, ctx.shift, ctx.M, ctx.C = buf[0], buf[1], buf[2], buf[3], M, C
    ctx.training, ctx.group_imgs = bool(training), (group_imgs if groups > 1 else 0)
    return ctx


def bn_ctx_group(ctx, g):
    """The context of group g of a grouped BatchNorm application as a plain one-batch context (views, no copy)."""
    if ctx is None or not ctx.group_imgs:
        return ctx
    out = BNCtx()
    out.mean, out.invstd, out.scale, out.shift = ctx.mean[g], ctx.invstd[g], ctx.scale[g], ctx.shift[g]
    out.M, out.C, out.training, out.group_imgs = ctx.M, ctx.C, ctx.training, 0
    return out


def bn_stats(x, gamma, beta, running_mean, running_var, training, eps=1e-5, momentum=0.1, groups=1):
    """groups > 1: x holds `groups` equally sized batches along dim 0, each normalised with its OWN statistics; the
    running statistics take the momentum updates in group order (as `groups` successive forwards would)."""
    C = x.shape[-1]
    M = x.numel() // C
    if x.shape[0] % groups:
        raise RuntimeError(f"bn_stats: batch of {x.shape[0]} does not split into {groups} groups")
    Mg, bg = M // groups, x.shape[0] // groups
    ctx = _bn_ctx(C, Mg, training, groups, bg, x.device)
    ws = _colred_ws(x.device, Mg, C, groups) if training else None
    if training and groups > 1:
        nat.call("diagan_bn_stats_grouped", ptr(x), Mg, C, groups, ptr(gamma), ptr(beta), eps, momentum, ptr(running_mean),
                 ptr(running_var), ptr(ctx.mean), ptr(ctx.invstd), ptr(ctx.scale), ptr(ctx.shift), ptr(ws), st())
        return ctx
    for g in range(groups):
        o = g * C * 4
        nat.call("diagan_bn_stats", ptr(x) + g * Mg * C * 4, Mg, C, ptr(gamma), ptr(beta), eps, momentum,
                 ptr(running_mean), ptr(running_var), 1 if training else 0, ptr(ctx.mean) + o, ptr(ctx.invstd) + o,
                 ptr(ctx.scale) + o, ptr(ctx.shift) + o, ptr(ws), st())
    return ctx


def bn_stats_fused(partials, tiles, M, gamma, beta, running_mean, running_var, eps=1e-5, momentum=0.1, groups=1,
                   group_imgs=0):
    """Training-mode BatchNorm context from the per-tile sums emitted by the producing conv's epilogue
    (groups > 1: the tiles of a group are contiguous; the groups are finalised in order by one launch)."""
    C = gamma.numel()
    if tiles % groups or M % groups:
        raise RuntimeError(f"bn_stats_fused: {tiles} tiles / {M} rows do not split into {groups} groups")
    tg, Mg = tiles // groups, M // groups
    ctx = _bn_ctx(C, Mg, True, groups, group_imgs, gamma.device)
    splits = nat.fn("diagan_bn_stats_fused_splits")(tg, C, groups)
    ws = torch.empty(groups * splits * 2 * C, dtype=torch.float64, device=gamma.device) if splits > 1 else None
    nat.call("diagan_bn_stats_fused", ptr(partials), tg, Mg, C, ptr(gamma), ptr(beta), eps, momentum,
             ptr(running_mean), ptr(running_var), ptr(ctx.mean), ptr(ctx.invstd), ptr(ctx.scale), ptr(ctx.shift), groups,
             ptr(ws), 0 if ws is None else ws.numel(), st())
    return ctx


def bn_bwd(g, x, ctx, relu, dgamma, dbeta, accumulate, residual=None, slope=0.0, drop=None, drop_scale=1.0):
    dx = torch.empty_like(x)
    coef = _f32((2 * ctx.C,), x.device)
    ws = _colred_ws(x.device, ctx.M, ctx.C)
    nat.call("diagan_bn_bwd", ptr(g), ptr(x), ctx.M, ctx.C, ptr(ctx.scale), ptr(ctx.shift), ptr(ctx.mean),
             ptr(ctx.invstd), 1 if ctx.training else 0, 1 if relu else 0, slope, ptr(drop), drop_scale, ptr(dgamma), ptr(dbeta), 1 if accumulate else 0,
             ptr(residual),
             ptr(dx), ptr(coef), ptr(ws), st())
    return dx


def colsum(x, out, accumulate):
    C = x.shape[-1]
    M = x.numel() // C
    ws = _colred_ws(x.device, M, C)
    nat.call("diagan_colsum", ptr(x), M, C, ptr(out), 1 if accumulate else 0, ptr(ws), st())
    return out


def upsample2x(x, pro=None):
    B, H, W, C = x.shape
    mode, scale, shift, group_imgs = (tuple(pro) + (0,))[:4] if pro is not None else (0, None, None, 0)
    out = _f32((B, 2 * H, 2 * W, C), x.device)
    nat.call("diagan_upsample2x", ptr(x), ptr(out), B, H, W, C, mode, ptr(scale), ptr(shift), group_imgs, st())
    return out


def upsample2x_bwd(g, residual=None):
    B, H2, W2, C = g.shape
    out = _f32((B, H2 // 2, W2 // 2, C), g.device)
    nat.call("diagan_upsample2x_bwd", ptr(g), ptr(out), B, H2 // 2, W2 // 2, C, ptr(residual), st())
    return out


def boxsum2(x, relu_in=False):
    """0.25 * 2x2 box sums of (relu) x on an (H+1) x (W+1) grid (zero outside the image): see diagan_boxsum2"""
    B, H, W, C = x.shape
    out = _f32((B, H + 1, W + 1, C), x.device)
    nat.call("diagan_boxsum2", ptr(x), ptr(out), B, H, W, C, 1 if relu_in else 0, st())
    return out


def avgpool2(x, residual=None, relu_in=False):
    B, H, W, C = x.shape
    out = _f32((B, H // 2, W // 2, C), x.device)
    nat.call("diagan_avgpool2", ptr(x), ptr(out), B, H, W, C, ptr(residual), 1 if relu_in else 0, st())
    return out


def avgpool2_bwd(g, residual=None):
    B, Ho, Wo, C = g.shape
    out = _f32((B, 2 * Ho, 2 * Wo, C), g.device)
    nat.call("diagan_avgpool2_bwd", ptr(g), ptr(out), B, 2 * Ho, 2 * Wo, C, ptr(residual), st())
    return out


def head_fwd(x, w, inv_sigma, bias, inv_sigma1=None):
    """inv_sigma1: second half of the batch uses it (two forwards batched into one)."""
    B, H, W, C = x.shape
    pooled = _f32((B, C), x.device)
    logit = _f32((B, 1), x.device)
    nat.call("diagan_head_fwd", ptr(x), ptr(w), ptr(inv_sigma), ptr(inv_sigma1), B // 2, ptr(bias), ptr(pooled),
             ptr(logit), B, H * W, C, st())
    return pooled, logit


def head_bwd(dlogit, w, inv_sigma, x, pooled, need_gx=True, need_wgrad=True, dbias=None, accumulate_bias=True,
             inv_sigma1=None):
    B, H, W, C = x.shape
    gx = torch.empty_like(x) if need_gx else None
    G = _f32((C,), x.device) if need_wgrad else None
    dot = torch.empty(1, dtype=torch.float64, device=x.device) if need_wgrad else None
    nat.call("diagan_head_bwd", ptr(dlogit), ptr(w), ptr(inv_sigma), ptr(inv_sigma1), B // 2, ptr(x), ptr(pooled),
             ptr(gx), ptr(G), ptr(dot), ptr(dbias), 1 if accumulate_bias else 0, B, H * W, C, st())
    return gx, G, dot


def act_fwd(x, slope, scale=None, shift=None, drop=None, drop_scale=1.0):
    """drop: keep-mask; drop_scale: 1 / (1 - p) for a 0 / 1 mask (torch's bernoulli_), 1 for a mask that carries its scale"""
    C = x.shape[-1]
    out = torch.empty_like(x)
    nat.call("diagan_act_fwd", ptr(x), ptr(scale), ptr(shift), slope, ptr(drop), drop_scale, ptr(out), x.numel() // C, C, st())
    return out


def act_bwd(g, x, slope, drop=None, drop_scale=1.0):
    out = torch.empty_like(x)
    nat.call("diagan_act_bwd", ptr(g), ptr(x), slope, ptr(drop), drop_scale, ptr(out), x.numel(), st())
    return out


def linear1_fwd(x, w, bias):
    B, C = x.shape
    logit = _f32((B, 1), x.device)
    nat.call("diagan_linear1_fwd", ptr(x), ptr(w), ptr(bias), ptr(logit), B, C, st())
    return logit


def linear1_wgrad(dlogit, x, dw, dbias):
    B, C = x.shape
    nat.call("diagan_linear1_wgrad", ptr(dlogit), ptr(x), ptr(dw), ptr(dbias), B, C, st())


def linear1_bwd_input(dlogit, w, B, C):
    gx = _f32((B, C), w.device)
    nat.call("diagan_linear1_bwd_input", ptr(dlogit), ptr(w), ptr(gx), B, C, st())
    return gx


def add(a, b, out=None):
    out = torch.empty_like(a) if out is None else out
    nat.call("diagan_add", ptr(a), ptr(b), ptr(out), a.numel(), st())
    return out


def loss_dis(out_real, out_fake, loss_type, gold=False, need_grad=True, d_real=None, d_fake=None):
    dev = out_real.device
    nr, nf = out_real.numel(), out_fake.numel()
    if need_grad and d_real is None:
        d_real, d_fake = _f32((nr,), dev), _f32((nf,), dev)
    out3 = _f32((3,), dev)
    nat.call("diagan_loss_dis", ptr(out_real), nr, ptr(out_fake), nf, LOSS_TYPES[loss_type], 1 if gold else 0,
             ptr(d_real), ptr(d_fake), ptr(out3), st())
    return out3, d_real, d_fake


def loss_gen(out_fake, loss_type, k=None, need_grad=True):
    dev = out_fake.device
    n = out_fake.numel()
    k = n if k is None else k
    d_fake = _f32((n,), dev) if need_grad else None
    out1 = _f32((1,), dev)
    nat.call("diagan_loss_gen", ptr(out_fake), n, k, LOSS_TYPES[loss_type], ptr(d_fake), ptr(out1), st())
    return out1, d_fake


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    bc1 = 1.0 - beta1 ** step
    bc2_sqrt = (1.0 - beta2 ** step) ** 0.5
    nat.call("diagan_adam_step", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, bc1, bc2_sqrt,
             grad_scale, st())


def adam_hyper_row(lr, beta1, beta2, eps, step, grad_scale=1.0):
    """the eight floats diagan_adam_step_dev reads: same host arithmetic as adam_step"""
    return [lr, beta1, beta2, eps, 1.0 - beta1 ** step, (1.0 - beta2 ** step) ** 0.5, grad_scale, 0.0]


def adam_step_dev(p, g, m, v, hyper_row):
    nat.call("diagan_adam_step_dev", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(hyper_row), st())
