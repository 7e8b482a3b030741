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

import os

import ctypes as _ct

from diagan import _native as nat
from diagan.ops import conv as K

_P, _I, _F = nat.c_void_p, nat.c_int, nat.c_f32
nat.register("diagan_pack_oihw", [_P, _F, _P, _P] + [_I] * 8 + [_P])
nat.register("diagan_unpack_oihw", [_P, _F, _P] + [_I] * 6 + [_P])
nat.register("diagan_parity_weights", [_P, _P, _P] + [_I] * 5 + [_P, _P, _I, _P])
# weight preparation in one launch each (csrc/weight_prep.hip, round 6): scale + pack (+ the data-gradient operand), and the parity
# classes' sub-kernels / their adjoint; DIAGAN_SG2_FUSED_PREP=0: the torch-op compositions of rounds 1-5
FUSED_PREP = os.environ.get("DIAGAN_SG2_FUSED_PREP", "1") == "1"

# The 3x3 / stride-1 layers of the StyleGAN2 ops take the Winograd kernels like the SNGAN layers do (+21 % on the 256 x 256
# iteration: 82 -> 99 images/s).  Measured against the oracle in float64 the two convolution paths are equally far from
# the truth: the three-iteration trajectory at 8 x 8 and 16 x 16 (tools/sg2_trajectory.py) and the ill-conditioned
# NoiseInjection strength gradients at 256 x 256 (tools/sg2_noise_grad.py); profiles/r02_sg2_winograd.md.
# DIAGAN_SG2_WINO=0 keeps these ops on the implicit GEMM.
SG2_WINO = os.environ.get("DIAGAN_SG2_WINO", "1") == "1"
# the parity classes of the stride-2 transposed gathers write straight into the interleaved result where the launch runs on the
# split-operand kernel (csrc/conv_gemm_x3b.hip, ConvGemmArgs::map); DIAGAN_SG2_OUT_MAP=0: compute each class, then copy it into place
OUT_MAP = os.environ.get("DIAGAN_SG2_OUT_MAP", "1") == "1"


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _wd(geom, wp):
    """data-gradient operand Wd[Ci][Kd] of packed forward weights: the one `pack_scaled` made beside wp (same launch, same values)
    where there is one"""
    wd = getattr(wp, '_diagan_wd', None)
    if wd is not None and tuple(wd.shape) == (geom.Ci, geom.Kd):
        return wd
    wd = torch.zeros((geom.Ci, geom.Kd), dtype=torch.float32, device=wp.device)
    K.pack_weights(_c(wp), geom.Co, geom.Ci, geom.R * geom.S, geom.Kp, geom.Kd, Wd=wd)
    return wd


# ---- stride-2 transposed gathers without the zeros ------------------------------------------------------------------
# A stride-2 transposed convolution (the generator's up-convolution, and the data gradient of the discriminator's
# stride-2 convolutions) reads y[oy] = sum_r x[(oy - r) / 2] w[r] over the taps r with oy - r even: as ONE implicit GEMM
# over all R*S taps, 3 of every 4 (tap, pixel) products of a 3x3 kernel multiply an inserted zero.  Split by output
# parity (cy, cx) it is four DENSE stride-1 convolutions of the low-resolution input with the sub-kernels
# w[cy::2, cx::2] (2x2, 2x1, 1x2 and 1x1 taps for a 3x3 kernel): 9/4 instead of 9 tap-products per output pixel.
# Each class runs on the same forward GEMM kernel; the class outputs are interleaved into the result.  The weight
# gradient splits the same way (strided slices of dy against the padded input, scattered back into the taps).
def _parity_taps(k):
    """[(output parity c, taps in correlation order)]: y[2m + c] = sum_u xpad[m + u] * w[taps[u]]"""
    return [(c, list(range(c, k, 2))[::-1]) for c in (0, 1) if c < k]


def _splits_stride2(geom, transposed_gather):
    return transposed_gather and geom.stride == 2 and geom.pad == 0


def _sub_problem(C, n_out, ny, nx):
    """dense stride-1 sub-convolution of one parity class.  The class needs the 'full' correlation (ny-1 / nx-1 zeros
    on each side); the kernels take ONE symmetric padding, so the larger of the two is used and the surplus border
    rows / columns of the result (products with padding only) are trimmed when the class is interleaved -- no padded
    copy of the input is ever made."""
    p = max(ny, nx) - 1
    return K.Geom('conv', C, n_out, ny, nx, 1, p), p - (ny - 1), p - (nx - 1)


def _up2_gather(x, w, n_out, R, S, C, out_hw):
    """out[B, oh, ow, n_out] = sum over (iy, ix, r, s, c) with oy = 2 iy + r, ox = 2 ix + s of
    x[b, iy, ix, c] * w[n][(r * S + s) * C + c]"""
    B, H, W, _ = x.shape
    full_h, full_w = 2 * (H - 1) + R, 2 * (W - 1) + S
    exact = (full_h, full_w) == tuple(out_hw) and R > 1 and S > 1
    out = (x.new_empty if exact else x.new_zeros)((B, out_hw[0], out_hw[1], n_out))
    w4 = w[:, : R * S * C].view(n_out, R, S, C)
    subs = _parity_buffers(w, n_out, R, S, C, adjoint=False) if FUSED_PREP and R <= 4 and S <= 4 else None
    for cy, ry in _parity_taps(R):
        for cx, sx in _parity_taps(S):
            ny, nx = len(ry), len(sx)
            sub, ty, tx = _sub_problem(C, n_out, ny, nx)
            if subs is not None:
                ws = subs[2 * cy + cx]
            else:
                ws = w4[:, ry][:, :, sx].reshape(n_out, ny * nx * C)
                if ws.shape[1] != sub.Kp:
                    ws = F.pad(ws, (0, sub.Kp - ws.shape[1]))
            if OUT_MAP and K.out_map_ok(sub, B, H, W):
                # the class interleaves itself: the split-operand kernel writes pixel (my, mx) of the class to (2 (my - ty) + cy,
                # 2 (mx - tx) + cx) and drops the surplus border of the symmetric padding (no copy pass; round 6)
                K.conv_fwd(sub, x, ws.contiguous(), wino=SG2_WINO, out=out,
                           out_map=(2, cy, cx, ty, ty + H + ny - 1, tx, tx + W + nx - 1))
                continue
            y = K.conv_fwd(sub, x, ws.contiguous(), wino=SG2_WINO)
            out[:, cy:full_h:2, cx:full_w:2] = y[:, ty: ty + H + ny - 1, tx: tx + W + nx - 1]
    return out


def _up2_wgrad(g, x, geom):
    """weight gradient of the stride-2 transposed convolution y = convT(x, w): g [B, 2(H-1)+R, 2(W-1)+S, Co]"""
    R, S, Ci, Co = geom.R, geom.S, geom.Ci, geom.Co
    fused = FUSED_PREP and R <= 4 and S <= 4
    parts, buf, kp, off = _parity_buffers(None, Co, R, S, Ci, adjoint=True, device=g.device) if fused else (None, None, None, None)
    grad = None if fused else g.new_zeros((Co, R, S, Ci))
    for cy, ry in _parity_taps(R):
        for cx, sx in _parity_taps(S):
            ny, nx = len(ry), len(sx)
            sub, ty, tx = _sub_problem(Ci, Co, ny, nx)
            part = parts[2 * cy + cx] if fused else torch.empty((Co, sub.Kp), dtype=torch.float32, device=g.device)
            gc = F.pad(g[:, cy::2, cx::2], (0, 0, tx, tx, ty, ty)) if (ty or tx) else g[:, cy::2, cx::2].contiguous()
            K.conv_wgrad(sub, gc, x, part, accumulate=False)
            if fused:
                continue
            part = part[:, : ny * nx * Ci].view(Co, ny, nx, Ci)
            for u, r in enumerate(ry):
                for v, s in enumerate(sx):
                    grad[:, r, s] = part[:, u, v]
    if fused:                                 # every tap of the full operand belongs to exactly one class: one gather, padding zeroed
        full = torch.empty((Co, geom.Kp), dtype=torch.float32, device=g.device)
        nat.call("diagan_parity_weights", None, nat.ptr(buf), nat.ptr(full), Co, R, S, Ci, geom.Kp, kp, off, 1, nat.current_stream())
        return full
    grad = grad.view(Co, R * S * Ci)
    return F.pad(grad, (0, geom.Kp - grad.shape[1])) if grad.shape[1] != geom.Kp else grad


def _parity_buffers(w, n, R, S, C, adjoint, device=None):
    """One buffer for the four parity classes' operands [n][Kp_class] (class index 2 cy + cx; classes without taps: None).
    adjoint False: filled from the packed operand w[n][>= R S C] by ONE launch (diagan_parity_weights) -> list of views;
    adjoint True: empty, for the classes' weight gradients -> (views, buffer, kp array, off array) for the reverse launch."""
    kp, off, total = (_ct.c_int * 4)(), (_ct.c_int * 4)(), 0
    for cy, ry in _parity_taps(R):
        for cx, sx in _parity_taps(S):
            sub, _, _ = _sub_problem(C, n, len(ry), len(sx))
            kp[2 * cy + cx], off[2 * cy + cx] = sub.Kp, total
            total += n * sub.Kp
    dev = w.device if w is not None else device
    buf = torch.empty(total, dtype=torch.float32, device=dev)
    views = [buf[off[c]: off[c] + n * kp[c]].view(n, kp[c]) if kp[c] else None for c in range(4)]
    if adjoint:
        return views, buf, kp, off
    nat.call("diagan_parity_weights", nat.ptr(_c(w)), nat.ptr(buf), None, n, R, S, C, w.shape[1], kp, off, 0, nat.current_stream())
    return views


class _Conv(Function):
    @staticmethod
    def forward(ctx, x, wp, geom):
        ctx.geom = geom
        ctx.wd_cache = getattr(wp, '_diagan_wd', None)       # (the data-gradient operand pack_scaled made beside wp)
        ctx.save_for_backward(x, wp)
        if _splits_stride2(geom, geom.kind == 'convT'):
            return _up2_gather(_c(x), _c(wp), geom.Co, geom.R, geom.S, geom.Ci, geom.out_hw(x.shape[1], x.shape[2]))
        return K.conv_fwd(geom, _c(x), _c(wp), wino=SG2_WINO)

    @staticmethod
    def backward(ctx, gy):
        x, wp = ctx.saved_tensors
        if ctx.wd_cache is not None:
            wp._diagan_wd = ctx.wd_cache
        gx = _DataGrad.apply(gy, wp, ctx.geom, tuple(x.shape[1:3])) if ctx.needs_input_grad[0] else None
        gw = _WeightGrad.apply(gy, x, ctx.geom) if ctx.needs_input_grad[1] else None
        return gx, gw, None


class _DataGrad(Function):
    @staticmethod
    def forward(ctx, g, wp, geom, in_hw):
        ctx.geom, ctx.in_hw = geom, in_hw
        ctx.save_for_backward(g, wp)
        if _splits_stride2(geom, geom.kind == 'conv'):
            return _up2_gather(_c(g), _wd(geom, wp), geom.Ci, geom.R, geom.S, geom.Co, in_hw)
        return K.conv_dgrad(geom, _c(g), _wd(geom, wp), in_hw, wino=SG2_WINO)

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
        if _splits_stride2(geom, geom.kind == 'convT'):
            return _up2_wgrad(_c(g), _c(x), geom)
        grad = torch.empty((geom.Co, geom.Kp), dtype=torch.float32, device=g.device)
        return K.conv_wgrad(geom, _c(g), _c(x), grad, accumulate=False)

    @staticmethod
    def backward(ctx, ggw):
        g, x = ctx.saved_tensors
        gg = _Conv.apply(x, ggw, ctx.geom) if ctx.needs_input_grad[0] else None
        gx = _DataGrad.apply(g, ggw, ctx.geom, tuple(x.shape[1:3])) if ctx.needs_input_grad[1] else None
        return gg, gx, None


class _PackScaled(Function):
    """(Wp[Co][Kp], Wd[Ci][Kd] or None) = pack(w_oihw * scale) in ONE launch; linear in w: the backward is _UnpackScaled, whose
    backward is this again (any order).  Wd is a by-product for the data gradients (not differentiable: they differentiate through Wp)."""

    @staticmethod
    def forward(ctx, w, scale, geom, want_wd):
        co, ci, r, s = w.shape
        ctx.meta = (float(scale), geom, tuple(w.shape))
        f32 = dict(dtype=torch.float32, device=w.device)
        wp = torch.empty((geom.Co, geom.Kp), **f32)
        wd = torch.empty((geom.Ci, geom.Kd), **f32) if want_wd else torch.empty(0, **f32)
        nat.call("diagan_pack_oihw", nat.ptr(_c(w)), float(scale), nat.ptr(wp), nat.ptr(wd) if want_wd else None, co, ci, r, s,
                 geom.Co, geom.Ci, geom.Kp, geom.Kd, nat.current_stream())
        ctx.mark_non_differentiable(wd)
        ctx.set_materialize_grads(False)           # (else the engine zero-fills a [Ci][Kd] "gradient" of wd for every backward)
        return wp, wd

    @staticmethod
    def backward(ctx, gwp, _gwd):
        scale, geom, shape = ctx.meta
        return (_UnpackScaled.apply(gwp, scale, geom, shape) if gwp is not None else None), None, None, None


class _UnpackScaled(Function):
    @staticmethod
    def forward(ctx, gwp, scale, geom, shape):
        ctx.meta = (scale, geom)
        co, ci, r, s = shape
        gw = torch.empty(shape, dtype=torch.float32, device=gwp.device)
        nat.call("diagan_unpack_oihw", nat.ptr(_c(gwp)), float(scale), nat.ptr(gw), co, ci, r, s, geom.Ci, geom.Kp, nat.current_stream())
        return gw

    @staticmethod
    def backward(ctx, ggw):
        scale, geom = ctx.meta
        return _PackScaled.apply(ggw, scale, geom, False)[0], None, None, None


def pack_scaled(w_oihw, scale, geom):
    """pack(w_oihw * scale, geom) in one launch; with gradients enabled the data-gradient operand of the same values rides along
    (picked up by _wd through the tensor's `_diagan_wd`)"""
    if not FUSED_PREP or not w_oihw.is_cuda:
        return pack(w_oihw * scale if scale != 1.0 else w_oihw, geom)
    want_wd = torch.is_grad_enabled()
    wp, wd = _PackScaled.apply(w_oihw, scale, geom, want_wd)
    if want_wd:
        wp._diagan_wd = wd
    return wp


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


def conv2d(x, w_oihw, stride=1, padding=0, scale=1.0):
    """F.conv2d(x, w * scale) on NHWC activations: x [B,H,W,Ci] (Ci % 4 == 0), w [Co', Ci', R, S] -> [B,Ho,Wo,roundup(Co',4)]"""
    geom = K.Geom('conv', x.shape[3], K.round_up(w_oihw.shape[0], 4), w_oihw.shape[2], w_oihw.shape[3], stride, padding)
    return _Conv.apply(x, pack_scaled(w_oihw, scale, geom), geom)


def conv_transpose2d(x, w_oihw, stride=2, padding=0, scale=1.0):
    """F.conv_transpose2d(x, (w * scale).transpose(0, 1)) on NHWC activations: w is given output-channel-major like conv2d's
    (the modulated convolution of the reference transposes it itself, stylegan2.py:243-248)."""
    geom = K.Geom('convT', x.shape[3], K.round_up(w_oihw.shape[0], 4), w_oihw.shape[2], w_oihw.shape[3], stride, padding)
    return _Conv.apply(x, pack_scaled(w_oihw, scale, geom), geom)


def linear(x, weight, scale=1.0):
    """F.linear(x, weight * scale) (no bias) as a 1x1 convolution over [B,1,1,Ci]"""
    b, ci = x.shape
    cp = K.round_up(ci, 4)
    if cp != ci:
        x = F.pad(x, (0, cp - ci))
    co = weight.shape[0]
    y = conv2d(x.view(b, 1, 1, cp), weight.view(co, ci, 1, 1), scale=scale)
    return y.view(b, -1)[:, :co]
