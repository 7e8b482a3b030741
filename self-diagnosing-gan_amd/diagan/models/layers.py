"""Layer objects of the MI355X engine: explicit forward / backward over the HIP kernels.

There is no autograd in the product path.  Every layer keeps its parameters in GEMM-ready HBM
layouts (packed weights Wp[Co][Kp], NHWC activations) inside ONE flat fp32 buffer per network
(`FlatNet`), so that zero_grad is one memset, Adam is one launch and the data-parallel gradient
all-reduce is one RCCL call per network.  state_dict()/load_state_dict() present the reference's
tensor shapes (OIHW conv weights, [out,in] linear weights, mimicry's sn_u / sn_sigma buffers).
"""

import os

import torch
import torch.nn as nn

from diagan.ops import conv as C
from diagan.ops import eltwise as E

# the pooled layers' weight gradient sums its 2x2 boxes in the loader (DIAGAN_WGRAD_BOX=0: a boxsum2 pass in front, as before)
WGRAD_BOX = os.environ.get("DIAGAN_WGRAD_BOX", "1") != "0"


def _r4(n):
    return (n + 3) // 4 * 4


class ConvLayer(nn.Module):
    """Conv2d / ConvTranspose2d (+ optional spectral norm) stored as packed Wp[Co_p][Kp].

    init: default PyTorch Conv init, then optional xavier_uniform_(gain) exactly like
    torch_mimicry's blocks (RNG consumption order: weight, bias, [xavier], [sn_u])."""

    def __init__(self, kind, in_ch, out_ch, ksize, stride=1, pad=0, bias=True, sn=False, xavier_gain=None):
        super().__init__()
        self.kind, self.in_ch, self.out_ch, self.ksize = kind, in_ch, out_ch, ksize
        self.Cip, self.Cop = _r4(in_ch), _r4(out_ch)
        self.geom = C.Geom(kind, self.Cip, self.Cop, ksize, ksize, stride, pad)
        self.sn = sn
        ref = (nn.Conv2d if kind == 'conv' else nn.ConvTranspose2d)(in_ch, out_ch, ksize, stride, pad, bias=bias)
        self._oihw_shape = tuple(ref.weight.shape)
        self.weight = nn.Parameter(self._pack(ref.weight.data))
        if bias:
            b = torch.zeros(self.Cop)
            b[:out_ch] = ref.bias.data
            self.bias = nn.Parameter(b)
        else:
            self.bias = None
        if sn:
            self.register_buffer('sn_u', torch.randn(1, out_ch))
            self.register_buffer('sn_sigma', torch.ones(1))
        self._wd = None
        self._wd_version = -1
        self._register_state_dict_hook(self._sd_hook)
        self._register_load_state_dict_pre_hook(self._load_hook)
        if xavier_gain is not None:
            self.xavier_(xavier_gain)

    def xavier_(self, gain):
        """nn.init.xavier_uniform_(weight, gain) on the reference-shaped tensor, then re-pack.  Kept a
        separate step so blocks can replay torch_mimicry's RNG order (convs created, then xavier)."""
        w = torch.empty(self._oihw_shape)
        nn.init.xavier_uniform_(w, gain)
        self.weight.data.copy_(self._pack(w).to(self.weight.device))
        return self

    # ---- layout conversion (checkpoint boundary only) ----
    def _pack(self, w):
        if self.kind == 'convT':
            w = w.permute(1, 0, 2, 3)
        Co, Ci, R, S = w.shape
        wp = w.permute(0, 2, 3, 1)
        wp = torch.nn.functional.pad(wp, (0, self.Cip - Ci))
        out = torch.zeros((self.Cop, self.geom.Kp), dtype=w.dtype, device=w.device)
        out[:Co, : R * S * self.Cip] = wp.reshape(Co, -1)
        return out

    def _unpack(self, wp):
        R = self.ksize
        w = wp[: self.out_ch, : R * R * self.Cip].reshape(self.out_ch, R, R, self.Cip)[..., : self.in_ch]
        w = w.permute(0, 3, 1, 2)
        if self.kind == 'convT':
            w = w.permute(1, 0, 2, 3)
        return w.contiguous()

    @staticmethod
    def _sd_hook(module, sd, prefix, local_metadata):
        sd[prefix + 'weight'] = module._unpack(sd[prefix + 'weight'])
        if module.bias is not None:
            sd[prefix + 'bias'] = sd[prefix + 'bias'][: module.out_ch].clone()
        return sd

    def _load_hook(self, sd, prefix, *args):
        k = prefix + 'weight'
        if k in sd and sd[k].dim() == 4:
            sd[k] = self._pack(sd[k].to(torch.float32))
        k = prefix + 'bias'
        if k in sd and sd[k].numel() == self.out_ch and self.out_ch != self.Cop:
            b = torch.zeros(self.Cop, dtype=torch.float32, device=sd[k].device)
            b[: self.out_ch] = sd[k]
            sd[k] = b
        net = getattr(self, '_net', None)
        if net is not None:
            net.param_version += 1

    # ---- compute ----
    class Ctx:
        __slots__ = ("wf", "wd", "u", "v", "state", "_keep", "row_scale", "pair", "wkey", "wver")

        def __init__(self):
            self.row_scale = None      # (inv_sigma0, inv_sigma1): two forwards batched into one GEMM
            self.pair = None           # their two per-forward SN contexts
            self.wkey = None           # which of the layer's operand sets this is: None (plain layer), slot number, 'pair'
            self.wver = None           # version of the operands wf / wd (what the batched Winograd transforms were made from)

    # ---- Winograd weights transformed ahead, many layers per launch (ops/conv.py: WinoWeightBatch) ----
    def _wsite(self, ctx, role):
        """(site, version) of this layer's forward ('f') / data-gradient ('d') launches with the operand set `ctx`, or
        (None, None) where no Winograd kernel can run (only 3x3 / stride 1 / pad 1 convolutions with Ci % 8 == 0 have one)"""
        net = getattr(self, '_net', None)
        g = self.geom
        if (net is None or not C.WINO_BATCH or ctx.wver is None or g.kind != 'conv' or g.R != 3 or g.S != 3 or g.stride != 1
                or g.pad != 1 or (g.Ci if role == 'f' else g.Co) % 8):
            return None, None
        sites = self.__dict__.setdefault('_wsites', {})
        key = (role, ctx.wkey, id(ctx) if self.sn else 0)
        site = sites.get(key)
        if site is None:
            batch = net.wino_batch((role, ctx.wkey))
            if self.sn:
                w_of = (lambda c=ctx: c.wf) if role == 'f' else (lambda c=ctx: c.wd)
            else:
                w_of = (lambda m=self: m.weight.data) if role == 'f' else (lambda m=self: m._wd)
            dims = (g.Co, g.Ci, g.Kp) if role == 'f' else (g.Ci, g.Co, g.Kd)
            site = sites[key] = batch.site(w_of, *dims)
        return site, ctx.wver

    def prepare(self, training, need_dgrad=True, slot=None):
        """Per-forward operand preparation.  SN layers: one power iteration + scaled packing (done for
        the whole network at once by SNBatch when `slot` is given)."""
        if self.sn and slot == 'pair':
            return self._pair_ctx
        if self.sn and slot is not None:
            return self._slot_ctx[slot]
        ctx = ConvLayer.Ctx()
        g = self.geom
        dev = self.weight.device
        if self.sn:
            u_buf = self.sn_u.view(-1)
            if self.Cop != self.out_ch:
                raise RuntimeError("spectral norm with padded output channels is not supported")
            ctx.u, ctx.v, ctx.state = C.sn_power_iter(self.weight.data, u_buf, self.sn_sigma, training=training)
            alloc = torch.zeros if g.Kp != g.R * g.S * g.Ci else torch.empty
            ctx.wf = alloc((g.Co, g.Kp), dtype=torch.float32, device=dev)
            ctx.wd = None
            if need_dgrad:
                allocd = torch.zeros if g.Kd != g.R * g.S * g.Co else torch.empty
                ctx.wd = allocd((g.Ci, g.Kd), dtype=torch.float32, device=dev)
            C.pack_weights(self.weight.data, g.Co, g.Ci, g.R * g.S, g.Kp, g.Kd, inv_sigma=ctx.state[1:],
                           Wf=ctx.wf, Wd=ctx.wd)
        else:
            ctx.u = ctx.v = ctx.state = None
            ctx.wf = self.weight.data
            ctx.wd = None
            ctx.wver = self._net.param_version if getattr(self, '_net', None) is not None else None
            if need_dgrad:
                ver = self._net.param_version if getattr(self, '_net', None) is not None else -2
                if self._wd is None or self._wd_version != ver or ver == -2:
                    if self._wd is None or self._wd.device != dev:
                        self._wd = torch.zeros((g.Ci, g.Kd), dtype=torch.float32, device=dev)
                    C.pack_weights(self.weight.data, g.Co, g.Ci, g.R * g.S, g.Kp, g.Kd, Wd=self._wd)
                    self._wd_version = ver
                ctx.wd = self._wd
        return ctx

    def _slot_ctx_or_none(self, slot):
        sc = getattr(self, '_slot_ctx', None)
        return sc[slot] if (sc is not None and isinstance(slot, int)) else None

    def _res_up(self, x, residual, res_up, want_stats, tile_cfg=0, pro=None):
        """res_up: `residual` is at half resolution and its bilinear x2 is to be added.  The Winograd kernel blends it in
        its epilogue; launches that take another kernel get the up-sampled tensor."""
        if not res_up:
            return residual, False
        group_imgs = pro[3] if (pro is not None and len(pro) > 3) else 0
        fused = tile_cfg in (9, 13) or (tile_cfg == 0 and C.res_up_fused(self.geom, x.shape[0], x.shape[1], x.shape[2],
                                                                  want_stats=want_stats, group_imgs=group_imgs))
        return (residual, True) if fused else (E.upsample2x(residual), False)

    def upin_fused(self, x, pro=None):
        """Will fwd / fwd_bn(..., up_in=True) on the half-resolution `x` run as one launch (the bilinear x2 folded into the
        F(4x4) kernel's input transform)?  Otherwise the caller up-samples first (E.upsample2x)."""
        group_imgs = pro[3] if (pro is not None and len(pro) > 3) else 0
        return (self.geom.kind == 'conv' and os.environ.get("DIAGAN_UPIN", "1") != "0"
                and C.upin_fused(self.geom, x.shape[0], x.shape[1], x.shape[2], group_imgs=group_imgs))

    def fwd(self, ctx, x, pro=None, residual=None, res_relu=False, tile_cfg=0, res_up=False, up_in=False):
        residual, res_up = self._res_up(x, residual, res_up, False, tile_cfg, pro)
        ws, wv = self._wsite(ctx, 'f')
        return C.conv_fwd(self.geom, x, ctx.wf, bias=None if self.bias is None else self.bias.data,
                          residual=residual, pro=pro, tile_cfg=tile_cfg, res_relu=res_relu, row_scale=ctx.row_scale,
                          res_up=res_up, up_in=up_in, wsite=ws, wversion=wv)

    def fwd_pool(self, ctx, x, pro=None, residual=None):
        """avg_pool2d(conv(pro(x)) + bias, 2) + residual (the end of a down-sampling DBlock): ONE launch on 9/16 of the
        Winograd products where the layer qualifies (C.pool_fused), else the convolution followed by diagan_avgpool2."""
        if C.pool_fused(self.geom, x.shape[0], x.shape[1], x.shape[2], pro):
            ws, wv = self._wsite(ctx, 'f')
            return C.conv_fwd(self.geom, x, ctx.wf, bias=None if self.bias is None else self.bias.data, residual=residual,
                              pro=pro, row_scale=ctx.row_scale, pool=True, wsite=ws, wversion=wv)
        return E.avgpool2(self.fwd(ctx, x, pro=pro), residual=residual)

    def fwd_bn(self, ctx, x, bn, training, pro=None, residual=None, groups=1, res_up=False, up_in=False):
        """Forward + the BatchNorm statistics of the layer that consumes the output, taken from the GEMM
        epilogue's per-tile sums (no second pass over the activation).  Returns (y, bn context).
        groups > 1: the batch is `groups` stacked batches with separate statistics (tiles never straddle groups)."""
        if not training:
            y = self.fwd(ctx, x, pro=pro, residual=residual, res_up=res_up, up_in=up_in)
            return y, bn.stats(y, False)
        if os.environ.get("DIAGAN_FUSED_BN_STATS", "1") == "0":      # diagnostic: statistics by a separate pass over y
            y = self.fwd(ctx, x, pro=pro, residual=residual, res_up=res_up, up_in=up_in)
            return y, bn.stats(y, True, groups=groups)
        residual, res_up = self._res_up(x, residual, res_up, True, pro=pro)
        ws, wv = self._wsite(ctx, 'f')
        y, stats = C.conv_fwd(self.geom, x, ctx.wf, bias=None if self.bias is None else self.bias.data,
                              residual=residual, pro=pro, row_scale=ctx.row_scale, want_stats=True, res_up=res_up, up_in=up_in,
                              wsite=ws, wversion=wv)
        M = y.numel() // y.shape[-1]
        if stats is None or stats[1] % groups or (M // groups) % (M // stats[1]):
            return y, bn.stats(y, True, groups=groups)
        return y, bn.stats_fused(stats[0], stats[1], M, groups=groups, group_imgs=y.shape[0] // groups)

    def dgrad_unpool_fused(self, B, in_hw):
        return C.unpool_fused(self.geom, B, in_hw[0], in_hw[1])

    def dgrad_unpool(self, ctx, dy_pooled, in_hw, residual=None, mask_src=None, mask_slope=0.0):
        """conv^T(avg_pool2d_backward(dy_pooled)): the data-gradient of a layer whose output was average-pooled, from the
        pooled gradient in one launch (callers check dgrad_unpool_fused first)"""
        ws, wv = self._wsite(ctx, 'd')
        return C.conv_dgrad(self.geom, dy_pooled, ctx.wd, in_hw, residual=residual, mask_src=mask_src,
                            mask_slope=mask_slope, row_scale=ctx.row_scale, unpool=True, wsite=ws, wversion=wv)

    def dgrad(self, ctx, dy, in_hw, residual=None, mask_src=None, mask_slope=0.0, res_unpool=False):
        """res_unpool: `residual` is a half-resolution tensor whose avg_pool2d_backward is added (callers check
        dgrad_res_unpool_fused first)"""
        ws, wv = self._wsite(ctx, 'd')
        return C.conv_dgrad(self.geom, dy, ctx.wd, in_hw, residual=residual, mask_src=mask_src,
                            mask_slope=mask_slope, row_scale=ctx.row_scale, wsite=ws, wversion=wv, res_unpool=res_unpool)

    def dgrad_res_unpool_fused(self, B, in_hw):
        return C.res_unpool_fused(self.geom, B, in_hw[0], in_hw[1])

    def wgrad_pooled_ok(self, dy_pooled, x):
        """The weight gradient of avg_pool2d(conv3x3(act(x)), 2) from the POOLED gradient: a 3x3 / stride 2 / pad 0 weight
        gradient over the (H+1) x (W+1) image of 2x2 box sums of act(x) -- a quarter of the multiply-accumulates, no
        transform.  Worth the box-sum pass from ~1 M input values on (smaller layers keep the up-sampled gradient)."""
        g = self.geom
        return (g.kind == 'conv' and g.R == 3 and g.S == 3 and g.stride == 1 and g.pad == 1 and x.shape[1] % 2 == 0
                and x.shape[2] % 2 == 0 and x.numel() >= (1 << 20)
                and (dy_pooled.numel() // dy_pooled.shape[-1]) % 64 == 0
                and os.environ.get("DIAGAN_WGRAD_POOLED", "1") != "0")

    def wgrad_pooled(self, ctx, dy_pooled, x, relu_in=False, slot=0):
        if getattr(self, '_geom_s2', None) is None:
            g = self.geom
            object.__setattr__(self, '_geom_s2', C.Geom('conv', g.Ci, g.Co, 3, 3, 2, 0))
        if WGRAD_BOX:      # the box sums are taken by the weight gradient's loader (round 5; same operands bit for bit)
            self.wgrad(ctx, dy_pooled, x, pro=(C.PRO_BOX_RELU if relu_in else C.PRO_BOX, None, None), slot=slot, geom=self._geom_s2)
        else:
            self.wgrad(ctx, dy_pooled, E.boxsum2(x, relu_in=relu_in), pro=None, slot=slot, geom=self._geom_s2)

    def wgrad(self, ctx, dy, x, pro=None, slot=0, geom=None):
        """Accumulates into weight.grad / bias.grad (views of the net's flat gradient buffer).
        Inside a network the split-K partials go to a per-layer slab and are reduced for all layers at
        once by WgradBatch.finish(); standalone layers reduce immediately."""
        net = getattr(self, '_net', None)
        if net is not None and self.sn and slot == 'pair':
            net.wgrad_batch.launch(self, slot, dy, x, pro, ctx.pair, segments=2, geom=geom)
            return
        if net is not None and (not self.sn or ctx is self._slot_ctx_or_none(slot)):
            net.wgrad_batch.launch(self, slot, dy, x, pro, ctx if self.sn else None, geom=geom)
            return
        sn = (self.weight.data, ctx.u, ctx.v, ctx.state) if self.sn else None
        C.conv_wgrad(geom if geom is not None else self.geom, dy, x, self.weight.grad, accumulate=True, pro=pro, sn=sn)
        if self.bias is not None:
            E.colsum(dy, self.bias.grad, accumulate=True)


class BatchNorm(nn.Module):
    """nn.BatchNorm2d parameters/buffers; statistics and backward run in elementwise.hip."""

    def __init__(self, ch, eps=1e-5, momentum=0.1):
        super().__init__()
        self.ch, self.eps, self.momentum = ch, eps, momentum
        self.weight = nn.Parameter(torch.ones(ch))
        self.bias = nn.Parameter(torch.zeros(ch))
        self.register_buffer('running_mean', torch.zeros(ch))
        self.register_buffer('running_var', torch.ones(ch))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))
        # the counter is only ever read through state_dict(): count on the host and fold the pending
        # increments into the buffer there (a device add per forward was 42 launches per global step)
        self._pending_batches = 0
        self.register_state_dict_pre_hook(BatchNorm._flush_counter)
        self._register_load_state_dict_pre_hook(self._drop_pending)

    @staticmethod
    def _flush_counter(module, prefix, keep_vars):
        if module._pending_batches:
            module.num_batches_tracked += module._pending_batches
            module._pending_batches = 0

    def _drop_pending(self, *args):
        self._pending_batches = 0

    def stats(self, x, training, groups=1):
        """groups > 1: `groups` batches stacked along dim 0, each normalised by its own statistics (forward only)."""
        if training:
            self._pending_batches += groups
        return E.bn_stats(x, self.weight.data, self.bias.data, self.running_mean, self.running_var, training,
                          self.eps, self.momentum, groups=groups if training else 1)

    def stats_fused(self, partials, tiles, M, groups=1, group_imgs=0):
        self._pending_batches += groups
        return E.bn_stats_fused(partials, tiles, M, self.weight.data, self.bias.data, self.running_mean,
                                self.running_var, self.eps, self.momentum, groups=groups, group_imgs=group_imgs)

    def bwd(self, g, x, ctx, relu, residual=None, slope=0.0, drop=None, drop_scale=1.0):
        return E.bn_bwd(g, x, ctx, relu, self.weight.grad, self.bias.grad, True, residual=residual, slope=slope,
                        drop=drop, drop_scale=drop_scale)


class LatentLinear(nn.Module):
    """Generator input layer nn.Linear(nz, bw*bw*ch) whose output is viewed [B, ch, bw, bw]
    (sngan l1; mnist.py:53 fc).  Stored as a 1x1 'conv' with its rows permuted to NHWC order so the
    GEMM writes the [B, bw, bw, ch] activation directly."""

    def __init__(self, nz, ch, bw, xavier_gain=None):
        super().__init__()
        self.nz, self.ch, self.bw = nz, ch, bw
        ref = nn.Linear(nz, bw * bw * ch)
        if xavier_gain is not None:
            nn.init.xavier_uniform_(ref.weight.data, xavier_gain)
        self.nzp = _r4(nz)
        self.geom = C.Geom('conv', self.nzp, bw * bw * ch, 1, 1, 1, 0)
        self.weight = nn.Parameter(self._pack(ref.weight.data))
        self.bias = nn.Parameter(self._perm(ref.bias.data.view(-1, 1)).view(-1).contiguous())
        self._register_state_dict_hook(self._sd_hook)
        self._register_load_state_dict_pre_hook(self._load_hook)

    def xavier_(self, gain):
        w = torch.empty(self.bw * self.bw * self.ch, self.nz)
        nn.init.xavier_uniform_(w, gain)
        self.weight.data.copy_(self._pack(w).to(self.weight.device))
        return self

    def _perm(self, w):      # rows (c, h, w) -> (h, w, c)
        return w.view(self.ch, self.bw * self.bw, -1).permute(1, 0, 2).reshape(self.ch * self.bw * self.bw, -1)

    def _unperm(self, w):
        return w.view(self.bw * self.bw, self.ch, -1).permute(1, 0, 2).reshape(self.ch * self.bw * self.bw, -1)

    def _pack(self, w):
        out = torch.zeros((w.shape[0], self.geom.Kp), dtype=w.dtype, device=w.device)
        out[:, : self.nz] = self._perm(w)
        return out

    @staticmethod
    def _sd_hook(module, sd, prefix, local_metadata):
        sd[prefix + 'weight'] = module._unperm(sd[prefix + 'weight'][:, : module.nz]).contiguous()
        sd[prefix + 'bias'] = module._unperm(sd[prefix + 'bias'].view(-1, 1)).view(-1).contiguous()
        return sd

    def _load_hook(self, sd, prefix, *args):
        k = prefix + 'weight'     # checkpoints always hold the reference layout [bw*bw*ch (c,h,w), nz]
        if k in sd:
            sd[k] = self._pack(sd[k].to(torch.float32))
        if prefix + 'bias' in sd:
            sd[prefix + 'bias'] = self._perm(sd[prefix + 'bias'].to(torch.float32).view(-1, 1)).view(-1).contiguous()
        net = getattr(self, '_net', None)
        if net is not None:
            net.param_version += 1

    def fwd(self, z):
        """z [B, nz] -> [B, bw, bw, ch]"""
        B = z.shape[0]
        if self.nzp != self.nz:
            z = torch.nn.functional.pad(z, (0, self.nzp - self.nz))
        x = z.contiguous().view(B, 1, 1, self.nzp)
        y = C.conv_fwd(self.geom, x, self.weight.data, bias=self.bias.data)
        return x, y.view(B, self.bw, self.bw, self.ch)

    sn = False

    def wgrad(self, x, dy, slot=0):
        B = x.shape[0]
        dy2 = dy.reshape(B, 1, 1, -1)
        net = getattr(self, '_net', None)
        if net is not None:
            net.wgrad_batch.launch(self, slot, dy2, x, None, None)
            return
        C.conv_wgrad(self.geom, dy2, x, self.weight.grad, accumulate=True)
        E.colsum(dy2, self.bias.grad, accumulate=True)


class HeadLinear(nn.Module):
    """SNLinear(C, 1) (or nn.Linear) on the globally pooled feature: the discriminator logit."""

    def __init__(self, in_ch, sn=True, xavier_gain=None):
        super().__init__()
        self.in_ch, self.sn = in_ch, sn
        ref = nn.Linear(in_ch, 1)
        if xavier_gain is not None:
            nn.init.xavier_uniform_(ref.weight.data, xavier_gain)
        self.weight = nn.Parameter(ref.weight.data.clone())      # [1, C] (already "packed")
        self.bias = nn.Parameter(torch.cat([ref.bias.data, torch.zeros(3)]))   # padded to 4 floats
        if sn:
            self.register_buffer('sn_u', torch.randn(1, 1))
            self.register_buffer('sn_sigma', torch.ones(1))
        self._register_state_dict_hook(self._sd_hook)
        self._register_load_state_dict_pre_hook(self._load_hook)

    def xavier_(self, gain):
        w = torch.empty(1, self.in_ch)
        nn.init.xavier_uniform_(w, gain)
        self.weight.data.copy_(w.to(self.weight.device))
        return self

    @staticmethod
    def _sd_hook(module, sd, prefix, local_metadata):
        sd[prefix + 'bias'] = sd[prefix + 'bias'][:1].clone()
        return sd

    def _load_hook(self, sd, prefix, *args):
        k = prefix + 'bias'
        if k in sd and sd[k].numel() == 1:
            sd[k] = torch.cat([sd[k].to(torch.float32), torch.zeros(3, device=sd[k].device)])

    class Ctx:
        __slots__ = ("u", "v", "state", "x", "pooled", "_keep", "pair")

        def __init__(self):
            self.pair = None

    def fwd(self, x, training, slot=None):
        ctx = HeadLinear.Ctx()
        inv = None
        if self.sn and slot == 'pair':
            p0, p1 = self._slot_ctx[0], self._slot_ctx[1]
            ctx.pair = (p0, p1)
            ctx.x = x
            ctx.pooled, logit = E.head_fwd(x, self.weight.data, p0.state[1:], self.bias.data, inv_sigma1=p1.state[1:])
            return ctx, logit
        if self.sn and slot is not None:
            pre = self._slot_ctx[slot]
            ctx.u, ctx.v, ctx.state = pre.u, pre.v, pre.state
            inv = ctx.state[1:]
        elif self.sn:
            ctx.u, ctx.v, ctx.state = C.sn_power_iter(self.weight.data, self.sn_u.view(-1), self.sn_sigma,
                                                      training=training)
            inv = ctx.state[1:]
        ctx.x = x
        ctx.pooled, logit = E.head_fwd(x, self.weight.data, inv, self.bias.data)
        return ctx, logit

    def bwd(self, ctx, dlogit, need_wgrad=True):
        if ctx.pair is not None:
            from diagan import _native as nat
            p0, p1 = ctx.pair
            gx, _, _ = E.head_bwd(dlogit, self.weight.data, p0.state[1:], ctx.x, ctx.pooled, need_gx=True,
                                  need_wgrad=False, inv_sigma1=p1.state[1:])
            if need_wgrad:
                h = dlogit.numel() // 2
                for p, sl in ((p0, slice(0, h)), (p1, slice(h, 2 * h))):
                    _, G, dot = E.head_bwd(dlogit[sl], self.weight.data, None, ctx.x[sl], ctx.pooled[sl], need_gx=False,
                                           need_wgrad=True, dbias=self.bias.grad)
                    nat.call("diagan_sn_grad_fix", nat.ptr(G), nat.ptr(dot), 1, nat.ptr(p.u), nat.ptr(p.v),
                             nat.ptr(p.state), nat.ptr(self.weight.grad), 1, self.in_ch, 1, nat.current_stream())
            return gx
        inv = ctx.state[1:] if self.sn else None
        gx, G, dot = E.head_bwd(dlogit, self.weight.data, inv, ctx.x, ctx.pooled, need_gx=True,
                                need_wgrad=need_wgrad, dbias=self.bias.grad if need_wgrad else None)
        if need_wgrad:
            if self.sn:
                from diagan import _native as nat
                nat.call("diagan_sn_grad_fix", nat.ptr(G), nat.ptr(dot), 1, nat.ptr(ctx.u), nat.ptr(ctx.v),
                         nat.ptr(ctx.state), nat.ptr(self.weight.grad), 1, self.in_ch, 1, nat.current_stream())
            else:
                E.add(self.weight.grad.view(-1), G, out=self.weight.grad.view(-1))
        return gx


class SNBatch:
    """All spectral-norm layers of one network prepared in 4 launches (diagan_sn_prepare_batched).

    Two context slots per layer: a D update runs two forwards (real, fake) before its backward, and each
    forward's (u, v, sigma, Wf, Wd) must survive until that forward's backward."""

    def __init__(self, net, layers, n_slots=2):
        import numpy as np
        from diagan import _native as nat
        self.net, self.layers, self.nat = net, layers, nat
        dev = net.flat_params.device
        self.slab_generation = net.slab_generation
        desc = np.dtype([('p', np.uint64, 9), ('i', np.int32, 6)])
        self.tables = []
        f32 = dict(dtype=torch.float32, device=dev)
        dims = []
        for slot in range(n_slots):
            tab = np.zeros(len(layers), dtype=desc)
            for li, m in enumerate(layers):
                if isinstance(m, ConvLayer):
                    g = m.geom
                    Co, Ci, RS, Kp, Kd = g.Co, g.Ci, g.R * g.S, g.Kp, g.Kd
                    wf, wd = torch.zeros((Co, Kp), **f32), torch.zeros((Ci, Kd), **f32)
                    ctx = ConvLayer.Ctx()
                else:                                   # HeadLinear: [1][C], used with inv_sigma only
                    Co, Ci, RS, Kp, Kd = 1, m.in_ch, 1, m.in_ch, 0
                    wf = wd = None
                    ctx = HeadLinear.Ctx()
                ctx.u, ctx.v, ctx.state = torch.zeros(Co, **f32), torch.zeros(Kp, **f32), torch.ones(2, **f32)
                if isinstance(m, ConvLayer):
                    ctx.wf, ctx.wd, ctx.wkey = wf, wd, slot
                work = torch.zeros(8 * Kp + Co, **f32)
                ctx._keep = (work,)
                if not hasattr(m, '_slot_ctx') or m._slot_ctx is None or len(m._slot_ctx) != n_slots:
                    m._slot_ctx = [None] * n_slots
                m._slot_ctx[slot] = ctx
                ptrs = [m.weight.data.data_ptr(), m.sn_u.data_ptr(), m.sn_sigma.data_ptr(), ctx.u.data_ptr(),
                        ctx.v.data_ptr(), ctx.state.data_ptr(), work.data_ptr(),
                        wf.data_ptr() if wf is not None else 0, wd.data_ptr() if wd is not None else 0]
                tab[li]['p'] = ptrs
                tab[li]['i'] = [Co, Ci, RS, Kp, Kd, 0]
                dims.append((Co, Ci, RS, Kp))
            self.tables.append(torch.from_numpy(tab.view(np.uint8).copy()).to(dev))
        self.max = [max(d[k] for d in dims) for k in range(4)]
        # "pair" mode: two forwards (different sigma) batched into one GEMM on the UN-normalised weight:
        # forward operand = master weight, data-gradient operand = un-normalised transpose pack (made
        # once per parameter update), per-forward 1/sigma applied in the GEMM epilogue.
        convs = [m for m in layers if isinstance(m, ConvLayer)]
        self.unit = torch.ones(2, **f32)
        ptab = np.zeros(len(convs), dtype=desc)
        for li, m in enumerate(convs):
            g = m.geom
            wd_raw = torch.zeros((g.Ci, g.Kd), **f32)
            pc = ConvLayer.Ctx()
            pc.wf, pc.wd = m.weight.data, wd_raw
            pc.u = pc.v = pc.state = None
            pc.row_scale = (m._slot_ctx[0].state[1:], m._slot_ctx[1].state[1:])
            pc.pair = (m._slot_ctx[0], m._slot_ctx[1])
            pc.wkey = 'pair'
            m._pair_ctx = pc
            ptab[li]['p'] = [m.weight.data.data_ptr(), 0, 0, 0, 0, self.unit.data_ptr(), 0, 0, wd_raw.data_ptr()]
            ptab[li]['i'] = [g.Co, g.Ci, g.R * g.S, g.Kp, g.Kd, 0]
        self.pair_table = torch.from_numpy(ptab.view(np.uint8).copy()).to(dev)
        self.n_convs = len(convs)
        self.pair_version = None
        self.convs = convs
        self.runs = 0                 # operand-set versions: every run() that re-packs W / sigma of its slot makes a new one
        self.slot_ver = {}

    def stale(self):
        self.net.flat_params                      # (builds the slabs if they do not exist yet)
        return self.slab_generation != self.net.slab_generation

    def run_pair(self, training, need_dgrad):
        """Power iterations of BOTH forwards (u is advanced twice, as two sequential forwards would),
        no operand packing; the shared un-normalised data-gradient operand is refreshed when stale."""
        self.run(0, training, -1)
        self.run(1, training, -1)
        if need_dgrad and self.pair_version != self.net.param_version:
            nat = self.nat
            nat.call("diagan_pack_batched", self.pair_table.data_ptr(), self.n_convs, self.max[0], self.max[1],
                     self.max[2], 1, nat.current_stream())
            self.pair_version = self.net.param_version
        for m in self.convs:          # pair mode runs on the un-normalised master weight: its operands change with the parameters
            m._pair_ctx.wver = ('p', self.net.param_version)

    def run(self, slot, training, write_wd):
        nat = self.nat
        nat.call("diagan_sn_prepare_batched", self.tables[slot].data_ptr(), len(self.layers), self.max[0],
                 self.max[1], self.max[2], self.max[3], 1e-12, 1 if training else 0,
                 write_wd if isinstance(write_wd, int) and not isinstance(write_wd, bool) else (1 if write_wd else 0),
                 nat.current_stream())
        if write_wd != -1:            # (-1: power iteration only, pair mode) W / sigma of this slot was re-packed
            self.runs += 1
            self.slot_ver[slot] = ('s', self.runs)
            for m in self.convs:
                m._slot_ctx[slot].wver = self.slot_ver[slot]


class WgradBatch:
    QUEUE_BYTES = int(os.environ.get("DIAGAN_WGRAD_QUEUE_BYTES", str(8 << 30)))     # early-flush threshold of a slot's queue
    """Deferred weight-gradient epilogue of one network.

    Every parameterised GEMM layer owns one split-K slab per context slot; conv_wgrad_kernel writes its
    partials (weights and, fused, the bias column sums) there during the backward pass, and finish()
    reduces all layers in two launches (diagan_wgrad_finish_batched), including the spectral-norm
    correction.  Replaces 4-5 small launches per layer per pass."""

    def __init__(self, net):
        self.net = net
        self.entries = {}        # (layer, slot) -> dict
        self.launched = {}       # slot -> list of layers launched in the current pass
        self.tables = {}         # (slot, tuple(layer ids)) -> (table tensor, n, total_blocks, any_sn)
        self.slab_generation = None
        # data parallelism (FlatNet.sync_grads): while `hold` is set, finish() only notes the slot; the reduction then
        # runs in two halves -- late layers first -- so that the exchange of the first half's slab range is in flight
        # under the second half's reduction (DDP's bucketed overlap, stylegan2/train_ffhq.py:572-585, for a flat slab)
        self.hold = False
        self.pending = []
        self.overlapped = 0      # updates whose reduction was split (tests)
        # Winograd weight gradients wait here until the end of the pass and then run as ONE launch per prologue mode
        # (ops/conv.py: conv_wgrad_batched); the queue keeps dy and x alive.  Memory: every batchable layer's dy and x stay
        # allocated until finish() instead of being freed as the backward pass moves on -- at most the activations + gradients
        # of one pass (SNGAN-64 at batch 64: ~1.3 GB, StyleGAN2 does not use this queue), bounded by QUEUE_BYTES below: a slot
        # whose queue passes it is flushed early (the layers so far launch as their own batch)
        self.queue = {}          # slot -> [(layer, dy, x, pro, segments, entry)]
        self.queued_bytes = {}   # slot -> bytes of dy + x the queue keeps alive
        self.batch_plans = {}    # (layer ids, shapes) -> splits per layer
        self.job_tables = {}     # (the same key, plan) -> constant columns of the batched launch's job table (ops/conv.py); lives and
                                 # dies with this network, so the layer ids in its keys cannot be recycled under it
        self.batched_launches = 0

    def _entry(self, layer, slot, M, segments=1, dy_shape=None, x_shape=None, geom=None):
        self.net.flat_grads
        if self.slab_generation != self.net.slab_generation:          # gradient slab was re-allocated
            self.entries.clear(), self.tables.clear(), self.job_tables.clear()
            self.slab_generation = self.net.slab_generation
        e = self.entries.get((layer, slot))
        if e is None or e['M'] != M:
            g = geom if geom is not None else layer.geom       # (an equivalent geometry with the same packed layout)
            n_w = g.Co * g.Kp
            has_bias = layer.bias is not None
            n_b = layer.bias.numel() if has_bias else 0
            if has_bias and layer.bias.grad.data_ptr() != layer.weight.grad.data_ptr() + 4 * n_w:
                raise RuntimeError("bias gradient does not follow the weight gradient in the flat slab")
            stride = n_w + n_b
            if dy_shape is not None and x_shape is not None:
                base = C.wgrad_splits_geom(g, dy_shape[0], x_shape[1], x_shape[2], dy_shape[1], dy_shape[2])
            else:
                base = C.wgrad_splits(M, g.Co, g.Kp)
            splits = max(segments, base // segments * segments)
            if segments == 1 and dy_shape is not None and C.small_co_wgrad(g):
                splits = C.small_co_wgrad_splits(dy_shape[0], dy_shape[1])
            dev = layer.weight.device
            e = dict(M=M, n_w=n_w, n_elem=stride, stride=stride, splits=splits, own_splits=splits, bias_off=n_w if has_bias else -1,
                     segments=segments,
                     slab=torch.empty(splits * stride, dtype=torch.float32, device=dev),
                     partials=torch.empty((segments, (stride + 1023) // 1024 + 1), dtype=torch.float64, device=dev))   # (>= blocks of any size + 1)
            self.entries[(layer, slot)] = e
            self.tables = {k: v for k, v in self.tables.items() if k[0] != slot}
        return e

    def launch(self, layer, slot, dy, x, pro, sn_ctx, segments=1, geom=None):
        """sn_ctx: None (plain layer), one SN context, or a tuple of `segments` contexts (one per
        batched forward; the pixel range is cut accordingly and each part gets its own correction).
        geom: gather geometry of THIS launch when it differs from the layer's (ConvLayer.wgrad_pooled)."""
        M = dy.numel() // dy.shape[-1]
        if self.hold and slot in self.pending:
            # a second backward pass into a slot whose reduction is being held back (a network that runs real and fake
            # through the same slot, e.g. MNIST_DCGAN_Discriminator): conv_wgrad_into WRITES the slab, so the first pass
            # has to be reduced into the gradient before its partials are overwritten
            self.pending.remove(slot)
            self.flush(slot)
            self._finish_layers(slot, self.launched.pop(slot, []))
        xs = C.wg_x_shape(x, pro)
        e = self._entry(layer, slot, M, segments, tuple(dy.shape), xs, geom)
        e['sn_ctx'] = sn_ctx
        self.launched.setdefault(slot, []).append(layer)
        if C.WGRAD_BATCH and dy.dim() == 4 and x.dim() == 4:
            g = geom if geom is not None else layer.geom
            cls = 0 if (segments == 1 and C.small_co_wgrad(g)) else C.wgrad_batch_class(
                g, xs[1], xs[2], dy.shape[1], dy.shape[2], int(pro[0]) if pro is not None else 0)
            if cls:
                q = self.queue.setdefault(slot, [])
                q.append((layer, dy, x, pro, segments, e, g, (cls, geom is not None)))
                self.queued_bytes[slot] = self.queued_bytes.get(slot, 0) + 4 * (dy.numel() + x.numel())
                if self.queued_bytes[slot] > self.QUEUE_BYTES:
                    self.flush(slot)
                return
        C.conv_wgrad_into(geom if geom is not None else layer.geom, dy, x, e['slab'], e['splits'], e['stride'], e['bias_off'],
                          pro=pro, segments=segments, pooled=geom is not None)

    def flush(self, slot):
        """launch the queued weight gradients of `slot`: one launch per kernel template (batch class; the pooled layers'
        strided form apart, for the FLOP accounting) and per wgrad_batch_max() layers, every layer with the split count the
        group's plan gives it; a group of one runs as an ordinary launch"""
        jobs = self.queue.pop(slot, None)
        self.queued_bytes.pop(slot, None)
        if not jobs:
            return
        groups = {}
        for job in jobs:
            groups.setdefault(job[7], []).append(job)
        nmax = C.wgrad_batch_max()

        def alone(job, pooled):
            layer, dy, x, pro, segments, e, g, _ = job
            if e['splits'] != e['own_splits']:       # (it ran in a batch before: back to its own chip-filling split count)
                e['splits'] = e['own_splits']
                e['slab'] = torch.empty(e['splits'] * e['stride'], dtype=torch.float32, device=dy.device)
                self.tables = {k: v for k, v in self.tables.items() if k[0] != slot}
            C.conv_wgrad_into(g, dy, x, e['slab'], e['splits'], e['stride'], e['bias_off'], pro=pro, segments=segments, pooled=pooled)

        for (cls, pooled), grp in groups.items():
            # (knob: a layer whose own chip-filling launch runs >= WGRAD_BATCH_MIN_STEPS K-steps per workgroup launches on its
            #  own; default off -- ops/conv.py has the sweep: the generator's three big layers are 0.92 -> 1.03 ms slower
            #  batched, and batching them still wins end to end through the slabs and launches it saves)
            small = []
            for job in grp:
                layer, dy, x, pro, segments, e, g, _ = job
                steps = C.wgrad_batch_shape(g, dy.shape[0], dy.shape[1], dy.shape[2], cls)[1]
                if -(-steps // max(e['own_splits'], 1)) >= C.WGRAD_BATCH_MIN_STEPS:
                    alone(job, pooled)
                else:
                    small.append(job)
            grp = small
            for lo in range(0, len(grp), nmax):
                part = grp[lo: lo + nmax]
                if len(part) == 1:
                    alone(part[0], pooled)
                    continue
                key = (cls, pooled) + tuple((id(j[0]), tuple(j[1].shape), j[4]) for j in part)
                plan = self.batch_plans.get(key)
                if plan is None:
                    desc, slots, fixed = [], 256, 8.0
                    for layer, dy, x, pro, segments, e, g, _ in part:
                        tiles, steps, slots, fixed = C.wgrad_batch_shape(g, dy.shape[0], dy.shape[1], dy.shape[2], cls)
                        desc.append((tiles, steps, segments))
                    plan = self.batch_plans[key] = C.batched_wgrad_splits(desc, slots, fixed)
                for (layer, dy, x, pro, segments, e, g, _), sp in zip(part, plan):
                    if e['splits'] != sp:                # the slab of this layer takes the batch's split count
                        e['splits'] = sp
                        e['slab'] = torch.empty(sp * e['stride'], dtype=torch.float32, device=dy.device)
                        self.tables = {k: v for k, v in self.tables.items() if k[0] != slot}
                name = None
                if cls >= 1000:
                    g0, d0 = part[0][6], part[0][1]
                    name = C._wgrad_kernel_name(g0.Co, g0.Kp, int(part[0][3][0]) if part[0][3] is not None else 0, d0.shape[1],
                                                d0.shape[2]).replace("conv_wgrad_kernel", "conv_wgrad_batched_kernel")
                    if pooled:
                        name += C.POOLED_TAG
                C.conv_wgrad_batched([(g, dy, x, e['slab'], e['splits'], e['stride'], e['bias_off'], pro, segments)
                                      for layer, dy, x, pro, segments, e, g, _ in part], key=(key, tuple(plan)),
                                     kernel_name=name, flop_scale=4.0 if pooled else 1.0, cache=self.job_tables)
                self.batched_launches += 1

    def finish(self, slot):
        self.flush(slot)             # the partial sums are computed now, whatever happens to their reduction
        if self.hold:
            self.pending.append(slot)
            return
        self._finish_layers(slot, self.launched.pop(slot, []))

    def split_for_overlap(self, slot):
        """(late layers, early layers, cut) of the layers launched in `slot` -- in launch order the backward pass reaches
        the LAST layers of the network first -- such that every late layer's gradient lies in flat_grads[cut:], every early
        one's in flat_grads[:cut], and the late part holds about half of the weight-gradient elements; None when the
        launch order does not partition the slab that way (then the reduction stays in one piece)."""
        layers = self.launched.get(slot, [])
        if len(layers) < 2:
            return None
        base = self.net.flat_grads.data_ptr()
        lo = [(l.weight.grad.data_ptr() - base) // 4 for l in layers]
        hi = [o + self.entries[(l, slot)]['n_elem'] for o, l in zip(lo, layers)]
        total, acc, best = sum(b - a for a, b in zip(lo, hi)), 0, None
        for h in range(1, len(layers)):            # late = layers[:h]: a valid cut leaves every early layer below it
            acc += hi[h - 1] - lo[h - 1]
            cut = min(lo[:h])
            if cut % 4 == 0 and all(b <= cut for b in hi[h:]):
                score = abs(2 * acc - total)
                if best is None or score < best[0]:
                    best = (score, h, cut)
        if best is None:
            return None
        _, h, cut = best
        return layers[:h], layers[h:], cut

    def _finish_layers(self, slot, layers):
        import numpy as np
        from diagan import _native as nat
        if not layers:
            return
        key = (slot, tuple(id(l) for l in layers))
        t = self.tables.get(key)
        if t is None:
            desc = np.dtype([('p', np.uint64, 12), ('stride', np.int64), ('i', np.int32, 6)])
            tab = np.zeros(len(layers), dtype=desc)
            any_sn, total_blocks = 0, 0
            for li, layer in enumerate(layers):
                e = self.entries[(layer, slot)]
                c = e['sn_ctx']
                ctxs = list(c) if isinstance(c, tuple) else [c]
                nctx = len(ctxs)
                per = e['splits'] // nctx                  # splits of each batched forward
                sn = ctxs[0] is not None
                any_sn |= int(sn)
                pp = [0] * 12
                for pi, cc in enumerate(ctxs):
                    pp[0 + pi] = e['slab'].data_ptr() + 4 * pi * per * e['stride']
                    if sn:
                        pp[2 + pi], pp[4 + pi], pp[6 + pi] = cc.u.data_ptr(), cc.v.data_ptr(), cc.state.data_ptr()
                    pp[8 + pi] = e['partials'][pi].data_ptr()
                if not sn and nctx > 1:                    # plain layer: one context over all splits
                    nctx, per = 1, e['splits']
                pp[10] = layer.weight.grad.data_ptr()
                pp[11] = layer.weight.data.data_ptr() if sn else 0
                assert 64 * e['stride'] < 2 ** 31, "wgrad finish: 16 splits of a layer must fit a 2 GiB buffer window"
                tab[li]['p'], tab[li]['stride'] = pp, e['stride']
                tab[li]['i'] = [per, e['n_elem'], e['n_w'], layer.geom.Kp, nctx, total_blocks]
                be = nat.fn("diagan_wgrad_finish_block_elems")(per)          # elements per workgroup of the finish kernels
                total_blocks += (e['n_elem'] + be - 1) // be
            t = (torch.from_numpy(tab.view(np.uint8).copy()).to(layers[0].weight.device), len(layers), total_blocks, any_sn)
            self.tables[key] = t
        nat.call("diagan_wgrad_finish_batched", t[0].data_ptr(), t[1], t[2], t[3], nat.current_stream())


class FlatNet(nn.Module):
    """Base of the engine's networks: owns the flat parameter / gradient buffers."""

    def __init__(self):
        super().__init__()
        self.param_version = 0        # bumped whenever parameters change (optimizer step, load)
        self.slab_generation = 0      # bumped whenever the flat slabs are re-allocated: descriptor tables that bake
                                      # raw pointers (SNBatch, WgradBatch) compare THIS, not id() of the tensors (ids
                                      # are recycled once the old slab is freed)
        self._flat = None
        self._flat_grad = None
        object.__setattr__(self, 'wgrad_batch', WgradBatch(self))
        object.__setattr__(self, '_wino_batches', {})

    def wino_batch(self, key):
        """the WinoWeightBatch of one part of a pass: key = ('f' | 'd', None | slot | 'pair')"""
        b = self._wino_batches.get(key)
        if b is None:
            b = self._wino_batches[key] = C.WinoWeightBatch()
        return b

    def _link_layers(self):
        for m in self.modules():
            if m is not self:
                object.__setattr__(m, '_net', self)

    def __deepcopy__(self, memo):
        """A copy must not inherit what was learned about the ORIGINAL's launches: the Winograd call sites close over the
        original's layers / contexts (their w_of would hand the copy the original's weights -- ADVICE r4) and the descriptor
        tables bake raw pointers into the original's slabs.  Sites and batches are dropped (re-learned on the copy's first
        pass, as after _apply) and the copy's slab generation moves on so that every baked table is rebuilt."""
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == 'wgrad_batch':            # (its entries hold per-layer slabs and queued dy / x of the ORIGINAL: a fresh one)
                continue
            new.__dict__[k] = {} if k == '_wino_batches' else (None if k in ('_flat', '_flat_grad') else copy.deepcopy(v, memo))
        new.__dict__['wgrad_batch'] = WgradBatch(new)
        for m in new.modules():
            if isinstance(m, ConvLayer):
                m.__dict__.pop('_wsites', None)
                m._wd = None
        # nn.Parameter.__deepcopy__ CLONES its data and drops .grad: the copy's parameters are no views of a slab any more.
        # Give the copy its own slabs (parameters re-pointed, gradients zero like a fresh network's) -- otherwise its fused
        # Adam / zero_grad / gradient exchange would act on a slab nobody reads
        if self._flat is not None:
            new._build_flat()
        new.slab_generation = self.slab_generation + 1
        return new

    def _build_flat(self):
        """(Re)allocate one contiguous fp32 slab for all parameters and one for all gradients and
        re-point every nn.Parameter (and its .grad) at 16-byte aligned views of them."""
        params = list(self.parameters())
        if not params:
            return
        dev = params[0].device
        total = sum(_r4(p.numel()) for p in params)
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        grad = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in params:
            n = p.numel()
            flat[off: off + n].copy_(p.data.reshape(-1))
            p.data = flat[off: off + n].view(p.shape)
            p.grad = grad[off: off + n].view(p.shape)
            off += _r4(n)
        self._flat, self._flat_grad = flat, grad
        self.param_version += 1
        self.slab_generation += 1

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn)
        self._build_flat()
        self._wino_batches.clear()
        for m in self.modules():
            if isinstance(m, ConvLayer):
                m._wd = None
                m.__dict__.pop('_wsites', None)
        return out

    def zero_grad(self, set_to_none=False):
        if self._flat_grad is None:
            self._build_flat()
        self._flat_grad.zero_()
        self.wgrad_batch.hold, self.wgrad_batch.pending = False, []
        self.wgrad_batch.queue.clear()
        self.wgrad_batch.queued_bytes.clear()

    @property
    def flat_params(self):
        if self._flat is None:
            self._build_flat()
        return self._flat

    @property
    def flat_grads(self):
        if self._flat_grad is None:
            self._build_flat()
        return self._flat_grad

    @property
    def device(self):
        return next(self.parameters()).device

    def sync_grads(self, optimizer=None, async_op=False):
        """Data-parallel gradient exchange: ONE all-reduce of the network's flat gradient slab over RCCL/xGMI
        (pattern: DistributedDataParallel in stylegan2/train_ffhq.py:572-585).  With the network's FusedAdam given, the
        ranks exchange the SUM and the optimiser applies 1/W as it reads the gradient (no extra pass over the slab);
        async_op leaves the collective in flight on the backend's stream until `optimizer.step()` waits for it, so the
        next update's forward / backward (phase 2: D_drs after D) runs beside it.
        When the train step held back the deferred weight-gradient reduction (`wgrad_batch.hold`), it runs here in two
        halves, late layers first, and the first half's slab range is exchanged while the second half is reduced."""
        from diagan.trainer import distributed as dist
        world = dist.get_world_size()
        wb = self.wgrad_batch
        held, wb.hold, wb.pending = wb.pending, False, []
        for slot in held[:-1]:
            wb.finish(slot)
        last = held[-1] if held else None
        if world == 1 or optimizer is None or not hasattr(optimizer, 'grad_scale'):
            if last is not None:
                wb.finish(last)
            if world > 1:
                dist.all_reduce_mean_(self.flat_grads)
            return
        optimizer.grad_scale = 1.0 / world
        parts = wb.split_for_overlap(last) if last is not None else None
        if parts is None:
            if last is not None:
                wb.finish(last)
            works = [dist.all_reduce_sum_(self.flat_grads, async_op=async_op)]
        else:
            # late layers reduced first; their slab range travels while the early layers are reduced
            late, early, cut = parts
            wb.launched.pop(last, None)
            flat = self.flat_grads
            wb._finish_layers(last, late)
            works = [dist.all_reduce_sum_(flat[cut:], async_op=True)]
            wb._finish_layers(last, early)
            works.append(dist.all_reduce_sum_(flat[:cut], async_op=True))
            wb.overlapped += 1
            if not async_op:
                for w in works:
                    if w is not None:
                        w.wait()
                works = []
        optimizer.pending = [w for w in works if w is not None] if async_op else None

    def export_grads(self):
        """Gradients in the reference's tensor shapes, keyed like state_dict() (tests, debugging)."""
        out = {}
        for name, m in self.named_modules():
            pre = name + '.' if name else ''
            if isinstance(m, ConvLayer):
                out[pre + 'weight'] = m._unpack(m.weight.grad)
                if m.bias is not None:
                    out[pre + 'bias'] = m.bias.grad[: m.out_ch].clone()
            elif isinstance(m, LatentLinear):
                out[pre + 'weight'] = m._unperm(m.weight.grad[:, : m.nz]).contiguous()
                out[pre + 'bias'] = m._unperm(m.bias.grad.view(-1, 1)).view(-1).contiguous()
            elif isinstance(m, HeadLinear):
                out[pre + 'weight'] = m.weight.grad.clone()
                out[pre + 'bias'] = m.bias.grad[:1].clone()
            elif isinstance(m, BatchNorm):
                out[pre + 'weight'] = m.weight.grad.clone()
                out[pre + 'bias'] = m.bias.grad.clone()
            elif hasattr(m, '_unperm') and hasattr(m, 'hw'):        # OutLinear
                out[pre + 'weight'] = m._unperm(m.weight.grad)
                out[pre + 'bias'] = m.bias.grad[:1].clone()
        return out

    def count_params(self):
        """Reference-visible parameter count (padding excluded)."""
        return sum(v.numel() for k, v in self.state_dict().items()
                   if not any(s in k for s in ('running_', 'num_batches', 'sn_u', 'sn_sigma')))
