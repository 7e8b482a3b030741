"""SNGAN generators / discriminators (32x32 CIFAR-10, 64x64 CelebA) on the HIP engine.

Architecture = torch_mimicry.nets.sngan.{SNGANGenerator32, SNGANDiscriminator32, SNGANGenerator64,
SNGANDiscriminator64} with its GBlock / DBlock / DBlockOptimized residual blocks, as selected by
the reference at diagan-pkg/diagan/models/predefined_models.py:17-92 (summarised in SURVEY §8
a2-a8; torch_mimicry itself is not vendored in the reference tree).  Module / parameter / buffer
names follow mimicry's so that state_dict keys line up (block2.c1.weight, block2.b1.running_mean,
block1.c_sc.sn_u, l5.sn_sigma, ...).

Each block has forward(x, training, save, ...) -> (out, ctx) and backward(ctx, g, ...) -> g_in with
the kernel fusion described in DESIGN.md (ReLU / BN-apply+ReLU in the conv loader, bias + shortcut
add in the conv epilogue, ReLU-backward mask in the dgrad epilogue).
"""
import math

import torch.nn as nn

from diagan.models.base import BaseDiscriminator, BaseGenerator
from diagan.models.layers import BatchNorm, ConvLayer, HeadLinear, LatentLinear, SNBatch
from diagan.ops import conv as C
from diagan.ops import eltwise as E

RELU = (C.PRO_RELU, None, None)


def _bn_pro(bn):
    return (C.PRO_AFFINE_RELU, bn.scale, bn.shift, bn.group_imgs)


class GBlock(nn.Module):
    def __init__(self, in_channels, out_channels, hidden_channels=None, upsample=False):
        super().__init__()
        hidden_channels = hidden_channels if hidden_channels is not None else out_channels
        self.learnable_sc = in_channels != out_channels or upsample
        self.upsample = upsample
        self.c1 = ConvLayer('conv', in_channels, hidden_channels, 3, 1, 1)
        self.c2 = ConvLayer('conv', hidden_channels, out_channels, 3, 1, 1)
        self.b1 = BatchNorm(in_channels)
        self.b2 = BatchNorm(hidden_channels)
        self.c1.xavier_(math.sqrt(2.0))
        self.c2.xavier_(math.sqrt(2.0))
        if self.learnable_sc:
            self.c_sc = ConvLayer('conv', in_channels, out_channels, 1, 1, 0)
            self.c_sc.xavier_(1.0)

    def forward(self, x, training, save=True, need_dgrad=True, bn1=None, next_bn=None, groups=1, save_group=None):
        """bn1: statistics of x if the producer already reduced them; next_bn: the BatchNorm module that will
        consume this block's output (its statistics are then taken from the last conv's epilogue).
        groups > 1: x stacks `groups` batches that are batch-normalised independently; forward only, except that the
        context of ONE of them (save_group) can be kept for a backward pass (views of the stacked tensors)."""
        ctx = {}
        if groups > 1 and save and save_group is None:
            raise RuntimeError("GBlock: a stacked forward keeps the context of one group only (save_group)")
        if bn1 is None:
            bn1 = self.b1.stats(x, training, groups=groups)
        k1 = self.c1.prepare(training, need_dgrad)
        if self.upsample and self.c1.upin_fused(x, _bn_pro(bn1)):
            # BN -> ReLU -> bilinear x2 -> c1 as ONE launch on the low-resolution input: the interpolation is part of the
            # Winograd input transform and the up-sampled tensor is never written (the backward pass of the ONE batch that has
            # one makes its own copy for c1's weight gradient)
            c1_in, c1_pro = None, None
            h1, bn2 = self.c1.fwd_bn(k1, x, self.b2, training, pro=_bn_pro(bn1), groups=groups, up_in=True)
        else:
            if self.upsample:
                c1_in, c1_pro = E.upsample2x(x, pro=_bn_pro(bn1)), None
            else:
                c1_in, c1_pro = x, _bn_pro(bn1)
            h1, bn2 = self.c1.fwd_bn(k1, c1_in, self.b2, training, pro=c1_pro, groups=groups)
        # shortcut: a 1x1 conv commutes with the (linear) bilinear upsampling, so c_sc runs on the LOW
        # resolution input (4x fewer FLOP): c_sc(up(x)) == up(c_sc(x)); the up-sampling itself is blended into c2's
        # epilogue from the low-resolution tensor (res_up), which is never written out at full resolution
        if self.learnable_sc:
            ksc = self.c_sc.prepare(training, need_dgrad)
            sc = self.c_sc.fwd(ksc, x)
        else:
            ksc, sc = None, x
        sc_up = self.learnable_sc and self.upsample
        k2 = self.c2.prepare(training, need_dgrad)
        bn_out = None
        if next_bn is not None:
            out, bn_out = self.c2.fwd_bn(k2, h1, next_bn, training, pro=_bn_pro(bn2), residual=sc, groups=groups,
                                         res_up=sc_up)
        else:
            out = self.c2.fwd(k2, h1, pro=_bn_pro(bn2), residual=sc, res_up=sc_up)
        if save and groups > 1:
            b = x.shape[0] // groups
            sl = slice(save_group * b, (save_group + 1) * b)
            g1 = E.bn_ctx_group(bn1, save_group)
            ctx = dict(x=x[sl], bn1=g1, c1_in=None if c1_in is None else c1_in[sl],
                       c1_pro=None if c1_pro is None else _bn_pro(g1), k1=k1, h1=h1[sl],
                       bn2=E.bn_ctx_group(bn2, save_group), ksc=ksc, k2=k2)
        elif save:
            ctx = dict(x=x, bn1=bn1, c1_in=c1_in, c1_pro=c1_pro, k1=k1, h1=h1, bn2=bn2, ksc=ksc, k2=k2)
        return out, ctx, bn_out

    def backward(self, ctx, gout):
        x, h1 = ctx['x'], ctx['h1']
        hw1 = h1.shape[1:3]
        self.c2.wgrad(ctx['k2'], gout, h1, pro=_bn_pro(ctx['bn2']))
        g_a2 = self.c2.dgrad(ctx['k2'], gout, hw1)
        g_h1 = self.b2.bwd(g_a2, h1, ctx['bn2'], relu=True)
        c1_in = ctx['c1_in']
        if c1_in is None:                    # the forward ran on the low-resolution input (up_in): c1's weight gradient wants
            c1_in = E.upsample2x(x, pro=_bn_pro(ctx['bn1']))       # the up-sampled activation of THIS batch
        self.c1.wgrad(ctx['k1'], g_h1, c1_in, pro=ctx['c1_pro'])
        g_c1in = self.c1.dgrad(ctx['k1'], g_h1, c1_in.shape[1:3])
        if self.learnable_sc:
            # gradient of the low-resolution shortcut: adjoint of the upsampling applied to gout
            g_sc = E.upsample2x_bwd(gout) if self.upsample else gout
            self.c_sc.wgrad(ctx['ksc'], g_sc, x)
            g_x_sc = self.c_sc.dgrad(ctx['ksc'], g_sc, x.shape[1:3])
        else:
            g_x_sc = gout
        g_a1 = E.upsample2x_bwd(g_c1in) if self.upsample else g_c1in
        return self.b1.bwd(g_a1, x, ctx['bn1'], relu=True, residual=g_x_sc)


class DBlock(nn.Module):
    """mimicry's DBlock applies nn.ReLU(True) to `h = x`, which mutates x in place, so the shortcut
    branch consumes relu(x) as well (SURVEY §7 'In-place ReLU aliasing')."""

    def __init__(self, in_channels, out_channels, hidden_channels=None, downsample=False):
        super().__init__()
        hidden_channels = hidden_channels if hidden_channels is not None else in_channels
        self.downsample = downsample
        self.learnable_sc = (in_channels != out_channels) or downsample
        self.c1 = ConvLayer('conv', in_channels, hidden_channels, 3, 1, 1, sn=True)
        self.c2 = ConvLayer('conv', hidden_channels, out_channels, 3, 1, 1, sn=True)
        self.c1.xavier_(math.sqrt(2.0))
        self.c2.xavier_(math.sqrt(2.0))
        if self.learnable_sc:
            self.c_sc = ConvLayer('conv', in_channels, out_channels, 1, 1, 0, sn=True)
            self.c_sc.xavier_(1.0)

    def forward(self, x, training, save=True, need_dgrad=True, slot=None):
        k1 = self.c1.prepare(training, need_dgrad, slot)
        k2 = self.c2.prepare(training, need_dgrad, slot)
        h1 = self.c1.fwd(k1, x, pro=RELU)
        xp = None
        # (the batched D(real)+D(fake) weight gradient cuts the pixel range in two segments of whole 32-pixel
        #  K-steps: keep the full-resolution shortcut when the pooled tensor is too small for that)
        lo_ok = (x.shape[0] * (x.shape[1] // 2) * (x.shape[2] // 2)) % 64 == 0
        if self.learnable_sc and self.downsample and lo_ok:
            # avg_pool2d(c_sc(relu x)) == c_sc(avg_pool2d(relu x)) for the 1x1 shortcut conv: pool first (4x fewer
            # FLOP and bytes in its forward / dgrad / wgrad), add the low-resolution shortcut in the pooling of c2
            ksc = self.c_sc.prepare(training, need_dgrad, slot)
            xp = E.avgpool2(x, relu_in=True)
            sc = self.c_sc.fwd(ksc, xp)
            out = self.c2.fwd_pool(k2, h1, pro=RELU, residual=sc)
        elif self.learnable_sc:
            ksc = self.c_sc.prepare(training, need_dgrad, slot)
            sc = self.c_sc.fwd(ksc, x, pro=RELU)
            h2 = self.c2.fwd(k2, h1, pro=RELU, residual=sc)
            out = E.avgpool2(h2) if self.downsample else h2
        else:
            ksc = None
            h2 = self.c2.fwd(k2, h1, pro=RELU, residual=x, res_relu=True)
            out = E.avgpool2(h2) if self.downsample else h2
        ctx = dict(x=x, xp=xp, h1=h1, k1=k1, k2=k2, ksc=ksc, slot=0 if slot is None else slot) if save else {}
        return out, ctx

    def backward(self, ctx, gout, need_wgrad=True, need_gx=True):
        x, xp, h1, slot = ctx['x'], ctx['xp'], ctx['h1'], ctx['slot']
        hw = x.shape[1:3]
        # c2's data-gradient through the pooling comes straight from the pooled gradient where the layer qualifies (nine
        # Winograd products instead of sixteen); the up-sampled gradient is then only needed by the weight gradients
        # ... and c2's weight gradient from the pooled gradient and 2x2 box sums of relu(h1) (a quarter of the products)
        unpool = self.downsample and self.c2.dgrad_unpool_fused(gout.shape[0], hw)
        pooled_w = self.downsample and need_wgrad and self.c2.wgrad_pooled_ok(gout, h1)
        lo_path = xp is not None
        need_full = self.downsample and ((need_wgrad and not pooled_w) or not unpool or not lo_path)
        g_full = E.avgpool2_bwd(gout) if need_full else gout
        if pooled_w:
            self.c2.wgrad_pooled(ctx['k2'], gout, h1, relu_in=True, slot=slot)
        elif need_wgrad:
            self.c2.wgrad(ctx['k2'], g_full, h1, pro=RELU, slot=slot)
        if unpool:
            g_h1 = self.c2.dgrad_unpool(ctx['k2'], gout, hw, mask_src=h1)
        else:
            g_h1 = self.c2.dgrad(ctx['k2'], g_full, hw, mask_src=h1)
        if need_wgrad:
            self.c1.wgrad(ctx['k1'], g_h1, x, pro=RELU, slot=slot)
            if xp is not None:
                self.c_sc.wgrad(ctx['ksc'], gout, xp, slot=slot)
            elif self.learnable_sc:
                self.c_sc.wgrad(ctx['ksc'], g_full, x, pro=RELU, slot=slot)
        if not need_gx:
            return None
        if xp is not None:
            g_xp = self.c_sc.dgrad(ctx['ksc'], gout, xp.shape[1:3])
            if self.c1.dgrad_res_unpool_fused(x.shape[0], hw):      # the pooled shortcut's gradient un-pooled by c1's epilogue
                return self.c1.dgrad(ctx['k1'], g_h1, hw, residual=g_xp, mask_src=x, res_unpool=True)
            return self.c1.dgrad(ctx['k1'], g_h1, hw, residual=E.avgpool2_bwd(g_xp), mask_src=x)
        if self.learnable_sc:
            tmp = self.c_sc.dgrad(ctx['ksc'], g_full, hw)
            return self.c1.dgrad(ctx['k1'], g_h1, hw, residual=tmp, mask_src=x)
        return self.c1.dgrad(ctx['k1'], g_h1, hw, residual=g_full, mask_src=x)


class DBlockOptimized(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.c1 = ConvLayer('conv', in_channels, out_channels, 3, 1, 1, sn=True)
        self.c2 = ConvLayer('conv', out_channels, out_channels, 3, 1, 1, sn=True)
        self.c_sc = ConvLayer('conv', in_channels, out_channels, 1, 1, 0, sn=True)
        self.c1.xavier_(math.sqrt(2.0))
        self.c2.xavier_(math.sqrt(2.0))
        self.c_sc.xavier_(1.0)

    def forward(self, x, training, save=True, need_dgrad=True, need_in_dgrad=False, slot=None):
        k1 = self.c1.prepare(training, need_dgrad and need_in_dgrad, slot)
        k2 = self.c2.prepare(training, need_dgrad, slot)
        ksc = self.c_sc.prepare(training, need_dgrad and need_in_dgrad, slot)
        h1 = self.c1.fwd(k1, x)
        xp = E.avgpool2(x)
        sc = self.c_sc.fwd(ksc, xp)
        out = self.c2.fwd_pool(k2, h1, pro=RELU, residual=sc)
        ctx = dict(x=x, xp=xp, h1=h1, k1=k1, k2=k2, ksc=ksc, slot=0 if slot is None else slot) if save else {}
        return out, ctx

    def backward(self, ctx, gout, need_wgrad=True, need_gx=True):
        x, xp, h1, slot = ctx['x'], ctx['xp'], ctx['h1'], ctx['slot']
        hw = x.shape[1:3]
        unpool = self.c2.dgrad_unpool_fused(gout.shape[0], hw)
        pooled_w = need_wgrad and self.c2.wgrad_pooled_ok(gout, h1)
        g_full = E.avgpool2_bwd(gout) if ((need_wgrad and not pooled_w) or not unpool) else None
        if need_wgrad:
            self.c_sc.wgrad(ctx['ksc'], gout, xp, slot=slot)
            if pooled_w:
                self.c2.wgrad_pooled(ctx['k2'], gout, h1, relu_in=True, slot=slot)
            else:
                self.c2.wgrad(ctx['k2'], g_full, h1, pro=RELU, slot=slot)
        if unpool:
            g_h1 = self.c2.dgrad_unpool(ctx['k2'], gout, hw, mask_src=h1)
        else:
            g_h1 = self.c2.dgrad(ctx['k2'], g_full, hw, mask_src=h1)
        if need_wgrad:
            self.c1.wgrad(ctx['k1'], g_h1, x, slot=slot)
        if not need_gx:
            return None
        g_xp = self.c_sc.dgrad(ctx['ksc'], gout, xp.shape[1:3])
        g_sc_full = E.avgpool2_bwd(g_xp)
        return self.c1.dgrad(ctx['k1'], g_h1, hw, residual=g_sc_full)


class SNGANBaseGenerator(BaseGenerator):
    out_channels = 3
    supports_stacked_forward = True

    def _blocks(self):
        raise NotImplementedError

    def forward_nhwc(self, z, training, save=True, out=None, groups=1, save_group=None):
        """groups > 1: z stacks `groups` noise batches; the result equals `groups` successive forwards (BatchNorm
        statistics and running-statistics updates per batch, in order) at the GEMM efficiency of the large batch.
        Forward only (save=False), or with the backward context of ONE batch (save=True, save_group)."""
        z = z.to(dtype=self.l1.weight.dtype)
        self.wino_batch(('f', None)).prepare(self.param_version)       # Winograd weights of all blocks in one launch
        x0, h = self.l1.fwd(z)
        bctx = []
        blocks = self._blocks()
        bn = None
        for i, blk in enumerate(blocks):
            nxt = blocks[i + 1].b1 if i + 1 < len(blocks) else self._last_bn
            h, c, bn = blk.forward(h, training, save=save, need_dgrad=save, bn1=bn, next_bn=nxt, groups=groups,
                                   save_group=save_group)
            bctx.append(c)
        k = self._last_conv.prepare(training, need_dgrad=save)
        y_pre = self._last_conv.fwd(k, h, pro=_bn_pro(bn))
        y = E.tanh_fwd(y_pre, out=out)
        if save and groups > 1:
            b = z.shape[0] // groups
            sl = slice(save_group * b, (save_group + 1) * b)
            ctx = dict(x0=x0[sl], bctx=bctx, h=h[sl], bn=E.bn_ctx_group(bn, save_group), k=k, y=y[sl])
        else:
            ctx = dict(x0=x0, bctx=bctx, h=h, bn=bn, k=k, y=y) if save else None
        return y, ctx

    def backward_nhwc(self, ctx, g_img):
        g_pre = E.tanh_bwd(ctx['y'], g_img)
        self.wino_batch(('d', None)).prepare(self.param_version)
        h = ctx['h']
        self._last_conv.wgrad(ctx['k'], g_pre, h, pro=_bn_pro(ctx['bn']))
        g_a = self._last_conv.dgrad(ctx['k'], g_pre, h.shape[1:3])
        g = self._last_bn.bwd(g_a, h, ctx['bn'], relu=True)
        for blk, c in zip(reversed(self._blocks()), reversed(ctx['bctx'])):
            g = blk.backward(c, g)
        self.l1.wgrad(ctx['x0'], g)
        self.wgrad_batch.finish(0)


class SNGANGenerator32(SNGANBaseGenerator):
    max_stacked_images = 1536     # largest activation [N,32,32,256] fp32

    def __init__(self, nz=128, ngf=256, bottom_width=4, loss_type='hinge', **kwargs):
        super().__init__(nz=nz, ngf=ngf, bottom_width=bottom_width, loss_type=loss_type, **kwargs)
        self.l1 = LatentLinear(nz, ngf, bottom_width)
        self.block2 = GBlock(ngf, ngf, upsample=True)
        self.block3 = GBlock(ngf, ngf, upsample=True)
        self.block4 = GBlock(ngf, ngf, upsample=True)
        self.b5 = BatchNorm(ngf)
        self.c5 = ConvLayer('conv', ngf, 3, 3, 1, 1)
        self.l1.xavier_(1.0)
        self.c5.xavier_(1.0)
        self._link_layers()

    def _blocks(self):
        return [self.block2, self.block3, self.block4]

    @property
    def _last_bn(self):
        return self.b5

    @property
    def _last_conv(self):
        return self.c5


class SNGANGenerator64(SNGANBaseGenerator):
    max_stacked_images = 768      # largest activation [N,64,64,128] fp32 (upsampled input of block5)

    def __init__(self, nz=128, ngf=1024, bottom_width=4, loss_type='hinge', **kwargs):
        super().__init__(nz=nz, ngf=ngf, bottom_width=bottom_width, loss_type=loss_type, **kwargs)
        self.l1 = LatentLinear(nz, ngf, bottom_width)
        self.block2 = GBlock(ngf, ngf >> 1, upsample=True)
        self.block3 = GBlock(ngf >> 1, ngf >> 2, upsample=True)
        self.block4 = GBlock(ngf >> 2, ngf >> 3, upsample=True)
        self.block5 = GBlock(ngf >> 3, ngf >> 4, upsample=True)
        self.b6 = BatchNorm(ngf >> 4)
        self.c6 = ConvLayer('conv', ngf >> 4, 3, 3, 1, 1)
        self.l1.xavier_(1.0)
        self.c6.xavier_(1.0)
        self._link_layers()

    def _blocks(self):
        return [self.block2, self.block3, self.block4, self.block5]

    @property
    def _last_bn(self):
        return self.b6

    @property
    def _last_conv(self):
        return self.c6


class SNGANBaseDiscriminator(BaseDiscriminator):
    in_channels_padded = 4
    pair_forward = True     # no BatchNorm in D: D(real) and D(fake) can share one batched pass

    def _blocks(self):
        raise NotImplementedError

    def _sn_prepare(self, slot, training, need_dgrad):
        """Spectral norm of every layer of the network for this forward: 4 launches in total."""
        sb = getattr(self, '_sn_batch', None)
        if sb is None or sb.stale():
            layers = [m for m in self.modules() if isinstance(m, ConvLayer) and m.sn] + [self._head]
            sb = SNBatch(self, layers)
            object.__setattr__(self, '_sn_batch', sb)
        if slot == 'pair':
            sb.run_pair(training, need_dgrad)
        else:
            sb.run(slot, training, 1 if need_dgrad else 0)

    def forward_nhwc(self, x, training, save=True, need_dgrad=True, need_in_dgrad=True, slot=0):
        self._sn_prepare(slot, training, need_dgrad)
        self.wino_batch(('f', slot)).prepare(self._wino_version(slot))
        blocks = self._blocks()
        h, c0 = blocks[0].forward(x, training, save=save, need_dgrad=need_dgrad, need_in_dgrad=need_in_dgrad,
                                  slot=slot)
        bctx = [c0]
        for blk in blocks[1:]:
            h, c = blk.forward(h, training, save=save, need_dgrad=need_dgrad, slot=slot)
            bctx.append(c)
        hctx, logit = self._head.fwd(h, training, slot=slot)
        return logit, (dict(bctx=bctx, hctx=hctx, slot=slot) if save else None)

    def _wino_version(self, slot):
        sb = self._sn_batch
        return ('p', self.param_version) if slot == 'pair' else sb.slot_ver.get(slot)

    def backward_nhwc(self, ctx, dlogit, need_wgrad=True, need_gx=False):
        self.wino_batch(('d', ctx['slot'])).prepare(self._wino_version(ctx['slot']))
        g = self._head.bwd(ctx['hctx'], dlogit, need_wgrad=need_wgrad)
        blocks = self._blocks()
        for i in range(len(blocks) - 1, -1, -1):
            g = blocks[i].backward(ctx['bctx'][i], g, need_wgrad=need_wgrad, need_gx=(need_gx or i > 0))
        if need_wgrad:
            self.wgrad_batch.finish(ctx['slot'])
        return g


class SNGANDiscriminator32(SNGANBaseDiscriminator):
    def __init__(self, ndf=128, loss_type='hinge', **kwargs):
        super().__init__(ndf=ndf, loss_type=loss_type, **kwargs)
        self.block1 = DBlockOptimized(3, ndf)
        self.block2 = DBlock(ndf, ndf, downsample=True)
        self.block3 = DBlock(ndf, ndf, downsample=False)
        self.block4 = DBlock(ndf, ndf, downsample=False)
        self.l5 = HeadLinear(ndf, sn=True)
        self.l5.xavier_(1.0)
        self._link_layers()

    def _blocks(self):
        return [self.block1, self.block2, self.block3, self.block4]

    @property
    def _head(self):
        return self.l5


class SNGANDiscriminator64(SNGANBaseDiscriminator):
    def __init__(self, ndf=1024, loss_type='hinge', **kwargs):
        super().__init__(ndf=ndf, loss_type=loss_type, **kwargs)
        self.block1 = DBlockOptimized(3, ndf >> 4)
        self.block2 = DBlock(ndf >> 4, ndf >> 3, downsample=True)
        self.block3 = DBlock(ndf >> 3, ndf >> 2, downsample=True)
        self.block4 = DBlock(ndf >> 2, ndf >> 1, downsample=True)
        self.block5 = DBlock(ndf >> 1, ndf, downsample=True)
        self.l6 = HeadLinear(ndf, sn=True)
        self.l6.xavier_(1.0)
        self._link_layers()

    def _blocks(self):
        return [self.block1, self.block2, self.block3, self.block4, self.block5]

    @property
    def _head(self):
        return self.l6
