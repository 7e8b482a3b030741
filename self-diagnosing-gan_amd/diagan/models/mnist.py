"""MNIST_DCGAN generator / discriminator on the HIP engine (BASELINE configs[0] plumbing model).

Reference: diagan-pkg/diagan/models/mnist.py:47-80 (generator: fc + 4 ConvTranspose2d with BN/ReLU,
tanh) and :155-223 (discriminator: PacGAN packing, 6 x [Conv3x3 (strides 2,1,2,1,2,1) (+BN) +
LeakyReLU(0.2) + Dropout(0.5)], Linear(8192, 1), get_feature).  Module names follow the reference's
nn.Sequential indices so state_dict keys line up (tconv.0.weight, conv.4.running_mean, out_d.weight).
`weights_init_3channel(self)` is a no-op in the reference (called on the root module, :33-39,74), so
the default PyTorch initialisation is kept.  `use_sn=True` (torch.nn.utils.spectral_norm) is not part
of the accelerated path.

Transposed convolutions run on the same implicit-GEMM kernel as every data-gradient (gather formula
(1,-1,+p,s)); BatchNorm+ReLU of the generator is fused into the next layer's loader; the
discriminator's BN -> LeakyReLU -> Dropout is one elementwise kernel (act_fwd) and its backward is
folded into the BatchNorm backward reduction.
"""
import torch
import torch.nn as nn

from diagan.models.base import BaseDiscriminator, BaseGenerator
from diagan.models.layers import BatchNorm, ConvLayer, LatentLinear, _r4
from diagan.models.topk_models import TopKGenerator
from diagan.ops import conv as C
from diagan.ops import eltwise as E


def _bn_pro(bn):
    return (C.PRO_AFFINE_RELU, bn.scale, bn.shift)


class MNIST_DCGAN_Generator(BaseGenerator, TopKGenerator):
    launch_bound = True      # ~1000 launches of ~10 us per global step: LogTrainer replays the step as one hipGraph

    def __init__(self, nz=100, nc=3, loss_type='hinge', topk=1, **kwargs):
        BaseGenerator.__init__(self, nz=100, ngf=128, bottom_width=4, loss_type=loss_type)
        TopKGenerator.__init__(self, use_topk=topk, decay_steps=10)
        print(f"Load MNIST_DCGAN_Generator reweight model loss_type: {loss_type} topk: {topk}")
        self.nz, self.out_channels = nz, nc
        self.fc = LatentLinear(nz, 384, 1)
        self.tconv = nn.ModuleDict({
            '0': ConvLayer('convT', 384, 192, 4, 1, 0, bias=False), '1': BatchNorm(192),
            '3': ConvLayer('convT', 192, 96, 4, 2, 1, bias=False), '4': BatchNorm(96),
            '6': ConvLayer('convT', 96, 48, 4, 2, 1, bias=False), '7': BatchNorm(48),
            '9': ConvLayer('convT', 48, nc, 4, 2, 1, bias=False),
        })
        self._link_layers()

    def forward_nhwc(self, z, training, save=True, out=None):
        t = self.tconv
        z = z.to(dtype=torch.float32)
        x0, h = self.fc.fwd(z)                                   # [B,1,1,384]
        k1 = t['0'].prepare(training, save)
        # BatchNorm statistics come out of the producing GEMM's epilogue (no second pass over y)
        y1, bn1 = t['0'].fwd_bn(k1, h, t['1'], training)         # [B,4,4,192]
        k2 = t['3'].prepare(training, save)
        y2, bn2 = t['3'].fwd_bn(k2, y1, t['4'], training, pro=_bn_pro(bn1))      # [B,8,8,96]
        k3 = t['6'].prepare(training, save)
        y3, bn3 = t['6'].fwd_bn(k3, y2, t['7'], training, pro=_bn_pro(bn2))      # [B,16,16,48]
        k4 = t['9'].prepare(training, save)
        y4 = t['9'].fwd(k4, y3, pro=_bn_pro(bn3))                # [B,32,32,4]
        img = E.tanh_fwd(y4, out=out)
        ctx = dict(x0=x0, h=h, y=(y1, y2, y3), bn=(bn1, bn2, bn3), k=(k1, k2, k3, k4), img=img) if save else None
        return img, ctx

    def backward_nhwc(self, ctx, g_img):
        t = self.tconv
        (y1, y2, y3), (bn1, bn2, bn3), (k1, k2, k3, k4) = ctx['y'], ctx['bn'], ctx['k']
        g4 = E.tanh_bwd(ctx['img'], g_img)
        t['9'].wgrad(k4, g4, y3, pro=_bn_pro(bn3))
        g = t['7'].bwd(t['9'].dgrad(k4, g4, y3.shape[1:3]), y3, bn3, relu=True)
        t['6'].wgrad(k3, g, y2, pro=_bn_pro(bn2))
        g = t['4'].bwd(t['6'].dgrad(k3, g, y2.shape[1:3]), y2, bn2, relu=True)
        t['3'].wgrad(k2, g, y1, pro=_bn_pro(bn1))
        g = t['1'].bwd(t['3'].dgrad(k2, g, y1.shape[1:3]), y1, bn1, relu=True)
        t['0'].wgrad(k1, g, ctx['h'])
        g = t['0'].dgrad(k1, g, (1, 1))
        self.fc.wgrad(ctx['x0'], g)
        self.wgrad_batch.finish(0)


class OutLinear(nn.Module):
    """nn.Linear(4*4*512, 1) on the NCHW-flattened feature (mnist.py:191,219); the weight is kept in
    NHWC-flatten order so it dots directly with the [B,4,4,512] activation."""

    def __init__(self, ch, hw):
        super().__init__()
        self.ch, self.hw = ch, hw
        ref = nn.Linear(ch * hw, 1)
        self.weight = nn.Parameter(self._perm(ref.weight.data))
        self.bias = nn.Parameter(torch.cat([ref.bias.data, torch.zeros(3)]))
        self._register_state_dict_hook(self._sd_hook)
        self._register_load_state_dict_pre_hook(self._load_hook)

    def _perm(self, w):        # (c, hw) -> (hw, c)
        return w.view(1, self.ch, self.hw).permute(0, 2, 1).reshape(1, -1).contiguous()

    def _unperm(self, w):
        return w.view(1, self.hw, self.ch).permute(0, 2, 1).reshape(1, -1).contiguous()

    @staticmethod
    def _sd_hook(module, sd, prefix, local_metadata):
        sd[prefix + 'weight'] = module._unperm(sd[prefix + 'weight'])
        sd[prefix + 'bias'] = sd[prefix + 'bias'][:1].clone()
        return sd

    def _load_hook(self, sd, prefix, *args):
        if prefix + 'weight' in sd:
            sd[prefix + 'weight'] = self._perm(sd[prefix + 'weight'].to(torch.float32))
        k = prefix + 'bias'
        if k in sd and sd[k].numel() == 1:
            sd[k] = torch.cat([sd[k].to(torch.float32), torch.zeros(3, device=sd[k].device)])


class MNIST_DCGAN_Discriminator(BaseDiscriminator):
    launch_bound = True      # ~1000 launches of ~10 us per global step: LogTrainer replays the step as one hipGraph

    CFG = ((None, 16, 2, False), (16, 32, 1, True), (32, 64, 2, True), (64, 128, 1, True), (128, 256, 2, True),
           (256, 512, 1, True))      # (in, out, stride, batch-norm)

    def __init__(self, nc=3, num_pack=1, use_sn=False, loss_type='hinge', use_gold=False, **kwargs):
        print(f"Load MNIST_DCGAN_Discriminator reweight model loss_type {loss_type}, num_pack: {num_pack}, "
              f"use_gold: {use_gold}")
        BaseDiscriminator.__init__(self, ndf=128, loss_type=loss_type)
        if use_sn:
            raise NotImplementedError("use_sn (torch.nn.utils.spectral_norm) is outside the accelerated path")
        self.nc, self.num_pack = nc, num_pack
        self.in_channels_padded = _r4(nc)            # NHWC images entering forward_nhwc (before packing)
        mods = {}
        idx = 0
        self._idx = []
        for cin, cout, stride, bn in self.CFG:
            cin = nc * num_pack if cin is None else cin
            mods[str(idx)] = ConvLayer('conv', cin, cout, 3, stride, 1, bias=False)
            self._idx.append((str(idx), str(idx + 1) if bn else None))
            if bn:
                mods[str(idx + 1)] = BatchNorm(cout)
            idx += 4 if bn else 3          # conv, [bn], lrelu, dropout
        self.conv = nn.ModuleDict(mods)
        self.out_d = OutLinear(512, 16)
        self.use_gold = use_gold
        self._link_layers()

    # ---- PacGAN packing (mnist.py:213-216): chunks of the batch concatenated on the channel axis
    def _pack_nhwc(self, x):
        if self.num_pack == 1:
            return x
        parts = torch.split(x[..., : self.nc], int(x.size(0) / self.num_pack))
        x = torch.cat(parts, dim=-1)
        return torch.nn.functional.pad(x, (0, _r4(self.nc * self.num_pack) - x.shape[-1])).contiguous()

    def forward_nhwc(self, x, training, save=True, need_dgrad=True, need_in_dgrad=True, slot=0, drop_masks=None):
        x = self._pack_nhwc(x)
        h, saved = x, []
        # the keep-masks of ALL layers of this forward from ONE bernoulli_ launch over a flat buffer (round 5: eight launches
        # before), cut in the layers' activation sizes
        flat_mask, moff = None, 0
        if training and drop_masks is None:
            total, hw = 0, tuple(x.shape[1:3])
            for ci, _ in self._idx:
                hw = self.conv[ci].geom.out_hw(*hw)
                total += x.shape[0] * hw[0] * hw[1] * self.conv[ci].geom.Co
            flat_mask = torch.empty(total, dtype=torch.float32, device=x.device).bernoulli_(0.5)
        for i, (ci, bi) in enumerate(self._idx):
            conv = self.conv[ci]
            k = conv.prepare(training, need_dgrad and (i > 0 or need_in_dgrad))
            if bi is not None:
                y, bn = conv.fwd_bn(k, h, self.conv[bi], training)      # statistics from the GEMM epilogue
            else:
                y, bn = conv.fwd(k, h), None
            drop, dsc = None, 1.0
            if training:      # Dropout(0.5): RNG is torch's (device generator).  An injected mask carries its 1 / (1 - p); the
                if drop_masks is not None:      # own one is the 0 / 1 keep-mask as bernoulli_ writes it, scaled inside the
                    drop = drop_masks[i]        # kernels that multiply by it (one launch per layer and pass less: round 5)
                else:
                    drop, dsc = flat_mask[moff: moff + y.numel()].view(y.shape), 2.0
                    moff += y.numel()
            a = E.act_fwd(y, 0.2, bn.scale if bn else None, bn.shift if bn else None, drop, dsc)
            saved.append((h, k, y, bn, (drop, dsc)))
            h = a
        B = h.shape[0]
        flat = h.view(B, -1)
        logit = E.linear1_fwd(flat, self.out_d.weight.data, self.out_d.bias.data)
        ctx = dict(saved=saved, flat=flat) if save else None
        return logit, ctx

    def state_dict_grad_out_d(self):
        """out_d.weight gradient in the reference's [1, 8192] (c, h, w) order."""
        return self.out_d._unperm(self.out_d.weight.grad)

    def forward(self, x, get_feature=False):
        xn = self.to_nhwc(x)
        if get_feature:       # NCHW-flattened 8192-vector of the last activation (mnist.py:218-221)
            _, ctx = self.forward_nhwc(xn, self.training, save=True, need_dgrad=False)
            B = xn.shape[0]
            return ctx['flat'].view(B, 16, 512).permute(0, 2, 1).reshape(B, -1).contiguous()
        logit, _ = self.forward_nhwc(xn, self.training, save=False, need_dgrad=False)
        return logit

    def backward_nhwc(self, ctx, dlogit, need_wgrad=True, need_gx=False):
        saved, flat = ctx['saved'], ctx['flat']
        B = flat.shape[0]
        w = self.out_d.weight.data
        if need_wgrad:
            E.linear1_wgrad(dlogit, flat, self.out_d.weight.grad, self.out_d.bias.grad)
        g = E.linear1_bwd_input(dlogit, w, B, flat.shape[1]).view(saved[-1][2].shape)
        for i in range(len(saved) - 1, -1, -1):
            h_in, k, y, bn, (drop, dsc) = saved[i]
            ci, bi = self._idx[i]
            if bn is not None:
                g_y = self.conv[bi].bwd(g, y, bn, relu=True, slope=0.2, drop=drop, drop_scale=dsc)
            else:
                g_y = E.act_bwd(g, y, 0.2, drop, dsc)
            if need_wgrad:
                self.conv[ci].wgrad(k, g_y, h_in)
            if i > 0 or need_gx:
                g = self.conv[ci].dgrad(k, g_y, h_in.shape[1:3])
            else:
                g = None
        if need_wgrad:
            self.wgrad_batch.finish(0)
        if g is not None and self.num_pack > 1:
            raise NotImplementedError("image gradient through PacGAN packing (num_pack > 1) is not implemented")
        return g
