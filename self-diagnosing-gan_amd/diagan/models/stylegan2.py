"""StyleGAN2 generator / discriminator (API of diagan-pkg/diagan/models/stylegan2.py = stylegan2/model.py:14-678),
SURVEY §8(f) rank 1.

Same module tree, parameter / buffer names and shapes as the reference (a rosinality checkpoint's `g`, `d`, `g_ema`
load with `load_state_dict`), same call signatures; images enter and leave as NCHW.  INSIDE, activations are NHWC fp32
-- the layout of this engine's convolution kernels and of the native `upfirdn2d` op ([major, H, W, minor]) -- and
every FLOP-carrying op is a HIP kernel behind the C ABI:

    conv2d / conv_transpose2d / linear   diagan.ops.diffconv  (implicit-GEMM MFMA kernels, any-order autograd)
    blur / up / down FIR                 diagan.models.op.upfirdn2d_nhwc
    bias + leaky ReLU * sqrt(2)          diagan.models.op.fused_leaky_relu

The modulated convolution (reference :224-265) is evaluated in its activation-side form: the reference folds the
style s[b,ci] and the demodulation d[b,co] into a per-sample weight and runs a grouped convolution with `batch`
groups; because both factors are constant over space, conv(x, W*s*d) = d * conv(x*s, W), which is ONE dense
convolution for the whole batch (M = B*H*W rows on the MFMA tiles instead of B small GEMMs) and
d[b,co] = rsqrt(sum_ci s[b,ci]^2 * sum_k (scale*W[co,ci,k])^2 + eps) is a [B,Ci]x[Ci,Co] product.  Output channel
counts that are not a multiple of 4 (the 3 RGB planes; 512+1 after the minibatch-stddev plane) are zero-padded
internally and sliced at the NCHW boundary."""
import math
import random

import torch
import torch.nn.functional as F
from torch import nn

from diagan.models.layers import FlatNet
from diagan.models.op.fused_act import FusedLeakyReLU, fused_leaky_relu, scale_rows, styled_bias_act, styled_bias_act_mod
from diagan.models.op import fused_tail as _tails
from diagan.models.op.fused_tail import (bias_act_add, bias_act_blur, blur_styled_act, blur_styled_act_ok, fork_fir, fromrgb, fromrgb_ok,
                                         torgb, torgb_ok)
from diagan.models.op.upfirdn2d import upfirdn2d_nhwc
from diagan.ops import diffconv as dc

SQRT2 = math.sqrt(2.0)
import os as _os
FUSED_SKIP = _os.environ.get("DIAGAN_SG2_FUSED_SKIP", "1") == "1"      # ResBlock skip: blur + sub-sampling in one FIR pass


def channel_table(multiplier):
    """resolution -> feature maps (reference :385-395 / :623-633)"""
    table = {res: 512 for res in (4, 8, 16, 32)}
    table.update({64 << i: (256 >> i) * multiplier for i in range(5)})
    return table


def make_kernel(k):
    """separable taps -> normalised 2-D FIR kernel (reference :22-30)"""
    k = torch.as_tensor(k, dtype=torch.float32)
    if k.dim() == 1:
        k = torch.outer(k, k)
    return k / k.sum()


def _resample_pad(taps, factor, kernel_size, mode):
    """(pad0, pad1) of the blur that accompanies a stride-`factor` convolution (reference :186-201, :565-570)"""
    if mode == 'up':
        p = (taps - factor) - (kernel_size - 1)
        return (p + 1) // 2 + factor - 1, p // 2 + 1
    p = (taps - factor) + (kernel_size - 1)
    return (p + 1) // 2, p // 2


def to_nhwc(images, channels=None):
    """[B,C,H,W] -> contiguous [B,H,W,C'] with C' = `channels` (zero planes appended) or C rounded up to 4"""
    x = images.permute(0, 2, 3, 1)
    c = x.shape[3]
    cp = channels or (c + 3) // 4 * 4
    return F.pad(x, (0, cp - c)) if cp != c else x.contiguous()


def to_nchw(x, channels):
    return x[..., :channels].permute(0, 3, 1, 2).contiguous()


class PixelNorm(nn.Module):
    def forward(self, input):
        return input * torch.rsqrt(input.square().mean(dim=1, keepdim=True) + 1e-8)


class _FIR(nn.Module):
    """upfirdn2d with a registered `kernel` buffer, applied to NHWC activations"""
    up = down = 1

    def __init__(self, kernel, pad):
        super().__init__()
        self.register_buffer("kernel", kernel)
        self.pad = pad

    def forward(self, x):
        return upfirdn2d_nhwc(x, self.kernel, up=self.up, down=self.down, pad=self.pad)


class Upsample(_FIR):
    def __init__(self, kernel, factor=2):
        k = make_kernel(kernel) * factor ** 2
        p = k.shape[0] - factor
        super().__init__(k, ((p + 1) // 2 + factor - 1, p // 2))
        self.factor = self.up = factor


class Downsample(_FIR):
    def __init__(self, kernel, factor=2):
        k = make_kernel(kernel)
        p = k.shape[0] - factor
        super().__init__(k, ((p + 1) // 2, p // 2))
        self.factor = self.down = factor


class Blur(_FIR):
    def __init__(self, kernel, pad, upsample_factor=1):
        k = make_kernel(kernel)
        super().__init__(k * upsample_factor ** 2 if upsample_factor > 1 else k, pad)


class _ChannelsLastLeakyReLU(FusedLeakyReLU):
    """FusedLeakyReLU (same `bias` parameter) for [..., channel] activations"""

    def forward(self, x):
        return fused_leaky_relu(x, self.bias, self.negative_slope, self.scale, bias_dim=-1)


class EqualConv2d(nn.Module):
    """equalised-learning-rate convolution (reference :94-129); NHWC in, NHWC out"""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.stride, self.padding = stride, padding
        self.bias = nn.Parameter(torch.zeros(out_channel)) if bias else None

    def forward(self, x, out_mul=1.0, stride=None):
        """out_mul: an extra factor on the OUTPUT, folded into the weight scale (ResBlock's 1 / sqrt 2 on its skip branch);
        stride: override (ResBlock's skip branch hands over an already sub-sampled input)"""
        y = dc.conv2d(x, self.weight, self.stride if stride is None else stride, self.padding, scale=self.scale * out_mul)
        if self.bias is not None:
            y = y + F.pad(self.bias, (0, y.shape[3] - self.bias.shape[0])) * out_mul
        return y

    def __repr__(self):
        o, i, k, _ = self.weight.shape
        return f"{type(self).__name__}({i}, {o}, {k}, stride={self.stride}, padding={self.padding})"


class EqualLinear(nn.Module):
    """equalised-learning-rate linear layer (reference :132-166) on [B, in_dim]"""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.full((out_dim,), float(bias_init))) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def _finish(self, out):
        bias = (self.bias * self.lr_mul if self.lr_mul != 1 else self.bias) if self.bias is not None else None
        if self.activation:
            return fused_leaky_relu(out, bias, bias_dim=-1)
        return out + bias if bias is not None else out

    def forward(self, input):
        if not self.activation and _tails.mod_linear_ok(input, self.weight):
            # product, equalised-learning-rate scale and bias in one launch (round 6; the styles' modulation layers come here)
            return _tails.mod_linear(input, self.weight, self.bias, self.scale, self.lr_mul)
        return self._finish(dc.linear(input, self.weight, scale=self.scale))

    def forward_spatial(self, x):
        """The same layer applied to the reference's `x_nchw.view(batch, -1)` when x is held as [B,H,W,C]: the
        weight's columns run (c, h, w), i.e. it IS an H x W convolution without padding."""
        b, h, w, c = x.shape
        out = dc.conv2d(x, self.weight.view(-1, c, h, w), scale=self.scale)
        return self._finish(out.view(b, -1)[:, : self.weight.shape[0]])

    def __repr__(self):
        return f"{type(self).__name__}({self.weight.shape[1]}, {self.weight.shape[0]})"


class ModulatedConv2d(nn.Module):
    """weight-(de)modulated convolution (reference :169-265), evaluated activation-side (module docstring)"""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False,
                 downsample=False, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.eps = 1e-8
        self.kernel_size, self.in_channel, self.out_channel = kernel_size, in_channel, out_channel
        self.upsample, self.downsample, self.demodulate = upsample, downsample, demodulate
        if upsample:
            self.blur = Blur(blur_kernel, pad=_resample_pad(len(blur_kernel), 2, kernel_size, 'up'), upsample_factor=2)
        if downsample:
            self.blur = Blur(blur_kernel, pad=_resample_pad(len(blur_kernel), 2, kernel_size, 'down'))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)

    def forward(self, x, style):
        y, d = self.forward_parts(x, style)
        return scale_rows(y, d) if d is not None else y

    def forward_parts(self, x, style, s=None, premodulated=False, blur=True):
        """(convolution of the modulated input, demodulation factors [B, Co] or None): the caller applies d -- StyledConv
        does it inside its fused noise + bias + activation pass.
        s: this layer's style self.modulation(style) when the caller has it already; premodulated: x arrives multiplied by it (the
        producing layer's tail did it on its way out); blur=False: an up-sampling layer returns the transposed convolution's output
        and leaves its Blur to the caller (StyledConv's one-pass blur + tail)"""
        if s is None:
            s = self.modulation(style)                               # [B, Ci]
        # (a VIEW, not self.weight[0]: the select's backward zero-fills a weight-sized tensor and copies into it, every pass; the
        #  equalised-learning-rate scale rides in the packing launch: dc.*(..., scale=))
        w = self.weight.view(self.weight.shape[1:])
        if self.downsample:
            x = self.blur(x)
        if not premodulated:
            x = scale_rows(x, s)
        if self.upsample:
            y = dc.conv_transpose2d(x, w, stride=2, padding=0, scale=self.scale)
            if blur:
                y = self.blur(y)
        elif self.downsample:
            y = dc.conv2d(x, w, stride=2, padding=0, scale=self.scale)
        else:
            y = dc.conv2d(x, w, stride=1, padding=self.padding, scale=self.scale)
        # demodulation 1 / sqrt(sum_ci s^2 sum_taps (scale w)^2 + eps) as a [B,Ci] x [Ci,Co] product
        if not self.demodulate:
            d = None
        elif _tails.demod_ok(s, w):
            d = _tails.demod(s, w, self.scale ** 2, self.eps)                                   # (one launch: models/op/fused_tail.py)
        else:
            d = torch.rsqrt(dc.linear(s.square(), w.square().sum((2, 3)), scale=self.scale ** 2) + self.eps)
        return y, d

    def __repr__(self):
        return (f"{type(self).__name__}({self.in_channel}, {self.out_channel}, {self.kernel_size}, "
                f"upsample={self.upsample}, downsample={self.downsample})")


class NoiseInjection(nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))

    @staticmethod
    def draw(image, noise=None):
        """the noise map as [B or 1, H, W, 1]: fresh N(0,1) (same draw count and order as the reference's
        [B,1,H,W]) or the given [B or 1, 1, H, W] tensor"""
        return NoiseInjection.draw_hw(image, image.shape[0], image.shape[1], image.shape[2], noise)

    @staticmethod
    def draw_hw(like, b, h, w, noise=None):
        if noise is None:
            return like.new_empty(b, h, w, 1).normal_()
        return noise.reshape(noise.shape[0], h, w, 1)

    def forward(self, image, noise=None):
        return image + self.weight * self.draw(image, noise)


class ConstantInput(nn.Module):
    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))

    def forward(self, input):
        return self.input.permute(0, 2, 3, 1).repeat(input.shape[0], 1, 1, 1)


class StyledConv(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, blur_kernel=[1, 3, 3, 1],
                 demodulate=True):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample,
                                    blur_kernel=blur_kernel, demodulate=demodulate)
        self.noise = NoiseInjection()
        self.activate = _ChannelsLastLeakyReLU(out_channel)

    def forward(self, input, style, noise=None, s=None, premodulated=False, post=None):
        """s / premodulated: see ModulatedConv2d.forward_parts; post [B, Co]: the NEXT layer's style -- the result is then the pair
        (y, y * post): that layer's modulated input leaves in this layer's tail pass (y is None where nothing needs it: no graph
        recorded, and the caller of an up-sampling layer only wants the modulated output)"""
        # conv -> * demod -> + strength * noise -> + bias -> leaky ReLU * sqrt(2): the last four in one launch
        act = self.activate
        if self.conv.upsample and blur_styled_act_ok(input, self.conv.blur.kernel):
            # ... and the Blur in front of them as well (round 6)
            y, d = self.conv.forward_parts(input, style, s, premodulated, blur=False)
            blur = self.conv.blur
            oh, ow = (y.shape[1] + blur.pad[0] + blur.pad[1] - blur.kernel.shape[0] + 1,
                      y.shape[2] + blur.pad[0] + blur.pad[1] - blur.kernel.shape[1] + 1)
            nz = self.noise.draw_hw(y, y.shape[0], oh, ow, noise)
            out = blur_styled_act(y, blur.kernel, blur.pad, d, nz, self.noise.weight, act.bias, act.negative_slope, act.scale, post)
            return (None, out) if post is not None else out
        y, d = self.conv.forward_parts(input, style, s, premodulated)
        nz = self.noise.draw(y, noise)
        if post is None:
            return styled_bias_act(y, d, nz, self.noise.weight, act.bias, act.negative_slope, act.scale)
        if y.is_cuda and y.shape[3] % 4 == 0:
            return styled_bias_act_mod(y, d, nz, self.noise.weight, act.bias, post, act.negative_slope, act.scale)
        out = styled_bias_act(y, d, nz, self.noise.weight, act.bias, act.negative_slope, act.scale)
        return out, scale_rows(out, post)


class ToRGB(nn.Module):
    """1x1 modulated convolution to 3 (internally 4) planes + the up-sampled running image (reference :332-351)"""

    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        if upsample:
            self.upsample = Upsample(blur_kernel)
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))

    def forward(self, input, style, skip=None):
        if torgb_ok(input) and self.conv.kernel_size == 1 and not self.conv.demodulate:
            # modulation, 1x1 convolution and bias in ONE read of the layer's input (round 6: models/op/fused_tail.py)
            out = torgb(input, self.conv.modulation(style), self.conv.weight.view(3, -1), self.bias.view(3), self.conv.scale)
        else:
            out = self.conv(input, style)
            out = out + F.pad(self.bias.view(3), (0, out.shape[3] - 3))
        if skip is not None:
            out = out + self.upsample(skip)
        return out


class StyleGANGenerator(FlatNet):
    def __init__(self, size=32, style_dim=512, n_mlp=8, channel_multiplier=2, blur_kernel=[1, 3, 3, 1], lr_mlp=0.01,
                 **kwargs):
        super().__init__()
        self.size, self.style_dim = size, style_dim
        self.style = nn.Sequential(PixelNorm(), *[EqualLinear(style_dim, style_dim, lr_mul=lr_mlp,
                                                              activation="fused_lrelu") for _ in range(n_mlp)])
        self.channels = channel_table(channel_multiplier)
        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1
        self.n_latent = self.log_size * 2 - 2

        self.input = ConstantInput(self.channels[4])
        self.conv1 = StyledConv(self.channels[4], self.channels[4], 3, style_dim, blur_kernel=blur_kernel)
        self.to_rgb1 = ToRGB(self.channels[4], style_dim, upsample=False)
        self.convs, self.upsamples, self.to_rgbs = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.noises = nn.Module()
        for layer in range(self.num_layers):
            res = 2 ** ((layer + 5) // 2)
            self.noises.register_buffer(f"noise_{layer}", torch.randn(1, 1, res, res))
        width = self.channels[4]
        for level in range(3, self.log_size + 1):
            nxt = self.channels[2 ** level]
            self.convs.append(StyledConv(width, nxt, 3, style_dim, upsample=True, blur_kernel=blur_kernel))
            self.convs.append(StyledConv(nxt, nxt, 3, style_dim, blur_kernel=blur_kernel))
            self.to_rgbs.append(ToRGB(nxt, style_dim))
            width = nxt

    # ---- reference helpers (:441-477) ----------------------------------------------------------------------
    def restore_checkpoint(self, ckpt_file, optimizer=None):
        print("load model:", ckpt_file)
        ckpt = torch.load(ckpt_file, map_location="cpu")
        self.load_state_dict(ckpt["g_ema"])

    def generate_images(self, num_images, device):
        images, _ = self.forward([torch.randn(num_images, self.style_dim, device=device)])
        return images

    def make_noise(self):
        device = self.input.input.device
        sizes = [4] + [2 ** level for level in range(3, self.log_size + 1) for _ in range(2)]
        return [torch.randn(1, 1, s, s, device=device) for s in sizes]

    def mean_latent(self, n_latent):
        z = torch.randn(n_latent, self.style_dim, device=self.input.input.device)
        return self.style(z).mean(0, keepdim=True)

    def get_latent(self, input):
        return self.style(input)

    def forward(self, styles, return_latents=False, inject_index=None, truncation=1, truncation_latent=None,
                input_is_latent=False, noise=None, randomize_noise=True):
        if not input_is_latent:
            styles = [self.style(s) for s in styles]
        if noise is None:
            noise = [None] * self.num_layers if randomize_noise else \
                [getattr(self.noises, f"noise_{i}") for i in range(self.num_layers)]
        if truncation < 1:
            styles = [truncation_latent + truncation * (s - truncation_latent) for s in styles]
        if len(styles) < 2:
            inject_index = self.n_latent
            latent = styles[0].unsqueeze(1).repeat(1, inject_index, 1) if styles[0].dim() < 3 else styles[0]
        else:
            if inject_index is None:
                inject_index = random.randint(1, self.n_latent - 1)
            latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                                styles[1].unsqueeze(1).repeat(1, self.n_latent - inject_index, 1)], 1)

        # the per-layer styles latent[:, i] as contiguous views of ONE transposed copy: each strided select was a copy in the forward and,
        # in a backward that reaches the latents (path-length regularisation), a zero-filled [B, n_latent, 512] tensor + an add
        lat = latent.transpose(0, 1).contiguous().unbind(0)
        if not (_tails.FUSED_TAILS and len(self.to_rgbs)):
            out = self.conv1(self.input(latent), lat[0], noise=noise[0])
            skip = self.to_rgb1(out, lat[1])
            for level, to_rgb in enumerate(self.to_rgbs):
                i = 1 + 2 * level
                out = self.convs[2 * level](out, lat[i], noise=noise[i])
                out = self.convs[2 * level + 1](out, lat[i + 1], noise=noise[i + 1])
                skip = to_rgb(out, lat[i + 2], skip)
            return to_nchw(skip, 3), (latent if return_latents else None)
        # Round 6: every styled layer hands its successor's style to its own tail, so the successor's modulated input leaves in the pass
        # that makes the activation (and comes back, in the backward, as ONE pass: models/op/fused_act.py: _StyledActMod); ToRGB reads the
        # un-modulated activation with its own style inside its kernel.  Same values, same order of random draws.
        s_next = self.convs[0].conv.modulation(lat[1])
        y, ym = self.conv1(self.input(latent), lat[0], noise=noise[0], post=s_next)
        skip = self.to_rgb1(y, lat[1])
        last = len(self.to_rgbs) - 1
        for level, to_rgb in enumerate(self.to_rgbs):
            i = 1 + 2 * level
            up, same = self.convs[2 * level], self.convs[2 * level + 1]
            s_up, s_same = s_next, same.conv.modulation(lat[i + 1])
            _, xm = up(ym, lat[i], noise=noise[i], s=s_up, premodulated=True, post=s_same)
            if level == last:
                y = same(xm, lat[i + 1], noise=noise[i + 1], s=s_same, premodulated=True)
            else:
                s_next = self.convs[2 * level + 2].conv.modulation(lat[i + 2])
                y, ym = same(xm, lat[i + 1], noise=noise[i + 1], s=s_same, premodulated=True, post=s_next)
            skip = to_rgb(y, lat[i + 2], skip)
        return to_nchw(skip, 3), (latent if return_latents else None)


class ConvLayer(nn.Sequential):
    """[Blur] + EqualConv2d + [FusedLeakyReLU] (reference :553-595); NHWC"""

    def __init__(self, in_channel, out_channel, kernel_size, downsample=False, blur_kernel=[1, 3, 3, 1], bias=True,
                 activate=True):
        layers = []
        if downsample:
            layers.append(Blur(blur_kernel, pad=_resample_pad(len(blur_kernel), 2, kernel_size, 'down')))
        self.padding = 0 if downsample else kernel_size // 2
        layers.append(EqualConv2d(in_channel, out_channel, kernel_size, padding=self.padding,
                                  stride=2 if downsample else 1, bias=bias and not activate))
        if activate:
            layers.append(_ChannelsLastLeakyReLU(out_channel, bias=bias))
        super().__init__(*layers)


class ResBlock(nn.Module):
    def __init__(self, in_channel, out_channel, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.conv1 = ConvLayer(in_channel, in_channel, 3)
        self.conv2 = ConvLayer(in_channel, out_channel, 3, downsample=True)
        self.skip = ConvLayer(in_channel, out_channel, 1, downsample=True, activate=False, bias=False)

    def forward(self, input):
        """(conv2(conv1(x)) + skip(x)) / sqrt 2 (reference :597-614) without the pass over the sum that the division costs
        (and its mirror image in the backward): the factor goes into what produces the two branches -- conv2's activation
        leaky_relu(.) * sqrt 2 runs with scale 1, the skip convolution with its weight scale divided by sqrt 2."""
        conv1, act1 = self.conv1
        blur2, conv2, act2 = self.conv2
        blur_s, conv_s = self.skip
        fused_skip = FUSED_SKIP and conv_s.stride == 2 and conv_s.padding == 0 and conv_s.weight.shape[2] == 1
        if fused_skip:     # (the input's two consumers as ONE autograd node: its backward adds their gradients in the filter's pass)
            input, down = fork_fir(input, blur_s.kernel, down=2, pad=blur_s.pad)
        # conv1's activation rides in the Blur's pass, conv2's in the pass that adds the skip branch (round 6: models/op/fused_tail.py)
        z = conv2(bias_act_blur(conv1(input), act1.bias, blur2.kernel, blur2.pad, act1.negative_slope, act1.scale))
        # skip branch (reference :553-595: Blur, then a 1x1 convolution of stride 2): the convolution reads every second pixel of the
        # blurred image, so the blur computes only those (upfirdn2d with down = 2, same taps and padding: the same values) and the
        # convolution runs at stride 1 on a quarter of the pixels
        if fused_skip:
            r = conv_s(down, out_mul=1.0 / SQRT2, stride=1)
        else:
            r = conv_s(blur_s(input), out_mul=1.0 / SQRT2)
        return bias_act_add(z, act2.bias, r, act2.negative_slope, act2.scale / SQRT2)


class StyleGANDiscriminator(FlatNet):
    def __init__(self, size=32, channel_multiplier=2, blur_kernel=[1, 3, 3, 1], **kwargs):
        super().__init__()
        channels = channel_table(channel_multiplier)
        blocks = [ConvLayer(3, channels[size], 1)]
        width = channels[size]
        for level in range(int(math.log(size, 2)), 2, -1):
            blocks.append(ResBlock(width, channels[2 ** (level - 1)], blur_kernel))
            width = channels[2 ** (level - 1)]
        self.convs = nn.Sequential(*blocks)
        self.stddev_group, self.stddev_feat = 4, 1
        self.final_conv = ConvLayer(width + 1, channels[4], 3)
        self.final_linear = nn.Sequential(EqualLinear(channels[4] * 4 * 4, channels[4], activation="fused_lrelu"),
                                          EqualLinear(channels[4], 1))

    def minibatch_stddev(self, x):
        """append the group-wise feature standard deviation as one more plane (reference :662-670; + 3 zero planes
        so that the channel count stays a multiple of 4 -- the next convolution's weights are padded to match)"""
        b, h, w, c = x.shape
        group = min(b, self.stddev_group)
        sd = x.view(group, -1, h, w, self.stddev_feat, c // self.stddev_feat)
        sd = torch.sqrt(sd.var(0, unbiased=False) + 1e-8)
        sd = sd.mean((1, 2, 4)).view(-1, 1, 1, self.stddev_feat).repeat(group, h, w, 1)
        pad = -(c + self.stddev_feat) % 4
        return torch.cat([x, sd] + ([x.new_zeros(b, h, w, pad)] if pad else []), 3)

    def forward(self, input):
        out = to_nhwc(input)
        blocks = list(self.convs)
        first = blocks[0]
        if len(first) == 2 and isinstance(first[0], EqualConv2d) and first[0].bias is None and fromrgb_ok(out, first[0].weight, first[1].bias):
            # the 1x1 convolution from RGB, its bias and activation as ONE write of the full-resolution tensor (round 6)
            out = fromrgb(out, first[0].weight, first[1].bias, first[0].scale, first[1].negative_slope, first[1].scale)
            blocks = blocks[1:]
        for blk in blocks:
            out = blk(out)
        out = self.final_conv(self.minibatch_stddev(out))
        out = self.final_linear[0].forward_spatial(out)
        return self.final_linear[1](out)


# names of stylegan2/model.py
Generator = StyleGANGenerator
Discriminator = StyleGANDiscriminator
