"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's StyleGAN2 generator / discriminator and of the
losses / regularisers of its trainer, as plain functions of a state dict (NCHW, torch CPU, autograd to any order).

Follows /root/reference/diagan-pkg/diagan/models/stylegan2.py (file:line cited per function; identical to
stylegan2/model.py) and /root/reference/stylegan2/train_ffhq.py:63-102.  The convolutions are stated the way the
reference states them (per-sample modulated weights, grouped conv2d / conv_transpose2d); the FIR filter and the fused
activation come from oracle/stylegan_ops.py.

PINNED: tests/golden/stylegan2.npz was produced by the reference classes themselves on CPU
(tools/gen_goldens_stylegan2.py; torch.utils.cpp_extension.load stubbed, the ops take the reference's own CPU
branches) from `seeded_state` weights, and tests/test_oracle_stylegan2.py reproduces it with this file."""
import math
import zlib

import numpy as np
import torch
import torch.nn.functional as F

from oracle.stylegan_ops import fused_leaky_relu, upfirdn2d


def channel_table(mult):                                                       # stylegan2.py:385-395
    return {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * mult, 128: 128 * mult, 256: 64 * mult, 512: 32 * mult,
            1024: 16 * mult}


def seeded_state(shapes, seed):
    """Deterministic, non-degenerate parameter values for a state dict given {key: shape}: every tensor is drawn from
    its own NumPy stream (seed, crc32(key)), so the values do not depend on key order.  Biases and noise strengths
    are given non-zero values so that every term of the forward pass matters; FIR `kernel` buffers are left to the
    module."""
    out = {}
    for key, shape in shapes.items():
        if key.endswith("kernel"):
            continue
        rs = np.random.RandomState((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 32))
        v = rs.standard_normal(tuple(shape)).astype(np.float32)
        if key.endswith("modulation.bias"):
            v = 1.0 + 0.1 * v
        elif key.endswith("bias") or key.endswith("noise.weight"):
            v = 0.1 * v
        elif key.startswith("style.") and key.endswith("weight"):
            v = v / 0.01                                                        # EqualLinear init with lr_mul 0.01
        out[key] = torch.from_numpy(v)
    return out


def _blur_kernel(taps=(1, 3, 3, 1), gain=1.0):                                  # make_kernel, stylegan2.py:22-30
    k = torch.tensor(taps, dtype=torch.float32)
    k = k[None, :] * k[:, None]
    return k / k.sum() * gain


def _fir(x, k, up=1, down=1, pad=(0, 0)):
    return upfirdn2d(x, k.to(x.dtype), up, up, down, down, pad[0], pad[1], pad[0], pad[1])


def equal_linear(sd, p, x, lr_mul=1.0, activation=False):                       # stylegan2.py:132-161
    w = sd[p + "weight"]
    scale = (1 / math.sqrt(w.shape[1])) * lr_mul
    if activation:
        return fused_leaky_relu(F.linear(x, w * scale), sd[p + "bias"] * lr_mul)
    return F.linear(x, w * scale, bias=sd[p + "bias"] * lr_mul)


def mapping(sd, z, n_mlp=8, lr_mlp=0.01):                                       # stylegan2.py:14-19, 371-380
    x = z * torch.rsqrt(torch.mean(z ** 2, dim=1, keepdim=True) + 1e-8)
    for i in range(n_mlp):
        x = equal_linear(sd, f"style.{i + 1}.", x, lr_mlp, True)
    return x


def modulated_conv(sd, p, x, style, demodulate=True, upsample=False):           # stylegan2.py:224-265
    W = sd[p + "weight"]                                                        # [1,Co,Ci,k,k]
    _, co, ci, k, _ = W.shape
    b, _, h, w = x.shape
    s = equal_linear(sd, p + "modulation.", style).view(b, 1, ci, 1, 1)
    weight = (1 / math.sqrt(ci * k * k)) * W * s
    if demodulate:
        demod = torch.rsqrt(weight.pow(2).sum([2, 3, 4]) + 1e-8)
        weight = weight * demod.view(b, co, 1, 1, 1)
    if upsample:
        wt = weight.transpose(1, 2).reshape(b * ci, co, k, k)
        out = F.conv_transpose2d(x.reshape(1, b * ci, h, w), wt, padding=0, stride=2, groups=b)
        out = out.view(b, co, out.shape[2], out.shape[3])
        pp = (4 - 2) - (k - 1)
        return _fir(out, _blur_kernel(gain=4.0), pad=((pp + 1) // 2 + 1, pp // 2 + 1))
    out = F.conv2d(x.reshape(1, b * ci, h, w), weight.view(b * co, ci, k, k), padding=k // 2, groups=b)
    return out.view(b, co, out.shape[2], out.shape[3])


def styled_conv(sd, p, x, style, noise, upsample=False):                        # stylegan2.py:268-329
    out = modulated_conv(sd, p + "conv.", x, style, upsample=upsample)
    out = out + sd[p + "noise.weight"] * noise
    return fused_leaky_relu(out, sd[p + "activate.bias"])


def to_rgb(sd, p, x, style, skip=None):                                         # stylegan2.py:332-351
    out = modulated_conv(sd, p + "conv.", x, style, demodulate=False) + sd[p + "bias"]
    if skip is not None:
        out = out + _fir(skip, _blur_kernel(gain=4.0), up=2, pad=(2, 1))
    return out


def generator(sd, size, styles, noise=None, input_is_latent=False, inject_index=None):
    """StyleGANGenerator.forward (stylegan2.py:479-550) with explicit noise maps (default: the `noises` buffers);
    returns (image, latent [B, n_latent, style_dim])."""
    log_size = int(math.log(size, 2))
    n_latent = log_size * 2 - 2
    if noise is None:
        noise = [sd[f"noises.noise_{i}"] for i in range((log_size - 2) * 2 + 1)]
    if not input_is_latent:
        styles = [mapping(sd, s) for s in styles]
    if len(styles) < 2:
        latent = styles[0].unsqueeze(1).repeat(1, n_latent, 1) if styles[0].dim() < 3 else styles[0]
    else:
        latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                            styles[1].unsqueeze(1).repeat(1, n_latent - inject_index, 1)], 1)
    out = sd["input.input"].repeat(latent.shape[0], 1, 1, 1)
    out = styled_conv(sd, "conv1.", out, latent[:, 0], noise[0])
    skip = to_rgb(sd, "to_rgb1.", out, latent[:, 1])
    i = 1
    for level in range(log_size - 2):
        out = styled_conv(sd, f"convs.{2 * level}.", out, latent[:, i], noise[i], upsample=True)
        out = styled_conv(sd, f"convs.{2 * level + 1}.", out, latent[:, i + 1], noise[i + 1])
        skip = to_rgb(sd, f"to_rgbs.{level}.", out, latent[:, i + 2], skip)
        i += 2
    return skip, latent


def conv_layer(sd, p, x, downsample=False, activate=True):                      # stylegan2.py:553-595
    i = 1 if downsample else 0                      # nn.Sequential positions: [Blur,] EqualConv2d [, FusedLeakyReLU]
    w = sd[f"{p}{i}.weight"]
    k = w.shape[2]
    if downsample:
        pp = (4 - 2) + (k - 1)
        x = _fir(x, _blur_kernel(), pad=((pp + 1) // 2, pp // 2))
    x = F.conv2d(x, w * (1 / math.sqrt(w.shape[1] * k * k)), bias=sd.get(f"{p}{i}.bias"),
                 stride=2 if downsample else 1, padding=0 if downsample else k // 2)
    if activate:
        x = fused_leaky_relu(x, sd.get(f"{p}{i + 1}.bias"))
    return x


def discriminator(sd, size, x):                                                 # stylegan2.py:619-678
    out = conv_layer(sd, "convs.0.", x)
    for j in range(int(math.log(size, 2)) - 2):
        p = f"convs.{j + 1}."
        y = conv_layer(sd, p + "conv2.", conv_layer(sd, p + "conv1.", out), downsample=True)
        out = (y + conv_layer(sd, p + "skip.", out, downsample=True, activate=False)) / math.sqrt(2)
    b, c, h, w = out.shape
    group = min(b, 4)
    sdv = out.view(group, -1, 1, c, h, w)
    sdv = torch.sqrt(sdv.var(0, unbiased=False) + 1e-8)
    sdv = sdv.mean([2, 3, 4], keepdims=True).squeeze(2).repeat(group, 1, h, w)
    out = conv_layer(sd, "final_conv.", torch.cat([out, sdv], 1))
    out = equal_linear(sd, "final_linear.0.", out.view(b, -1), activation=True)
    return equal_linear(sd, "final_linear.1.", out)


# ---- losses / regularisers of the trainer (stylegan2/train_ffhq.py:63-102) ----------------------------------------
def d_logistic_loss(real_pred, fake_pred):
    return F.softplus(-real_pred).mean() + F.softplus(fake_pred).mean()


def g_nonsaturating_loss(fake_pred):
    return F.softplus(-fake_pred).mean()


def d_r1_loss(real_pred, real_img):
    grad_real, = torch.autograd.grad(outputs=real_pred.sum(), inputs=real_img, create_graph=True)
    return grad_real.pow(2).reshape(grad_real.shape[0], -1).sum(1).mean()


def g_path_regularize(fake_img, latents, mean_path_length, pl_noise, decay=0.01):
    """`pl_noise` is the reference's `torch.randn_like(fake_img)` made explicit"""
    noise = pl_noise / math.sqrt(fake_img.shape[2] * fake_img.shape[3])
    grad, = torch.autograd.grad(outputs=(fake_img * noise).sum(), inputs=latents, create_graph=True)
    path_lengths = torch.sqrt(grad.pow(2).sum(2).mean(1))
    path_mean = mean_path_length + decay * (path_lengths.mean() - mean_path_length)
    path_penalty = (path_lengths - path_mean).pow(2).mean()
    return path_penalty, path_mean.detach(), path_lengths


# ---- parameter / buffer inventory (names and shapes of the reference modules' state_dict) -------------------------
def generator_shapes(size, style_dim=512, n_mlp=8, mult=2):
    ch = channel_table(mult)
    log_size = int(math.log(size, 2))
    shapes = {}
    for i in range(n_mlp):
        shapes[f"style.{i + 1}.weight"], shapes[f"style.{i + 1}.bias"] = (style_dim, style_dim), (style_dim,)
    shapes["input.input"] = (1, ch[4], 4, 4)

    def styled(p, ci, co, k=3):
        shapes[p + "conv.weight"] = (1, co, ci, k, k)
        shapes[p + "conv.modulation.weight"], shapes[p + "conv.modulation.bias"] = (ci, style_dim), (ci,)

    def layer(p, ci, co):
        styled(p, ci, co)
        shapes[p + "noise.weight"], shapes[p + "activate.bias"] = (1,), (co,)

    def rgb(p, ci):
        styled(p, ci, 3, 1)
        shapes[p + "bias"] = (1, 3, 1, 1)

    layer("conv1.", ch[4], ch[4])
    rgb("to_rgb1.", ch[4])
    width = ch[4]
    for level in range(3, log_size + 1):
        nxt = ch[2 ** level]
        layer(f"convs.{2 * (level - 3)}.", width, nxt)
        layer(f"convs.{2 * (level - 3) + 1}.", nxt, nxt)
        rgb(f"to_rgbs.{level - 3}.", nxt)
        width = nxt
    for i in range((log_size - 2) * 2 + 1):
        res = 2 ** ((i + 5) // 2)
        shapes[f"noises.noise_{i}"] = (1, 1, res, res)
    return shapes


def discriminator_shapes(size, mult=2):
    ch = channel_table(mult)
    shapes = {"convs.0.0.weight": (ch[size], 3, 1, 1), "convs.0.1.bias": (ch[size],)}
    width = ch[size]
    for j, level in enumerate(range(int(math.log(size, 2)), 2, -1)):
        nxt = ch[2 ** (level - 1)]
        p = f"convs.{j + 1}."
        shapes[p + "conv1.0.weight"], shapes[p + "conv1.1.bias"] = (width, width, 3, 3), (width,)
        shapes[p + "conv2.1.weight"], shapes[p + "conv2.2.bias"] = (nxt, width, 3, 3), (nxt,)
        shapes[p + "skip.1.weight"] = (nxt, width, 1, 1)
        width = nxt
    shapes["final_conv.0.weight"], shapes["final_conv.1.bias"] = (ch[4], width + 1, 3, 3), (ch[4],)
    shapes["final_linear.0.weight"], shapes["final_linear.0.bias"] = (ch[4], ch[4] * 16), (ch[4],)
    shapes["final_linear.1.weight"], shapes["final_linear.1.bias"] = (1, ch[4]), (1,)
    return shapes
