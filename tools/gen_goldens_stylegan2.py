#!/usr/bin/env python
"""Golden vectors for the StyleGAN2 row from the reference's OWN classes, run on CPU.

`StyleGANGenerator` / `StyleGANDiscriminator` are loaded from /root/reference/diagan-pkg/diagan/models/stylegan2.py
and the loss / regulariser functions from /root/reference/stylegan2/train_ffhq.py by file, with
torch.utils.cpp_extension.load stubbed BEFORE the import: nothing is JIT-compiled or written next to the reference
sources, and on CPU tensors the reference's ops take their own CPU branches (`upfirdn2d_native`, F.leaky_relu).
Weights come from oracle.stylegan2.seeded_state (a seed, not a file), so the fixture holds only inputs and outputs.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_goldens_stylegan2.py        (build container only)
"""
import ast
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..")
OUT = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

from oracle.stylegan2 import seeded_state  # noqa: E402

SIZE, SEED_G, SEED_D = 16, 11, 12


def load_file(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def reference_modules():
    import torch.utils.cpp_extension as cpp
    real_load = cpp.load
    cpp.load = lambda *a, **k: types.SimpleNamespace()
    try:
        opdir = os.path.join(REF, "diagan-pkg/diagan/models/op")
        for pkg in ("diagan", "diagan.models", "diagan.models.op"):          # empty namespace shells
            sys.modules[pkg] = types.ModuleType(pkg)
        fa = load_file("diagan.models.op.fused_act", os.path.join(opdir, "fused_act.py"))
        up = load_file("diagan.models.op.upfirdn2d", os.path.join(opdir, "upfirdn2d.py"))
        op = sys.modules["diagan.models.op"]
        op.FusedLeakyReLU, op.fused_leaky_relu, op.upfirdn2d = fa.FusedLeakyReLU, fa.fused_leaky_relu, up.upfirdn2d
        model = load_file("_ref_stylegan2", os.path.join(REF, "diagan-pkg/diagan/models/stylegan2.py"))
    finally:
        cpp.load = real_load
    # the trainer module imports lmdb / torchvision / wandb at the top: take only its pure loss functions
    src = open(os.path.join(REF, "stylegan2/train_ffhq.py")).read()
    wanted = {"d_logistic_loss", "d_r1_loss", "g_nonsaturating_loss", "g_path_regularize"}
    tree = ast.parse(src)
    tree.body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    import math
    from torch import autograd
    from torch.nn import functional as F
    ns = dict(math=math, torch=torch, autograd=autograd, F=F)
    exec(compile(tree, "train_ffhq.py", "exec"), ns)
    return model, types.SimpleNamespace(**{k: ns[k] for k in wanted})


def load_seeded(net, seed):
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(seeded_state(shapes, seed), strict=False)


def grad_norms(net):
    return {k: float(p.grad.double().norm()) for k, p in net.named_parameters() if p.grad is not None}


def main():
    model, losses = reference_modules()
    torch.manual_seed(0)
    G = model.StyleGANGenerator(size=SIZE)
    D = model.StyleGANDiscriminator(size=SIZE)
    load_seeded(G, SEED_G)
    load_seeded(D, SEED_D)
    gen = torch.Generator().manual_seed(5)
    out = dict(size=np.array(SIZE), seed_g=np.array(SEED_G), seed_d=np.array(SEED_D))

    def put_norms(tag, norms):
        out[f"{tag}_keys"] = np.array(sorted(norms))
        out[f"{tag}_norms"] = np.array([norms[k] for k in sorted(norms)])

    # generator: single latent code, fixed noise buffers; and style mixing at a fixed crossover
    z1, z2 = torch.randn(4, 512, generator=gen), torch.randn(4, 512, generator=gen)
    with torch.no_grad():
        img, lat = G([z1], return_latents=True, randomize_noise=False)
        img_mix, _ = G([z1, z2], inject_index=3, randomize_noise=False)
    out.update(z1=z1.numpy(), z2=z2.numpy(), g_image=img.numpy(), g_latent=lat.numpy(), g_image_mix=img_mix.numpy())

    # discriminator logits, logistic loss and its parameter gradients
    x = torch.randn(8, 3, SIZE, SIZE, generator=gen)
    xf = torch.randn(8, 3, SIZE, SIZE, generator=gen)
    D.zero_grad()
    real_pred, fake_pred = D(x), D(xf)
    d_loss = losses.d_logistic_loss(real_pred, fake_pred)
    d_loss.backward()
    out.update(d_real=x.numpy(), d_fake=xf.numpy(), d_real_pred=real_pred.detach().numpy(),
               d_fake_pred=fake_pred.detach().numpy(), d_loss=np.array(d_loss.item()))
    put_norms("d_loss_grad", grad_norms(D))
    out["d_loss_grad_last"] = D.final_linear[1].weight.grad.numpy().copy()

    # R1 (train_ffhq.py:238-246): second-order gradients of the discriminator
    D.zero_grad()
    xr = x.clone().requires_grad_(True)
    real_pred = D(xr)
    r1 = losses.d_r1_loss(real_pred, xr)
    (10.0 / 2 * r1 * 16 + 0 * real_pred[0]).backward()
    out["r1"] = np.array(r1.item())
    put_norms("r1_grad", grad_norms(D))
    out["r1_grad_first"] = D.convs[0][0].weight.grad.numpy().copy()

    # generator loss through the discriminator
    G.zero_grad()
    for p in D.parameters():
        p.requires_grad_(False)
    fake, _ = G([z1], randomize_noise=False)
    g_loss = losses.g_nonsaturating_loss(D(fake))
    g_loss.backward()
    out["g_loss"] = np.array(g_loss.item())
    put_norms("g_loss_grad", grad_norms(G))
    out["g_loss_grad_rgb_bias"] = G.to_rgbs[-1].bias.grad.numpy().copy()

    # path-length regularisation (train_ffhq.py:267-284): second-order gradients of the generator
    G.zero_grad()
    zp = torch.randn(2, 512, generator=gen)
    pl_noise = torch.randn(2, 3, SIZE, SIZE, generator=gen)
    fake, latents = G([zp], return_latents=True, randomize_noise=False)
    real_randn_like = torch.randn_like
    torch.randn_like = lambda t: pl_noise                  # the reference draws its projection noise inside
    try:
        path_loss, mean_path, path_lengths = losses.g_path_regularize(fake, latents, 0.3)
    finally:
        torch.randn_like = real_randn_like
    (2.0 * 4 * path_loss + 0 * fake[0, 0, 0, 0]).backward()
    out.update(zp=zp.numpy(), pl_noise=pl_noise.numpy(), path_loss=np.array(path_loss.item()),
               mean_path=np.array(mean_path.item()), path_lengths=path_lengths.detach().numpy())
    put_norms("path_grad", grad_norms(G))
    out["path_grad_input"] = G.input.input.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "stylegan2.npz"), **out)
    print({k: (v.shape if hasattr(v, 'shape') else v) for k, v in out.items() if not k.endswith('keys')})


def compact(t, limit=4096):
    """(sum, |sum|, stride, every stride-th value) of a large tensor: the form tests/test_oracle_models.expand_check reads"""
    a = np.asarray(t.detach().cpu().numpy() if torch.is_tensor(t) else t, dtype=np.float64).reshape(-1)
    if a.size <= limit:
        return a.astype(np.float32)
    k = -(-a.size // 4096)
    return np.concatenate([[a.sum(), np.abs(a).sum(), float(k)], a[::k][:4096]])


def main_256(size=256, cm=1, batch=2):
    """BASELINE configs[4]'s resolution (VERDICT r1 Missing 7): the 256x256 pyramid, whose upper layers have Ci != Co
    (512 -> 256 -> 128 -> 64 channels at channel_multiplier 1; diagan-pkg/diagan/models/stylegan2.py:224-265) -- never
    reached by the size-16 / 32 vectors, where every layer has 512 channels.  Reduced to what fits a fixture: batch 2,
    channel_multiplier 1, images / gradients as checksums + strided samples, per-parameter gradient norms in full."""
    model, losses = reference_modules()
    torch.manual_seed(0)
    G = model.StyleGANGenerator(size=size, channel_multiplier=cm)
    D = model.StyleGANDiscriminator(size=size, channel_multiplier=cm)
    load_seeded(G, SEED_G + 100)
    load_seeded(D, SEED_D + 100)
    gen = torch.Generator().manual_seed(7)
    out = dict(size=np.array(size), channel_multiplier=np.array(cm), seed_g=np.array(SEED_G + 100),
               seed_d=np.array(SEED_D + 100), batch=np.array(batch))

    def put_norms(tag, norms):
        out[f"{tag}_keys"] = np.array(sorted(norms))
        out[f"{tag}_norms"] = np.array([norms[k] for k in sorted(norms)])

    z1, z2 = torch.randn(batch, 512, generator=gen), torch.randn(batch, 512, generator=gen)
    with torch.no_grad():
        img, lat = G([z1], return_latents=True, randomize_noise=False)
        img_mix, _ = G([z1, z2], inject_index=5, randomize_noise=False)
    out.update(z1=z1.numpy(), z2=z2.numpy(), g_image=compact(img), g_latent=lat.numpy()[:, :2], g_image_mix=compact(img_mix),
               g_image_corner=img[:, :, :8, :8].numpy())
    # discriminator on the generator's own (detached) images and on seeded "real" ones
    x = torch.randn(batch, 3, size, size, generator=gen).clamp_(-2, 2) * 0.5
    D.zero_grad()
    real_pred, fake_pred = D(x), D(img.detach())
    d_loss = losses.d_logistic_loss(real_pred, fake_pred)
    d_loss.backward()
    out.update(d_real_seed=np.array(7), d_real_pred=real_pred.detach().numpy(), d_fake_pred=fake_pred.detach().numpy(),
               d_loss=np.array(d_loss.item()), d_real_check=compact(x))
    put_norms("d_loss_grad", grad_norms(D))
    # R1
    D.zero_grad()
    xr = x.clone().requires_grad_(True)
    real_pred = D(xr)
    r1 = losses.d_r1_loss(real_pred, xr)
    (10.0 / 2 * r1 * 16 + 0 * real_pred[0]).backward()
    out["r1"] = np.array(r1.item())
    put_norms("r1_grad", grad_norms(D))
    # generator loss through the discriminator
    G.zero_grad()
    for p in D.parameters():
        p.requires_grad_(False)
    fake, _ = G([z1], randomize_noise=False)
    g_loss = losses.g_nonsaturating_loss(D(fake))
    g_loss.backward()
    out["g_loss"] = np.array(g_loss.item())
    put_norms("g_loss_grad", grad_norms(G))
    # path-length regularisation on one image
    G.zero_grad()
    zp = torch.randn(1, 512, generator=gen)
    pl_noise = torch.randn(1, 3, size, size, generator=gen)
    fake, latents = G([zp], return_latents=True, randomize_noise=False)
    real_randn_like = torch.randn_like
    torch.randn_like = lambda t: pl_noise
    try:
        path_loss, mean_path, path_lengths = losses.g_path_regularize(fake, latents, 0.3)
    finally:
        torch.randn_like = real_randn_like
    (2.0 * 4 * path_loss + 0 * fake[0, 0, 0, 0]).backward()
    out.update(zp=zp.numpy(), pl_noise_seed=np.array(7), pl_noise_check=compact(pl_noise), path_loss=np.array(path_loss.item()),
               mean_path=np.array(mean_path.item()), path_lengths=path_lengths.detach().numpy())
    put_norms("path_grad", grad_norms(G))

    # How well-conditioned each NoiseInjection strength's gradient is.  The strength is ONE scalar whose gradient sums
    # g * noise over B*C*H*W elements of either sign (up to 8 M terms, |sum| down to 1/5000 of the sum of |terms|): any
    # fp32 evaluation is only good to ~1e-6 of the sum of |terms|.  That sum is recorded here so the consumer can state
    # its tolerance against it: the strength is expanded to one value per element, whose .grad then holds the terms.
    def noise_term_sums(z, backward):
        mods = {n: m for n, m in G.named_modules() if type(m).__name__ == "NoiseInjection"}
        shapes, saved = {}, {n: m.weight for n, m in mods.items()}
        hooks = [m.register_forward_hook(lambda mod, inp, o, n=n: shapes.__setitem__(n, tuple(o.shape))) for n, m in mods.items()]
        with torch.no_grad():
            G([z], randomize_noise=False)
        for h in hooks:
            h.remove()
        for n, m in mods.items():
            m.weight = torch.nn.Parameter(saved[n].data.reshape(1, 1, 1, 1).expand(shapes[n]).clone())
        G.zero_grad()
        backward()
        sums = {n + ".weight": (float(m.weight.grad.double().abs().sum()), float(m.weight.grad.double().sum())) for n, m in mods.items()}
        for n, m in mods.items():
            m.weight = saved[n]
        return sums

    def g_pass():
        fake, _ = G([z1], randomize_noise=False)
        losses.g_nonsaturating_loss(D(fake)).backward()

    def path_pass():
        fake, latents = G([zp], return_latents=True, randomize_noise=False)
        torch.randn_like = lambda t: pl_noise
        try:
            path_loss, _, _ = losses.g_path_regularize(fake, latents, 0.3)
        finally:
            torch.randn_like = real_randn_like
        (2.0 * 4 * path_loss + 0 * fake[0, 0, 0, 0]).backward()

    for tag, sums in (("g_loss_grad", noise_term_sums(z1, g_pass)), ("path_grad", noise_term_sums(zp, path_pass))):
        keys = [str(k) for k in out[f"{tag}_keys"]]
        for k, (a, s_) in sums.items():                       # the expanded pass must reproduce the scalar gradient
            assert abs(abs(s_) - out[f"{tag}_norms"][keys.index(k)]) <= 2e-5 * a, (tag, k, s_, a)
        out[f"{tag}_noise_keys"] = np.array(sorted(sums))
        out[f"{tag}_noise_abs"] = np.array([sums[k][0] for k in sorted(sums)])
    np.savez_compressed(os.path.join(OUT, "stylegan2_256.npz"), **out)
    print({k: (v.shape if hasattr(v, 'shape') else v) for k, v in out.items() if not k.endswith('keys')})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "256":
        main_256()
    else:
        main()
