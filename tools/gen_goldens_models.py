#!/usr/bin/env python
"""Golden vectors from the reference's own model-side code that is importable with torch_mimicry
stubbed out: MNIST_DCGAN_{Generator,Discriminator} (diagan-pkg/diagan/models/mnist.py),
GOLD losses (gold_reweight_models.py:10-61), TopKGenerator (topk_models.py:15-38).

The stubs only provide base classes / names; every number stored below is computed by reference code
(or by torch itself).  Weights are NOT stored (12 MB): the vectors hold per-tensor checksums of the
reference model built under a fixed seed, and the oracle must rebuild bit-identical weights from the
same seed (same RNG consumption order) before its outputs are compared.
Run through tools/gen_goldens.py (build container only).
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference/diagan-pkg"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
sys.dont_write_bytecode = True
if REF not in sys.path:
    sys.path.insert(0, REF)


def install_stubs():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    class BaseGenerator(nn.Module):
        def __init__(self, nz, ngf, bottom_width, loss_type, **kw):
            super().__init__()
            self.nz, self.ngf, self.bottom_width, self.loss_type = nz, ngf, bottom_width, loss_type

        # The two base-class methods the reference's in-tree train_step (mnist.py:82-152) calls.  They belong to torch_mimicry 0.1.16
        # (nets/gan/gan.py: generate_images draws torch.randn((num_images, nz)) and runs forward; compute_gan_loss with loss_type 'ns' is
        # modules/losses.py: ns_loss_gen = BCE-with-logits against ones) -- restated here because the package is absent; the STEP that
        # the second witness pins (gen_gstep below) is the reference's own code.
        def generate_images(self, num_images, device=None):
            return self.forward(torch.randn((num_images, self.nz), device=device))

        def compute_gan_loss(self, output):
            assert self.loss_type == 'ns'
            return F.binary_cross_entropy_with_logits(output, torch.ones_like(output))

    class BaseDiscriminator(nn.Module):
        def __init__(self, ndf, loss_type, **kw):
            super().__init__()
            self.ndf, self.loss_type = ndf, loss_type

    mmc = mod("torch_mimicry")
    nets = mod("torch_mimicry.nets")
    gan_pkg = mod("torch_mimicry.nets.gan")
    gan = mod("torch_mimicry.nets.gan.gan")
    gan.BaseGenerator, gan.BaseDiscriminator = BaseGenerator, BaseDiscriminator
    gan_pkg.gan = gan
    nets.gan = gan_pkg
    for fam, names in (("sngan", ["SNGANGenerator32", "SNGANGenerator64", "SNGANDiscriminator32", "SNGANDiscriminator64"]),
                       ("infomax_gan", ["InfoMaxGANGenerator32", "InfoMaxGANGenerator64", "InfoMaxGANDiscriminator32",
                                        "InfoMaxGANDiscriminator64"]),
                       ("ssgan", ["SSGANGenerator32", "SSGANGenerator64", "SSGANDiscriminator32", "SSGANDiscriminator64"])):
        m = mod(f"torch_mimicry.nets.{fam}")
        for n in names:
            setattr(m, n, type(n, (nn.Module,), {}))
        setattr(nets, fam, m)
    modules = mod("torch_mimicry.modules")
    losses = mod("torch_mimicry.modules.losses")
    losses.hinge_loss_dis = lambda output_fake, output_real: None      # never called for the goldens
    losses.minimax_loss_dis = lambda output_fake, output_real, **kw: None
    modules.losses = losses
    mmc.nets, mmc.modules = nets, modules


def checksums(sd):
    return {k: np.array([v.double().sum().item(), v.double().abs().sum().item()]) for k, v in sd.items()
            if v.dtype.is_floating_point}


def compact(a, n=4096):
    """Large arrays are stored as {sum, abs-sum, every k-th element}: enough to pin them, small on disk."""
    a = np.asarray(a)
    flat = a.reshape(-1)
    if flat.size <= n:
        return a
    k = flat.size // n
    return np.concatenate([[flat.astype(np.float64).sum(), np.abs(flat.astype(np.float64)).sum(), k], flat[::k][:n]])


def main():
    install_stubs()
    from diagan.models.mnist import MNIST_DCGAN_Discriminator, MNIST_DCGAN_Generator
    from diagan.models import gold_reweight_models as gold
    from diagan.models.topk_models import TopKGenerator

    out = {}
    torch.manual_seed(11)
    netG = MNIST_DCGAN_Generator(loss_type='ns')
    netD = MNIST_DCGAN_Discriminator(loss_type='ns')
    netD2 = MNIST_DCGAN_Discriminator(loss_type='hinge', num_pack=2)
    for tag, net in (("G", netG), ("D", netD), ("D2", netD2)):
        for k, v in checksums(net.state_dict()).items():
            out[f"ck_{tag}_{k}"] = v
    g = torch.Generator().manual_seed(5)
    z = torch.randn(6, 100, generator=g)
    x = torch.rand(6, 3, 32, 32, generator=g) * 2 - 1
    out["z"] = z.numpy()          # x is regenerated from the same generator seed in the tests
    # eval mode: BN running stats, no dropout -> deterministic
    netG.eval(), netD.eval(), netD2.eval()
    with torch.no_grad():
        out["G_eval"] = netG(z).numpy()[:2]
        out["D_eval"] = netD(x).numpy()
        out["D_feature"] = netD(x, get_feature=True).numpy()
        out["D2_eval_pack2"] = netD2(x).numpy()
    # train-mode generator forward (BN batch statistics) + running-stat update
    netG.train()
    with torch.no_grad():
        out["G_train"] = netG(z).numpy()[:2]
    out["G_bn1_running_mean"] = netG.tconv[1].running_mean.numpy().copy()
    out["G_bn1_running_var"] = netG.tconv[1].running_var.numpy().copy()
    # gradients through G (train mode) and D (eval mode: no dropout RNG) of a fixed scalar
    netG.zero_grad(), netD.zero_grad()
    img = netG(z)
    loss = (netD(img).view(-1) * torch.linspace(-1, 1, 6)).sum()
    loss.backward()
    out["loss_GD"] = np.array(loss.item())
    out["gG_fc_weight"] = compact(netG.fc.weight.grad.numpy().copy())
    out["gG_tconv0"] = compact(netG.tconv[0].weight.grad.numpy().copy())
    out["gG_tconv9"] = compact(netG.tconv[9].weight.grad.numpy().copy())
    out["gG_bn1_weight"] = compact(netG.tconv[1].weight.grad.numpy().copy())
    out["gD_conv0"] = compact(netD.conv[0].weight.grad.numpy().copy())
    out["gD_conv19"] = compact(netD.conv[19].weight.grad.numpy().copy())
    out["gD_out_d"] = compact(netD.out_d.weight.grad.numpy().copy())
    # ---- second witness for the generator's train step: the reference's OWN in-tree restatement of the step
    # (mnist.py:82-152: zero_grad, generate_images, netD, get_topk, compute_gan_loss, backward, optG.step, log) executes on the pair
    # above -- G in train mode, D in eval mode (no dropout draw), top-k rate 0.75 so that get_topk really selects (4 of 6 logits)
    class Log:
        def __init__(self):
            self.m = {}

        def add_metric(self, name, value, group=None, **kw):
            self.m[name] = float(value)

    netG.train(), netD.eval()
    netG.topk_rate = 0.75
    optG = torch.optim.Adam(netG.parameters(), 2e-4, betas=(0.0, 0.9))
    torch.manual_seed(77)                  # the noise generate_images draws
    log = netG.train_step(real_batch=(x,), netD=netD, optG=optG, log_data=Log(), device=torch.device('cpu'))
    out["gstep_errG"] = np.array(log.m['errG'])
    out["gstep_grad_fc"] = compact(netG.fc.weight.grad.numpy().copy())
    out["gstep_grad_tconv3"] = compact(netG.tconv[3].weight.grad.numpy().copy())
    out["gstep_grad_tconv9"] = compact(netG.tconv[9].weight.grad.numpy().copy())
    out["gstep_grad_bn4_bias"] = compact(netG.tconv[4].bias.grad.numpy().copy())
    for k, v in checksums({k: v for k, v in netG.state_dict().items() if 'running' not in k and 'num_batches' not in k}).items():
        out[f"gstep_post_{k}"] = v
    np.savez_compressed(os.path.join(OUT, "dcgan.npz"), **out)

    # GOLD losses + top-k
    lo = {}
    g = torch.Generator().manual_seed(9)
    r = torch.randn(64, 1, generator=g) * 2
    f = torch.randn(64, 1, generator=g) * 2
    lo["real"], lo["fake"] = r.numpy(), f.numpy()
    for name, fn in (("ns", gold.gold_reweighted_minimax_loss_dis), ("hinge", gold.gold_reweighted_hinge_loss_dis)):
        rr, ff = r.clone().requires_grad_(True), f.clone().requires_grad_(True)
        L = fn(output_fake=ff, output_real=rr)
        L.backward()
        lo[f"gold_{name}_loss"] = np.array(L.item())
        lo[f"gold_{name}_dreal"], lo[f"gold_{name}_dfake"] = rr.grad.numpy(), ff.grad.numpy()
    t = TopKGenerator(use_topk=True)
    rates = []
    for step in (0, 781, 782, 7820, 782 * 68, 782 * 69, 782 * 500):
        t.decay_topk_rate(step, epoch_steps=782)
        rates.append(t.topk_rate)
    lo["topk_steps"] = np.array([0, 781, 782, 7820, 782 * 68, 782 * 69, 782 * 500])
    lo["topk_rates"] = np.array(rates, dtype=np.float64)
    t.topk_rate = 0.77
    vals, idx = t.get_topk(f, return_index=True)
    lo["topk_vals"], lo["topk_idx"] = vals.numpy(), idx.numpy()
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **lo)


if __name__ == "__main__":
    main()


def gen_drs():
    """DRS acceptance (diagan-pkg/diagan/models/drs.py:36-55) on scripted logits: which indices survive."""
    from diagan.models.drs import DRS

    class FakeG:
        def generate_images(self, n, device=None):
            return torch.arange(n, dtype=torch.float32).view(n, 1)

    class FakeD:
        def __init__(self):
            self.g = torch.Generator().manual_seed(23)

        def __call__(self, imgs):
            return torch.randn(imgs.shape[0], 1, generator=self.g) * 1.5

    np.random.seed(7)
    drs = DRS(FakeG(), FakeD(), device='cpu')
    out = {"maximum_after_init": np.array(drs.maximum)}
    g = torch.Generator().manual_seed(29)
    for i in range(3):
        ldr = (torch.randn(256, 1, generator=g) * 2).numpy()
        kept = drs.sub_rejection_sampler(torch.arange(256, dtype=torch.float32).view(256, 1), ldr)
        out[f"ldr{i}"], out[f"kept{i}"] = ldr, kept.numpy().reshape(-1)
    out["maximum_final"] = np.array(drs.maximum)
    np.savez_compressed(os.path.join(OUT, "drs.npz"), **out)


if __name__ == "__main__":
    gen_drs()


def gen_pr():
    """Precision / recall (diagan-pkg/diagan/trainer/compute_pr.py) on seeded feature sets, device='cpu'."""
    import contextlib
    import io
    from diagan.trainer import compute_pr as ref
    rng = np.random.default_rng(11)
    real = rng.normal(size=(384, 64)).astype(np.float32)
    fake = (rng.normal(size=(320, 64)) * 1.1 + 0.25).astype(np.float32)
    with contextlib.redirect_stdout(io.StringIO()):
        pr = ref.compute_pr(real, fake, nearest_k=5, device='cpu')
        part = ref.compute_partial_recall(real[:100], fake, nearest_k=5, device='cpu')
    np.savez_compressed(os.path.join(OUT, "pr.npz"), real=real, fake=fake, nearest_k=5,
                        dist=ref.compute_pairwise_distance(real[:64], fake[:48], device='cpu'),
                        radii=ref.compute_nearest_neighbour_distances(real, 5, device='cpu'),
                        kth=ref.get_kth_value(np.abs(real[:32]), 3, device='cpu'),
                        precision=pr['precision'], recall=pr['recall'], partial_recall=part['recall'])


if __name__ == "__main__":
    gen_pr()
