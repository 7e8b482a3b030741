"""CPU: the model-side oracle (oracle/nets.py) against vectors produced by the reference's own
MNIST_DCGAN / GOLD / TopK code (tools/gen_goldens_models.py, torch_mimicry stubbed)."""
import os

import numpy as np
import pytest
import torch

from oracle import nets as O


def expand_check(got, packed, tol):
    """Compare against the compact form written by gen_goldens_models.compact()."""
    got = np.asarray(got)
    flat = got.reshape(-1)
    if packed.shape == got.shape:
        np.testing.assert_allclose(got, packed, rtol=tol, atol=tol * (np.abs(packed).max() + 1e-30))
        return
    s, a, k = packed[0], packed[1], int(packed[2])
    scale = a / flat.size
    assert abs(flat.astype(np.float64).sum() - s) <= tol * a + 1e-12
    assert abs(np.abs(flat.astype(np.float64)).sum() - a) <= tol * a
    np.testing.assert_allclose(flat[::k][:4096], packed[3:], rtol=0, atol=tol * 50 * scale + 1e-12)


@pytest.fixture(scope="module")
def models(golden_dir):
    g = np.load(os.path.join(golden_dir, "dcgan.npz"))
    torch.manual_seed(11)
    netG = O.MNIST_DCGAN_Generator(loss_type='ns')
    netD = O.MNIST_DCGAN_Discriminator(loss_type='ns')
    netD2 = O.MNIST_DCGAN_Discriminator(loss_type='hinge', num_pack=2)
    gen = torch.Generator().manual_seed(5)
    z = torch.randn(6, 100, generator=gen)
    x = torch.rand(6, 3, 32, 32, generator=gen) * 2 - 1
    return g, netG, netD, netD2, z, x


def test_same_seed_weights_bit_identical(models):
    g, netG, netD, netD2, z, x = models
    for tag, net in (("G", netG), ("D", netD), ("D2", netD2)):
        for k, v in net.state_dict().items():
            if v.dtype.is_floating_point:
                ck = g[f"ck_{tag}_{k}"]
                assert v.double().sum().item() == ck[0] and v.double().abs().sum().item() == ck[1], (tag, k)
    assert np.array_equal(z.numpy(), g["z"])
    assert sum(p.numel() for p in netG.parameters()) == 1590048
    assert sum(p.numel() for p in netD.parameters()) == 1581937


def test_dcgan_forward_and_grads(models):
    g, netG, netD, netD2, z, x = models
    netG.eval(), netD.eval(), netD2.eval()
    with torch.no_grad():
        np.testing.assert_allclose(netG(z).numpy()[:2], g["G_eval"], atol=1e-6)
        np.testing.assert_allclose(netD(x).numpy(), g["D_eval"], atol=1e-5)
        np.testing.assert_allclose(netD(x, get_feature=True).numpy(), g["D_feature"], atol=1e-5)
        np.testing.assert_allclose(netD2(x).numpy(), g["D2_eval_pack2"], atol=1e-5)
    netG.train()
    with torch.no_grad():
        np.testing.assert_allclose(netG(z).numpy()[:2], g["G_train"], atol=1e-6)
    np.testing.assert_allclose(netG.tconv[1].running_mean.numpy(), g["G_bn1_running_mean"], atol=1e-7)
    np.testing.assert_allclose(netG.tconv[1].running_var.numpy(), g["G_bn1_running_var"], atol=1e-7)
    netG.zero_grad(), netD.zero_grad()
    loss = (netD(netG(z)).view(-1) * torch.linspace(-1, 1, 6)).sum()
    loss.backward()
    assert abs(loss.item() - float(g["loss_GD"])) < 1e-5
    expand_check(netG.fc.weight.grad.numpy(), g["gG_fc_weight"], 1e-5)
    expand_check(netG.tconv[0].weight.grad.numpy(), g["gG_tconv0"], 1e-5)
    expand_check(netG.tconv[9].weight.grad.numpy(), g["gG_tconv9"], 1e-5)
    expand_check(netG.tconv[1].weight.grad.numpy(), g["gG_bn1_weight"], 1e-5)
    expand_check(netD.conv[0].weight.grad.numpy(), g["gD_conv0"], 1e-5)
    expand_check(netD.conv[19].weight.grad.numpy(), g["gD_conv19"], 1e-5)
    expand_check(netD.out_d.weight.grad.numpy(), g["gD_out_d"], 1e-5)


def test_gold_losses_and_topk(golden_dir):
    g = np.load(os.path.join(golden_dir, "losses.npz"))
    r, f = torch.from_numpy(g["real"]), torch.from_numpy(g["fake"])
    for name in ("ns", "hinge"):
        rr, ff = r.clone().requires_grad_(True), f.clone().requires_grad_(True)
        L = O.dis_loss(name, rr, ff, gold=True)
        L.backward()
        assert abs(L.item() - float(g[f"gold_{name}_loss"])) < 1e-6
        np.testing.assert_allclose(rr.grad.numpy(), g[f"gold_{name}_dreal"], atol=1e-7)
        np.testing.assert_allclose(ff.grad.numpy(), g[f"gold_{name}_dfake"], atol=1e-7)
    rates = [O.topk_rate_at(int(s), 782) for s in g["topk_steps"]]
    assert rates == list(g["topk_rates"])
    vals = O.get_topk(f, 0.77)
    assert np.array_equal(vals.numpy(), g["topk_vals"])
    # the product's host-side TopKGenerator follows the same rule
    from diagan.models.topk_models import TopKGenerator
    t = TopKGenerator(use_topk=True)
    got = []
    for s in g["topk_steps"]:
        t.decay_topk_rate(int(s), epoch_steps=782)
        got.append(t.topk_rate)
    assert got == list(g["topk_rates"])


def test_generator_train_step_against_the_references_own_step(golden_dir):
    """Second, independent witness for the base generator step: the fixture's `gstep_*` entries were written with the
    reference's OWN in-tree restatement of the step executing (`diagan-pkg/diagan/models/mnist.py:82-152`: zero_grad, generate_images,
    netD, get_topk, compute_gan_loss, backward, optG.step, log -- tools/gen_goldens_models.py), top-k rate 0.75 (4 of 6 logits selected),
    Adam(2e-4, (0, 0.9)).  The oracle's hand-written step (oracle/nets.py: _BaseG.train_step) must give the same loss, gradients and
    post-Adam parameters from the same seeds."""
    g = np.load(os.path.join(golden_dir, "dcgan.npz"))
    torch.manual_seed(11)
    netG = O.MNIST_DCGAN_Generator(loss_type='ns', topk=True)
    netD = O.MNIST_DCGAN_Discriminator(loss_type='ns')
    O.MNIST_DCGAN_Discriminator(loss_type='hinge', num_pack=2)          # (the fixture's third network: same consumption of the seed)
    gen = torch.Generator().manual_seed(5)
    torch.randn(6, 100, generator=gen)
    x = torch.rand(6, 3, 32, 32, generator=gen) * 2 - 1
    netG.train(), netD.eval()
    optG = torch.optim.Adam(netG.parameters(), 2e-4, betas=(0.0, 0.9))
    torch.manual_seed(77)
    errG = netG.train_step((x,), netD, optG, topk_rate=0.75)
    assert abs(errG - float(g["gstep_errG"])) < 1e-6
    expand_check(netG.fc.weight.grad.numpy(), g["gstep_grad_fc"], 1e-5)
    expand_check(netG.tconv[3].weight.grad.numpy(), g["gstep_grad_tconv3"], 1e-5)
    expand_check(netG.tconv[9].weight.grad.numpy(), g["gstep_grad_tconv9"], 1e-5)
    expand_check(netG.tconv[4].bias.grad.numpy(), g["gstep_grad_bn4_bias"], 1e-5)
    n = 0
    for k, v in netG.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            continue
        ck = g[f"gstep_post_{k}"]
        assert abs(v.double().sum().item() - ck[0]) <= 1e-6 * ck[1] + 1e-9, k
        assert abs(v.double().abs().sum().item() - ck[1]) <= 1e-6 * ck[1], k
        n += 1
    assert n == 12
