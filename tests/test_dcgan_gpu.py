"""GPU: MNIST_DCGAN (BASELINE configs[0] model) on the HIP engine against
 (a) vectors produced by the REFERENCE's own classes (tests/golden/dcgan.npz), and
 (b) the CPU oracle for full train steps with injected noise / dropout masks."""
import os

import numpy as np
import pytest
import torch

from oracle import nets as O
from test_oracle_models import expand_check

pytestmark = pytest.mark.gpu


class Log:
    def __init__(self):
        self.m = {}

    def add_metric(self, name, value, group=None, precision=4):
        self.m[name] = value


def build(seed=11):
    from diagan.models.mnist import MNIST_DCGAN_Discriminator, MNIST_DCGAN_Generator
    torch.manual_seed(seed)
    netG = MNIST_DCGAN_Generator(loss_type='ns')
    netD = MNIST_DCGAN_Discriminator(loss_type='ns')
    netD2 = MNIST_DCGAN_Discriminator(loss_type='hinge', num_pack=2)
    return netG, netD, netD2


def test_same_seed_state_dict_matches_reference_checksums(golden_dir):
    g = np.load(os.path.join(golden_dir, "dcgan.npz"))
    netG, netD, netD2 = build()
    for tag, net in (("G", netG), ("D", netD), ("D2", netD2)):
        sd = net.state_dict()
        keys = [k[len(f"ck_{tag}_"):] for k in g.files if k.startswith(f"ck_{tag}_")]
        assert sorted(k for k, v in sd.items() if v.dtype.is_floating_point) == sorted(keys)
        for k in keys:
            ck = g[f"ck_{tag}_{k}"]
            # same values; the sum order differs for re-laid-out tensors, hence 1e-12 relative instead of ==
            assert abs(sd[k].double().sum().item() - ck[0]) <= 1e-12 * ck[1] and \
                abs(sd[k].double().abs().sum().item() - ck[1]) <= 1e-12 * ck[1], (tag, k)
    assert netG.count_params() == 1590048 and netD.count_params() == 1581937


def test_forward_and_grads_vs_reference_goldens(golden_dir):
    g = np.load(os.path.join(golden_dir, "dcgan.npz"))
    netG, netD, netD2 = build()
    for n in (netG, netD, netD2):
        n.to('cuda')
    gen = torch.Generator().manual_seed(5)
    z = torch.randn(6, 100, generator=gen)
    x = torch.rand(6, 3, 32, 32, generator=gen) * 2 - 1
    assert np.array_equal(z.numpy(), g["z"])
    z, x = z.cuda(), x.cuda()
    netG.eval(), netD.eval(), netD2.eval()
    np.testing.assert_allclose(netG(z).cpu().numpy()[:2], g["G_eval"], atol=1e-4)
    np.testing.assert_allclose(netD(x).cpu().numpy(), g["D_eval"], atol=1e-3)
    np.testing.assert_allclose(netD(x, get_feature=True).cpu().numpy(), g["D_feature"], atol=1e-3)
    np.testing.assert_allclose(netD2(x).cpu().numpy(), g["D2_eval_pack2"], atol=1e-3)
    netG.train()
    img, gctx = netG.forward_nhwc(z, True, save=True)
    from diagan.ops import eltwise as E
    np.testing.assert_allclose(E.nhwc_to_nchw(img, 3).cpu().numpy()[:2], g["G_train"], atol=1e-4)
    np.testing.assert_allclose(netG.tconv['1'].running_mean.cpu().numpy(), g["G_bn1_running_mean"], atol=1e-6)
    np.testing.assert_allclose(netG.tconv['1'].running_var.cpu().numpy(), g["G_bn1_running_var"], atol=1e-6)
    # loss = sum(D(G(z)) * linspace(-1, 1)), D in eval mode: gradients of G and D
    netG.zero_grad(), netD.zero_grad()
    logit, dctx = netD.forward_nhwc(img, False, save=True, need_dgrad=True, need_in_dgrad=True)
    w = torch.linspace(-1, 1, 6).cuda()
    assert abs((logit.view(-1) * w).sum().item() - float(g["loss_GD"])) < 1e-3
    g_img = netD.backward_nhwc(dctx, w.contiguous(), need_wgrad=True, need_gx=True)
    netG.backward_nhwc(gctx, g_img)
    gG, gD = netG.export_grads(), netD.export_grads()
    expand_check(gG['fc.weight'].cpu().numpy(), g["gG_fc_weight"], 2e-3)
    expand_check(gG['tconv.0.weight'].cpu().numpy(), g["gG_tconv0"], 2e-3)
    expand_check(gG['tconv.9.weight'].cpu().numpy(), g["gG_tconv9"], 2e-3)
    expand_check(gG['tconv.1.weight'].cpu().numpy(), g["gG_bn1_weight"], 2e-3)
    expand_check(gD['conv.0.weight'].cpu().numpy(), g["gD_conv0"], 2e-3)
    expand_check(gD['conv.19.weight'].cpu().numpy(), g["gD_conv19"], 2e-3)
    expand_check(netD.state_dict_grad_out_d().cpu().numpy(), g["gD_out_d"], 2e-3)


def test_train_steps_vs_oracle_with_injected_dropout():
    """Full D and G updates in training mode (BN batch stats, dropout) with the dropout masks injected on
    both sides: losses within 1e-3."""
    oG, oD, ooptG, ooptD = O.make_pair('color_mnist', 'ns', seed=3)
    from diagan.models.predefined_models import get_gan_model
    torch.manual_seed(3)
    netG, netD, optG, optD = get_gan_model('color_mnist', model='mnist_dcgan', loss_type='ns')
    assert all(torch.equal(a, b) for a, b in zip(oG.state_dict().values(), netG.state_dict().values()))
    netG.to('cuda'), netD.to('cuda')
    gen = torch.Generator().manual_seed(4)
    B = 8
    x = torch.rand(B, 3, 32, 32, generator=gen) * 2 - 1
    z = torch.randn(B, 100, generator=gen)
    shapes = [(B, 16, 16, 16), (B, 32, 16, 16), (B, 64, 8, 8), (B, 128, 8, 8), (B, 256, 4, 4), (B, 512, 4, 4)]
    masks = [[(torch.rand(s, generator=gen) >= 0.5).float() * 2 for s in shapes] for _ in range(2)]   # real, fake

    # oracle D step with the same masks: replace its Dropout modules by fixed multiplications
    class FixedDrop(torch.nn.Module):
        def __init__(self, seq):
            super().__init__()
            self.seq, self.i = seq, 0

        def forward(self, t):
            m = self.seq[self.i % len(self.seq)]
            self.i += 1
            return t * m

    drops = [i for i, m in enumerate(oD.conv) if isinstance(m, torch.nn.Dropout)]
    for li, i in enumerate(drops):
        oD.conv[i] = FixedDrop([masks[0][li], masks[1][li]])
    errD, D_x, D_Gz = oD.train_step((x, None), oG, ooptD, noise=z)

    # engine D step: same masks (NHWC), two-pass path
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    netD.zero_grad()
    from diagan.ops import eltwise as E
    out_r, cr = netD.forward_nhwc(netD.to_nhwc(x.cuda()), True, save=True, need_in_dgrad=False,
                                  drop_masks=[nh(m) for m in masks[0]])
    fake, _ = netG.generate_images_nhwc(B, noise=z.cuda(), save=False)
    out_f, cf = netD.forward_nhwc(fake, True, save=True, need_in_dgrad=False, drop_masks=[nh(m) for m in masks[1]])
    out3, dr, df = E.loss_dis(out_r, out_f, 'ns')
    assert abs(out3[0].item() - errD) < 1e-3 and abs(out3[1].item() - D_x) < 1e-3 and abs(out3[2].item() - D_Gz) < 1e-3
    netD.backward_nhwc(cr, dr, need_wgrad=True)
    netD.backward_nhwc(cf, df, need_wgrad=True)
    gr = netD.export_grads()
    for k, p in oD.named_parameters():
        kk = k.replace('conv.', 'conv.') 
        a, b = gr[kk].double().cpu(), p.grad.double()
        assert (a - b).norm() <= 2e-2 * (b.norm() + 1e-12), k


def test_trainer_smoke_color_mnist(tmp_path):
    """BASELINE configs[0] plumbing: mnist_dcgan phase-1 loop (n_dis=1) runs through LogTrainer."""
    from diagan.datasets.predefined import get_predefined_dataset
    from diagan.models.predefined_models import get_gan_model
    from diagan.trainer.trainer import LogTrainer
    torch.manual_seed(0)
    netG, netD, optG, optD = get_gan_model('color_mnist', model='mnist_dcgan', loss_type='ns')
    ds = get_predefined_dataset('color_mnist', num_data=256)
    dl = torch.utils.data.DataLoader(ds, batch_size=64, shuffle=True)
    os.environ["DIAGAN_QUIET"] = "1"
    t = LogTrainer(output_path=tmp_path, netD=netD, netG=netG, optD=optD, optG=optG, dataloader=dl, num_steps=6,
                   log_dir=str(tmp_path), n_dis=1, lr_decay='linear', device='cuda', print_steps=3, save_steps=100,
                   logit_save_steps=2, save_logit_after=2, stop_save_logit_after=6)
    t.train()
    assert t.logit_records['netD_eval'].steps == [2, 4, 6]
    d = t.logit_records['netD_eval'].to_dict()
    assert all(np.isfinite(v).all() and (v != 0).all() for v in d.values())
