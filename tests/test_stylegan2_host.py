"""CPU: host-side logic of the StyleGAN2 row -- module tree / state-dict inventory, sampler choice, mixing noise,
EMA -- no kernels involved."""
import random

import numpy as np
import torch

from oracle import stylegan2 as O


def test_state_dict_has_reference_names_and_shapes():
    from diagan.models import stylegan2 as M
    for size in (8, 16, 64):
        g, d = M.StyleGANGenerator(size=size), M.StyleGANDiscriminator(size=size)
        strip = lambda sd: {k: tuple(v.shape) for k, v in sd.items() if not k.endswith("kernel")}
        assert strip(g.state_dict()) == O.generator_shapes(size)
        assert strip(d.state_dict()) == O.discriminator_shapes(size)
        kernels = [k for k in g.state_dict() if k.endswith("kernel")]
        assert len(kernels) == 2 * (g.log_size - 2)          # one Blur per up-conv, one Upsample per ToRGB
        assert g.n_latent == 2 * g.log_size - 2 and g.num_layers == 2 * (g.log_size - 2) + 1
    assert M.Generator is M.StyleGANGenerator and M.Discriminator is M.StyleGANDiscriminator


def test_fir_kernels_and_pads_match_reference_formulas():
    from diagan.models import stylegan2 as M
    k = M.make_kernel([1, 3, 3, 1])
    assert abs(float(k.sum()) - 1) < 1e-6 and k.shape == (4, 4)
    assert M.Upsample([1, 3, 3, 1]).pad == (2, 1) and float(M.Upsample([1, 3, 3, 1]).kernel.sum()) == 4.0
    assert M.Downsample([1, 3, 3, 1]).pad == (1, 1)
    conv = M.ModulatedConv2d(8, 8, 3, 16, upsample=True)
    assert conv.blur.pad == (1, 1) and abs(float(conv.blur.kernel.sum()) - 4) < 1e-6
    assert M.ConvLayer(8, 8, 3, downsample=True)[0].pad == (2, 2)
    assert M.ConvLayer(8, 8, 1, downsample=True, activate=False, bias=False)[0].pad == (1, 1)


def test_trainer_helpers():
    from diagan.trainer import stylegan2 as TR
    ds = list(range(10))
    assert isinstance(TR.data_sampler(ds, True, False), torch.utils.data.RandomSampler)
    assert isinstance(TR.data_sampler(ds, False, False), torch.utils.data.SequentialSampler)
    w = TR.data_sampler(ds, True, False, weights=np.linspace(0, 1, 10))
    assert isinstance(w, torch.utils.data.WeightedRandomSampler) and w.num_samples == 10 and w.replacement
    # under data parallelism the weights stay in force (the reference drops them): same order on every rank, strided
    from diagan.datasets.sampler import ShardedSampler
    sh = TR.data_sampler(ds, True, True, weights=np.array([0.0] * 9 + [1.0]))
    assert isinstance(sh, ShardedSampler) and set(iter(sh)) == {9}
    random.seed(0)
    kinds = {len(TR.mixing_noise(2, 8, 0.9, "cpu")) for _ in range(50)}
    assert kinds == {1, 2}
    assert len(TR.mixing_noise(2, 8, 0.0, "cpu")) == 1 and TR.make_noise(3, 8, 1, "cpu").shape == (3, 8)
    a, b = torch.nn.Linear(3, 2), torch.nn.Linear(3, 2)
    before = a.weight.detach().clone()
    TR.accumulate(a, b, 0.75)
    assert torch.allclose(a.weight, 0.75 * before + 0.25 * b.weight)
    # losses: plain functions of logits
    r, f = torch.tensor([[1.0], [-2.0]]), torch.tensor([[0.5], [3.0]])
    assert abs(TR.d_logistic_loss(r, f).item() - O.d_logistic_loss(r, f).item()) < 1e-7
    assert abs(TR.g_nonsaturating_loss(f).item() - O.g_nonsaturating_loss(f).item()) < 1e-7


def test_factory_builds_stylegan_for_ffhq():
    from diagan.models import stylegan2 as M
    from diagan.models.predefined_models import get_gan_model
    from diagan.optim import FusedAdam
    netG, netD, netD_drs, optG, optD, optD_drs = get_gan_model('ffhq', model='stylegan', drs=True)
    assert isinstance(netG, M.StyleGANGenerator) and netG.size == 256
    assert isinstance(netD, M.StyleGANDiscriminator) and isinstance(netD_drs, M.StyleGANDiscriminator)
    assert netD is not netD_drs and all(isinstance(o, FusedAdam) for o in (optG, optD, optD_drs))
    assert optG.param_groups[0]['lr'] == 2e-4 and tuple(optG.param_groups[0]['betas']) == (0.0, 0.9)


def test_command_line_defaults_follow_the_reference():
    from diagan.stylegan2_cli import SIZES, build_parser
    p1, p2 = build_parser(1).parse_args([]), build_parser(2).parse_args([])
    assert (p1.r1, p2.r1) == (0.1, 10) and (p1.save_logit_after, p1.stop_save_logit_after) == (195000, 200000)
    assert p2.save_logit_after == 1000000 and p2.p1_step == 200000 and not hasattr(p2, 'stop_save_logit_after')
    for a in (p1, p2):
        assert (a.iter, a.batch, a.n_sample, a.size, a.path_regularize, a.path_batch_shrink) == (800000, 16, 64, 32, 2, 2)
        assert (a.d_reg_every, a.g_reg_every, a.mixing, a.lr, a.channel_multiplier) == (16, 4, 0.9, 0.002, 2)
        assert (a.work_dir, a.exp_name, a.seed, a.logit_save_steps, a.dataset) == ("./exp_results", "test", 1, 100, "cifar10")
    assert SIZES == {'cifar10': 32, 'celeba': 64, 'utk_faces': 64, 'imagenet': 128, 'ffhq': 256}
