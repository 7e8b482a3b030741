"""Model factory (API of diagan-pkg/diagan/models/predefined_models.py:17-183).

`get_gan_model(dataset_name, model, loss_type, gold, drs, **kw)` returns `(netG, netD, optG, optD)`, or with
`drs=True` `(netG, netD, netD_drs, optG, optD, optD_drs)` where D_drs is a second discriminator of the same
architecture trained with the 'ns' loss.  Architectures and Adam settings per dataset are the reference's; they are
declared as data below, and the per-dataset helpers of the reference (`get_cifar10_gen`, ...) are generated from the
table.  Optimisers are `FusedAdam` (one launch per step over the network's flat parameter slab).  'ffhq' builds the StyleGAN2 pair at 256^2
(reference :153-163; extra keywords -- loss_type, gold, ... -- are swallowed by the classes' **kwargs as there).  Model
families outside the accelerated path (infomax_gan, ssgan, toy, inclusive) raise NotImplementedError."""
from diagan.optim import FusedAdam


def _sngan(name):
    def load():
        from diagan.models import sngan
        return getattr(sngan, name)
    return load


def _named(module, name):
    def load():
        import importlib
        return getattr(importlib.import_module(module), name)
    return load


# dataset -> architecture family, (lr, betas), generator / discriminator classes (plain, top-k or GOLD variant) and
# constructor keywords fixed by the dataset
RECIPES = {
    'cifar10': dict(family='sngan', adam=(2e-4, (0.0, 0.9)),
                    gen=_sngan('SNGANGenerator32'), gen_topk=_named('diagan.models.topk_models', 'TopkSNGANGenerator32'),
                    disc=_sngan('SNGANDiscriminator32'),
                    disc_gold=_named('diagan.models.gold_reweight_models', 'GoldSNGANDiscriminator32')),
    'celeba': dict(family='sngan', adam=(2e-4, (0.0, 0.9)),
                   gen=_sngan('SNGANGenerator64'), gen_topk=_named('diagan.models.topk_models', 'TopkSNGANGenerator64'),
                   disc=_sngan('SNGANDiscriminator64'),
                   disc_gold=_named('diagan.models.gold_reweight_models', 'GoldSNGANDiscriminator64')),
    'color_mnist': dict(family='mnist_dcgan', adam=(1e-4, (0.5, 0.9)), fixed={},
                        gen=_named('diagan.models.mnist', 'MNIST_DCGAN_Generator'),
                        disc=_named('diagan.models.mnist', 'MNIST_DCGAN_Discriminator')),
    'mnist_fmnist': dict(family='mnist_dcgan', adam=(1e-4, (0.5, 0.9)), fixed=dict(nc=1),
                         gen=_named('diagan.models.mnist', 'MNIST_DCGAN_Generator'),
                         disc=_named('diagan.models.mnist', 'MNIST_DCGAN_Discriminator')),
    'ffhq': dict(family='stylegan', adam=(2e-4, (0.0, 0.9)), fixed=dict(size=256),
                 gen=_named('diagan.models.stylegan2', 'StyleGANGenerator'),
                 disc=_named('diagan.models.stylegan2', 'StyleGANDiscriminator')),
}
NOT_ACCELERATED = ('25gaussian',)            # toy MLPs


def _optimizer(net, recipe):
    lr, betas = recipe['adam']
    return FusedAdam(net, lr, betas=betas)


def _recipe(dataset_name, model):
    if dataset_name in NOT_ACCELERATED:
        raise NotImplementedError(f"dataset '{dataset_name}' uses a model family outside the accelerated hot path")
    recipe = RECIPES[dataset_name]
    if recipe['family'] == 'sngan' and model != 'sngan':
        raise NotImplementedError(f"model '{model}' is outside the accelerated hot path (SURVEY §8: sngan and "
                                  "mnist_dcgan are in scope)")
    return recipe


def build_generator(dataset_name, model='sngan', loss_type='hinge', gold=False, topk=False, num_pack=1,
                    reweight=False, **kwargs):
    recipe = _recipe(dataset_name, model)
    if recipe['family'] == 'stylegan':
        netG = recipe['gen']()(**recipe['fixed'], **kwargs)
    elif recipe['family'] == 'sngan':
        netG = recipe['gen_topk']()(loss_type=loss_type, topk=topk, **kwargs) if topk else \
            recipe['gen']()(loss_type=loss_type, **kwargs)
    else:
        if kwargs.get('inclusive'):
            raise NotImplementedError("InclusiveMNISTDCGANGenerator is a baseline outside the hot path (SURVEY §2)")
        netG = recipe['gen']()(loss_type=loss_type, topk=topk, **recipe['fixed'], **kwargs)
    return netG, _optimizer(netG, recipe)


def build_discriminator(dataset_name, model='sngan', loss_type='hinge', gold=False, topk=False, num_pack=1, **kwargs):
    recipe = _recipe(dataset_name, model)
    if recipe['family'] == 'stylegan':
        netD = recipe['disc']()(**recipe['fixed'], **kwargs)
    elif recipe['family'] == 'sngan':
        netD = recipe['disc_gold' if gold else 'disc']()(loss_type=loss_type, **kwargs)
    else:
        netD = recipe['disc']()(use_gold=gold, loss_type=loss_type, num_pack=num_pack, **recipe['fixed'], **kwargs)
    return netD, _optimizer(netD, recipe)


def _bind(builder, dataset_name):
    def helper(**kwargs):
        return builder(dataset_name, **kwargs)
    helper.__doc__ = f"{builder.__name__}('{dataset_name}', ...): (network, optimiser)"
    return helper


# the reference's per-dataset helper names and its DATASET_DICT, generated from the table
DATASET_DICT = {}
for _name in list(RECIPES) + list(NOT_ACCELERATED):
    DATASET_DICT[_name] = (_bind(build_generator, _name), _bind(build_discriminator, _name))
    globals()[f'get_{_name}_gen'], globals()[f'get_{_name}_disc'] = DATASET_DICT[_name]


def get_gan_model(dataset_name, model='sngan', loss_type="hinge", gold=False, drs=False, **kwargs):
    make_g, make_d = DATASET_DICT[dataset_name]
    netG, optG = make_g(model=model, loss_type=loss_type, gold=gold, **kwargs)
    netD, optD = make_d(model=model, loss_type=loss_type, gold=gold, **kwargs)
    if not drs:
        return netG, netD, optG, optD
    netD_drs, optD_drs = make_d(model=model, loss_type='ns', **kwargs)      # the DRS critic: always 'ns', never GOLD
    return netG, netD, netD_drs, optG, optD, optD_drs
