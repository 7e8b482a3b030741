"""Model factory of the reference (diagan-pkg/diagan/models/predefined_models.py:17-183).

Same entry point and tuple shapes: get_gan_model(dataset_name, model, loss_type, gold, drs, **kw)
-> (netG, netD, optG, optD) or (netG, netD, netD_drs, optG, optD, optD_drs); same Adam
hyper-parameters per dataset.  Model families outside the hot path (infomax_gan, ssgan, toy,
stylegan, inclusive) raise NotImplementedError naming SURVEY §8."""
from diagan.models import sngan
from diagan.models.gold_reweight_models import GoldSNGANDiscriminator32, GoldSNGANDiscriminator64
from diagan.models.topk_models import TopkSNGANGenerator32, TopkSNGANGenerator64
from diagan.optim import FusedAdam


def _only_sngan(model):
    if model != 'sngan':
        raise NotImplementedError(f"model '{model}' is outside the accelerated hot path (SURVEY §8: sngan and "
                                  "mnist_dcgan are in scope)")


def get_cifar10_gen(model='sngan', loss_type='hinge', gold=False, topk=False, **kwargs):
    _only_sngan(model)
    netG = TopkSNGANGenerator32(loss_type=loss_type, topk=topk, **kwargs) if topk else \
        sngan.SNGANGenerator32(loss_type=loss_type, **kwargs)
    return netG, FusedAdam(netG, 2e-4, betas=(0.0, 0.9))


def get_cifar10_disc(model='sngan', loss_type='hinge', gold=False, topk=False, **kwargs):
    _only_sngan(model)
    netD = GoldSNGANDiscriminator32(loss_type=loss_type, **kwargs) if gold else \
        sngan.SNGANDiscriminator32(loss_type=loss_type, **kwargs)
    return netD, FusedAdam(netD, 2e-4, betas=(0.0, 0.9))


def get_celeba_gen(model='sngan', loss_type='hinge', gold=False, topk=False, **kwargs):
    _only_sngan(model)
    netG = TopkSNGANGenerator64(loss_type=loss_type, topk=topk, **kwargs) if topk else \
        sngan.SNGANGenerator64(loss_type=loss_type, **kwargs)
    return netG, FusedAdam(netG, 2e-4, betas=(0.0, 0.9))


def get_celeba_disc(model='sngan', loss_type='hinge', gold=False, topk=False, **kwargs):
    _only_sngan(model)
    netD = GoldSNGANDiscriminator64(loss_type=loss_type, **kwargs) if gold else \
        sngan.SNGANDiscriminator64(loss_type=loss_type, **kwargs)
    return netD, FusedAdam(netD, 2e-4, betas=(0.0, 0.9))


def get_color_mnist_gen(model='mnist_dcgan', reweight=False, loss_type='ns', gold=False, num_pack=1, topk=False,
                        **kwargs):
    from diagan.models.mnist import MNIST_DCGAN_Generator
    if kwargs.get('inclusive'):
        raise NotImplementedError("InclusiveMNISTDCGANGenerator is a baseline outside the hot path (SURVEY §2)")
    netG = MNIST_DCGAN_Generator(loss_type=loss_type, topk=topk, **kwargs)
    return netG, FusedAdam(netG, 1e-4, betas=(0.5, 0.9))


def get_color_mnist_disc(model='mnist_dcgan', loss_type='hinge', gold=False, num_pack=1, topk=False, **kwargs):
    from diagan.models.mnist import MNIST_DCGAN_Discriminator
    netD = MNIST_DCGAN_Discriminator(use_gold=gold, loss_type=loss_type, num_pack=num_pack, **kwargs)
    return netD, FusedAdam(netD, 1e-4, betas=(0.5, 0.9))


def get_mnist_fmnist_gen(model='mnist_dcgan', loss_type='hinge', gold=False, num_pack=1, topk=False, **kwargs):
    from diagan.models.mnist import MNIST_DCGAN_Generator
    netG = MNIST_DCGAN_Generator(nc=1, loss_type=loss_type, topk=topk, **kwargs)
    return netG, FusedAdam(netG, 1e-4, betas=(0.5, 0.9))


def get_mnist_fmnist_disc(model='mnist_dcgan', loss_type='hinge', gold=False, num_pack=1, topk=False, **kwargs):
    from diagan.models.mnist import MNIST_DCGAN_Discriminator
    netD = MNIST_DCGAN_Discriminator(nc=1, use_gold=gold, loss_type=loss_type, num_pack=num_pack, **kwargs)
    return netD, FusedAdam(netD, 1e-4, betas=(0.5, 0.9))


def _out_of_scope(name):
    def fn(**kwargs):
        raise NotImplementedError(f"dataset '{name}' uses a model family outside the accelerated hot path "
                                  "(SURVEY §8(f): StyleGAN2 is the ranked 'next' row)")
    return fn


DATASET_DICT = {
    'celeba': (get_celeba_gen, get_celeba_disc),
    'cifar10': (get_cifar10_gen, get_cifar10_disc),
    'color_mnist': (get_color_mnist_gen, get_color_mnist_disc),
    'mnist_fmnist': (get_mnist_fmnist_gen, get_mnist_fmnist_disc),
    '25gaussian': (_out_of_scope('25gaussian'), _out_of_scope('25gaussian')),
    'ffhq': (_out_of_scope('ffhq'), _out_of_scope('ffhq')),
}


def get_gan_model(dataset_name, model='sngan', loss_type="hinge", gold=False, drs=False, **kwargs):
    netG_fn, netD_fn = DATASET_DICT[dataset_name]
    netG, optG = netG_fn(model=model, loss_type=loss_type, gold=gold, **kwargs)
    netD, optD = netD_fn(model=model, loss_type=loss_type, gold=gold, **kwargs)
    if drs:
        netD_drs, optD_drs = netD_fn(model=model, loss_type='ns', **kwargs)
        return netG, netD, netD_drs, optG, optD, optD_drs
    return netG, netD, optG, optD
