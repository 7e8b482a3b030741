"""TEST INFRASTRUCTURE ONLY -- plain-PyTorch (CPU, fp32, autograd) restatement of the model side
of the hot path.  Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.

What is restated and from where:
  * MNIST_DCGAN_{Generator,Discriminator}, GOLD losses, TopKGenerator: the in-tree reference
    files diagan-pkg/diagan/models/mnist.py:47-80,155-223, gold_reweight_models.py:10-61,
    topk_models.py:15-38.  PINNED: tests/golden/dcgan.npz + losses.npz were produced by running
    those reference classes (torch_mimicry stubbed) and this file reproduces them.
    The base generator STEP (_BaseG.train_step below) has a second witness since round 6: dcgan.npz's
    `gstep_*` entries were written with the reference's own in-tree restatement of the step executing
    (mnist.py:82-152; tools/gen_goldens_models.py), and tests/test_oracle_models.py holds this file's step to them.
  * SNGAN generators / discriminators (32, 64), GBlock / DBlock / DBlockOptimized, SNConv2d /
    SNLinear, base losses, base train steps: these live in torch-mimicry==0.1.16
    (requirements.txt:72), which is NOT in /root/reference and not installable here.  They are
    restated from the published mimicry algorithm as summarised in SURVEY.md §8 a2-a8, a11-a14.
    PARITY UNPINNED versus upstream for these classes: no golden vector of the real package
    exists in this repository; the call sites that fix their interface are
    diagan-pkg/diagan/models/predefined_models.py:17-92 and trainer/trainer.py:250-291.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------
# losses (torch_mimicry.modules.losses; GOLD variants gold_reweight_models.py:10-61)
# ---------------------------------------------------------------------------------------------
def minimax_loss_dis(output_fake, output_real, real_label_val=1.0, fake_label_val=0.0):
    fake_labels = torch.full((output_fake.shape[0], 1), fake_label_val)
    real_labels = torch.full((output_real.shape[0], 1), real_label_val)
    errD_fake = F.binary_cross_entropy_with_logits(output_fake, fake_labels)
    errD_real = F.binary_cross_entropy_with_logits(output_real, real_labels)
    return errD_real + errD_fake


def minimax_loss_gen(output_fake, real_label_val=1.0):
    real_labels = torch.full((output_fake.shape[0], 1), real_label_val)
    return F.binary_cross_entropy_with_logits(output_fake, real_labels)


def ns_loss_gen(output_fake):
    output_fake = torch.sigmoid(output_fake)
    return -torch.mean(torch.log(output_fake + 1e-8))


def hinge_loss_dis(output_fake, output_real):
    return F.relu(1.0 - output_real).mean() + F.relu(1.0 + output_fake).mean()


def hinge_loss_gen(output_fake):
    return -output_fake.mean()


def wasserstein_loss_dis(output_real, output_fake):
    return -1.0 * output_real.mean() + output_fake.mean()


def wasserstein_loss_gen(output_fake):
    return -output_fake.mean()


def compute_gold_reweight(output_fake, d=1):                       # gold_reweight_models.py:10-13
    with torch.no_grad():
        return output_fake ** d


def gold_reweighted_minimax_loss_dis(output_fake, output_real):     # gold_reweight_models.py:21-51
    w = compute_gold_reweight(output_fake)
    errD_fake = F.binary_cross_entropy_with_logits(output_fake, torch.zeros_like(output_fake), reduction='none')
    errD_fake = torch.mean(w.view(-1) * errD_fake.view(-1))
    errD_real = torch.mean(F.binary_cross_entropy_with_logits(output_real, torch.ones_like(output_real),
                                                              reduction='none'))
    return errD_real + errD_fake


def gold_reweighted_hinge_loss_dis(output_fake, output_real):       # gold_reweight_models.py:54-61
    w = compute_gold_reweight(output_fake)
    fake_out = F.relu(1.0 + output_fake)
    return F.relu(1.0 - output_real).mean() + (w.view(-1) * fake_out.view(-1)).mean()


def dis_loss(loss_type, output_real, output_fake, gold=False):
    if gold:
        return {'hinge': gold_reweighted_hinge_loss_dis, 'ns': gold_reweighted_minimax_loss_dis}[loss_type](
            output_fake=output_fake, output_real=output_real)
    if loss_type in ('gan', 'ns'):
        return minimax_loss_dis(output_fake=output_fake, output_real=output_real)
    if loss_type == 'hinge':
        return hinge_loss_dis(output_fake=output_fake, output_real=output_real)
    if loss_type == 'wasserstein':
        return wasserstein_loss_dis(output_fake=output_fake, output_real=output_real)
    raise ValueError(loss_type)


def gen_loss(loss_type, output):
    return {'gan': minimax_loss_gen, 'ns': ns_loss_gen, 'hinge': hinge_loss_gen,
            'wasserstein': wasserstein_loss_gen}[loss_type](output)


def get_topk(x, topk_rate):                                         # topk_models.py:31-38
    k = int(topk_rate * x.size(0))
    return torch.topk(x, k=k, dim=0)[0]


def topk_rate_at(step, epoch_steps, decay_rate=0.99, min_rate=0.5):  # topk_models.py:23-29
    return max(decay_rate ** (step // epoch_steps), min_rate)


# ---------------------------------------------------------------------------------------------
# spectral norm layers (torch_mimicry.modules.spectral_norm / layers)
# ---------------------------------------------------------------------------------------------
class _SpectralNorm:
    def _sn_init(self, n_dim, num_iters=1, eps=1e-12):
        self.num_iters, self.eps = num_iters, eps
        self.register_buffer('sn_u', torch.randn(1, n_dim))
        self.register_buffer('sn_sigma', torch.ones(1))

    def sn_weights(self):
        W = self.weight.view(self.weight.shape[0], -1)
        u = self.sn_u
        with torch.no_grad():
            for _ in range(self.num_iters):
                v = F.normalize(torch.matmul(u, W), eps=self.eps)
                u = F.normalize(torch.matmul(v, W.t()), eps=self.eps)
        sigma = torch.mm(u, torch.mm(W, v.t()))          # gradient flows through W only
        if self.training:
            with torch.no_grad():
                self.sn_sigma[:] = sigma
                self.sn_u[:] = u
        return self.weight / sigma


class SNConv2d(nn.Conv2d, _SpectralNorm):
    def __init__(self, in_channels, out_channels, *args, **kwargs):
        nn.Conv2d.__init__(self, in_channels, out_channels, *args, **kwargs)
        self._sn_init(out_channels)

    def forward(self, x):
        return F.conv2d(x, self.sn_weights(), self.bias, self.stride, self.padding, self.dilation, self.groups)


class SNLinear(nn.Linear, _SpectralNorm):
    def __init__(self, in_features, out_features, *args, **kwargs):
        nn.Linear.__init__(self, in_features, out_features, *args, **kwargs)
        self._sn_init(out_features)

    def forward(self, x):
        return F.linear(x, self.sn_weights(), self.bias)


# ---------------------------------------------------------------------------------------------
# residual blocks (torch_mimicry.modules.resblocks)
# ---------------------------------------------------------------------------------------------
class GBlock(nn.Module):
    def __init__(self, in_channels, out_channels, hidden_channels=None, upsample=False):
        super().__init__()
        hidden_channels = hidden_channels if hidden_channels is not None else out_channels
        self.learnable_sc = in_channels != out_channels or upsample
        self.upsample = upsample
        self.c1 = nn.Conv2d(in_channels, hidden_channels, 3, 1, padding=1)
        self.c2 = nn.Conv2d(hidden_channels, out_channels, 3, 1, padding=1)
        self.b1 = nn.BatchNorm2d(in_channels)
        self.b2 = nn.BatchNorm2d(hidden_channels)
        nn.init.xavier_uniform_(self.c1.weight.data, math.sqrt(2.0))
        nn.init.xavier_uniform_(self.c2.weight.data, math.sqrt(2.0))
        if self.learnable_sc:
            self.c_sc = nn.Conv2d(in_channels, out_channels, 1, 1, padding=0)
            nn.init.xavier_uniform_(self.c_sc.weight.data, 1.0)

    def _up(self, x, conv):
        return conv(F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False))

    def forward(self, x):
        h = F.relu(self.b1(x))
        h = self._up(h, self.c1) if self.upsample else self.c1(h)
        h = F.relu(self.b2(h))
        h = self.c2(h)
        if self.learnable_sc:
            sc = self._up(x, self.c_sc) if self.upsample else self.c_sc(x)
        else:
            sc = x
        return h + sc


class DBlock(nn.Module):
    """nn.ReLU(True) on `h = x` mutates x in place, so the shortcut sees relu(x)."""

    def __init__(self, in_channels, out_channels, hidden_channels=None, downsample=False):
        super().__init__()
        hidden_channels = hidden_channels if hidden_channels is not None else in_channels
        self.downsample = downsample
        self.learnable_sc = (in_channels != out_channels) or downsample
        self.c1 = SNConv2d(in_channels, hidden_channels, 3, 1, 1)
        self.c2 = SNConv2d(hidden_channels, out_channels, 3, 1, 1)
        nn.init.xavier_uniform_(self.c1.weight.data, math.sqrt(2.0))
        nn.init.xavier_uniform_(self.c2.weight.data, math.sqrt(2.0))
        if self.learnable_sc:
            self.c_sc = SNConv2d(in_channels, out_channels, 1, 1, 0)
            nn.init.xavier_uniform_(self.c_sc.weight.data, 1.0)

    def forward(self, x):
        a = F.relu(x)                       # the in-place ReLU aliasing, made explicit
        h = self.c1(a)
        h = self.c2(F.relu(h))
        if self.downsample:
            h = F.avg_pool2d(h, 2)
        if self.learnable_sc:
            sc = self.c_sc(a)
            sc = F.avg_pool2d(sc, 2) if self.downsample else sc
        else:
            sc = a
        return h + sc


class DBlockOptimized(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.c1 = SNConv2d(in_channels, out_channels, 3, 1, 1)
        self.c2 = SNConv2d(out_channels, out_channels, 3, 1, 1)
        self.c_sc = SNConv2d(in_channels, out_channels, 1, 1, 0)
        nn.init.xavier_uniform_(self.c1.weight.data, math.sqrt(2.0))
        nn.init.xavier_uniform_(self.c2.weight.data, math.sqrt(2.0))
        nn.init.xavier_uniform_(self.c_sc.weight.data, 1.0)

    def forward(self, x):
        h = self.c1(x)
        h = self.c2(F.relu(h))
        h = F.avg_pool2d(h, 2)
        return h + self.c_sc(F.avg_pool2d(x, 2))


# ---------------------------------------------------------------------------------------------
# networks (torch_mimicry.nets.sngan) and base train steps (a11-a13)
# ---------------------------------------------------------------------------------------------
class _BaseG(nn.Module):
    def generate_images(self, num_images, noise=None):
        if noise is None:
            noise = torch.randn((num_images, self.nz))
        return self.forward(noise)

    def compute_gan_loss(self, output):
        return gen_loss(self.loss_type, output)

    def train_step(self, real_batch, netD, optG, noise=None, topk_rate=1.0):
        self.zero_grad()
        batch_size = real_batch[0].shape[0]
        fake_images = self.generate_images(batch_size, noise=noise)
        output = netD(fake_images)
        if topk_rate < 1.0 or getattr(self, 'use_topk', False):
            output = get_topk(output, topk_rate)
        errG = self.compute_gan_loss(output)
        errG.backward()
        optG.step()
        return errG.item()


class _BaseD(nn.Module):
    use_gold = False

    def compute_gan_loss(self, output_real, output_fake):
        return dis_loss(self.loss_type, output_real, output_fake, gold=self.use_gold)

    def train_step(self, real_batch, netG, optD, noise=None):
        self.zero_grad()
        real_images = real_batch[0]
        batch_size = real_images.shape[0]
        output_real = self.forward(real_images)
        fake_images = netG.generate_images(batch_size, noise=noise).detach()
        output_fake = self.forward(fake_images)
        errD = self.compute_gan_loss(output_real=output_real, output_fake=output_fake)
        errD.backward()
        optD.step()
        D_x = torch.sigmoid(output_real).mean().item()
        D_Gz = torch.sigmoid(output_fake).mean().item()
        return errD.item(), D_x, D_Gz


class SNGANGenerator32(_BaseG):
    def __init__(self, nz=128, ngf=256, bottom_width=4, loss_type='hinge'):
        super().__init__()
        self.nz, self.ngf, self.bottom_width, self.loss_type = nz, ngf, bottom_width, loss_type
        self.l1 = nn.Linear(nz, (bottom_width ** 2) * ngf)
        self.block2 = GBlock(ngf, ngf, upsample=True)
        self.block3 = GBlock(ngf, ngf, upsample=True)
        self.block4 = GBlock(ngf, ngf, upsample=True)
        self.b5 = nn.BatchNorm2d(ngf)
        self.c5 = nn.Conv2d(ngf, 3, 3, 1, padding=1)
        nn.init.xavier_uniform_(self.l1.weight.data, 1.0)
        nn.init.xavier_uniform_(self.c5.weight.data, 1.0)

    def forward(self, x):
        h = self.l1(x).view(x.shape[0], -1, self.bottom_width, self.bottom_width)
        h = self.block4(self.block3(self.block2(h)))
        return torch.tanh(self.c5(F.relu(self.b5(h))))


class SNGANDiscriminator32(_BaseD):
    def __init__(self, ndf=128, loss_type='hinge'):
        super().__init__()
        self.ndf, self.loss_type = ndf, loss_type
        self.block1 = DBlockOptimized(3, ndf)
        self.block2 = DBlock(ndf, ndf, downsample=True)
        self.block3 = DBlock(ndf, ndf, downsample=False)
        self.block4 = DBlock(ndf, ndf, downsample=False)
        self.l5 = SNLinear(ndf, 1)
        nn.init.xavier_uniform_(self.l5.weight.data, 1.0)

    def forward(self, x):
        h = self.block4(self.block3(self.block2(self.block1(x))))
        h = torch.sum(F.relu(h), dim=(2, 3))
        return self.l5(h)


class SNGANGenerator64(_BaseG):
    def __init__(self, nz=128, ngf=1024, bottom_width=4, loss_type='hinge'):
        super().__init__()
        self.nz, self.ngf, self.bottom_width, self.loss_type = nz, ngf, bottom_width, loss_type
        self.l1 = nn.Linear(nz, (bottom_width ** 2) * ngf)
        self.block2 = GBlock(ngf, ngf >> 1, upsample=True)
        self.block3 = GBlock(ngf >> 1, ngf >> 2, upsample=True)
        self.block4 = GBlock(ngf >> 2, ngf >> 3, upsample=True)
        self.block5 = GBlock(ngf >> 3, ngf >> 4, upsample=True)
        self.b6 = nn.BatchNorm2d(ngf >> 4)
        self.c6 = nn.Conv2d(ngf >> 4, 3, 3, 1, padding=1)
        nn.init.xavier_uniform_(self.l1.weight.data, 1.0)
        nn.init.xavier_uniform_(self.c6.weight.data, 1.0)

    def forward(self, x):
        h = self.l1(x).view(x.shape[0], -1, self.bottom_width, self.bottom_width)
        h = self.block5(self.block4(self.block3(self.block2(h))))
        return torch.tanh(self.c6(F.relu(self.b6(h))))


class SNGANDiscriminator64(_BaseD):
    def __init__(self, ndf=1024, loss_type='hinge'):
        super().__init__()
        self.ndf, self.loss_type = ndf, loss_type
        self.block1 = DBlockOptimized(3, ndf >> 4)
        self.block2 = DBlock(ndf >> 4, ndf >> 3, downsample=True)
        self.block3 = DBlock(ndf >> 3, ndf >> 2, downsample=True)
        self.block4 = DBlock(ndf >> 2, ndf >> 1, downsample=True)
        self.block5 = DBlock(ndf >> 1, ndf, downsample=True)
        self.l6 = SNLinear(ndf, 1)
        nn.init.xavier_uniform_(self.l6.weight.data, 1.0)

    def forward(self, x):
        h = self.block5(self.block4(self.block3(self.block2(self.block1(x)))))
        h = torch.sum(F.relu(h), dim=(2, 3))
        return self.l6(h)


# ---------------------------------------------------------------------------------------------
# MNIST_DCGAN (in-tree reference: diagan-pkg/diagan/models/mnist.py:47-80,155-223)
# ---------------------------------------------------------------------------------------------
class MNIST_DCGAN_Generator(_BaseG):
    """weights_init_3channel(self) is called on the root module, i.e. it is a no-op
    (mnist.py:33-39,74): default PyTorch initialisation is what the reference trains from."""

    def __init__(self, nz=100, nc=3, loss_type='hinge', topk=False):
        super().__init__()
        self.nz, self.loss_type, self.use_topk = nz, loss_type, topk
        self.fc = nn.Linear(nz, 384)
        self.tconv = nn.Sequential(
            nn.ConvTranspose2d(384, 192, 4, 1, 0, bias=False), nn.BatchNorm2d(192), nn.ReLU(True),
            nn.ConvTranspose2d(192, 96, 4, 2, 1, bias=False), nn.BatchNorm2d(96), nn.ReLU(True),
            nn.ConvTranspose2d(96, 48, 4, 2, 1, bias=False), nn.BatchNorm2d(48), nn.ReLU(True),
            nn.ConvTranspose2d(48, nc, 4, 2, 1, bias=False), nn.Tanh())

    def forward(self, x):
        return self.tconv(self.fc(x).view(-1, 384, 1, 1))


class MNIST_DCGAN_Discriminator(_BaseD):
    def __init__(self, nc=3, num_pack=1, loss_type='hinge', use_gold=False):
        super().__init__()
        self.num_pack, self.loss_type, self.use_gold = num_pack, loss_type, use_gold
        layers = [nn.Conv2d(nc * num_pack, 16, 3, 2, 1, bias=False), nn.LeakyReLU(0.2, inplace=True),
                  nn.Dropout(0.5, inplace=False)]
        for cin, cout, stride in ((16, 32, 1), (32, 64, 2), (64, 128, 1), (128, 256, 2), (256, 512, 1)):
            layers += [nn.Conv2d(cin, cout, 3, stride, 1, bias=False), nn.BatchNorm2d(cout),
                       nn.LeakyReLU(0.2, inplace=True), nn.Dropout(0.5, inplace=False)]
        self.conv = nn.Sequential(*layers)
        self.out_d = nn.Linear(4 * 4 * 512, 1)

    def forward(self, x, get_feature=False):
        batch_size = x.size(0)
        packed = torch.cat(torch.split(x, int(batch_size / self.num_pack)), dim=1)
        x = self.conv(packed).view(-1, 4 * 4 * 512)
        return x if get_feature else self.out_d(x)


def make_pair(dataset, loss_type='ns', seed=1):
    """Same construction order as get_gan_model (predefined_models.py:175-183): G first, then D."""
    torch.manual_seed(seed)
    if dataset == 'cifar10':
        netG, netD = SNGANGenerator32(loss_type=loss_type), SNGANDiscriminator32(loss_type=loss_type)
        lr, betas = 2e-4, (0.0, 0.9)
    elif dataset == 'celeba':
        netG, netD = SNGANGenerator64(loss_type=loss_type), SNGANDiscriminator64(loss_type=loss_type)
        lr, betas = 2e-4, (0.0, 0.9)
    elif dataset == 'color_mnist':
        netG, netD = MNIST_DCGAN_Generator(loss_type=loss_type), MNIST_DCGAN_Discriminator(loss_type=loss_type)
        lr, betas = 1e-4, (0.5, 0.9)
    else:
        raise ValueError(dataset)
    optG = torch.optim.Adam(netG.parameters(), lr, betas=betas)
    optD = torch.optim.Adam(netD.parameters(), lr, betas=betas)
    return netG, netD, optG, optD
