"""StyleGAN2 training for the self-diagnosing pipeline.

Phase 1 = stylegan2/train_ffhq.py: non-saturating logistic GAN with lazy R1 (every `d_reg_every` steps) and lazy
path-length regularisation (every `g_reg_every`), EMA generator, and the per-index discriminator logit record
(`get_logit`, train_ffhq.py:128-146) that phase 2 scores.  Phase 2 = stylegan2/train_ffhq_phase2.py: the same loop
fed by a score-weighted sampler, plus a second discriminator `D_drs` trained on uniformly sampled data
(train_ffhq_phase2.py:226-270).  Function names and argument meaning follow the reference so that its training
scripts read the same; `StyleGAN2Trainer` holds the loop both scripts share.

Data parallelism: one process per GPU.  The reference wraps the nets in DistributedDataParallel (bucketed all-reduce
hooked into backward); here every network owns ONE flat gradient slab (FlatNet), so a step's gradients are averaged
with a single RCCL all-reduce of that slab after backward -- for these sizes (G 30 M / D 29 M parameters at 256^2:
~120 MB) one large collective is what xGMI's point-to-point links want.  The logit record is gathered with
`concat_all_gather` exactly as the reference does."""
import math
import pickle
import random
from collections import defaultdict
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F
from torch import autograd
from torch.utils import data

from diagan.datasets.sampler import ShardedSampler
from diagan.optim import FusedAdam
from diagan.trainer import distributed as dist
from diagan.trainer.distributed import get_rank, get_world_size, reduce_loss_dict, reduce_sum
from diagan.utils.settings import quiesce_gc


def data_sampler(dataset, shuffle, distributed, weights=None):
    """train_ffhq.py:34-45 / train_ffhq_phase2.py:35-46.  One deliberate difference: the reference tests
    `distributed` first, so a multi-GPU phase 2 silently trains on UNweighted data; here the weights
    stay in force -- every rank draws the same weighted order (shared CPU seed) and keeps every W-th index."""
    if weights is not None:
        weighted = data.WeightedRandomSampler(weights, len(weights), replacement=True)
        return ShardedSampler(weighted, get_rank(), get_world_size()) if distributed else weighted
    if distributed:
        return data.distributed.DistributedSampler(dataset, shuffle=shuffle)
    return data.RandomSampler(dataset) if shuffle else data.SequentialSampler(dataset)


def requires_grad(model, flag=True):
    for p in model.parameters():
        p.requires_grad = flag


def accumulate(model1, model2, decay=0.999):
    """EMA of every parameter (train_ffhq.py:53-58): one fused update over the flat slabs when both nets have them"""
    if hasattr(model1, 'flat_params') and hasattr(model2, 'flat_params') \
            and model1.flat_params.numel() == model2.flat_params.numel():
        model1.flat_params.mul_(decay).add_(model2.flat_params, alpha=1 - decay)
        model1.param_version += 1
        return
    par2 = dict(model2.named_parameters())
    for k, p in model1.named_parameters():
        p.data.mul_(decay).add_(par2[k].data, alpha=1 - decay)


def sample_data(loader):
    while True:
        for batch in loader:
            yield batch


def d_logistic_loss(real_pred, fake_pred):
    return F.softplus(-real_pred).mean() + F.softplus(fake_pred).mean()


def d_r1_loss(real_pred, real_img):
    grad_real, = autograd.grad(outputs=real_pred.sum(), inputs=real_img, create_graph=True)
    return grad_real.pow(2).reshape(grad_real.shape[0], -1).sum(1).mean()


def g_nonsaturating_loss(fake_pred):
    return F.softplus(-fake_pred).mean()


def g_path_regularize(fake_img, latents, mean_path_length, decay=0.01, noise=None):
    """train_ffhq.py:88-102; `noise` (extension) replaces the internal randn_like draw, for tests"""
    if noise is None:
        noise = torch.randn_like(fake_img)
    noise = noise / math.sqrt(fake_img.shape[2] * fake_img.shape[3])
    grad, = autograd.grad(outputs=(fake_img * noise).sum(), inputs=latents, create_graph=True)
    path_lengths = torch.sqrt(grad.pow(2).sum(2).mean(1))
    path_mean = mean_path_length + decay * (path_lengths.mean() - mean_path_length)
    path_penalty = (path_lengths - path_mean).pow(2).mean()
    return path_penalty, path_mean.detach(), path_lengths


def make_noise(batch, latent_dim, n_noise, device):
    if n_noise == 1:
        return torch.randn(batch, latent_dim, device=device)
    return torch.randn(n_noise, batch, latent_dim, device=device).unbind(0)


def mixing_noise(batch, latent_dim, prob, device):
    if prob > 0 and random.random() < prob:
        return make_noise(batch, latent_dim, 2, device)
    return [make_noise(batch, latent_dim, 1, device)]


@torch.no_grad()
def concat_all_gather(tensor):
    """rank-major concatenation of per-rank tensors (train_ffhq.py:150-161); no gradient"""
    return dist.all_gather_cat(tensor)


def get_logit(dataloader, netD, device):
    """D's logit of every training sample, indexed by dataset position (train_ffhq.py:128-146).  The loader yields
    (image, index) pairs; every rank scores its share and the pieces are all-gathered."""
    logit_list = np.zeros(len(dataloader.dataset))
    was_training = netD.training
    netD.eval()
    with torch.no_grad():
        for batch in dataloader:
            images, idx = batch[0].to(device), batch[-1].to(device)
            logit_r = netD(images).view(-1)
            idx_all = concat_all_gather(idx)
            logit_r = concat_all_gather(logit_r)
            logit_list[idx_all.cpu().numpy()] = logit_r.cpu().numpy()
    netD.train(was_training)
    return logit_list


def save_logit(logits_dict, output_path):
    for name, logits in logits_dict.items():
        with open(Path(output_path) / f'logits_{name}.pkl', 'wb') as f:
            pickle.dump(logits, f)


def make_optimizers(generator, discriminator, lr=0.002, g_reg_every=4, d_reg_every=16):
    """Adam with the lazy-regularisation correction of train_ffhq.py:536-549 (lr and betas rescaled by
    every/(every+1)); one fused launch per step over each network's flat slab"""
    g_ratio, d_ratio = g_reg_every / (g_reg_every + 1), d_reg_every / (d_reg_every + 1)
    g_optim = FusedAdam(generator, lr * g_ratio, betas=(0 ** g_ratio, 0.99 ** g_ratio))
    d_optim = FusedAdam(discriminator, lr * d_ratio, betas=(0 ** d_ratio, 0.99 ** d_ratio))
    return g_optim, d_optim


class StyleGAN2Trainer:
    """The loop of train_ffhq.py:163-382 and, with `drs_discriminator` set, of train_ffhq_phase2.py:144-400.

    `args` carries the reference's flags (iter, start_iter, batch, latent, mixing, r1, d_reg_every, g_reg_every,
    path_regularize, path_batch_shrink, logit_save_steps, save_logit_after, stop_save_logit_after, n_sample).
    Adaptive augmentation (non_leaking.py) is not part of this row and `args.augment` must be off."""

    def __init__(self, args, loader, generator, discriminator, g_optim, d_optim, g_ema, device, output_path,
                 drs_loader=None, drs_discriminator=None, drs_d_optim=None, log_every=100, checkpoint_every=5000):
        if getattr(args, 'augment', False):
            raise NotImplementedError("adaptive discriminator augmentation is outside the accelerated path")
        self.args, self.device, self.output_path = args, device, Path(output_path)
        self.loader, self.drs_loader = loader, drs_loader
        self.G, self.D, self.g_ema, self.D_drs = generator, discriminator, g_ema, drs_discriminator
        self.g_optim, self.d_optim, self.drs_d_optim = g_optim, d_optim, drs_d_optim
        self.log_every, self.checkpoint_every = log_every, checkpoint_every
        self.mean_path_length = 0
        self.mean_path_length_avg = 0
        self.logit_results = defaultdict(dict)
        self.accum = 0.5 ** (32 / (10 * 1000))
        self.history = []
        self._iters = {}
        if get_world_size() > 1:
            for net in (generator, discriminator, drs_discriminator, g_ema):
                if net is not None:
                    dist.broadcast_module_(net)

    # ---- pieces of one iteration ---------------------------------------------------------------------------
    def _next(self, which):
        loader = self.loader if which == 'main' else self.drs_loader
        it = self._iters.get(which)
        if it is None:
            it = self._iters[which] = iter(loader)
        try:
            batch = next(it)
        except StopIteration:
            it = self._iters[which] = iter(loader)
            batch = next(it)
        return batch[0].to(self.device, non_blocking=True)

    @staticmethod
    def _step(net, optimizer, loss):
        net.zero_grad()
        loss.backward()
        net.sync_grads(optimizer)
        optimizer.step()                                # DDP's gradient averaging: one collective per step

    def _d_update(self, D, optim, real_img, fake_img, regularize, tag, losses):
        a = self.args
        fake_pred, real_pred = D(fake_img), D(real_img)
        d_loss = d_logistic_loss(real_pred, fake_pred)
        losses[tag] = d_loss
        if tag == 'd':
            losses['real_score'], losses['fake_score'] = real_pred.mean(), fake_pred.mean()
        self._step(D, optim, d_loss)
        if regularize:
            real_img = real_img.detach().requires_grad_(True)
            real_pred = D(real_img)
            r1_loss = d_r1_loss(real_pred, real_img)
            self._step(D, optim, a.r1 / 2 * r1_loss * a.d_reg_every + 0 * real_pred[0])
            if tag == 'd':
                self.r1_loss = r1_loss.detach()

    def _g_update(self, i, losses):
        a = self.args
        noise = mixing_noise(a.batch, a.latent, a.mixing, self.device)
        fake_img, _ = self.G(noise)
        g_loss = g_nonsaturating_loss(self.D(fake_img))
        losses['g'] = g_loss
        self._step(self.G, self.g_optim, g_loss)
        if i % a.g_reg_every == 0:
            path_batch = max(1, a.batch // a.path_batch_shrink)
            noise = mixing_noise(path_batch, a.latent, a.mixing, self.device)
            fake_img, latents = self.G(noise, return_latents=True)
            path_loss, self.mean_path_length, path_lengths = g_path_regularize(fake_img, latents,
                                                                               self.mean_path_length)
            weighted = a.path_regularize * a.g_reg_every * path_loss
            if a.path_batch_shrink:
                weighted = weighted + 0 * fake_img[0, 0, 0, 0]
            self._step(self.G, self.g_optim, weighted)
            self.mean_path_length_avg = reduce_sum(self.mean_path_length).item() / get_world_size()
            self.path_loss, self.path_lengths = path_loss.detach(), path_lengths.detach()

    def train_step(self, i):
        """one iteration i of the reference loop; returns the dict of (device) loss scalars"""
        a, dev = self.args, self.device
        losses = {}
        real_img = self._next('main')
        nets_d = [(self.D, self.d_optim, real_img, 'd')]
        if self.D_drs is not None:
            nets_d.append((self.D_drs, self.drs_d_optim, self._next('drs'), 'drs_d'))
        requires_grad(self.G, False)
        for D, *_ in nets_d:
            requires_grad(D, True)
        with torch.no_grad():
            fake_img, _ = self.G(mixing_noise(a.batch, a.latent, a.mixing, dev))
        for D, optim, real, tag in nets_d:
            self._d_update(D, optim, real, fake_img, i % a.d_reg_every == 0, tag, losses)
        requires_grad(self.G, True)
        for D, *_ in nets_d:
            requires_grad(D, False)
        self._g_update(i, losses)
        accumulate(self.g_ema, self.G, self.accum)
        losses['r1'] = self.r1_loss
        losses['path'], losses['path_length'] = self.path_loss, self.path_lengths.mean()
        return losses

    # ---- the loop --------------------------------------------------------------------------------------------
    def train(self):
        a = self.args
        zero = torch.tensor(0.0, device=self.device)
        self.r1_loss, self.path_loss, self.path_lengths = zero, zero, zero
        sample_z = torch.randn(a.n_sample, a.latent, device=self.device)
        for idx in range(a.iter):                     # the reference's `pbar = range(args.iter)`: i stops at iter - 1
            i = idx + a.start_iter                    # for a fresh run and at iter for a resumed one
            if i > a.iter:
                print("Done!")
                break
            losses = self.train_step(i)
            if idx == 0:
                quiesce_gc()                          # (the first iteration built every autograd site and launch table)
            if self.D_drs is None and i % a.logit_save_steps == 0 and a.save_logit_after <= i <= a.stop_save_logit_after:
                print(f'save logit step: {i}')
                logit_list = get_logit(dataloader=self.loader, netD=self.D, device=self.device)
                if get_rank() == 0:
                    self.logit_results['netD'][i] = logit_list
                    self.output_path.mkdir(parents=True, exist_ok=True)
                    save_logit(self.logit_results, self.output_path)
            if i % self.log_every == 0 or i == a.iter:
                reduced = reduce_loss_dict(losses)
                if get_rank() == 0:
                    vals = {k: float(v.detach().mean()) for k, v in reduced.items()}
                    vals['step'] = i
                    self.history.append(vals)
                    print("; ".join(f"{k}: {vals[k]:.4f}" for k in ('d', 'drs_d', 'g', 'r1', 'path') if k in vals)
                          + f"; mean path: {self.mean_path_length_avg:.4f}")
            if get_rank() == 0 and i > 0 and i % self.checkpoint_every == 0:
                self.save_checkpoint(i)
        return sample_z

    def save_checkpoint(self, i):
        """same keys as the reference's checkpoint (train_ffhq.py:367-380 / phase2 :389-401)"""
        path = self.output_path / 'checkpoint'
        path.mkdir(parents=True, exist_ok=True)
        ckpt = {"g": self.G.state_dict(), "d": self.D.state_dict(), "g_ema": self.g_ema.state_dict(),
                "g_optim": self.g_optim.state_dict(), "d_optim": self.d_optim.state_dict(), "args": self.args,
                "ada_aug_p": 0.0}
        if self.D_drs is not None:
            ckpt["drs_d"], ckpt["drs_d_optim"] = self.D_drs.state_dict(), self.drs_d_optim.state_dict()
        torch.save(ckpt, path / f"{str(i).zfill(6)}.pt")
        return path / f"{str(i).zfill(6)}.pt"
