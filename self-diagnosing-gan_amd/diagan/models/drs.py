"""Discriminator Rejection Sampling at evaluation time (reference: diagan-pkg/diagan/models/drs.py:9-68).

Wraps a generator and the D_drs discriminator trained in phase 2: images come from the HIP generator, the
log-density-ratio from the HIP discriminator; the acceptance arithmetic (running maximum, percentile
gamma, sigmoid, np.random.rand draw) is the reference's host-side NumPy, so that for the same logits and
NumPy RNG state the same samples are accepted."""
import numpy as np
import torch
import torch.nn as nn


def sigmoid(x):
    return 1 / (1 + np.exp(-x))


class DRS(nn.Module):
    def __init__(self, netG, netD, device, gamma=None, percentile=80):
        super().__init__()
        self.netG, self.netD = netG, netD
        self.maximum = -100000
        self.device = device
        self.batch_size = 256
        self.percentile = percentile
        self.gamma = gamma
        self.init_drs()

    def get_fake_samples_and_ldr(self, num_data):
        with torch.no_grad():
            imgs = self.netG.generate_images(num_data, device=self.device)
            netD_out = self.netD(imgs)
            if type(netD_out) is tuple:
                netD_out = netD_out[0]
            ldr = netD_out.detach().cpu().numpy()
        return imgs, ldr

    def init_drs(self):
        for _ in range(50):                       # burn-in estimate of the maximum logit (drs.py:29-34)
            _, ldr = self.get_fake_samples_and_ldr(self.batch_size)
            self.maximum = max(self.maximum, ldr.max())

    def acceptance(self, ldr, eps=1e-6):
        """Boolean accept mask for one batch of logits (drs.py:36-55)."""
        self.maximum = max(self.maximum, ldr.max())
        ldr_max = ldr - self.maximum
        F = ldr_max - np.log(1 - np.exp(ldr_max - eps))
        gamma = np.percentile(F, self.percentile) if self.gamma is None else self.gamma
        sigF = sigmoid(F - gamma)
        psi = np.random.rand(len(sigF))
        return np.array([bool(sigF[i] > psi[i]) for i in range(len(sigF))])

    def sub_rejection_sampler(self, fake_samples, ldr, eps=1e-6):
        keep = self.acceptance(ldr, eps)
        idx = torch.from_numpy(np.nonzero(keep)[0]).to(fake_samples.device)
        return fake_samples.detach().index_select(0, idx).cpu()

    def generate_images(self, num_images, device=None):
        out, n = [], 0
        while n < num_images:
            fake_samples, ldrs = self.get_fake_samples_and_ldr(self.batch_size)
            acc = self.sub_rejection_sampler(fake_samples, ldrs)
            out.append(acc)
            n += acc.size(0)
        return torch.cat(out, dim=0)[:num_images]
