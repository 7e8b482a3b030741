"""Discriminator Rejection Sampling at evaluation time.

Public surface of the reference wrapper (diagan-pkg/diagan/models/drs.py:9-68): `DRS(netG, netD, device, gamma=None,
percentile=80)` behaves like a generator -- `generate_images(n)` -- whose samples have passed the D_drs acceptance
test.  Images come from the HIP generator and logits from the HIP discriminator; the acceptance arithmetic is host
NumPy on 256 values per round, deliberately: the draw from NumPy's global generator is part of the observable
behaviour (same logits + same `np.random` state => same accepted samples, pinned by tests/golden/drs.npz).

Acceptance of a sample with log-density-ratio l, given the running maximum M of all logits seen so far:
    F = (l - M) - log(1 - exp(l - M - eps));   accept iff sigmoid(F - gamma) > u,  u ~ U[0, 1)
where gamma is fixed or the `percentile`-th percentile of F over the current round."""
import numpy as np
import torch
import torch.nn as nn

ROUND = 256          # samples per round of generation (the reference's batch_size)
BURN_IN_ROUNDS = 50  # rounds used to estimate the maximum logit before sampling starts


def acceptance_probability(ldr, running_max, gamma=None, percentile=80, eps=1e-6):
    """NumPy array of acceptance probabilities for one round of logits (drs.py:41-49 of the reference)."""
    shifted = ldr - running_max
    F = shifted - np.log(1 - np.exp(shifted - eps))
    if gamma is None:
        gamma = np.percentile(F, percentile)
    return 1 / (1 + np.exp(-(F - gamma)))


class DRS(nn.Module):
    def __init__(self, netG, netD, device, gamma=None, percentile=80):
        super().__init__()
        self.netG, self.netD, self.device = netG, netD, device
        self.gamma, self.percentile = gamma, percentile
        self.batch_size = ROUND
        self.maximum = -100000
        self.init_drs()

    # -- generator + discriminator round -------------------------------------------------------------------
    def get_fake_samples_and_ldr(self, num_data):
        with torch.no_grad():
            fake = self.netG.generate_images(num_data, device=self.device)
            logit = self.netD(fake)
            logit = logit[0] if type(logit) is tuple else logit        # D may return (logit, features)
        return fake, logit.detach().cpu().numpy()

    def _observe(self, ldr):
        self.maximum = max(self.maximum, ldr.max())

    def init_drs(self):
        for _ in range(BURN_IN_ROUNDS):
            self._observe(self.get_fake_samples_and_ldr(self.batch_size)[1])

    # -- acceptance ------------------------------------------------------------------------------------------
    def acceptance(self, ldr, eps=1e-6):
        """Boolean mask of the accepted samples of one round; consumes len(ldr) draws of np.random."""
        self._observe(ldr)
        prob = acceptance_probability(ldr, self.maximum, self.gamma, self.percentile, eps)
        draw = np.random.rand(len(prob))
        return np.fromiter((bool(p > u) for p, u in zip(prob, draw)), dtype=bool, count=len(prob))

    def sub_rejection_sampler(self, fake_samples, ldr, eps=1e-6):
        kept = torch.from_numpy(np.flatnonzero(self.acceptance(ldr, eps))).to(fake_samples.device)
        return fake_samples.detach().index_select(0, kept).cpu()

    def generate_images(self, num_images, device=None):
        rounds, have = [], 0
        while have < num_images:
            accepted = self.sub_rejection_sampler(*self.get_fake_samples_and_ldr(self.batch_size))
            rounds.append(accepted)
            have += accepted.size(0)
        return torch.cat(rounds, dim=0)[:num_images]
