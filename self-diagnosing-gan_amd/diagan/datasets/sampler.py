"""Phase-2 resampling (reference: train_mimicry_phase2.py:21-34).

The sampler stays on the host on purpose: `WeightedRandomSampler` draws
`torch.multinomial(weights.double(), N, replacement=True)` from the global CPU generator, so using
the very same call with bit-identical float64 weights (the exact-f64 HIP scorer) gives bit-exact
sample-index assignments.
"""
from torch.utils import data


def floor_weights(weights, eps=1e-6):
    """weight_list = [eps if i < eps else i for i in weights]  (train_mimicry_phase2.py:23)"""
    return [eps if i < eps else i for i in weights]


def make_weighted_sampler(weights, eps=1e-6):
    weight_list = floor_weights(weights, eps)
    return data.WeightedRandomSampler(weight_list, len(weight_list), replacement=True)


class ShardedSampler(data.Sampler):
    """Data-parallel view of any index sampler: rank r takes positions r, r+W, r+2W, ... of the
    order drawn by the wrapped sampler.  Every rank draws the SAME order (same CPU seed), which
    fixes the reference's DDP bug where phase-2 weights are silently dropped
    (stylegan2/train_ffhq_phase2.py:35-40 returns DistributedSampler before looking at weights)."""

    def __init__(self, base_sampler, rank, world_size):
        self.base, self.rank, self.world = base_sampler, rank, world_size

    def __iter__(self):
        order = list(iter(self.base))
        n = (len(order) // self.world) * self.world          # drop the ragged tail evenly
        return iter(order[self.rank:n:self.world])

    def __len__(self):
        return len(self.base) // self.world
