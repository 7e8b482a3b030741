"""Seeding (API of diagan-pkg/diagan/utils/settings.py:8-18: `set_seed(seed=3)`, `None` = leave every generator alone).

One seed feeds, in this order, Python's `random`, the hash seed exported to child processes, NumPy's legacy global
generator and torch's CPU and device generators -- the same generators the reference seeds, so host-side draws
(sampler order, DRS acceptance, NumPy noise) repeat for a given seed.  There is no cuDNN to configure here: the HIP
kernels of this engine are deterministic by construction (fixed-order reductions, no float atomics)."""
import os
import random

import numpy
import torch

_BANNER = '=======> Using Fixed Random Seed: {} <========'      # printed by the reference; log scrapers rely on it


def _device_generators(seed):
    if not torch.cuda.is_available():
        return
    torch.cuda.manual_seed(seed)            # current device, then all of them, as the reference calls both
    torch.cuda.manual_seed_all(seed)


def set_seed(seed=3):
    if seed is None:
        return
    print(_BANNER.format(seed))
    seeders = (random.seed,
               lambda s: os.environ.__setitem__('PYTHONHASHSEED', str(s)),
               numpy.random.seed,
               torch.manual_seed,
               _device_generators)
    for apply in seeders:
        apply(seed)


def quiesce_gc():
    """After the networks, optimisers and their launch tables exist (a quarter of a million tracked Python objects for an SNGAN pair):
    collect once and move everything alive to the permanent generation.  Python's cyclic collector otherwise walks all of them on
    every full collection -- ~50 ms, in the middle of a training step whenever the allocation counters say so (measured: 12.1 -> 14.1
    ms per SNGAN-32 step in a 30-step window that contains one).  The engine's own per-step objects are acyclic; later collections
    only see what was made after this call."""
    import gc
    gc.collect()
    gc.freeze()

