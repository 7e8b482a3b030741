"""Dataset wrapper API of the reference (diagan-pkg/diagan/datasets/predefined.py:17-36).

`WeightedDataset` is identical in behaviour: items are (data, target, weight, index).
torchvision is not a dependency of the hot path, so `get_predefined_dataset` serves synthetic
tensors of the real datasets' shapes ('cifar10' 32x32 N=50000, 'celeba' 64x64 N=162770,
'color_mnist' 32x32 N=60000) unless a tensor dataset is supplied by the caller.
"""
import numpy as np
import torch
from torch.utils.data import Dataset

DATASET_SHAPES = {
    # name: (N, C, H, W)   sizes: inclusive_gan.py:92-95 (cifar10 / celeba), MNIST train split
    'cifar10': (50000, 3, 32, 32),
    'celeba': (162770, 3, 64, 64),
    'color_mnist': (60000, 3, 32, 32),
    'mnist_fmnist': (60000, 1, 32, 32),
}


class WeightedDataset(Dataset):
    def __init__(self, dataset, weights=None):
        self.dataset = dataset
        self.weights = weights if weights is not None else np.ones(len(dataset))

    def __getitem__(self, index):
        data, target = self.dataset.__getitem__(index)
        return data, target, self.weights[index], index

    def __len__(self):
        return len(self.dataset)

    def fetch_range(self, lo, hi):
        """(data[lo:hi], indices) of a contiguous index range without per-item Python -- the logit pass uses it when the
        wrapped dataset is tensor-backed (it offers `fetch_range` itself); None otherwise (the pass then walks a loader)."""
        f = getattr(self.dataset, 'fetch_range', None)
        data = f(lo, hi) if f is not None else None
        return None if data is None else (data, torch.arange(lo, hi))


class SyntheticImages(Dataset):
    """Deterministic stand-in with the input contract of datasets/transform.py:9-10:
    float32 CHW in [-1, 1] (ToTensor + Normalize(0.5, 0.5))."""

    def __init__(self, num, shape, seed=1234, materialize=True):
        self.num, self.shape, self.seed = num, tuple(shape), seed
        self.data = None
        if materialize:
            g = torch.Generator().manual_seed(seed)
            self.data = torch.rand((num,) + self.shape, generator=g) * 2 - 1

    def __getitem__(self, index):
        if self.data is not None:
            return self.data[index], 0
        g = torch.Generator().manual_seed(self.seed * 1000003 + int(index))
        return torch.rand(self.shape, generator=g) * 2 - 1, 0

    def __len__(self):
        return self.num

    def fetch_range(self, lo, hi):
        return self.data[lo:hi] if self.data is not None else None


def get_predefined_dataset(dataset_name, root=None, weights=None, num_data=None, dataset=None, **kwargs):
    if dataset is None:
        n, c, h, w = DATASET_SHAPES[dataset_name]
        n = num_data if num_data is not None else n
        dataset = SyntheticImages(n, (c, h, w), materialize=n * c * h * w <= (1 << 28))
    return WeightedDataset(dataset=dataset, weights=weights)
