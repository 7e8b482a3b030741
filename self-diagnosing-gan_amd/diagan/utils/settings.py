"""Seed contract of the reference (diagan-pkg/diagan/utils/settings.py:8-18)."""
import os
import random

import numpy as np
import torch


def set_seed(seed=3):
    if seed is not None:
        print(f'=======> Using Fixed Random Seed: {seed} <========')
        random.seed(seed)
        os.environ['PYTHONHASHSEED'] = str(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed(seed)
            torch.cuda.manual_seed_all(seed)
