"""CPU: oracle/stylegan_ops.py against vectors produced by the reference's own CPU statements."""
import os

import numpy as np
import torch

from oracle import stylegan_ops as S


def test_upfirdn2d_cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "stylegan_ops.npz"))
    for name in [str(n) for n in g["names"]]:
        x = torch.from_numpy(g[f"{name}_x"]).requires_grad_(True)
        k = torch.from_numpy(g[f"{name}_k"])
        u, d, px0, px1, py0, py1 = [int(v) for v in g[f"{name}_cfg"]]
        y = S.upfirdn2d(x, k, u, u, d, d, px0, px1, py0, py1)
        assert y.shape == g[f"{name}_y"].shape, name
        np.testing.assert_allclose(y.detach().numpy(), g[f"{name}_y"], atol=1e-5, err_msg=name)
        (y * torch.from_numpy(g[f"{name}_cot"])).sum().backward()
        np.testing.assert_allclose(x.grad.numpy(), g[f"{name}_gx"], atol=1e-5, err_msg=name)


def test_fused_leaky_relu(golden_dir):
    g = np.load(os.path.join(golden_dir, "stylegan_ops.npz"))
    for tag in ("4d", "2d"):
        x = torch.from_numpy(g[f"flr_{tag}_x"]).requires_grad_(True)
        b = torch.from_numpy(g[f"flr_{tag}_b"]).requires_grad_(True)
        y = S.fused_leaky_relu(x, b, 0.2, 2 ** 0.5)
        np.testing.assert_allclose(y.detach().numpy(), g[f"flr_{tag}_y"], atol=1e-6)
        (y * torch.from_numpy(g[f"flr_{tag}_cot"])).sum().backward()
        np.testing.assert_allclose(x.grad.numpy(), g[f"flr_{tag}_gx"], atol=1e-6)
        np.testing.assert_allclose(b.grad.numpy(), g[f"flr_{tag}_gb"], atol=1e-5)
    y = S.fused_leaky_relu(torch.from_numpy(g["flr_nobias_x"]), None, 0.2, 2 ** 0.5)
    np.testing.assert_allclose(y.numpy(), g["flr_nobias_y"], atol=1e-6)
