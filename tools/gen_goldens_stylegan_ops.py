#!/usr/bin/env python
"""Golden vectors for the StyleGAN2 native ops from the reference's own CPU statements:
`upfirdn2d_native` (diagan-pkg/diagan/models/op/upfirdn2d.py:159-200) and the CPU branch of
`fused_leaky_relu` (op/fused_act.py:106-118).  torch.utils.cpp_extension.load is stubbed BEFORE the
import so that nothing is JIT-compiled or written next to the reference sources.
Run through tools/gen_goldens.py (build container only)."""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/diagan-pkg"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
sys.dont_write_bytecode = True


def main():
    import torch.utils.cpp_extension as cpp
    real_load = cpp.load
    cpp.load = lambda *a, **k: types.SimpleNamespace()          # no JIT, no files
    try:
        if REF not in sys.path:
            sys.path.insert(0, REF)
        import importlib.util
        def load_file(name, path):
            spec = importlib.util.spec_from_file_location(name, path)
            m = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(m)
            return m
        up = load_file("_ref_upfirdn2d", os.path.join(REF, "diagan/models/op/upfirdn2d.py"))
        fa = load_file("_ref_fused_act", os.path.join(REF, "diagan/models/op/fused_act.py"))
    finally:
        cpp.load = real_load

    out = {}
    g = torch.Generator().manual_seed(17)
    k1 = torch.tensor([1.0, 3.0, 3.0, 1.0])
    blur = k1[None, :] * k1[:, None]
    blur = blur / blur.sum()
    asym = torch.randn(3, 5, generator=g)
    cases = [
        # name, shape, kernel, up, down, (pad_x0, pad_x1, pad_y0, pad_y1)
        ("blur_p21", (2, 3, 8, 8), blur, 1, 1, (2, 1, 2, 1)),
        ("upsample2", (2, 3, 8, 8), blur * 4, 2, 1, (2, 1, 2, 1)),
        ("downsample2", (2, 3, 8, 8), blur, 1, 2, (1, 1, 1, 1)),
        ("asym_k_up3_down2", (1, 2, 7, 5), asym, 3, 2, (2, 3, 1, 0)),
        ("negative_pad", (1, 2, 9, 9), blur, 1, 1, (-1, 2, 0, -2)),
        ("big", (1, 6, 32, 32), blur * 4, 2, 1, (2, 1, 2, 1)),
    ]
    names = []
    for name, shape, kern, u, d, pad in cases:
        x = torch.randn(*shape, generator=g).requires_grad_(True)
        y = up.upfirdn2d_native(x, kern, u, u, d, d, *pad)
        cot = torch.randn(y.shape, generator=g)
        (y * cot).sum().backward()
        names.append(name)
        out[f"{name}_x"], out[f"{name}_k"] = x.detach().numpy(), kern.numpy()
        out[f"{name}_cfg"] = np.array([u, d, *pad])
        out[f"{name}_y"], out[f"{name}_cot"], out[f"{name}_gx"] = y.detach().numpy(), cot.numpy(), x.grad.numpy()
    out["names"] = np.array(names)

    # fused_leaky_relu, CPU branch of the reference (slope fixed at 0.2 there)
    for tag, shape in (("4d", (2, 8, 5, 5)), ("2d", (4, 8))):
        x = torch.randn(*shape, generator=g).requires_grad_(True)
        b = torch.randn(8, generator=g).requires_grad_(True)
        y = fa.fused_leaky_relu(x, b, 0.2, 2 ** 0.5)
        cot = torch.randn(y.shape, generator=g)
        (y * cot).sum().backward()
        out[f"flr_{tag}_x"], out[f"flr_{tag}_b"], out[f"flr_{tag}_y"] = x.detach().numpy(), b.detach().numpy(), y.detach().numpy()
        out[f"flr_{tag}_cot"], out[f"flr_{tag}_gx"], out[f"flr_{tag}_gb"] = cot.numpy(), x.grad.numpy(), b.grad.numpy()
    x = torch.randn(3, 4, 6, 6, generator=g)
    out["flr_nobias_x"], out["flr_nobias_y"] = x.numpy(), fa.fused_leaky_relu(x, None, 0.2, 2 ** 0.5).numpy()
    np.savez_compressed(os.path.join(OUT, "stylegan_ops.npz"), **out)


if __name__ == "__main__":
    main()
