"""StyleGAN2 native ops of the reference (diagan-pkg/diagan/models/op/__init__.py) on HIP:
`FusedLeakyReLU`, `fused_leaky_relu`, `upfirdn2d` with the same call signatures and autograd
behaviour (first and second order), backed by csrc/stylegan_ops.hip through the C ABI.
No JIT compilation at import (the reference runs torch.utils.cpp_extension.load here)."""
from .fused_act import FusedLeakyReLU, fused_leaky_relu  # noqa: F401
from .upfirdn2d import upfirdn2d  # noqa: F401
from . import fused_tail  # noqa: F401,E402  (round 6: activation passes folded into their neighbours; registers its entry points)
