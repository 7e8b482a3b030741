#!/usr/bin/env python
"""Launch times of the X3 build of the F(4x4,3x3) kernel (and of its compile-time ablation variants, DIAGAN_LIB_PATH) on the two
dominant SNGAN-32 shapes; no error check (ablated builds compute garbage).  GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from diagan.ops import conv as C


def timeit(f, iters=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


torch.manual_seed(0)
dev = "cuda"
out = []
for B, H, W, Ci, Co in [(64, 32, 32, 256, 256), (384, 32, 32, 256, 256), (384, 64, 64, 64, 64)]:
    geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
    x = torch.randn(B, H, W, Ci, device=dev)
    xh = torch.randn(B, H // 2, W // 2, Ci, device=dev)
    wp = torch.randn(Co, geom.Kp, device=dev) * (9 * Ci) ** -0.5
    sc, sh = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.3
    C.set_winograd4x('--fp32' not in sys.argv)
    t = [timeit(lambda: C.conv_fwd(geom, x, wp, tile_cfg=13)),
         timeit(lambda: C.conv_fwd(geom, x, wp, pro=(C.PRO_AFFINE_RELU, sc, sh), tile_cfg=13)),
         timeit(lambda: C.conv_fwd(geom, xh, wp, pro=(C.PRO_AFFINE_RELU, sc, sh), up_in=True))]
    out.append(f"{B}x{H}x{W} {Ci}->{Co}: plain {t[0]:7.1f}  bn {t[1]:7.1f}  upin {t[2]:7.1f} us")
print(" | ".join(out), flush=True)
