#!/usr/bin/env python
"""A few launches of each F(4x4) mode on the stacked SNGAN-32 generator shape (384 x 32x32, 256 -> 256), for rocprofv3 --pmc
passes (tools/w4_pmc.sh): MODE 0 plain (<0,0>), MODE 0 with the BatchNorm prologue + statistics + half-resolution residual
(<2,0>: GBlock c2), MODE 3 with the BatchNorm prologue + statistics (<2,3>: GBlock c1 on the half-resolution input)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from diagan.ops import conv as C
dev = "cuda"
B, H, W, Ci, Co = 384, 32, 32, 256, 256
geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
x = torch.randn(B, H, W, Ci, device=dev)
xl = torch.randn(B, H // 2, W // 2, Ci, device=dev)
wp = torch.randn(Co, geom.Kp, device=dev) * (9 * Ci) ** -0.5
sc, sh = torch.rand(6, Ci, device=dev) + 0.5, torch.randn(6, Ci, device=dev) * 0.3
pro = (C.PRO_AFFINE_RELU, sc, sh, B // 6)
bias = torch.randn(Co, device=dev)
resh = torch.randn(B, H // 2, W // 2, Co, device=dev)
for _ in range(int(os.environ.get("W4_REPS", "4"))):
    C.conv_fwd(geom, x, wp, tile_cfg=13)
    C.conv_fwd(geom, x, wp, bias=bias, residual=resh, res_up=True, pro=pro, tile_cfg=13, want_stats=True)
    C.conv_fwd(geom, xl, wp, bias=bias, pro=pro, up_in=True, want_stats=True)
torch.cuda.synchronize()
