#!/usr/bin/env python
"""Timing of the F(4x4,3x3) kernel (tile_cfg 13) against F(2x2,3x3) (9) on the big SNGAN launches (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from diagan.ops import conv as C
PEAK = 157.3e12

def timeit(f, iters=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

torch.manual_seed(0)
dev = "cuda"
shapes = [(64, 32, 32, 256, 256), (384, 32, 32, 256, 256), (384, 16, 16, 256, 256), (128, 32, 32, 128, 128), (384, 64, 64, 64, 64),
          (384, 32, 32, 128, 128), (64, 64, 64, 64, 64), (384, 8, 8, 256, 256), (64, 16, 16, 256, 256), (384, 16, 16, 256, 256)]
if os.environ.get('W4_SHORT'):
    shapes = [shapes[0], shapes[1], shapes[4], shapes[3]]
for B, H, W, Ci, Co in shapes:
    geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
    x = torch.randn(B, H, W, Ci, device=dev)
    wp = torch.randn(Co, geom.Kp, device=dev) * (9 * Ci) ** -0.5
    sc, sh = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.3
    res = torch.randn(B, H, W, Co, device=dev)
    bias = torch.randn(Co, device=dev)
    flop = 2.0 * B * H * W * Co * 9 * Ci
    resh = torch.randn(B, H // 2, W // 2, Co, device=dev)
    for name, f in (("plain", lambda cfg: C.conv_fwd(geom, x, wp, tile_cfg=cfg)),
                    ("bn+relu+bias+res_up", lambda cfg: C.conv_fwd(geom, x, wp, bias=bias, residual=resh, res_up=True, pro=(C.PRO_AFFINE_RELU, sc, sh), tile_cfg=cfg)),
                    ("bn+relu+bias+res", lambda cfg: C.conv_fwd(geom, x, wp, bias=bias, residual=res, pro=(C.PRO_AFFINE_RELU, sc, sh), tile_cfg=cfg))):
        y13, y9 = f(13), f(9)
        e = ((y13 - y9).abs().max() / y9.abs().max()).item()
        if not e < 1e-3:       # seen twice on one (11 % slower) box and never again: say WHICH result is off, and where
            ref = f(1)
            for tag, y in (("first F(4x4)", y13), ("first F(2x2)", y9), ("second F(4x4)", f(13)), ("second F(2x2)", f(9))):
                d = (y - ref).abs() > 1e-3 * ref.abs().max()
                if d.any():
                    w = d.nonzero()
                    print(f"   MISMATCH {tag} vs implicit GEMM: {w.shape[0]} values, images {w[:, 0].unique().tolist()[:16]}, "
                          f"rows {w[:, 1].unique().tolist()[:16]}, channels {w[:, 3].min().item()}..{w[:, 3].max().item()}", flush=True)
        t9, t13 = timeit(lambda: f(9)), timeit(lambda: f(13))
        print(f"B={B:3d} {H:2d}x{W:2d} Ci={Ci:4d} Co={Co:4d} {name:18s} err {e:.1e} | F(2x2) {t9*1e6:8.1f} us (MFMA {flop/2.25/t9/PEAK:5.1%}) | "
              f"F(4x4) {t13*1e6:8.1f} us {flop/t13/1e12:6.1f} TF-eq (MFMA {flop/4/t13/PEAK:5.1%})  {t9/t13:4.2f}x", flush=True)
