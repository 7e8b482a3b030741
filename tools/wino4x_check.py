#!/usr/bin/env python
"""Round 5 gate: the F(4x4,3x3) kernel with its frequency GEMMs on the bf16 matrix pipe (X3: exact three-piece operand split,
conv_wino4.hip) against the fp32-MFMA build of the same kernel: error of both against float64 F.conv2d, and launch times (GPU box).
  python tools/wino4x_check.py [--short]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
import torch.nn.functional as F
from diagan.ops import conv as C
PEAK = 157.3e12


def timeit(f, iters=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def ref64(x, wp, Ci, Co, pro, sc, sh, up):
    """float64 reference on the first 2 images (CPU)"""
    xs = x[:2].double().cpu().permute(0, 3, 1, 2)
    w = wp[:, :9 * Ci].double().cpu().view(Co, 3, 3, Ci).permute(0, 3, 1, 2)
    if pro == C.PRO_AFFINE_RELU:
        xs = F.relu(xs * sc.double().cpu().view(1, -1, 1, 1) + sh.double().cpu().view(1, -1, 1, 1))
    if up:
        xs = F.interpolate(xs, scale_factor=2, mode='bilinear', align_corners=False)
    return F.conv2d(xs, w, padding=1).permute(0, 2, 3, 1)


torch.manual_seed(0)
dev = "cuda"
shapes = [(64, 32, 32, 256, 256), (384, 32, 32, 256, 256), (384, 16, 16, 256, 256), (128, 32, 32, 128, 128), (384, 64, 64, 64, 64),
          (384, 32, 32, 128, 128), (64, 64, 64, 64, 64), (64, 16, 16, 512, 512)]
if '--short' in sys.argv:
    shapes = shapes[:2] + [shapes[4]]
for B, H, W, Ci, Co in shapes:
    geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
    x = torch.randn(B, H, W, Ci, device=dev)
    xh = torch.randn(B, H // 2, W // 2, Ci, device=dev)
    wp = torch.randn(Co, geom.Kp, device=dev) * (9 * Ci) ** -0.5
    sc, sh = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.3
    flop = 2.0 * B * H * W * Co * 9 * Ci
    cases = (("plain", lambda: C.conv_fwd(geom, x, wp, tile_cfg=13), (x, 0, False)),
             ("bn+relu", lambda: C.conv_fwd(geom, x, wp, pro=(C.PRO_AFFINE_RELU, sc, sh), tile_cfg=13), (x, C.PRO_AFFINE_RELU, False)),
             ("upin bn+relu", lambda: C.conv_fwd(geom, xh, wp, pro=(C.PRO_AFFINE_RELU, sc, sh), tile_cfg=15, up_in=True), (xh, C.PRO_AFFINE_RELU, True)))
    for name, f, (xin, pro, up) in cases:
        try:
            C.set_winograd4x(False)
            y0 = f()
            t0 = timeit(f)
            C.set_winograd4x(True)
            y1 = f()
            t1 = timeit(f)
        finally:
            C.set_winograd4x(None)
        r = ref64(xin, wp, Ci, Co, pro, sc, sh, up)
        scale = r.abs().max().item()
        e0 = (y0[:2].double().cpu() - r).abs().max().item() / scale
        e1 = (y1[:2].double().cpu() - r).abs().max().item() / scale
        d = ((y1 - y0).abs().max() / y0.abs().max()).item()
        print(f"B={B:3d} {H:2d}x{W:2d} Ci={Ci:4d} Co={Co:4d} {name:13s} | fp32 {t0*1e6:8.1f} us err64 {e0:.1e} | X3 {t1*1e6:8.1f} us err64 {e1:.1e} "
              f"(X3 - fp32 {d:.1e}) | {t0/t1:4.2f}x  fp32-eq MFMA {flop/4/t1/PEAK:5.1%}", flush=True)
