"""wgrad time vs pixel count at fixed (Co, K): fixed cost vs per-K-step rate (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from diagan.ops import conv as C
from bench_conv import timeit
for (Ci, Co, H) in ((128, 128, 32), (256, 256, 32)):
    pts = []
    for B in (16, 32, 64, 128, 256):
        geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
        x = torch.randn(B, H, H, Ci, device="cuda"); dy = torch.randn(B, H, H, Co, device="cuda")
        M = B * H * H
        splits = C.wgrad_splits(M, Co, geom.Kp)
        stride = Co * geom.Kp + Co
        slab = torch.empty(splits * stride, device="cuda")
        for _ in range(2):
            t = timeit(lambda: C.conv_wgrad_into(geom, dy, x, slab, splits, stride, Co * geom.Kp, pro=(1, None, None)), iters=20)
        steps = M // 32 // splits
        pts.append((steps, t))
        print(f"M={M} N={Co} K={9*Ci} splits={splits} steps/block={steps}: {t*1e6:7.1f} us {2.0*M*Co*9*Ci/t/1e12:6.1f} TF", flush=True)
    (s0, t0), (s1, t1) = pts[1], pts[-1]
    b = (t1 - t0) / (s1 - s0); a_ = t0 - b * s0
    print(f"  fit: fixed {a_*1e6:.1f} us + {b*1e6:.3f} us per K-step (ideal at 154 TF, 2 waves/SIMD: {8192/2.35e3:.3f} us)")
