"""time(K) = a + b*K for the 128x128 kernel at fixed M, N: separates per-tile fixed cost from the K loop (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from diagan.ops import conv as C
from bench_conv import timeit
CFG = int(os.environ.get("CFG", "1"))
SHAPES = ((64, 32, 256), (128, 32, 128)) if CFG == 1 else ((128, 16, 128), (64, 16, 256), (128, 8, 128))
for (B, H, Co) in SHAPES:
    pts = []
    for Ci in (32, 64, 128, 256, 512):
        geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
        x = torch.randn(B, H, H, Ci, device="cuda"); wp = torch.randn(Co, geom.Kp, device="cuda") * 0.05
        y = torch.empty(B, H, H, Co, device="cuda")
        for _ in range(2):
            t = timeit(lambda: C.conv_fwd(geom, x, wp, out=y, tile_cfg=CFG), iters=30)
        K = 9 * Ci
        pts.append((K, t))
        print(f"M={B*H*H} N={Co} K={K}: {t*1e6:7.1f} us {2.0*B*H*H*Co*K/t/1e12:6.1f} TF", flush=True)
    (k0, t0), (k1, t1) = pts[1], pts[-1]
    b = (t1 - t0) / (k1 - k0)
    a = t0 - b * k0
    M = B * H * H
    print(f"  fit: fixed {a*1e6:.1f} us + {b*1e9:.2f} ns per k  -> loop rate {2.0*M*Co/b/1e12:.1f} TF, fixed share at K=2304: {a/(a+b*2304):.1%}")
