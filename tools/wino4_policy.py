#!/usr/bin/env python
"""Launch-size policy of the F(4x4,3x3) kernel: SNGAN-32 / SNGAN-64 3x3 layer shapes (forward stacked x6 and x1, pair pass,
data-gradient) timed with the automatic F(2x2) choice (tile_cfg 9 incl. its own split-K policy) against tile_cfg 13 with
forced channel splits 1 / 2 / 4 (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from diagan.ops import conv as C
from diagan import _native as nat

def timeit(f, iters=8):
    for _ in range(2): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-6 * 1e9   # us

dev = "cuda"
shapes = []
for B in (64, 128, 384):
    shapes += [(B, 8, 8, 256, 256), (B, 16, 16, 256, 256), (B, 32, 32, 256, 256), (B, 32, 32, 128, 128), (B, 16, 16, 128, 128),
               (B, 8, 8, 1024, 512), (B, 8, 8, 512, 512), (B, 16, 16, 512, 256), (B, 16, 16, 256, 256), (B, 32, 32, 256, 128),
               (B, 32, 32, 128, 128), (B, 64, 64, 128, 64), (B, 64, 64, 64, 64), (B, 16, 16, 256, 512), (B, 32, 32, 128, 256),
               (B, 64, 64, 64, 128)]
seen = set()
for B, H, W, Ci, Co in shapes:
    if (B, H, W, Ci, Co) in seen or B * H * W * max(Ci, Co) * 4 >= 2 ** 31:
        continue
    seen.add((B, H, W, Ci, Co))
    geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
    x = torch.randn(B, H, W, Ci, device=dev)
    wp = torch.randn(Co, geom.Kp, device=dev) * (9 * Ci) ** -0.5
    wgs4 = -(-B * (H // 4) * (W // 4) // 32) * -(-Co // 64)
    nat.call("diagan_conv_gemm_tune", 0, -1, 0)
    t9 = timeit(lambda: C.conv_fwd(geom, x, wp, tile_cfg=9))
    C.set_winograd4(False)
    ta = timeit(lambda: C.conv_fwd(geom, x, wp))
    C.set_winograd4(None)
    res = {}
    for ks in (1, 2, 4):
        if Ci // 8 // ks < 4:
            continue
        nat.call("diagan_conv_gemm_tune", ks if ks > 1 else 0, -1, 0)
        try:
            res[ks] = timeit(lambda: C.conv_fwd(geom, x, wp, tile_cfg=13))
        except RuntimeError as e:
            res[ks] = float('nan')
    nat.call("diagan_conv_gemm_tune", 0, -1, 0)
    best = min(res, key=lambda k: res[k])
    print(f"B={B:3d} {H:2d}x{W:2d} {Ci:4d}->{Co:4d} wgs4={wgs4:5d} | auto(no F4) {ta:8.1f} cfg9 {t9:8.1f} | F(4x4) " +
          " ".join(f"ks{k} {v:8.1f}" for k, v in res.items()) + f" | best ks{best} {min(ta, t9)/res[best]:4.2f}x", flush=True)
