#!/usr/bin/env python
"""Diagnostic (GPU box): replay tools/wino4_time.py's exact call sequence (W4_SHORT shapes, three variants, timing loops between
the first calls) REPS times and check every first call of a shape against the implicit GEMM."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from diagan.ops import conv as C

dev = "cuda"
torch.manual_seed(0)
shapes = [(64, 32, 32, 256, 256), (384, 32, 32, 256, 256), (384, 64, 64, 64, 64), (128, 32, 32, 128, 128)]
bad = 0
for rep in range(int(os.environ.get("REPS", 6))):
    for B, H, W, Ci, Co in shapes:
        geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
        x = torch.randn(B, H, W, Ci, device=dev)
        wp = torch.randn(Co, geom.Kp, device=dev) * (9 * Ci) ** -0.5
        sc, sh = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.3
        res = torch.randn(B, H, W, Co, device=dev)
        bias = torch.randn(Co, device=dev)
        resh = torch.randn(B, H // 2, W // 2, Co, device=dev)
        for name, f in (("plain", lambda cfg: C.conv_fwd(geom, x, wp, tile_cfg=cfg)),
                        ("bn+relu+bias+res_up", lambda cfg: C.conv_fwd(geom, x, wp, bias=bias, residual=resh, res_up=True, pro=(C.PRO_AFFINE_RELU, sc, sh), tile_cfg=cfg)),
                        ("bn+relu+bias+res", lambda cfg: C.conv_fwd(geom, x, wp, bias=bias, residual=res, pro=(C.PRO_AFFINE_RELU, sc, sh), tile_cfg=cfg))):
            y13, y9 = f(13), f(9)
            e = ((y13 - y9).abs().max() / f(9).abs().max()).item()
            if not e < 1e-3:
                bad += 1
                ref = f(1)
                scale = ref.abs().max().item()
                for tag, y in (("first 13", y13), ("first 9", y9), ("again 13", f(13)), ("again 9", f(9))):
                    d = (y - ref).abs()
                    w = (d > 1e-3 * scale).nonzero()
                    if w.shape[0]:
                        rows = torch.unique(w[:, 0] * H * W + w[:, 1] * W + w[:, 2])
                        cols = torch.unique(w[:, 3])
                        print(f"rep {rep} B={B} {H}x{W} {Ci}->{Co} {name}: {tag} err {d.max().item() / scale:.2e}; {w.shape[0]} bad values, "
                              f"{rows.numel()} pixels [{rows.min().item()}..{rows.max().item()}], {cols.numel()} channels "
                              f"[{cols.min().item()}..{cols.max().item()}] nan {torch.isnan(y).sum().item()}", flush=True)
                        print("   first bad pixels:", rows[:24].tolist(), flush=True)
            for cfg in (9, 13):
                for _ in range(12):
                    f(cfg)
    print(f"rep {rep}: {bad} mismatches so far", flush=True)
