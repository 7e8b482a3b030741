#!/usr/bin/env python
"""Tile-configuration x split-K sweep of the conv GEMM kernel on the SNGAN-64 / SNGAN-32 layer shapes (GPU box).

All variants of one shape are timed in ONE process in interleaved rounds (median over rounds); every variant's output
is compared with the 64x64 tile's.  Prints one line per (shape, variant) and the best variant per shape."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from diagan import _native as nat
from diagan.ops import conv as C

PEAK = 157.3e12
# name, B, H, W, Ci, Co, R, pro   (B = 128: D(real)+D(fake) pair pass; B = 320: stacked generator forward of a step)
SHAPES = [
    ("D64.b1.c2.pair", 128, 64, 64, 64, 64, 3, 1),
    ("D64.b2.c1.pair", 128, 32, 32, 64, 64, 3, 1),
    ("D64.b2.c2.pair", 128, 32, 32, 64, 128, 3, 1),
    ("D64.b3.c1.pair", 128, 16, 16, 128, 128, 3, 1),
    ("D64.b3.c2.pair", 128, 16, 16, 128, 256, 3, 1),
    ("D64.b4.c1.pair", 128, 8, 8, 256, 256, 3, 1),
    ("D64.b4.c2.pair", 128, 8, 8, 256, 512, 3, 1),
    ("D64.b5.c1.pair", 128, 4, 4, 512, 512, 3, 1),
    ("D64.b5.c2.pair", 128, 4, 4, 512, 1024, 3, 1),
    ("D64.b5.c2.dgrad", 128, 4, 4, 1024, 512, 3, 0),
    ("D64.b4.c2.dgrad", 128, 8, 8, 512, 256, 3, 0),
    ("G64.b2.c1", 64, 8, 8, 1024, 512, 3, 0),
    ("G64.b2.c2", 64, 8, 8, 512, 512, 3, 0),
    ("G64.b3.c1", 64, 16, 16, 512, 256, 3, 0),
    ("G64.b3.c2", 64, 16, 16, 256, 256, 3, 0),
    ("G64.b4.c1", 64, 32, 32, 256, 128, 3, 0),
    ("G64.b4.c2", 64, 32, 32, 128, 128, 3, 0),
    ("G64.b5.c1", 64, 64, 64, 128, 64, 3, 0),
    ("G64.b5.c2", 64, 64, 64, 64, 64, 3, 0),
    ("G64.b2.c1.x5", 320, 8, 8, 1024, 512, 3, 0),
    ("G64.b3.c1.x5", 320, 16, 16, 512, 256, 3, 0),
    ("G64.b5.c2.x5", 320, 64, 64, 64, 64, 3, 0),
    ("D32.b1.c2.pair", 128, 32, 32, 128, 128, 3, 1),
    ("D32.b2.c1.pair", 128, 16, 16, 128, 128, 3, 1),
    ("D32.b3.c1.pair", 128, 8, 8, 128, 128, 3, 1),
    ("G32.b4.c1", 64, 32, 32, 256, 256, 3, 0),
    ("G32.b3.c1", 64, 16, 16, 256, 256, 3, 0),
    ("G32.b2.c1", 64, 8, 8, 256, 256, 3, 0),
]
# (tile cfg, forced split-K (0 = auto), tune flags (-1 = production default), LDS delta in bytes (timing only when < 0))
VARIANTS = [(3, 0, 0, 0), (3, 0, 1, 0), (3, 0, 3, 0), (7, 0, 0, 0), (7, 0, 3, 0), (1, 0, 0, 0), (1, 0, 3, 0),
            (8, 0, 0, 0), (8, 0, 3, 0), (5, 0, 0, 0), (5, 0, 3, 0), (1, 2, 3, 0), (1, 4, 3, 0),
            (3, 0, 0, 8192), (3, 0, 0, 20480), (7, 0, 3, 8192)]
if os.environ.get("VARIANTS"):
    VARIANTS = [tuple(int(v) for v in t.split(",")) for t in os.environ["VARIANTS"].split(";")]


def main():
    only = set(sys.argv[1:])
    rounds, iters = 5, 8
    for name, B, H, W, Ci, Co, R, pro in SHAPES:
        if only and name not in only:
            continue
        geom = C.Geom("conv", Ci, Co, R, R, 1, R // 2)
        x = torch.randn(B, H, W, Ci, device="cuda")
        wp = torch.randn(Co, geom.Kp, device="cuda") * 0.05
        M = B * H * W
        flop = 2.0 * M * Co * R * R * Ci
        prot = (pro, None, None) if pro else None
        ref = None
        outs, runs = {}, []
        for var in VARIANTS:
            cfg, ks, fl, dl = var
            if cfg == 6 and geom.Kp % 64:
                continue
            if ks and (geom.Kp // 64 < ks or ks * M * Co > (16 << 20)):
                continue
            y = torch.empty(B, H, W, Co, device="cuda")

            def run(cfg=cfg, ks=ks, fl=fl, dl=dl, y=y):
                nat.call("diagan_conv_gemm_tune", ks, fl, dl)
                C.conv_fwd(geom, x, wp, out=y, tile_cfg=cfg, pro=prot)
            run()
            torch.cuda.synchronize()
            if ref is None:
                ref = y.clone()
            err = (y - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
            runs.append((var, run, err))
        nat.call("diagan_conv_gemm_tune", 0, -1, 0)
        times = {v: [] for v, _, _ in runs}
        for _ in range(rounds):
            for var, run, _ in runs:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                run()
                s.record()
                for _ in range(iters):
                    run()
                e.record()
                torch.cuda.synchronize()
                times[var].append(s.elapsed_time(e) / iters * 1e-3)
        nat.call("diagan_conv_gemm_tune", 0, -1, 0)
        auto = nat.fn("diagan_conv_gemm_pick_cfg")(M, Co, geom.Kp, 1)
        best = None
        for var, _, err in runs:
            cfg, ks, fl, dl = var
            t = sorted(times[var])[len(times[var]) // 2]
            tf = flop / t / 1e12
            if dl >= 0 and (best is None or t < best[0]):
                best = (t, var)
            print(f"{name:16s} M={M:7d} N={Co:5d} K={R*R*Ci:5d} cfg{cfg} ks{ks} fl{fl} lds{dl:+d}: {t*1e6:8.1f} us {tf:6.1f} TF "
                  f"{tf*1e12/PEAK:5.1%}  err {err:.1e}{'  <- auto' if cfg == auto and ks == 0 and fl == 0 and dl == 0 else ''}"
                  f"{'  (timing only)' if dl < 0 else ''}", flush=True)
        print(f"  best {name}: {best[1]} {best[0]*1e6:.1f} us {flop/best[0]/1e12:.1f} TF", flush=True)


if __name__ == "__main__":
    main()
