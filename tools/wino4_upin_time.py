#!/usr/bin/env python
"""tile_cfg 15 (F(4x4) on the bilinear x2 of a half-resolution input, conv_wino4.hip MODE 3) against the two launches it
replaces (diagan_upsample2x with the BatchNorm + ReLU prologue, then tile_cfg 13) on the stacked SNGAN generator shapes:
error against float64 on a slice and against the two-launch form, and the times of both (GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
import torch.nn.functional as F
from diagan.ops import conv as C
from diagan.ops import eltwise as E


def timeit(f, iters=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def relerr(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def main():
    torch.manual_seed(0)
    dev = "cuda"
    # B, Hl, Wl, Ci, Co: SNGAN-32 blocks 4 / 3 / 2 and SNGAN-64 blocks 5 / 4 / 3 of the stacked forward (6 x 64 images)
    shapes = [(384, 16, 16, 256, 256), (384, 8, 8, 256, 256), (384, 4, 4, 256, 256), (384, 32, 32, 128, 64),
              (384, 16, 16, 256, 128), (384, 8, 8, 512, 256), (64, 16, 16, 256, 256)]
    C.set_winograd4('force-pool')
    for B, Hl, Wl, Ci, Co in shapes:
        geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
        x = torch.randn(B, Hl, Wl, Ci, device=dev)
        wp = torch.zeros(Co, geom.Kp, device=dev)
        wp[:, : 9 * Ci] = torch.randn(Co, 9 * Ci, device=dev) * (9 * Ci) ** -0.5
        bias = torch.randn(Co, device=dev)
        G = 6 if B % 6 == 0 else 1
        sc, sh = torch.rand(G, Ci, device=dev) + 0.5, torch.randn(G, Ci, device=dev) * 0.3
        pro = (C.PRO_AFFINE_RELU, sc, sh, B // G) if G > 1 else (C.PRO_AFFINE_RELU, sc[0].contiguous(), sh[0].contiguous())

        def fused():
            return C.conv_fwd(geom, x, wp, bias=bias, pro=pro, up_in=True, want_stats=True)[0]

        def two():
            return C.conv_fwd(geom, E.upsample2x(x, pro=pro), wp, bias=bias, tile_cfg=13, want_stats=True)[0]

        def conv_only(u):
            return C.conv_fwd(geom, u, wp, bias=bias, tile_cfg=13, want_stats=True)[0]
        a, b = fused(), two()
        torch.cuda.synchronize()
        nb = 2
        xs = x[:nb].double().permute(0, 3, 1, 2) * sc[0].double().view(1, -1, 1, 1) + sh[0].double().view(1, -1, 1, 1)
        up = F.interpolate(F.relu(xs), scale_factor=2, mode='bilinear', align_corners=False)
        w64 = wp[:, : 9 * Ci].double().view(Co, 3, 3, Ci).permute(0, 3, 1, 2)
        ref = F.conv2d(up, w64, bias.double(), padding=1).permute(0, 2, 3, 1)
        u = E.upsample2x(x, pro=pro)
        t_f, t_2, t_c, t_u = timeit(fused), timeit(two), timeit(lambda: conv_only(u)), timeit(lambda: E.upsample2x(x, pro=pro))
        flop = 2.0 * B * 4 * Hl * Wl * Co * 9 * Ci
        print(f"B={B:3d} {Hl:2d}x{Wl:2d}->x2 Ci={Ci:4d} Co={Co:4d} | vs f64: fused {relerr(a[:nb].double(), ref):.2e} two-launch "
              f"{relerr(b[:nb].double(), ref):.2e} | fused vs two {relerr(a, b):.2e} | fused {t_f*1e6:8.1f} us ({flop/4/t_f/157.3e12:5.1%} MFMA) | "
              f"upsample {t_u*1e6:7.1f} + conv {t_c*1e6:8.1f} = {t_2*1e6:8.1f} us | {t_2/t_f:4.2f}x", flush=True)
    C.set_winograd4(None)


if __name__ == "__main__":
    main()
