#!/usr/bin/env python
"""Winograd F(4x4,3x3) kernel (tile_cfg 13) against the implicit-GEMM kernel (tile_cfg 7) and float64 on the same inputs:
max error relative to the output scale for the fused prologue / epilogue variants, and timings against F(2x2,3x3)
(tile_cfg 9) and the automatic choice (GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from diagan.ops import conv as C

PEAK = 157.3e12
QUICK = "--quick" in sys.argv


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
    shapes = [(2, 8, 8, 16, 64), (3, 4, 12, 8, 24), (2, 16, 16, 32, 64), (5, 16, 16, 128, 72), (64, 32, 32, 256, 256),
              (384, 32, 32, 256, 256), (128, 32, 32, 128, 128), (64, 64, 64, 64, 64), (384, 16, 16, 256, 256), (64, 16, 16, 256, 256),
              (64, 64, 64, 128, 64), (128, 16, 16, 128, 256)]
    if QUICK:
        shapes = shapes[:5]
    worst = 0.0
    for B, H, W, Ci, Co in shapes:
        geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
        x = torch.randn(B, H, W, Ci, device=dev)
        wp = torch.zeros(Co, geom.Kp, device=dev)
        wp[:, : 9 * Ci] = torch.randn(Co, 9 * Ci, device=dev) * (9 * Ci) ** -0.5
        wd = torch.zeros(Ci, geom.Kd, device=dev)
        C.pack_weights(wp, Co, Ci, 9, geom.Kp, geom.Kd, Wd=wd)
        bias = torch.randn(Co, device=dev)
        res = torch.randn(B, H, W, Co, device=dev)
        resh = torch.randn(B, H // 2, W // 2, Co, device=dev)
        sc, sh = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.3
        dy = torch.randn(B, H, W, Co, device=dev)
        msk = torch.randn(B, H, W, Ci, device=dev)
        resx = torch.randn(B, H, W, Ci, device=dev)
        s0, s1 = torch.tensor([0.7], device=dev), torch.tensor([1.3], device=dev)
        big = B * H * W >= 16384
        variants = {
            "fwd plain": lambda cfg: C.conv_fwd(geom, x, wp, tile_cfg=cfg),
            "fwd relu+bias": lambda cfg: C.conv_fwd(geom, x, wp, bias=bias, pro=(C.PRO_RELU, None, None), tile_cfg=cfg),
            "fwd lrelu+res(relu)": lambda cfg: C.conv_fwd(geom, x, wp, residual=res, res_relu=True,
                                                          pro=(C.PRO_LRELU, None, None), tile_cfg=cfg),
            "fwd bn+relu+bias+res": lambda cfg: C.conv_fwd(geom, x, wp, bias=bias, residual=res,
                                                           pro=(C.PRO_AFFINE_RELU, sc, sh), tile_cfg=cfg),
            "fwd affine": lambda cfg: C.conv_fwd(geom, x, wp, pro=(C.PRO_AFFINE, sc, sh), tile_cfg=cfg),
            "dgrad plain": lambda cfg: C.conv_dgrad(geom, dy, wd, (H, W), tile_cfg=cfg),
            "dgrad mask+res": lambda cfg: C.conv_dgrad(geom, dy, wd, (H, W), residual=resx, mask_src=msk, tile_cfg=cfg),
        }
        if B % 2 == 0:
            variants["fwd pair scales"] = lambda cfg: C.conv_fwd(geom, x, wp, pro=(C.PRO_RELU, None, None),
                                                                 row_scale=(s0, s1), tile_cfg=cfg)
        for name, f in variants.items():
            if name.startswith("dgrad") and Co % 8:
                continue                      # the data-gradient's input channels are the layer's Co
            ref = f(7)
            got = f(13)
            torch.cuda.synchronize()
            e = relerr(got, ref)
            worst = max(worst, e)
            line = f"B={B:3d} {H:2d}x{W:2d} Ci={Ci:4d} Co={Co:4d} {name:22s} err {e:.2e}"
            if big and name in ("fwd plain", "fwd bn+relu+bias+res", "dgrad mask+res"):
                flop = 2.0 * B * H * W * Co * 9 * Ci
                t9, t13 = timeit(lambda: f(9)), timeit(lambda: f(13))
                line += (f" | F(2x2) {t9*1e6:8.1f} us {flop/t9/1e12:6.1f} TF-eq (MFMA {flop/2.25/t9/PEAK:5.1%}) | F(4x4) {t13*1e6:8.1f} us "
                         f"{flop/t13/1e12:6.1f} TF-eq (MFMA {flop/4/t13/PEAK:5.1%})  {t9/t13:4.2f}x")
            print(line, flush=True)
        # the half-resolution residual (GBlock shortcut), against the F(2x2) kernel's blend
        if H % 4 == 0 and W % 4 == 0:
            e = relerr(C.conv_fwd(geom, x, wp, bias=bias, residual=resh, res_up=True, tile_cfg=13),
                       C.conv_fwd(geom, x, wp, bias=bias, residual=resh, res_up=True, tile_cfg=9))
            worst = max(worst, e)
            print(f"B={B:3d} {H:2d}x{W:2d} Ci={Ci:4d} Co={Co:4d} {'fwd res_up':22s} err {e:.2e}", flush=True)
        # float64 truth on a slice of the batch
        nb = min(B, 4)
        x64 = x[:nb].double().permute(0, 3, 1, 2)
        w64 = wp[:, : 9 * Ci].double().view(Co, 3, 3, Ci).permute(0, 3, 1, 2)
        ref64 = torch.nn.functional.conv2d(x64, w64, padding=1).permute(0, 2, 3, 1)
        errs = {cfg: relerr(C.conv_fwd(geom, x[:nb].contiguous(), wp, tile_cfg=cfg).double(), ref64) for cfg in (7, 9, 13)}
        print(f"B={B:3d} {H:2d}x{W:2d} Ci={Ci:4d} Co={Co:4d} {'vs float64':22s} implicit GEMM {errs[7]:.2e}  F(2x2) {errs[9]:.2e}  "
              f"F(4x4) {errs[13]:.2e}", flush=True)
        # fused BatchNorm statistics
        y7, st7 = C.conv_fwd(geom, x, wp, bias=bias, tile_cfg=7, want_stats=True)
        y13, st13 = C.conv_fwd(geom, x, wp, bias=bias, tile_cfg=13, want_stats=True)
        e = relerr(st13[0].sum(0), st7[0].sum(0))
        worst = max(worst, e)
        print(f"B={B:3d} {H:2d}x{W:2d} Ci={Ci:4d} Co={Co:4d} {'fwd stats (col sums)':22s} err {e:.2e}", flush=True)
        # stacked forward: per-group affine prologue
        if B % 4 == 0 and (B // 4) * H * W % 512 == 0:
            gsc, gsh = torch.rand(4, Ci, device=dev) + 0.5, torch.randn(4, Ci, device=dev) * 0.3
            pro = (C.PRO_AFFINE_RELU, gsc, gsh, B // 4)
            e = relerr(C.conv_fwd(geom, x, wp, pro=pro, tile_cfg=13), C.conv_fwd(geom, x, wp, pro=pro, tile_cfg=7))
            worst = max(worst, e)
            print(f"B={B:3d} {H:2d}x{W:2d} Ci={Ci:4d} Co={Co:4d} {'fwd grouped bn+relu':22s} err {e:.2e}", flush=True)
    print(f"WORST relative error vs the implicit GEMM: {worst:.2e}")


if __name__ == "__main__":
    main()
