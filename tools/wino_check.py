#!/usr/bin/env python
"""Winograd F(2x2,3x3) kernel (tile_cfg 9) against the implicit-GEMM kernel (tile_cfg 7 / auto) on the same inputs:
max error relative to the output scale for every fused prologue / epilogue variant, and timings (GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from diagan.ops import conv as C

PEAK = 157.3e12


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
    shapes = [(4, 8, 8, 64, 64), (2, 16, 16, 32, 64), (4, 4, 16, 16, 8), (3, 6, 10, 16, 24), (2, 4, 4, 8, 4), (5, 16, 16, 128, 72), (64, 32, 32, 256, 256),
              (128, 64, 64, 64, 64), (64, 8, 8, 1024, 512), (128, 4, 4, 512, 1024), (128, 16, 16, 128, 256),
              (320, 32, 32, 256, 128), (64, 64, 64, 128, 64)]
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
        sc, sh = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.3
        dy = torch.randn(B, H, W, Co, device=dev)
        msk = torch.randn(B, H, W, Ci, device=dev)
        resx = torch.randn(B, H, W, Ci, device=dev)
        s0, s1 = torch.tensor([0.7], device=dev), torch.tensor([1.3], device=dev)
        big = B * H * W >= 2048
        from diagan import _native as nat
        staged = False        # (the staged-input kernel, tile_cfg 10, was retired in round 3)
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
            got = f(9)
            torch.cuda.synchronize()
            e = relerr(got, ref)
            worst = max(worst, e)
            line = f"B={B:3d} {H:2d}x{W:2d} Ci={Ci:4d} Co={Co:4d} {name:22s} err {e:.2e}"
            if staged and not (name.startswith("dgrad") and Co % 16):
                es = relerr(f(10), ref)
                worst = max(worst, es)
                line += f" staged {es:.2e}"
            if big and name in ("fwd plain", "fwd bn+relu+bias+res", "dgrad mask+res"):
                flop = 2.0 * B * H * W * Co * 9 * Ci
                ta, tw = timeit(lambda: f(0)), timeit(lambda: f(9))
                line += (f" | auto {ta*1e6:8.1f} us {flop/ta/1e12:6.1f} TF | winograd {tw*1e6:8.1f} us "
                         f"{flop/tw/1e12:6.1f} TF-equivalent ({ta/tw:4.2f}x; MFMA util {flop/2.25/tw/PEAK:5.1%})")
                if staged:
                    ts = timeit(lambda: f(10))
                    line += f" | staged {ts*1e6:8.1f} us {flop/ts/1e12:6.1f} TF-eq (util {flop/2.25/ts/PEAK:5.1%})"
            print(line, flush=True)
        # fused BatchNorm statistics
        y7, st7 = C.conv_fwd(geom, x, wp, bias=bias, tile_cfg=7, want_stats=True)
        y9, st9 = C.conv_fwd(geom, x, wp, bias=bias, tile_cfg=9, want_stats=True)
        a, b = st7[0].sum(0), st9[0].sum(0)
        e = relerr(b, a)
        worst = max(worst, e)
        if staged:
            y10, st10 = C.conv_fwd(geom, x, wp, bias=bias, tile_cfg=10, want_stats=True)
            e = max(e, relerr(st10[0].sum(0), a), relerr(y10, y7))
            worst = max(worst, e)
        print(f"B={B:3d} {H:2d}x{W:2d} Ci={Ci:4d} Co={Co:4d} {'fwd stats (col sums)':22s} err {e:.2e}", flush=True)
        # stacked forward: per-group affine prologue
        if B % 4 == 0 and (B // 4) * H * W % 256 == 0:
            gsc, gsh = torch.rand(4, Ci, device=dev) + 0.5, torch.randn(4, Ci, device=dev) * 0.3
            pro = (C.PRO_AFFINE_RELU, gsc, gsh, B // 4)
            e = relerr(C.conv_fwd(geom, x, wp, pro=pro, tile_cfg=9), C.conv_fwd(geom, x, wp, pro=pro, tile_cfg=7))
            if staged:
                e = max(e, relerr(C.conv_fwd(geom, x, wp, pro=pro, tile_cfg=10), C.conv_fwd(geom, x, wp, pro=pro, tile_cfg=7)))
            worst = max(worst, e)
            print(f"B={B:3d} {H:2d}x{W:2d} Ci={Ci:4d} Co={Co:4d} {'fwd grouped bn+relu':22s} err {e:.2e}", flush=True)
    # ---- split-K of the forward kernel (few output tiles, long channel loop) ----
    from diagan import _native as nat
    for B, H, W, Ci, Co in ((64, 8, 8, 1024, 512), (128, 4, 4, 512, 1024), (128, 8, 8, 256, 256), (128, 8, 8, 128, 128)):
        geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
        x = torch.randn(B, H, W, Ci, device=dev)
        wp = torch.randn(Co, geom.Kp, device=dev) * (9 * Ci) ** -0.5
        bias, res = torch.randn(Co, device=dev), torch.randn(B, H, W, Co, device=dev)
        f = lambda cfg: C.conv_fwd(geom, x, wp, bias=bias, residual=res, pro=(C.PRO_RELU, None, None), tile_cfg=cfg)
        ref = f(7)
        flop = 2.0 * B * H * W * Co * 9 * Ci
        line = f"B={B:3d} {H:2d}x{W:2d} Ci={Ci:4d} Co={Co:4d} split-K:"
        for ks in (1, 2, 4):
            nat.call("diagan_conv_gemm_tune", ks, -1, 0)
            e = relerr(f(9), ref)
            worst = max(worst, e)
            t = timeit(lambda: f(9))
            line += f"  ks{ks} err {e:.1e} {t*1e6:7.1f} us"
        if False:
            line += " | staged"
            for ks in (1, 2, 4):
                nat.call("diagan_conv_gemm_tune", ks, -1, 0)
                e = relerr(f(10), ref)
                worst = max(worst, e)
                t = timeit(lambda: f(10))
                line += f"  ks{ks} err {e:.1e} {t*1e6:7.1f} us"
        nat.call("diagan_conv_gemm_tune", 0, -1, 0)
        C.set_winograd(False)
        td = timeit(lambda: f(0))
        C.set_winograd(None)
        ta = timeit(lambda: f(0))
        print(line + f"  | implicit GEMM {td*1e6:7.1f} us | auto {ta*1e6:7.1f} us ({flop/ta/1e12:5.1f} TF-eq)", flush=True)
    # ---- weight gradient: Winograd F(3x3,2x2) against the implicit-GEMM weight gradient (same slab API) ----
    for B, H, W, Ci, Co in shapes:
        if Ci < 16 or Co < 16:
            continue
        geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
        x = torch.randn(B, H, W, Ci, device=dev)
        dy = torch.randn(B, H, W, Co, device=dev)
        sc, sh = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.3
        n_w = Co * geom.Kp
        stride = n_w + Co

        def run(wino, pro, segments=1):
            C.set_winograd(wino)
            try:
                splits = max(segments, C.wgrad_splits_geom(geom, B, H, W, H, W) // segments * segments)
                slab = torch.full((splits * stride,), float("nan"), device=dev)
                C.conv_wgrad_into(geom, dy, x, slab, splits, stride, n_w, pro=pro, segments=segments)
                torch.cuda.synchronize()
                return slab.view(splits, stride).sum(0), splits
            finally:
                C.set_winograd(None)
        for name, pro in (("wgrad plain", None), ("wgrad relu", (C.PRO_RELU, None, None)),
                          ("wgrad bn+relu", (C.PRO_AFFINE_RELU, sc, sh))):
            ref, s0_ = run(False, pro)
            got, s1_ = run(True, pro)
            e = relerr(got[:n_w], ref[:n_w])
            eb = relerr(got[n_w:], ref[n_w:])
            worst = max(worst, e, eb)
            line = f"B={B:3d} {H:2d}x{W:2d} Ci={Ci:4d} Co={Co:4d} {name:22s} err {e:.2e} bias err {eb:.2e} splits {s0_}/{s1_}"
            if B * H * W >= 2048 and name != "wgrad relu":
                flop = 2.0 * B * H * W * Co * 9 * Ci
                ta = timeit(lambda: run(False, pro))
                tw = timeit(lambda: run(True, pro))
                line += (f" | gemm {ta*1e6:8.1f} us | winograd {tw*1e6:8.1f} us (incl. slab alloc + sum; {ta/tw:4.2f}x)")
            print(line, flush=True)
        if B % 2 == 0 and (B // 2) * H * W % 32 == 0:
            ref, _ = run(False, (C.PRO_RELU, None, None), segments=2)
            got, _ = run(True, (C.PRO_RELU, None, None), segments=2)
            e = relerr(got, ref)
            worst = max(worst, e)
            print(f"B={B:3d} {H:2d}x{W:2d} Ci={Ci:4d} Co={Co:4d} {'wgrad 2 segments':22s} err {e:.2e}", flush=True)
    print(f"worst relative error {worst:.2e}")
    return 0 if worst < 2e-5 else 1


if __name__ == "__main__":
    sys.exit(main())
