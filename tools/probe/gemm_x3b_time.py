#!/usr/bin/env python
"""Round 6: the 128 x 128 split-operand implicit GEMM (csrc/conv_gemm_x3b.hip, tile_cfg 17) against the fp32 implicit GEMM
(tile_cfg 1 / the automatic fp32 choice) on the StyleGAN2 256 x 256, batch 32 launch shapes that no Winograd kernel takes:
error of both against float64 on a small case (also through an output map), then launch times.
    python tools/probe/gemm_x3b_time.py [--reps 5]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def ref64(x, w_oihw, stride, pad):
    return F.conv2d(x.double().permute(0, 3, 1, 2).cpu(), w_oihw.double().cpu(), stride=stride, padding=pad).permute(0, 2, 3, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    from diagan.ops import conv as K
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)

    def make(B, H, W, Ci, Co, R, S):
        x = torch.randn(B, H, W, Ci, device=dev, generator=g)
        w = torch.randn(Co, Ci, R, S, device=dev, generator=g) / (Ci * R * S) ** 0.5
        return x, w

    short = os.environ.get("XB_SHORT") == "1"          # ablation builds: three shapes, times only
    print("== error against float64 (max |err| / max |ref|) ==")
    for (B, H, W, Ci, Co, R, S, st, pd) in [] if short else [(2, 33, 33, 64, 128, 3, 3, 2, 0), (3, 20, 24, 96, 72, 2, 2, 1, 1), (2, 16, 16, 128, 256, 1, 1, 1, 0),
                                            (2, 19, 21, 32, 64, 2, 1, 1, 1)]:
        x, w = make(B, H, W, Ci, Co, R, S)
        geom = K.Geom('conv', Ci, K.round_up(Co, 4), R, S, st, pd)
        wp = K.pack_oihw(w, geom.Kp)
        ref = ref64(x, w, st, pd)
        sc = ref.abs().max().item()
        row = []
        for cfg in (1, 17):
            y = K.conv_fwd(geom, x, wp, tile_cfg=cfg, wino=False)
            row.append((y.double().cpu()[..., :Co] - ref).abs().max().item() / sc)
        print(f"  {R}x{S} s{st} p{pd} {Ci}->{Co} on {B}x{H}x{W}: fp32 tile_cfg 1 {row[0]:.2e}   split-operand tile_cfg 17 {row[1]:.2e}")
    # output map: a 2x2 / pad 1 class written to the even pixels of a larger tensor, border rows / columns trimmed
    if short:
        return timing(a, K, dev, make, [(32, 257, 257, 128, 256, 3, 3, 2, 0), (32, 64, 64, 512, 256, 2, 2, 1, 1), (32, 64, 64, 512, 256, 1, 1, 1, 0)], (17,))
    B, H, W, Ci, Co = 2, 12, 12, 64, 128
    x, w = make(B, H, W, Ci, Co, 2, 2)
    geom = K.Geom('conv', Ci, Co, 2, 2, 1, 1)
    wp = K.pack_oihw(w, geom.Kp)
    full = K.conv_fwd(geom, x, wp, tile_cfg=17, wino=False)                      # [B, H+1, W+1, Co]
    out = torch.full((B, 2 * H + 1, 2 * W + 1, Co), -7.0, device=dev)
    K.conv_fwd(geom, x, wp, tile_cfg=17, wino=False, out=out, out_map=(2, 0, 0, 0, H + 1, 0, W + 1))
    ok = torch.equal(out[:, 0::2, 0::2], full) and bool((out[:, 1::2] == -7).all()) and bool((out[:, :, 1::2] == -7).all())
    out2 = torch.full((B, 2 * H - 1, 2 * W - 1, Co), -7.0, device=dev)
    K.conv_fwd(geom, x, wp, tile_cfg=17, wino=False, out=out2, out_map=(2, 1, 1, 1, H, 1, W))
    ok2 = torch.equal(out2[:, 1::2, 1::2], full[:, 1:H, 1:W]) and bool((out2[:, 0::2] == -7).all())
    print(f"  output map: whole grid to even pixels {'OK' if ok else 'MISMATCH'}, trimmed window to odd pixels {'OK' if ok2 else 'MISMATCH'}")

    shapes = [(32, 257, 257, 128, 256, 3, 3, 2, 0), (32, 129, 129, 256, 512, 3, 3, 2, 0), (32, 65, 65, 512, 512, 3, 3, 2, 0),
              (32, 128, 128, 256, 128, 2, 2, 1, 1), (32, 64, 64, 512, 256, 2, 2, 1, 1), (32, 32, 32, 512, 512, 2, 2, 1, 1),
              (32, 128, 128, 256, 128, 2, 1, 1, 1), (32, 64, 64, 512, 256, 1, 2, 1, 1),
              (32, 64, 64, 512, 256, 1, 1, 1, 0), (32, 128, 128, 256, 128, 1, 1, 1, 0), (32, 16, 16, 512, 512, 2, 2, 1, 1),
              (32, 255, 255, 128, 256, 1, 1, 2, 0)]
    timing(a, K, dev, make, shapes, (1, 17))


def timing(a, K, dev, make, shapes, cfgs):
    print("== launch times (us; TFLOP/s in direct-convolution FLOP) ==")
    for (B, H, W, Ci, Co, R, S, st, pd) in shapes:
        x, w = make(B, H, W, Ci, Co, R, S)
        geom = K.Geom('conv', Ci, Co, R, S, st, pd)
        wp = K.pack_oihw(w, geom.Kp)
        Ho, Wo = geom.out_hw(H, W)
        flop = 2.0 * B * Ho * Wo * Co * R * S * Ci
        out = torch.empty((B, Ho, Wo, Co), device=dev)
        row = []
        for cfg in cfgs:
            for _ in range(2):
                K.conv_fwd(geom, x, wp, tile_cfg=cfg, wino=False, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                K.conv_fwd(geom, x, wp, tile_cfg=cfg, wino=False, out=out)
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) / a.reps * 1e3
            row.append((us, flop / us / 1e6))
        if len(row) == 1:
            print(f"  {R}x{S} s{st} p{pd} {Ci:4d}->{Co:4d} on {B}x{H}x{W}: tile_cfg {cfgs[0]} {row[0][0]:8.1f} us {row[0][1]:6.1f}")
            continue
        print(f"  {R}x{S} s{st} p{pd} {Ci:4d}->{Co:4d} on {B}x{H}x{W}: tile_cfg 1 {row[0][0]:8.1f} us {row[0][1]:6.1f}   "
              f"tile_cfg 17 {row[1][0]:8.1f} us {row[1][1]:6.1f}   x{row[0][0] / row[1][0]:.2f}")


if __name__ == "__main__":
    main()
