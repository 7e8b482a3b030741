#!/usr/bin/env python
"""Round 6: the split-operand weight gradient (csrc/conv_wgrad_x3.hip) against the fp32 implicit-GEMM weight gradient
(conv_wgrad_kernel<128,128,0,*>): error of both against float64 on small cases (run with DIAGAN_WGRAD_X3_MIN_MAC=0 so that
the small cases qualify), then launch times on the StyleGAN2 256 x 256, batch 32 shapes.
    DIAGAN_WGRAD_X3_MIN_MAC=0 python tools/probe/wgrad_x3_time.py [--reps 5]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def ref64(x, dy, Co, Ci, R, S, st, pd):
    """d(loss)/dW for loss = <conv(x, W), dy>, float64 on the host, in the packed layout [Co][(r, s, c)]"""
    xd = x.double().permute(0, 3, 1, 2).cpu()
    gd = dy.double().permute(0, 3, 1, 2).cpu()
    w = torch.zeros(Co, Ci, R, S, dtype=torch.float64, requires_grad=True)
    (F.conv2d(xd, w, stride=st, padding=pd) * gd).sum().backward()
    return w.grad.permute(0, 2, 3, 1).reshape(Co, R * S * Ci)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--no_time", action="store_true")
    a = ap.parse_args()
    from diagan.ops import conv as K
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)

    def make(B, H, W, Ci, Co, R, S, st, pd):
        geom = K.Geom('conv', Ci, Co, R, S, st, pd)
        Ho, Wo = geom.out_hw(H, W)
        x = torch.randn(B, H, W, Ci, device=dev, generator=g)
        dy = torch.randn(B, Ho, Wo, Co, device=dev, generator=g)
        return geom, x, dy

    print("== error against float64 (max |err| / max |ref|) ==")
    bad = 0
    for case in [(2, 33, 33, 128, 128, 3, 3, 2, 0), (3, 20, 24, 64, 128, 2, 2, 1, 1), (2, 16, 16, 128, 256, 1, 1, 1, 0),
                 (2, 19, 21, 64, 128, 2, 1, 1, 1), (2, 19, 21, 64, 128, 1, 2, 1, 1), (1, 5, 5, 128, 128, 1, 1, 1, 0),
                 (5, 3, 3, 32, 128, 2, 2, 1, 1), (2, 31, 17, 128, 128, 3, 3, 2, 1), (3, 9, 40, 32, 256, 2, 2, 1, 0)]:
        B, H, W, Ci, Co, R, S, st, pd = case
        geom, x, dy = make(*case)
        ref = ref64(x, dy, Co, Ci, R, S, st, pd)
        sc = ref.abs().max().item()
        row = []
        for on in (False, True):
            K.set_wgrad_x3(on)
            grad = torch.zeros(Co, geom.Kp, device=dev)
            K.conv_wgrad(geom, dy, x, grad, False)
            used = K.wgrad_uses_x3(geom, B, H, W, dy.shape[1], dy.shape[2])
            row.append(((grad.double().cpu() - ref).abs().max().item() / sc, used))
        flag = "" if (row[1][1] and row[1][0] < 4 * max(row[0][0], 1e-7)) else "   <-- CHECK"
        bad += 1 if flag else 0
        print(f"  {R}x{S} s{st} p{pd} {Ci}->{Co} on {B}x{H}x{W}: fp32 {row[0][0]:.2e}   split-operand {row[1][0]:.2e} (taken: {row[1][1]}){flag}")
    print("  cases to check:", bad)
    if a.no_time:
        return
    shapes = [(32, 257, 257, 128, 256, 3, 3, 2, 0), (32, 129, 129, 256, 512, 3, 3, 2, 0), (32, 65, 65, 512, 512, 3, 3, 2, 0),
              (32, 33, 33, 512, 512, 3, 3, 2, 0),
              (32, 128, 128, 256, 128, 2, 2, 1, 1), (32, 64, 64, 512, 256, 2, 2, 1, 1), (32, 32, 32, 512, 512, 2, 2, 1, 1),
              (32, 128, 128, 256, 128, 2, 1, 1, 1), (32, 64, 64, 512, 256, 1, 2, 1, 1),
              (32, 128, 128, 128, 256, 1, 1, 1, 0), (32, 64, 64, 256, 512, 1, 1, 1, 0), (32, 128, 128, 256, 128, 1, 1, 1, 0)]
    print("== launch times (us incl. the second-stage sum; TFLOP/s in direct-convolution FLOP) ==")
    for case in shapes:
        B, H, W, Ci, Co, R, S, st, pd = case
        geom, x, dy = make(*case)
        Ho, Wo = dy.shape[1], dy.shape[2]
        flop = 2.0 * B * Ho * Wo * Co * R * S * Ci
        grad = torch.zeros(Co, geom.Kp, device=dev)
        row = []
        for on in (False, True):
            K.set_wgrad_x3(on)
            for _ in range(2):
                K.conv_wgrad(geom, dy, x, grad, False)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                K.conv_wgrad(geom, dy, x, grad, False)
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) / a.reps * 1e3
            row.append((us, flop / us / 1e6))
        print(f"  {R}x{S} s{st} p{pd} {Ci:4d}->{Co:4d} on {B}x{H}x{W}: fp32 {row[0][0]:8.1f} us {row[0][1]:6.1f}   "
              f"split-operand {row[1][0]:8.1f} us {row[1][1]:6.1f}   x{row[0][0] / row[1][0]:.2f}  "
              f"(splits {K.wgrad_splits_geom(geom, B, H, W, Ho, Wo)})")
    K.set_wgrad_x3(None)


if __name__ == "__main__":
    main()
