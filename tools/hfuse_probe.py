#!/usr/bin/env python
"""VERDICT r3 item 2 (go / no-go): how much could ONE grid holding a layer's data-gradient and weight-gradient launches gain
over two launches back to back?  Upper bound without writing the fused kernel: the same two launches issued on two HIP streams
with NO dependency between them (no fork / join per pair: N pairs are queued on each stream, one synchronisation at the end),
against the N pairs on one stream.  Shapes: the small maps of SNGAN-32's / SNGAN-64's discriminators in the paired
D(real)|D(fake) pass (batch 128), where conv_gemm_kernel<64,64,...> is fixed-cost bound (GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from diagan.ops import conv as C


def run(B, H, W, Ci, Co, N=200):
    dev = "cuda"
    geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
    x = torch.randn(B, H, W, Ci, device=dev)
    dy = torch.randn(B, H, W, Co, device=dev)
    wp = torch.randn(Co, geom.Kp, device=dev) * 0.03
    wd = torch.zeros(Ci, geom.Kd, device=dev)
    C.pack_weights(wp, Co, Ci, 9, geom.Kp, geom.Kd, Wd=wd)
    splits = C.wgrad_splits_geom(geom, B, H, W, H, W)
    stride = Co * geom.Kp
    slab = torch.empty(splits * stride, device=dev)
    dx = torch.empty(B, H, W, Ci, device=dev)
    msk = torch.randn(B, H, W, Ci, device=dev)

    def dgrad():
        C.conv_dgrad(geom, dy, wd, (H, W), mask_src=msk, out=dx)

    def wgrad():
        C.conv_wgrad_into(geom, dy, x, slab, splits, stride, -1, pro=(C.PRO_RELU, None, None))

    def timed(f):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / N * 1e6

    def serial():
        for _ in range(N):
            dgrad()
            wgrad()

    def only(g):
        def f():
            for _ in range(N):
                g()
        return f
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def two_streams():
        for _ in range(N):
            with torch.cuda.stream(s1):
                dgrad()
            with torch.cuda.stream(s2):
                wgrad()
    for f in (serial, two_streams):
        f()
    res = {}
    for rep in range(3):
        for name, f in (("dgrad", only(dgrad)), ("wgrad", only(wgrad)), ("serial", serial), ("two_streams", two_streams)):
            res.setdefault(name, []).append(timed(f))
    m = {k: min(v) for k, v in res.items()}
    print(f"B={B:3d} {H:2d}x{W:2d} Ci={Ci:4d} Co={Co:4d} splits={splits:3d} | dgrad {m['dgrad']:6.1f} us  wgrad {m['wgrad']:6.1f} us  "
          f"pair on one stream {m['serial']:6.1f} us  on two streams {m['two_streams']:6.1f} us  "
          f"({m['serial'] / m['two_streams']:4.2f}x, saves {m['serial'] - m['two_streams']:5.1f} us per pair)", flush=True)


if __name__ == "__main__":
    for shape in [(128, 8, 8, 128, 128), (128, 16, 16, 128, 128), (64, 8, 8, 256, 256), (64, 4, 4, 256, 256),
                  (128, 8, 8, 256, 256), (128, 4, 4, 512, 512), (128, 2, 2, 1024, 1024), (128, 16, 16, 128, 256)]:
        run(*shape)
