#!/usr/bin/env python
"""Round 6: what the split-operand weight gradient would be worth on the SNGAN discriminators' pooled layers if their box sums were
materialised (plain stride-2 launches on the (H+1) x (W+1) grid): an upper bound for a box-sum loader in that kernel -- 1.3-1.55x on
128 -> 128 at 33 x 33 / 17 x 17, i.e. <= 0.2 ms of SNGAN-32's 12 ms step; not built (the batched launches, their prologues and bias
columns stay on the fp32 kernels).
    python tools/probe/wgrad_x3_sngan_shapes.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "self-diagnosing-gan_amd"))
import torch
from diagan.ops import conv as K
dev = torch.device("cuda", 0)
def timed(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (B, H, Ci, Co) in [(128, 33, 128, 128), (128, 17, 128, 128), (128, 65, 64, 128), (128, 33, 256, 256), (128, 65, 128, 128)]:
    geom = K.Geom('conv', Ci, Co, 3, 3, 2, 0)
    Ho = (H - 3) // 2 + 1
    x = torch.randn(B, H, H, Ci, device=dev); dy = torch.randn(B, Ho, Ho, Co, device=dev)
    grad = torch.zeros(Co, geom.Kp, device=dev)
    row = []
    for on in (0, 2):
        K.set_wgrad_x3(on)
        row.append(timed(lambda: K.conv_wgrad(geom, dy, x, grad, False)))
    K.set_wgrad_x3(None)
    fl = 2.0 * B * Ho * Ho * Co * 9 * Ci
    print(f"B{B} {H}x{H} {Ci}->{Co}: fp32 {row[0]:7.1f} us {fl/row[0]/1e6:6.1f} TF   x3 {row[1]:7.1f} us {fl/row[1]/1e6:6.1f} TF  x{row[0]/row[1]:.2f}  splits {K.wgrad_splits_geom(geom, B, H, H, Ho, Ho)}")
