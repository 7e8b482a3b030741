#!/usr/bin/env python
"""Diagnostic (GPU box; library built with -DDIAGAN_W4_STAMP: tools/build_variant.sh stamp conv_wino4.hip "-DDIAGAN_W4_STAMP",
run with DIAGAN_LIB_PATH=gpurun_variants/libdiagan_stamp.so): where a K-step of the F(4x4) kernel goes -- per-wave s_memtime
deltas of the step's phases summed over the K loop, median / p10 / p90 over all waves of all workgroups, in shader cycles per
K-step (MFMA-bound: 4608 per SIMD = 2 waves x 36 x 64)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import numpy as np
import torch
from diagan import _native as nat
from diagan.ops import conv as C

PHASES = ["wait: first weight unit", "slots 0..ROW_AT (frag reads, load issue, MFMAs)", "wait: input loads", "row pass (+ drain)",
          "slots ROW_AT+1..COL_AT", "column pass (reads, math, writes drained)", "slots COL_AT+1..NS-1", "LDS drain at the end",
          "barrier", "(loop overhead behind the barrier)"]


def main():
    dev = "cuda"
    slots = 1 << 19
    buf = torch.zeros(slots * 8, dtype=torch.int64, device=dev)
    cases = [("plain", 13, dict()), ("bn+relu", 13, dict(pro=True)), ("upin bn+relu", 15, dict(pro=True, up=True))]
    for B, H, W, Ci, Co in ((64, 32, 32, 256, 256), (384, 32, 32, 256, 256)):
        geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
        wp = torch.randn(Co, geom.Kp, device=dev) * (9 * Ci) ** -0.5
        sc, sh = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.3
        for name, cfg, kw in cases:
            up = kw.get('up', False)
            x = torch.randn(B, H // 2 if up else H, W // 2 if up else W, Ci, device=dev)
            pro = (C.PRO_AFFINE_RELU, sc, sh) if kw.get('pro') else None
            f = (lambda: C.conv_fwd(geom, x, wp, pro=pro, up_in=True)) if up else (lambda: C.conv_fwd(geom, x, wp, pro=pro, tile_cfg=cfg))
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            nwg = (B * H * W // 512) * (Co // 64)
            nk = Ci // 8
            buf.zero_()
            nat.call("diagan_conv_gemm_set_stamp_buffer", buf.data_ptr(), slots)
            f()
            torch.cuda.synchronize()
            nat.call("diagan_conv_gemm_set_stamp_buffer", None, 0)
            t = buf[: nwg * 128].cpu().numpy().reshape(nwg, 8, 16).astype(np.float64)
            assert (t[:, :, 13] == nk).all(), "stamp slots do not line up with the launch"
            print(f"B={B} {H}x{W} {Ci}->{Co} {name}: {nwg} workgroups x 8 waves, {nk} K-steps; cycles per K-step")
            tot = 0.0
            for i, ph in enumerate(PHASES):
                v = t[:, :, i].reshape(-1) / nk
                tot += np.median(v)
                print(f"   {ph:50s} median {np.median(v):8.0f}   p10 {np.percentile(v, 10):8.0f}   p90 {np.percentile(v, 90):8.0f}")
            loop = t[:, :, 12].reshape(-1) / nk
            print(f"   {'sum of medians / K loop per step (median)':50s} {tot:8.0f} / {np.median(loop):8.0f}")
            print(f"   set-up + first stage {np.median(t[:, :, 14]):8.0f} cycles, K loop {np.median(t[:, :, 12]):8.0f}, "
                  f"epilogue {np.median(t[:, :, 15]):8.0f}  (per workgroup; medians over waves)")
            te = buf[nwg * 128: 2 * nwg * 128].cpu().numpy().reshape(nwg, 8, 16).astype(np.float64)
            names = ["scalar set-up", "p0: products parked", "p0: barrier", "p0: reads, transform, stores issued", "p0: barrier",
                     "p1: products parked", "p1: barrier", "p1: reads, transform, stores issued", "p1: barrier", "(p1 loop overhead)"]
            order = [0, 1, 2, 3, 4, 9, 5, 6, 7, 8]
            print("   epilogue phases (cycles; waves 0-3 own column half 0 = phase p0, waves 4-7 phase p1):")
            for i in order:
                print(f"      {names[i]:40s} waves 0-3 {np.median(te[:, :4, i]):8.0f}   waves 4-7 {np.median(te[:, 4:, i]):8.0f}")
            # waves 0-3 against their SIMD partners 4-7
            for i in range(len(PHASES)):
                a, b = t[:, :4, i].reshape(-1) / nk, t[:, 4:, i].reshape(-1) / nk
                print(f"   {PHASES[i]:50s} waves 0-3 {np.median(a):8.0f}   waves 4-7 {np.median(b):8.0f}")


if __name__ == "__main__":
    main()
