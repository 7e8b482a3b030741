#!/usr/bin/env python
"""F(4x4,3x3) kernel: launch time against the number of K-steps (input channels) at a fixed output, normal and with the
epilogue switched off (diagnostic build): slope = cost of a K-step, intercept = fixed cost per workgroup (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from diagan.ops import conv as C
from diagan import _native as nat

def timeit(f, iters=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

dev = "cuda"
ablate = "--ablate" in sys.argv
modes = {"full": 0}
if ablate:
    modes.update({"no epilogue": 512, "no stores": 1024, "no transform/loads": 48, "no transform/loads/epilogue": 48 | 512})
for cfg in (13, 9):
    for B, H, W, Co in ((384, 32, 32, 256), (128, 32, 32, 128)):
        for name, bits in (modes.items() if cfg == 13 else [("full", 0)]):
            if ablate:
                nat.call("diagan_conv_gemm_tune", 0, bits, 0)
            ts = {}
            for Ci in (64, 128, 256, 512):
                geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
                x = torch.randn(B, H, W, Ci, device=dev)
                wp = torch.randn(Co, geom.Kp, device=dev) * (9 * Ci) ** -0.5
                ts[Ci] = timeit(lambda: C.conv_fwd(geom, x, wp, tile_cfg=cfg))
            rows = 512 if cfg == 13 else 256
            rounds = (B * H * W // rows) * (Co // 64) / 256
            slope = (ts[512] - ts[128]) / (384 / 8) / rounds          # seconds per K-step of 8 channels per round
            icpt = ts[128] / rounds - slope * 16
            mf = 4608 if cfg == 13 else 4096   # MFMA issue cycles per SIMD and K-step: 2 waves x (36 | 32) instructions x 64 cycles
            print(f"cfg {cfg} B={B} {H}x{W} Co={Co} {name:28s} " + " ".join(f"Ci={c}: {t*1e6:7.1f} us" for c, t in ts.items()) +
                  f" | per K-step {slope*1e6:6.3f} us = {slope*2.2e9:6.0f} cyc (MFMA {mf:.0f}), fixed per workgroup round {icpt*1e6:6.2f} us", flush=True)
if ablate:
    nat.call("diagan_conv_gemm_tune", 0, -1, 0)
