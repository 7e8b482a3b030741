#!/usr/bin/env python
"""Diagnostic (GPU box; library built with tools/build_variant.sh stamp conv_gemm_x3b.hip "-DXB_STAMP", run with
DIAGAN_LIB_PATH=gpurun_variants/libdiagan_stamp.so): where a K-step of the 128 x 128 split-operand implicit GEMM goes --
per-wave s_memtime deltas summed over the K loop, medians over all waves."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

PHASES = ["issue of the next step's global loads", "MFMA phase (48 MFMAs, 40 fragment reads)", "barrier behind the MFMAs",
          "wait for the global loads", "split + 12 LDS writes (drained)", "barrier behind the writes"]


def main():
    from diagan import _native as nat
    from diagan.ops import conv as K
    dev = torch.device("cuda", 0)
    slots = 1 << 22
    buf = torch.zeros(slots, dtype=torch.int64, device=dev)
    for (B, H, W, Ci, Co, R, S, st, pd) in [(32, 257, 257, 128, 256, 3, 3, 2, 0), (32, 64, 64, 512, 256, 2, 2, 1, 1), (32, 64, 64, 512, 256, 1, 1, 1, 0)]:
        x = torch.randn(B, H, W, Ci, device=dev)
        geom = K.Geom('conv', Ci, Co, R, S, st, pd)
        wp = torch.randn(Co, geom.Kp, device=dev) * (R * S * Ci) ** -0.5
        Ho, Wo = geom.out_hw(H, W)
        out = torch.empty((B, Ho, Wo, Co), device=dev)
        for _ in range(3):
            K.conv_fwd(geom, x, wp, tile_cfg=17, wino=False, out=out)
        torch.cuda.synchronize()
        nwg = ((B * Ho * Wo + 127) // 128) * ((Co + 127) // 128)
        nk = R * S * Ci // 32
        buf.zero_()
        nat.call("diagan_conv_gemm_set_stamp_buffer", buf.data_ptr(), slots // 16)
        K.conv_fwd(geom, x, wp, tile_cfg=17, wino=False, out=out)
        torch.cuda.synchronize()
        nat.call("diagan_conv_gemm_set_stamp_buffer", None, 0)
        if os.environ.get("DIAGAN_GEMM_X3B_FORM") == "2":
            nwg = ((B * Ho * Wo + 255) // 256) * ((Co + 127) // 128)
            nk = 2 * nk                                   # sub-steps (K-step, 128-row half)
            t = buf[: nwg * 64].cpu().numpy().reshape(nwg, 8, 8).astype(np.float64)
            print(f"{R}x{S} s{st} {Ci}->{Co} on {B}x{H}x{W}: {nwg} workgroups, {nk} sub-steps; producer / consumer form, cycles per sub-step (median, p10, p90)")
            for who, sl, names in (("consumers", slice(0, 4), {0: "fragment reads + 48 MFMAs", 5: "barrier"}),
                                   ("producers", slice(4, 8), {0: "DMA issue (4 x 1 KB)", 1: "issue of 4 loads", 2: "wait for the loads of this step",
                                                               3: "split + 12 LDS writes (drained)", 4: "wait for the DMA", 5: "barrier"})):
                tot = 0.0
                for i, nm in names.items():
                    v = t[:, sl, i].reshape(-1) / nk
                    tot += np.median(v)
                    print(f"   {who:10s} {nm:40s} {np.median(v):8.0f} {np.percentile(v, 10):8.0f} {np.percentile(v, 90):8.0f}")
                print(f"   {who:10s} {'sum':40s} {tot:8.0f}")
            print(f"   epilogue {np.median(t[:, 0:4, 7]):8.0f} cycles")
            continue
        t = buf[: nwg * 32].cpu().numpy().reshape(nwg, 4, 8).astype(np.float64)
        print(f"{R}x{S} s{st} {Ci}->{Co} on {B}x{H}x{W}: {nwg} workgroups x 4 waves, {nk} K-steps; cycles per K-step (median, p10, p90 over waves)")
        tot = 0.0
        for i, ph in enumerate(PHASES):
            v = t[:, :, i].reshape(-1) / nk
            tot += np.median(v)
            print(f"   {ph:50s} {np.median(v):8.0f} {np.percentile(v, 10):8.0f} {np.percentile(v, 90):8.0f}")
        print(f"   {'sum / whole loop per step':50s} {tot:8.0f} / {np.median(t[:, :, 6]) / nk:8.0f};  epilogue {np.median(t[:, :, 7]):8.0f} cycles")


if __name__ == "__main__":
    main()
