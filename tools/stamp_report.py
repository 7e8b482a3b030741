#!/usr/bin/env python
"""Where a conv GEMM launch spends its time (GPU box): per-workgroup s_memtime / s_memrealtime stamps of the
diagnostic (STAMP) build of conv_gemm_kernel, see diagan_conv_gemm_set_stamp_buffer in include/diagan_hip.h.

For every shape: launch wall time (HIP events, production kernel), then from the stamped build the distribution over
workgroups of: start offset after the first workgroup's start, loader set-up, first tile (loads + LDS stores + barrier),
K loop (total and per K-step), epilogue issue, store drain, and the end offset of the last workgroup; plus how many
workgroups each CU received.  Shares, not lengths, are what the stamped build is good for (its fences forbid overlaps
the production kernel has)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import numpy as np
import torch
from diagan import _native as nat
from diagan.ops import conv as C

SHAPES = [
    # name, B, H, W, Ci, Co, R, pro, variants (tile cfg, tune flags, LDS delta)
    ("D32.b3.c1.pair  M=8192 N=128 K=1152", 128, 8, 8, 128, 128, 3, 1, ((3, 0, 0), (7, 0, 0), (7, 3, 0))),
    ("D64.b1.c2.pair  M=524288 N=64 K=576", 128, 64, 64, 64, 64, 3, 1, ((3, 0, 0), (3, 3, 0), (7, 3, 0), (3, 0, 8192), (8, 3, 0), (5, 3, 0))),
    ("G64.b3.c1       M=16384 N=256 K=4608", 64, 16, 16, 512, 256, 3, 0, ((3, 0, 0), (7, 3, 0), (1, 0, 0))),
    ("G32.b4.c1       M=65536 N=256 K=2304", 64, 32, 32, 256, 256, 3, 0, ((1, 0, 0), (1, 3, 0), (7, 3, 0))),
]


def concurrency(cu, r0, r1):
    """per CU: maximum and time-average number of workgroups resident together (sweep over entry / exit stamps)"""
    mx, avg = [], []
    for c in np.unique(cu):
        m = cu == c
        ev = sorted([(t, 1) for t in r0[m]] + [(t, -1) for t in r1[m]])
        cur = best = 0
        area, last = 0.0, ev[0][0]
        for t, d in ev:
            area += cur * (t - last)
            last = t
            cur += d
            best = max(best, cur)
        mx.append(best)
        avg.append(area / max(ev[-1][0] - ev[0][0], 1))
    return np.array(mx), np.array(avg)


def q(a):
    a = np.sort(np.asarray(a, dtype=np.float64))
    return f"min {a[0]:9.0f}  med {a[len(a)//2]:9.0f}  p90 {a[int(len(a)*0.9)]:9.0f}  max {a[-1]:9.0f}"


def main():
    slots = 1 << 17
    buf = torch.zeros(slots * 8, dtype=torch.int64, device="cuda")
    for name, B, H, W, Ci, Co, R, pro, cfgs in SHAPES:
        geom = C.Geom("conv", Ci, Co, R, R, 1, R // 2)
        x = torch.randn(B, H, W, Ci, device="cuda")
        wp = torch.randn(Co, geom.Kp, device="cuda") * 0.05
        y = torch.empty(B, H, W, Co, device="cuda")
        M = B * H * W
        flop = 2.0 * M * Co * R * R * Ci
        prot = (pro, None, None) if pro else None
        for cfg, flags, ldsd in cfgs:
            nat.call("diagan_conv_gemm_tune", 0, flags, ldsd)
            bm, bn = nat.fn("diagan_conv_gemm_tile_rows")(cfg), nat.fn("diagan_conv_gemm_tile_cols")(cfg)
            f = lambda: C.conv_fwd(geom, x, wp, out=y, tile_cfg=cfg, pro=prot)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                f()
            e.record()
            torch.cuda.synchronize()
            wall = s.elapsed_time(e) / 20 * 1e3
            ks = nat.fn("diagan_conv_gemm_pick_ksplit")(M, Co, geom.Kp, cfg)
            nwg = -(-M // bm) * -(-Co // bn) * ks
            nk = geom.Kp // 32 // ks
            buf.zero_()
            nat.call("diagan_conv_gemm_set_stamp_buffer", buf.data_ptr(), slots)
            try:
                for _ in range(3):          # the last launch's stamps are read (caches as warm as in the timed loop)
                    f()
                torch.cuda.synchronize()
            finally:
                nat.call("diagan_conv_gemm_set_stamp_buffer", None, 0)
            st = buf.view(-1, 8)[:nwg].cpu().numpy().astype(np.uint64)
            r0, r1 = st[:, 0].astype(np.float64), st[:, 1].astype(np.float64)
            t = st[:, 2:7].astype(np.float64)
            hw = st[:, 7]
            cu = ((hw >> np.uint64(32)) & np.uint64(0xf)) * np.uint64(1024) + ((hw >> np.uint64(8)) & np.uint64(0xf)) \
                + (((hw >> np.uint64(13)) & np.uint64(0x7)) << np.uint64(4)) + (((hw >> np.uint64(12)) & np.uint64(1)) << np.uint64(7))
            per_cu = np.unique(cu, return_counts=True)[1]
            span = (r1.max() - r0.min()) * 10.0          # ns (100 MHz counter)
            clk = t.sum(1) / np.maximum((r1 - r0) * 10.0, 1.0)      # cycles per ns of the workgroup's life = GHz
            cmax, cavg = concurrency(cu, r0, r1)
            np.save(os.path.join(ROOT, "gpurun_out", f"stamps_{name.split()[0]}_cfg{cfg}_fl{flags}_lds{ldsd}.npy"), st)
            print(f"== {name}  cfg{cfg} ({bm}x{bn}) tune flags {flags} LDS delta {ldsd} ksplit {ks}: {nwg} workgroups, {nk} K-steps each; wall {wall:.1f} us "
                  f"= {flop/wall/1e6:.1f} TF; stamped launch first-start -> last-end {span/1e3:.1f} us; "
                  f"clock ~{np.median(clk):.2f} GHz; CUs used {len(per_cu)}, workgroups per CU {per_cu.min()}..{per_cu.max()}")
            print(f"   workgroups resident together on a CU: max {cmax.min()}..{cmax.max()}, time-average {cavg.mean():.2f}")
            print(f"   start after first start [ns] {q((r0 - r0.min()) * 10)}")
            print(f"   end before last end     [ns] {q((r1.max() - r1) * 10)}")
            print(f"   life of a workgroup     [ns] {q((r1 - r0) * 10)}")
            for i, lab in enumerate(("loader set-up", "first tile", "K loop", "epilogue issue", "store drain")):
                print(f"   {lab:15s} [cycles] {q(t[:, i])}   share of life {np.median(t[:, i] / t.sum(1)):6.1%}")
            print(f"   K loop per K-step [cycles] {q(t[:, 2] / max(nk, 1))}  (MFMA-bound, one wave per SIMD: {16 * 64 * (bm * bn // 4096)})")
            sys.stdout.flush()
    nat.call("diagan_conv_gemm_tune", 0, -1, 0)


if __name__ == "__main__":
    main()
