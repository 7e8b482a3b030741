"""Split-K / tile sweep on the small D-block shapes (GPU box).  One process per DIAGAN_KSPLIT value."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
    import torch
    from diagan.ops import conv as C
    from bench_conv import timeit
    for (B, H, Ci, Co) in ((128, 8, 128, 128), (64, 8, 256, 256), (128, 16, 128, 128)):
        geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
        x = torch.randn(B, H, H, Ci, device="cuda"); wp = torch.randn(Co, geom.Kp, device="cuda") * 0.05
        y = torch.empty(B, H, H, Co, device="cuda")
        flop = 2.0 * B * H * H * Co * 9 * Ci
        for cfg in (1, 3):
            t = timeit(lambda: C.conv_fwd(geom, x, wp, out=y, tile_cfg=cfg, pro=(1, None, None)), iters=50)
            print(f"ksplit={os.environ.get('DIAGAN_KSPLIT','auto'):4s} cfg{cfg} M={B*H*H} N={Co} K={9*Ci}: {t*1e6:7.1f} us {flop/t/1e12:6.1f} TF", flush=True)
else:
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    for ks in ("auto", "1", "2", "3", "4", "6", "9", "12"):
        env = dict(os.environ)
        if ks != "auto":
            env["DIAGAN_KSPLIT"] = ks
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, cwd=os.path.join(ROOT, "tools"))
