"""Cost of the fused prologue / epilogue variants of conv_gemm on the two big G-block shapes (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from diagan.ops import conv as C
def timeit(f, iters=20, warm=3):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
shapes = ((64, 32, 256, 256), (64, 16, 256, 256), (128, 32, 128, 128), (128, 8, 128, 128))
for (B, H, Ci, Co) in shapes:
    geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
    x = torch.randn(B, H, H, Ci, device="cuda"); wp = torch.randn(Co, geom.Kp, device="cuda") * 0.05
    y = torch.empty(B, H, H, Co, device="cuda"); res = torch.randn(B, H, H, Co, device="cuda")
    sc = torch.rand(Ci, device="cuda") + 0.5; sh = torch.randn(Ci, device="cuda")
    bias = torch.randn(Co, device="cuda")
    flop = 2.0 * B * H * H * Co * 9 * Ci
    for name, kw in (("pro0", {}), ("pro0+bias", dict(bias=bias)), ("pro0+res", dict(residual=res)),
                     ("pro0+stats", dict(want_stats=True)), ("pro1", dict(pro=(1, None, None))),
                     ("pro2", dict(pro=(2, sc, sh))), ("pro2+res+bias", dict(pro=(2, sc, sh), residual=res, bias=bias)),
                     ("pro2+res+bias+stats", dict(pro=(2, sc, sh), residual=res, bias=bias, want_stats=True))):
        t = timeit(lambda: C.conv_fwd(geom, x, wp, out=y, **kw))
        print(f"M={B*H*H} N={Co} K={9*Ci} {name:20s} {t*1e6:8.1f} us {flop/t/1e12:6.1f} TF", flush=True)
