import os, sys
sys.path.insert(0, "self-diagnosing-gan_amd")
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
    return s.elapsed_time(e) / iters * 1e3
dev="cuda"
for B,H,W,Ci,Co in [(384,32,32,256,256),(384,16,16,256,256),(384,64,64,64,64),(384,32,32,128,128),(128,32,32,128,128)]:
    geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
    x = torch.randn(B,H,W,Ci, device=dev)
    xb = x.view(B,H,W,Ci//8,8).permute(0,3,1,2,4).contiguous()       # [B][Ci/8][H][W][8]
    wp = torch.randn(Co, geom.Kp, device=dev) * (9*Ci)**-0.5
    sc, sh = torch.rand(Ci, device=dev)+0.5, torch.randn(Ci, device=dev)*0.3
    resh = torch.randn(B,H//2,W//2,Co, device=dev); bias = torch.randn(Co, device=dev)
    for name, kw in (("plain", {}), ("bn+res_up+stats", dict(bias=bias, residual=resh, res_up=True, pro=(C.PRO_AFFINE_RELU, sc, sh), want_stats=True))):
        def run(inp):
            r = C.conv_fwd(geom, inp, wp, tile_cfg=13, **kw)
            return r[0] if isinstance(r, tuple) else r
        ts0, ts1 = [], []
        for rep in range(4):                      # interleaved rounds in one process (cdna guide rule 24)
            nat.call("diagan_conv_gemm_tune", 0, -1, 0)
            ref = run(x); ts0.append(timeit(lambda: run(x)))
            nat.call("diagan_conv_gemm_tune", 0, 8192, 0)
            got = run(xb.view(B,H,W,Ci)); ts1.append(timeit(lambda: run(xb.view(B,H,W,Ci))))
        nat.call("diagan_conv_gemm_tune", 0, -1, 0)
        err = (got-ref).abs().max().item()/ref.abs().max().item()
        t0, t1 = min(ts0), min(ts1)
        print(f"B={B} {H}x{W} {Ci}->{Co} {name:16s} NHWC {t0:8.1f} us (rounds {' '.join(f'{t:.0f}' for t in ts0)}) | channel-blocked input {t1:8.1f} us ({' '.join(f'{t:.0f}' for t in ts1)}) {t0/t1:4.2f}x err {err:.1e}", flush=True)
