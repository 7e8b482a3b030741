#!/usr/bin/env python
"""bf16x6 mode of the conv GEMM (diagan_set_mfma_mode(1)) beside the exact fp32 MFMA: error against float64 on a
small convolution, and time / TFLOP/s on the SNGAN shapes (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
import torch.nn.functional as F
from diagan.ops import conv as C

def mode(m):
    C.set_mfma_mode(m)


def run(geom, x, w, pro=None, cfg=0):
    return C.conv_fwd(geom, x, w, pro=pro, tile_cfg=cfg)


dev = torch.device("cuda")
torch.manual_seed(0)
# ---- accuracy -------------------------------------------------------------------------------------------------
for (B, H, Ci, Co, k, cfg) in ((4, 16, 64, 128, 3, 1), (4, 16, 64, 128, 3, 3), (2, 8, 256, 256, 3, 1), (8, 8, 32, 64, 1, 3)):
    x = torch.randn(B, H, H, Ci)
    w = torch.randn(Co, Ci, k, k) / (Ci * k * k) ** 0.5
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=k // 2).permute(0, 2, 3, 1)
    geom = C.Geom('conv', Ci, Co, k, k, 1, k // 2)
    wp = C.pack_oihw(w, geom.Kp).to(dev)
    errs = []
    for m in (0, 1):
        mode(m)
        y = run(geom, x.to(dev).contiguous(), wp, cfg=cfg).cpu().double()
        errs.append(float((y - ref).abs().mean() / ref.abs().mean()))
    print(f"B={B} H={H} Ci={Ci} Co={Co} k={k} tile={'128' if cfg == 1 else '64'}: mean|err|/mean|y|  fp32 MFMA {errs[0]:.2e}   bf16x6 {errs[1]:.2e}")
# prologue path (BN affine + ReLU) through the split
x = torch.randn(4, 16, 16, 64); w = torch.randn(128, 64, 3, 3) / 24.0
sc, sh = torch.rand(64) + 0.5, torch.randn(64) * 0.1
xa = torch.relu(x * sc + sh)
ref = F.conv2d(xa.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
geom = C.Geom('conv', 64, 128, 3, 3, 1, 1)
wp = C.pack_oihw(w, geom.Kp).to(dev)
for m in (0, 1):
    mode(m)
    y = run(geom, x.to(dev), wp, pro=(C.PRO_AFFINE_RELU, sc.to(dev), sh.to(dev))).cpu().double()
    print(f"affine+relu prologue, mode {m}: {float((y - ref).abs().mean() / ref.abs().mean()):.2e}")
# ---- speed ----------------------------------------------------------------------------------------------------
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

for (B, H, Ci, Co, k) in ((64, 32, 256, 256, 3), (320, 32, 256, 256, 3), (128, 32, 128, 128, 3), (128, 16, 128, 128, 3),
                          (64, 64, 64, 64, 3), (32, 256, 128, 128, 3)):
    geom = C.Geom('conv', Ci, Co, k, k, 1, k // 2)
    x = torch.randn(B, H, H, Ci, device=dev)
    wp = torch.randn(Co, geom.Kp, device=dev) * 0.02
    out = torch.empty(B, H, H, Co, device=dev)
    flop = 2.0 * B * H * H * Co * k * k * Ci
    res = []
    for m in (0, 1):
        mode(m)
        t = timeit(lambda: C.conv_fwd(geom, x, wp, out=out))
        res.append((t, flop / t / 1e12))
    print(f"M={B*H*H:8d} N={Co:4d} K={k*k*Ci:5d}: fp32 MFMA {res[0][0]*1e6:8.1f} us {res[0][1]:6.1f} TF | bf16x6 {res[1][0]*1e6:8.1f} us {res[1][1]:6.1f} TF  ({res[0][0]/res[1][0]:.2f}x)")
mode(0)

# ---- weight gradient ------------------------------------------------------------------------------------------
print("weight gradient:")
for (B, H, Ci, Co, k, pro) in ((4, 16, 128, 128, 3, None), (2, 8, 256, 256, 3, 'bn'), (3, 12, 128, 256, 3, None), (4, 16, 128, 128, 1, None)):
    x = torch.randn(B, H, H, Ci)
    dy = torch.randn(B, H, H, Co)
    geom = C.Geom('conv', Ci, Co, k, k, 1, k // 2)
    prot, xa = None, x
    if pro == 'bn':
        sc, sh = torch.rand(Ci) + 0.5, torch.randn(Ci) * 0.1
        prot, xa = (C.PRO_AFFINE_RELU, sc.to(dev), sh.to(dev)), torch.relu(x * sc + sh)
    xr = xa.permute(0, 3, 1, 2).double().requires_grad_(False)
    wref = torch.zeros(Co, Ci, k, k, dtype=torch.float64, requires_grad=True)
    yy = F.conv2d(xr, wref, padding=k // 2)
    (yy * dy.permute(0, 3, 1, 2).double()).sum().backward()
    ref = C.pack_oihw(wref.grad, geom.Kp)
    errs = []
    for m in (0, 1):
        mode(m)
        grad = torch.zeros(Co, geom.Kp, device=dev)
        C.conv_wgrad(geom, dy.to(dev), x.to(dev), grad, accumulate=False, pro=prot)
        errs.append(float((grad.cpu().double() - ref).abs().mean() / ref.abs().mean()))
    print(f"B={B} H={H} Ci={Ci} Co={Co} k={k} pro={pro}: mean|err|/mean|g|  fp32 MFMA {errs[0]:.2e}   bf16x6 {errs[1]:.2e}")
for (B, H, Ci, Co, k) in ((128, 32, 128, 128, 3), (64, 32, 256, 256, 3), (64, 16, 256, 256, 3), (32, 256, 128, 128, 3), (32, 64, 512, 512, 3)):
    geom = C.Geom('conv', Ci, Co, k, k, 1, k // 2)
    x = torch.randn(B, H, H, Ci, device=dev)
    dy = torch.randn(B, H, H, Co, device=dev)
    grad = torch.zeros(Co, geom.Kp, device=dev)
    flop = 2.0 * B * H * H * Co * k * k * Ci
    res = []
    for m in (0, 1):
        mode(m)
        t = timeit(lambda: C.conv_wgrad(geom, dy, x, grad, accumulate=False), n=10)
        res.append((t, flop / t / 1e12))
    print(f"M={B*H*H:8d} N={Co:4d} K={k*k*Ci:5d}: fp32 MFMA {res[0][0]*1e6:8.1f} us {res[0][1]:6.1f} TF | bf16x6 {res[1][0]*1e6:8.1f} us {res[1][1]:6.1f} TF  ({res[0][0]/res[1][0]:.2f}x)")
mode(0)
