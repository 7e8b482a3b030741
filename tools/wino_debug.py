#!/usr/bin/env python
"""Diagnostic: the first LDS stage (V planes, U planes) of workgroup 0 of the Winograd kernel against a host model."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import numpy as np, torch
from diagan import _native as nat
from diagan.ops import conv as C
torch.manual_seed(0)
B, H, W, Ci, Co = 4, 8, 8, 16, 64
geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
x = torch.randn(B, H, W, Ci, device="cuda")
wp = torch.zeros(Co, geom.Kp, device="cuda"); wp[:, :9 * Ci] = torch.randn(Co, 9 * Ci, device="cuda")
buf = torch.zeros(1 << 16, dtype=torch.float32, device="cuda")
nat.call("diagan_conv_gemm_set_stamp_buffer", buf.data_ptr(), 1 << 20)
try:
    C.conv_fwd(geom, x, wp, tile_cfg=9)
    torch.cuda.synchronize()
finally:
    nat.call("diagan_conv_gemm_set_stamp_buffer", None, 0)
lds = buf[:16384].cpu().numpy()
print("nonzero in dump", int((buf != 0).sum().item()))
dbg = buf[16384:16384 + 64 * 16].cpu().numpy().reshape(64, 16)
for t in (0, 1, 2, 3, 4, 9):
    r = dbg[t]
    print("tid", t, "marker", r[0], "lt lq lr", r[1:4], "off0 %d inv0 %x" % (r[4:5].view(np.uint32)[0], r[5:6].view(np.uint32)[0]),
          "ra1", r[6], "MT t0 nk Ci TW TH", r[7:13], "off1 %d inv1 %x" % (r[13:14].view(np.uint32)[0], r[14:15].view(np.uint32)[0]), "vslot", r[15])
print("x[0,0,0,:4]", x[0, 0, 0, :4].tolist(), "x[0,0,1,:4]", x[0, 0, 1, :4].tolist())
V, U = lds[:8192].reshape(32, 64, 4), lds[8192:].reshape(32, 64, 4)
xn, wn = x.cpu().numpy(), wp.cpu().numpy()[:, :9 * Ci].reshape(Co, 3, 3, Ci)
Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
TW, TH = W // 2, H // 2
Vexp = np.zeros((32, 64, 4)); Uexp = np.zeros((32, 64, 4))
xp = np.pad(xn, ((0, 0), (1, 1), (1, 1), (0, 0)))
for t in range(64):
    b, ty, tx = t // (TH * TW), (t // TW) % TH, t % TW
    d = xp[b, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4, :8].astype(np.float64)      # [4][4][8]
    v = np.einsum('ir,rck,jc->ijk', Bt, d, Bt)
    for i in range(4):
        for j in range(4):
            for q in range(2):
                p = (i * 4 + j) * 2 + q
                Vexp[p, t ^ (q | (i << 1))] = v[i, j, 4 * q:4 * q + 4]
for co in range(64):
    g = wn[co, :, :, :8].astype(np.float64)                                     # [3][3][8]
    u = np.einsum('ir,rsk,js->ijk', G, g, G)
    for i in range(4):
        for j in range(4):
            for q in range(2):
                Uexp[(i * 4 + j) * 2 + q, co] = u[i, j, 4 * q:4 * q + 4]
print("V max err", np.abs(V - Vexp).max(), "U max err", np.abs(U - Uexp).max())
bad = np.argwhere(np.abs(V - Vexp) > 1e-4)
print("bad V entries", len(bad), bad[:12].tolist())
badu = np.argwhere(np.abs(U - Uexp) > 1e-4)
print("bad U entries", len(badu), badu[:12].tolist())
if len(badu):
    p, c, e = badu[0]; print("U got", U[p, c], "exp", Uexp[p, c])
    for pp in range(32):
        for cc in range(64):
            if np.allclose(U[pp, cc], Uexp[p, c], atol=1e-5): print("  found expected U", (p, c), "at", (pp, cc))
if len(bad):
    p, t, e = bad[0]; print("V got", V[p, t], "exp", Vexp[p, t])
    for pp in range(32):
        for tt in range(64):
            if np.allclose(V[pp, tt], Vexp[p, t], atol=1e-5): print("  found expected V", (p, t), "at", (pp, tt))
