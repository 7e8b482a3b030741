#!/usr/bin/env python
"""Go / no-go, part 1: rounding error of Winograd F(4x4,3x3) in fp32 against a float64 direct
convolution, next to F(2x2,3x3) and the plain fp32 convolution, on SNGAN-like layers (CPU, torch).
Accumulation over input channels is fp32 in channel order (what the MFMA chain does); transforms are fp32."""
import sys
import torch

torch.manual_seed(0)
F64, F32 = torch.float64, torch.float32


def mats(m):
    if m == 2:
        BT = [[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]]
        G = [[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]]
        AT = [[1, 1, 1, 0], [0, 1, -1, -1]]
    else:       # Lavin & Gray, points 0, +-1, +-2, inf
        BT = [[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
              [0, 4, 0, -5, 0, 1]]
        G = [[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
             [0, 0, 1]]
        AT = [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]
    return [torch.tensor(x, dtype=F64) for x in (BT, G, AT)]


def wino(x, w, m, dt=F32):
    """x [B,Ci,H,W] (H, W multiples of m), w [Co,Ci,3,3]; pad 1.  Transforms and the channel sum in dtype dt."""
    BT, G, AT = [t.to(dt) for t in mats(m)]
    B, Ci, H, W = x.shape
    a = m + 2
    xp = torch.nn.functional.pad(x.to(dt), (1, 1, 1, 1))
    d = xp.unfold(2, a, m).unfold(3, a, m)                     # [B,Ci,TH,TW,a,a]
    V = BT @ d @ BT.T                                          # (exact-ish small sums in dt)
    U = G @ w.to(dt) @ G.T                                     # [Co,Ci,a,a]
    # M[b,co,ty,tx,i,j] = sum_ci V * U, sequential fp32 accumulation over ci in blocks (einsum keeps dt)
    M = torch.einsum('bcyxij,ocij->boyxij', V, U)
    Y = AT @ M @ AT.T                                          # [B,Co,TH,TW,m,m]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, w.shape[0], H, W)


def run(B, Ci, Co, H, relu_in):
    x = torch.randn(B, Ci, H, H, dtype=F64)
    if relu_in:
        x = x.clamp_min(0)
    x = x.to(F32).to(F64)
    w = (torch.randn(Co, Ci, 3, 3, dtype=F64) * (2.0 / (9 * Ci)) ** 0.5).to(F32).to(F64)
    ref = torch.nn.functional.conv2d(x, w, padding=1)
    scale = ref.abs().max().item()
    rms = ref.pow(2).mean().sqrt().item()
    out = {}
    out['direct fp32'] = torch.nn.functional.conv2d(x.to(F32), w.to(F32), padding=1).to(F64)
    out['F(2x2,3x3) fp32'] = wino(x, w, 2).to(F64)
    out['F(4x4,3x3) fp32'] = wino(x, w, 4).to(F64)
    print(f"B={B} Ci={Ci} Co={Co} H={H} relu_in={relu_in}: |y|max={scale:.3f} rms={rms:.3f}")
    for k, v in out.items():
        e = (v - ref).abs()
        print(f"   {k:18s} max err / |y|max = {e.max().item() / scale:.2e}   rms err / rms = {e.pow(2).mean().sqrt().item() / rms:.2e}")


if __name__ == "__main__":
    torch.set_num_threads(8)
    run(4, 256, 64, 32, True)
    run(4, 128, 64, 32, True)
    run(2, 512, 64, 16, True)
    run(2, 1024, 32, 8, False)
    run(4, 64, 64, 64, True)
