#!/usr/bin/env python
"""Go / no-go, part 1: rounding error of a Winograd F(3x3,4x4) WEIGHT gradient in fp32 against a float64
direct weight gradient, next to F(3x3,2x2) (what conv_wgrad_wino_kernel runs) and the plain fp32 gradient (CPU, torch).
By transposition of y = A^T[(G g G^T) . (B^T d B)]A:   dg = G^T [ sum_tiles (A dy A^T) . (B^T d B) ] G,
i.e. both operands are transformed per tile (dy with A, the input patch with B^T), the products are summed over ALL tiles of
the batch in the transform domain (fp32, tile order: what an MFMA accumulator chain does) and ONE inverse transform with G^T
follows per split."""
import sys
import torch

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from wino_f4_error import mats, F32, F64

torch.manual_seed(0)


def wino_wgrad(x, dy, m, dt=F32, splits=1):
    BT, G, AT = [t.to(dt) for t in mats(m)]
    B, Ci, H, W = x.shape
    a = m + 2
    xp = torch.nn.functional.pad(x.to(dt), (1, 1, 1, 1))
    d = xp.unfold(2, a, m).unfold(3, a, m)                         # [B,Ci,TH,TW,a,a]
    V = BT @ d @ BT.T
    g = dy.to(dt).unfold(2, m, m).unfold(3, m, m)                  # [B,Co,TH,TW,m,m]
    Wt = AT.T @ g @ AT                                             # A dy A^T: [B,Co,TH,TW,a,a]
    Vf = V.permute(1, 4, 5, 0, 2, 3).reshape(Ci, a, a, -1)         # [Ci,a,a,T]
    Wf = Wt.permute(1, 4, 5, 0, 2, 3).reshape(dy.shape[1], a, a, -1)
    T = Vf.shape[-1]
    out = 0
    for s in range(splits):                                        # each split: one fp32 chain over its tiles, then G^T . G
        sl = slice(s * T // splits, (s + 1) * T // splits)
        M = torch.einsum('oijt,cijt->ocij', Wf[..., sl], Vf[..., sl])
        out = out + G.T @ M @ G
    return out


def run(B, Ci, Co, H, relu_in, splits):
    x = torch.randn(B, Ci, H, H, dtype=F64)
    if relu_in:
        x = x.clamp_min(0)
    x = x.to(F32).to(F64)
    dy = (torch.randn(B, Co, H, H, dtype=F64) * 1e-2).to(F32).to(F64)
    xr = torch.nn.functional.pad(x, (1, 1, 1, 1)).unfold(2, 3, 1).unfold(3, 3, 1)          # [B,Ci,H,W,3,3]
    ref = torch.einsum('bohw,bchwrs->ocrs', dy, xr)
    scale, rms = ref.abs().max().item(), ref.pow(2).mean().sqrt().item()
    out = {'direct fp32 (one chain)': torch.einsum('bohw,bchwrs->ocrs', dy.to(F32), xr.to(F32)).to(F64),
           f'F(3x3,2x2) fp32, {splits} splits': wino_wgrad(x, dy, 2, splits=splits).to(F64),
           f'F(3x3,4x4) fp32, {splits} splits': wino_wgrad(x, dy, 4, splits=splits).to(F64),
           'F(3x3,4x4) float64 (algorithm check)': wino_wgrad(x, dy, 4, dt=F64)}
    print(f"B={B} Ci={Ci} Co={Co} H={H} relu_in={relu_in}: |dW|max={scale:.3e} rms={rms:.3e}  (pixels per weight: {B * H * H})")
    for k, v in out.items():
        e = (v - ref).abs()
        print(f"   {k:38s} max err / |dW|max = {e.max().item() / scale:.2e}   rms err / rms = {e.pow(2).mean().sqrt().item() / rms:.2e}")


if __name__ == "__main__":
    run(16, 32, 32, 32, True, 8)
    run(64, 16, 16, 32, True, 32)
    run(16, 32, 32, 64, False, 32)
    run(64, 32, 32, 8, True, 4)
