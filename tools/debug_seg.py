import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from diagan.ops import conv as C
torch.manual_seed(0)
B, H, W, Ci, Co = 16, 8, 8, 128, 128
geom = C.Geom('conv', Ci, Co, 3, 3, 1, 1)
x = torch.randn(B, H, W, Ci, device='cuda'); dy = torch.randn(B, H, W, Co, device='cuda')
n_w = Co * geom.Kp; stride = n_w + Co
for segs in (1, 2):
    splits = max(segs, C.wgrad_splits(B*H*W, Co, geom.Kp) // segs * segs)
    slab = torch.full((splits, stride), float('nan'), device='cuda')
    C.conv_wgrad_into(geom, dy, x, slab, splits, stride, n_w, segments=segs)
    torch.cuda.synchronize()
    print('segs', segs, 'splits', splits, 'row abs sums', slab[:, :n_w].abs().sum(1).tolist())
    print('   bias sums', slab[:, n_w:].abs().sum(1).tolist())
    if segs == 2:
        h = splits // 2
        g0 = torch.zeros(Co, geom.Kp, device='cuda'); g1 = torch.zeros(Co, geom.Kp, device='cuda')
        C.conv_wgrad(geom, dy[:B//2].contiguous(), x[:B//2].contiguous(), g0, False)
        C.conv_wgrad(geom, dy[B//2:].contiguous(), x[B//2:].contiguous(), g1, False)
        print('half0 err', (slab[:h, :n_w].sum(0) - g0.view(-1)).abs().max().item(), 'half1 err', (slab[h:, :n_w].sum(0) - g1.view(-1)).abs().max().item(), g0.abs().max().item())
        print('bias half1 err', (slab[h:, n_w:].sum(0) - dy[B//2:].sum((0,1,2))).abs().max().item())
