import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch, torch.nn.functional as F
from oracle import nets as O
from diagan.models.predefined_models import get_gan_model
from diagan.ops import eltwise as E

def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return f"relmax {(a-b).abs().max().item()/(b.abs().max().item()+1e-30):.2e} relL2 {((a-b).norm()/(b.norm()+1e-30)).item():.2e}"

oG, oD, _, _ = O.make_pair('cifar10', 'ns', seed=1)
torch.manual_seed(1)
netG, netD, optG, optD = get_gan_model('cifar10', model='sngan', loss_type='ns')
netD.load_state_dict(oD.state_dict()); netD.to('cuda')
B = 8
g = torch.Generator().manual_seed(3)
x = torch.rand(B, 3, 32, 32, generator=g) * 2 - 1
xr = x.clone().requires_grad_(True)
hs = []
h = xr
for blk in (oD.block1, oD.block2, oD.block3, oD.block4):
    h = blk(h); h.retain_grad(); hs.append(h)
lo = oD.l5(torch.sum(F.relu(h), dim=(2, 3)))
lo.sum().backward()
rec = {}
for i, blk in enumerate(netD._blocks()):
    orig = blk.backward
    def mk(i, orig):
        def f(ctx, gout, **kw):
            rec[f'gout{i}'] = gout
            r = orig(ctx, gout, **kw)
            rec[f'gx{i}'] = r
            return r
        return f
    blk.backward = mk(i, orig)
logit, dctx = netD.forward_nhwc(E.nchw_to_nhwc(x.cuda(), 4), True, save=True, need_dgrad=True, need_in_dgrad=True)
gx = netD.backward_nhwc(dctx, torch.ones(B, device='cuda'), need_wgrad=True, need_gx=True)
for i in range(4):
    print(f'block{i+1} gout', rel(rec[f'gout{i}'].permute(0,3,1,2), hs[i].grad))
print('gx image', rel(E.nhwc_to_nchw(gx, 3), xr.grad))
# per-pixel error map of block1's gout (= block2's gx)
a = rec['gout0'].permute(0,3,1,2).double().cpu(); b = hs[0].grad.double()
err = (a-b).abs().amax(dim=(0,1))
print((err / b.abs().max()).numpy().round(4))
d = (a-b).abs()
idx = (d == d.max()).nonzero()[0].tolist()
print('argmax', idx, 'hip', a[tuple(idx)].item(), 'ref', b[tuple(idx)].item())
print('oracle block1 out at idx', hs[0][tuple(idx)].item())
hip_x = dctx['bctx'][1]['x'].permute(0,3,1,2).cpu()
print('hip    block1 out at idx', hip_x[tuple(idx)].item())
print('num elements |x|<1e-6:', (hs[0].abs() < 1e-6).sum().item(), 'of', hs[0].numel())
