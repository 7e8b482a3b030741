import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from oracle import nets as O
from diagan.models.predefined_models import get_gan_model
from diagan.ops import eltwise as E

def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return (a-b).abs().max().item()/(b.abs().max().item()+1e-30), ((a-b).norm()/(b.norm()+1e-30)).item()

dataset, res, loss = 'cifar10', 32, 'ns'
oG, oD, ooptG, ooptD = O.make_pair(dataset, loss, seed=1)
torch.manual_seed(1)
netG, netD, optG, optD = get_gan_model(dataset, model='sngan', loss_type=loss)
netG.load_state_dict(oG.state_dict()); netD.load_state_dict(oD.state_dict())
netG.to('cuda'); netD.to('cuda')
B = 8
g = torch.Generator().manual_seed(3)
x = torch.rand(B, 3, res, res, generator=g) * 2 - 1
z = torch.randn(B, 128, generator=g)
gi = torch.randn(B, 3, res, res, generator=g)
# --- G isolated
img = oG(z); img.backward(gi)
netG.zero_grad()
y, ctx = netG.forward_nhwc(z.cuda(), True, save=True)
print('G fwd', rel(E.nhwc_to_nchw(y, 3), img.detach()))
netG.backward_nhwc(ctx, E.nchw_to_nhwc(gi.cuda(), 4))
gr = netG.export_grads()
for k, p in oG.named_parameters():
    print(f"G {k:22s} relmax {rel(gr[k], p.grad)[0]:.2e} relL2 {rel(gr[k], p.grad)[1]:.2e}")
# --- D isolated: input gradient and weight grads with dlogit = ones
xr = x.clone().requires_grad_(True)
oD.zero_grad()
lo = oD(xr); lo.sum().backward()
netD.zero_grad()
logit, dctx = netD.forward_nhwc(E.nchw_to_nhwc(x.cuda(), 4), True, save=True, need_dgrad=True, need_in_dgrad=True)
print('D fwd', rel(logit, lo.detach()))
gx = netD.backward_nhwc(dctx, torch.ones(B, device='cuda'), need_wgrad=True, need_gx=True)
print('D gx', rel(E.nhwc_to_nchw(gx, 3), xr.grad))
gr = netD.export_grads()
for k, p in oD.named_parameters():
    print(f"D {k:22s} relmax {rel(gr[k], p.grad)[0]:.2e} relL2 {rel(gr[k], p.grad)[1]:.2e}")
