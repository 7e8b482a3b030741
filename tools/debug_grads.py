import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from oracle import nets as O
from diagan.models.predefined_models import get_gan_model

class Log:
    def __init__(self): self.m = {}
    def add_metric(self, name, value, group=None, precision=4): self.m[name] = value

dataset, res, loss = (sys.argv[1:] + ['cifar10', '32', 'ns'])[:3]
res = int(res)
oG, oD, ooptG, ooptD = O.make_pair(dataset, loss, seed=1)
torch.manual_seed(1)
netG, netD, optG, optD = get_gan_model(dataset, model='sngan', loss_type=loss)
netG.load_state_dict(oG.state_dict()); netD.load_state_dict(oD.state_dict())
netG.to('cuda'); netD.to('cuda')
B = 8
g = torch.Generator().manual_seed(3)
x = torch.rand(B, 3, res, res, generator=g) * 2 - 1
zd = torch.randn(B, 128, generator=g); zg = torch.randn(B, 128, generator=g)
errD, D_x, D_Gz = oD.train_step((x, None), oG, ooptD, noise=zd)
log = netD.train_step(real_batch=(x.cuda(), None), netG=netG, optD=optD, log_data=Log(), device='cuda', noise=zd.cuda())
print('errD', errD, log.m['errD'].item())
gr = netD.export_grads()
for k, p in oD.named_parameters():
    a, b = gr[k].double().cpu(), p.grad.double()
    print(f"D {k:24s} err {(a-b).abs().max().item():.3e} scale {b.abs().max().item():.3e} rel {(a-b).abs().max().item()/(b.abs().max().item()+1e-30):.2e}")
errG = oG.train_step((x, None), oD, ooptG, noise=zg)
log = netG.train_step(real_batch=(x.cuda(), None), netD=netD, optG=optG, log_data=Log(), device='cuda', noise=zg.cuda())
print('errG', errG, log.m['errG'].item())
gr = netG.export_grads()
for k, p in oG.named_parameters():
    a, b = gr[k].double().cpu(), p.grad.double()
    print(f"G {k:24s} err {(a-b).abs().max().item():.3e} scale {b.abs().max().item():.3e} rel {(a-b).abs().max().item()/(b.abs().max().item()+1e-30):.2e}")
