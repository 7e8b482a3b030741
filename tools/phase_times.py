import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from diagan.models.predefined_models import get_gan_model
from diagan.ops import eltwise as E

class L:
    def add_metric(self, *a, **k): pass

ds = sys.argv[1] if len(sys.argv) > 1 else 'cifar10'
res = 32 if ds == 'cifar10' else 64
torch.manual_seed(1)
netG, netD, optG, optD = get_gan_model(ds, model='sngan', loss_type='ns')
netG.to('cuda'); netD.to('cuda')
x = torch.rand(64, 3, res, res, device='cuda') * 2 - 1
def T(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
print('D step ms', T(lambda: netD.train_step(real_batch=(x, None), netG=netG, optD=optD, log_data=L(), device='cuda')))
print('G step ms', T(lambda: netG.train_step(real_batch=(x, None), netD=netD, optG=optG, log_data=L(), device='cuda')))
xn = E.nchw_to_nhwc(x, 4)
print('G fwd (no save) ms', T(lambda: netG.generate_images_nhwc(64, save=False)))
print('G fwd (save) ms', T(lambda: netG.generate_images_nhwc(64, save=True)))
print('D fwd ms', T(lambda: netD.forward_nhwc(xn, True, save=True, need_dgrad=True, need_in_dgrad=False)))
def dfb():
    netD.zero_grad()
    lo, ctx = netD.forward_nhwc(xn, True, save=True, need_dgrad=True, need_in_dgrad=False)
    netD.backward_nhwc(ctx, torch.ones(64, device='cuda'), need_wgrad=True, need_gx=False)
print('D fwd+bwd ms', T(dfb))
def gfb():
    netG.zero_grad()
    y, ctx = netG.generate_images_nhwc(64, save=True)
    netG.backward_nhwc(ctx, y)
print('G fwd+bwd ms', T(gfb))
import time
t0 = time.perf_counter()
for _ in range(10):
    netD.train_step(real_batch=(x, None), netG=netG, optD=optD, log_data=L(), device='cuda')
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host-side launch time per D step ms', (t1 - t0) * 100, 'drain ms', (t2 - t1) * 1000)
