#!/usr/bin/env python
"""is the SNGAN-64 generator's distance to float64 (images 3.5e-6 against the CPU's 8.8e-7, gradients 1.2x -
2.2x the CPU's from the same upstream gradient) the length of the fp32 accumulation chains?  Test without a new kernel: a
channel split of a launch IS a blocked sum -- `diagan_conv_gemm_tune(force_ksplit = s)` cuts every forward / data-gradient
accumulation chain into s pieces that a second stage adds in a fixed order, DIAGAN_WGRAD_MINSTEPS=1 (environment, read once) lets
the weight-gradient policy use its maximum of splits.  One process per mode (the float64 oracle is computed once and cached):

    python tools/chain_length_probe.py <mode>      mode: default | ksplit4 | ksplit8 | nowino | nowino_ksplit8
"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from oracle import nets as O
from diagan import _native as nat
from diagan.ops import conv as C
from diagan.ops import eltwise as E
from diagan.models.predefined_models import get_gan_model

mode = sys.argv[1] if len(sys.argv) > 1 else "default"
dataset, res, B = "celeba", 64, 64
torch.set_num_threads(32)
oG, oD, _, _ = O.make_pair(dataset, "ns", seed=1)
torch.manual_seed(1)
netG, netD, optG, optD = get_gan_model(dataset, model='sngan', loss_type="ns")
netG.load_state_dict(oG.state_dict())
netG.to('cuda')
g = torch.Generator().manual_seed(5)
z = torch.randn(B, 128, generator=g)
up = torch.randn(B, 3, res, res, generator=g) * 1e-3
cache = "/tmp/chain_probe_f64.pt"
for n in (oG, netG):
    n.train()
if os.path.exists(cache):
    ref = torch.load(cache)
else:
    dG = copy.deepcopy(oG).double().train()
    img64 = dG(z.double()); img64.backward(up.double())
    img32 = oG(z); img32.backward(up)
    ref = dict(img64=img64.detach(), g64={k: p.grad for k, p in dG.named_parameters()}, img32=img32.detach(),
               g32={k: p.grad for k, p in oG.named_parameters()})
    torch.save(ref, cache)


def rel(a, b):
    return (a.detach().double().cpu() - b.detach().double()).norm().item() / (b.double().norm().item() + 1e-30)


if "nowino" in mode:
    C.set_winograd(False)
ks = [int(t[6:]) for t in mode.split("_") if t.startswith("ksplit")]
if ks:
    nat.call("diagan_conv_gemm_tune", ks[0], -1, 0)
netG.zero_grad()
himg, ctx = netG.forward_nhwc(z.cuda(), True, save=True)
netG.backward_nhwc(ctx, E.nchw_to_nhwc(up.cuda(), 4))
gr = netG.export_grads()
wscale = max(v.norm().item() for v in ref['g64'].values())
hip, o32 = [], []
for k, p64 in ref['g64'].items():
    if p64.norm().item() < 1e-6 * wscale:
        continue
    hip.append(rel(gr[k], p64)); o32.append(rel(ref['g32'][k], p64))
rms = lambda v: (sum(e * e for e in v) / len(v)) ** 0.5
print(f"CHAIN mode={mode:16s} wgrad_minsteps={os.environ.get('DIAGAN_WGRAD_MINSTEPS', '4')} | images: hip {rel(E.nhwc_to_nchw(himg, 3), ref['img64']):.2e} "
      f"oracle32 {rel(ref['img32'], ref['img64']):.2e} | G gradients from the same upstream gradient, rms over parameters: hip {rms(hip):.2e} "
      f"oracle32 {rms(o32):.2e} (max hip {max(hip):.2e}, oracle32 {max(o32):.2e})", flush=True)
