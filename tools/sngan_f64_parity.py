#!/usr/bin/env python
"""How close is the HIP path to the truth, compared with the fp32 CPU oracle?  (GPU box)
One D and one G update of SNGAN-32 / SNGAN-64 at batch 64 from identical weights, images and noise, three ways: the HIP
engine, oracle/nets.py in float32 and oracle/nets.py in float64.  Per parameter: relative L2 distance of each fp32
gradient to the float64 one.  ReLU-mask flips of near-zero pre-activations hit BOTH fp32 implementations; what is asked
is that the HIP path is not systematically further from float64 than plain PyTorch fp32 is."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import nets as O
from diagan.ops import conv as C


class Log:
    def __init__(self): self.m = {}
    def add_metric(self, name, value, group=None, precision=4): self.m[name] = value


def run(dataset, res, mode):
    from diagan.models.predefined_models import get_gan_model
    C.set_winograd(None); C.set_winograd4(None)
    if mode == "no F(4x4)": C.set_winograd4(False)
    if mode == "implicit GEMM only": C.set_winograd(False)
    oG, oD, ooptG, ooptD = O.make_pair(dataset, "ns", seed=1)
    torch.manual_seed(1)
    netG, netD, optG, optD = get_gan_model(dataset, model='sngan', loss_type="ns")
    netG.load_state_dict(oG.state_dict()); netD.load_state_dict(oD.state_dict())
    netG.to('cuda'); netD.to('cuda')
    dG, dD = copy.deepcopy(oG).double(), copy.deepcopy(oD).double()
    doptG = torch.optim.Adam(dG.parameters(), 2e-4, betas=(0.0, 0.9)); doptD = torch.optim.Adam(dD.parameters(), 2e-4, betas=(0.0, 0.9))
    B = 64
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, 3, res, res, generator=g) * 2 - 1
    zd, zg = torch.randn(B, 128, generator=g), torch.randn(B, 128, generator=g)
    e32 = oD.train_step((x, None), oG, ooptD, noise=zd)[0]
    e64 = dD.train_step((x.double(), None), dG, doptD, noise=zd.double())[0]
    log = netD.train_step(real_batch=(x.cuda(), None), netG=netG, optD=optD, log_data=Log(), device='cuda', noise=zd.cuda())
    eh = log.m['errD'].item()
    print(f"{dataset} [{mode}] errD: hip-f64 {abs(eh - e64):.2e}  oracle32-f64 {abs(e32 - e64):.2e}")
    worst = 0.0
    gr = netD.export_grads()
    rows = []
    for (k, p32), (_, p64) in zip(oD.named_parameters(), dD.named_parameters()):
        n = p64.grad.norm().item() + 1e-30
        rows.append(("D " + k, (gr[k].double().cpu() - p64.grad).norm().item() / n, (p32.grad.double() - p64.grad).norm().item() / n))
    g32 = oG.train_step((x, None), oD, ooptG, noise=zg)
    g64 = dG.train_step((x.double(), None), dD, doptG, noise=zg.double())
    log = netG.train_step(real_batch=(x.cuda(), None), netD=netD, optG=optG, log_data=Log(), device='cuda', noise=zg.cuda())
    print(f"{dataset} [{mode}] errG: hip-f64 {abs(log.m['errG'].item() - g64):.2e}  oracle32-f64 {abs(g32 - g64):.2e}")
    gr = netG.export_grads()
    wscale = max(p.grad.norm().item() for p in dG.parameters())
    for (k, p32), (_, p64) in zip(oG.named_parameters(), dG.named_parameters()):
        n = p64.grad.norm().item()
        if n < 1e-6 * wscale:
            continue                        # (conv biases in front of a BatchNorm: exactly-zero true gradient)
        rows.append(("G " + k, (gr[k].double().cpu() - p64.grad).norm().item() / n, (p32.grad.double() - p64.grad).norm().item() / n))
    for k, a, b in rows:
        print(f"   {k:28s} hip {a:.2e}  oracle32 {b:.2e}  ratio {a / max(b, 1e-30):6.2f}")
    print(f"{dataset} [{mode}] max ratio {max(a / max(b, 1e-30) for _, a, b in rows):.2f}, max hip rel err {max(a for _, a, _ in rows):.2e}, "
          f"max oracle32 rel err {max(b for _, _, b in rows):.2e}")
    C.set_winograd(None); C.set_winograd4(None)


if __name__ == "__main__":
    torch.set_num_threads(16)
    for dataset, res in (("cifar10", 32), ("celeba", 64)):
        for mode in ("default", "no F(4x4)", "implicit GEMM only"):
            run(dataset, res, mode)
