"""Throughput of the logit-record pass `_get_logit` over a CIFAR-sized synthetic dataset (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from diagan.cli import make_loader
from diagan.datasets.predefined import get_predefined_dataset
from diagan.models.predefined_models import get_gan_model
from diagan.trainer.trainer import LogTrainer
from diagan.utils.plot import LogitRecord
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
netG, netD, optG, optD = get_gan_model('cifar10', model='sngan', loss_type='ns')
ds = get_predefined_dataset('cifar10', num_data=N)
for bs in (64, 512):
    dl = make_loader(ds, bs)
    t = LogTrainer(output_path="gpurun_out/lp", netD=netD, netG=netG, optD=optD, optG=optG, dataloader=dl, num_steps=1,
                   log_dir="gpurun_out/lp", device='cuda')
    rec = LogitRecord(N, capacity=4, device=t.device)
    t._get_logit(netD, eval_mode=True, record=rec, step=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t._get_logit(netD, eval_mode=True, record=rec, step=1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"_get_logit N={N} batch {bs}: {dt:.2f} s = {N/dt:.0f} images/s")
