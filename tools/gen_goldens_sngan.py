#!/usr/bin/env python
"""tests/golden/sngan.npz from a REAL torch-mimicry install -- the pin the SNGAN rows of SURVEY §8 (a2-a8, a11-a14) lack.

The reference takes its SNGAN networks, residual blocks, spectral norm, base losses and train steps from
`torch-mimicry==0.1.16` (requirements.txt:72; call sites diagan-pkg/diagan/models/predefined_models.py:14-92).  That
package is not vendored under /root/reference and cannot be installed offline, so `oracle/nets.py` restates it from
the published algorithm and every SNGAN parity claim of this repo says "parity unpinned".  This script turns a
machine that HAS the package into the missing fixture:

    pip install torch-mimicry==0.1.16        # on a machine with network access
    python tools/gen_goldens_sngan.py        # writes tests/golden/sngan.npz (< 1 MB)
    python -m pytest tests/test_oracle_models.py tests/test_sngan_gpu.py -k golden

It never imports anything of this repo's restatement: the vectors come from mimicry's own classes.  Without the
package it exits with status 3 and writes nothing (it never fakes a fixture).  What it records, per resolution
(32: CIFAR-10 nets, 64: CelebA nets), with `torch.manual_seed(SEED)` before each construction:

  * state-dict key order and per-tensor checksums (sum, |sum|) of the seeded initialisation -> init order and init
    functions (xavier gains, sn_u draws);
  * inputs z, x (seeded) and G(z) / D(x) in eval mode and in train mode -> forward arithmetic incl. BatchNorm batch
    statistics and the spectral-norm power iteration;
  * sn_u and sn_sigma of every SN layer after two train-mode forwards;
  * one D train step and one G train step (ns and hinge losses): errD / errG, D(x), D(G(z)) and the gradients of the
    first and the last parameterised layer of the stepped network, plus a parameter checksum after the Adam update;
  * the key list of a checkpoint written by `save_checkpoint` (file layout of f3).
Large tensors are stored in the compact form of tools/gen_goldens_models.compact (sum, |sum|, stride, samples)."""
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "sngan.npz")
SEED, B = 11, 4


def compact(t, limit=16384):
    a = np.asarray(t.detach().cpu().numpy() if torch.is_tensor(t) else t, dtype=np.float64).reshape(-1)
    if a.size <= limit:
        return a.astype(np.float32) if a.size else a
    k = -(-a.size // 4096)
    return np.concatenate([[a.sum(), np.abs(a).sum(), float(k)], a[::k][:4096]])


def checks(sd, tag, out):
    out[f"{tag}_keys"] = np.array(list(sd.keys()))
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            out[f"ck_{tag}_{k}"] = np.array([v.double().sum().item(), v.double().abs().sum().item()])


class _Log:
    """stands in for mimicry's MetricLog (the train steps only call add_metric)"""

    def __init__(self):
        self.m = {}

    def add_metric(self, name, value, group=None, precision=4):
        self.m[name] = float(value)


def main():
    try:
        import torch_mimicry as mmc                                    # noqa: F401
        from torch_mimicry.nets import sngan
    except ImportError as e:
        print(f"torch_mimicry is not importable here ({e}); nothing written.  Install torch-mimicry==0.1.16 on a "
              f"machine with network access and re-run.", file=sys.stderr)
        return 3
    out = {"mimicry_version": np.array(getattr(mmc, "__version__", "unknown")), "seed": np.array(SEED), "batch": np.array(B)}
    for res, G, D in ((32, sngan.SNGANGenerator32, sngan.SNGANDiscriminator32),
                      (64, sngan.SNGANGenerator64, sngan.SNGANDiscriminator64)):
        for loss in ("ns", "hinge"):
            tag = f"r{res}_{loss}"
            torch.manual_seed(SEED)
            netG = G(loss_type=loss)
            netD = D(loss_type=loss)
            optG = torch.optim.Adam(netG.parameters(), 2e-4, betas=(0.0, 0.9))      # predefined_models.py:32,51,70,89
            optD = torch.optim.Adam(netD.parameters(), 2e-4, betas=(0.0, 0.9))
            if loss == "ns":
                checks(netG.state_dict(), f"G{res}", out)
                checks(netD.state_dict(), f"D{res}", out)
            gen = torch.Generator().manual_seed(5)
            z = torch.randn(B, 128, generator=gen)
            x = torch.rand(B, 3, res, res, generator=gen) * 2 - 1
            zd, zg = torch.randn(B, 128, generator=gen), torch.randn(B, 128, generator=gen)
            out[f"{tag}_z"], out[f"{tag}_x"], out[f"{tag}_zd"], out[f"{tag}_zg"] = z.numpy(), x.numpy(), zd.numpy(), zg.numpy()
            netG.eval(), netD.eval()
            with torch.no_grad():
                out[f"{tag}_G_eval"] = compact(netG(z))
                out[f"{tag}_D_eval"] = netD(x).numpy()
            netG.train(), netD.train()
            with torch.no_grad():
                out[f"{tag}_G_train"] = compact(netG(z))
                out[f"{tag}_D_train1"] = netD(x).numpy()
                out[f"{tag}_D_train2"] = netD(x).numpy()
            for k, v in netD.state_dict().items():
                if "sn_u" in k or "sn_sigma" in k:
                    out[f"{tag}_after2_{k}"] = v.numpy().copy()
            # one D step and one G step with the noise injected through torch's global generator: mimicry draws
            # `torch.randn((n, nz), device=device)` inside generate_images, so seed right before each step
            torch.manual_seed(SEED + 1)
            log = netD.train_step(real_batch=(x, None), netG=netG, optD=optD, log_data=_Log(), device=torch.device("cpu"))
            out[f"{tag}_errD"], out[f"{tag}_Dx"], out[f"{tag}_DGz"] = (np.array(log.m[k]) for k in ("errD", "D(x)", "D(G(z))"))
            pD = list(netD.named_parameters())
            for name, p in (pD[0], pD[-2], pD[-1]):
                out[f"{tag}_gradD_{name}"] = compact(p.grad)
            out[f"{tag}_D_after_step"] = np.array([sum(p.double().sum().item() for p in netD.parameters()),
                                                   sum(p.double().abs().sum().item() for p in netD.parameters())])
            torch.manual_seed(SEED + 2)
            log = netG.train_step(real_batch=(x, None), netD=netD, optG=optG, log_data=_Log(), device=torch.device("cpu"))
            out[f"{tag}_errG"] = np.array(log.m["errG"])
            pG = list(netG.named_parameters())
            for name, p in (pG[0], pG[1], pG[-2]):
                out[f"{tag}_gradG_{name}"] = compact(p.grad)
            out[f"{tag}_G_after_step"] = np.array([sum(p.double().sum().item() for p in netG.parameters()),
                                                   sum(p.double().abs().sum().item() for p in netG.parameters())])
            if res == 32 and loss == "ns":
                import tempfile
                with tempfile.TemporaryDirectory() as d:
                    ck = os.path.join(d, "netD")
                    netD.save_checkpoint(directory=ck, global_step=7, optimizer=optD)
                    f = os.path.join(ck, os.listdir(ck)[0])
                    out["ckpt_file_name"] = np.array(os.path.basename(f))
                    sd = torch.load(f, map_location="cpu", weights_only=False)
                    out["ckpt_top_keys"] = np.array(sorted(sd.keys()))
                    out["ckpt_model_keys"] = np.array(list(sd["model_state_dict"].keys()))
                    out["ckpt_opt_state_n"] = np.array(len(sd["optimizer_state_dict"]["state"]))
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {len(out)} arrays, {os.path.getsize(OUT) / 1e3:.0f} kB")
    return 0


if __name__ == "__main__":
    sys.exit(main())
