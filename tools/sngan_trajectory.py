#!/usr/bin/env python
"""SNGAN training trajectories against float64 (VERDICT r4 "Next 4"; the loop of
diagan-pkg/diagan/trainer/trainer.py:238-299).  GPU box.

N global steps (n_dis = 5 D updates + 1 G update, Adam, batch 64) from identical weights, with the SAME injected real batches and
latent noise, four ways:
  hip default     the engine as shipped (Winograd F(4x4,3x3) / F(2x2,3x3) where they qualify)
  hip exact-fp32  DIAGAN_WINO=0 DIAGAN_WINO4=0 DIAGAN_GEMM_X3=0: the fp32 implicit GEMM everywhere (an fmaf chain per output)
  cpu fp32        oracle/nets.py (plain PyTorch CPU autograd)
  cpu float64     oracle/nets.py in double: the reference trajectory
Per step: |errD - errD64| (mean over the step's D updates), |errG - errG64|, and the distance of all parameters to the float64
run, relative to how far the float64 run has moved from the common start.  What is asked of the default build: it drifts from
float64 no faster than 2x the exact-fp32 build and no faster than 3x the CPU fp32 oracle does.

  python tools/sngan_trajectory.py [steps] [dataset]      ->  table (profiles/r05_trajectory.md is made from it)"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-diagnosing-gan_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch


class Log:
    def __init__(self):
        self.m = {}

    def add_metric(self, name, value, group=None, precision=4):
        self.m[name] = value


def _flat64(named):
    return torch.cat([v.detach().double().cpu().reshape(-1) for _, v in named])


SKETCH_K = 16384


def sketch(p, k=SKETCH_K, seed=1234):
    """count-sketch of a parameter vector: E |sketch(x)|^2 = |x|^2 with a relative standard deviation of sqrt(2 / k) ~ 1.1 % -- lets a
    committed fixture stand in for the 5.3 M-parameter float64 trajectory (tests/golden/sngan_trajectory.npz, 0.7 MB)"""
    g = torch.Generator().manual_seed(seed)
    n = p.numel()
    bucket = torch.randint(0, k, (n,), generator=g)
    sign = torch.randint(0, 2, (n,), generator=g).double() * 2 - 1
    return torch.zeros(k, dtype=torch.float64).index_add_(0, bucket, p.double().cpu() * sign)


def trajectories(dataset="cifar10", steps=20, B=64, n_dis=5, loss="ns", seed=1, hip_modes=("default", "exact-fp32"), golden=None):
    """-> {name: {"errD": [steps], "errG": [steps], "dist": [steps]}} with dist / errors measured against the float64 run, and
    "moved": how far the float64 parameters are from the start after each step (norms over G and D together).
    golden: a dict made by tools/gen_goldens_trajectory.py from THIS function's CPU runs (float64 trajectory as count-sketches, the
    CPU fp32 run's distances) -- the CPU legs, minutes of float64 autograd, are then not repeated and the HIP runs are measured
    against the sketches (distances to ~1 %)"""
    from oracle import nets as O
    from diagan.models.predefined_models import get_gan_model
    from diagan.ops import conv as C
    res = 32 if dataset == "cifar10" else 64
    oG, oD, ooptG, ooptD = O.make_pair(dataset, loss, seed=seed)
    start = {"G": copy.deepcopy(oG.state_dict()), "D": copy.deepcopy(oD.state_dict())}
    g = torch.Generator().manual_seed(5)
    reals = [[torch.rand(B, 3, res, res, generator=g) * 2 - 1 for _ in range(n_dis)] for _ in range(steps)]
    zd = [[torch.randn(B, 128, generator=g) for _ in range(n_dis)] for _ in range(steps)]
    zg = [torch.randn(B, 128, generator=g) for _ in range(steps)]
    names_G = [k for k, _ in oG.named_parameters()]
    names_D = [k for k, _ in oD.named_parameters()]

    def cpu_run(double):
        G, D = copy.deepcopy(oG), copy.deepcopy(oD)
        if double:
            G, D = G.double(), D.double()
        optG = torch.optim.Adam(G.parameters(), 2e-4, betas=(0.0, 0.9))
        optD = torch.optim.Adam(D.parameters(), 2e-4, betas=(0.0, 0.9))
        cast = (lambda t: t.double()) if double else (lambda t: t)
        eD, eG, params = [], [], []
        for s in range(steps):
            ed = [D.train_step((cast(reals[s][i]), None), G, optD, noise=cast(zd[s][i]))[0] for i in range(n_dis)]
            eg = G.train_step((cast(reals[s][-1]), None), D, optG, noise=cast(zg[s]))
            eD.append(float(sum(ed) / n_dis))
            eG.append(float(eg))
            params.append(torch.cat([_flat64(G.named_parameters()), _flat64(D.named_parameters())]))
        return eD, eG, params

    def hip_run(mode):
        C.set_winograd(None)
        C.set_winograd4(None)
        C.set_gemm_x3(None)
        if mode == "exact-fp32":
            C.set_winograd(False)
            C.set_winograd4(False)
            C.set_gemm_x3(False)               # (the split-operand implicit GEMM of round 5 is fp32-grade, not an fmaf chain)
        try:
            torch.manual_seed(seed)
            netG, netD, optG, optD = get_gan_model(dataset, model='sngan', loss_type=loss)
            netG.load_state_dict(start["G"])
            netD.load_state_dict(start["D"])
            netG.to('cuda')
            netD.to('cuda')
            eD, eG, params = [], [], []
            for s in range(steps):
                ed = []
                for i in range(n_dis):
                    log = netD.train_step(real_batch=(reals[s][i].cuda(), None), netG=netG, optD=optD, log_data=Log(),
                                          device='cuda', noise=zd[s][i].cuda())
                    ed.append(log.m['errD'].item())
                log = netG.train_step(real_batch=(reals[s][-1].cuda(), None), netD=netD, optG=optG, log_data=Log(),
                                      device='cuda', noise=zg[s].cuda())
                eD.append(sum(ed) / n_dis)
                eG.append(log.m['errG'].item())
                sdG, sdD = netG.state_dict(), netD.state_dict()
                params.append(torch.cat([torch.cat([sdG[k].detach().double().cpu().reshape(-1) for k in names_G]),
                                         torch.cat([sdD[k].detach().double().cpu().reshape(-1) for k in names_D])]))
            return eD, eG, params
        finally:
            C.set_winograd(None)
            C.set_winograd4(None)
            C.set_gemm_x3(None)

    p0 = torch.cat([_flat64(oG.named_parameters()), _flat64(oD.named_parameters())])
    if golden is not None:
        import numpy as np
        assert int(golden["steps"]) >= steps and str(golden["dataset"]) == dataset and int(golden["n"]) == p0.numel()
        out = {"moved": [float(v) for v in golden["moved"][:steps]], "param_norm": float(golden["param_norm"]),
               "errD64": [float(v) for v in golden["errD64"][:steps]], "errG64": [float(v) for v in golden["errG64"][:steps]],
               "cpu fp32": {k: [float(v) for v in golden["cpu_fp32_" + k][:steps]] for k in ("errD", "errG", "dist")}}
        sk64 = torch.from_numpy(np.asarray(golden["sketch64"]))
        for m in hip_modes:
            eD, eG, params = hip_run(m)
            out["hip " + m] = {"errD": [abs(a - b) for a, b in zip(eD, out["errD64"])], "errG": [abs(a - b) for a, b in zip(eG, out["errG64"])],
                               "dist": [(sketch(p) - sk64[i]).norm().item() / max(out["moved"][i], 1e-30) for i, p in enumerate(params)]}
        return out
    runs = {"cpu float64": cpu_run(True), "cpu fp32": cpu_run(False)}
    for m in hip_modes:
        runs["hip " + m] = hip_run(m)
    eD64, eG64, p64 = runs["cpu float64"]
    moved = [(p - p0).norm().item() for p in p64]
    out = {"moved": moved, "param_norm": p0.norm().item(), "errD64": eD64, "errG64": eG64, "_p64": p64, "_n": p0.numel()}
    for name, (eD, eG, params) in runs.items():
        if name == "cpu float64":
            continue
        out[name] = {"errD": [abs(a - b) for a, b in zip(eD, eD64)], "errG": [abs(a - b) for a, b in zip(eG, eG64)],
                     "dist": [(p - q).norm().item() / max(m, 1e-30) for p, q, m in zip(params, p64, moved)]}
    return out


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dataset = sys.argv[2] if len(sys.argv) > 2 else "cifar10"
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    r = trajectories(dataset, steps)
    names = [k for k in r if k not in ("moved", "param_norm", "errD64", "errG64", "_p64", "_n")]
    print(f"SNGAN {dataset}, {steps} global steps (5 D + 1 G updates, batch 64, 'ns' loss); |parameters| = {r['param_norm']:.1f}")
    print("step  errD64   errG64   moved    | " + " | ".join(f"{n:>15s}: dD      dG      dist  " for n in names))
    for s in range(steps):
        row = f"{s + 1:4d} {r['errD64'][s]:8.4f} {r['errG64'][s]:8.4f} {r['moved'][s]:8.4f} | "
        row += " | ".join(f"{'':>15s}  {r[n]['errD'][s]:.1e} {r[n]['errG'][s]:.1e} {r[n]['dist'][s]:.1e}" for n in names)
        print(row)
    last = {n: r[n]['dist'][-1] for n in names}
    print("distance to float64 after the last step, relative to the float64 run's own movement: " +
          ", ".join(f"{n} {v:.2e}" for n, v in last.items()))
    d, e, c = last.get("hip default"), last.get("hip exact-fp32"), last.get("cpu fp32")
    print(f"default / exact-fp32 = {d / e:.2f} (bar 2), default / cpu fp32 = {d / c:.2f} (bar 3)")


if __name__ == "__main__":
    main()
