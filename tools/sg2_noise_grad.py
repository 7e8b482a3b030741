"""How accurate are the NoiseInjection strength gradients at 256 x 256 (tests/golden/stylegan2_256.npz's case)?
Each is ONE scalar: the sum of g * noise over B*C*H*W elements of either sign.  Truth = the pinned oracle evaluated in
float64; compared: the oracle in fp32 on CPU (= the reference's own arithmetic: it reproduces the golden's values),
the HIP engine on the implicit-GEMM kernels, the HIP engine with the 3x3 layers on the Winograd kernels.  Deviations
are printed in units of 1e-6 of the sum of |terms| (the fixture's `*_noise_abs`), the natural scale of an fp32 sum.
    python tools/sg2_noise_grad.py          (GPU box; ~1 min of CPU for the float64 pass)"""
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import numpy as np
import torch

from oracle import stylegan2 as O

warnings.filterwarnings("ignore")
G256 = np.load(os.path.join(ROOT, "tests", "golden", "stylegan2_256.npz"))
SIZE, CM, BATCH = int(G256["size"]), int(G256["channel_multiplier"]), int(G256["batch"])
KEYS = [str(k) for k in G256["path_grad_noise_keys"]]


def path_inputs():
    gen = torch.Generator().manual_seed(7)
    torch.randn(BATCH, 512, generator=gen), torch.randn(BATCH, 512, generator=gen)
    torch.randn(BATCH, 3, SIZE, SIZE, generator=gen)
    return torch.randn(1, 512, generator=gen), torch.randn(1, 3, SIZE, SIZE, generator=gen)


def oracle_pass(dt):
    sg = {k: v.to(dt) for k, v in O.seeded_state(O.generator_shapes(SIZE, mult=CM), int(G256["seed_g"])).items()}
    sd = {k: v.to(dt) for k, v in O.seeded_state(O.discriminator_shapes(SIZE, mult=CM), int(G256["seed_d"])).items()}
    pg = {k: v.clone().requires_grad_(not k.startswith("noises.")) for k, v in sg.items()}
    fake, _ = O.generator(pg, SIZE, [torch.from_numpy(G256["z1"]).to(dt)])
    O.g_nonsaturating_loss(O.discriminator(sd, SIZE, fake)).backward()
    out = {"g_loss_grad": [float(pg[k].grad.double().norm()) for k in KEYS]}
    for k in pg:
        pg[k].grad = None
    zp, pl_noise = path_inputs()
    fake, lat = O.generator(pg, SIZE, [zp.to(dt)])
    pl, _, _ = O.g_path_regularize(fake, lat, 0.3, pl_noise.to(dt))
    (2.0 * 4 * pl + 0 * fake[0, 0, 0, 0]).backward()
    out["path_grad"] = [float(pg[k].grad.double().norm()) for k in KEYS]
    return out


def engine_pass(wino):
    from diagan.models import stylegan2 as M
    from diagan.ops import diffconv as DC
    from diagan.trainer import stylegan2 as TR
    DC.SG2_WINO = bool(wino)
    G = M.StyleGANGenerator(size=SIZE, channel_multiplier=CM)
    D = M.StyleGANDiscriminator(size=SIZE, channel_multiplier=CM)
    G.load_state_dict(O.seeded_state(O.generator_shapes(SIZE, mult=CM), int(G256["seed_g"])), strict=False)
    D.load_state_dict(O.seeded_state(O.discriminator_shapes(SIZE, mult=CM), int(G256["seed_d"])), strict=False)
    G.cuda(), D.cuda()
    TR.requires_grad(D, False)
    P = dict(G.named_parameters())
    fake, _ = G([torch.from_numpy(G256["z1"]).cuda()], randomize_noise=False)
    TR.g_nonsaturating_loss(D(fake)).backward()
    out = {"g_loss_grad": [float(P[k].grad.double().norm()) for k in KEYS]}
    G.zero_grad()
    zp, pl_noise = path_inputs()
    fake, lat = G([zp.cuda()], return_latents=True, randomize_noise=False)
    pl, _, _ = TR.g_path_regularize(fake, lat, 0.3, noise=pl_noise.cuda())
    (2.0 * 4 * pl + 0 * fake[0, 0, 0, 0]).backward()
    out["path_grad"] = [float(P[k].grad.double().norm()) for k in KEYS]
    return out


def main():
    truth = oracle_pass(torch.float64)
    runs = [("cpu fp32", oracle_pass(torch.float32))]
    if torch.cuda.is_available():
        runs += [("implicit", engine_pass(False)), ("winograd", engine_pass(True))]
    for tag in ("g_loss_grad", "path_grad"):
        gk = [str(k) for k in G256[f"{tag}_keys"]]
        print(f"{tag}: |value - float64| in units of 1e-6 x sum|terms|")
        print(f"  {'strength':22s} {'float64':>12s} {'sum|terms|':>11s} {'cond':>7s} | {'golden':>7s} " + " ".join(f"{n:>8s}" for n, _ in runs))
        for i, k in enumerate(KEYS):
            t, a = truth[tag][i], float(G256[f"{tag}_noise_abs"][i])
            gold = float(G256[f"{tag}_norms"][gk.index(k)])
            print(f"  {k:22s} {t:12.5e} {a:11.4e} {a / t:7.0f} | {abs(gold - t) / a * 1e6:7.2f} "
                  + " ".join(f"{abs(r[tag][i] - t) / a * 1e6:8.2f}" for _, r in runs))


if __name__ == "__main__":
    main()
