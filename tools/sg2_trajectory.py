"""Where does the three-iteration StyleGAN2 trajectory (tests/test_stylegan2_gpu.py::test_training_trajectory_vs_oracle)
end up, measured against the oracle evaluated in float64: the oracle in fp32 (the reference's own arithmetic on CPU),
the HIP engine on the implicit-GEMM kernels, the HIP engine with the 3x3 layers on the Winograd kernels.
    python tools/sg2_trajectory.py [size] [batch] [iters]"""
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch

from oracle import stylegan2 as O

warnings.filterwarnings("ignore")
GR, DR = 4 / 5, 16 / 17


def draws(size, batch, iters):
    gen = torch.Generator().manual_seed(77)
    out = []
    for _ in range(iters):
        real = torch.rand(batch, 3, size, size, generator=gen) * 2 - 1
        z = [torch.randn(batch, 512, generator=gen) for _ in range(4)]
        out.append((real, z, torch.randn(2, 3, size, size, generator=gen)))
    return out


def oracle_run(dtype, size, batch, iters):
    sg = O.seeded_state(O.generator_shapes(size), 31)
    sd_ = O.seeded_state(O.discriminator_shapes(size), 32)
    pg = {k: v.clone().to(dtype).requires_grad_(not k.startswith("noises.")) for k, v in sg.items()}
    pd = {k: v.clone().to(dtype).requires_grad_(True) for k, v in sd_.items()}
    og = torch.optim.Adam([v for v in pg.values() if v.requires_grad], lr=0.002 * GR, betas=(0 ** GR, 0.99 ** GR))
    od = torch.optim.Adam(list(pd.values()), lr=0.002 * DR, betas=(0 ** DR, 0.99 ** DR))

    def step(opt, loss):
        opt.zero_grad()
        loss.backward()
        opt.step()

    mp, losses = 0, []
    for real, z, pl_noise in draws(size, batch, iters):
        real, z, pl_noise = real.to(dtype), [t.to(dtype) for t in z], pl_noise.to(dtype)
        with torch.no_grad():
            fake, _ = O.generator(pg, size, [z[0]])
        d_loss = O.d_logistic_loss(O.discriminator(pd, size, real), O.discriminator(pd, size, fake))
        step(od, d_loss)
        xr = real.clone().requires_grad_(True)
        rp = O.discriminator(pd, size, xr)
        r1 = O.d_r1_loss(rp, xr)
        step(od, 10.0 / 2 * r1 * 16 + 0 * rp[0])
        fake, _ = O.generator(pg, size, [z[1], z[2]], inject_index=2)
        g_loss = O.g_nonsaturating_loss(O.discriminator({k: v.detach() for k, v in pd.items()}, size, fake))
        step(og, g_loss)
        fake, lat = O.generator(pg, size, [z[3][:2]])
        pl, mp, _ = O.g_path_regularize(fake, lat, mp, pl_noise)
        step(og, 2.0 * 4 * pl + 0 * fake[0, 0, 0, 0])
        losses.append([float(d_loss), float(r1), float(g_loss), float(pl)])
    return losses, float(mp), {k: v.detach() for k, v in pg.items()}, {k: v.detach() for k, v in pd.items()}


def engine_run(wino, size, batch, iters):
    from diagan.models import stylegan2 as M
    from diagan.ops import diffconv as DC
    from diagan.trainer import stylegan2 as TR
    DC.SG2_WINO = bool(wino)
    sg = O.seeded_state(O.generator_shapes(size), 31)
    sd_ = O.seeded_state(O.discriminator_shapes(size), 32)
    G, D = M.StyleGANGenerator(size=size), M.StyleGANDiscriminator(size=size)
    G.load_state_dict(sg, strict=False), D.load_state_dict(sd_, strict=False)
    G.cuda(), D.cuda()
    g_optim, d_optim = TR.make_optimizers(G, D, lr=0.002, g_reg_every=4, d_reg_every=16)
    mp, losses = 0, []
    for real, z, pl_noise in draws(size, batch, iters):
        TR.requires_grad(G, False), TR.requires_grad(D, True)
        with torch.no_grad():
            fake, _ = G([z[0].cuda()], randomize_noise=False)
        d_loss = TR.d_logistic_loss(D(real.cuda()), D(fake))
        TR.StyleGAN2Trainer._step(D, d_optim, d_loss)
        x = real.cuda().requires_grad_(True)
        rp = D(x)
        r1 = TR.d_r1_loss(rp, x)
        TR.StyleGAN2Trainer._step(D, d_optim, 10.0 / 2 * r1 * 16 + 0 * rp[0])
        TR.requires_grad(G, True), TR.requires_grad(D, False)
        fake, _ = G([z[1].cuda(), z[2].cuda()], inject_index=2, randomize_noise=False)
        g_loss = TR.g_nonsaturating_loss(D(fake))
        TR.StyleGAN2Trainer._step(G, g_optim, g_loss)
        fake, lat = G([z[3][:2].cuda()], return_latents=True, randomize_noise=False)
        pl, mp, _ = TR.g_path_regularize(fake, lat, mp, noise=pl_noise.cuda())
        TR.StyleGAN2Trainer._step(G, g_optim, 2.0 * 4 * pl + 0 * fake[0, 0, 0, 0])
        losses.append([float(d_loss), float(r1), float(g_loss), float(pl)])
    cpu = lambda net: {k: v.detach().cpu() for k, v in net.state_dict().items()}
    return losses, float(mp), cpu(G), cpu(D)


def rel_err(p, ref):
    num = den = 0.0
    for k, v in p.items():
        if k in ref:
            num += float((v.double() - ref[k].double()).pow(2).sum())
            den += float(ref[k].double().pow(2).sum())
    return (num / den) ** 0.5


def main():
    size, batch, iters = (int(a) for a in (sys.argv[1:4] + ["8", "4", "3"][len(sys.argv) - 1:]))
    truth = oracle_run(torch.float64, size, batch, iters)
    runs = [("oracle fp32 (CPU)", oracle_run(torch.float32, size, batch, iters))]
    if torch.cuda.is_available():
        runs += [("engine, implicit GEMM", engine_run(False, size, batch, iters)),
                 ("engine, Winograd 3x3", engine_run(True, size, batch, iters))]
    print(f"size {size}, batch {batch}: deviation from the float64 oracle, (a - b) / max(1, |b|)")
    print(f"{'':24s}" + "".join(f"| it{i} d       r1      g       path    " for i in range(iters)) + "| mean_path  G params  D params")
    for name, (losses, mp, g, d) in runs:
        row = f"{name:24s}"
        for la, lb in zip(losses, truth[0]):
            row += "| " + " ".join(f"{abs(a - b) / max(1.0, abs(b)):7.1e}" for a, b in zip(la, lb)) + " "
        row += f"| {abs(mp - truth[1]) / abs(truth[1]):7.1e}    {rel_err(g, truth[2]):7.1e}   {rel_err(d, truth[3]):7.1e}"
        print(row, flush=True)


if __name__ == "__main__":
    main()
