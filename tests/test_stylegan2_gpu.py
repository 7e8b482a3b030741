"""GPU: the StyleGAN2 row -- the any-order-differentiable convolution ops over the HIP GEMM
kernels, and StyleGANGenerator / StyleGANDiscriminator with the trainer's losses and second-order regularisers,
against the reference-generated vectors (tests/golden/stylegan2.npz) and the pinned oracle (oracle/stylegan2.py).
Tolerance: 1e-3 relative (fp32; the engine sums in MFMA tile order and evaluates the modulated convolution
activation-side)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import stylegan2 as O

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "stylegan2.npz"))
SIZE = int(GOLD["size"])
T = lambda k: torch.from_numpy(GOLD[k]).cuda()


def close(a, b, rtol=1e-3, what=""):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol * max(np.abs(b).max(), 1e-30), err_msg=what)


# ---- ops ---------------------------------------------------------------------------------------------------------
CASES = [  # kind, B, H, W, Ci, Co, k, stride, pad
    ("conv", 2, 8, 8, 32, 64, 3, 1, 1),
    ("conv", 3, 11, 9, 16, 32, 3, 2, 0),        # blurred odd-sized map into the stride-2 convolution
    ("conv", 2, 9, 9, 32, 16, 1, 2, 0),         # the residual skip
    ("conv", 2, 10, 12, 16, 16, 3, 2, 0),       # last input row / column never read: zero data gradient there
    ("conv", 4, 4, 4, 32, 8, 4, 1, 0),          # flatten + linear as a 4x4 convolution
    ("conv", 2, 6, 6, 4, 32, 1, 1, 0),          # from RGB (3 planes + 1 zero plane)
    ("convT", 2, 4, 4, 32, 64, 3, 2, 0),
    ("convT", 3, 8, 8, 64, 32, 3, 2, 0),
]


def _ref_op(kind, x, w, stride, pad):
    if kind == "conv":
        return F.conv2d(x, w, stride=stride, padding=pad)
    return F.conv_transpose2d(x, w.transpose(0, 1), stride=stride, padding=pad)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(map(str, c)))
def test_diffconv_first_and_second_order(case):
    """y, dy/dx, dy/dw and the gradients OF a function of the input gradient (R1's structure) and of the weight
    gradient, against torch autograd over F.conv2d / F.conv_transpose2d on the CPU in float64."""
    from diagan.ops import diffconv as dc
    kind, B, H, W, Ci, Co, k, stride, pad = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x0 = torch.randn(B, Ci, H, W, generator=g, dtype=torch.float64)
    w0 = torch.randn(Co, Ci, k, k, generator=g, dtype=torch.float64) / (Ci * k * k) ** 0.5

    def run(x, w, op, nhwc):
        x = x.clone().requires_grad_(True)
        w = w.clone().requires_grad_(True)
        y = op(x.permute(0, 2, 3, 1).contiguous() if nhwc else x, w)
        if nhwc:
            y = y.permute(0, 3, 1, 2)
        cot = torch.cos(torch.arange(y.numel(), dtype=torch.float64).view(y.shape)).to(y)
        gx, gw = torch.autograd.grad((y * cot).sum() + 0.5 * (y ** 2).sum(), (x, w), create_graph=True)
        penalty = (gx ** 2).sum() + (gw ** 2).sum()
        ggx, ggw = torch.autograd.grad(penalty, (x, w))
        return [t.detach().cpu().double() for t in (y, gx, gw, ggx, ggw)]

    ours = run(x0.float().cuda(), w0.float().cuda(),
               (lambda x, w: dc.conv2d(x, w, stride, pad)) if kind == "conv" else
               (lambda x, w: dc.conv_transpose2d(x, w, stride, pad)), True)
    ref = run(x0, w0, lambda x, w: _ref_op(kind, x, w, stride, pad), False)
    for name, a, b in zip(("y", "dx", "dw", "d(penalty)/dx", "d(penalty)/dw"), ours, ref):
        close(a, b, what=f"{case} {name}")


def test_linear_matches_torch():
    from diagan.ops import diffconv as dc
    g = torch.Generator().manual_seed(3)
    x = torch.randn(5, 512, generator=g)
    w = torch.randn(7, 512, generator=g)
    close(dc.linear(x.cuda(), w.cuda()), F.linear(x.double(), w.double()))
    x = torch.randn(3, 10, generator=g)
    w = torch.randn(6, 10, generator=g)
    close(dc.linear(x.cuda(), w.cuda()), F.linear(x.double(), w.double()))


# ---- model -------------------------------------------------------------------------------------------------------
def build(kind, seed):
    from diagan.models import stylegan2 as M
    net = (M.StyleGANGenerator if kind == "g" else M.StyleGANDiscriminator)(size=SIZE)
    shapes = O.generator_shapes(SIZE) if kind == "g" else O.discriminator_shapes(SIZE)
    # names and shapes are the reference's: the oracle's inventory loads strictly (FIR kernels are buffers the
    # module builds itself, exactly as in the reference)
    missing, unexpected = net.load_state_dict(O.seeded_state(shapes, seed), strict=False)
    assert not unexpected and all(k.endswith("kernel") for k in missing), (missing, unexpected)
    return net.cuda()


def grad_norms(net):
    return {k: float(p.grad.double().norm()) for k, p in net.named_parameters()}


def check_norms(tag, net, rtol=2e-3):
    ours = grad_norms(net)
    keys = [str(k) for k in GOLD[f"{tag}_keys"]]
    assert sorted(ours) == keys
    got, want = np.array([ours[k] for k in keys]), GOLD[f"{tag}_norms"]
    # a NoiseInjection strength is ONE scalar whose gradient sums B*H*W*C products of either sign (524 288 terms
    # here, result ~1e-2): its value depends on the fp32 summation order at the 1e-3..1e-2 level on any device
    tol = lambda k: 1e-2 if k.endswith("noise.weight") else rtol
    bad = [(k, a, b) for k, a, b in zip(keys, got, want) if abs(a - b) > tol(k) * abs(b) + 1e-6]
    assert not bad, f"{tag}: gradient norms off: {bad}"


def test_generator_forward_vs_reference():
    G = build("g", int(GOLD["seed_g"]))
    with torch.no_grad():
        img, lat = G([T("z1")], return_latents=True, randomize_noise=False)
        mix, none = G([T("z1"), T("z2")], inject_index=3, randomize_noise=False)
    assert none is None and img.shape == (4, 3, SIZE, SIZE) and img.is_contiguous()
    close(lat, GOLD["g_latent"], what="latent")
    close(img, GOLD["g_image"], what="image")
    close(mix, GOLD["g_image_mix"], what="style mixing")
    # random noise path: draws fresh noise every call
    with torch.no_grad():
        a, _ = G([T("z1")])
        b, _ = G([T("z1")])
    assert not torch.equal(a, b)


def test_discriminator_loss_and_gradients_vs_reference():
    from diagan.trainer import stylegan2 as TR
    D = build("d", int(GOLD["seed_d"]))
    D.zero_grad()
    rp, fp = D(T("d_real")), D(T("d_fake"))
    close(rp, GOLD["d_real_pred"], what="real logits")
    close(fp, GOLD["d_fake_pred"], what="fake logits")
    loss = TR.d_logistic_loss(rp, fp)
    close(loss, GOLD["d_loss"])
    loss.backward()
    check_norms("d_loss_grad", D)
    close(D.final_linear[1].weight.grad, GOLD["d_loss_grad_last"])
    # the gradients live in the flat slab the fused Adam steps over
    assert D.final_linear[1].weight.grad.data_ptr() >= D.flat_grads.data_ptr()
    assert float(D.flat_grads.double().norm()) > 0


def test_r1_second_order_vs_reference():
    from diagan.trainer import stylegan2 as TR
    D = build("d", int(GOLD["seed_d"]))
    D.zero_grad()
    x = T("d_real").clone().requires_grad_(True)
    rp = D(x)
    r1 = TR.d_r1_loss(rp, x)
    close(r1, GOLD["r1"])
    (10.0 / 2 * r1 * 16 + 0 * rp[0]).backward()
    check_norms("r1_grad", D)
    close(D.convs[0][0].weight.grad, GOLD["r1_grad_first"])


def test_generator_loss_and_path_length_vs_reference():
    from diagan.trainer import stylegan2 as TR
    G, D = build("g", int(GOLD["seed_g"])), build("d", int(GOLD["seed_d"]))
    TR.requires_grad(D, False)
    G.zero_grad()
    fake, _ = G([T("z1")], randomize_noise=False)
    loss = TR.g_nonsaturating_loss(D(fake))
    close(loss, GOLD["g_loss"])
    loss.backward()
    check_norms("g_loss_grad", G)
    close(G.to_rgbs[-1].bias.grad, GOLD["g_loss_grad_rgb_bias"])

    G.zero_grad()
    fake, lat = G([T("zp")], return_latents=True, randomize_noise=False)
    pl, mean_path, lengths = TR.g_path_regularize(fake, lat, 0.3, noise=T("pl_noise"))
    close(lengths, GOLD["path_lengths"])
    close(pl, GOLD["path_loss"])
    close(mean_path, GOLD["mean_path"])
    (2.0 * 4 * pl + 0 * fake[0, 0, 0, 0]).backward()
    check_norms("path_grad", G)
    close(G.input.input.grad, GOLD["path_grad_input"])


def test_other_resolutions_vs_oracle():
    """size 8 and 32 (different pyramid depths; batch 5 = a ragged minibatch-stddev group of 1... the reference
    requires batch % group == 0, so 6 with group 4 is invalid there too: use 4 and 8)"""
    from diagan.models import stylegan2 as M
    for size, batch in ((8, 4), (32, 8)):
        G, D = M.StyleGANGenerator(size=size), M.StyleGANDiscriminator(size=size)
        sg = O.seeded_state(O.generator_shapes(size), 21)
        sd_ = O.seeded_state(O.discriminator_shapes(size), 22)
        G.load_state_dict(sg, strict=False), D.load_state_dict(sd_, strict=False)
        G.cuda(), D.cuda()
        z = torch.randn(batch, 512, generator=torch.Generator().manual_seed(size))
        with torch.no_grad():
            img, _ = G([z.cuda()], randomize_noise=False)
            ref, _ = O.generator(sg, size, [z])
            close(img, ref, what=f"G size {size}")
            close(D(img), O.discriminator(sd_, size, ref), what=f"D size {size}")


def test_training_trajectory_vs_oracle():
    """Three full iterations (D step, R1 step, G step, path-length step; Adam with the reference's lazy-regularisation
    hyper-parameters) on the HIP engine with FusedAdam over the flat slabs, against the same iterations on the oracle
    with torch.optim.Adam: per-iteration losses and the final parameters."""
    from diagan.models import stylegan2 as M
    from diagan.trainer import stylegan2 as TR
    size, batch, iters = 8, 4, 3
    gen = torch.Generator().manual_seed(77)
    sg = O.seeded_state(O.generator_shapes(size), 31)
    sd_ = O.seeded_state(O.discriminator_shapes(size), 32)
    G, D = M.StyleGANGenerator(size=size), M.StyleGANDiscriminator(size=size)
    G.load_state_dict(sg, strict=False), D.load_state_dict(sd_, strict=False)
    G.cuda(), D.cuda()
    g_optim, d_optim = TR.make_optimizers(G, D, lr=0.002, g_reg_every=4, d_reg_every=16)

    pg = {k: v.clone().requires_grad_(not k.startswith("noises.")) for k, v in sg.items()}
    pd = {k: v.clone().requires_grad_(True) for k, v in sd_.items()}
    gr, dr = 4 / 5, 16 / 17
    og = torch.optim.Adam([v for v in pg.values() if v.requires_grad], lr=0.002 * gr, betas=(0 ** gr, 0.99 ** gr))
    od = torch.optim.Adam(list(pd.values()), lr=0.002 * dr, betas=(0 ** dr, 0.99 ** dr))

    def step_ref(opt, loss):
        opt.zero_grad()
        loss.backward()
        opt.step()

    mean_path, mean_path_ref = 0, 0
    for it in range(iters):
        real = torch.rand(batch, 3, size, size, generator=gen) * 2 - 1
        z = [torch.randn(batch, 512, generator=gen) for _ in range(4)]
        pl_noise = torch.randn(2, 3, size, size, generator=gen)
        # ---- engine ----
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
        pl, mean_path, _ = TR.g_path_regularize(fake, lat, mean_path, noise=pl_noise.cuda())
        TR.StyleGAN2Trainer._step(G, g_optim, 2.0 * 4 * pl + 0 * fake[0, 0, 0, 0])
        # ---- oracle ----
        with torch.no_grad():
            fake_r, _ = O.generator(pg, size, [z[0]])
        d_loss_r = O.d_logistic_loss(O.discriminator(pd, size, real), O.discriminator(pd, size, fake_r))
        step_ref(od, d_loss_r)
        xr = real.clone().requires_grad_(True)
        rpr = O.discriminator(pd, size, xr)
        r1_r = O.d_r1_loss(rpr, xr)
        step_ref(od, 10.0 / 2 * r1_r * 16 + 0 * rpr[0])
        fake_r, _ = O.generator(pg, size, [z[1], z[2]], inject_index=2)
        frozen = {k: v.detach() for k, v in pd.items()}
        g_loss_r = O.g_nonsaturating_loss(O.discriminator(frozen, size, fake_r))
        step_ref(og, g_loss_r)
        fake_r, lat_r = O.generator(pg, size, [z[3][:2]])
        pl_r, mean_path_ref, _ = O.g_path_regularize(fake_r, lat_r, mean_path_ref, pl_noise)
        step_ref(og, 2.0 * 4 * pl_r + 0 * fake_r[0, 0, 0, 0])
        for name, a, b in (("d", d_loss, d_loss_r), ("r1", r1, r1_r), ("g", g_loss, g_loss_r), ("path", pl, pl_r)):
            assert abs(float(a) - float(b)) <= 5e-3 * max(1.0, abs(float(b))), (it, name, float(a), float(b))
    assert abs(float(mean_path) - float(mean_path_ref)) < 1e-3 * abs(float(mean_path_ref))

    def rel_err(net, ref):
        num = den = 0.0
        for k, v in net.state_dict().items():
            if k in ref:
                num += float((v.cpu().double() - ref[k].detach().double()).pow(2).sum())
                den += float(ref[k].detach().double().pow(2).sum())
        return (num / den) ** 0.5
    # Adam's first steps move every weight by ~lr * sign(gradient): elements whose gradient is within rounding of
    # zero may step the other way, so the comparison is a global relative L2 error, not element-wise
    assert rel_err(D, pd) < 1e-3 and rel_err(G, pg) < 1e-3
    assert float((D.state_dict()["final_linear.1.weight"].cpu() - sd_["final_linear.1.weight"]).abs().max()) > 1e-3


# ---- BASELINE configs[4]'s resolution: 256 x 256 (VERDICT r1 Missing 7) ----------------------------------------------
# tests/golden/stylegan2_256.npz comes from the reference's own classes at size 256, channel_multiplier 1, batch 2
# (tools/gen_goldens_stylegan2.py 256): the upper pyramid layers have Ci != Co (512 -> 256 -> 128 -> 64), which the
# size-16 / 32 vectors never reach.  Images are stored as checksums + strided samples (test_oracle_models.expand_check).
G256_PATH = os.path.join(os.path.dirname(__file__), "golden", "stylegan2_256.npz")
needs_256 = pytest.mark.skipif(not os.path.exists(G256_PATH), reason="tests/golden/stylegan2_256.npz not generated")


def _build256(kind, g256):
    from diagan.models import stylegan2 as M
    size, cm = int(g256["size"]), int(g256["channel_multiplier"])
    net = (M.StyleGANGenerator if kind == "g" else M.StyleGANDiscriminator)(size=size, channel_multiplier=cm)
    shapes = O.generator_shapes(size, mult=cm) if kind == "g" else O.discriminator_shapes(size, mult=cm)
    missing, unexpected = net.load_state_dict(O.seeded_state(shapes, int(g256["seed_g" if kind == "g" else "seed_d"])), strict=False)
    assert not unexpected and all(k.endswith("kernel") for k in missing), (missing, unexpected)
    return net.cuda()


def _check_norms256(g256, tag, net, rtol=2e-3):
    ours = grad_norms(net)
    keys = [str(k) for k in g256[f"{tag}_keys"]]
    assert sorted(ours) == keys
    got, want = np.array([ours[k] for k in keys]), g256[f"{tag}_norms"]
    # a NoiseInjection strength is ONE scalar: the sum of B*H*W*C products g * noise of either sign (up to 8 M terms,
    # |sum| down to 1/5000 of the sum of |terms|), so an fp32 evaluation is only good to some 1e-5 of the sum of |terms|
    # on any device.  Against the oracle in float64 (tools/sg2_noise_grad.py, profiles/r02_sg2_winograd.md) the
    # reference's own CPU values -- this fixture -- are off by up to 27e-6 of it, this engine by up to 28e-6 on either
    # convolution path (<= 10e-6 on all but the well-conditioned sums, where rtol covers it).  The fixture records the
    # sum of |terms| (`*_noise_abs`, from the reference with the strength expanded per element); the allowance next to
    # rtol is 5e-5 of it: the two measured deviations added, since fixture and engine may err to opposite sides.
    if f"{tag}_noise_abs" in g256.files:
        term_sum = dict(zip((str(k) for k in g256[f"{tag}_noise_keys"]), g256[f"{tag}_noise_abs"]))
        tol = lambda k, b: rtol * abs(b) + (5e-5 * term_sum[k] if k in term_sum else 1e-6)
    else:
        tol = lambda k, b: (1e-2 * abs(b) + 1e-4) if k.endswith("noise.weight") else (rtol * abs(b) + 1e-6)
    bad = [(k, a, b) for k, a, b in zip(keys, got, want) if abs(a - b) > tol(k, b)]
    assert not bad, f"{tag}: gradient norms off: {bad}"


@needs_256
def test_256_generator_discriminator_losses_and_regularisers_vs_reference():
    from test_oracle_models import expand_check
    from diagan.trainer import stylegan2 as TR
    g = np.load(G256_PATH)
    size, batch = int(g["size"]), int(g["batch"])
    G, D = _build256("g", g), _build256("d", g)
    z1, z2 = torch.from_numpy(g["z1"]).cuda(), torch.from_numpy(g["z2"]).cuda()
    with torch.no_grad():
        img, lat = G([z1], return_latents=True, randomize_noise=False)
        mix, _ = G([z1, z2], inject_index=5, randomize_noise=False)
    assert img.shape == (batch, 3, size, size)
    close(lat[:, :2], g["g_latent"], what="latent")
    close(img[:, :, :8, :8], g["g_image_corner"], what="image corner")
    expand_check(img.cpu().numpy(), g["g_image"], 1e-3)
    expand_check(mix.cpu().numpy(), g["g_image_mix"], 1e-3)
    # the "real" batch is re-drawn from its seed (a 393 216-value tensor is not stored); its checksum is
    gen = torch.Generator().manual_seed(7)
    torch.randn(batch, 512, generator=gen), torch.randn(batch, 512, generator=gen)                 # z1, z2
    x = torch.randn(batch, 3, size, size, generator=gen).clamp_(-2, 2) * 0.5
    expand_check(x.numpy(), g["d_real_check"], 1e-6)
    D.zero_grad()
    rp, fp = D(x.cuda()), D(img.detach())
    close(rp, g["d_real_pred"], what="real logits")
    close(fp, g["d_fake_pred"], what="fake logits")
    d_loss = TR.d_logistic_loss(rp, fp)
    close(d_loss, g["d_loss"])
    d_loss.backward()
    _check_norms256(g, "d_loss_grad", D)
    D.zero_grad()
    xr = x.cuda().requires_grad_(True)
    rp = D(xr)
    r1 = TR.d_r1_loss(rp, xr)
    close(r1, g["r1"])
    (10.0 / 2 * r1 * 16 + 0 * rp[0]).backward()
    _check_norms256(g, "r1_grad", D)
    TR.requires_grad(D, False)
    G.zero_grad()
    fake, _ = G([z1], randomize_noise=False)
    g_loss = TR.g_nonsaturating_loss(D(fake))
    close(g_loss, g["g_loss"])
    g_loss.backward()
    _check_norms256(g, "g_loss_grad", G)
    G.zero_grad()
    zp = torch.randn(1, 512, generator=gen)
    pl_noise = torch.randn(1, 3, size, size, generator=gen)
    np.testing.assert_array_equal(zp.numpy(), g["zp"])
    expand_check(pl_noise.numpy(), g["pl_noise_check"], 1e-6)
    fake, lat = G([zp.cuda()], return_latents=True, randomize_noise=False)
    pl, mean_path, lengths = TR.g_path_regularize(fake, lat, 0.3, noise=pl_noise.cuda())
    close(lengths, g["path_lengths"])
    close(pl, g["path_loss"])
    (2.0 * 4 * pl + 0 * fake[0, 0, 0, 0]).backward()
    _check_norms256(g, "path_grad", G)


# ---- BASELINE configs[4] as it is trained: 256 x 256, channel_multiplier 2, batch 32 (stylegan2/train_ffhq.py:393,442) --------
@pytest.mark.timeout(1800)
def test_real_configuration_256_cm2_batch32_vs_oracle():
    """The configuration the reference trains: at batch 32 every convolution takes the launch sizes of the
    real run (the large Winograd / F(4x4) launches, the parity-split transposed gathers at 128^2 and 256^2), which the
    channel_multiplier-1 / batch-2 fixture never reaches.  The pinned oracle (oracle/stylegan2.py: float32 PyTorch on the host,
    stated the reference's way with per-sample grouped convolutions) evaluates the SAME 32 latents and images: generator
    images, discriminator logits (whose minibatch-stddev layer couples the samples in groups of 4, so the whole batch is
    compared), both losses; then one full training iteration of the HIP engine at this configuration with R1 and
    path-length regularisation must leave finite parameters that moved."""
    from diagan.models import stylegan2 as M
    from diagan.trainer import stylegan2 as TR
    size, cm, batch = 256, 2, 32
    sg = O.seeded_state(O.generator_shapes(size, mult=cm), 41)
    sd_ = O.seeded_state(O.discriminator_shapes(size, mult=cm), 42)
    G, D = M.StyleGANGenerator(size=size, channel_multiplier=cm), M.StyleGANDiscriminator(size=size, channel_multiplier=cm)
    G.load_state_dict(sg, strict=False), D.load_state_dict(sd_, strict=False)
    G.cuda(), D.cuda()
    gen = torch.Generator().manual_seed(256)
    z = torch.randn(batch, 512, generator=gen)
    x = torch.randn(batch, 3, size, size, generator=gen).clamp_(-2, 2) * 0.5
    torch.set_num_threads(max(torch.get_num_threads(), 16))
    with torch.no_grad():
        img, _ = G([z.cuda()], randomize_noise=False)
        ref, _ = O.generator(sg, size, [z])
        assert img.shape == (batch, 3, size, size)
        close(img, ref, what="generator images, 256^2 cm 2 batch 32")
        fp, rp = D(img), D(x.cuda())
        fp_ref, rp_ref = O.discriminator(sd_, size, ref), O.discriminator(sd_, size, x)
        close(fp, fp_ref, what="fake logits")
        close(rp, rp_ref, what="real logits")
        close(TR.d_logistic_loss(rp, fp), O.d_logistic_loss(rp_ref, fp_ref))
        close(TR.g_nonsaturating_loss(fp), O.g_nonsaturating_loss(fp_ref))
    # one full iteration of the trainer at this configuration (D step + R1, G step + path length): finite, and the step was taken
    import types
    g_ema = M.StyleGANGenerator(size=size, channel_multiplier=cm).cuda().eval()
    TR.accumulate(g_ema, G, 0)
    g_optim, d_optim = TR.make_optimizers(G, D)
    a = types.SimpleNamespace(iter=10 ** 9, start_iter=0, batch=batch, latent=512, mixing=0.9, r1=10.0, d_reg_every=1,
                              g_reg_every=1, path_regularize=2.0, path_batch_shrink=2, logit_save_steps=10 ** 9,
                              save_logit_after=10 ** 9, stop_save_logit_after=0, n_sample=16, augment=False)
    ds = torch.utils.data.TensorDataset(x.cuda(), torch.arange(batch, device='cuda'))
    loader = torch.utils.data.DataLoader(ds, batch_size=batch, shuffle=False, drop_last=True)
    tr = TR.StyleGAN2Trainer(a, loader, G, D, g_optim, d_optim, g_ema, torch.device('cuda'), "/tmp/diagan_test_sg2_real")
    w0, d0 = G.flat_params.clone(), D.flat_params.clone()
    tr.train_step(0)
    torch.cuda.synchronize()
    assert torch.isfinite(G.flat_params).all() and torch.isfinite(D.flat_params).all()
    assert float((G.flat_params - w0).abs().max()) > 1e-4 and float((D.flat_params - d0).abs().max()) > 1e-4
    assert all(torch.isfinite(v).all() for v in (tr.r1_loss, tr.path_loss, tr.path_lengths))

