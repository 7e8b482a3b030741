"""The StyleGAN2 oracle (oracle/stylegan2.py) against the vectors the reference's own classes produced on CPU
(tests/golden/stylegan2.npz, tools/gen_goldens_stylegan2.py): forward passes, losses, and the second-order
gradients of R1 and of the path-length penalty."""
import os

import numpy as np
import pytest
import torch

from oracle import stylegan2 as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "stylegan2.npz"))
SIZE = int(G["size"])
T = lambda k: torch.from_numpy(G[k])


def params(shapes, seed, grad=True):
    sd = O.seeded_state(shapes, seed)
    for k, v in sd.items():
        if grad and not k.startswith("noises."):
            v.requires_grad_(True)
    return sd


def norms(sd):
    ks = sorted(k for k, v in sd.items() if v.grad is not None)
    return ks, np.array([float(sd[k].grad.double().norm()) for k in ks])


def check_norms(tag, sd, rtol=2e-4):
    ks, ns = norms(sd)
    assert ks == list(G[f"{tag}_keys"])
    np.testing.assert_allclose(ns, G[f"{tag}_norms"], rtol=rtol, atol=1e-7)


def test_state_inventory_matches_reference_modules():
    # the golden file lists every parameter that received a gradient: all of them
    assert set(G["g_loss_grad_keys"]) == {k for k in O.generator_shapes(SIZE) if not k.startswith("noises.")}
    assert set(G["d_loss_grad_keys"]) == set(O.discriminator_shapes(SIZE))


def test_generator_forward_and_mixing():
    sd = params(O.generator_shapes(SIZE), int(G["seed_g"]), grad=False)
    with torch.no_grad():
        img, lat = O.generator(sd, SIZE, [T("z1")])
        mix, _ = O.generator(sd, SIZE, [T("z1"), T("z2")], inject_index=3)
    np.testing.assert_allclose(lat.numpy(), G["g_latent"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(img.numpy(), G["g_image"], rtol=1e-3, atol=1e-3 * np.abs(G["g_image"]).max())
    np.testing.assert_allclose(mix.numpy(), G["g_image_mix"], rtol=1e-3, atol=1e-3 * np.abs(G["g_image_mix"]).max())


def test_discriminator_loss_and_gradients():
    sd = params(O.discriminator_shapes(SIZE), int(G["seed_d"]))
    rp, fp = O.discriminator(sd, SIZE, T("d_real")), O.discriminator(sd, SIZE, T("d_fake"))
    np.testing.assert_allclose(rp.detach().numpy(), G["d_real_pred"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(fp.detach().numpy(), G["d_fake_pred"], rtol=1e-3, atol=1e-3)
    loss = O.d_logistic_loss(rp, fp)
    assert abs(loss.item() - float(G["d_loss"])) < 1e-3 * max(1.0, abs(float(G["d_loss"])))
    loss.backward()
    check_norms("d_loss_grad", sd)
    np.testing.assert_allclose(sd["final_linear.1.weight"].grad.numpy(), G["d_loss_grad_last"], rtol=1e-3, atol=1e-5)


def test_r1_second_order():
    sd = params(O.discriminator_shapes(SIZE), int(G["seed_d"]))
    x = T("d_real").clone().requires_grad_(True)
    rp = O.discriminator(sd, SIZE, x)
    r1 = O.d_r1_loss(rp, x)
    assert abs(r1.item() - float(G["r1"])) < 1e-3 * abs(float(G["r1"]))
    (10.0 / 2 * r1 * 16 + 0 * rp[0]).backward()
    check_norms("r1_grad", sd)
    ref = G["r1_grad_first"]
    np.testing.assert_allclose(sd["convs.0.0.weight"].grad.numpy(), ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())


def test_generator_loss_and_path_length_second_order():
    sg = params(O.generator_shapes(SIZE), int(G["seed_g"]))
    sdd = params(O.discriminator_shapes(SIZE), int(G["seed_d"]), grad=False)
    fake, _ = O.generator(sg, SIZE, [T("z1")])
    loss = O.g_nonsaturating_loss(O.discriminator(sdd, SIZE, fake))
    assert abs(loss.item() - float(G["g_loss"])) < 1e-3 * max(1.0, abs(float(G["g_loss"])))
    loss.backward()
    check_norms("g_loss_grad", sg, rtol=1e-3)
    np.testing.assert_allclose(sg["to_rgbs.1.bias"].grad.numpy(), G["g_loss_grad_rgb_bias"], rtol=1e-3, atol=1e-6)

    for v in sg.values():
        v.grad = None
    fake, lat = O.generator(sg, SIZE, [T("zp")])
    pl, mean_path, lengths = O.g_path_regularize(fake, lat, 0.3, T("pl_noise"))
    np.testing.assert_allclose(lengths.detach().numpy(), G["path_lengths"], rtol=1e-3)
    assert abs(pl.item() - float(G["path_loss"])) < 1e-3 * abs(float(G["path_loss"]))
    assert abs(mean_path.item() - float(G["mean_path"])) < 1e-5
    (2.0 * 4 * pl + 0 * fake[0, 0, 0, 0]).backward()
    check_norms("path_grad", sg, rtol=1e-3)
    ref = G["path_grad_input"]
    np.testing.assert_allclose(sg["input.input"].grad.numpy(), ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())


def test_oracle_at_256_with_unequal_channel_counts():
    """BASELINE configs[4]'s resolution: the oracle's generator / discriminator forward and the logistic loss at 256 x 256,
    channel_multiplier 1 (Ci != Co layers 512 -> 256 -> 128 -> 64) against the reference's own classes
    (tests/golden/stylegan2_256.npz, tools/gen_goldens_stylegan2.py 256)."""
    from test_oracle_models import expand_check
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "stylegan2_256.npz"))
    size, cm = int(g["size"]), int(g["channel_multiplier"])
    sg = O.seeded_state(O.generator_shapes(size, mult=cm), int(g["seed_g"]))
    sd = O.seeded_state(O.discriminator_shapes(size, mult=cm), int(g["seed_d"]))
    assert set(g["g_loss_grad_keys"]) == {k for k in sg if not k.startswith("noises.")}
    assert set(g["d_loss_grad_keys"]) == set(sd)
    with torch.no_grad():
        img, lat = O.generator(sg, size, [torch.from_numpy(g["z1"])])
        np.testing.assert_allclose(img[:, :, :8, :8].numpy(), g["g_image_corner"], rtol=1e-4, atol=1e-5)
        expand_check(img.numpy(), g["g_image"], 1e-4)
        np.testing.assert_allclose(O.discriminator(sd, size, img).numpy(), g["d_fake_pred"], rtol=1e-4, atol=1e-5)
