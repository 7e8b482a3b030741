"""GPU: the HIP fused_bias_act / upfirdn2d ops (reference operator API, diagan/models/op) against the
reference-generated goldens and the oracle, including first- and second-order autograd."""
import os

import numpy as np
import pytest
import torch

from oracle import stylegan_ops as S

pytestmark = pytest.mark.gpu


def test_upfirdn2d_forward_backward_vs_reference(golden_dir):
    from diagan.models.op import upfirdn2d
    from diagan.models.op.upfirdn2d import UpFirDn2d
    g = np.load(os.path.join(golden_dir, "stylegan_ops.npz"))
    for name in [str(n) for n in g["names"]]:
        x = torch.from_numpy(g[f"{name}_x"]).cuda().requires_grad_(True)
        k = torch.from_numpy(g[f"{name}_k"]).cuda()
        u, d, px0, px1, py0, py1 = [int(v) for v in g[f"{name}_cfg"]]
        y = UpFirDn2d.apply(x, k, (u, u), (d, d), (px0, px1, py0, py1))
        assert tuple(y.shape) == g[f"{name}_y"].shape, name
        np.testing.assert_allclose(y.detach().cpu().numpy(), g[f"{name}_y"], atol=1e-5, err_msg=name)
        (y * torch.from_numpy(g[f"{name}_cot"]).cuda()).sum().backward()
        np.testing.assert_allclose(x.grad.cpu().numpy(), g[f"{name}_gx"], atol=1e-5, err_msg=name)
    # the public helper (symmetric pads) = StyleGAN2's Upsample / Blur / Downsample call forms
    x = torch.randn(2, 4, 16, 16, device="cuda")
    k = torch.from_numpy(g["upsample2_k"]).cuda()
    ref = S.upfirdn2d(x.cpu(), k.cpu(), 2, 2, 1, 1, 2, 1, 2, 1)
    np.testing.assert_allclose(upfirdn2d(x, k, up=2, down=1, pad=(2, 1)).cpu().numpy(), ref.numpy(), atol=1e-5)


def test_upfirdn2d_double_backward():
    """R1 / path-length regularisation differentiate the gradient: grad-of-grad vs autograd on the oracle."""
    from diagan.models.op import upfirdn2d
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 2, 6, 6, generator=g)
    k = torch.rand(4, 4, generator=g)
    w = torch.randn(1, 2, 12, 12, generator=g)

    def run(fn, x, k, w):
        x = x.clone().requires_grad_(True)
        y = fn(x, k)
        gx, = torch.autograd.grad((y * w).sum() + (y ** 2).sum(), x, create_graph=True)
        ggx, = torch.autograd.grad((gx ** 2).sum(), x)
        return gx.detach().cpu(), ggx.cpu()

    a = run(lambda x, k: upfirdn2d(x, k, up=2, down=1, pad=(2, 1)), x.cuda(), k.cuda(), w.cuda())
    b = run(lambda x, k: S.upfirdn2d(x, k, 2, 2, 1, 1, 2, 1, 2, 1), x, k, w)
    np.testing.assert_allclose(a[0].numpy(), b[0].numpy(), atol=1e-4)
    np.testing.assert_allclose(a[1].numpy(), b[1].numpy(), atol=1e-3)


def test_fused_leaky_relu_vs_reference(golden_dir):
    from diagan.models.op import FusedLeakyReLU, fused_leaky_relu
    g = np.load(os.path.join(golden_dir, "stylegan_ops.npz"))
    for tag in ("4d", "2d"):
        x = torch.from_numpy(g[f"flr_{tag}_x"]).cuda().requires_grad_(True)
        b = torch.from_numpy(g[f"flr_{tag}_b"]).cuda().requires_grad_(True)
        y = fused_leaky_relu(x, b, 0.2, 2 ** 0.5)
        np.testing.assert_allclose(y.detach().cpu().numpy(), g[f"flr_{tag}_y"], atol=1e-6)
        (y * torch.from_numpy(g[f"flr_{tag}_cot"]).cuda()).sum().backward()
        np.testing.assert_allclose(x.grad.cpu().numpy(), g[f"flr_{tag}_gx"], atol=1e-6)
        np.testing.assert_allclose(b.grad.cpu().numpy(), g[f"flr_{tag}_gb"], atol=1e-5)
    y = fused_leaky_relu(torch.from_numpy(g["flr_nobias_x"]).cuda(), None, 0.2, 2 ** 0.5)
    np.testing.assert_allclose(y.cpu().numpy(), g["flr_nobias_y"], atol=1e-6)
    m = FusedLeakyReLU(8).cuda()
    assert m(torch.randn(2, 8, 3, 3, device="cuda")).shape == (2, 8, 3, 3)


def test_fused_bias_act_table_and_second_order():
    from diagan.models.op.fused_act import fused_bias_act, fused_leaky_relu
    g = torch.Generator().manual_seed(2)
    x, ref, b = torch.randn(3, 5, 4, 4, generator=g), torch.randn(3, 5, 4, 4, generator=g), torch.randn(5, generator=g)
    for act, grad in ((1, 0), (1, 1), (1, 2), (3, 0), (3, 1), (3, 2)):
        got = fused_bias_act(x.cuda(), b.cuda(), ref.cuda(), act, grad, 0.3, 1.7)
        exp = S.fused_bias_act(x, b, ref, act, grad, 0.3, 1.7)
        np.testing.assert_allclose(got.cpu().numpy(), exp.numpy(), atol=1e-6, err_msg=str((act, grad)))
    with pytest.raises(RuntimeError):
        fused_bias_act(x, b, ref, 3, 0, 0.2, 1.0)            # CPU tensor: the reference's CHECK_CUDA

    def run(fn, x, b):
        x, b = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = fn(x, b)
        gx, = torch.autograd.grad((y ** 2).sum(), x, create_graph=True)
        ggb, = torch.autograd.grad((gx ** 2).sum(), b)
        return gx.detach().cpu(), ggb.cpu()

    a = run(lambda x, b: fused_leaky_relu(x, b, 0.2, 2 ** 0.5), x.cuda(), b.cuda())
    c = run(lambda x, b: S.fused_leaky_relu(x, b, 0.2, 2 ** 0.5), x, b)
    np.testing.assert_allclose(a[0].numpy(), c[0].numpy(), atol=1e-4)
    np.testing.assert_allclose(a[1].numpy(), c[1].numpy(), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("shape,kshape,pad", [
    ((2, 9, 11, 8), (4, 4), (2, 1)),        # StyleGAN2 blur, ragged width (11 + 3 - 4 + 1 = 11 columns: 2 full groups + 3)
    ((1, 6, 6, 128), (4, 4), (1, 1)),
    ((3, 7, 5, 4), (3, 4), (2, 2)),         # rectangular filter
    ((2, 10, 10, 12), (4, 4), (-1, 2)),     # cropping pad
    ((1, 1, 1, 4), (4, 4), (3, 3)),         # single pixel: every tap out of range somewhere
])
def test_channels_last_fir_fast_path_vs_oracle(shape, kshape, pad):
    """the 4-channel x 4-column register-window kernel that serves up = down = 1 on [B, H, W, C] activations"""
    from diagan.models.op.upfirdn2d import upfirdn2d_nhwc
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g)
    k = torch.randn(*kshape, generator=g)
    xg = x.cuda().requires_grad_(True)
    y = upfirdn2d_nhwc(xg, k.cuda(), pad=pad)
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    ref = S.upfirdn2d(xr, k, 1, 1, 1, 1, pad[0], pad[1], pad[0], pad[1])
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.detach().permute(0, 2, 3, 1).numpy(), atol=1e-5)
    cot = torch.randn(ref.shape, generator=g)
    (ref * cot).sum().backward()
    (y * cot.permute(0, 2, 3, 1).cuda()).sum().backward()          # adjoint: the same fast path, flipped filter
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.permute(0, 2, 3, 1).numpy(), atol=1e-5)


def test_channels_last_bias_act_fast_path_vs_oracle():
    from diagan.models.op.fused_act import fused_bias_act, fused_leaky_relu
    g = torch.Generator().manual_seed(9)
    for shape in ((2, 5, 5, 8), (3, 12), (1, 3, 3, 516)):
        x = torch.randn(*shape, generator=g)
        b = torch.randn(shape[-1], generator=g)
        r = torch.randn(*shape, generator=g)
        to_cf = (lambda t: t.permute(0, 3, 1, 2)) if len(shape) == 4 else (lambda t: t)
        back = (lambda t: t.permute(0, 2, 3, 1)) if len(shape) == 4 else (lambda t: t)
        for act, grad, ref_t in ((3, 0, None), (3, 1, r), (1, 0, None), (3, 2, r)):
            want = S.fused_bias_act(to_cf(x), b, to_cf(ref_t) if ref_t is not None else None, act, grad, 0.2, 1.5)
            got = fused_bias_act(x.cuda(), b.cuda(), ref_t.cuda() if ref_t is not None else None, act, grad, 0.2, 1.5,
                                 bias_dim=-1)
            np.testing.assert_allclose(got.cpu().numpy(), back(want).numpy(), atol=1e-6, err_msg=f"{shape} {act}{grad}")
        xg, bg = x.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
        xr, br = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
        (fused_leaky_relu(xg, bg, bias_dim=-1) ** 2).sum().backward()
        (S.fused_leaky_relu(to_cf(xr), br) ** 2).sum().backward()
        np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), atol=1e-5)
        np.testing.assert_allclose(bg.grad.cpu().numpy(), br.grad.numpy(), rtol=1e-4, atol=1e-4)


def test_styled_bias_act_first_and_second_order():
    """the one-launch StyledConv tail against its three-pass composition, values and gradients to second order"""
    from diagan.models.op.fused_act import styled_bias_act
    g = torch.Generator().manual_seed(4)
    B, H, W, C = 3, 5, 6, 8
    vals = dict(x=torch.randn(B, H, W, C, generator=g), d=torch.rand(B, C, generator=g) + 0.5,
                s=torch.randn(1, generator=g), b=torch.randn(C, generator=g))
    for noise in (torch.randn(B, H, W, 1, generator=g), torch.randn(1, H, W, 1, generator=g)):
        def run(fused, dev):
            t = {k: v.to(dev).double().requires_grad_(True) if not fused else v.to(dev).requires_grad_(True)
                 for k, v in vals.items()}
            n = noise.to(dev) if fused else noise.to(dev).double()
            if fused:
                y = styled_bias_act(t['x'], t['d'], n, t['s'], t['b'])
            else:
                pre = t['x'] * t['d'][:, None, None, :] + t['s'] * n + t['b']
                y = torch.nn.functional.leaky_relu(pre, 0.2) * 2 ** 0.5
            cot = torch.sin(torch.arange(y.numel(), device=dev, dtype=y.dtype).view(y.shape))
            first = torch.autograd.grad((y * cot).sum(), list(t.values()), create_graph=True)
            second = torch.autograd.grad(sum((f ** 2).sum() for f in first), [t['x'], t['d']])
            return [v.detach().cpu().double() for v in (y, *first, *second)]
        ours, ref = run(True, "cuda"), run(False, "cpu")
        for i, (a, b) in enumerate(zip(ours, ref)):
            np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=2e-4, atol=2e-4, err_msg=f"output {i}")
    # optional operands
    x = vals['x'].cuda()
    np.testing.assert_allclose(styled_bias_act(x).cpu().numpy(),
                               (torch.nn.functional.leaky_relu(vals['x'], 0.2) * 2 ** 0.5).numpy(), atol=1e-6)


@pytest.mark.parametrize("shape", [(2, 4, 4, 4), (3, 9, 7, 64), (2, 33, 31, 128), (1, 5, 5, 512), (2, 3, 3, 12)])
def test_rowdot_and_scale_rows_closed_pair(shape):
    """S(x, s) = x * s[n, c] and R(a, b) = sum_pixels a * b (one-pass HIP kernel; torch fallback for channel counts
    that are not a power of two): values and gradients to second order against plain torch in float64"""
    from diagan.models.op.fused_act import rowdot, scale_rows
    g = torch.Generator().manual_seed(sum(shape))
    x0, a0 = torch.randn(*shape, generator=g), torch.randn(*shape, generator=g)
    s0 = torch.randn(shape[0], shape[3], generator=g)

    def run(fused, dev, dt):
        x, a, s = (t.to(dev, dt).requires_grad_(True) for t in (x0, a0, s0))
        y = scale_rows(x, s) if fused else x * s[:, None, None, :]
        r = rowdot(y, a) if fused else (y * a).sum((1, 2))
        first = torch.autograd.grad((r ** 2).sum() + (y ** 3).sum(), (x, a, s), create_graph=True)
        second = torch.autograd.grad(sum((f ** 2).sum() for f in first), (x, a, s))
        return [t.detach().cpu().double() for t in (y, r, *first, *second)]

    ours, ref = run(True, "cuda", torch.float32), run(False, "cpu", torch.float64)
    for i, (p, q) in enumerate(zip(ours, ref)):
        np.testing.assert_allclose(p.numpy(), q.numpy(), rtol=1e-3, atol=1e-3 * float(q.abs().max()), err_msg=f"output {i}")


@pytest.mark.parametrize("shape", [(3, 5, 6, 8), (2, 33, 31, 128), (4, 16, 16, 512), (2, 7, 9, 4)])
@pytest.mark.parametrize("noise_kind", ["per_image", "shared", "none"])
def test_styled_bias_act_one_pass_first_order_backward(shape, noise_kind):
    """round 4: a plain backward (no graph recorded) of the StyledConv tail / of bias + leaky ReLU takes ONE pass over the
    incoming gradient (diagan_styled_bias_act_bwd: gate, demodulated gradient, d(demod), d(bias), d(noise strength)); it must
    equal the float64 composition, and the differentiable path that R1 / path-length penalties use (create_graph=True)."""
    from diagan.models.op import fused_act as FA
    B, H, W, C = shape
    g = torch.Generator().manual_seed(sum(shape))
    vals = dict(x=torch.randn(B, H, W, C, generator=g), d=torch.rand(B, C, generator=g) + 0.5, s=torch.randn(1, generator=g),
                b=torch.randn(C, generator=g))
    noise = {"per_image": torch.randn(B, H, W, 1, generator=g), "shared": torch.randn(1, H, W, 1, generator=g),
             "none": None}[noise_kind]
    cot = torch.sin(torch.arange(B * H * W * C, dtype=torch.float64).view(B, H, W, C))

    def ref():
        t = {k: v.double().requires_grad_(True) for k, v in vals.items()}
        pre = t['x'] * t['d'][:, None, None, :] + t['b']
        if noise is not None:
            pre = pre + t['s'] * noise.double()
        y = torch.nn.functional.leaky_relu(pre, 0.2) * 2 ** 0.5
        (y * cot).sum().backward()
        return [t[k].grad for k in ('x', 'd', 'b')] + ([t['s'].grad] if noise is not None else [])

    def ours(create_graph):
        t = {k: v.cuda().requires_grad_(True) for k, v in vals.items()}
        y = FA.styled_bias_act(t['x'], t['d'], noise.cuda() if noise is not None else None,
                               t['s'] if noise is not None else None, t['b'])
        keys = ['x', 'd', 'b'] + (['s'] if noise is not None else [])
        gr = torch.autograd.grad((y * cot.cuda().float()).sum(), [t[k] for k in keys], create_graph=create_graph)
        return [v.detach().double().cpu() for v in gr]

    want, fast, slow = ref(), ours(False), ours(True)
    for i, (a, b, c) in enumerate(zip(fast, want, slow)):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-6, f"one-pass backward, gradient {i}"
        assert float((c - b).abs().max()) <= 2e-5 * scale + 1e-6, f"differentiable backward, gradient {i}"
    # bias + leaky ReLU alone (FusedLeakyReLU of the discriminator's convolutions), channels-last
    x, b = vals['x'].cuda().requires_grad_(True), vals['b'].cuda().requires_grad_(True)
    (FA.fused_leaky_relu(x, b, bias_dim=-1) * cot.cuda().float()).sum().backward()
    xr, br = vals['x'].double().requires_grad_(True), vals['b'].double().requires_grad_(True)
    (torch.nn.functional.leaky_relu(xr + br, 0.2) * 2 ** 0.5 * cot).sum().backward()
    assert float((x.grad.double().cpu() - xr.grad).abs().max()) <= 1e-6
    assert float((b.grad.double().cpu() - br.grad).abs().max()) <= 2e-5 * float(br.grad.abs().max()) + 1e-6


@pytest.mark.parametrize("shape,pad", [((3, 9, 11, 8), (2, 2)), ((2, 33, 31, 128), (2, 2)), ((2, 16, 16, 512), (1, 1)), ((1, 5, 7, 4), (2, 1))])
def test_activation_folded_into_the_blur_and_into_the_residual_add(shape, pad):
    """round 6 (models/op/fused_tail.py): blur(leaky_relu(z + bias) * scale) in the blur's pass and leaky_relu(z + bias) * scale + r in
    one pass are BIT-identical to the two launches each replaces -- values, first-order gradients (plain backward: the one-pass gate
    from the pre-activation) and the gradients of a gradient penalty (create_graph: the differentiable composition)."""
    from diagan.models.op import fused_act as FA, fused_tail as FT
    from diagan.models.op.upfirdn2d import upfirdn2d_nhwc
    B, H, W, C = shape
    g = torch.Generator().manual_seed(sum(shape))
    k1 = torch.tensor([1.0, 3.0, 3.0, 1.0])
    kern = (torch.outer(k1, k1) / 64).cuda()
    z0, b0 = torch.randn(B, H, W, C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    z0[0, 0, 0, 0] = -b0[0]                      # a pre-activation of exactly zero takes the slope branch in both forms

    def two(z, b):
        return upfirdn2d_nhwc(FA.fused_leaky_relu(z, b, 0.2, 1.3, bias_dim=-1), kern, pad=pad)

    def one(z, b):
        return FT.bias_act_blur(z, b, kern, pad, 0.2, 1.3)

    def run(f, second):
        z, b = z0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        y = f(z, b)
        cot = torch.cos(torch.arange(y.numel(), device="cuda", dtype=torch.float32)).view(y.shape)
        if not second:
            gz, gb = torch.autograd.grad((y * cot).sum(), [z, b])
            return y.detach(), gz, gb
        # a gradient penalty (R1's shape): the backward is differentiated with respect to the cotangent it was given
        w = torch.ones_like(y, requires_grad=True)
        g1, = torch.autograd.grad((y * w * cot).sum(), [z], create_graph=True)
        gw, = torch.autograd.grad((g1 * g1).sum(), [w])
        return y.detach(), g1.detach(), gw
    for second in (False, True):
        for i, (p, q) in enumerate(zip(run(one, second), run(two, second))):
            if not second and i == 2:      # the bias gradient: the same numbers added block by block in another (fixed) order
                assert float((p - q).abs().max()) <= 2e-6 * float(q.abs().max()) + 1e-6
                continue
            assert torch.equal(p, q), f"bias_act_blur second={second} output {i}: max diff {(p - q).abs().max().item():.3e}"
    a1, a2 = run(one, False), run(one, False)
    assert all(torch.equal(u, v) for u, v in zip(a1, a2)), "the one-pass backward must repeat bit for bit"

    r0 = torch.randn(B, H, W, C, generator=g).cuda()

    def two_add(z, b, r):
        return FA.fused_leaky_relu(z, b, 0.2, 0.9, bias_dim=-1) + r

    def one_add(z, b, r):
        return FT.bias_act_add(z, b, r, 0.2, 0.9)

    def run_add(f, second):
        z, b, r = (t.clone().requires_grad_(True) for t in (z0, b0, r0))
        y = f(z, b, r)
        cot = torch.cos(torch.arange(y.numel(), device="cuda", dtype=torch.float32)).view(y.shape)
        if not second:
            return (y.detach(),) + torch.autograd.grad((y * cot).sum(), [z, b, r])
        w = torch.ones_like(y, requires_grad=True)
        g1, = torch.autograd.grad((y * w).sum(), [z], create_graph=True)
        gw, = torch.autograd.grad((g1 * g1 * cot).sum(), [w])
        return y.detach(), g1.detach(), gw
    for second in (False, True):
        for i, (p, q) in enumerate(zip(run_add(one_add, second), run_add(two_add, second))):
            assert torch.equal(p, q), f"bias_act_add second={second} output {i}: max diff {(p - q).abs().max().item():.3e}"


@pytest.mark.parametrize("shape", [(3, 11, 13, 8), (2, 35, 33, 128), (2, 19, 19, 512)])
@pytest.mark.parametrize("noise_kind", ["per_image", "shared", "none"])
@pytest.mark.parametrize("post", [False, True])
def test_styled_tail_folded_into_the_blur_without_a_graph(shape, noise_kind, post):
    """round 6: the generator's up-sampling StyledConv when nothing records a graph -- Blur, demodulation, noise, bias, leaky ReLU and
    the next layer's style in ONE pass (diagan_fir_styled_act) -- bit-identical to upfirdn2d + styled_bias_act + scale_rows"""
    from diagan.models.op import fused_act as FA, fused_tail as FT
    from diagan.models.op.upfirdn2d import upfirdn2d_nhwc
    B, H, W, C = shape
    g = torch.Generator().manual_seed(sum(shape))
    k1 = torch.tensor([1.0, 3.0, 3.0, 1.0])
    kern = (torch.outer(k1, k1) / 16).cuda()
    pad = (1, 1)
    x = torch.randn(B, H, W, C, generator=g).cuda()
    d, s, b = (torch.rand(B, C, generator=g) + 0.5).cuda(), torch.randn(1, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    ps = torch.randn(B, C, generator=g).cuda() if post else None
    oh, ow = H + 2 - 4 + 1, W + 2 - 4 + 1
    noise = {"per_image": torch.randn(B, oh, ow, 1, generator=g), "shared": torch.randn(1, oh, ow, 1, generator=g), "none": None}[noise_kind]
    noise = noise.cuda() if noise is not None else None
    with torch.no_grad():
        assert FT.blur_styled_act_ok(x, kern)
        want = FA.styled_bias_act(upfirdn2d_nhwc(x, kern, pad=pad), d, noise, s if noise is not None else None, b)
        if post:
            want = FA.scale_rows(want, ps)
        got = FT.blur_styled_act(x, kern, pad, d, noise, s if noise is not None else None, b, post=ps)
    assert torch.equal(got, want), (got - want).abs().max().item()
    assert not FT.blur_styled_act_ok(x.requires_grad_(True), kern) or not torch.is_grad_enabled()


@pytest.mark.parametrize("shape", [(3, 5, 6, 8), (2, 33, 31, 128), (2, 16, 16, 512), (2, 9, 7, 256), (1, 3, 3, 1024), (2, 4, 4, 4)])
def test_torgb_in_one_pass(shape):
    """round 6: ToRGB (modulation, 1x1 convolution to 3 planes, bias) as ONE read of its input; the plain backward as one read + one
    write (diagan_torgb_bwd); a differentiated backward (create_graph) as the composition of the convolution ops -- all against float64"""
    from diagan.models.op import fused_tail as FT
    B, H, W, C = shape
    g = torch.Generator().manual_seed(sum(shape))
    vals = dict(x=torch.randn(B, H, W, C, generator=g), s=torch.randn(B, C, generator=g) + 1.0, w=torch.randn(3, C, generator=g),
                b=torch.randn(3, generator=g))
    scale = C ** -0.5
    cot = torch.sin(torch.arange(B * H * W * 3, dtype=torch.float64).view(B, H, W, 3))
    pen_w = torch.cos(torch.arange(B * H * W * C, dtype=torch.float64).view(B, H, W, C))

    def ref(second):
        t = {k: v.double().requires_grad_(True) for k, v in vals.items()}
        y = torch.einsum('bhwc,oc->bhwo', t['x'] * t['s'][:, None, None, :], t['w'] * scale) + t['b']
        if not second:
            return [y.detach()] + list(torch.autograd.grad((y * cot).sum(), [t[k] for k in 'xswb']))
        gx, = torch.autograd.grad((y * cot).sum(), [t['x']], create_graph=True)
        return [y.detach()] + list(torch.autograd.grad((gx.square() * pen_w).sum(), [t['s'], t['w']]))

    def ours(second):
        t = {k: v.cuda().requires_grad_(True) for k, v in vals.items()}
        y = FT.torgb(t['x'], t['s'], t['w'], t['b'], scale)
        assert y.shape == (B, H, W, 4) and float(y.detach()[..., 3].abs().max()) == 0.0
        c4 = torch.nn.functional.pad(cot, (0, 1)).float().cuda()
        if not second:
            gr = torch.autograd.grad((y * c4).sum(), [t[k] for k in 'xswb'])
        else:
            gx, = torch.autograd.grad((y * c4).sum(), [t['x']], create_graph=True)
            gr = torch.autograd.grad((gx.square() * pen_w.float().cuda()).sum(), [t['s'], t['w']])
        return [y[..., :3].detach().double().cpu()] + [v.detach().double().cpu() for v in gr]

    for second in (False, True):
        for i, (a, b) in enumerate(zip(ours(second), ref(second))):
            sc = float(b.abs().max()) + 1e-12
            assert float((a - b).abs().max()) <= 3e-5 * sc + 1e-6, f"second={second} output {i}: {float((a - b).abs().max()):.3e} of {sc:.3e}"


@pytest.mark.parametrize("shape", [(2, 16, 16, 8), (2, 33, 31, 128), (1, 10, 12, 512)])
def test_fork_into_a_filter_accumulates_in_the_filters_pass(shape):
    """round 6: a tensor with two consumers, one of them a resampling filter (ResBlock's input): the node's plain backward adds the other
    consumer's gradient inside the adjoint filter's pass -- bit-identical to the adjoint followed by autograd's accumulation; the
    differentiated backward stays the differentiable sum"""
    from diagan.models.op import fused_tail as FT
    from diagan.models.op.upfirdn2d import upfirdn2d_nhwc
    B, H, W, C = shape
    g = torch.Generator().manual_seed(sum(shape))
    k1 = torch.tensor([1.0, 3.0, 3.0, 1.0])
    kern = (torch.outer(k1, k1) / 64).cuda()
    x0 = torch.randn(B, H, W, C, generator=g).cuda()

    def two(x):
        return x, upfirdn2d_nhwc(x, kern, down=2, pad=(1, 1))

    def one(x):
        return FT.fork_fir(x, kern, down=2, pad=(1, 1))

    def run(f, second):
        x = x0.clone().requires_grad_(True)
        a, d = f(x * 1.0)
        ca = torch.cos(torch.arange(a.numel(), device="cuda", dtype=torch.float32)).view(a.shape)
        cd = torch.sin(torch.arange(d.numel(), device="cuda", dtype=torch.float32)).view(d.shape)
        if not second:
            gx, = torch.autograd.grad((a.square() * ca).sum() + (d * cd).sum(), [x])
            return a.detach(), d.detach(), gx
        wd = torch.ones_like(d, requires_grad=True)
        g1, = torch.autograd.grad((a * ca).sum() + (d * wd * cd).sum(), [x], create_graph=True)
        gw, = torch.autograd.grad(g1.square().sum(), [wd])
        return a.detach(), d.detach(), g1.detach(), gw
    for second in (False, True):
        for i, (p, q) in enumerate(zip(run(one, second), run(two, second))):
            assert torch.equal(p, q), f"second={second} output {i}: max diff {(p - q).abs().max().item():.3e}"


@pytest.mark.parametrize("shape", [(3, 5, 6, 8), (2, 33, 31, 128), (2, 16, 16, 512)])
@pytest.mark.parametrize("noise_kind", ["per_image", "shared", "none"])
@pytest.mark.parametrize("both", [True, False])
def test_styled_tail_that_leaves_the_next_layers_modulated_input(shape, noise_kind, both):
    """round 6: (y, y * post) from one launch and, in a plain backward, ONE pass for gate, gradient, d(demod), d(bias), d(strength),
    d(post) from the two incoming gradients (`both` False: y has no consumer of its own) -- element-wise results bit-identical to
    styled_bias_act followed by scale_rows, the per-channel sums to rounding; the differentiated backward against the same"""
    from diagan.models.op import fused_act as FA
    B, H, W, C = shape
    g = torch.Generator().manual_seed(sum(shape) + both)
    vals = dict(x=torch.randn(B, H, W, C, generator=g), d=torch.rand(B, C, generator=g) + 0.5, s=torch.randn(1, generator=g),
                b=torch.randn(C, generator=g), p=torch.randn(B, C, generator=g) + 1.0)
    noise = {"per_image": torch.randn(B, H, W, 1, generator=g), "shared": torch.randn(1, H, W, 1, generator=g), "none": None}[noise_kind]
    noise = noise.cuda() if noise is not None else None
    c1 = torch.sin(torch.arange(B * H * W * C, dtype=torch.float32)).view(B, H, W, C).cuda()
    c2 = torch.cos(torch.arange(B * H * W * C, dtype=torch.float32) * 0.7).view(B, H, W, C).cuda()
    keys = ['x', 'd', 'b', 'p'] + (['s'] if noise is not None else [])

    def run(fused, second):
        t = {k: v.cuda().requires_grad_(True) for k, v in vals.items()}
        st = t['s'] if noise is not None else None
        if fused:
            y, ym = FA.styled_bias_act_mod(t['x'], t['d'], noise, st, t['b'], t['p'])
        else:
            y = FA.styled_bias_act(t['x'], t['d'], noise, st, t['b'])
            ym = FA.scale_rows(y, t['p'])
        loss = (ym * c2).sum() + ((y * c1).sum() if both else 0.0)
        if not second:
            return [y.detach(), ym.detach()] + list(torch.autograd.grad(loss, [t[k] for k in keys]))
        gx, = torch.autograd.grad(loss, [t['x']], create_graph=True)
        return [gx.detach()] + list(torch.autograd.grad((gx.square() * c1).sum(), [t['d'], t['p']]))

    for second in (False, True):
        for i, (a, b) in enumerate(zip(run(True, second), run(False, second))):
            exact = (not second and i < 3) or (second and i == 0)          # y, ym, gx: element-wise
            if exact:
                assert torch.equal(a, b), f"second={second} output {i}: {(a - b).abs().max().item():.3e}"
            else:
                assert float((a - b).abs().max()) <= 3e-5 * float(b.abs().max()) + 1e-5, f"second={second} output {i}"


@pytest.mark.parametrize("shape", [(3, 5, 6, 8), (2, 33, 31, 128), (2, 16, 16, 256), (1, 7, 9, 4)])
@pytest.mark.parametrize("need_gx", [True, False])
def test_fromrgb_in_one_pass(shape, need_gx):
    """round 6: the discriminator's first ConvLayer (1x1 convolution from RGB, bias, leaky ReLU * sqrt 2) as one write of its output; the
    plain backward as one read of gy and y; a differentiated backward (R1: gradient with respect to the images, then through it) as
    gate + convolution ops -- against float64"""
    from diagan.models.op import fused_tail as FT
    B, H, W, C = shape
    g = torch.Generator().manual_seed(sum(shape))
    vals = dict(x=torch.nn.functional.pad(torch.randn(B, H, W, 3, generator=g), (0, 1)), w=torch.randn(C, 3, 1, 1, generator=g),
                b=torch.randn(C, generator=g))
    wscale = 3 ** -0.5
    cot = torch.sin(torch.arange(B * H * W * C, dtype=torch.float64).view(B, H, W, C))

    def ref(second):
        t = {k: v.double().requires_grad_(True) for k, v in vals.items()}
        y = torch.nn.functional.leaky_relu(torch.einsum('bhwi,ci->bhwc', t['x'][..., :3], t['w'].view(C, 3) * wscale) + t['b'], 0.2) * 2 ** 0.5
        if not second:
            gr = torch.autograd.grad((y * cot).sum(), [t['w'], t['b']] + ([t['x']] if need_gx else []))
            return [y.detach()] + [v if v.shape[-1] != 4 or v.dim() != 4 else v[..., :3] for v in gr]
        gx, = torch.autograd.grad((y * cot).sum(), [t['x']], create_graph=True)
        return [y.detach()] + list(torch.autograd.grad(gx[..., :3].square().sum(), [t['w']]))

    def ours(second):
        t = {k: v.cuda().requires_grad_(k != 'x' or need_gx or second) for k, v in vals.items()}
        assert FT.fromrgb_ok(t['x'], t['w'], t['b'])
        y = FT.fromrgb(t['x'], t['w'], t['b'], wscale)
        if not second:
            gr = torch.autograd.grad((y * cot.float().cuda()).sum(), [t['w'], t['b']] + ([t['x']] if need_gx else []))
            gr = [v if v.dim() != 4 or v.shape[-1] != 4 else v[..., :3] for v in gr]
        else:
            gx, = torch.autograd.grad((y * cot.float().cuda()).sum(), [t['x']], create_graph=True)
            gr = torch.autograd.grad(gx[..., :3].square().sum(), [t['w']])
        return [y.detach().double().cpu()] + [v.detach().double().cpu() for v in gr]

    for second in (False, True):
        for i, (a, b) in enumerate(zip(ours(second), ref(second))):
            sc = float(b.abs().max()) + 1e-12
            assert float((a - b).abs().max()) <= 3e-5 * sc + 1e-6, f"second={second} output {i}: {float((a - b).abs().max()):.3e} of {sc:.3e}"


@pytest.mark.parametrize("B,K,C,lr_mul", [(32, 512, 512, 1.0), (16, 512, 128, 1.0), (5, 64, 3, 0.01), (40, 512, 1, 1.0)])
def test_small_linear_in_one_launch(B, K, C, lr_mul):
    """round 6: EqualLinear without activation (the styles' modulation layers) as one launch, plain backward one launch, a differentiated
    backward through the recomputed differentiable form -- against float64"""
    from diagan.models.op import fused_tail as FT
    g = torch.Generator().manual_seed(B + K + C)
    vals = dict(x=torch.randn(B, K, generator=g), w=torch.randn(C, K, generator=g), b=torch.randn(C, generator=g))
    scale = K ** -0.5 * lr_mul
    cot = torch.sin(torch.arange(B * C, dtype=torch.float64).view(B, C))

    def run(dev, dt, second):
        t = {k: v.to(dev, dt).requires_grad_(True) for k, v in vals.items()}
        if dev == "cuda":
            y = FT.mod_linear(t['x'], t['w'], t['b'], scale, lr_mul)
        else:
            y = t['x'] @ (t['w'] * scale).t() + t['b'] * lr_mul
        c = cot.to(dev, dt)
        if not second:
            return [y.detach()] + list(torch.autograd.grad((y * c).sum(), [t['x'], t['w'], t['b']]))
        gx, = torch.autograd.grad((y * c).sum(), [t['x']], create_graph=True)
        return [y.detach()] + list(torch.autograd.grad(gx.square().sum(), [t['w']]))
    for second in (False, True):
        for i, (a, b) in enumerate(zip(run("cuda", torch.float32, second), run("cpu", torch.float64, second))):
            sc = float(b.abs().max()) + 1e-12
            assert float((a.double().cpu() - b).abs().max()) <= 2e-5 * sc + 1e-6, f"second={second} output {i}"


@pytest.mark.parametrize("B,Ci,Co,k", [(32, 512, 512, 3), (16, 256, 128, 3), (3, 8, 12, 1), (7, 64, 4, 3)])
def test_demodulation_in_one_launch(B, Ci, Co, k):
    """round 6: d = rsqrt(scale^2 * (s^2) @ (sum_taps w^2).T + eps) as one launch forward, one backward (d(s), d(w)); the differentiated
    backward through the recomputed differentiable form -- against float64"""
    from diagan.models.op import fused_tail as FT
    g = torch.Generator().manual_seed(B + Ci + Co)
    vals = dict(s=torch.randn(B, Ci, generator=g) + 1.0, w=torch.randn(Co, Ci, k, k, generator=g))
    scale2, eps = 1.0 / (Ci * k * k), 1e-8
    cot = torch.sin(torch.arange(B * Co, dtype=torch.float64).view(B, Co))

    def run(dev, dt, second):
        t = {kk: v.to(dev, dt).requires_grad_(True) for kk, v in vals.items()}
        if dev == "cuda":
            assert FT.demod_ok(t['s'], t['w'])
            d = FT.demod(t['s'], t['w'], scale2, eps)
        else:
            d = torch.rsqrt(t['s'].square() @ t['w'].square().sum((2, 3)).t() * scale2 + eps)
        c = cot.to(dev, dt)
        if not second:
            return [d.detach()] + list(torch.autograd.grad((d * c).sum(), [t['s'], t['w']]))
        gs, = torch.autograd.grad((d * c).sum(), [t['s']], create_graph=True)
        return [d.detach()] + list(torch.autograd.grad(gs.square().sum(), [t['w'], t['s']]))
    for second in (False, True):
        for i, (a, b) in enumerate(zip(run("cuda", torch.float32, second), run("cpu", torch.float64, second))):
            sc = float(b.abs().max()) + 1e-12
            assert float((a.double().cpu() - b).abs().max()) <= 5e-5 * sc + 1e-6, f"second={second} output {i}: {float((a.double().cpu() - b).abs().max()):.3e} of {sc:.3e}"
