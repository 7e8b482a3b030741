"""GPU: SNGAN-32 / SNGAN-64 forward, backward and train steps of the HIP engine against the
plain-PyTorch CPU oracle (oracle/nets.py) on identical weights, images and noise.
Tolerances: logits/losses 1e-3 absolute (BASELINE.json north_star), gradients 1e-3 relative to
their max magnitude."""
import copy

import pytest
import torch

from oracle import nets as O

pytestmark = pytest.mark.gpu


class Log:
    def __init__(self):
        self.m = {}

    def add_metric(self, name, value, group=None, precision=4):
        self.m[name] = value


def build(dataset, loss_type, seed=1):
    from diagan.models.predefined_models import get_gan_model
    oG, oD, ooptG, ooptD = O.make_pair(dataset, loss_type, seed=seed)
    torch.manual_seed(seed)
    netG, netD, optG, optD = get_gan_model(dataset, model='sngan', loss_type=loss_type)
    netG.load_state_dict(oG.state_dict())
    netD.load_state_dict(oD.state_dict())
    netG.to('cuda')
    netD.to('cuda')
    return (oG, oD, ooptG, ooptD), (netG, netD, optG, optD)


def relclose(a, b, tol, what=""):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    scale = b.abs().max().item() + 1e-20
    err = (a - b).abs().max().item()
    assert err <= tol * scale, f"{what}: max err {err:.3e}, scale {scale:.3e}"


def l2close(a, b, tol, what="", floor=0.0):
    """Relative L2 error.  Used for gradients that pass through ReLU masks: a pre-activation that is
    +1e-7 in one fp32 implementation and -1e-7 in the other flips one mask element (observed: 1 of
    262144), which is an O(1) error in a single element and ~1e-3 in L2 -- not a kernel error."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    err = (a - b).norm().item()
    assert err <= tol * (b.norm().item() + floor), f"{what}: L2 err {err:.3e}, norm {b.norm().item():.3e}"


def is_dead_bias(k):
    """Conv biases that feed a BatchNorm have an exactly-zero true gradient (BN removes the mean)."""
    return k.startswith('block') and k.endswith('.bias') and ('.c1.' in k or '.c2.' in k or '.c_sc.' in k)


def test_same_seed_init_matches_oracle():
    """RNG consumption order of the constructors replays torch_mimicry's (set_seed contract)."""
    from diagan.models.predefined_models import get_gan_model
    oG, oD, _, _ = O.make_pair('cifar10', 'ns', seed=7)
    torch.manual_seed(7)
    netG, netD, _, _ = get_gan_model('cifar10', model='sngan', loss_type='ns')
    for o, n in ((oG, netG), (oD, netD)):
        so, sn = o.state_dict(), n.state_dict()
        assert list(so.keys()) == list(sn.keys())
        for k in so:
            assert so[k].shape == sn[k].shape, k
            assert torch.equal(so[k], sn[k].cpu()), k
    assert netG.count_params() == sum(p.numel() for p in oG.parameters())
    assert netD.count_params() == sum(p.numel() for p in oD.parameters())


@pytest.mark.parametrize("dataset,res", [("cifar10", 32), ("celeba", 64)])
def test_forward_eval_and_train(dataset, res):
    (oG, oD, _, _), (netG, netD, _, _) = build(dataset, 'ns')
    B = 4
    g = torch.Generator().manual_seed(0)
    z = torch.randn(B, 128, generator=g)
    x = torch.rand(B, 3, res, res, generator=g) * 2 - 1
    for mode in ("eval", "train"):
        for n in (oG, oD, netG, netD):
            n.train(mode == "train")
        with torch.no_grad():
            ref_img, ref_logit = oG(z), oD(x)
        img, logit = netG(z.cuda()), netD(x.cuda())
        assert img.shape == ref_img.shape and logit.shape == ref_logit.shape
        assert (img.cpu() - ref_img).abs().max() < 1e-3, mode
        assert (logit.cpu() - ref_logit).abs().max() < 1e-3, mode
    # training-mode forwards updated BN running stats and SN u / sigma identically
    sG, sD = netG.state_dict(), netD.state_dict()
    for k, v in oG.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            relclose(sG[k].float(), v.float(), 1e-4, k)
    for k, v in oD.state_dict().items():
        if 'sn_' in k:
            relclose(sD[k], v, 1e-4, k)


@pytest.mark.parametrize("mode", ["fp32", "fp32-implicit-gemm"])
@pytest.mark.parametrize("dataset,res,loss", [("cifar10", 32, "ns"), ("cifar10", 32, "hinge"), ("celeba", 64, "ns")])
def test_train_steps_match_oracle(dataset, res, loss, mode):
    """two D + G updates against the CPU oracle in every arithmetic mode of the GEMM kernels: the default (fp32 MFMA,
    Winograd where it qualifies) and fp32 MFMA on the implicit GEMM only"""
    from diagan.ops import conv as C
    C.set_winograd(False if mode == "fp32-implicit-gemm" else None)
    try:
        _train_steps_match_oracle(dataset, res, loss)
    finally:
        C.set_winograd(None)


def _train_steps_match_oracle(dataset, res, loss):
    (oG, oD, ooptG, ooptD), (netG, netD, optG, optD) = build(dataset, loss)
    B = 8 if res == 32 else 4
    g = torch.Generator().manual_seed(3)
    for step in range(2):
        # step 0 starts from identical parameters: 1e-3 absolute (observed ~1e-6).  Later steps start
        # from parameters that already went through Adam with beta1 = 0 (update = lr * g/|g|), which
        # turns rounding noise on near-zero gradients into +-lr parameter differences: the loss
        # trajectories of ANY two fp32 implementations separate at that rate, so the bound is relative.
        tol = (lambda ref: 1e-3) if step == 0 else (lambda ref: 5e-3 * max(1.0, abs(ref)))
        x = torch.rand(B, 3, res, res, generator=g) * 2 - 1
        zd = torch.randn(B, 128, generator=g)
        zg = torch.randn(B, 128, generator=g)
        # ---- D step
        errD, D_x, D_Gz = oD.train_step((x, None), oG, ooptD, noise=zd)
        log = netD.train_step(real_batch=(x.cuda(), None), netG=netG, optD=optD, log_data=Log(), device='cuda',
                              noise=zd.cuda())
        assert abs(log.m['errD'].item() - errD) < tol(errD), (step, log.m['errD'].item(), errD)
        assert abs(log.m['D(x)'].item() - D_x) < 1e-3 and abs(log.m['D(G(z))'].item() - D_Gz) < 1e-3
        if step == 0:
            gr = netD.export_grads()
            for k, p in oD.named_parameters():
                l2close(gr[k], p.grad, 1e-2, f"D grad {k}")
        # ---- G step
        errG = oG.train_step((x, None), oD, ooptG, noise=zg)
        log = netG.train_step(real_batch=(x.cuda(), None), netD=netD, optG=optG, log_data=Log(), device='cuda',
                              noise=zg.cuda())
        assert abs(log.m['errG'].item() - errG) < tol(errG), (step, log.m['errG'].item(), errG)
        if step == 0:
            gr = netG.export_grads()
            wscale = max(p.grad.norm().item() for p in oG.parameters())
            for k, p in oG.named_parameters():
                if is_dead_bias(k):
                    assert gr[k].abs().max().item() < 1e-4 * wscale, k
                else:
                    l2close(gr[k], p.grad, 5e-2, f"G grad {k}")
    # after two Adam updates of each net the parameters still agree
    sG, sD = netG.state_dict(), netD.state_dict()
    for k, v in oG.state_dict().items():
        if v.dtype.is_floating_point:
            assert (sG[k].cpu() - v).abs().max() < 2e-3, k
    for k, v in oD.state_dict().items():
        assert (sD[k].cpu() - v).abs().max() < 2e-3, k


@pytest.mark.parametrize("dataset,res", [("cifar10", 32), ("celeba", 64)])
def test_full_batch_updates_match_oracle_through_the_fused_kernels(dataset, res):
    """One D and one G update at the REAL batch size (64; the D pair pass stacks 128 images) against the CPU oracle.  The
    small batches of the tests above never reach the launch sizes at which the engine switches to its fused launches --
    convolution + average pool in nine Winograd products (tile_cfg 11), its data gradient from the pooled gradient (12),
    the weight gradient over box sums, the shortcut's bilinear x2 blended into c2's epilogue -- so this is the test in
    which those run inside the networks; it asserts that they are indeed selected."""
    from diagan.ops import conv as C
    (oG, oD, ooptG, ooptD), (netG, netD, optG, optD) = build(dataset, "ns")
    B = 64
    c2 = netD.block1.c2
    relu = (C.PRO_RELU, None, None)
    if True:
        assert C.pool_fused(c2.geom, 2 * B, res, res, relu) and C.unpool_fused(c2.geom, 2 * B, res, res)
        last = netG.block4 if res == 32 else netG.block5             # the generator's last up-sampling block, at full size
        assert C.res_up_fused(last.c2.geom, B, res, res, want_stats=True)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, 3, res, res, generator=g) * 2 - 1
    zd, zg = torch.randn(B, 128, generator=g), torch.randn(B, 128, generator=g)
    errD, D_x, D_Gz = oD.train_step((x, None), oG, ooptD, noise=zd)
    log = netD.train_step(real_batch=(x.cuda(), None), netG=netG, optD=optD, log_data=Log(), device='cuda', noise=zd.cuda())
    assert abs(log.m['errD'].item() - errD) < 1e-3, (log.m['errD'].item(), errD)
    assert abs(log.m['D(x)'].item() - D_x) < 1e-3 and abs(log.m['D(G(z))'].item() - D_Gz) < 1e-3
    gr = netD.export_grads()
    for k, p in oD.named_parameters():
        l2close(gr[k], p.grad, 1e-2, f"D grad {k}")
    errG = oG.train_step((x, None), oD, ooptG, noise=zg)
    log = netG.train_step(real_batch=(x.cuda(), None), netD=netD, optG=optG, log_data=Log(), device='cuda', noise=zg.cuda())
    assert abs(log.m['errG'].item() - errG) < 5e-3 * max(1.0, abs(errG)), (log.m['errG'].item(), errG)
    gr = netG.export_grads()
    wscale = max(p.grad.norm().item() for p in oG.parameters())
    for k, p in oG.named_parameters():
        if is_dead_bias(k):
            assert gr[k].abs().max().item() < 1e-4 * wscale, k
        else:
            l2close(gr[k], p.grad, 5e-2, f"G grad {k}")


@pytest.fixture
def wino4_mode(request):
    """'off': the F(4x4,3x3) kernels (tile_cfg 13 / 15 and the pooled launches on them) and the split-operand implicit GEMM of
    round 5 (tile_cfg 16) are not selected -- the arithmetic the rounds before 3 made their parity claims with (F(2x2) + the exact
    fp32 implicit GEMM); restored afterwards"""
    from diagan.ops import conv as C
    if request.param == "off":
        C.set_winograd4(False)
        C.set_gemm_x3(False)
    yield request.param
    C.set_winograd4(None)
    C.set_gemm_x3(None)


@pytest.mark.parametrize("dataset,res,wino4_mode", [("cifar10", 32, "default"), ("celeba", 64, "default"), ("cifar10", 32, "off")],
                         indirect=["wino4_mode"])
def test_full_batch_gradients_against_float64(dataset, res, wino4_mode):
    """The same D and G updates at batch 64, measured against the TRUTH: oracle/nets.py evaluated in float64 from the same
    weights, images and noise.  Bounds are per-parameter relative L2 distances to the float64 gradient,
    set from what was measured on MI355X for BOTH fp32 implementations (tools/sngan_f64_parity.py,
    profiles/r03_f64_parity.md): D-update gradients HIP 2.3e-5 - 3.4e-4, CPU oracle in fp32 <= 3.4e-4 (the 3.4e-4 is ONE
    parameter, block1.c_sc.weight, where a single ReLU mask element within rounding of zero flips against float64: the fp32
    oracle flips it always, the HIP path with some kernel selections and not with others -- bound 1e-3 for both); G-update gradients (through D's and G's masks: a pre-activation within rounding of zero flips an O(1)
    mask element) HIP 2.2e-3 - 6.2e-3 depending on the kernel selection, fp32 oracle <= 6.0e-3; bound 1.2e-2 for both.  The fp32 oracle is run beside the HIP path and held to the same
    bounds, so the allowance is a statement about fp32, not about this engine; and over a whole network the HIP path may not
    be systematically further from float64 than plain PyTorch fp32 (rms over the parameters within a per-configuration bar,
    see `bar` below: 2x on SNGAN-32, 1x with the F(4x4) kernel off, 7x on SNGAN-64)."""
    (oG, oD, ooptG, ooptD), (netG, netD, optG, optD) = build(dataset, "ns")
    dG, dD = copy.deepcopy(oG).double(), copy.deepcopy(oD).double()
    doptG = torch.optim.Adam(dG.parameters(), 2e-4, betas=(0.0, 0.9))
    doptD = torch.optim.Adam(dD.parameters(), 2e-4, betas=(0.0, 0.9))
    # rms(HIP) / rms(fp32 oracle) over a network's parameters, per configuration (round 5: was 8 for all).  Measured on MI355X
    # (profiles/r05_trajectory.md, D / G update): SNGAN-32 default 1.01 / 1.04, SNGAN-32 without F(4x4) 0.002 / 0.31, SNGAN-64
    # default 5.1 / 4.8 (round 3: 6.9 on the G update; explained in profiles/r04_f64_parity.md -- transform rounding of the
    # Winograd launches plus K = 9216 single-accumulator chains on the small maps -- and 300x below the contract's 1e-3)
    bar = {("cifar10", "default"): 2.0, ("cifar10", "off"): 1.0, ("celeba", "default"): 7.0}[(dataset, wino4_mode)]
    B = 64
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, 3, res, res, generator=g) * 2 - 1
    zd, zg = torch.randn(B, 128, generator=g), torch.randn(B, 128, generator=g)

    def rel(a, b):
        return (a.detach().double().cpu() - b).norm().item() / (b.norm().item() + 1e-30)

    def rms(v):
        return (sum(e * e for e in v) / len(v)) ** 0.5

    e32 = oD.train_step((x, None), oG, ooptD, noise=zd)[0]
    e64 = dD.train_step((x.double(), None), dG, doptD, noise=zd.double())[0]
    log = netD.train_step(real_batch=(x.cuda(), None), netG=netG, optD=optD, log_data=Log(), device='cuda', noise=zd.cuda())
    assert abs(log.m['errD'].item() - e64) < 1e-5 and abs(e32 - e64) < 1e-5
    gr = netD.export_grads()
    hip, o32 = [], []
    for (k, p32), (_, p64) in zip(oD.named_parameters(), dD.named_parameters()):
        hip.append(rel(gr[k], p64.grad))
        o32.append(rel(p32.grad, p64.grad))
        assert hip[-1] < 1e-3, f"D grad {k}: {hip[-1]:.2e} from float64 (fp32 oracle: {o32[-1]:.2e})"
        assert o32[-1] < 1e-3, f"(oracle fp32) D grad {k}: {o32[-1]:.2e} from float64"
    print(f"f64-parity {dataset} {wino4_mode} D update: rms hip {rms(hip):.3e} oracle32 {rms(o32):.3e} ratio {rms(hip) / rms(o32):.3f}")
    assert rms(hip) <= bar * rms(o32) + 1e-6, (rms(hip), rms(o32))
    g32 = oG.train_step((x, None), oD, ooptG, noise=zg)
    g64 = dG.train_step((x.double(), None), dD, doptG, noise=zg.double())
    log = netG.train_step(real_batch=(x.cuda(), None), netD=netD, optG=optG, log_data=Log(), device='cuda', noise=zg.cuda())
    assert abs(log.m['errG'].item() - g64) < 1e-5 and abs(g32 - g64) < 1e-5
    gr = netG.export_grads()
    wscale = max(p.grad.norm().item() for p in dG.parameters())
    hip, o32 = [], []
    for (k, p32), (_, p64) in zip(oG.named_parameters(), dG.named_parameters()):
        if p64.grad.norm().item() < 1e-6 * wscale:          # conv biases in front of a BatchNorm: exactly-zero true gradient
            assert gr[k].abs().max().item() < 1e-4 * wscale, k
            continue
        hip.append(rel(gr[k], p64.grad))
        o32.append(rel(p32.grad, p64.grad))
        assert hip[-1] < 1.2e-2, f"G grad {k}: {hip[-1]:.2e} from float64 (fp32 oracle: {o32[-1]:.2e})"
        assert o32[-1] < 1.2e-2, f"(oracle fp32) G grad {k}: {o32[-1]:.2e} from float64"
    print(f"f64-parity {dataset} {wino4_mode} G update: rms hip {rms(hip):.3e} oracle32 {rms(o32):.3e} ratio {rms(hip) / rms(o32):.3f}")
    assert rms(hip) <= bar * rms(o32) + 1e-6, (rms(hip), rms(o32))


def test_generator_backward_isolated():
    """Same upstream gradient into both generators: no ReLU-flip noise from D on the path."""
    from diagan.ops import eltwise as E
    (oG, _, _, _), (netG, _, _, _) = build("cifar10", "ns")
    g = torch.Generator().manual_seed(11)
    z = torch.randn(8, 128, generator=g)
    gi = torch.randn(8, 3, 32, 32, generator=g)
    oG(z).backward(gi)
    netG.zero_grad()
    y, ctx = netG.forward_nhwc(z.cuda(), True, save=True)
    netG.backward_nhwc(ctx, E.nchw_to_nhwc(gi.cuda(), 4))
    gr = netG.export_grads()
    wscale = max(p.grad.abs().max().item() for p in oG.parameters())
    for k, p in oG.named_parameters():
        if is_dead_bias(k):
            assert gr[k].abs().max().item() < 1e-4 * wscale, k
        else:
            l2close(gr[k], p.grad, 1e-2, f"G grad {k}")   # ReLU-flip noise bound, see l2close


def test_discriminator_backward_isolated():
    from diagan.ops import eltwise as E
    (_, oD, _, _), (_, netD, _, _) = build("cifar10", "hinge")
    g = torch.Generator().manual_seed(12)
    x = torch.rand(8, 3, 32, 32, generator=g) * 2 - 1
    dl = torch.randn(8, generator=g)
    xr = x.clone().requires_grad_(True)
    oD(xr).view(-1).mul(dl).sum().backward()
    netD.zero_grad()
    logit, ctx = netD.forward_nhwc(E.nchw_to_nhwc(x.cuda(), 4), True, save=True, need_dgrad=True, need_in_dgrad=True)
    gx = netD.backward_nhwc(ctx, dl.cuda(), need_wgrad=True, need_gx=True)
    l2close(E.nhwc_to_nchw(gx, 3), xr.grad, 2e-2, "image gradient")
    gr = netD.export_grads()
    for k, p in oD.named_parameters():
        l2close(gr[k], p.grad, 1e-2, f"D grad {k}")
    # the deepest blocks see no flips at all: tight check that the kernels are exact there
    for k, p in oD.named_parameters():
        if k.startswith('block4') or k.startswith('l5'):
            relclose(gr[k], p.grad, 1e-4, f"D grad {k}")


def test_pair_forward_equals_two_pass():
    """The batched real+fake D pass (one GEMM per layer, per-half 1/sigma in the epilogue) gives the same
    update as two sequential forwards: same losses, same gradients, same SN buffers."""
    import copy
    (_, _, _, _), (netG, netD, optG, optD) = build("cifar10", "hinge")
    g = torch.Generator().manual_seed(21)
    x = (torch.rand(8, 3, 32, 32, generator=g) * 2 - 1).cuda()
    z = torch.randn(8, 128, generator=g).cuda()
    sd = copy.deepcopy(netD.state_dict())
    sdG = copy.deepcopy(netG.state_dict())
    res = {}
    for mode in (True, False):
        netD.load_state_dict(sd)
        netG.load_state_dict(sdG)
        netD.pair_forward = mode
        log = netD.train_step(real_batch=(x, None), netG=netG, optD=optD, log_data=Log(), device='cuda', noise=z)
        res[mode] = (log.m['errD'].item(), {k: v.clone() for k, v in netD.export_grads().items()},
                     {k: v.clone() for k, v in netD.state_dict().items() if 'sn_' in k})
    netD.pair_forward = True
    assert abs(res[True][0] - res[False][0]) < 1e-5
    for k in res[True][1]:
        relclose(res[True][1][k], res[False][1][k], 1e-4, f"grad {k}")
    for k in res[True][2]:
        relclose(res[True][2][k], res[False][2][k], 1e-6, k)


def test_topk_and_gold_losses_vs_oracle():
    from diagan.ops import eltwise as E
    g = torch.Generator().manual_seed(5)
    r = torch.randn(64, 1, generator=g) * 2
    f = torch.randn(64, 1, generator=g) * 2
    for loss in ('ns', 'hinge', 'gan', 'wasserstein'):
        for gold in ((False, True) if loss in ('ns', 'hinge') else (False,)):
            rr, ff = r.clone().requires_grad_(True), f.clone().requires_grad_(True)
            ref = O.dis_loss(loss, rr, ff, gold=gold)
            ref.backward()
            out3, dr, df = E.loss_dis(r.cuda(), f.cuda(), loss, gold=gold)
            assert abs(out3[0].item() - ref.item()) < 1e-5, (loss, gold)
            assert (dr.cpu() - rr.grad.view(-1)).abs().max() < 1e-6
            assert (df.cpu() - ff.grad.view(-1)).abs().max() < 1e-6
            assert abs(out3[1].item() - torch.sigmoid(r).mean().item()) < 1e-6
        for rate in (1.0, 0.77, 0.5):
            ff = f.clone().requires_grad_(True)
            k = int(rate * 64)
            ref = O.gen_loss(loss, torch.topk(ff, k, dim=0)[0])
            ref.backward()
            out1, df = E.loss_gen(f.cuda(), loss, k=k)
            assert abs(out1[0].item() - ref.item()) < 1e-5, (loss, rate)
            assert (df.cpu() - ff.grad.view(-1)).abs().max() < 1e-6


def test_checkpoint_roundtrip(tmp_path):
    (_, _, _, _), (netG, netD, optG, optD) = build("cifar10", "ns")
    x = torch.rand(4, 3, 32, 32).cuda() * 2 - 1
    netD.train_step(real_batch=(x, None), netG=netG, optD=optD, log_data=Log(), device='cuda')
    netD.save_checkpoint(str(tmp_path / "netD"), 7, optD)
    from diagan.models.predefined_models import get_gan_model
    _, netD2, _, optD2 = get_gan_model('cifar10', model='sngan', loss_type='ns')
    netD2.to('cuda')
    step = netD2.restore_checkpoint(str(tmp_path / "netD" / "netD_7_steps.pth"), optD2)
    assert step == 7
    netD.eval(); netD2.eval()
    assert torch.equal(netD(x), netD2(x))
    ck = torch.load(str(tmp_path / "netD" / "netD_7_steps.pth"), weights_only=False)
    assert ck['model_state_dict']['block1.c1.weight'].shape == (128, 3, 3, 3)
    assert ck['model_state_dict']['l5.weight'].shape == (1, 128)


def test_gold_and_topk_train_steps_vs_oracle():
    """GOLD re-weighted D loss (train_mimicry_phase2.py --gold) and top-k G training (--topk) through the
    real train steps, against the oracle with the same switches."""
    from diagan.models.predefined_models import get_gan_model
    oG, oD, ooptG, ooptD = O.make_pair('cifar10', 'ns', seed=2)
    oD.use_gold = True
    torch.manual_seed(2)
    netG, netD, optG, optD = get_gan_model('cifar10', model='sngan', loss_type='ns', gold=True, topk=True)
    netG.load_state_dict(oG.state_dict()), netD.load_state_dict(oD.state_dict())
    netG.to('cuda'), netD.to('cuda')
    assert netD.use_gold and netG.use_topk
    netG.decay_topk_rate(782 * 30, epoch_steps=782)          # rate 0.99^30 = 0.7397 -> k = 5 of 8
    g = torch.Generator().manual_seed(8)
    x = torch.rand(8, 3, 32, 32, generator=g) * 2 - 1
    zd, zg = torch.randn(8, 128, generator=g), torch.randn(8, 128, generator=g)
    errD, _, _ = oD.train_step((x, None), oG, ooptD, noise=zd)
    log = netD.train_step(real_batch=(x.cuda(), None), netG=netG, optD=optD, log_data=Log(), device='cuda',
                          noise=zd.cuda())
    assert abs(log.m['errD'].item() - errD) < 1e-3
    gr = netD.export_grads()
    for k, p in oD.named_parameters():
        l2close(gr[k], p.grad, 1e-2, f"gold D grad {k}")
    errG = oG.train_step((x, None), oD, ooptG, noise=zg, topk_rate=netG.topk_rate)
    log = netG.train_step(real_batch=(x.cuda(), None), netD=netD, optG=optG, log_data=Log(), device='cuda',
                          noise=zg.cuda())
    assert abs(log.m['errG'].item() - errG) < 5e-3 * max(1.0, abs(errG))


def test_resume_from_torch_adam_state(tmp_path):
    """Checkpoint wire format: a FusedAdam resumes from a torch.optim.Adam state_dict taken
    over the reference-shaped parameters (what a mimicry checkpoint's 'optimizer_state_dict' holds), and writes one
    a torch.optim.Adam accepts.  The oracle (plain torch.optim.Adam, lr 2e-4, betas (0, 0.9)) warms up alone for
    two D and one G update; both sides then continue from its model + optimizer state and must stay together."""
    (oG, oD, ooptG, ooptD), (netG, netD, optG, optD) = build('cifar10', 'hinge')
    assert [n for n, _ in netD.named_parameters()] == [n for n, _ in oD.named_parameters()]
    assert [n for n, _ in netG.named_parameters()] == [n for n, _ in oG.named_parameters()]
    g = torch.Generator().manual_seed(9)
    B = 8
    batch = lambda: (torch.rand(B, 3, 32, 32, generator=g) * 2 - 1, torch.randn(B, 128, generator=g))
    for _ in range(2):
        x, z = batch()
        oD.train_step((x, None), oG, ooptD, noise=z)
    x, z = batch()
    oG.train_step((x, None), oD, ooptG, noise=z)
    # genuine-format checkpoint files written by torch, restored through the engine's own restore_checkpoint
    for net, opt, name in ((oD, ooptD, 'netD'), (oG, ooptG, 'netG')):
        torch.save({'model_state_dict': net.state_dict(), 'optimizer_state_dict': opt.state_dict(), 'global_step': 3},
                   tmp_path / f'{name}_3_steps.pth')
    assert netD.restore_checkpoint(ckpt_file=str(tmp_path / 'netD_3_steps.pth'), optimizer=optD) == 3
    assert netG.restore_checkpoint(ckpt_file=str(tmp_path / 'netG_3_steps.pth'), optimizer=optG) == 3
    assert optD._step == 2 and optG._step == 1
    # moments round-trip exactly through the packed device layout
    back = optD.state_dict()
    ref = ooptD.state_dict()
    for i in ref['state']:
        assert torch.equal(back['state'][i]['exp_avg_sq'], ref['state'][i]['exp_avg_sq'].cpu()), i
    torch.optim.Adam(list(copy.deepcopy(oD).parameters()), lr=2e-4, betas=(0.0, 0.9)).load_state_dict(back)
    # continue on both sides from the restored state
    x, z = batch()
    errD, _, _ = oD.train_step((x, None), oG, ooptD, noise=z)
    log = netD.train_step(real_batch=(x.cuda(), None), netG=netG, optD=optD, log_data=Log(), device='cuda', noise=z.cuda())
    assert abs(log.m['errD'].item() - errD) < 1e-3
    x, z = batch()
    errG = oG.train_step((x, None), oD, ooptG, noise=z)
    log = netG.train_step(real_batch=(x.cuda(), None), netD=netD, optG=optG, log_data=Log(), device='cuda', noise=z.cuda())
    assert abs(log.m['errG'].item() - errG) < 5e-3 * max(1.0, abs(errG))
    sD, sG = netD.state_dict(), netG.state_dict()
    for k, v in oD.state_dict().items():
        assert (sD[k].cpu() - v).abs().max() < 2e-3, k
    for k, v in oG.state_dict().items():
        if v.dtype.is_floating_point:
            assert (sG[k].cpu() - v).abs().max() < 2e-3, k


@pytest.mark.parametrize("winograd,tol", [(False, 1e-5), (True, 1e-5)])
@pytest.mark.parametrize("dataset", ["cifar10", "celeba"])
def test_stacked_generator_forward_equals_successive_forwards(dataset, winograd, tol):
    """prefetch_fakes: the n_dis generator forwards of a global step as ONE stacked forward with per-batch BatchNorm
    statistics -- same images, same running statistics (momentum chain in order), same RNG state afterwards.
    The stacked (3x larger) launches and the single-batch ones do not all take the same kernel (launch-size policy: Winograd
    or not, and since round 3 the 64x64 tile with one or two K-groups, which sums K in two halves), which moves the images
    by rounding: 1e-5 (2e-6 while both sides took the same implicit-GEMM kernel)."""
    from diagan.models import base as MB
    from diagan.ops import conv as C
    C.set_winograd(winograd)
    try:
        _stacked_forward_check(dataset, tol)
    finally:
        C.set_winograd(None)


def test_stacked_forward_with_a_batch_size_not_divisible_by_four():
    """--batch_size 50, n_dis 5 (+ the generator's own batch): six groups of 50 images.  At 8x8 a group is 3200 GEMM rows,
    not a multiple of the Winograd kernel's 256-row tile although the launch is large enough to be given that kernel:
    the launch must fall back to a tile that divides the group (diagan_conv_gemm_pick_cfg_grouped), not raise."""
    _stacked_forward_check('cifar10', 2e-5, B=50, n=6, short=True)


def _stacked_forward_check(dataset, tol, B=8, n=3, short=False):
    from diagan.models import base as MB
    (_, _, _, _), (netG, _, _, _) = build(dataset, 'ns')
    ref = copy.deepcopy(netG)
    torch.manual_seed(21)
    netG.prefetch_fakes(n, B, device='cuda')
    got = [netG.generate_images_nhwc(B)[0] for _ in range(n)]
    after = torch.randn(4, device='cuda')
    assert netG._fake_pool == []                                   # pool exhausted: the next call is an ordinary forward
    torch.manual_seed(21)
    want = [ref.generate_images_nhwc(B)[0] for _ in range(n)]      # ref never prefetched: n ordinary forwards
    assert torch.equal(after, torch.randn(4, device='cuda'))       # the generator consumed the same random numbers
    for a, b in zip(got, want):
        assert a.shape == b.shape
        assert (a - b).abs().max().item() < tol
    sa, sb = netG.state_dict(), ref.state_dict()
    for k in sb:
        if 'running' in k:
            assert (sa[k] - sb[k]).abs().max().item() < tol / 2, k
    if short:
        return
    # a stack that would not fit the 2 GiB tensor limit is cut into several stacked forwards: same images again
    netG.load_state_dict(ref.state_dict())                      # same running statistics as before the chunked run ...
    ref2 = copy.deepcopy(ref)
    netG.max_stacked_images = 2 * B
    torch.manual_seed(22)
    netG.prefetch_fakes(n, B, device='cuda')
    assert len(netG._fake_pool) == n
    chunked = [netG.generate_images_nhwc(B)[0] for _ in range(n)]
    torch.manual_seed(22)
    for a in chunked:
        assert (a - ref2.generate_images_nhwc(B)[0]).abs().max().item() < tol
    # ragged request / changed parameters / eval mode invalidate the pool instead of serving stale images
    netG.prefetch_fakes(n, B, device='cuda')
    assert netG.generate_images_nhwc(B - 1)[0].shape[0] == B - 1 and not netG._fake_pool
    netG.prefetch_fakes(n, B, device='cuda')
    netG.param_version += 1
    netG.generate_images_nhwc(B)
    assert not netG._fake_pool


def test_d_updates_with_prefetched_fakes_match(monkeypatch):
    """Two D updates fed from the pool equal two D updates that run their own generator forward."""
    (_, _, _, _), (netG, netD, optG, optD) = build('cifar10', 'hinge')
    (_, _, _, _), (netG2, netD2, _, optD2) = build('cifar10', 'hinge')          # same seed: identical twins
    g = torch.Generator().manual_seed(4)
    xs = [(torch.rand(8, 3, 32, 32, generator=g) * 2 - 1).cuda() for _ in range(2)]
    torch.manual_seed(33)
    netG.prefetch_fakes(2, 8, device='cuda')
    logs = [netD.train_step(real_batch=(x, None), netG=netG, optD=optD, log_data=Log(), device='cuda') for x in xs]
    torch.manual_seed(33)
    logs2 = [netD2.train_step(real_batch=(x, None), netG=netG2, optD=optD2, log_data=Log(), device='cuda') for x in xs]
    for a, b in zip(logs, logs2):
        assert abs(a.m['errD'].item() - b.m['errD'].item()) < 1e-5
    assert (netD.flat_params - netD2.flat_params).abs().max().item() < 1e-6


def test_generator_update_stacked_onto_the_fake_batches(monkeypatch):
    """DIAGAN_STACK_G_STEP=1: the generator update's own forward is the last batch of the stacked forward that makes the
    n_dis fake batches (same weights; its noise is drawn right after theirs, as successive updates would) and keeps the
    context for its backward.  Same fakes, same generator gradients and same device-generator state as the separate path."""
    from diagan.models import base
    B, n = 16, 3

    def run(stack):
        monkeypatch.setattr(base, "STACK_G_STEP", stack)
        (_, _, _, _), (netG, netD, optG, optD) = build("cifar10", "ns", seed=7)
        netG.train(), netD.train()
        torch.cuda.manual_seed(123)
        netG.prefetch_fakes(n, B, device='cuda', g_step=True)
        fakes = [netG.generate_images_nhwc(B)[0].clone() for _ in range(n)]
        x = torch.zeros(B, 3, 32, 32)
        netG.train_step(real_batch=(x.cuda(), None), netD=netD, optG=optG, log_data=Log(), device='cuda')
        return fakes, netG.export_grads(), netG.state_dict(), torch.cuda.get_rng_state()

    fa, ga, sa, ra = run(False)
    fb, gb, sb, rb = run(True)
    assert torch.equal(ra, rb)
    for a, b in zip(fa, fb):
        relclose(a, b, 1e-5, "fake batch")
    # (the update's fake batch differs by rounding between the two paths -- other launch sizes, other kernels -- which flips
    #  a few of D's ReLU masks: gradients agree in L2 like any two fp32 implementations, see l2close)
    wscale = max(v.norm().item() for v in ga.values())
    for k in ga:
        if is_dead_bias(k):
            assert gb[k].abs().max().item() < 1e-4 * wscale, k
        else:
            l2close(gb[k], ga[k], 1e-2, f"G grad {k}")
    num = sum(float((sb[k].double() - v.double()).pow(2).sum()) for k, v in sa.items() if v.dtype.is_floating_point)
    den = sum(float(v.double().pow(2).sum()) for v in sa.values() if v.dtype.is_floating_point)
    assert (num / den) ** 0.5 < 1e-3
    for k in ("block2.b1.running_mean", "b5.running_var"):            # BatchNorm saw n + 1 batches, in order, on both paths
        relclose(sb[k], sa[k], 1e-5, k)


@pytest.mark.parametrize("dataset,nup", [("cifar10", 3), ("celeba", 4)])
def test_stacked_forward_folds_the_upsampling_into_c1(dataset, nup, monkeypatch):
    """VERDICT r3 item 1 (mimicry GBlock._upsample_conv as selected at predefined_models.py:19,57): in the stacked
    generator forward of a global step (6 x 64 images) every GBlock's BN -> ReLU -> bilinear x2 -> c1 is ONE launch of the
    F(4x4) kernel on the low-resolution input (tile_cfg 15) -- `upsample2x` is not called for c1 at all -- and the images,
    the BatchNorm running statistics and the generator update's gradients equal the two-launch form (DIAGAN_UPIN=0)."""
    from diagan.ops import conv as C
    from diagan.ops import eltwise as E
    B, n = 64, 5

    def run(fused):
        monkeypatch.setenv("DIAGAN_UPIN", "1" if fused else "0")
        (_, _, _, _), (netG, netD, optG, optD) = build(dataset, "ns", seed=9)
        netG.train(), netD.train()
        calls, orig = [], E.upsample2x
        monkeypatch.setattr(E, "upsample2x", lambda x, pro=None: (calls.append(tuple(x.shape)), orig(x, pro=pro))[1])
        C.TIMER = C.KernelTimer()
        try:
            torch.cuda.manual_seed(77)
            netG.prefetch_fakes(n, B, device='cuda', g_step=True)
            torch.cuda.synchronize()
            names = [r[0] for r in C.TIMER.records]
        finally:
            C.TIMER = None
        fakes = [netG.generate_images_nhwc(B)[0].clone() for _ in range(n)]
        res = fakes[0].shape[1]
        netG.train_step(real_batch=(torch.zeros(B, 3, res, res).cuda(), None), netD=netD, optG=optG, log_data=Log(),
                        device='cuda')
        monkeypatch.setattr(E, "upsample2x", orig)
        return names, calls, fakes, netG.export_grads(), netG.state_dict()

    names_f, calls_f, fakes_f, g_f, sd_f = run(True)
    names_t, calls_t, fakes_t, g_t, sd_t = run(False)
    assert names_f.count("conv_wino4_kernel<2,3,false>") == nup and "conv_wino4_kernel<2,3,false>" not in names_t
    # fused: the stacked forward up-samples nothing; only the update's backward makes its own batch-64 copies for c1's
    # weight gradient.  Two-launch form: one stacked up-sampling per block in the forward.
    assert all(s[0] == B for s in calls_f) and len(calls_f) == nup, calls_f
    assert sum(1 for s in calls_t if s[0] == (n + 1) * B) == nup, calls_t
    # (both forms run the F(4x4) transforms in fp32 -- ~1e-5 of the output scale per layer, tests/test_wino4_gpu.py -- but round
    #  differently: the interpolation before or inside the input transform)
    for a, b in zip(fakes_f, fakes_t):
        relclose(a, b, 1e-4, "fake batch")
    for k in sd_f:
        if 'running' in k:
            relclose(sd_f[k], sd_t[k], 1e-4, k)
    wscale = max(v.norm().item() for v in g_t.values())
    for k in g_t:
        if is_dead_bias(k):
            assert g_f[k].abs().max().item() < 1e-4 * wscale, k
        else:
            l2close(g_f[k], g_t[k], 2e-2, f"G grad {k}")


def test_deep_copy_after_forwards_uses_its_own_weights():
    """a FlatNet deep-copied AFTER its Winograd call sites were learned kept sites whose weight getters returned
    the ORIGINAL's weights; once the two diverge the copy convolved with the wrong ones.  The copy is perturbed, run, and
    compared with a freshly built network that loaded the copy's state dict (never shared anything with the original)."""
    from diagan.models.predefined_models import get_gan_model
    (_, _, _, _), (netG, _, _, _) = build('cifar10', 'ns')
    torch.manual_seed(3)
    for _ in range(3):                                   # sites are learned on the first pass, batches used from the second
        netG.generate_images_nhwc(8)
    twin = copy.deepcopy(netG)
    with torch.no_grad():
        twin.flat_params.mul_(1.25)
        twin.param_version += 1
    fresh = get_gan_model('cifar10', model='sngan', loss_type='ns')[0]
    fresh.load_state_dict(twin.state_dict())
    fresh.to('cuda')
    outs = []
    for net in (twin, fresh, netG):
        net.eval()
        torch.manual_seed(4)
        for _ in range(2):
            y = net.generate_images_nhwc(8)[0]
        outs.append(y)
    relclose(outs[0], outs[1], 1e-5, "copy vs freshly loaded")
    assert (outs[0] - outs[2]).abs().max().item() > 1e-3          # and it really differs from the original's images


def test_five_step_trajectory_drifts_no_faster_than_the_exact_fp32_build_and_the_cpu_oracle():
    """SURVEY section 4 item 3 (the loop of trainer.py:238-299) for the headline network: five global steps (5 D + 1 G updates
    each, Adam, batch 64) from the same weights with the same injected batches and noise -- the engine as shipped (Winograd
    F(4x4) / F(2x2)), the engine with the implicit GEMM everywhere, the CPU oracle in fp32 -- each measured against the CPU
    oracle in float64 (tools/sngan_trajectory.py; 20 steps: profiles/r05_trajectory.md).  With beta1 = 0 Adam's first steps are
    lr * g / |g|: EVERY fp32 implementation is 4-7 % of the float64 run's own movement away after one step (sign flips of
    near-zero gradient entries) and the losses decorrelate within ~10 steps, so what is asserted is relative: the default build
    is at no step more than 3x plain PyTorch fp32 away, and over the five steps it drifts no faster than 2x the exact-fp32 build
    OR no faster than 1.25x plain PyTorch fp32 (the arithmetic class the reference itself runs in).  The second alternative is
    there because the exact-fp32 build's own distance moves between 0.009 and 0.037 after step 1 on the same inputs from build to
    build -- sign flips it happens to share with float64 -- while the Winograd build sits at PyTorch fp32's level, 0.05-0.08
    (profiles/r05_trajectory.md has the table); the first step's losses agree to the contract's 1e-3."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import numpy as np
    from sngan_trajectory import trajectories
    # the CPU legs (oracle/nets.py in float64 and fp32: 280 s of float64 autograd) come from tests/golden/sngan_trajectory.npz, made
    # by tools/gen_goldens_trajectory.py from the same function: the float64 trajectory as count-sketches (distances to ~1 %)
    golden = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sngan_trajectory.npz")))
    r = trajectories("cifar10", steps=5, golden=golden)
    d, e, c = (r[k]["dist"] for k in ("hip default", "hip exact-fp32", "cpu fp32"))
    print("trajectory dist:", [f"{v:.3e}" for v in d], [f"{v:.3e}" for v in e], [f"{v:.3e}" for v in c])
    for k in ("hip default", "hip exact-fp32", "cpu fp32"):
        assert r[k]["errD"][0] < 1e-3 and r[k]["errG"][0] < 1e-3, (k, r[k]["errD"][0], r[k]["errG"][0])
    for s in range(5):
        assert d[s] <= 3.0 * c[s] + 0.01, (s, d[s], c[s])
    assert sum(d) <= max(2.0 * sum(e), 1.25 * sum(c)) + 0.01, (sum(d) / 5, sum(e) / 5, sum(c) / 5)
    assert d[-1] < 0.25                          # and it is still the same run, not a diverged one


@pytest.mark.parametrize("dataset", ["cifar10", "celeba"])
def test_batched_weight_gradient_launches_equal_the_per_layer_ones(dataset):
    """Round 5: the Winograd weight gradients of a backward pass run as ONE launch per prologue mode at the end of the pass
    (diagan_conv_wgrad_batched; the layers share the chip's workgroups, so their split counts -- the order of the fp32 sums
    -- differ from the per-layer launches').  Same weights, images and noise with the batching on and off: losses equal,
    gradients equal to fp32 summation-order rounding, and the batched path really ran."""
    from diagan.ops import conv as C
    res = 32 if dataset == "cifar10" else 64
    g = torch.Generator().manual_seed(5)
    x = torch.rand(64, 3, res, res, generator=g) * 2 - 1
    zd, zg = torch.randn(64, 128, generator=g), torch.randn(64, 128, generator=g)
    out = {}
    try:
        for on in (True, False):
            C.WGRAD_BATCH = on
            (_, _, _, _), (netG, netD, optG, optD) = build(dataset, "ns")
            log = netD.train_step(real_batch=(x.cuda(), None), netG=netG, optD=optD, log_data=Log(), device='cuda', noise=zd.cuda())
            gd, ed = netD.export_grads(), log.m['errD'].item()
            log = netG.train_step(real_batch=(x.cuda(), None), netD=netD, optG=optG, log_data=Log(), device='cuda', noise=zg.cuda())
            out[on] = (gd, ed, netG.export_grads(), log.m['errG'].item(),
                       netD.wgrad_batch.batched_launches + netG.wgrad_batch.batched_launches)
    finally:
        C.WGRAD_BATCH = True
    assert out[True][4] >= 2 and out[False][4] == 0
    assert out[True][1] == out[False][1]                       # the D forward does not depend on the switch
    for k, v in out[False][0].items():
        relclose(out[True][0][k], v, 2e-5, f"D grad {k}")
    # (the G update runs through a discriminator whose Adam step saw gradients that differ in the last bits: ReLU masks of
    #  near-zero pre-activations may flip, hence the L2 form -- see l2close)
    for k, v in out[False][2].items():
        if not is_dead_bias(k):
            l2close(out[True][2][k], v, 2e-3, f"G grad {k}", floor=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("dataset", ["cifar10", "celeba"])
def test_box_sums_in_the_weight_gradients_loader_change_no_bit(dataset):
    """Round 5: the weight gradient of a convolution + average pool layer gathers from the 2x2 box sums of its input; the sums are
    now taken by the kernel's loader (prologue modes 5 / 6 of diagan_conv_wgrad) instead of a diagan_boxsum2 pass in front.  Same
    operands in the same order: a D update with the switch on and off gives the same gradients bit for bit."""
    from diagan.models import layers as L
    res = 32 if dataset == "cifar10" else 64
    g = torch.Generator().manual_seed(6)
    x = torch.rand(64, 3, res, res, generator=g) * 2 - 1
    zd = torch.randn(64, 128, generator=g)
    out = {}
    try:
        for on in (True, False):
            L.WGRAD_BOX = on
            (_, _, _, _), (netG, netD, optG, optD) = build(dataset, "ns")
            log = netD.train_step(real_batch=(x.cuda(), None), netG=netG, optD=optD, log_data=Log(), device='cuda', noise=zd.cuda())
            out[on] = (netD.export_grads(), log.m['errD'].item())
    finally:
        L.WGRAD_BOX = True
    assert out[True][1] == out[False][1]
    for k, v in out[False][0].items():
        assert torch.equal(out[True][0][k], v), k
