#!/usr/bin/env python
"""Headline benchmark: images/sec of the SNGAN G+D training step (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started bare with --gpus N > 1 (no WORLD_SIZE in the environment) the script launches the second form itself as a
CHILD process (one rank per GPU, like the reference's `torch.distributed.launch --nproc_per_node=4`, README.md:149 /
stylegan2/train_ffhq.py:503-506), relays rank 0's JSON line and exits with the child's code.  A world size that
differs from --gpus is an error (non-zero exit, no JSON line).

One "step" = one global step of LogTrainer.train (diagan-pkg/diagan/trainer/trainer.py:250-299):
n_dis = 5 discriminator updates + 1 generator update (+ the LR scheduler) at batch 64 PER GPU
(weak scaling, as stylegan2/train_ffhq.py:393 "batch sizes for each gpus"), on synthetic images that
are resident in HBM before the timed region.  Default workload: BASELINE.json configs[1]
(CIFAR-10 SNGAN phase-1 'ns' loss, 32x32).  Prints ONE JSON line on rank 0.

Extra objects on the line:
  roofline      dominant kernel (most GPU time among the GEMM kernels, found in the last warm-up step):
                algorithmic FLOP / launch time, measured with HIP events around every launch of THAT
                kernel inside the timed region; the table of the other GEMM kernels comes from two
                un-timed steps after it (bracketing every launch costs ~2 ms of host time per step)
  cpu_baseline  the CPU restatement (oracle/nets.py, kind "port") timed on this box's host cores,
                rank 0 at N = 1 only, on a bounded sample of the same workload
"""
import argparse
import contextlib
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))

import torch

MFMA_F32_PEAK = 157.3e12      # MI355X_MICROARCH.md: dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)
HBM_PEAK = 8.0e12
BF16_X3_PEAK = 2.5e15 / 6     # split-operand kernels (conv_gemm_x3.hip, conv_gemm_x3b.hip, the X3 build of conv_wino4.hip): dense bf16
                              # MFMA peak / six piece products per fp32 product = 416.7 TFLOP/s of fp32-equivalent products


def kernel_peak(name):
    """FLOP/s peak of the matrix pipe the kernel `name` runs its products on"""
    if name.startswith(("conv_gemm_x3", "conv_wgrad_x3")) or (name.startswith("conv_wino4_kernel") and name.rstrip().endswith(",true>")):
        return BF16_X3_PEAK
    return MFMA_F32_PEAK

WORKLOADS = {
    # name: (dataset key, resolution, description)
    'sngan32': ('cifar10', 32, "CIFAR-10 SNGAN phase-1 ns-loss bs=64, synthetic 32x32"),
    'sngan64': ('celeba', 64, "CelebA-64 SNGAN phase-1 ns-loss bs=64, synthetic 64x64"),
    'dcgan': ('color_mnist', 32, "Colored-MNIST mnist_dcgan phase-1 bs=64, synthetic 32x32 (BASELINE configs[0] shape)"),
    # BASELINE configs[4] shape; one "step" = one iteration of stylegan2/train_ffhq.py (D step, G step, EMA, R1 every
    # 16th and path-length every 4th iteration; --phase 2 adds D_drs); --batch_size defaults to 32 for this workload
    'stylegan2': ('ffhq', 256, "FFHQ-256 StyleGAN2 (stylegan2/train_ffhq.py iteration) bs=32, synthetic 256x256"),
}


def make_stylegan2_step(size, batch, phase, device):
    """the iteration of diagan/trainer/stylegan2.py on synthetic (image, index) pairs; returns (step, nets, optimizers)"""
    import types
    from diagan.models.stylegan2 import StyleGANDiscriminator, StyleGANGenerator
    from diagan.trainer import stylegan2 as TR
    from diagan.utils.settings import set_seed
    with contextlib.redirect_stdout(sys.stderr):        # (the reference's seed banner: stdout carries the JSON line only)
        set_seed(1)
    G, D = StyleGANGenerator(size=size).to(device), StyleGANDiscriminator(size=size).to(device)
    g_ema = StyleGANGenerator(size=size).to(device).eval()
    TR.accumulate(g_ema, G, 0)
    g_optim, d_optim = TR.make_optimizers(G, D)
    a = types.SimpleNamespace(iter=10 ** 9, start_iter=0, batch=batch, latent=512, mixing=0.9, r1=10.0, d_reg_every=16,
                              g_reg_every=4, path_regularize=2.0, path_batch_shrink=2, logit_save_steps=10 ** 9,
                              save_logit_after=10 ** 9, stop_save_logit_after=0, n_sample=16, augment=False)
    images = (torch.rand(4 * batch, 3, size, size, generator=torch.Generator().manual_seed(1234)) * 2 - 1).to(device)
    ds = torch.utils.data.TensorDataset(images, torch.arange(len(images), device=device))     # resident in HBM
    mk = lambda: torch.utils.data.DataLoader(ds, batch_size=batch, shuffle=True, drop_last=True)
    extra = {}
    if phase == 2:
        D2 = StyleGANDiscriminator(size=size).to(device)
        extra = dict(drs_loader=mk(), drs_discriminator=D2, drs_d_optim=TR.make_optimizers(G, D2)[1])
    tr = TR.StyleGAN2Trainer(a, mk(), G, D, g_optim, d_optim, g_ema, device, "/tmp/diagan_bench_sg2", **extra)
    zero = torch.tensor(0.0, device=device)
    tr.r1_loss, tr.path_loss, tr.path_lengths = zero, zero, zero
    state = dict(i=0)

    def step():
        tr.train_step(state['i'])
        state['i'] += 1
    return step


class NullLog:
    def add_metric(self, *a, **k):
        pass


def build_models(dataset, loss_type, phase, device):
    from diagan.models.predefined_models import get_gan_model
    from diagan.utils.settings import set_seed
    with contextlib.redirect_stdout(sys.stderr):        # (the reference's seed banner: stdout carries the JSON line only)
        set_seed(1)
    if phase == 2:
        netG, netD, netD_drs, optG, optD, optD_drs = get_gan_model(dataset, model='sngan', loss_type=loss_type, drs=True)
    else:
        netG, netD, optG, optD = get_gan_model(dataset, model='sngan', loss_type=loss_type)
        netD_drs = optD_drs = None
    for n in (netG, netD, netD_drs):
        if n is not None:
            n.to(device)
    return netG, netD, netD_drs, optG, optD, optD_drs


def make_global_step(netG, netD, netD_drs, optG, optD, optD_drs, batches, n_dis, num_steps, device):
    """The body of LogTrainer.train's while-loop without logging / checkpoint I/O."""
    from diagan.trainer.scheduler import DRS_LRScheduler
    sched = DRS_LRScheduler('linear', [o for o in (optD, optD_drs, optG) if o is not None], num_steps)
    log = NullLog()
    state = dict(step=0, cursor=0)

    def fetch():
        b = batches[state['cursor'] % len(batches)]
        state['cursor'] += 1
        return (b, None)

    def device_part():
        netG.prefetch_fakes(n_dis * (2 if netD_drs is not None else 1), batches[0].shape[0], device=device, g_step=True)   # as LogTrainer._updates
        from diagan.trainer import distributed as dist
        overlap = netD_drs is not None and dist.get_world_size() > 1      # as LogTrainer._updates
        for i in range(n_dis):
            real = fetch()
            netD.train_step(real_batch=real, netG=netG, optD=optD, log_data=log, global_step=state['step'],
                            device=device, **(dict(defer_step=True) if overlap else {}))
            if netD_drs is not None:
                netD_drs.train_step(real_batch=fetch(), netG=netG, optD=optD_drs, log_data=log,
                                    global_step=state['step'], device=device)
            if overlap:
                optD.step()
            if i == n_dis - 1:
                netG.train_step(real_batch=real, netD=netD, optG=optG, log_data=log, global_step=state['step'],
                                device=device)

    def host_part():
        state['step'] += 1
        sched.step(log, state['step'])

    def step():
        device_part()
        host_part()

    step.device_part, step.host_part = device_part, host_part      # --graph replays the first, runs the second
    return step


def cpu_baseline(dataset, res, loss_type, batch, n_dis, budget_s=25.0):
    """The CPU restatement of the same global step on the host cores this process may use.
    Bounded sample: D updates and G updates are timed separately (at least one of each, more while
    the budget lasts) and combined as one global step = n_dis * t_D + t_G."""
    from oracle import nets as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    try:                                     # cgroup CPU quota of the box (e.g. "1600000 100000" = 16 cores)
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            avail = max(1, min(avail, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    cores = max(1, min(avail, 64))           # beyond the quota / 64 threads torch CPU convs only oversubscribe
    torch.set_num_threads(cores)
    oG, oD, ooptG, ooptD = O.make_pair(dataset, loss_type, seed=1)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(batch, 3, res, res, generator=g) * 2 - 1
    t_start = time.time()
    oD.train_step((x, None), oG, ooptD)      # warm-up: thread pool, allocator (not timed)
    warm = time.time() - t_start
    td, tg = [], []
    while True:
        t0 = time.time(); oD.train_step((x, None), oG, ooptD); td.append(time.time() - t0)
        t0 = time.time(); oG.train_step((x, None), oD, ooptG); tg.append(time.time() - t0)
        if time.time() - t_start > budget_s or len(td) >= 5:
            break
    t_d, t_g = min(td), min(tg)
    t_step = n_dis * t_d + t_g
    return {"value": round(batch / t_step, 3), "unit": "images/s", "cores": cores, "cpu_model": cpu_model_name(),
            "kind": "port",
            "sample": f"oracle/nets.py (torch CPU ops, {cores} threads of {avail} available): best of {len(td)} "
                      f"D update(s) {t_d:.2f} s and G update(s) {t_g:.2f} s at bs={batch}; one global step = "
                      f"{n_dis}*t_D + t_G = {t_step:.2f} s (warm-up D update {warm:.1f} s not counted)"}


WINO_MAC_RATIO = 16.0 / 36.0    # Winograd F(2x2,3x3): multiply-accumulates executed per direct-convolution multiply-accumulate


WINO_POOL_MAC_RATIO = 9.0 / 36.0    # ... followed by a 2x2 average pool (conv_wino_pool.hip): 9 of the 16 products suffice


def executed_flop(name, flop):
    """FLOP the matrix pipe actually executes for `flop` algorithmic (direct-convolution, 2*M*Co*R*S*Ci) FLOP"""
    if name.startswith("conv_wino_pool_kernel"):
        return flop * WINO_POOL_MAC_RATIO
    if name.startswith("conv_wino4_kernel"):     # Winograd F(4x4,3x3): 36 products per 4x4 output tile instead of 144
        m = re.match(r"conv_wino4_kernel<\s*\d+\s*,\s*(\d+)", name)     # <prologue, MODE[, bf16-split build]>
        return flop * (25.0 / 144.0 if m and m.group(1) in ("1", "2") else 0.25)     # ... 25 with the 2x2 average pool folded in
    if name.endswith("[pooled gradient]"):     # weight gradient through the average pool as a strided convolution over box sums
        return flop * 0.25
    return flop * WINO_MAC_RATIO if name.startswith(("conv_wino_kernel", "conv_wgrad_wino_kernel", "conv_wgrad_wino_batched_kernel")) else flop


def conv_block_excluded(name, shape):
    """'SNGAN-64 conv blocks' = the 3x3 / 1x1 convolutions of the residual blocks of a4 + a5, forward and
    backward.  Out: the generator's latent linear l1 (N = 16384), its last conv c6 (forward: the 4-output-channel
    kernel with the BatchNorm prologue; weight gradient: the 4-channel weight-gradient kernel).  c6's data-gradient
    (1.2 GFLOP of 4726) has the same (M, N, K) as block1.c1's forward and stays in numerator and denominator alike."""
    if shape is not None and len(shape) >= 4:
        M, N, K, tag = shape[:4]
        if N == 16384:
            return True
        if name == "conv3x3_co4_kernel" and str(tag).startswith("pro2"):
            return True
    return name == "conv3x3_co4_wgrad_kernel"


def sngan64_leg(args, device, steps=10, warmup=3):
    """north_star's kernel target, on the driver's own line: >= 60 % of the fp32 MFMA roofline on the SNGAN 64x64 conv
    blocks at bs = 64 (reference nets: diagan-pkg/diagan/models/predefined_models.py:57-59,76-78).  Un-scored leg run
    after the timed region: `steps` global steps of the CelebA-64 configuration for images/s, then two more with every
    GEMM launch bracketed by HIP events: numerator = algorithmic FLOP of the residual-block convolutions (forward, data
    and weight gradients), denominator = the summed launch time of exactly those launches."""
    from diagan.ops import conv as C
    dataset, res, desc = WORKLOADS['sngan64']
    nets = build_models(dataset, args.loss_type, 1, device)
    gen = torch.Generator().manual_seed(1234)
    batches = [(torch.rand(args.batch_size, 3, res, res, generator=gen) * 2 - 1).to(device) for _ in range(2 * args.n_dis)]
    step = make_global_step(*nets, batches, args.n_dis, num_steps=75000, device=device)
    for _ in range(warmup):
        step()
    from diagan.utils.settings import quiesce_gc
    quiesce_gc()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    C.TIMER = full = C.KernelTimer()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    C.TIMER = None
    flop = secs = xflop = 0.0
    flop_all = secs_all = 0.0
    per_kernel = {}
    for name, f, s, e, shape in full.records:
        dt = s.elapsed_time(e) * 1e-3
        flop_all += f
        secs_all += dt
        if conv_block_excluded(name, shape):
            continue
        flop += f
        xflop += executed_flop(name, f)
        secs += dt
        d = per_kernel.setdefault(name, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += f
        d[2] += dt
    tf = flop / secs / 1e12
    return {"workload": desc, "steps": steps, "warmup": warmup,
            "images_per_s": round(args.batch_size * steps / el, 2), "ms_per_step": round(el / steps * 1e3, 3),
            "peak": round(MFMA_F32_PEAK / 1e12, 1),
            "frac": round(xflop / secs / MFMA_F32_PEAK, 4),                  # == frac_executed (one convention on this line)
            "frac_executed": round(xflop / secs / MFMA_F32_PEAK, 4), "executed_tflops": round(xflop / secs / 1e12, 2),
            "frac_algorithmic": round(tf * 1e12 / MFMA_F32_PEAK, 4), "algorithmic_tflops": round(tf, 2),
            "conv_block_gflop_per_step": round(flop / 2 / 1e9, 1), "conv_block_kernel_ms_per_step": round(secs / 2 * 1e3, 3),
            "all_gemm_tflops": round(flop_all / secs_all / 1e12, 2),
            "definition": "residual-block 3x3/1x1 convs of SNGANGenerator64 + SNGANDiscriminator64, fwd + dgrad + wgrad, over "
                          "the summed HIP-event launch time of those launches in 2 un-timed global steps; l1, c6 fwd/wgrad, "
                          "head, BN, SN, loss, Adam excluded from both.  frac = frac_executed: multiply-accumulates the "
                          "kernels EXECUTE on the matrix pipe (Winograd F(2x2) launches at 16/36 of the direct convolution, "
                          "F(4x4) and the pooled launches at 9/36) x 2 / time / 157.3 TFLOP/s -- how busy the hardware is "
                          "(north_star's 60 % bar is NOT met on this basis).  frac_algorithmic: the direct convolution's "
                          "2*M*Co*R*S*Ci of SURVEY 8(d) in full / time / peak -- the work the reference's conv costs; it "
                          "exceeds 1 because the Winograd kernels skip 56-75 % of those products",
            "kernels": {k: {"launches": v[0], "algorithmic_tflops": round(v[1] / v[2] / 1e12, 2),
                            "executed_tflops": round(executed_flop(k, v[1]) / v[2] / 1e12, 2),
                            "frac_of_own_pipe": round(executed_flop(k, v[1]) / v[2] / kernel_peak(k), 4),
                            "ms_per_step": round(v[2] / 2 * 1e3, 3)}
                        for k, v in sorted(per_kernel.items())}}


def phase2_leg(args, device, steps=10, warmup=3):
    """BASELINE.json configs[2] on the driver's own line: the SNGAN-32 global step of phase 2 -- n_dis x (D update + D_drs
    update) + one G update (reference loop: diagan-pkg/diagan/trainer/trainer.py:267-277, CLI train_mimicry_phase2.py:102-153)
    -- at batch 64 on one GPU.  Un-scored leg after the timed region."""
    dataset, res, desc = WORKLOADS['sngan32']
    nets = build_models(dataset, args.loss_type, 2, device)
    gen = torch.Generator().manual_seed(4321)
    batches = [(torch.rand(args.batch_size, 3, res, res, generator=gen) * 2 - 1).to(device) for _ in range(4 * args.n_dis)]
    step = make_global_step(*nets, batches, args.n_dis, num_steps=50000, device=device)
    for _ in range(warmup):
        step()
    from diagan.utils.settings import quiesce_gc
    quiesce_gc()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return {"workload": desc + " + D_drs (phase 2, ldr_conf_0.3_ratio_50 sampling is host-side and off the step)",
            "steps": steps, "warmup": warmup, "images_per_s": round(args.batch_size * steps / el, 2),
            "ms_per_step": round(el / steps * 1e3, 3)}


def stylegan2_leg(device, size=256, batch=32, warmup=3, steps=8, table_iters=4):
    """BASELINE.json configs[4] on the driver's own line: iterations of stylegan2/train_ffhq.py (D step, G step, EMA; lazy R1
    every 16th and path-length regularisation every 4th iteration; reference loop stylegan2/train_ffhq.py:163-382, nets
    diagan-pkg/diagan/models/stylegan2.py:224-265,274-285,597-614) at 256 x 256, batch 32, one GPU.  Un-scored leg after the timed
    region.  Iteration 0 (warm-up) runs both regularisers; the timed iterations 3 .. 10 hold two path-length passes (the steady
    1-in-4) and no R1 pass (steady state: 1 in 16).  gemm_*: HIP events around every convolution / weight-gradient launch of
    `table_iters` further iterations (one path-length pass among four)."""
    from diagan.ops import conv as C
    step = make_stylegan2_step(size, batch, 1, device)
    for _ in range(warmup):
        step()
    from diagan.utils.settings import quiesce_gc
    quiesce_gc()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    C.TIMER = full = C.KernelTimer()
    try:
        for _ in range(table_iters):
            step()
        torch.cuda.synchronize()
    finally:
        C.TIMER = None
    # the OPT-IN two-piece mode of the large split-operand kernels (DIAGAN_X3_PIECES=2; NOT the default, operands at ~2^-16: DESIGN 3.2) over
    # a window of the same shape: iterations 15, 16 (R1 + path length) un-timed, 17 .. 24 timed (two path-length passes, no R1)
    two = None
    try:
        C.set_x3_pieces(2)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t1
        two = {"images_per_s": round(batch * steps / el2, 2), "ms_per_iter": round(el2 / steps * 1e3, 3),
               "note": "opt-in (DIAGAN_X3_PIECES=2): third piece pair of the large split-operand kernels dropped, 1.5e-5-2e-5 of the output "
                       "scale per layer against float64 (three pieces: 3e-7-7e-7); the parity tests pass in this mode as well; not the default"}
    except Exception as e:       # noqa: BLE001  (an extra: never fails the leg)
        two = {"error": repr(e)}
    finally:
        C.set_x3_pieces(None)
    flop = xflop = secs = pipe_s = 0.0
    for name, d in full.summary().items():
        flop += d['flop']
        xflop += executed_flop(name, d['flop'])
        pipe_s += executed_flop(name, d['flop']) / kernel_peak(name)       # seconds of the kernel's OWN pipe at its peak
        secs += d['seconds']
    ms_iter, gemm_ms = el / steps * 1e3, secs / table_iters * 1e3
    return {"workload": WORKLOADS['stylegan2'][2], "steps": steps, "warmup": warmup,
            "images_per_s": round(batch * steps / el, 2), "ms_per_iter": round(ms_iter, 3),
            "gemm_ms_per_iter": round(gemm_ms, 3), "non_gemm_ms": round(ms_iter - gemm_ms, 3),
            "gemm_frac_executed": round(pipe_s / secs, 4),
            "gemm_executed_tflops": round(xflop / secs / 1e12, 2),
            "gemm_frac_algorithmic": round(flop / secs / MFMA_F32_PEAK, 4),
            "opt_in_two_pieces": two,
            "definition": "gemm_*: convolution, data- and weight-gradient launches (implicit GEMM + Winograd kernels) by HIP events "
                          "in 4 un-timed iterations; non_gemm_ms = ms_per_iter - gemm_ms_per_iter (FIR, activations, modulation, "
                          "autograd glue, optimiser); gemm_frac_executed = matrix-pipe seconds at peak / launch seconds, every kernel priced "
                          "against the pipe it runs on (fp32 MFMA 157.3 TFLOP/s; the split-operand kernels 416.7 TFLOP/s fp32-equivalent); "
                          "gemm_frac_algorithmic: direct-convolution FLOP / time / 157.3 TFLOP/s"}


def logit_pass_leg(device, N=50000, loader_batch=64):
    """`_get_logit` images/s, reported separately -- the full pass of D over the dataset that fills one row of
    the logit record (reference: trainer.py:142-156; eval mode, as train_mimicry_phase1.py's --save_eval_logits default), on
    a CIFAR-10-sized synthetic dataset (N = 50 000, 32x32) with the training loader's batch of 64."""
    import tempfile
    from diagan.cli import make_loader
    from diagan.datasets.predefined import get_predefined_dataset
    from diagan.models.predefined_models import get_gan_model
    from diagan.trainer.trainer import LogTrainer
    from diagan.utils.plot import LogitRecord
    netG, netD, optG, optD = get_gan_model('cifar10', model='sngan', loss_type='ns')
    ds = get_predefined_dataset('cifar10', num_data=N)
    dl = make_loader(ds, loader_batch)
    with tempfile.TemporaryDirectory() as tmp:
        t = LogTrainer(output_path=tmp, netD=netD, netG=netG, optD=optD, optG=optG, dataloader=dl, num_steps=1, log_dir=tmp,
                       device=device)
        rec = LogitRecord(N, capacity=4, device=t.device)
        t._get_logit(netD, eval_mode=True, record=rec, step=0)         # warm-up pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        row = t._get_logit(netD, eval_mode=True, record=rec, step=1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        finite = bool(torch.isfinite(row).all().item())
        t.logger.close_writers()
    return {"images_per_s": round(N / dt, 1), "seconds": round(dt, 4), "N": N, "loader_batch": loader_batch,
            "eval_batch": t.logit_eval_batch, "mode": "eval", "record_row_finite": finite,
            "includes": "host -> device copy of the images (dataset resident in host memory), D forward, scatter by index"}


def cpu_model_name():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def scorer_leg(device, N=50000, T=50):
    """Second half of BASELINE.json's metric: LDR-score max-abs-err of the HIP scorer against the oracle
    on a SURVEY §8(d)-shaped record (T=50 snapshots of N=50000 logits), plus its run time."""
    import numpy as np
    from diagan.utils.plot import LogitRecord, ldr_scores_device
    from oracle import scorer as osc
    rng = np.random.default_rng(0)
    mu, sg = rng.normal(0.0, 2.0, size=N), rng.uniform(0.05, 1.0, size=N)
    logits = {35000 + 100 * t: (mu + sg * rng.normal(size=N)).astype(np.float32).astype(np.float64) for t in range(T)}
    rec = LogitRecord.from_dict(logits, device=device)
    rows = rec.window(0, 10 ** 9)
    ldr_scores_device(rows)                                   # warm-up
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        stats, conf, tv = ldr_scores_device(rows)
    e.record()
    torch.cuda.synchronize()
    gpu_ms = s.elapsed_time(e) / 10
    # what the phase-2 command line consumes: the four statistics + ONE ldr_conf row (calculate_scores(keys=[...]))
    ldr_scores_device(rows, t_values=[0.3])
    torch.cuda.synchronize()
    s.record()
    for _ in range(10):
        ldr_scores_device(rows, t_values=[0.3])
    e.record()
    torch.cuda.synchronize()
    gpu_ms_one = s.elapsed_time(e) / 10
    t0 = time.time()
    ref = osc.calculate_scores_c(logits, 0, 10 ** 9)
    cpu_ms = (time.time() - t0) * 1e3
    err = 0.0
    for k in ("ldr", "ldrd", "ldrv", "ldrm"):
        err = max(err, float(np.abs(stats[k].cpu().numpy() - ref[k]).max()))
    for j, t in enumerate(tv):
        if f"{t:.1f}" in ("0.3", "5.0"):
            err = max(err, float(np.abs(conf[j].cpu().numpy() - ref[f"ldr_conf_{t:.1f}_ratio_50"]).max()))
    bytes_alg = 8.0 * T * N + 8.0 * N * (4 + len(tv))
    return {"record": f"T={T} x N={N} float64", "max_abs_err": err, "gpu_ms_all_103_scores": round(gpu_ms, 4),
            "gpu_ms_requested_key_only": round(gpu_ms_one, 4),
            "cpu_oracle_ms_1_core": round(cpu_ms, 1), "algorithmic_GBps": round(bytes_alg / gpu_ms / 1e6, 1)}


def self_launch(args):
    """--gpus N > 1 without a torch.distributed.run environment: start the N ranks as a child process group and relay
    its output.  The parent only counts devices and spawns: it never execs (torch.cuda.device_count() may bring the HIP
    runtime up in this process on builds without amdsmi), and every rank is a fresh child of torch.distributed.run.
    A rank whose native RCCL context does not come up in time leaves with a non-zero status and a flag file
    (diagan/trainer/distributed.py::_StartupWatchdog); the job is then started ONCE more, as new processes, with the
    exchange on torch.distributed's process group."""
    import socket
    import subprocess
    import tempfile
    ndev = torch.cuda.device_count()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or args.gpus) // args.gpus)))
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (the device path has no CPU fallback)")
    if ndev < args.gpus and "DIAGAN_DIST_BACKEND" not in env:
        # fewer devices than ranks (a one-GPU test box): the ranks share devices, which RCCL refuses -> gloo;
        # a functional run of the N-rank path, not a scaling measurement (config.devices says so)
        print(f"WARNING: --gpus {args.gpus} on a node with {ndev} device(s): ranks share devices, exchange over gloo",
              file=sys.stderr)
        env["DIAGAN_DIST_BACKEND"] = "gloo"
    flag = os.path.join(tempfile.mkdtemp(prefix="diagan_bench_"), "native_comm_timeout")
    env["DIAGAN_COMM_TIMEOUT_FLAG"] = flag
    rc = 1
    for attempt in range(2):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        rc = subprocess.run(cmd, env=env).returncode
        if rc == 0 or not os.path.exists(flag) or env.get("DIAGAN_COMM", "").lower() == "torch":
            break
        print("WARNING: the native RCCL context did not come up; starting the ranks again with DIAGAN_COMM=torch",
              file=sys.stderr)
        os.remove(flag)
        env["DIAGAN_COMM"] = "torch"
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="sngan32", choices=sorted(WORKLOADS))
    ap.add_argument("--phase", type=int, default=1, choices=(1, 2))
    ap.add_argument("--loss_type", default="ns")
    ap.add_argument("--batch_size", type=int, default=64)
    ap.add_argument("--n_dis", type=int, default=5)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_kernel_timer", action="store_true")
    ap.add_argument("--no_sngan64_leg", action="store_true",
                    help="skip the extra (un-scored) SNGAN-64 conv-block roofline leg of the default sngan32 run")
    ap.add_argument("--graph", action="store_true",
                    help="replay the global step as one hipGraph (launch-bound workloads: dcgan); single GPU only")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    from diagan.trainer import distributed as dist
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:             # before any rendezvous: every rank sees the same mismatch and leaves
        if int(os.environ.get("RANK", "0")) == 0:
            print(f"ERROR: --gpus {args.gpus} but WORLD_SIZE={env_world}: refusing to report a line for a job of a "
                  f"different size", file=sys.stderr)
        sys.exit(2)
    rank, local_rank, world = dist.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the device path has no CPU fallback)")
    dev_index = local_rank % torch.cuda.device_count()     # == local_rank on a real multi-GPU node
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dataset, res, desc = WORKLOADS[args.workload]
    devices = [dev_index]
    if world > 1:
        devices = [int(d) for d in dist.all_gather(dev_index)]

    if args.workload == 'stylegan2':
        if args.graph:
            raise SystemExit("--graph is for the SNGAN / DCGAN steps")
        if args.batch_size == 64:
            args.batch_size = 32              # the reference's per-GPU batch for this configuration
        args.n_dis, args.no_cpu_baseline = 1, True      # one D update per iteration; no CPU restatement is timed
        args.loss_type = 'logistic (non-saturating) + lazy R1 / path-length'
        step = make_stylegan2_step(res, args.batch_size, args.phase, device)
    else:
        netG, netD, netD_drs, optG, optD, optD_drs = build_models(dataset, args.loss_type, args.phase, device)
        if world > 1:
            for n in (netG, netD, netD_drs):
                if n is not None:
                    dist.broadcast_module_(n)
            dist.seed_device_per_rank(1)        # identical replicas, per-rank latent noise (shared CPU generators)
        gen = torch.Generator().manual_seed(1234 + rank)          # SURVEY §8(d) synthetic inputs
        pool = 2 * args.n_dis
        batches = [(torch.rand(args.batch_size, 3, res, res, generator=gen) * 2 - 1).to(device) for _ in range(pool)]
        step = make_global_step(netG, netD, netD_drs, optG, optD, optD_drs, batches, args.n_dis,
                                num_steps=50000 if dataset == 'cifar10' else 75000, device=device)

    from diagan.ops import conv as C
    # Roofline instrumentation: the last warm-up step is bracketed launch by launch to find the dominant
    # GEMM kernel; inside the timed region only THAT kernel's launches carry HIP events (bracketing all
    # ~250 GEMM launches of a step costs ~2 ms of host time per step); the per-kernel table in the output
    # comes from two more un-timed steps after the timed region.
    timer, dominant = None, None
    for i in range(args.warmup):
        if i == args.warmup - 1 and not args.no_kernel_timer:
            C.TIMER = C.KernelTimer()
        step()
    dominant_outside = False
    if C.TIMER is not None:
        torch.cuda.synchronize()
        dominant, dstat = max(C.TIMER.summary().items(), key=lambda kv: kv[1]['seconds'])
        # launch-bound workloads (MNIST-DCGAN: hundreds of ~10 us launches of the dominant kernel per step): a pair of HIP
        # events per launch would slow the very thing being timed (16 vs 10 ms per step); its launches are then bracketed
        # in the un-timed steps after the timed region instead
        dominant_outside = dstat['launches'] > 60
        C.TIMER = None
    if not args.no_kernel_timer:
        timer = C.KernelTimer(only=(set() if dominant_outside else {dominant}) if dominant else None)
    eager_step = step
    if args.graph:
        # the captured launches read the n_dis real batches from fixed tensors; every replay gets fresh data copied in
        from diagan.utils.graph import GraphedStep
        graphed = GraphedStep(eager_step.device_part, (netG, netD, netD_drs), (optG, optD, optD_drs), warmup=2,
                              after=eager_step.host_part)
        graphed.capture()
        step = graphed
        timer = None            # launches inside a graph carry no events; the kernel table comes from eager steps below
    from diagan.utils.settings import quiesce_gc
    quiesce_gc()          # as LogTrainer.train / StyleGAN2Trainer.train do after their first step (no full GC pass over the nets mid-run)
    dist.synchronize()
    torch.cuda.synchronize()
    C.TIMER = timer
    t0, c0 = time.perf_counter(), time.thread_time()
    for _ in range(args.steps):
        step()
    issue_s, issue_cpu_s = time.perf_counter() - t0, time.thread_time() - c0      # the host's launch loop alone
    torch.cuda.synchronize()
    dist.synchronize()
    elapsed = time.perf_counter() - t0
    C.TIMER = None
    host_rows = [(issue_s, issue_cpu_s, len(os.sched_getaffinity(0)))]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = t.item()
        host_rows = dist.all_gather(host_rows[0])
    # two more UN-timed steps on EVERY rank (each step contains the gradient all-reduces): every GEMM launch is
    # bracketed for the per-kernel table rank 0 prints
    summ_all = None
    if args.graph and not args.no_kernel_timer:
        timer = C.KernelTimer(only={dominant} if dominant else None)      # roofline leg of a graph run: eager steps
        C.TIMER = timer
        for _ in range(2):
            eager_step()
        torch.cuda.synchronize()
        C.TIMER = None
    if timer is not None:
        C.TIMER = full = C.KernelTimer()
        for _ in range(2):
            eager_step()
        torch.cuda.synchronize()
        C.TIMER = None
        summ_all = full.summary()
        if dominant_outside:
            timer = full
        dist.synchronize()
    # Extra leg: north_star's >= 60 % MFMA-roofline target on the SNGAN-64 conv blocks (default workload, one GPU only)
    s64, s64_error = None, None
    if (world == 1 and args.workload == 'sngan32' and args.phase == 1 and not args.no_sngan64_leg and not args.graph
            and not args.no_kernel_timer):
        try:
            s64 = sngan64_leg(args, device)
        except Exception as e:          # the extra leg must never cost the scored line
            s64_error = repr(e)
        finally:
            C.TIMER = None
    # Extra legs (default workload, one GPU): the two SURVEY 8(d) figures the scored line does not carry -- phase 2 (+ D_drs,
    # BASELINE configs[2]) and the logit-record pass.  Un-scored, after the timed region, exception-proof.
    extra_legs = {}
    if (world == 1 and args.workload == 'sngan32' and args.phase == 1 and not args.no_sngan64_leg and not args.graph):
        for key, leg in (("phase2", lambda: phase2_leg(args, device)), ("logit_pass", lambda: logit_pass_leg(device)),
                         ("stylegan2", lambda: stylegan2_leg(device))):
            try:
                extra_legs[key] = leg()
            except Exception as e:
                extra_legs[key] = {"error": repr(e)}
    if rank != 0:
        return

    images = args.batch_size * args.steps * world
    line = {
        "metric": "images/sec (G+D step)",
        "value": round(images / elapsed, 2),
        "unit": "images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": desc + (" + D_drs (phase 2)" if args.phase == 2 else ""),
                   "global_batch": args.batch_size * world, "n_dis": args.n_dis, "loss_type": args.loss_type,
                   "parallelism": f"dp{world}", "launch": "hipGraph replay" if args.graph else "eager",
                   "comm": dist.comm_description(), "ranks": world, "devices": devices,
                   "steps_per_s": round(args.steps / elapsed, 3),
                   "D_updates_per_s": round(args.steps * args.n_dis / elapsed, 3)},
    }
    # the host side of the timed region, rank by rank: wall time of the launch loop before the final synchronisation (the
    # loop is throttled by the HIP queue once that is ~80 ms deep: profiles/r03_host_time.md) and the CPU time the launching
    # thread itself consumed -- with N ranks sharing the box's hardware threads the second is the host margin per step
    line["host"] = {"launch_loop_ms_per_step": [round(r[0] / args.steps * 1e3, 3) for r in host_rows],
                    "launch_thread_cpu_ms_per_step": [round(r[1] / args.steps * 1e3, 3) for r in host_rows],
                    "hardware_threads_available": host_rows[0][2], "cpu_count": os.cpu_count()}
    if timer is not None:
        summ = timer.summary()
        dom = max(summ.items(), key=lambda kv: kv[1]['seconds'])
        name, d = dom
        algorithmic = d['flop'] / d['seconds'] / 1e12
        achieved = executed_flop(name, d['flop']) / d['seconds'] / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.workload, {}).get(name)
            except Exception:
                traffic = None
        # algorithmic bytes of the dominant kernel's launches (3x3 Winograd kernels: input + output + weights, fp32; the
        # pooled / up-sampled-input forms read or write a quarter of the pixels): what `traffic` is to be compared with
        alg_bytes = None
        if name.startswith("conv_wino"):
            tot = 0.0
            for key, v in timer.by_shape().items():
                if key[0] != name or len(key) < 4:
                    continue
                M, Co, K = key[1], key[2], key[3]
                # parsed family and template arguments, not name suffixes (conv_wino_pool_kernel's LAST template
                # argument is its column-block count, its second one says whether it is the un-pooling data gradient)
                m4 = re.match(r"conv_wino4\w*_kernel<\s*\d+\s*,\s*(\d+)", name)
                mp = re.match(r"conv_wino_pool_kernel<\s*\d+\s*,\s*(true|false|[01])", name)
                mode = int(m4.group(1)) if m4 else None
                unpool = mp is not None and mp.group(1) in ("true", "1")
                m_in = M / 4 if (mode in (2, 3) or unpool) else M
                m_out = M / 4 if (mode == 1 or (mp is not None and not unpool)) else M
                tot += v['launches'] * 4.0 * (m_in * K / 9 + m_out * Co + Co * K)
            alg_bytes = tot / d['launches'] if tot else None
        peak = kernel_peak(name)                 # the pipe THIS kernel runs on (fp32 MFMA, or bf16 MFMA with split operands)
        x3 = peak != MFMA_F32_PEAK
        line["roofline"] = {
            "kernel": name, "bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak / 1e12, 1),
            "unit": "TFLOP/s", "frac": round(achieved * 1e12 / peak, 4), "frac_executed": round(achieved * 1e12 / peak, 4),
            "frac_algorithmic": round(algorithmic * 1e12 / peak, 4), "traffic": traffic,
            "algorithmic_bytes_per_launch": round(alg_bytes) if alg_bytes else None,
            "traffic_ratio": round(traffic / alg_bytes, 3) if (traffic and alg_bytes) else None,
            "traffic_source": "profiles/pmc_traffic.json (static: separate rocprofv3 --pmc passes of this command, "
                              "2*FETCH_SIZE + WRITE_SIZE per launch; not measured by this run)" if traffic else None,
            "pipe": ("bf16 MFMA with every fp32 operand split in three (six piece products per fp32 product): the dense bf16 "
                     "peak / 6 = 416.7 TFLOP/s of fp32-equivalent products, which is the peak frac is quoted against"
                     if x3 else "fp32 MFMA (v_mfma_f32_32x32x2_f32)"),
            "launches": d['launches'], "avg_launch_us": round(d['seconds'] / d['launches'] * 1e6, 2),
            "algorithmic_gflop_per_launch": round(d['flop'] / d['launches'] / 1e9, 3),
            "executed_gflop_per_launch": round(executed_flop(name, d['flop']) / d['launches'] / 1e9, 3),
            "algorithmic_tflops": round(algorithmic, 2),
            "accounting": "achieved / frac / frac_executed = multiply-accumulates the kernel EXECUTES on the matrix pipe (x2) "
                          "per second: 16/36 of the direct convolution's 2*M*Co*9*Ci for conv_wino_kernel (Winograd "
                          "F(2x2,3x3)), 9/36 for conv_wino4_kernel (F(4x4,3x3)) and conv_wino_pool_kernel (convolution + "
                          "average pool); algorithmic_tflops / frac_algorithmic count the direct convolution in full (the "
                          "convolution the reference runs) and may exceed the peak",
            "all_gemm_kernels_2_untimed_steps": {
                k: {"launches": v['launches'], "algorithmic_tflops": round(v['flop'] / v['seconds'] / 1e12, 2),
                    "executed_tflops": round(executed_flop(k, v['flop']) / v['seconds'] / 1e12, 2),
                    "frac_of_own_pipe": round(executed_flop(k, v['flop']) / v['seconds'] / kernel_peak(k), 4),
                    "ms_per_step": round(v['seconds'] / 2 * 1e3, 3)} for k, v in sorted(summ_all.items())},
        }
    if s64 is not None:
        line["sngan64_conv_blocks"] = s64
    elif s64_error:
        line["sngan64_conv_blocks"] = {"error": s64_error}
    for key, val in extra_legs.items():
        line[key] = val
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(dataset, res, args.loss_type, args.batch_size, args.n_dis)
        line["ldr_scorer"] = scorer_leg(device)
        try:                                   # BASELINE configs[3]'s record size (CelebA: N = 162 770)
            line["ldr_scorer_celeba"] = scorer_leg(device, N=162770)
        except Exception as e:
            line["ldr_scorer_celeba"] = {"error": repr(e)}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
