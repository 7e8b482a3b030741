"""Command-line front ends of the two Dia-GAN phases on the MI355X engine.

The reference's `train_mimicry_phase1.py` (flags :29-51, per-dataset schedule :82-92, trainer wiring :104-126) and
`train_mimicry_phase2.py` (flags :39-56, scorer call :87-93, weighted sampler :21-34, trainer wiring :128-153) keep
their flag names, defaults and on-disk layout; the scripts of the same names at the repository root are two-line
wrappers around `phase1()` / `phase2()` below.  Flags are declared as data (one table per phase plus the shared
ones) so that the contract is testable (`tests/test_host_logic.py` compares names and defaults).

Multi-GPU: `python -m torch.distributed.run --nproc-per-node N train_mimicry_phaseK.py ...` -- one process per GPU
over RCCL; `--batch_size` is per GPU; `--gpu` only applies to a single process.
"""
import argparse
import os
import pickle
from pathlib import Path

import torch
from torch.utils import data

from diagan.datasets.predefined import get_predefined_dataset
from diagan.datasets.sampler import ShardedSampler, make_weighted_sampler
from diagan.models.predefined_models import get_gan_model
from diagan.trainer import distributed as dist
from diagan.trainer.trainer import LogTrainer
from diagan.utils.plot import calculate_scores, print_num_params
from diagan.utils.settings import set_seed

_FLAG = "store_true"     # marker for boolean switches in the tables below

# (option strings, default, type or _FLAG, help)
SHARED_FLAGS = [
    (("--dataset", "-d"), "cifar10", str, None),
    (("--root", "-r"), "./dataset/cifar10", str, "dataset dir"),
    (("--work_dir",), "./exp_results", str, "output dir"),
    (("--model",), "sngan", str, "network model"),
    (("--loss_type",), "hinge", str, "loss type"),
    (("--gpu",), "0", str, "id(s) for CUDA_VISIBLE_DEVICES (single process only)"),
    (("--batch_size",), 64, int, None),
    (("--seed",), 1, int, None),
    (("--decay",), "linear", str, None),
    (("--n_dis",), 5, int, None),
    (("--topk",), False, _FLAG, None),
    # not in the reference: smoke runs on synthetic data, loader workers, checkpoint cadence
    (("--num_data",), None, int, "synthetic dataset size (default: the real dataset size)"),
    (("--num_workers",), 0, int, None),
    (("--save_steps",), 1000, int, None),
]
PHASE1_FLAGS = [
    (("--exp_name",), "cifar10", str, "exp name"),
    (("--num_pack",), 1, int, None),
    (("--download_dataset",), False, _FLAG, None),
    (("--num_steps",), 100000, int, None),
    (("--logit_save_steps",), 100, int, None),
    (("--imb_factor",), 0.1, float, None),
    (("--celeba_class_attr",), "glass", str, None),
    (("--ckpt_step",), None, int, None),
    (("--no_save_logits",), False, _FLAG, None),
    (("--save_logit_after",), 30000, int, None),
    (("--stop_save_logit_after",), 60000, int, None),
    (("--max_steps",), None, int, "override the per-dataset step schedule (smoke runs)"),
]
PHASE2_FLAGS = [
    (("--exp_name",), None, str, "exp name"),
    (("--baseline_exp_name",), None, str, "exp name"),
    (("--p1_step",), 40000, int, None),
    (("--num_steps",), 80000, int, None),
    (("--resample_score",), None, str, None),
    (("--gold",), False, _FLAG, None),
    (("--window",), 5000, int, "score window in steps (reference: 5000)"),
    (("--compat_fetch_quirk",), False, _FLAG, "reproduce mimicry's _fetch_data: an exhausted D_drs iterator restarts on the "
                                              "WEIGHTED main loader (what the reference's runs did after their first "
                                              "epoch); default: D_drs keeps its own uniform loader"),
]
# per-dataset phase-1 schedule the reference hard-codes after parsing (train_mimicry_phase1.py:82-92)
PHASE1_SCHEDULE = {
    "celeba": dict(num_steps=75000, logit_save_steps=100, save_logit_after=55000, stop_save_logit_after=60000),
    "cifar10": dict(num_steps=50000, logit_save_steps=100, save_logit_after=35000, stop_save_logit_after=40000),
}


def make_parser(*tables):
    parser = argparse.ArgumentParser()
    for table in tables:
        for names, default, kind, text in table:
            if kind is _FLAG:
                parser.add_argument(*names, action="store_true", help=text)
            else:
                parser.add_argument(*names, default=default, type=kind, help=text)
    return parser


def phase1_parser():
    return make_parser(SHARED_FLAGS, PHASE1_FLAGS)


def phase2_parser():
    return make_parser(SHARED_FLAGS, PHASE2_FLAGS)


def make_loader(dataset, batch_size, num_workers=0, weights=None, floor=1e-6):
    """Phase 1: uniform shuffling.  Phase 2: `WeightedRandomSampler` over weights floored at 1e-6
    (train_mimicry_phase2.py:21-34).  Under data parallelism every rank walks the SAME sampler order (shared CPU
    seed) and keeps every W-th index, so the phase-2 weights stay in force on every rank (the reference's DDP path
    drops them)."""
    sampler = None if weights is None else make_weighted_sampler(weights, floor)
    world = dist.get_world_size()
    if world > 1:
        sampler = ShardedSampler(sampler if sampler is not None else data.RandomSampler(dataset), dist.get_rank(), world)
    # pinned staging only with worker processes (the reference always runs 8 of them): in the main process every
    # batch would pay a synchronous host allocation (~2 ms per tensor, 40 ms per global step)
    return data.DataLoader(dataset=dataset, batch_size=batch_size, shuffle=sampler is None, sampler=sampler,
                           num_workers=num_workers, pin_memory=num_workers > 0)


class _Run:
    """Process / device / output-directory state shared by both phases."""

    def __init__(self, args):
        self.rank, self.local_rank, self.world = dist.init_from_env()
        if self.world == 1:
            os.environ['CUDA_VISIBLE_DEVICES'] = args.gpu
        self.out_dir = f'{args.work_dir}/{args.exp_name}'
        self.save_path = Path(self.out_dir)
        self.save_path.mkdir(parents=True, exist_ok=True)
        set_seed(args.seed)
        self.seed = args.seed
        if not torch.cuda.is_available():
            raise SystemExit("the Dia-GAN engine needs an MI355X (no CPU fallback)")
        index = self.local_rank % torch.cuda.device_count() if self.world > 1 else 0
        self.device = torch.device("cuda", index)
        torch.cuda.set_device(self.device)

    def replicate(self, *nets):
        """Data parallel start: identical parameters and buffers on every rank."""
        if self.world > 1:
            for net in nets:
                net.to(self.device)
                dist.broadcast_module_(net)
            # identical replicas, different latent noise / dropout masks per rank (the CPU generators stay shared)
            dist.seed_device_per_rank(self.seed)


def _checkpoint(root, net, step):
    return root / f'checkpoints/{net}/{net}_{step}_steps.pth'


def phase1(argv=None):
    """Train G/D and record per-sample discriminator logits."""
    args = phase1_parser().parse_args(argv)
    run = _Run(args)
    netG, netD, optG, optD = get_gan_model(dataset_name=args.dataset, model=args.model, loss_type=args.loss_type,
                                           topk=args.topk)
    print_num_params(netG, netD)
    train_set = get_predefined_dataset(dataset_name=args.dataset, root=args.root, num_data=args.num_data)
    loader = make_loader(train_set, args.batch_size, args.num_workers)

    for key, value in PHASE1_SCHEDULE.get(args.dataset, {}).items():
        setattr(args, key, value)
    if args.max_steps:                       # shrink the whole schedule proportionally
        ratio = args.max_steps / args.num_steps
        args.save_logit_after = int(args.save_logit_after * ratio)
        args.stop_save_logit_after = int(args.stop_save_logit_after * ratio)
        args.logit_save_steps = max(1, int(args.logit_save_steps * ratio))
        args.num_steps = args.max_steps
    print(args)

    resume = {}
    if args.ckpt_step:
        resume = dict(netG_ckpt_file=_checkpoint(run.save_path, 'netG', args.ckpt_step),
                      netD_ckpt_file=_checkpoint(run.save_path, 'netD', args.ckpt_step))
    run.replicate(netG, netD)
    trainer = LogTrainer(output_path=run.save_path, log_dir=run.out_dir, device=run.device, dataloader=loader,
                         netD=netD, netG=netG, optD=optD, optG=optG, n_dis=args.n_dis, num_steps=args.num_steps,
                         lr_decay=args.decay, topk=args.topk, print_steps=10, save_steps=args.save_steps,
                         logit_save_steps=args.logit_save_steps, save_logits=not args.no_save_logits,
                         save_logit_after=args.save_logit_after, stop_save_logit_after=args.stop_save_logit_after,
                         netG_ckpt_file=resume.get('netG_ckpt_file'), netD_ckpt_file=resume.get('netD_ckpt_file'))
    trainer.train()
    return trainer


def phase2(argv=None):
    """Score the phase-1 logit record, resample by the score, fine-tune G/D and train the DRS discriminator."""
    args = phase2_parser().parse_args(argv)
    run = _Run(args)
    baseline = Path(f'{args.work_dir}/{args.baseline_exp_name}')

    weights = None
    if not args.gold:
        record = baseline / 'logits_netD_eval.pkl'
        print(f'Use logit from: {record}')
        with open(record, "rb") as f:
            logits = pickle.load(f)
        # window [p1_step - 5000, p1_step) of the record (train_mimicry_phase2.py:90-92); scored on the device
        scores = calculate_scores(logits, start_epoch=args.p1_step - args.window, end_epoch=args.p1_step,
                                  device=run.device, keys=[args.resample_score])
        weights = scores[args.resample_score]
        print(f'sample_weights mean: {weights.mean()}, var: {weights.var()}, max: {weights.max()}, min: {weights.min()}')

    netG, netD, netD_drs, optG, optD, optD_drs = get_gan_model(dataset_name=args.dataset, model=args.model,
                                                               loss_type=args.loss_type, drs=True, topk=args.topk,
                                                               gold=args.gold)
    start = {net: str(_checkpoint(baseline, net, args.p1_step)) for net in ('netG', 'netD')}
    drs_start = start['netD']               # sic: D_drs starts from the phase-1 netD file (reference :100-101)
    print(f'model: {args.model} - netD_drs_ckpt_path: {drs_start}')
    print_num_params(netG, netD)

    def fresh_set():
        return get_predefined_dataset(dataset_name=args.dataset, root=args.root, weights=None, num_data=args.num_data)
    loader = make_loader(fresh_set(), args.batch_size, args.num_workers, weights=weights)
    loader_drs = make_loader(fresh_set(), args.batch_size, args.num_workers)
    print(args)

    trainer = LogTrainer(output_path=run.save_path, log_dir=run.out_dir, device=run.device, dataloader=loader,
                         dataloader_drs=loader_drs, netD=netD, netG=netG, netD_drs=netD_drs, optD=optD, optG=optG,
                         optD_drs=optD_drs, netG_ckpt_file=start['netG'], netD_ckpt_file=start['netD'],
                         netD_drs_ckpt_file=drs_start, n_dis=args.n_dis, num_steps=args.num_steps,
                         lr_decay=args.decay, topk=args.topk, gold=args.gold, gold_step=args.p1_step, print_steps=10,
                         save_steps=args.save_steps, save_logits=False, compat_fetch_quirk=args.compat_fetch_quirk)
    trainer.train()
    return trainer


# ---- Colored-MNIST / mnist_dcgan front ends (BASELINE configs[0]) ----------------------------------------------------
# train_mimicry_color_mnist_phase1.py (flags :48-67, trainer wiring :106-124) and train_mimicry_color_mnist_phase2.py
# (flags :41-61, scorer call :93-97, trainer wiring :128-151).  What differs from the CIFAR/CelebA scripts: ns / hinge
# default losses, n_dis = 1, no LR decay, 20 000 steps, logits recorded in TRAIN mode (`save_eval_logits=False`) and
# only when the discriminator is not packed, checkpoints every 1000 and sample grids every 100 steps, `--topk` an int.
COLOR_MNIST_SHARED = [
    (("--dataset", "-d"), "color_mnist", str, None),
    (("--root", "-r"), "./dataset/colour_mnist", str, "dataset dir"),
    (("--work_dir",), "./exp_results", str, "output dir"),
    (("--exp_name",), "colour_mnist", str, "exp name"),
    (("--model",), "mnistgan", str, "network model"),
    (("--gpu",), "0", str, "id(s) for CUDA_VISIBLE_DEVICES (single process only)"),
    (("--num_pack",), 1, int, None),
    (("--batch_size",), 64, int, None),
    (("--seed",), 1, int, None),
    (("--num_steps",), 20000, int, None),
    (("--logit_save_steps",), 100, int, None),
    (("--decay",), "None", str, None),
    (("--n_dis",), 1, int, None),
    (("--major_ratio",), 0.99, float, None),
    (("--num_data",), 10000, int, None),
    (("--resample_score",), None, str, None),
    (("--num_workers",), 0, int, None),             # not in the reference
]
COLOR_MNIST_PHASE1 = [
    (("--loss_type",), "ns", str, "loss type"),
    (("--use_clipping",), False, _FLAG, None),
    (("--topk",), 0, int, None),
]
COLOR_MNIST_PHASE2 = [
    (("--loss_type",), "hinge", str, "loss type"),
    (("--baseline_exp_name",), "colour_mnist", str, "exp name"),
    (("--p1_step",), 10000, int, None),
    (("--use_eval_logits",), None, int, None),
    (("--compat_fetch_quirk",), False, _FLAG, "as in train_mimicry_phase2.py: restart an exhausted D_drs iterator on the "
                                              "main (weighted) loader, like mimicry's _fetch_data"),
]


def color_mnist_phase1_parser():
    return make_parser(COLOR_MNIST_SHARED, COLOR_MNIST_PHASE1)


def color_mnist_phase2_parser():
    return make_parser(COLOR_MNIST_SHARED, COLOR_MNIST_PHASE2)


def floor_or_clip_weights(weights, clip=False, eps=1e-1):
    """train_mimicry_color_mnist_phase1.py:22-32: weights floored at `eps`, or clipped to
    [max(mean - 2 var, eps), mean + 2 var] (sic: the variance, not the standard deviation)"""
    import numpy as np
    weights = np.asarray(weights, dtype=np.float64)
    if not clip:
        return np.maximum(weights, eps)
    mean, var = weights.mean(), weights.var()
    return np.clip(weights, max(mean - 2 * var, eps), mean + 2 * var)


def _plain_weighted_loader(dataset, batch_size, num_workers, weights=None):
    """the colour-MNIST scripts hand the weights to WeightedRandomSampler as they are (phase 2 :24-37)"""
    sampler = None if weights is None else data.WeightedRandomSampler(weights, len(weights), replacement=True)
    world = dist.get_world_size()
    if world > 1:
        sampler = ShardedSampler(sampler if sampler is not None else data.RandomSampler(dataset), dist.get_rank(), world)
    return data.DataLoader(dataset=dataset, batch_size=batch_size, shuffle=sampler is None, sampler=sampler,
                           num_workers=num_workers, pin_memory=num_workers > 0)


def color_mnist_phase1(argv=None, dataset=None):
    args = color_mnist_phase1_parser().parse_args(argv)
    run = _Run(args)
    netG, netD, optG, optD = get_gan_model(dataset_name=args.dataset, model=args.model, num_pack=args.num_pack,
                                           loss_type=args.loss_type, topk=args.topk == 1)
    print_num_params(netG, netD)
    train_set = get_predefined_dataset(dataset_name=args.dataset, root=args.root, weights=None,
                                       major_ratio=args.major_ratio, num_data=args.num_data, dataset=dataset)
    loader = _plain_weighted_loader(train_set, args.batch_size, args.num_workers)
    print(args)
    run.replicate(netG, netD)
    trainer = LogTrainer(output_path=run.save_path, logit_save_steps=args.logit_save_steps, netD=netD, netG=netG,
                         optD=optD, optG=optG, n_dis=args.n_dis, num_steps=args.num_steps, save_steps=1000,
                         vis_steps=100, lr_decay=args.decay, dataloader=loader, log_dir=run.out_dir, print_steps=10,
                         device=run.device, topk=args.topk, save_logits=args.num_pack == 1, save_eval_logits=False)
    trainer.train()
    return trainer


def color_mnist_phase2(argv=None, dataset=None):
    args = color_mnist_phase2_parser().parse_args(argv)
    run = _Run(args)
    baseline = Path(f'{args.work_dir}/{args.baseline_exp_name}')
    netG, netD, netD_drs, optG, optD, optD_drs = get_gan_model(dataset_name=args.dataset, model=args.model, drs=True,
                                                               loss_type=args.loss_type)
    start = {net: str(_checkpoint(baseline, net, args.p1_step)) for net in ('netG', 'netD')}
    record = baseline / ('logits_netD_eval.pkl' if args.use_eval_logits == 1 else 'logits_netD_train.pkl')
    print(f'Use logit from: {record}')
    with open(record, "rb") as f:
        logits = pickle.load(f)
    scores = calculate_scores(logits, start_epoch=args.p1_step - 5000, end_epoch=args.p1_step, device=run.device,
                              keys=None if args.resample_score is None else [args.resample_score])
    weights = scores[args.resample_score] if args.resample_score is not None else None
    if weights is not None:
        print(f'sample_weights mean: {weights.mean()}, var: {weights.var()}, max: {weights.max()}, min: {weights.min()}')
    print_num_params(netG, netD)

    def fresh_set():
        return get_predefined_dataset(dataset_name=args.dataset, root=args.root, weights=None,
                                      major_ratio=args.major_ratio, num_data=args.num_data, dataset=dataset)
    loader = _plain_weighted_loader(fresh_set(), args.batch_size, args.num_workers, weights=weights)
    loader_drs = _plain_weighted_loader(fresh_set(), args.batch_size, args.num_workers)
    print(args, start['netG'], start['netD'], start['netD'])
    trainer = LogTrainer(output_path=run.save_path, logit_save_steps=args.logit_save_steps, netD=netD, netG=netG,
                         optD=optD, optG=optG, netG_ckpt_file=start['netG'], netD_ckpt_file=start['netD'],
                         netD_drs_ckpt_file=start['netD'], netD_drs=netD_drs, optD_drs=optD_drs,
                         dataloader_drs=loader_drs, n_dis=args.n_dis, num_steps=args.num_steps, save_steps=1000,
                         vis_steps=100, lr_decay=args.decay, dataloader=loader, log_dir=run.out_dir, print_steps=10,
                         device=run.device, save_logits=False, compat_fetch_quirk=args.compat_fetch_quirk)
    trainer.train()
    return trainer
