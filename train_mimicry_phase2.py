#!/usr/bin/env python
"""Phase 2 of Dia-GAN: score the phase-1 logit record, resample by the score, fine-tune G/D and
train the DRS discriminator.

CLI surface of the reference's train_mimicry_phase2.py (flags :39-56; scorer call :87-93; sampler
:21-34; trainer wiring :128-153) on the MI355X engine.  The score window is [p1_step - 5000, p1_step)
(:90-92); `--resample_score` is a key of calculate_scores' dict, e.g. ldr_conf_0.3_ratio_50.
"""
import argparse
import os
import pickle
import sys
from pathlib import Path

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))

import torch
from torch.utils import data

from diagan.datasets.predefined import get_predefined_dataset
from diagan.datasets.sampler import ShardedSampler, make_weighted_sampler
from diagan.models.predefined_models import get_gan_model
from diagan.trainer import distributed as dist
from diagan.trainer.trainer import LogTrainer
from diagan.utils.plot import calculate_scores, print_num_params
from diagan.utils.settings import set_seed


def get_dataloader(dataset, batch_size=128, weights=None, eps=1e-6, num_workers=8):
    """WeightedRandomSampler over floored weights (reference :21-34).  Under data parallelism every rank
    draws the same multinomial order from the shared CPU seed and keeps every W-th index, so the
    phase-2 weights stay in force (the reference's DDP path drops them)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    sampler = make_weighted_sampler(weights, eps) if weights is not None else None
    if world > 1:
        sampler = ShardedSampler(sampler if sampler is not None else data.RandomSampler(dataset), rank, world)
    return data.DataLoader(dataset=dataset, batch_size=batch_size, shuffle=False if sampler else True,
                           sampler=sampler, num_workers=num_workers, pin_memory=True)


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--dataset", "-d", default="cifar10", type=str)
    parser.add_argument("--root", "-r", default="./dataset/cifar10", type=str, help="dataset dir")
    parser.add_argument("--work_dir", default="./exp_results", type=str, help="output dir")
    parser.add_argument("--exp_name", type=str, help="exp name")
    parser.add_argument("--baseline_exp_name", type=str, help="exp name")
    parser.add_argument('--p1_step', default=40000, type=int)
    parser.add_argument("--model", default="sngan", type=str, help="network model")
    parser.add_argument("--loss_type", default="hinge", type=str, help="loss type")
    parser.add_argument('--gpu', default='0', type=str, help='id(s) for CUDA_VISIBLE_DEVICES (single process only)')
    parser.add_argument('--num_steps', default=80000, type=int)
    parser.add_argument('--batch_size', default=64, type=int)
    parser.add_argument('--seed', default=1, type=int)
    parser.add_argument('--decay', default='linear', type=str)
    parser.add_argument('--n_dis', default=5, type=int)
    parser.add_argument('--resample_score', type=str)
    parser.add_argument('--gold', action='store_true')
    parser.add_argument('--topk', action='store_true')
    # additions
    parser.add_argument('--num_data', type=int)
    parser.add_argument('--window', default=5000, type=int, help='score window in steps (reference: 5000)')
    parser.add_argument('--num_workers', default=0, type=int)
    parser.add_argument('--save_steps', default=1000, type=int)
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    rank, local_rank, world = dist.init_from_env()
    if world == 1:
        os.environ['CUDA_VISIBLE_DEVICES'] = args.gpu
    output_dir = f'{args.work_dir}/{args.exp_name}'
    save_path = Path(output_dir)
    save_path.mkdir(parents=True, exist_ok=True)
    baseline_save_path = Path(f'{args.work_dir}/{args.baseline_exp_name}')

    set_seed(args.seed)
    if not torch.cuda.is_available():
        raise SystemExit("the Dia-GAN engine needs an MI355X (no CPU fallback)")
    device = torch.device("cuda", (local_rank % torch.cuda.device_count()) if world > 1 else 0)
    torch.cuda.set_device(device)

    if not args.gold:
        logit_path = baseline_save_path / 'logits_netD_eval.pkl'
        print(f'Use logit from: {logit_path}')
        with open(logit_path, "rb") as f:
            logits = pickle.load(f)
        score_dict = calculate_scores(logits, start_epoch=args.p1_step - args.window, end_epoch=args.p1_step,
                                      device=device)
        sample_weights = score_dict[args.resample_score]
        print(f'sample_weights mean: {sample_weights.mean()}, var: {sample_weights.var()}, '
              f'max: {sample_weights.max()}, min: {sample_weights.min()}')
    else:
        sample_weights = None

    netG_ckpt_path = baseline_save_path / f'checkpoints/netG/netG_{args.p1_step}_steps.pth'
    netD_ckpt_path = baseline_save_path / f'checkpoints/netD/netD_{args.p1_step}_steps.pth'
    netD_drs_ckpt_path = baseline_save_path / f'checkpoints/netD/netD_{args.p1_step}_steps.pth'   # sic: the netD file
    netG, netD, netD_drs, optG, optD, optD_drs = get_gan_model(dataset_name=args.dataset, model=args.model,
                                                               loss_type=args.loss_type, drs=True, topk=args.topk,
                                                               gold=args.gold)
    print(f'model: {args.model} - netD_drs_ckpt_path: {netD_drs_ckpt_path}')
    print_num_params(netG, netD)

    ds_train = get_predefined_dataset(dataset_name=args.dataset, root=args.root, weights=None, num_data=args.num_data)
    dl_train = get_dataloader(ds_train, batch_size=args.batch_size, weights=sample_weights,
                              num_workers=args.num_workers)
    ds_drs = get_predefined_dataset(dataset_name=args.dataset, root=args.root, weights=None, num_data=args.num_data)
    dl_drs = get_dataloader(ds_drs, batch_size=args.batch_size, weights=None, num_workers=args.num_workers)
    print(args)

    trainer = LogTrainer(output_path=save_path, netD=netD, netG=netG, optD=optD, optG=optG,
                         netG_ckpt_file=str(netG_ckpt_path), netD_ckpt_file=str(netD_ckpt_path),
                         netD_drs_ckpt_file=str(netD_drs_ckpt_path), netD_drs=netD_drs, optD_drs=optD_drs,
                         dataloader_drs=dl_drs, n_dis=args.n_dis, num_steps=args.num_steps,
                         save_steps=args.save_steps, lr_decay=args.decay, dataloader=dl_train, log_dir=output_dir,
                         print_steps=10, device=device, topk=args.topk, gold=args.gold, gold_step=args.p1_step,
                         save_logits=False)
    trainer.train()
    return trainer


if __name__ == '__main__':
    main()
