#!/usr/bin/env python
"""Phase 1 of Dia-GAN: train G/D and record per-sample discriminator logits.

CLI surface of the reference's train_mimicry_phase1.py (flags :29-51, dataset overrides :82-92,
trainer wiring :104-126) on the MI355X engine.  New flags: --num_data / --max_steps (smoke runs on
synthetic data; datasets themselves are outside the accelerated path).
Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N train_mimicry_phase1.py ...`
(one process per GPU, RCCL); batch_size is per GPU.
"""
import argparse
import os
import sys
from pathlib import Path

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))

import torch
from torch.utils import data

from diagan.datasets.predefined import get_predefined_dataset
from diagan.datasets.sampler import ShardedSampler
from diagan.models.predefined_models import get_gan_model
from diagan.trainer import distributed as dist
from diagan.trainer.trainer import LogTrainer
from diagan.utils.plot import print_num_params
from diagan.utils.settings import set_seed


def get_dataloader(dataset, batch_size=128, num_workers=8):
    rank, world = dist.get_rank(), dist.get_world_size()
    if world > 1:     # same shuffled order on every rank (shared CPU seed), rank r takes every W-th index
        sampler = ShardedSampler(data.RandomSampler(dataset), rank, world)
        return data.DataLoader(dataset=dataset, batch_size=batch_size, sampler=sampler, num_workers=num_workers,
                               pin_memory=True)
    return data.DataLoader(dataset=dataset, batch_size=batch_size, shuffle=True, num_workers=num_workers,
                           pin_memory=True)


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--dataset", "-d", default="cifar10", type=str)
    parser.add_argument("--root", "-r", default="./dataset/cifar10", type=str, help="dataset dir")
    parser.add_argument("--work_dir", default="./exp_results", type=str, help="output dir")
    parser.add_argument("--exp_name", default="cifar10", type=str, help="exp name")
    parser.add_argument("--model", default="sngan", type=str, help="network model")
    parser.add_argument("--loss_type", default="hinge", type=str, help="loss type")
    parser.add_argument('--gpu', default='0', type=str, help='id(s) for CUDA_VISIBLE_DEVICES (single process only)')
    parser.add_argument('--num_pack', default=1, type=int)
    parser.add_argument('--batch_size', default=64, type=int)
    parser.add_argument('--seed', default=1, type=int)
    parser.add_argument('--download_dataset', action='store_true')
    parser.add_argument('--topk', action='store_true')
    parser.add_argument('--num_steps', default=100000, type=int)
    parser.add_argument('--logit_save_steps', default=100, type=int)
    parser.add_argument('--decay', default='linear', type=str)
    parser.add_argument('--n_dis', default=5, type=int)
    parser.add_argument('--imb_factor', default=0.1, type=float)
    parser.add_argument('--celeba_class_attr', default='glass', type=str)
    parser.add_argument('--ckpt_step', type=int)
    parser.add_argument('--no_save_logits', action='store_true')
    parser.add_argument('--save_logit_after', default=30000, type=int)
    parser.add_argument('--stop_save_logit_after', default=60000, type=int)
    # additions
    parser.add_argument('--num_data', type=int, help='synthetic dataset size (default: the real dataset size)')
    parser.add_argument('--max_steps', type=int, help='override the per-dataset step schedule (smoke runs)')
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

    set_seed(args.seed)
    if not torch.cuda.is_available():
        raise SystemExit("the Dia-GAN engine needs an MI355X (no CPU fallback)")
    device = torch.device("cuda", (local_rank % torch.cuda.device_count()) if world > 1 else 0)
    torch.cuda.set_device(device)

    netG, netD, optG, optD = get_gan_model(dataset_name=args.dataset, model=args.model, loss_type=args.loss_type,
                                           topk=args.topk)
    print_num_params(netG, netD)
    ds_train = get_predefined_dataset(dataset_name=args.dataset, root=args.root, num_data=args.num_data)
    dl_train = get_dataloader(ds_train, batch_size=args.batch_size, num_workers=args.num_workers)

    if args.dataset == 'celeba':
        args.num_steps, args.logit_save_steps = 75000, 100
        args.save_logit_after, args.stop_save_logit_after = 55000, 60000
    if args.dataset == 'cifar10':
        args.num_steps, args.logit_save_steps = 50000, 100
        args.save_logit_after, args.stop_save_logit_after = 35000, 40000
    if args.max_steps:
        scale = args.max_steps / args.num_steps
        args.save_logit_after = int(args.save_logit_after * scale)
        args.stop_save_logit_after = int(args.stop_save_logit_after * scale)
        args.logit_save_steps = max(1, int(args.logit_save_steps * scale))
        args.num_steps = args.max_steps
    print(args)

    if args.ckpt_step:
        netG_ckpt_file = save_path / f'checkpoints/netG/netG_{args.ckpt_step}_steps.pth'
        netD_ckpt_file = save_path / f'checkpoints/netD/netD_{args.ckpt_step}_steps.pth'
    else:
        netG_ckpt_file = netD_ckpt_file = None
    if world > 1:
        netG.to(device), netD.to(device)
        dist.broadcast_module_(netG)
        dist.broadcast_module_(netD)

    trainer = LogTrainer(output_path=save_path, logit_save_steps=args.logit_save_steps,
                         netG_ckpt_file=netG_ckpt_file, netD_ckpt_file=netD_ckpt_file, netD=netD, netG=netG,
                         optD=optD, optG=optG, n_dis=args.n_dis, num_steps=args.num_steps,
                         save_steps=args.save_steps, lr_decay=args.decay, dataloader=dl_train, log_dir=output_dir,
                         print_steps=10, device=device, topk=args.topk, save_logits=not args.no_save_logits,
                         save_logit_after=args.save_logit_after, stop_save_logit_after=args.stop_save_logit_after)
    trainer.train()
    return trainer


if __name__ == '__main__':
    main()
