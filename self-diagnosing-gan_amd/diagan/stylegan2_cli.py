"""Command line of the StyleGAN2 trainers (flags of stylegan2/train_ffhq.py:387-488 and
stylegan2/train_ffhq_phase2.py:409-512), phase 1 and phase 2 behind one table.

    python stylegan2/train_ffhq.py        -d ffhq --batch 32 --exp_name base ...
    python stylegan2/train_ffhq_phase2.py -d ffhq --baseline_exp_name base --resample_score ldr_conf --p1_step 200000 ...

One process per GPU (`python -m torch.distributed.run --nproc-per-node N ...`); rendezvous and the RCCL process group
come from diagan.trainer.distributed.  There is no lmdb / torchvision in this image: unless a dataset object is
handed to `main`, images are synthetic tensors of the dataset's resolution served as (image, index) pairs, the item
format of the reference's MultiResolutionDataset as its trainers consume it."""
import argparse
import os
import pickle
from pathlib import Path

import torch
from torch.utils import data

from diagan.trainer import distributed as dist
from diagan.trainer import stylegan2 as TR
from diagan.utils.settings import set_seed

SIZES = {'cifar10': 32, 'celeba': 64, 'utk_faces': 64, 'imagenet': 128, 'ffhq': 256}      # train_ffhq.py:510-523

COMMON = [
    # flag(s), kwargs
    (("--dataset", "-d"), dict(default="cifar10", type=str)),
    (("--root", "-r"), dict(default="./dataset/cifar10", type=str, help="dataset dir")),
    (("--iter",), dict(type=int, default=800000, help="total training iterations")),
    (("--batch",), dict(type=int, default=16, help="batch sizes for each gpus")),
    (("--n_sample",), dict(type=int, default=64, help="number of the samples generated during training")),
    (("--size",), dict(type=int, default=32, help="image sizes for the model")),
    (("--path_regularize",), dict(type=float, default=2, help="weight of the path length regularization")),
    (("--path_batch_shrink",), dict(type=int, default=2, help="batch size reducing factor for the path length "
                                                               "regularization")),
    (("--d_reg_every",), dict(type=int, default=16, help="interval of the applying r1 regularization")),
    (("--g_reg_every",), dict(type=int, default=4, help="interval of the applying path length regularization")),
    (("--mixing",), dict(type=float, default=0.9, help="probability of latent code mixing")),
    (("--ckpt",), dict(type=str, default=None, help="path to the checkpoints to resume training")),
    (("--lr",), dict(type=float, default=0.002, help="learning rate")),
    (("--channel_multiplier",), dict(type=int, default=2, help="config-f = 2, else = 1")),
    (("--wandb",), dict(action="store_true", help="accepted for compatibility; not used")),
    (("--local_rank",), dict(type=int, default=0, help="local rank for distributed training")),
    (("--augment",), dict(action="store_true", help="non leaking augmentation (not part of the accelerated path)")),
    (("--augment_p",), dict(type=float, default=0)),
    (("--ada_target",), dict(type=float, default=0.6)),
    (("--ada_length",), dict(type=int, default=500 * 1000)),
    (("--ada_every",), dict(type=int, default=256)),
    (("--work_dir",), dict(default="./exp_results", type=str, help="output dir")),
    (("--exp_name",), dict(default="test", type=str, help="exp name")),
    (("--seed",), dict(default=1, type=int)),
    (("--gpu",), dict(type=str)),
    (("--logit_save_steps",), dict(default=100, type=int)),
    # extensions (no reference counterpart)
    (("--num_data",), dict(default=None, type=int, help="size of the synthetic dataset")),
    (("--rank_noise",), dict(action="store_true", help="data parallel: seed every rank's DEVICE generator with seed + rank "
                                                      "(distinct latents / noise maps per rank); default: the reference's "
                                                      "behaviour, one seed for all ranks (stylegan2/train_ffhq.py:497)")),
    (("--log_every",), dict(default=100, type=int)),
    (("--checkpoint_every",), dict(default=5000, type=int)),
]
PHASE = {
    1: [(("--r1",), dict(type=float, default=0.1, help="weight of the r1 regularization")),
        (("--save_logit_after",), dict(default=195000, type=int)),
        (("--stop_save_logit_after",), dict(default=200000, type=int))],
    2: [(("--r1",), dict(type=float, default=10, help="weight of the r1 regularization")),
        (("--save_logit_after",), dict(default=1000000, type=int)),
        (("--baseline_exp_name",), dict(type=str)),
        (("--resample_score",), dict(type=str)),
        (("--p1_step",), dict(default=200000, type=int))],
}


def build_parser(phase):
    parser = argparse.ArgumentParser(description="StyleGAN2 trainer" + (" (phase 2)" if phase == 2 else ""))
    for flags, kw in COMMON + PHASE[phase]:
        parser.add_argument(*flags, **kw)
    return parser


class IndexedImages(data.Dataset):
    """(image in [-1, 1], index) pairs"""

    def __init__(self, num, size, seed=1234):
        g = torch.Generator().manual_seed(seed)
        self.images = torch.rand((num, 3, size, size), generator=g) * 2 - 1

    def __len__(self):
        return len(self.images)

    def __getitem__(self, i):
        return self.images[i], i


def _loader(dataset, args, weights=None):
    return data.DataLoader(dataset, batch_size=args.batch, drop_last=True,
                           sampler=TR.data_sampler(dataset, shuffle=True, distributed=args.distributed, weights=weights))


def main(phase, argv=None, dataset=None):
    from diagan.models.stylegan2 import Discriminator, Generator
    args = build_parser(phase).parse_args(argv)
    print(args)
    if args.gpu:
        os.environ['HIP_VISIBLE_DEVICES'] = args.gpu
    save_path = Path(f'{args.work_dir}/{args.exp_name}')
    save_path.mkdir(parents=True, exist_ok=True)
    set_seed(args.seed)
    rank, local_rank, world = dist.init_from_env()
    args.distributed = world > 1
    device = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
    args.latent, args.n_mlp, args.start_iter = 512, 8, 0
    if not hasattr(args, 'stop_save_logit_after'):
        args.stop_save_logit_after = -1
    if args.dataset not in SIZES:
        raise AttributeError(f'{args.dataset} not supported')
    args.size = SIZES[args.dataset]

    def make_d():
        return Discriminator(args.size, channel_multiplier=args.channel_multiplier).to(device)

    def make_g():
        return Generator(args.size, args.latent, args.n_mlp, channel_multiplier=args.channel_multiplier).to(device)

    generator, discriminator, g_ema = make_g(), make_d(), make_g()
    g_ema.eval()
    TR.accumulate(g_ema, generator, 0)
    g_optim, d_optim = TR.make_optimizers(generator, discriminator, args.lr, args.g_reg_every, args.d_reg_every)
    dist.seed_device_per_rank(args.seed, enable=args.rank_noise)       # after the (identically seeded) initialisation
    extra = {}
    ckpt_path = args.ckpt
    if phase == 2:
        drs_d = make_d()
        drs_optim = TR.make_optimizers(generator, drs_d, args.lr, args.g_reg_every, args.d_reg_every)[1]
        extra = dict(drs_discriminator=drs_d, drs_d_optim=drs_optim)
        if not ckpt_path:
            ckpt_path = f'{args.work_dir}/{args.baseline_exp_name}/checkpoint/{str(args.p1_step).zfill(6)}.pt'
    if ckpt_path is not None:
        print("load model:", ckpt_path)
        ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
        try:
            args.start_iter = int(os.path.splitext(os.path.basename(ckpt_path))[0]) + 1
        except ValueError:
            pass
        generator.load_state_dict(ckpt["g"])
        discriminator.load_state_dict(ckpt["d"])
        g_ema.load_state_dict(ckpt["g_ema"])
        g_optim.load_state_dict(ckpt["g_optim"])
        d_optim.load_state_dict(ckpt["d_optim"])
        if phase == 2:          # D_drs starts from the phase-1 discriminator and its optimiser state (:604-610)
            extra['drs_discriminator'].load_state_dict(ckpt["d"])
            extra['drs_d_optim'].load_state_dict(ckpt["d_optim"])
        print(f'start_iter: {args.start_iter}')

    if dataset is None:
        dataset = IndexedImages(args.num_data or 4096, args.size)
    weights = None
    if phase == 2:
        from diagan.utils.plot import calculate_scores
        logit_path = f'{args.work_dir}/{args.baseline_exp_name}/logits_netD.pkl'
        print(f'Use logit from: {logit_path}')
        with open(logit_path, "rb") as f:
            logits = pickle.load(f)
        window = 5000                                                               # train_ffhq_phase2.py:649-652
        score_dict = calculate_scores(logits, start_epoch=args.p1_step - window, end_epoch=args.p1_step + 1,
                                      keys=[args.resample_score])
        weights = score_dict[args.resample_score]
        print(f'weight_list max: {weights.max()} min: {weights.min()} mean: {weights.mean()} var: {weights.var()}')
        extra['drs_loader'] = _loader(dataset, args)
    loader = _loader(dataset, args, weights)
    trainer = TR.StyleGAN2Trainer(args, loader, generator, discriminator, g_optim, d_optim, g_ema, device, save_path,
                                  log_every=args.log_every, checkpoint_every=args.checkpoint_every, **extra)
    trainer.train()
    dump = os.environ.get("DIAGAN_SG2_DUMP")        # test hook: final flat parameters of every rank
    if dump:
        final = {"g": generator.flat_params.cpu(), "d": discriminator.flat_params.cpu()}
        if phase == 2:
            final["drs_d"] = extra['drs_discriminator'].flat_params.cpu()
            final["first_indices"] = list(iter(loader.sampler))[:8]
        torch.save(final, os.path.join(dump, f"rank{rank}_phase{phase}_final.pt"))
    return trainer
