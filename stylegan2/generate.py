#!/usr/bin/env python
"""Sample from a trained StyleGAN2 generator (flags of the reference's stylegan2/generate.py:33-64) on the HIP engine.

    python stylegan2/generate.py --size 256 --ckpt exp_results/base/checkpoint/200000.pt --pics 20 --truncation 0.7

torchvision is not part of this image, so instead of PNG grids every batch is written as a float tensor in [-1, 1]
(`<out>/<index>.pt`, NCHW), which `torchvision.utils.save_image(..., normalize=True, range=(-1, 1))` turns into the
reference's picture."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "self-diagnosing-gan_amd"))

import torch  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser(description="Generate samples from the generator")
    p.add_argument("--size", type=int, default=1024, help="output image size of the generator")
    p.add_argument("--sample", type=int, default=1, help="number of samples to be generated for each image")
    p.add_argument("--pics", type=int, default=20, help="number of images to be generated")
    p.add_argument("--truncation", type=float, default=1, help="truncation ratio")
    p.add_argument("--truncation_mean", type=int, default=4096, help="number of vectors to calculate mean for the truncation")
    p.add_argument("--ckpt", type=str, default="stylegan2-ffhq-config-f.pt", help="path to the model checkpoint")
    p.add_argument("--channel_multiplier", type=int, default=2, help="config-f = 2, else = 1")
    p.add_argument("--out", type=str, default="sample", help="output directory (not in the reference: it writes ./sample)")
    return p


def generate(args, g_ema, device, mean_latent):
    os.makedirs(args.out, exist_ok=True)
    paths = []
    with torch.no_grad():
        g_ema.eval()
        for i in range(args.pics):
            sample_z = torch.randn(args.sample, args.latent, device=device)
            sample, _ = g_ema([sample_z], truncation=args.truncation, truncation_latent=mean_latent)
            paths.append(os.path.join(args.out, f"{str(i).zfill(6)}.pt"))
            torch.save(sample.clamp(-1, 1).cpu(), paths[-1])
    return paths


def main(argv=None):
    from diagan.models.stylegan2 import Generator
    args = build_parser().parse_args(argv)
    args.latent, args.n_mlp = 512, 8
    device = torch.device("cuda")
    g_ema = Generator(args.size, args.latent, args.n_mlp, channel_multiplier=args.channel_multiplier).to(device)
    checkpoint = torch.load(args.ckpt, map_location="cpu", weights_only=False)
    g_ema.load_state_dict(checkpoint["g_ema"])
    mean_latent = None
    if args.truncation < 1:
        with torch.no_grad():
            mean_latent = g_ema.mean_latent(args.truncation_mean)
    return generate(args, g_ema, device, mean_latent)


if __name__ == "__main__":
    main()
