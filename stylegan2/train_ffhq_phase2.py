#!/usr/bin/env python
"""StyleGAN2 phase 2: score-weighted resampling + D_drs (entry point of the reference's
stylegan2/train_ffhq_phase2.py): see diagan/stylegan2_cli.py."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "self-diagnosing-gan_amd"))

from diagan.stylegan2_cli import build_parser as _bp, main as _main  # noqa: E402


def build_parser():
    return _bp(2)


def main(argv=None, dataset=None):
    return _main(2, argv, dataset)


if __name__ == "__main__":
    main()
