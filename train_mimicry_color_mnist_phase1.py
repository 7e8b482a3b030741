#!/usr/bin/env python
"""Colored-MNIST / mnist_dcgan phase 1 on the MI355X engine (same flags as the reference's script of this name): see
diagan/cli.py."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "self-diagnosing-gan_amd"))

from diagan.cli import color_mnist_phase1 as main, color_mnist_phase1_parser as build_parser  # noqa: E402,F401

if __name__ == '__main__':
    main()
