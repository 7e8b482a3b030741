"""Names of the reference's stylegan2/model.py (`Generator`, `Discriminator`): the HIP-backed classes."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "self-diagnosing-gan_amd"))

from diagan.models.stylegan2 import Discriminator, Generator  # noqa: E402,F401
