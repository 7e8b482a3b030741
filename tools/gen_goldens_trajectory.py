#!/usr/bin/env python
"""tests/golden/sngan_trajectory.npz: the CPU legs of tools/sngan_trajectory.py (oracle/nets.py in float64 and in fp32: five global
steps of SNGAN-32, 5 D + 1 G updates each, batch 64, injected batches and noise) computed ONCE -- the float64 parameter trajectory as
count-sketches (16384 buckets per step: distances to ~1 %), how far it moved, its losses, and the CPU fp32 run's losses / distances.
tests/test_sngan_gpu.py::test_five_step_trajectory... then runs only the two HIP builds (the float64 leg was 280 s of the GPU suite).
CPU only, ~5-10 minutes:   python tools/gen_goldens_trajectory.py [steps [output.npz]]
(the committed file was made on the GPU box's host -- EPYC 9575F, 16 threads: the CPU fp32 leg is the yardstick of that box's PyTorch CPU
arithmetic; the float64 leg is machine-independent to ~1e-12)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from sngan_trajectory import sketch, trajectories  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
torch.set_num_threads(min(16, os.cpu_count() or 1))
r = trajectories("cifar10", steps=steps, hip_modes=())
out = dict(dataset="cifar10", steps=steps, n=r["_n"], param_norm=r["param_norm"], moved=np.array(r["moved"]),
           errD64=np.array(r["errD64"]), errG64=np.array(r["errG64"]), sketch64=torch.stack([sketch(p) for p in r["_p64"]]).numpy(),
           cpu_fp32_errD=np.array(r["cpu fp32"]["errD"]), cpu_fp32_errG=np.array(r["cpu fp32"]["errG"]),
           cpu_fp32_dist=np.array(r["cpu fp32"]["dist"]))
# self-check of the sketch: the CPU fp32 run's exact distances against their sketched form
path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden", "sngan_trajectory.npz")
np.savez_compressed(path, **out)
print("wrote", path, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})
print("moved", out["moved"], "cpu fp32 dist", out["cpu_fp32_dist"])
