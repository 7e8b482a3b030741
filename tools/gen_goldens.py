#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own code.

Runs only in the build container (needs /root/reference).  Nothing of the reference's source is
copied: the outputs are seeded inputs + the arrays the reference functions return.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tools/gen_goldens.py

Vectors
  scorer_main.npz    calculate_scores (diagan-pkg/diagan/utils/plot.py:220-249), T=50, N=1024
  scorer_edges.npz   same: T=2, constant / all-negative columns, ragged N=7, window edges, N=1
  sampler.npz        WeightedRandomSampler indices (train_mimicry_phase2.py:21-25) for scorer_main weights
  scheduler.npz      DRS_LRScheduler.step (diagan-pkg/diagan/trainer/scheduler.py:80-106)
  dcgan.npz / losses.npz   see gen_dcgan() / gen_losses()  (stub-imported torch_mimicry)
"""
import os
import sys
import types

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
REF = "/root/reference/diagan-pkg"
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")

import numpy as np
import torch


def synth_record(N, steps, seed):
    """Synthetic logit record shaped like SURVEY §8(d): logit[t,i] = fp32(N(mu_i, sigma_i))."""
    rng = np.random.default_rng(seed)
    mu = rng.normal(0.0, 2.0, size=N)
    sg = rng.uniform(0.05, 1.0, size=N)
    return {s: (mu + sg * rng.normal(size=N)).astype(np.float32).astype(np.float64) for s in steps}


def pack_scores(sd):
    keys = list(sd.keys())
    return keys, np.stack([np.asarray(sd[k], dtype=np.float64) for k in keys])


def gen_scorer():
    from diagan.utils.plot import calculate_scores
    steps = list(range(35000, 40001, 100))          # 51 snapshots, bounds inclusive (trainer.py:328)
    logits = synth_record(1024, steps, seed=0)
    sd = calculate_scores(logits, start_epoch=35000, end_epoch=40000)
    keys, vals = pack_scores(sd)
    rec32 = np.stack([logits[s] for s in steps]).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "scorer_main.npz"), steps=np.array(steps), rec32=rec32,
                        start=35000, end=40000, keys=np.array(keys), values=vals)

    # sampler golden: weights from the main case (ldr_conf_0.3_ratio_50), seed 1234
    w = sd['ldr_conf_0.3_ratio_50']
    weight_list = [1e-6 if i < 1e-6 else i for i in w]          # train_mimicry_phase2.py:23
    from torch.utils import data
    torch.manual_seed(1234)
    sampler = data.WeightedRandomSampler(weight_list, len(weight_list), replacement=True)
    idx = np.array(list(iter(sampler)), dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "sampler.npz"), seed=1234, key='ldr_conf_0.3_ratio_50', indices=idx)

    # edge cases
    edge = {}
    # (a) T == 2
    lg = synth_record(33, [100, 200, 300], seed=1)
    k, v = pack_scores(calculate_scores(lg, 100, 300))
    edge.update(a_steps=np.array([100, 200, 300]), a_rec=np.stack([lg[s] for s in (100, 200, 300)]),
                a_start=100, a_end=300, a_values=v)
    # (b) constant column (std 0), all-negative column (hits the 1e-2 floor), ragged N=7
    steps_b = list(range(0, 1000, 100))
    lg = synth_record(7, steps_b, seed=2)
    for s in steps_b:
        lg[s][2] = 0.75
        lg[s][5] = -abs(lg[s][5]) - 3.0
    k, v = pack_scores(calculate_scores(lg, 0, 1000))
    edge.update(b_steps=np.array(steps_b), b_rec=np.stack([lg[s] for s in steps_b]), b_start=0, b_end=1000,
                b_values=v)
    # (c) window edges: keys exactly at start (included) and end (excluded); unsorted insertion order
    steps_c = [500, 300, 400, 700, 600, 800]
    lg = synth_record(65, steps_c, seed=3)
    k, v = pack_scores(calculate_scores(lg, 400, 800))
    edge.update(c_steps=np.array(steps_c), c_rec=np.stack([lg[s] for s in steps_c]), c_start=400, c_end=800,
                c_values=v)
    # (d) N == 1 (NumPy switches to pairwise summation along the reduced axis)
    steps_d = list(range(0, 5000, 100))
    lg = synth_record(1, steps_d, seed=4)
    k, v = pack_scores(calculate_scores(lg, 0, 5000))
    edge.update(d_steps=np.array(steps_d), d_rec=np.stack([lg[s] for s in steps_d]), d_start=0, d_end=5000,
                d_values=v)
    edge['keys'] = np.array(k)
    np.savez_compressed(os.path.join(OUT, "scorer_edges.npz"), **edge)


class _Opt:
    def __init__(self, lr):
        self.param_groups = [{'lr': lr}]


class _Log:
    def __init__(self):
        self.m = {}

    def add_metric(self, name, value, group=None, precision=4):
        self.m[name] = value


def gen_scheduler():
    from diagan.trainer.scheduler import DRS_LRScheduler
    out = {}
    for tag, decay in (("linear", "linear"), ("none", None)):
        opts = [_Opt(2e-4), _Opt(1e-4), _Opt(2e-4)]
        sch = DRS_LRScheduler(lr_decay=decay, optimizers=opts, num_steps=50000)
        steps = [0, 1, 2, 12345, 40000, 49999, 50000, 50001]
        lrs = []
        for s in steps:
            log = sch.step(_Log(), s)
            lrs.append([o.param_groups[0]['lr'] for o in opts] + [log.m['lr_0'], log.m['lr_1'], log.m['lr_2']])
        out[f"{tag}_steps"] = np.array(steps)
        out[f"{tag}_lrs"] = np.array(lrs, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "scheduler.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    gen_scorer()
    gen_scheduler()
    import runpy
    for extra in ("gen_goldens_models.py", "gen_goldens_stylegan_ops.py"):
        if os.path.exists(os.path.join(HERE, extra)):
            runpy.run_path(os.path.join(HERE, extra), run_name="__main__")
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
