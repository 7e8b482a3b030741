"""GPU end-to-end: the StyleGAN2 trainers through their command lines -- phase 1 trains, records
per-index discriminator logits and writes a reference-keyed checkpoint; phase 2 resumes from it, scores the record
(LDR scorer kernel), samples by weight and trains D_drs beside D; two ranks stay in lock-step."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, "stylegan2"))


def _phase1(work, extra=()):
    import train_ffhq
    return train_ffhq.main(["-d", "cifar10", "--batch", "4", "--iter", "7", "--num_data", "24", "--work_dir", str(work),
                            "--exp_name", "base", "--logit_save_steps", "2", "--save_logit_after", "2",
                            "--stop_save_logit_after", "6", "--checkpoint_every", "6", "--log_every", "2",
                            "--d_reg_every", "2", "--g_reg_every", "2", "--r1", "10", *extra])


@pytest.mark.timeout(900)
def test_phase1_then_phase2(tmp_path):
    tr = _phase1(tmp_path)
    assert tr.history and all(np.isfinite(list(h.values())).all() for h in tr.history)
    assert {'d', 'g', 'r1', 'path', 'real_score', 'fake_score', 'path_length'} <= set(tr.history[-1])
    assert tr.history[-1]['r1'] > 0 and tr.history[-1]['path'] > 0 and tr.mean_path_length_avg > 0
    logits = pickle.load(open(tmp_path / "base" / "logits_netD.pkl", "rb"))
    assert sorted(logits) == [2, 4, 6] and all(v.shape == (24,) and np.isfinite(v).all() for v in logits.values())
    assert np.abs(logits[6]).max() > 0 and not np.array_equal(logits[2], logits[6])
    ckpt = torch.load(tmp_path / "base" / "checkpoint" / "000006.pt", map_location="cpu", weights_only=False)
    assert set(ckpt) == {"g", "d", "g_ema", "g_optim", "d_optim", "args", "ada_aug_p"}
    # the EMA generator moved away from its initial copy, towards the trained one
    assert not torch.equal(ckpt["g"]["conv1.conv.weight"], ckpt["g_ema"]["conv1.conv.weight"])
    # optimiser state is torch.optim.Adam's wire format
    opt = torch.optim.Adam([torch.nn.Parameter(v.clone().float()) for k, v in ckpt["d"].items()
                            if not k.endswith("kernel")], lr=1e-3)
    opt.load_state_dict(ckpt["d_optim"])

    import train_ffhq_phase2
    tr2 = train_ffhq_phase2.main(["-d", "cifar10", "--batch", "4", "--iter", "10", "--num_data", "24", "--work_dir",
                                  str(tmp_path), "--exp_name", "p2", "--baseline_exp_name", "base", "--p1_step", "6",
                                  "--resample_score", "ldrm", "--log_every", "1", "--checkpoint_every", "9",
                                  "--d_reg_every", "2", "--g_reg_every", "2"])
    assert tr2.args.start_iter == 7 and [h['step'] for h in tr2.history] == [7, 8, 9, 10]
    assert all('drs_d' in h and np.isfinite(h['drs_d']) for h in tr2.history)
    sampler = tr2.loader.sampler
    assert isinstance(sampler, torch.utils.data.WeightedRandomSampler) and len(sampler.weights) == 24
    ck2 = torch.load(tmp_path / "p2" / "checkpoint" / "000009.pt", map_location="cpu", weights_only=False)
    assert {"drs_d", "drs_d_optim"} <= set(ck2)
    # sampling script on the phase-2 checkpoint's EMA generator, with truncation towards the mean latent
    import generate
    files = generate.main(["--size", "32", "--ckpt", str(tmp_path / "p2" / "checkpoint" / "000009.pt"), "--pics", "2",
                           "--sample", "3", "--truncation", "0.7", "--truncation_mean", "64", "--out", str(tmp_path / "s")])
    imgs = [torch.load(f) for f in files]
    assert len(imgs) == 2 and all(i.shape == (3, 3, 32, 32) and float(i.abs().max()) <= 1 for i in imgs)
    assert not torch.equal(imgs[0], imgs[1])
    # D_drs started from D's phase-1 weights and has since been trained on uniformly sampled data
    assert not torch.equal(ck2["drs_d"]["final_linear.1.weight"], ckpt["d"]["final_linear.1.weight"])
    assert not torch.equal(ck2["drs_d"]["final_linear.1.weight"], ck2["d"]["final_linear.1.weight"])


@pytest.mark.timeout(900)
def test_two_ranks_stay_in_lock_step(tmp_path):
    """two processes on the one GPU, gloo: gradient slabs are averaged once per step, the logit record is gathered
    from both ranks' shards (DistributedSampler) and the replicas end with identical weights"""
    env = dict(os.environ, DIAGAN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "stylegan2", "train_ffhq.py"), "-d", "cifar10",
           "--batch", "4", "--iter", "5", "--num_data", "16", "--work_dir", str(tmp_path), "--exp_name", "dp",
           "--logit_save_steps", "2", "--save_logit_after", "2", "--stop_save_logit_after", "4", "--checkpoint_every",
           "4", "--log_every", "2", "--d_reg_every", "2", "--g_reg_every", "2", "--r1", "10"]
    env["DIAGAN_SG2_DUMP"] = str(tmp_path)
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    logits = pickle.load(open(tmp_path / "dp" / "logits_netD.pkl", "rb"))
    assert sorted(logits) == [2, 4] and all(np.abs(v).min() > 0 for v in logits.values())     # every index was filled
    a, b = (torch.load(tmp_path / f"rank{r}_phase1_final.pt") for r in (0, 1))
    assert torch.equal(a["g"], b["g"]) and torch.equal(a["d"], b["d"])

    # phase 2 on two ranks: resumes from the rank-0 checkpoint, keeps the score weights under data parallelism (every
    # rank takes its stride of ONE weighted draw) and trains D_drs in lock-step too
    cmd2 = cmd[:cmd.index(os.path.join(ROOT, "stylegan2", "train_ffhq.py"))] + [
        os.path.join(ROOT, "stylegan2", "train_ffhq_phase2.py"), "-d", "cifar10", "--batch", "4", "--iter", "7",
        "--num_data", "16", "--work_dir", str(tmp_path), "--exp_name", "dp2", "--baseline_exp_name", "dp", "--p1_step",
        "4", "--resample_score", "ldrv", "--log_every", "1", "--checkpoint_every", "100", "--d_reg_every", "2",
        "--g_reg_every", "2"]
    cmd2[cmd2.index("29533")] = "29534"
    out = subprocess.run(cmd2, env=env, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    a, b = (torch.load(tmp_path / f"rank{r}_phase2_final.pt") for r in (0, 1))
    for k in ("g", "d", "drs_d"):
        assert torch.equal(a[k], b[k]), k
    assert not torch.equal(a["drs_d"], a["d"])
    assert len(a["first_indices"]) > 0 and a["first_indices"] != b["first_indices"]       # disjoint strides of one draw


@pytest.mark.timeout(600)
def test_bench_line_for_the_stylegan2_workload():
    """bench.py --workload stylegan2: one JSON line with the contract's keys (iteration = the trainer's train_step)"""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "stylegan2", "--steps", "2",
                          "--warmup", "1", "--batch_size", "4"], capture_output=True, text=True, timeout=500)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["metric"] == "images/sec (G+D step)" and line["value"] > 0 and line["n_gpus"] == 1
    assert "StyleGAN2" in line["config"]["workload"] and line["config"]["global_batch"] == 4
    assert line["roofline"]["bound"] == "mfma" and 0 < line["roofline"]["frac"] < 1
    assert "cpu_baseline" not in line
    assert line["dtype"] == "f32"
