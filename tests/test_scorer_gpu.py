"""GPU: the HIP scorer / logit record against the oracle and the reference-generated goldens."""
import os

import numpy as np
import pytest
import torch

from oracle import scorer as osc

pytestmark = pytest.mark.gpu


def _rec(N, T, seed):
    rng = np.random.default_rng(seed)
    mu = rng.normal(0.0, 2.0, size=N)
    sg = rng.uniform(0.05, 1.0, size=N)
    steps = [1000 + 100 * t for t in range(T)]
    return {s: (mu + sg * rng.normal(size=N)).astype(np.float32).astype(np.float64) for s in steps}


def test_golden_main_bit_exact(golden_dir):
    from diagan.utils.plot import calculate_scores
    g = np.load(os.path.join(golden_dir, "scorer_main.npz"))
    logits = {int(s): g["rec32"][i].astype(np.float64) for i, s in enumerate(g["steps"])}
    sd = calculate_scores(logits, int(g["start"]), int(g["end"]))
    keys = [str(k) for k in g["keys"]]
    assert list(sd.keys()) == keys
    for j, k in enumerate(keys):
        assert sd[k].dtype == np.float64
        assert np.array_equal(sd[k], g["values"][j]), k


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_golden_edges_bit_exact(golden_dir, case):
    from diagan.utils.plot import calculate_scores
    g = np.load(os.path.join(golden_dir, "scorer_edges.npz"))
    logits = {int(s): g[f"{case}_rec"][i] for i, s in enumerate(g[f"{case}_steps"])}
    sd = calculate_scores(logits, int(g[f"{case}_start"]), int(g[f"{case}_end"]))
    for j, k in enumerate([str(k) for k in g["keys"]]):
        assert np.array_equal(sd[k], g[f"{case}_values"][j]), (case, k)


def test_golden_edge_n_equals_one(golden_dir):
    """Edge set (d): ONE sample.  NumPy reduces the then-contiguous T axis pairwise, so its own result differs from a
    sequential sum in the last bits (tests/test_oracle_scorer.py::test_n_equals_one_is_pairwise_in_numpy); the HIP
    scorer walks T sequentially like the C oracle: bit-equal to the oracle, 1e-14 relative to the reference's vector."""
    from diagan.utils.plot import calculate_scores
    g = np.load(os.path.join(golden_dir, "scorer_edges.npz"))
    logits = {int(s): g["d_rec"][i] for i, s in enumerate(g["d_steps"])}
    sd = calculate_scores(logits, int(g["d_start"]), int(g["d_end"]))
    ref = osc.calculate_scores_c(logits, int(g["d_start"]), int(g["d_end"]))
    for j, k in enumerate([str(k) for k in g["keys"]]):
        assert np.array_equal(sd[k], ref[k]), k
        np.testing.assert_allclose(sd[k], g["d_values"][j], rtol=1e-14, atol=1e-15, err_msg=k)


# N = 162 770: the CelebA training set (BASELINE configs[3]; inclusive_gan.py:94-95), T = 50 snapshots = 65 MB
@pytest.mark.parametrize("N,T", [(1, 3), (63, 2), (257, 7), (5000, 50), (50000, 50), (162770, 50)])
def test_vs_oracle_bit_exact(N, T):
    from diagan.utils.plot import calculate_scores
    logits = _rec(N, T, seed=N + T)
    sd = calculate_scores(logits, 0, 10 ** 9)
    ref = osc.calculate_scores_c(logits, 0, 10 ** 9)
    for k in ref:
        assert np.array_equal(sd[k], ref[k]), k


def test_sampler_indices_bit_exact(golden_dir):
    """Sample-index assignments of phase 2 are identical to the reference's."""
    from diagan.utils.plot import calculate_scores
    g = np.load(os.path.join(golden_dir, "scorer_main.npz"))
    s = np.load(os.path.join(golden_dir, "sampler.npz"))
    logits = {int(st): g["rec32"][i].astype(np.float64) for i, st in enumerate(g["steps"])}
    sd = calculate_scores(logits, int(g["start"]), int(g["end"]))
    from diagan.datasets.sampler import make_weighted_sampler
    torch.manual_seed(int(s["seed"]))
    sampler = make_weighted_sampler(sd[str(s["key"])])
    assert np.array_equal(np.array(list(iter(sampler))), s["indices"])


def test_too_few_snapshots_raises():
    from diagan.utils.plot import calculate_scores
    with pytest.raises(ValueError):
        calculate_scores({0: np.zeros(4)}, 0, 10)


def test_f32_path_close():
    from diagan.utils.plot import LogitRecord, ldr_scores_device
    logits = _rec(4097, 50, seed=5)
    rec = LogitRecord.from_dict(logits)
    ref = osc.calculate_scores_c(logits, 0, 10 ** 9)
    stats, conf, tv = ldr_scores_device(rec.window(0, 10 ** 9), exact=False)
    for k in ("ldr", "ldrd", "ldrv", "ldrm"):
        np.testing.assert_allclose(stats[k].cpu().numpy(), ref[k], rtol=2e-5, atol=2e-5)
    for j, t in enumerate(tv):
        np.testing.assert_allclose(conf[j].cpu().numpy(), ref[f'ldr_conf_{t:.1f}_ratio_50'], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("T", [2, 3, 5, 9])
def test_f32_path_small_T(T):
    from diagan.utils.plot import LogitRecord, ldr_scores_device
    logits = _rec(100, T, seed=T)
    rec = LogitRecord.from_dict(logits)
    ref = osc.calculate_scores_c(logits, 0, 10 ** 9)
    stats, conf, tv = ldr_scores_device(rec.window(0, 10 ** 9), exact=False)
    for k in ("ldr", "ldrd", "ldrv", "ldrm"):
        np.testing.assert_allclose(stats[k].cpu().numpy(), ref[k], rtol=2e-5, atol=2e-5)


def test_logit_scatter_and_record_roundtrip():
    from diagan.utils.plot import LogitRecord
    N = 1000
    rec = LogitRecord(N, capacity=2)
    rng = np.random.default_rng(0)
    expect = {}
    for step in (100, 200, 300):          # forces one growth
        r = rec.new_snapshot(step)
        perm = rng.permutation(N)
        lg = rng.normal(size=N).astype(np.float32)
        for lo in range(0, N, 64):        # ragged last batch
            sl = slice(lo, min(lo + 64, N))
            rec.scatter(r, torch.from_numpy(perm[sl]), torch.from_numpy(lg[sl]).cuda().view(-1, 1))
        row = np.zeros(N)
        osc.logit_scatter(lg, perm, row)
        expect[step] = row
    rec.check_bounds()
    d = rec.to_dict()
    assert list(d.keys()) == [100, 200, 300]
    for s in expect:
        assert d[s].dtype == np.float64 and np.array_equal(d[s], expect[s])
    rec.scatter(0, torch.tensor([N]), torch.zeros(1).cuda())
    with pytest.raises(IndexError):
        rec.check_bounds()


def test_requested_keys_only_bit_exact(golden_dir):
    """calculate_scores(..., keys=[...]) -- what the phase-2 command lines ask for (train_mimicry_phase2.py:93 reads ONE of the
    103 scores) -- returns the four statistics and exactly the requested confidence scores, bit-identical to the full call."""
    from diagan.utils.plot import calculate_scores
    g = np.load(os.path.join(golden_dir, "scorer_main.npz"))
    logits = {int(s): g["rec32"][i].astype(np.float64) for i, s in enumerate(g["steps"])}
    start, end = int(g["start"]), int(g["end"])
    full = calculate_scores(logits, start, end)
    want = ["ldr_conf_0.3_ratio_50", "ldr_conf_5.0_ratio_50", "ldrm"]
    part = calculate_scores(logits, start, end, keys=want)
    assert sorted(part) == sorted(["ldr", "ldrd", "ldrv", "ldrm", "ldr_conf_0.3_ratio_50", "ldr_conf_5.0_ratio_50"])
    for k in part:
        assert np.array_equal(part[k], full[k]), k
    only_stats = calculate_scores(logits, start, end, keys=["ldr"])
    assert sorted(only_stats) == ["ldr", "ldrd", "ldrm", "ldrv"] and np.array_equal(only_stats["ldr"], full["ldr"])
    with pytest.raises(KeyError):
        calculate_scores(logits, start, end, keys=["ldr_conf_0.35_ratio_50"])
