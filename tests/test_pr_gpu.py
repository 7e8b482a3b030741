"""GPU: precision / recall in feature space against vectors produced by the reference's
own compute_pr.py on CPU (tests/golden/pr.npz, tools/gen_goldens_models.py::gen_pr).

Distances differ from torch's CPU matmul by fp32 summation order only; a sample whose distance sits within
that rounding of a radius may flip its `<`, so the fractions are allowed +-2 samples."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "pr.npz"))


def test_pairwise_distance_and_kth(g):
    from diagan.trainer import compute_pr as pr
    d = pr.compute_pairwise_distance(g["real"][:64], g["fake"][:48], device="cuda")
    assert d.shape == (64, 48) and d.dtype == np.float32
    np.testing.assert_allclose(d, g["dist"], rtol=2e-5, atol=2e-4)
    dxx = pr.compute_pairwise_distance(g["real"][:40], device="cuda")
    assert np.abs(np.diag(dxx)).max() < 1e-3
    np.testing.assert_array_equal(pr.get_kth_value(np.abs(g["real"][:32]), 3, device="cuda"), g["kth"])


def test_radii_precision_recall(g, capsys):
    from diagan.trainer import compute_pr as pr
    k = int(g["nearest_k"])
    np.testing.assert_allclose(pr.compute_nearest_neighbour_distances(g["real"], k, device="cuda"), g["radii"],
                               rtol=2e-5, atol=2e-4)
    out = pr.compute_pr(g["real"], g["fake"], k, device="cuda")
    assert "Num real: 384 Num fake: 320" in capsys.readouterr().out        # the reference prints this line
    assert abs(out["precision"] - float(g["precision"])) <= 2.0 / 320
    assert abs(out["recall"] - float(g["recall"])) <= 2.0 / 384
    part = pr.compute_partial_recall(g["real"][:100], g["fake"], k, device="cuda")
    assert abs(part["recall"] - float(g["partial_recall"])) <= 2.0 / 100


def test_row_blocks_and_size_independent_properties(monkeypatch):
    """Row-blocked evaluation (forced small blocks) equals the single-block one; identical sets give
    precision = recall = 1; far-apart sets give 0."""
    from diagan.trainer import compute_pr as pr
    rng = np.random.default_rng(0)
    a = rng.normal(size=(500, 96)).astype(np.float32)
    b = rng.normal(size=(450, 96)).astype(np.float32)
    full = pr.compute_pr(a, b, 3, device="cuda")
    monkeypatch.setattr(pr, "_ROW_BLOCK", 128)
    blocked = pr.compute_pr(a, b, 3, device="cuda")
    assert blocked == full
    same = pr.compute_pr(a, a.copy(), 3, device="cuda")
    assert same["precision"] == 1.0 and same["recall"] == 1.0
    far = pr.compute_pr(a, b + 100.0, 3, device="cuda")
    assert far["precision"] == 0.0 and far["recall"] == 0.0
    with pytest.raises(RuntimeError):
        pr.compute_pr(a, b, 3, device="cpu")
