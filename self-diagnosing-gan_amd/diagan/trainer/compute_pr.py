"""Precision / recall of two feature sets on the HIP engine (reference: diagan-pkg/diagan/trainer/compute_pr.py,
modified there from clovaai/generative-evaluation-prdc).  Same function names, arguments and return types.

The pairwise squared distances |x|^2 - 2 x.y + |y|^2 come from the conv GEMM kernel (a 1x1 geometry on the fp32
matrix cores); `compute_pr` / `compute_partial_recall` never bring the N x N matrix to the host: k-th neighbour
radii and the two `.any()` reductions run on the device over row blocks of the matrix (csrc/pr_metrics.hip).
"""
import numpy as np
import torch

from diagan import _native as nat
from diagan.ops import conv as C

__all__ = ['compute_pr']

P, I = nat.c_void_p, nat.c_int
nat.register("diagan_row_sqnorm", [P, P, I, I, I, P])
nat.register("diagan_kth_smallest_rows", [P, P, I, I, I, I, P, P])
nat.register("diagan_any_lt_rows", [P, P, P, P, I, I, I, P, P])
nat.register("diagan_any_lt_cols", [P, P, P, P, I, I, I, P, P])

_ROW_BLOCK = 8192


def _dev(device):
    if device is None or str(device) == 'cpu':
        raise RuntimeError("compute_pr: the HIP engine needs a GPU device (no CPU fallback)")
    return torch.device(device)


class _Features:
    """Feature matrix on the device in the two layouts the GEMM wants: rows as 'pixels' [1,1,N,Dp] and rows as
    packed 'weights' [N][Kp] (zero padded), plus the squared row norms."""

    def __init__(self, data, device):
        x = torch.as_tensor(np.ascontiguousarray(data, dtype=np.float32)).to(device)
        if x.dim() != 2:
            raise RuntimeError(f"features must be [N, feature_dim], got {tuple(x.shape)}")
        self.N, self.D = x.shape
        self.Kp = C.round_up(self.D, 32)
        self.packed = torch.zeros((self.N, self.Kp), dtype=torch.float32, device=device)
        self.packed[:, :self.D] = x
        self.norm = torch.empty(self.N, dtype=torch.float32, device=device)
        nat.call("diagan_row_sqnorm", nat.ptr(self.packed), nat.ptr(self.norm), self.N, self.D, self.Kp,
                 nat.current_stream())

    def rows(self, lo, hi):
        return self.packed[lo:hi].view(1, 1, hi - lo, self.Kp)


def _row_blocks(a, b):
    """Yield (lo, hi, T) with T[r][c] = |b_c|^2 - 2 a_r.b_c for the rows lo..hi of a (fp32, on the device)."""
    if a.Kp != b.Kp:
        raise RuntimeError(f"feature dimensions differ: {a.D} vs {b.D}")
    geom = C.Geom("conv", a.Kp, b.N, 1, 1, 1, 0)
    step = max(1, min(_ROW_BLOCK, (1 << 29) // max(b.N, 1) - 1))      # every tensor stays below 2 GiB
    for lo in range(0, a.N, step):
        hi = min(lo + step, a.N)
        T = C.conv_fwd(geom, a.rows(lo, hi), b.packed, bias=b.norm, out_scale=-2.0)
        yield lo, hi, T.view(hi - lo, b.N)


def compute_pairwise_distance(data_x, data_y=None, device=None):
    """numpy [N, feature_dim] (x2) -> numpy [Nx, Ny] of squared distances (reference compute_pr.py:11-31)."""
    device = _dev(device)
    a = _Features(data_x, device)
    b = a if data_y is None else _Features(data_y, device)
    out = np.empty((a.N, b.N), dtype=np.float32)
    for lo, hi, T in _row_blocks(a, b):
        out[lo:hi] = (T + a.norm[lo:hi, None]).cpu().numpy()
    return out


def get_kth_value(unsorted, k, axis=-1, device=None):
    """k-th smallest value along the last axis (reference compute_pr.py:34-50)."""
    device = _dev(device)
    u = torch.as_tensor(np.ascontiguousarray(unsorted, dtype=np.float32)).to(device)
    if axis not in (-1, u.dim() - 1):
        raise NotImplementedError("get_kth_value: only the last axis (the reference's only use)")
    flat = u.reshape(-1, u.shape[-1])
    out = torch.empty(flat.shape[0], dtype=torch.float32, device=device)
    nat.call("diagan_kth_smallest_rows", nat.ptr(flat), None, flat.shape[0], flat.shape[1], flat.shape[1], int(k),
             nat.ptr(out), nat.current_stream())
    return out.reshape(u.shape[:-1]).cpu().numpy()


def _radii(f, nearest_k):
    """Distance of every sample to its nearest_k-th neighbour within its own set (device tensor [N])."""
    r = torch.empty(f.N, dtype=torch.float32, device=f.norm.device)
    for lo, hi, T in _row_blocks(f, f):
        nat.call("diagan_kth_smallest_rows", nat.ptr(T), nat.ptr(f.norm[lo:hi]), hi - lo, f.N, f.N, nearest_k + 1,
                 nat.ptr(r[lo:hi]), nat.current_stream())
    return r


def compute_nearest_neighbour_distances(input_features, nearest_k, device=None):
    """Distances to the k-th nearest neighbours (reference compute_pr.py:53-64)."""
    device = _dev(device)
    return _radii(_Features(input_features, device), nearest_k).cpu().numpy()


def _recall(real, fake, fake_radii):
    """mean_i any_j D[i, j] < fake_radii[j]   (reference compute_pr.py:90-93)."""
    hit = torch.empty(real.N, dtype=torch.float32, device=real.norm.device)
    for lo, hi, T in _row_blocks(real, fake):
        nat.call("diagan_any_lt_rows", nat.ptr(T), nat.ptr(real.norm[lo:hi]), nat.ptr(fake_radii), None, hi - lo, fake.N,
                 fake.N, nat.ptr(hit[lo:hi]), nat.current_stream())
    return hit.double().mean().item()


def compute_pr(real_features, fake_features, nearest_k, device=None):
    """Precision and recall of two manifolds (reference compute_pr.py:67-95)."""
    device = _dev(device)
    print('Num real: {} Num fake: {}'.format(real_features.shape[0], fake_features.shape[0]))
    real, fake = _Features(real_features, device), _Features(fake_features, device)
    real_radii, fake_radii = _radii(real, nearest_k), _radii(fake, nearest_k)
    # precision: mean_j any_i D[i, j] < real_radii[i]  -- a column reduction, OR-ed over the row blocks
    prec = torch.zeros(fake.N, dtype=torch.float32, device=device)
    part = torch.empty_like(prec)
    for lo, hi, T in _row_blocks(real, fake):
        nat.call("diagan_any_lt_cols", nat.ptr(T), nat.ptr(real.norm[lo:hi]), None, nat.ptr(real_radii[lo:hi]), hi - lo,
                 fake.N, fake.N, nat.ptr(part), nat.current_stream())
        prec = torch.maximum(prec, part)
    precision = prec.double().mean().item()
    recall = _recall(real, fake, fake_radii)
    return dict(precision=precision, recall=recall)


def compute_partial_recall(partial_real_features, fake_features, nearest_k, device=None):
    """Recall of a subset of the real features (reference compute_pr.py:100-124)."""
    device = _dev(device)
    print('Num real: {} Num fake: {}'.format(partial_real_features.shape[0], fake_features.shape[0]))
    real, fake = _Features(partial_real_features, device), _Features(fake_features, device)
    return dict(recall=_recall(real, fake, _radii(fake, nearest_k)))
