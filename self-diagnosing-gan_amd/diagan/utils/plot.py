"""LDR scorer front-end (reference: diagan-pkg/diagan/utils/plot.py:220-249).

`calculate_scores` keeps the reference signature and return value (dict of 103 float64 [N]
arrays) but the arithmetic runs in the HIP kernel `diagan_ldr_scores_f64`, which is bit-exact
with the NumPy reference.  `LogitRecord` is the HBM-resident form of the reference's
`logit_results[name]` dict (trainer.py:224,338): snapshots are scattered straight into a device
matrix and never cross PCIe until they are pickled.
"""
import numpy as np
import torch

from diagan import _native as nat

FLOOR = 1e-2     # clip_min lower bound, plot.py:230
RATIO = 50.0     # clip_max_ratio ratio used for every ldr_conf key, plot.py:248


def conf_t_values():
    """The 99 confidence multipliers, exactly as plot.py:247 builds them."""
    return np.arange(0.1, 10.0, 0.1)


def conf_key(t):
    return f'ldr_conf_{t:.1f}_ratio_50'


class LogitRecord:
    """Resident [T_cap, N] logit record (float64 rows = snapshots, columns = dataset index)."""

    def __init__(self, num_data, capacity=64, device='cuda', dtype=torch.float64):
        self.N = int(num_data)
        self.device = torch.device(device)
        self.dtype = dtype
        self.buf = torch.zeros((capacity, self.N), dtype=dtype, device=self.device)
        self.steps = []                       # row r holds snapshot of global step steps[r]
        self._oob = torch.zeros(1, dtype=torch.int32, device=self.device)

    def _grow(self):
        nb = torch.zeros((self.buf.shape[0] * 2, self.N), dtype=self.dtype, device=self.device)
        nb[: self.buf.shape[0]].copy_(self.buf)
        self.buf = nb

    def new_snapshot(self, step):
        """Start the snapshot of `step`; rows start as zeros like np.zeros(N) (trainer.py:144)."""
        if step in self.steps:
            r = self.steps.index(step)
        else:
            if len(self.steps) == self.buf.shape[0]:
                self._grow()
            self.steps.append(step)
            r = len(self.steps) - 1
        self.buf[r].zero_()
        return r

    def scatter(self, row, idx, logit):
        """rec[row, idx] = logit  (trainer.py:154) on the current stream."""
        logit = logit.reshape(-1)
        if logit.dtype != torch.float32:
            logit = logit.float()
        logit = logit.contiguous()
        idx = idx.to(device=self.device, dtype=torch.int64).contiguous()
        if idx.numel() != logit.numel():
            raise RuntimeError(f"scatter: {idx.numel()} indices for {logit.numel()} logits")
        nat.call("diagan_logit_scatter", nat.ptr(logit), nat.ptr(idx), idx.numel(),
                 self.buf[row].data_ptr(), self.N, 1 if self.dtype == torch.float64 else 0,
                 nat.ptr(self._oob), nat.current_stream())

    def check_bounds(self):
        n = int(self._oob.item())
        if n:
            raise IndexError(f"{n} dataset indices were outside [0, {self.N})")

    def to_dict(self):
        """Host dict{step -> float64 ndarray[N]}: the pickle layout of trainer.py:138-140."""
        host = self.buf[: len(self.steps)].to(torch.float64).cpu().numpy()
        return {s: host[r].copy() for r, s in enumerate(self.steps)}

    @classmethod
    def from_dict(cls, logits, device='cuda'):
        steps = list(logits.keys())
        n = len(np.asarray(logits[steps[0]]))
        rec = cls(n, capacity=max(len(steps), 1), device=device)
        host = np.ascontiguousarray(np.stack([np.asarray(logits[s], dtype=np.float64) for s in steps]))
        rec.buf[: len(steps)].copy_(torch.from_numpy(host))
        rec.steps = steps
        return rec

    def window(self, start_epoch, end_epoch):
        """Rows with start <= step < end in insertion order (dict order, plot.py:239)."""
        rows = [r for r, s in enumerate(self.steps) if s >= start_epoch and s < end_epoch]
        if not rows:
            return self.buf[:0]
        if rows == list(range(rows[0], rows[0] + len(rows))):
            return self.buf[rows[0]: rows[0] + len(rows)]       # contiguous view, no copy
        return self.buf[torch.tensor(rows, device=self.device)]


def ldr_scores_device(rec_rows, t_values=None, want_stats=True, exact=True):
    """Run the scorer on a device [T, N] window. Returns (stats dict, conf [n_t, N]) on device."""
    T, N = rec_rows.shape
    if T < 2:
        raise ValueError(f"calculate_scores needs at least 2 snapshots in the window, got {T}")
    if rec_rows.stride(1) != 1:
        rec_rows = rec_rows.contiguous()
    dev = rec_rows.device
    if exact:
        if rec_rows.dtype != torch.float64:
            rec_rows = rec_rows.double()
        dt, fn_name, ws_elt = torch.float64, "diagan_ldr_scores_f64", 8
    else:
        if rec_rows.dtype != torch.float32:
            rec_rows = rec_rows.float()
        dt, fn_name, ws_elt = torch.float32, "diagan_ldr_scores_f32", 4
    tv = conf_t_values() if t_values is None else np.asarray(t_values, dtype=np.float64)
    n_t = len(tv)
    stats = torch.empty((4, N), dtype=dt, device=dev) if want_stats else None
    conf = torch.empty((n_t, N), dtype=dt, device=dev) if n_t else None
    tdev = torch.from_numpy(tv).to(device=dev, dtype=dt) if n_t else None
    ws = torch.empty(max(n_t, 1) * ws_elt, dtype=torch.uint8, device=dev)
    sp = [stats[k].data_ptr() if want_stats else None for k in range(4)]
    nat.call(fn_name, nat.ptr(rec_rows), T, N, rec_rows.stride(0), sp[0], sp[1], sp[2], sp[3],
             nat.ptr(tdev), n_t, nat.ptr(conf), FLOOR, RATIO, nat.ptr(ws), nat.current_stream())
    out = {}
    if want_stats:
        out = {'ldr': stats[0], 'ldrd': stats[1], 'ldrv': stats[2], 'ldrm': stats[3]}
    return out, conf, tv


def calculate_scores(logits, start_epoch=50, end_epoch=75, clip_val=1.5, conf=1, device='cuda',
                     exact=True, keys=None):
    """Drop-in for plot.py:220 -- same keys, float64 ndarray values, window [start, end).

    `logits` is the reference's dict{step -> ndarray[N]} or a resident LogitRecord.
    keys (not in the reference): the scores the caller will read -- the phase-2 command lines consume ONE of the 103
    (train_mimicry_phase2.py:93 `scores[args.resample_score]`); then only the four statistics and the requested
    `ldr_conf_<t>_ratio_50` rows are computed, copied to the host and returned (same values, bit for bit)."""
    rec = logits if isinstance(logits, LogitRecord) else LogitRecord.from_dict(logits, device=device)
    rows = rec.window(start_epoch, end_epoch)
    print(f'calculate_scores -- start_epoch: {start_epoch} end_epoch: {end_epoch} '
          f'logits_arr: {tuple(rows.shape)}')
    t_values = None
    if keys is not None:
        by_key = {conf_key(t): t for t in conf_t_values()}
        unknown = [k for k in keys if k not in by_key and k not in ('ldr', 'ldrd', 'ldrv', 'ldrm')]
        if unknown:
            raise KeyError(f"calculate_scores: unknown score key(s) {unknown}")
        t_values = [by_key[k] for k in keys if k in by_key]
    stats, conf_dev, tv = ldr_scores_device(rows, t_values=t_values, exact=exact)
    score_dict = dict()
    host_stats = torch.stack([stats[k] for k in ('ldr', 'ldrd', 'ldrv', 'ldrm')]).double().cpu().numpy()
    for j, k in enumerate(('ldr', 'ldrd', 'ldrv', 'ldrm')):
        score_dict[k] = host_stats[j]
    if conf_dev is not None:
        host_conf = conf_dev.double().cpu().numpy()
        for j, t in enumerate(tv):
            score_dict[conf_key(t)] = host_conf[j]
    return score_dict


def print_num_params(netG, netD):
    """plot.py helper used by the CLIs (plot.py:107-110; train_mimicry_phase1.py:74)."""
    count = lambda n: n.count_params() if hasattr(n, 'count_params') else sum(p.numel() for p in n.parameters())
    gen_trainable_parameters, disc_trainable_parameters = count(netG), count(netD)
    print(f'gen_trainable_parameters: {gen_trainable_parameters}, '
          f'disc_trainable_parameters: {disc_trainable_parameters}')
