"""Fused Adam over a network's flat parameter slab.

`get_gan_model` returns this in place of torch.optim.Adam
(diagan-pkg/diagan/models/predefined_models.py:32,51,70,89,114,123): same hyper-parameters and
update rule, same `param_groups[0]['lr']` handle the DRS_LRScheduler writes to
(trainer/scheduler.py:75), one HIP launch per step instead of ~10 ATen kernels per parameter.
"""
import torch

from diagan.ops import eltwise as E


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, net, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.net = net
        params = list(net.parameters())
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._m = None
        self._v = None
        self._step = 0
        self._hyper_dev = None        # hipGraph replay: one device row of hyper-parameters per captured update
        self._capturing = False
        self._cursor = 0
        self._updates_per_replay = 0
        # data parallelism: gradients arrive as the SUM over W ranks (distributed.all_reduce_sum_); the 1/W of the mean
        # is applied as the kernel reads the gradient.  `pending` is an optional async collective to wait for first.
        self.grad_scale = 1.0
        self.pending = None

    def _ensure_state(self):
        flat = self.net.flat_params
        if self._m is None or self._m.device != flat.device or self._m.numel() != flat.numel():
            m_old, v_old = self._m, self._v
            self._m = torch.zeros_like(flat)
            self._v = torch.zeros_like(flat)
            if m_old is not None and m_old.numel() == flat.numel():
                self._m.copy_(m_old)
                self._v.copy_(v_old)

    @torch.no_grad()
    def step(self, closure=None):
        self._ensure_state()
        g = self.param_groups[0]
        if self.pending is not None:         # the gradient exchange(s) issued asynchronously by FlatNet.sync_grads
            for w in (self.pending if isinstance(self.pending, (list, tuple)) else [self.pending]):
                w.wait()
            self.pending = None
        if self._capturing:
            # stream capture: the launch reads its hyper-parameters from device row `cursor`, which
            # `before_replay` fills with the values of the update it will stand for
            if self._cursor >= self._hyper_dev.shape[0]:
                raise RuntimeError("FusedAdam: more optimiser steps in one captured region than rows were reserved")
            E.adam_step_dev(self.net.flat_params, self.net.flat_grads, self._m, self._v, self._hyper_dev[self._cursor])
            self._cursor += 1
        else:
            self._step += 1
            E.adam_step(self.net.flat_params, self.net.flat_grads, self._m, self._v, g['lr'], g['betas'][0],
                        g['betas'][1], g['eps'], self._step, grad_scale=self.grad_scale)
        self.net.param_version += 1

    # ---- hipGraph support (diagan/utils/graph.py) -------------------------------------------------------------
    def capture_begin(self, max_updates=64):
        self._ensure_state()
        self._hyper_dev = torch.zeros((max_updates, 8), dtype=torch.float32, device=self.net.flat_params.device)
        self._capturing, self._cursor = True, 0

    def capture_end(self):
        self._capturing = False
        self._updates_per_replay = self._cursor

    def before_replay(self):
        """write the hyper-parameter rows of the next `updates_per_replay` updates (current lr; bias corrections
        of steps step+1 ..) and advance the step counter -- the host side of what the replayed launches will do"""
        g = self.param_groups[0]
        k = self._updates_per_replay
        if k == 0:
            return
        rows = [E.adam_hyper_row(g['lr'], g['betas'][0], g['betas'][1], g['eps'], self._step + 1 + i, self.grad_scale)
                for i in range(k)]
        self._hyper_dev[:k].copy_(torch.tensor(rows, dtype=torch.float32), non_blocking=False)
        self._step += k

    def zero_grad(self, set_to_none=False):
        self.net.zero_grad()

    # ---- checkpoint boundary: the wire format is torch.optim.Adam's ---------------------------------------
    # mimicry checkpoints hold `optimizer.state_dict()` of a torch.optim.Adam over `net.parameters()`
    # (diagan-pkg/diagan/trainer/trainer.py:158-204 restores it): per-parameter `exp_avg` / `exp_avg_sq` in the
    # parameter's own (reference) shape, keyed by position.  The flat moment slabs are converted with the same
    # state-dict hooks that convert the parameters themselves (OIHW <-> packed Wp[Co][Kp], latent-linear row order,
    # channel padding), so a FusedAdam resumes from a genuine Adam state and vice versa (SURVEY §8(f) rank 3;
    # verified against the oracle's torch.optim.Adam, not against a real mimicry install).
    def _views(self, flat):
        base = self.net.flat_params
        out = {}
        for name, p in self.net.named_parameters():
            off = (p.data_ptr() - base.data_ptr()) // base.element_size()
            out[name] = flat[off: off + p.numel()].view(p.shape)
        return out

    def _layout_hooks(self):
        for prefix, mod in self.net.named_modules():
            if hasattr(mod, '_sd_hook') and hasattr(mod, '_load_hook'):
                yield (prefix + '.' if prefix else ''), mod

    def _export(self, flat):
        """flat moment slab -> {parameter name: tensor in the reference's shape} (CPU)."""
        sd = {name: v.detach().clone() for name, v in self._views(flat).items()}
        for prefix, mod in self._layout_hooks():
            mod._sd_hook(mod, sd, prefix, None)
        return {k: v.cpu() for k, v in sd.items()}

    def _import(self, by_name, flat):
        sd = {k: v.to(device=flat.device, dtype=flat.dtype) for k, v in by_name.items()}
        for prefix, mod in self._layout_hooks():
            mod._load_hook(sd, prefix)
        flat.zero_()
        for name, view in self._views(flat).items():
            view.copy_(sd[name].view(view.shape))

    def state_dict(self):
        self._ensure_state()
        names = [n for n, _ in self.net.named_parameters()]
        state = {}
        if self._step > 0:
            m, v = self._export(self._m), self._export(self._v)
            state = {i: {'step': torch.tensor(float(self._step)), 'exp_avg': m[n], 'exp_avg_sq': v[n]}
                     for i, n in enumerate(names)}
        group = {k: v for k, v in self.param_groups[0].items() if k != 'params'}
        for key, default in (('weight_decay', 0), ('amsgrad', False), ('maximize', False), ('foreach', None),
                             ('capturable', False), ('differentiable', False), ('fused', None)):
            group.setdefault(key, default)
        group['params'] = list(range(len(names)))
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        self._ensure_state()
        if sd.get('fused_adam'):                     # flat format written by earlier builds of this engine
            if sd['exp_avg'].numel() != self._m.numel():
                raise RuntimeError("optimizer state size mismatch")
            self._m.copy_(sd['exp_avg'])
            self._v.copy_(sd['exp_avg_sq'])
            self._step = int(sd['step'])
        else:
            names = [n for n, _ in self.net.named_parameters()]
            order = sd['param_groups'][0]['params']
            if len(order) != len(names):
                raise RuntimeError(f"optimizer state has {len(order)} parameters, the network has {len(names)}")
            state = sd['state']
            if not state:
                self._m.zero_(), self._v.zero_()
                self._step = 0
            else:
                missing = [names[i] for i, key in enumerate(order) if key not in state]
                if missing:
                    raise RuntimeError(f"optimizer state lacks entries for {missing[:3]}...")
                self._import({names[i]: state[key]['exp_avg'] for i, key in enumerate(order)}, self._m)
                self._import({names[i]: state[key]['exp_avg_sq'] for i, key in enumerate(order)}, self._v)
                steps = {int(float(state[key]['step'])) for key in order}
                if len(steps) != 1:
                    raise RuntimeError(f"per-parameter step counts differ: {sorted(steps)}")
                self._step = steps.pop()
        saved = {k: v for k, v in sd['param_groups'][0].items() if k in ('lr', 'betas', 'eps')}
        if 'betas' in saved:
            saved['betas'] = tuple(saved['betas'])
        self.param_groups[0].update(saved)
