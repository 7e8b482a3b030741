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
        self._step += 1
        E.adam_step(self.net.flat_params, self.net.flat_grads, self._m, self._v, g['lr'], g['betas'][0],
                    g['betas'][1], g['eps'], self._step)
        self.net.param_version += 1

    def zero_grad(self, set_to_none=False):
        self.net.zero_grad()

    # checkpoint: flat moments + step (optimizer_state_dict entry of mimicry's checkpoint dict)
    def state_dict(self):
        self._ensure_state()
        return {
            'fused_adam': True,
            'step': self._step,
            'exp_avg': self._m.detach().cpu(),
            'exp_avg_sq': self._v.detach().cpu(),
            'param_groups': [{k: v for k, v in g.items() if k != 'params'} for g in self.param_groups],
        }

    def load_state_dict(self, sd):
        if not sd.get('fused_adam'):
            raise RuntimeError("optimizer state is not a FusedAdam state (loading torch.optim.Adam state of a "
                               "genuine mimicry checkpoint is listed as 'next' in SURVEY §8(f) rank 3)")
        self._ensure_state()
        if sd['exp_avg'].numel() != self._m.numel():
            raise RuntimeError("optimizer state size mismatch")
        self._m.copy_(sd['exp_avg'])
        self._v.copy_(sd['exp_avg_sq'])
        self._step = int(sd['step'])
        for g, saved in zip(self.param_groups, sd['param_groups']):
            g.update(saved)
