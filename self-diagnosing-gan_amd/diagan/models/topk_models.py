"""Top-k generator training (reference: diagan-pkg/diagan/models/topk_models.py:15-194).

`TopKGenerator` keeps the reference's attributes and decay rule; the top-k selection itself is
fused into the generator loss kernel (diagan_loss_gen selects the k largest logits and only those
receive gradient), so train_step is the base class' (the reference's override at :46-116 is the
base step with one get_topk line added)."""
import torch

from diagan.models import sngan


class TopKGenerator:
    def __init__(self, use_topk=False, decay_steps=2000):
        self.use_topk = use_topk
        self.topk_rate = 1
        self.decay_rate = 0.99
        self.decay_steps = 2000    # unused in the reference as well (:21)
        self.min_topk_rate = 0.5

    def decay_topk_rate(self, step, epoch_steps=None):
        assert self.use_topk
        epoch = step // (epoch_steps if epoch_steps else self.decay_steps)
        self.topk_rate = max(self.decay_rate ** epoch, self.min_topk_rate)

    def get_topk(self, x, return_index=False):
        """Tensor-level helper kept for API parity (topk_models.py:31-38)."""
        k = int(self.topk_rate * x.size(0))
        vals, idx = torch.topk(x, k=k, dim=0)
        return (vals, idx) if return_index else vals


class TopkSNGANGenerator32(sngan.SNGANGenerator32, TopKGenerator):
    def __init__(self, topk=False, **kwargs):
        sngan.SNGANGenerator32.__init__(self, **kwargs)
        TopKGenerator.__init__(self, use_topk=topk)
        print(f"Load SNGAN32 model topk: {topk} loss: {self.loss_type}")


class TopkSNGANGenerator64(sngan.SNGANGenerator64, TopKGenerator):
    def __init__(self, topk=False, **kwargs):
        sngan.SNGANGenerator64.__init__(self, **kwargs)
        TopKGenerator.__init__(self, use_topk=topk)
        print(f"Load SNGAN64 model topk: {topk} loss: {self.loss_type}")
