"""GOLD re-weighted discriminators (reference: diagan-pkg/diagan/models/gold_reweight_models.py:63-86).

The re-weighted losses themselves (:10-61) are the `gold` flag of the fused loss kernel
(csrc/train_ops.hip, diagan_loss_dis): the fake term of each sample is multiplied by the detached
logit before the mean, for 'ns' and 'hinge'."""
from diagan.models import sngan


class GoldDiscriminator:
    """Mixin: always trains with the GOLD loss (the reference's compute_gan_loss at :68-74)."""

    def _init_gold(self, loss_type):
        assert loss_type in ["hinge", "ns"]
        self.use_gold = True


class GoldSNGANDiscriminator32(sngan.SNGANDiscriminator32, GoldDiscriminator):
    def __init__(self, loss_type='ns', **kwargs):
        print("Load SNGAN32 GOLD model")
        sngan.SNGANDiscriminator32.__init__(self, loss_type=loss_type, **kwargs)
        self._init_gold(loss_type)


class GoldSNGANDiscriminator64(sngan.SNGANDiscriminator64, GoldDiscriminator):
    def __init__(self, loss_type='ns', **kwargs):
        print("Load SNGAN64 GOLD model")
        sngan.SNGANDiscriminator64.__init__(self, loss_type=loss_type, **kwargs)
        self._init_gold(loss_type)
