"""Learning-rate schedule of the reference trainer: exponential decay with a floor.

    lr_k = max(lr_0 * gamma**k, minimum)        (reference amc_dl/torch_plus/example.py:4-13)

stepped once per BATCH by `OptimizerScheduler.step`; with the reference's settings (1e-3, 0.9999, 1e-5)
the floor is reached after ~46k batches."""
from torch.optim.lr_scheduler import ExponentialLR


class MinExponentialLR(ExponentialLR):
    """`ExponentialLR` whose closed form is clamped from below by `minimum` (kept in `self.min`)."""

    def __init__(self, optimizer, gamma, minimum, last_epoch=-1):
        self.min = float(minimum)
        # like the reference, always start the schedule from step 0 whatever `last_epoch` says
        ExponentialLR.__init__(self, optimizer, gamma, last_epoch=-1)

    def get_lr(self):
        k = self.last_epoch
        decayed = (base * self.gamma ** k for base in self.base_lrs)
        return [lr if lr > self.min else self.min for lr in decayed]
