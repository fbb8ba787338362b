"""MinExponentialLR (reference amc_dl/torch_plus/example.py:4-13): lr_k = max(lr0 * gamma^k, minimum)."""
from torch.optim.lr_scheduler import ExponentialLR


class MinExponentialLR(ExponentialLR):

    def __init__(self, optimizer, gamma, minimum, last_epoch=-1):
        self.min = minimum
        super().__init__(optimizer, gamma, last_epoch=-1)

    def get_lr(self):
        return [max(base * self.gamma ** self.last_epoch, self.min) for base in self.base_lrs]
