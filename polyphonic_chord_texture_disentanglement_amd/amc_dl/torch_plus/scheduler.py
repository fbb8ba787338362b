"""Per-batch schedulers (reference amc_dl/torch_plus/scheduler.py): a step counter that only
advances in 'train' mode, teacher-forcing / KL schedules driven by it, the optimiser + LR pair,
and the dict-of-schedulers the trainer queries once per batch."""
from .train_utils import scheduled_sampling


class _Scheduler:

    def __init__(self, step=0, mode='train'):
        self._step = step
        self._mode = mode

    def _update_step(self):
        if self._mode == 'train':
            self._step += 1
        elif self._mode != 'val':
            raise NotImplementedError

    def step(self):
        raise NotImplementedError

    def train(self):
        self._mode = 'train'

    def eval(self):
        self._mode = 'val'


class ConstantScheduler(_Scheduler):
    """scheduler.py:28-36"""

    def __init__(self, param, step=0.):
        super().__init__(step)
        self.param = param

    def step(self):
        self._update_step()
        return self.param


class TeacherForcingScheduler(_Scheduler):
    """scheduler.py:39-54: value = f(step counter, high, low), read BEFORE the counter advances."""

    def __init__(self, high, low, f=scheduled_sampling, step=0):
        super().__init__(step)
        self.high, self.low = high, low
        self.schedule_f = f

    def get_tfr(self):
        return self.schedule_f(self._step, self.high, self.low)

    def step(self):
        value = self.get_tfr()
        self._update_step()
        return value


class OptimizerScheduler(_Scheduler):
    """scheduler.py:57-74: optimizer.step() then LR scheduler.step(), once per BATCH."""

    def __init__(self, optimizer, scheduler, clip, step=0):
        super().__init__(step)
        self.optimizer, self.scheduler, self.clip = optimizer, scheduler, clip

    def optimizer_zero_grad(self):
        self.optimizer.zero_grad()

    def step(self, require_zero_grad=False):
        self.optimizer.step()
        self.scheduler.step()
        if require_zero_grad:
            self.optimizer_zero_grad()
        self._update_step()


class ParameterScheduler(_Scheduler):
    """scheduler.py:77-99: {name: scheduler} -> {name: value} per batch."""

    def __init__(self, step=0, mode='train', **schedulers):
        super().__init__(step)
        self.schedulers = schedulers
        self.mode = mode

    def train(self):
        self.mode = 'train'
        for s in self.schedulers.values():
            s.train()

    def eval(self):
        self.mode = 'val'
        for s in self.schedulers.values():
            s.eval()

    def step(self, require_zero_grad=False):
        return {k: s.step() for k, s in self.schedulers.items()}
