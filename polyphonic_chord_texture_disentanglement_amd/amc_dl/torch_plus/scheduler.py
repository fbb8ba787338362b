"""Per-batch schedulers of the trainer (API of the reference's amc_dl/torch_plus/scheduler.py).

All of them share one piece of state: a batch counter that advances only while the scheduler is in
'train' mode (`scheduler.py:10-16`), so validation passes see frozen values.  The teacher-forcing and
KL-weight schedules are functions of that integer counter (`scheduler.py:48-54`), which is why the
published settings saturate after two batches."""
from .train_utils import scheduled_sampling

_MODES = ('train', 'val')


class _Scheduler:
    """Counter + mode.  Subclasses implement `step()` and call `_update_step()` once per batch."""

    def __init__(self, step=0, mode='train'):
        self._step, self._mode = step, mode

    def train(self):
        self._mode = 'train'

    def eval(self):
        self._mode = 'val'

    def _update_step(self):
        if self._mode not in _MODES:
            raise NotImplementedError(self._mode)
        self._step += int(self._mode == 'train')

    def step(self):
        raise NotImplementedError


class ConstantScheduler(_Scheduler):
    """Always the same value (`weights=[1, 0.5]` in train.py:62)."""

    def __init__(self, param, step=0.):
        _Scheduler.__init__(self, step)
        self.param = param

    def step(self):
        self._update_step()
        return self.param


class TeacherForcingScheduler(_Scheduler):
    """value_k = f(k, high, low) with k the number of TRAINING batches seen so far; the value is read
    before the counter moves (scheduler.py:48-54)."""

    def __init__(self, high, low, f=scheduled_sampling, step=0):
        _Scheduler.__init__(self, step)
        self.high, self.low, self.schedule_f = high, low, f

    def get_tfr(self):
        return self.schedule_f(self._step, self.high, self.low)

    def step(self):
        current = self.get_tfr()
        self._update_step()
        return current


class OptimizerScheduler(_Scheduler):
    """Bundles optimizer, LR scheduler and the gradient-clipping threshold; `step()` = optimizer.step()
    followed by one LR-scheduler step (per batch, scheduler.py:69-74)."""

    def __init__(self, optimizer, scheduler, clip, step=0):
        _Scheduler.__init__(self, step)
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
    """Named schedulers -> the keyword arguments of `model('train', ...)` for this batch."""

    def __init__(self, step=0, mode='train', **schedulers):
        _Scheduler.__init__(self, step)
        self.schedulers, self.mode = schedulers, mode

    def _set_mode(self, mode):
        self.mode = mode
        for sch in self.schedulers.values():
            (sch.train if mode == 'train' else sch.eval)()

    def train(self):
        self._set_mode('train')

    def eval(self):
        self._set_mode('val')

    def step(self, require_zero_grad=False):
        return {name: sch.step() for name, sch in self.schedulers.items()}

    # ---- resume support (the reference saves no scheduler state: run(start_*) only resets the trainer's counters,
    # module.py:195-198, so a resumed run restarts teacher forcing and the KL weight from step 0)
    def state_dict(self):
        return {'step': self._step, 'mode': self.mode, 'schedulers': {n: sch._step for n, sch in self.schedulers.items()}}

    def load_state_dict(self, state):
        self._step = state['step']
        for n, k in state['schedulers'].items():
            self.schedulers[n]._step = k
        self._set_mode(state.get('mode', 'train'))
