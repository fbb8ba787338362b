"""`amc_dl.torch_plus` trainer surface for the MI355X path.

Exports the names the reference package exports (its `__init__` re-exports the model base class, the
training loop, the per-batch schedulers, the path / loader / writer helpers and `MinExponentialLR`), so
`from amc_dl.torch_plus import ...` lines of a training script keep working."""
from . import example, manager, module, scheduler, train_utils  # noqa: F401

MinExponentialLR = example.MinExponentialLR
PytorchModel, TrainingInterface = module.PytorchModel, module.TrainingInterface
LogPathManager, DataLoaders, SummaryWriters = manager.LogPathManager, manager.DataLoaders, manager.SummaryWriters
ConstantScheduler = scheduler.ConstantScheduler
TeacherForcingScheduler = scheduler.TeacherForcingScheduler
OptimizerScheduler = scheduler.OptimizerScheduler
ParameterScheduler = scheduler.ParameterScheduler

__all__ = ['PytorchModel', 'TrainingInterface', 'ConstantScheduler', 'TeacherForcingScheduler', 'OptimizerScheduler',
           'ParameterScheduler', 'LogPathManager', 'DataLoaders', 'SummaryWriters', 'MinExponentialLR']
