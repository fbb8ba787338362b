"""Trainer surface of the reference's `amc_dl.torch_plus` package (same importable names,
amc_dl/torch_plus/__init__.py:1-5), re-implemented for the one-process-per-GPU MI355X path."""
from .module import PytorchModel, TrainingInterface
from .scheduler import ConstantScheduler, TeacherForcingScheduler, OptimizerScheduler, ParameterScheduler
from .manager import LogPathManager, DataLoaders, SummaryWriters
from .example import MinExponentialLR
