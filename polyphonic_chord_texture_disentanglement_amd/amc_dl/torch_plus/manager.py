"""Result directories, loader holder and scalar writers (API of the reference's
amc_dl/torch_plus/manager.py).  tensorboardX is optional: when it is not installed the writers append
the scalars to `<writer dir>/<loss name>/scalars.jsonl` instead."""
import datetime
import json
import os
import shutil

import torch

from .train_utils import join_fn

try:
    from tensorboardX import SummaryWriter as _TensorboardWriter
except Exception:                                   # noqa: BLE001  (absent in the build image)
    _TensorboardWriter = None


class LogPathManager:
    """`./<log_path_name>_<date>_<HHMMSS>/` with `writers/` and `models/` inside; `readme_fn` (the training
    script, train.py:19,48) is copied there as readme.txt.  Checkpoint names: `<model>_{epoch,valid,final}.pt`."""

    def __init__(self, readme_fn=None, log_path_name='result', with_date=True, with_time=True,
                 writer_folder='writers', model_folder='models'):
        stamp = [log_path_name,
                 str(datetime.date.today()) if with_date else '',
                 datetime.datetime.now().strftime('%H%M%S') if with_time else '']
        self.log_path = os.path.join('.', '_'.join(stamp))
        self.writer_path = os.path.join(self.log_path, writer_folder)
        self.model_path = os.path.join(self.log_path, model_folder)
        for folder in (self.log_path, self.writer_path, self.model_path):
            self.create_path(folder)
        if readme_fn is not None:
            shutil.copyfile(readme_fn, os.path.join(self.log_path, 'readme.txt'))

    @staticmethod
    def create_path(path):
        os.makedirs(path, exist_ok=True)

    def _checkpoint(self, model_name, kind):
        return os.path.join(self.model_path, join_fn(model_name, kind, ext='pt'))

    def epoch_model_path(self, model_name):
        return self._checkpoint(model_name, 'epoch')

    def valid_model_path(self, model_name):
        return self._checkpoint(model_name, 'valid')

    def final_model_path(self, model_name):
        return self._checkpoint(model_name, 'final')


class DataLoaders:
    """Holds the train / validation loaders, their lengths and batch sizes, and the target device."""

    def __init__(self, train_loader, val_loader, bs_train, bs_val, device=None):
        self.train_loader, self.val_loader = train_loader, val_loader
        self.num_train_batch, self.num_val_batch = len(train_loader), len(val_loader)
        self.bs_train, self.bs_val = bs_train, bs_val
        self.device = device if device is not None else torch.device('cuda' if torch.cuda.is_available() else 'cpu')

    @staticmethod
    def get_loaders(seed, bs_train, bs_val, portion=8, shift_low=-6, shift_high=5, num_bar=2, contain_chord=True):
        raise NotImplementedError

    def batch_to_inputs(self, *input):
        raise NotImplementedError

    @staticmethod
    def _get_ith_batch(i, loader):
        for index, batch in enumerate(loader):
            if index == i:
                return batch
        raise IndexError('loader has no batch %d' % i)

    def get_ith_train_batch(self, i):
        return self._get_ith_batch(i, self.train_loader)

    def get_ith_val_batch(self, i):
        return self._get_ith_batch(i, self.val_loader)


class _JsonlWriter:
    """`add_scalar(tag, value, step)` sink used when tensorboardX is unavailable."""

    def __init__(self, folder):
        os.makedirs(folder, exist_ok=True)
        self.path = os.path.join(folder, 'scalars.jsonl')
        self.scalars = []

    def add_scalar(self, tag, val, step):
        record = {'tag': tag, 'value': float(val), 'step': int(step)}
        self.scalars.append((record['tag'], record['value'], record['step']))
        with open(self.path, 'a') as fh:
            fh.write(json.dumps(record) + '\n')


class SummaryWriters:
    """One writer per loss name (writer_names[0] must be 'loss'); `tags` maps a tag key to the indices of the
    writers that log it (None = all); the written tag is '<task>_<key>' for task in ('train', 'val')."""

    def __init__(self, writer_names, tags, log_path, tasks=('train', 'val')):
        assert writer_names[0] == 'loss'
        self.log_path, self.writer_names = log_path, writer_names
        everything = tuple(range(len(writer_names)))
        self.tags = {key: everything if ids is None else ids for key, ids in tags.items()}
        sink = _TensorboardWriter or _JsonlWriter
        self.writers = {name: sink(os.path.join(log_path, name)) for name in writer_names}
        self.all_tags = {task: {task + '_' + key: ids for key, ids in self.tags.items()} for task in tasks}

    def single_write(self, name, tag, val, step):
        self.writers[name].add_scalar(tag, val, step)

    def write_tag(self, task, tag, vals, step):
        ids = self.all_tags[task][tag]
        assert len(ids) == len(vals)
        for index, val in zip(ids, vals):
            self.single_write(self.writer_names[index], tag, val, step)

    def write_task(self, task, vals_dic, step):
        for tag, ids in self.all_tags[task].items():
            self.write_tag(task, tag, [vals_dic[self.writer_names[index]] for index in ids], step)
