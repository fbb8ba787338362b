"""Result paths, data-loader holder and scalar writers (reference amc_dl/torch_plus/manager.py).
tensorboardX is optional here: without it the writers keep the scalars in memory / a jsonl file."""
import datetime
import json
import os
import shutil

import torch

from .train_utils import join_fn

try:                                    # not installed in the build image; never required
    from tensorboardX import SummaryWriter as _TbWriter
except Exception:                       # noqa: BLE001
    _TbWriter = None


class LogPathManager:
    """manager.py:12-48: ./result_<date>_<time>/{writers,models}; copies `readme_fn` as readme.txt."""

    def __init__(self, readme_fn=None, log_path_name='result', with_date=True, with_time=True,
                 writer_folder='writers', model_folder='models'):
        date = str(datetime.date.today()) if with_date else ''
        ctime = datetime.datetime.now().time().strftime('%H%M%S') if with_time else ''
        self.log_path = os.path.join('.', '_'.join([log_path_name, date, ctime]))
        self.writer_path = os.path.join(self.log_path, writer_folder)
        self.model_path = os.path.join(self.log_path, model_folder)
        for p in (self.log_path, self.writer_path, self.model_path):
            LogPathManager.create_path(p)
        if readme_fn is not None:
            shutil.copyfile(readme_fn, os.path.join(self.log_path, 'readme.txt'))

    @staticmethod
    def create_path(path):
        os.makedirs(path, exist_ok=True)

    def _model_fn(self, model_name, kind):
        return os.path.join(self.model_path, join_fn(model_name, kind, ext='pt'))

    def epoch_model_path(self, model_name):
        return self._model_fn(model_name, 'epoch')

    def valid_model_path(self, model_name):
        return self._model_fn(model_name, 'valid')

    def final_model_path(self, model_name):
        return self._model_fn(model_name, 'final')


class DataLoaders:
    """manager.py:51-86"""

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
        for ind, batch in enumerate(loader):
            if ind == i:
                return batch
        raise IndexError(i)

    def get_ith_train_batch(self, i):
        return DataLoaders._get_ith_batch(i, self.train_loader)

    def get_ith_val_batch(self, i):
        return DataLoaders._get_ith_batch(i, self.val_loader)


class _ScalarLog:
    """Stand-in for tensorboardX.SummaryWriter.add_scalar: appends to <dir>/scalars.jsonl."""

    def __init__(self, path):
        os.makedirs(path, exist_ok=True)
        self.fn = os.path.join(path, 'scalars.jsonl')
        self.scalars = []

    def add_scalar(self, tag, val, step):
        self.scalars.append((tag, float(val), int(step)))
        with open(self.fn, 'a') as f:
            f.write(json.dumps({'tag': tag, 'value': float(val), 'step': int(step)}) + '\n')


class SummaryWriters:
    """manager.py:89-135: one writer per loss name; tags '<task>_<key>' -> tuple of writer indices."""

    def __init__(self, writer_names, tags, log_path, tasks=('train', 'val')):
        assert writer_names[0] == 'loss'
        self.log_path = log_path
        self.writer_names = writer_names
        self.tags = {k: (tuple(range(len(writer_names))) if v is None else v) for k, v in tags.items()}
        make = _TbWriter if _TbWriter is not None else _ScalarLog
        self.writers = {n: make(os.path.join(log_path, n)) for n in writer_names}
        self.all_tags = {task: {'_'.join([task, k]): v for k, v in self.tags.items()} for task in tasks}

    def single_write(self, name, tag, val, step):
        self.writers[name].add_scalar(tag, val, step)

    def write_tag(self, task, tag, vals, step):
        ids = self.all_tags[task][tag]
        assert len(vals) == len(ids)
        for i, v in zip(ids, vals):
            self.single_write(self.writer_names[i], tag, v, step)

    def write_task(self, task, vals_dic, step):
        for tag, ids in self.all_tags[task].items():
            self.write_tag(task, tag, [vals_dic[self.writer_names[i]] for i in ids], step)
