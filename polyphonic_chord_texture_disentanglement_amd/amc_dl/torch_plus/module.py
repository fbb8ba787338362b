"""PytorchModel base class and the TrainingInterface step loop (reference
amc_dl/torch_plus/module.py).  Same public surface; differences are confined to what one process
per GPU needs: `parallel=True` means torch.distributed data parallelism with an RCCL gradient
all-reduce (instead of nn.DataParallel, module.py:67-68), the 11 logged scalars leave the device
in ONE transfer per step (instead of 22 .item() syncs, module.py:113-124), and the fused
clip+Adam kernel is used when the optimizer provides it."""
import time

import torch
from torch import nn

from .train_utils import epoch_time


class PytorchModel(nn.Module):
    """module.py:8-57"""

    def __init__(self, name, device):
        self.name = name
        super().__init__()
        if device is None:
            device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
        self.device = device

    def run(self, *input):
        raise NotImplementedError

    def loss(self, *input, **kwargs):
        raise NotImplementedError

    def inference(self, *input):
        raise NotImplementedError

    def loss_function(self, *input):
        raise NotImplementedError

    def forward(self, mode, *input, **kwargs):
        if mode in ('run', 0):
            return self.run(*input, **kwargs)
        if mode in ('loss', 'train', 1):
            return self.loss(*input, **kwargs)
        if mode in ('inference', 'eval', 'val', 2):
            return self.inference(*input, **kwargs)
        raise NotImplementedError

    def load_model(self, model_path, map_location=None):
        dic = torch.load(model_path, map_location=self.device if map_location is None else map_location)
        self.load_state_dict({k.replace('module.', ''): v for k, v in dic.items()})
        self.to(self.device)

    @staticmethod
    def init_model(*inputs):
        raise NotImplementedError


class TrainingInterface:
    """module.py:60-213"""

    def __init__(self, device, model, parallel, log_path_mng, data_loaders, summary_writers, opt_scheduler,
                 param_scheduler, n_epoch, **kwargs):
        self.model = model
        self.model.device = device
        self.model.to(device)
        self.parallel = bool(parallel)
        self.path_mng = log_path_mng
        self.summary_writers = summary_writers
        self.data_loaders = data_loaders
        self.opt_scheduler = opt_scheduler
        self.param_scheduler = param_scheduler
        self.device = device
        self.n_epoch = n_epoch
        self.epoch = self.train_step = self.val_step = 0
        self.grad_sync = None           # set to a dist.GradSync for multi-GPU data parallel
        for k, v in kwargs.items():
            setattr(self, k, v)
        if self.parallel and self.grad_sync is None:
            from ...dist import GradSync
            self.grad_sync = GradSync(self.model)

    name = property(lambda self: self.model.name)
    log_path = property(lambda self: self.path_mng.log_path)
    model_path = property(lambda self: self.path_mng.model_path)
    writer_path = property(lambda self: self.path_mng.writer_path)
    writer_names = property(lambda self: self.summary_writers.writer_names)

    def _init_loss_dic(self):
        return {k: 0. for k in self.writer_names}

    @staticmethod
    def _host_values(loss_items):
        """all scalars -> python floats with one device->host copy"""
        return torch.stack([l.detach().reshape(()) for l in loss_items]).tolist()

    def _accumulate_loss_dic(self, loss_dic, loss_items):
        assert len(self.writer_names) == len(loss_items)
        vals = loss_items if isinstance(loss_items[0], float) else self._host_values(loss_items)
        for k, v in zip(self.writer_names, vals):
            loss_dic[k] += v
        return loss_dic

    def _write_loss_to_dic(self, loss_items):
        assert len(self.writer_names) == len(loss_items)
        vals = loss_items if isinstance(loss_items[0], float) else self._host_values(loss_items)
        return dict(zip(self.writer_names, vals))

    def _batch_to_inputs(self, batch):
        raise NotImplementedError

    def _sum_parallel_loss(self, loss):
        """module.py:152-159: data-parallel loss = mean of the replicas' scalars."""
        if self.parallel and self.grad_sync is not None:
            return self.grad_sync.mean_scalars(loss)
        return loss

    def _clip_and_step(self):
        opt = self.opt_scheduler.optimizer
        if hasattr(opt, 'clip_and_step'):                      # fused HIP clip + Adam (one pass)
            opt.clip_and_step(self.opt_scheduler.clip)
            self.opt_scheduler.scheduler.step()
            self.opt_scheduler._update_step()
        else:                                                  # the reference's two calls (module.py:142-144)
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.opt_scheduler.clip)
            self.opt_scheduler.step()

    def train(self, **kwargs):
        self.model.train()
        self.param_scheduler.train()
        epoch_loss_dic = self._init_loss_dic()
        for batch in self.data_loaders.train_loader:
            inputs = self._batch_to_inputs(batch)
            self.opt_scheduler.optimizer_zero_grad()
            input_params = self.param_scheduler.step()
            outputs = self.model('train', *inputs, **input_params)
            outputs[0].backward()
            if self.grad_sync is not None:
                self.grad_sync.all_reduce_grads()
            self._clip_and_step()
            vals = self._host_values(self._sum_parallel_loss(outputs))
            self._accumulate_loss_dic(epoch_loss_dic, vals)
            self.summary_writers.write_task('train', self._write_loss_to_dic(vals), self.train_step)
            self.train_step += 1
        return epoch_loss_dic

    def eval(self):
        self.model.eval()
        self.param_scheduler.eval()
        epoch_loss_dic = self._init_loss_dic()
        for batch in self.data_loaders.val_loader:
            inputs = self._batch_to_inputs(batch)
            input_params = self.param_scheduler.step()
            with torch.no_grad():
                outputs = self._sum_parallel_loss(self.model('train', *inputs, **input_params))
            vals = self._host_values(outputs)
            self._accumulate_loss_dic(epoch_loss_dic, vals)
            self.summary_writers.write_task('val', self._write_loss_to_dic(vals), self.val_step)
            self.val_step += 1
        return epoch_loss_dic

    def save_model(self, fn):
        torch.save(self.model.state_dict(), fn)

    def epoch_report(self, start_time, end_time, train_loss, valid_loss):
        mins, secs = epoch_time(start_time, end_time)
        print(f'Epoch: {self.epoch + 1:02} | Time: {mins}m {secs}s', flush=True)
        print(f'\tTrain Loss: {train_loss:.3f}', flush=True)
        print(f'\t Valid. Loss: {valid_loss:.3f}', flush=True)

    def run(self, start_epoch=0, start_train_step=0, start_val_step=0):
        self.epoch, self.train_step, self.val_step = start_epoch, start_train_step, start_val_step
        best_valid_loss = float('inf')
        for _ in range(self.n_epoch):
            t0 = time.time()
            train_loss = self.train()['loss']
            val_loss = self.eval()['loss']
            t1 = time.time()
            self.save_model(self.path_mng.epoch_model_path(self.name))
            if val_loss < best_valid_loss:
                best_valid_loss = val_loss
                self.save_model(self.path_mng.valid_model_path(self.name))
            self.epoch_report(t0, t1, train_loss, val_loss)
            self.epoch += 1
        self.save_model(self.path_mng.final_model_path(self.name))
        print('Model saved.')
