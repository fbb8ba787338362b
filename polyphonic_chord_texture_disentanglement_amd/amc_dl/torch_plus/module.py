"""Model base class and training loop of the trainer surface (API of the reference's
amc_dl/torch_plus/module.py), built for one process per MI355X:

* `PytorchModel.forward(mode, ...)` dispatches 'run'/0, 'loss'/'train'/1, 'inference'/'eval'/'val'/2
  exactly like module.py:36-44.
* `TrainingInterface` keeps the constructor, properties, `train/eval/run/save_model/epoch_report` and the
  per-batch order of module.py:129-150 (zero_grad -> parameter schedule -> model('train', ...) -> backward
  -> clip -> optimizer + LR step -> log).  `parallel=True` means torch.distributed data parallelism with
  ONE RCCL all-reduce of the flat gradient bucket (not nn.DataParallel, module.py:67-68); the 11 logged
  scalars leave the device in one transfer per batch (the reference does 22 `.item()` syncs,
  module.py:113-124); a `FusedClipAdam` optimizer clips and steps in two kernel launches.
"""
import time

import torch
from torch import nn

from .train_utils import epoch_time

_RUN, _LOSS, _INFER = ('run', 0), ('loss', 'train', 1), ('inference', 'eval', 'val', 2)


class PytorchModel(nn.Module):

    def __init__(self, name, device):
        self.name = name
        nn.Module.__init__(self)
        self.device = device if device is not None else torch.device('cuda' if torch.cuda.is_available() else 'cpu')

    # subclasses provide these four
    def run(self, *input):
        raise NotImplementedError

    def loss(self, *input, **kwargs):
        raise NotImplementedError

    def inference(self, *input):
        raise NotImplementedError

    def loss_function(self, *input):
        raise NotImplementedError

    def forward(self, mode, *input, **kwargs):
        for keys, fn in ((_RUN, self.run), (_LOSS, self.loss), (_INFER, self.inference)):
            if mode in keys:
                return fn(*input, **kwargs)
        raise NotImplementedError

    def load_model(self, model_path, map_location=None):
        """state_dict file -> parameters; 'module.' prefixes of DataParallel checkpoints are dropped."""
        state = torch.load(model_path, map_location=map_location if map_location is not None else self.device)
        self.load_state_dict({key.replace('module.', ''): val for key, val in state.items()})
        self.to(self.device)

    @staticmethod
    def init_model(*inputs):
        raise NotImplementedError


class TrainingInterface:

    def __init__(self, device, model, parallel, log_path_mng, data_loaders, summary_writers, opt_scheduler,
                 param_scheduler, n_epoch, **kwargs):
        model.device = device
        self.model = model.to(device)
        self.device, self.parallel, self.n_epoch = device, bool(parallel), n_epoch
        self.path_mng, self.summary_writers, self.data_loaders = log_path_mng, summary_writers, data_loaders
        self.opt_scheduler, self.param_scheduler = opt_scheduler, param_scheduler
        self.epoch = self.train_step = self.val_step = 0
        self.grad_sync = None                          # dist.GradSync when data parallel
        self._pending_logs = []
        self.__dict__.update(kwargs)
        if self.parallel and self.grad_sync is None:
            from ...dist import GradSync
            self.grad_sync = GradSync(self.model, getattr(opt_scheduler, 'optimizer', None))

    # ---- read-only views the reference exposes as properties
    name = property(lambda self: self.model.name)
    log_path = property(lambda self: self.path_mng.log_path)
    model_path = property(lambda self: self.path_mng.model_path)
    writer_path = property(lambda self: self.path_mng.writer_path)
    writer_names = property(lambda self: self.summary_writers.writer_names)

    # ---- loss bookkeeping
    @staticmethod
    def _host_values(loss_items):
        """scalar tensors -> python floats with ONE device-to-host copy"""
        if loss_items and isinstance(loss_items[0], float):
            return list(loss_items)
        return torch.stack([item.detach().reshape(()) for item in loss_items]).tolist()

    def _init_loss_dic(self):
        return dict.fromkeys(self.writer_names, 0.)

    def _write_loss_to_dic(self, loss_items):
        assert len(loss_items) == len(self.writer_names)
        return dict(zip(self.writer_names, self._host_values(loss_items)))

    def _accumulate_loss_dic(self, loss_dic, loss_items):
        for key, val in self._write_loss_to_dic(loss_items).items():
            loss_dic[key] += val
        return loss_dic

    def _sum_parallel_loss(self, loss):
        """data-parallel reporting = mean of the replicas' scalars (module.py:152-159)"""
        return self.grad_sync.mean_scalars(loss) if (self.parallel and self.grad_sync is not None) else loss

    def _batch_to_inputs(self, batch):
        raise NotImplementedError

    # ---- one optimisation step
    def _clip_and_step(self):
        sched, opt = self.opt_scheduler, self.opt_scheduler.optimizer
        if hasattr(opt, 'clip_and_step'):
            opt.clip_and_step(sched.clip)              # fused HIP global-norm clip + Adam
            sched.scheduler.step()
            sched._update_step()
        else:                                          # torch optimizer: the reference's two calls
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), sched.clip)
            sched.step()

    # ---- logging without stalling the pipeline.  The reference reads 22 `.item()`s per batch (module.py:113-124); one
    # blocking read still makes the host wait for the whole step before it can enqueue the next one (8 ms of enqueue
    # serialised behind 16 ms of GPU work at B = 512).  Here the 11 scalars are copied into a pinned buffer with an async
    # D2H and an event; the entry is consumed when the NEXT batch logs (or at the end of the pass), so the host runs one
    # step ahead of the device.  `log_lag = 0` restores the blocking behaviour.
    log_lag = 1

    def _log(self, task, outputs, loss_dic, step):
        vals = torch.stack([item.detach().reshape(()) for item in self._sum_parallel_loss(outputs)])
        if not vals.is_cuda or self.log_lag <= 0:
            return self._consume_log(task, vals.tolist(), loss_dic, step)
        host = torch.empty(vals.shape, dtype=vals.dtype, pin_memory=True)
        host.copy_(vals, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pending_logs.append((task, host, ev, loss_dic, step))
        while len(self._pending_logs) > self.log_lag:
            self._pop_log()

    def _pop_log(self):
        task, host, ev, loss_dic, step = self._pending_logs.pop(0)
        ev.synchronize()
        self._consume_log(task, host.tolist(), loss_dic, step)

    def _flush_logs(self):
        while self._pending_logs:
            self._pop_log()

    def _consume_log(self, task, vals, loss_dic, step):
        self._accumulate_loss_dic(loss_dic, vals)
        if self.is_main:
            self.summary_writers.write_task(task, self._write_loss_to_dic(vals), step)

    # ---- the step as one hipGraph launch (graph_step.GraphedTrainStep).  graph_step = 'auto' (default; PTV_GRAPH_STEP overrides):
    # replay only where it was MEASURED faster than the eager step.  Round 3 that was per-GPU batch <= 128; since the backward passes
    # moved behind one C call each (round 5) the eager step wins at 128 and above (BENCH_r05: B = 128 eager 4.10 ms vs replayed 4.36,
    # B = 256 4.94 vs 5.40, B = 512 7.06 vs 7.66 -- the capture routes sibling-stream edges through the origin stream, see
    # graph_step.py), so the crossover sits below 128: the default replays at B <= 64 only (bench.py reports the eager / replayed pair
    # at 64, 128 and 256 every round: `extra.train_teacher_forced_b*`).  True / False force it.  Only steps whose teacher-forcing
    # ratios are all exactly 1 can replay (coin flips pick kernels on the host); every other step runs eagerly.
    graph_step = 'auto'
    GRAPH_AUTO_MAX_BATCH = 64
    # the training LOOP (not the optimiser) opts into optim.freeze_gc() once its working set exists: the first generation-2 collection
    # otherwise lands around the 20th step as a ~100-ms stall.  Process-wide, idempotent, undone by optim.unfreeze_gc(); None = never.
    freeze_gc_after_steps = 3

    def _graphed(self, inputs, params):
        import os
        mode = os.environ.get('PTV_GRAPH_STEP', self.graph_step)
        mode = {'1': True, 'on': True, '0': False, 'off': False}.get(str(mode).lower(), mode)
        opt = self.opt_scheduler.optimizer
        if mode is False or not hasattr(opt, 'clip_and_step') or not inputs[0].is_cuda:
            return None
        tfr = (params.get('tfr1', 0.), params.get('tfr2', 0.), params.get('tfr3', 0.))
        if any(float(t) != 1.0 for t in tfr) or 'weights' not in params:
            return None
        B = inputs[0].shape[0]
        if mode == 'auto' and (B > self.GRAPH_AUTO_MAX_BATCH or (self.grad_sync is not None and self.grad_sync.active)):
            return None                                        # (data parallel: only when asked for -- the early, overlapped exchange needs the eager step)
        gs = self.__dict__.setdefault('_graph_steps', {})
        key = (B, tuple(float(w) for w in params['weights']), float(self.opt_scheduler.clip))
        if key not in gs:
            from ...graph_step import GraphedTrainStep
            if len(gs) >= 2:                                   # (a ragged last batch gets its own graph; more shapes than that: stay eager)
                return None
            gs[key] = GraphedTrainStep(self.model, opt, B, clip=self.opt_scheduler.clip, tfr=tfr, weights=params['weights'],
                                       grad_sync=self.grad_sync)
        return gs[key]

    def train(self, **kwargs):
        self.model.train()
        self.param_scheduler.train()
        epoch_loss_dic = self._init_loss_dic()
        for batch in self.data_loaders.train_loader:
            inputs = self._batch_to_inputs(batch)
            params = self.param_scheduler.step()
            g = self._graphed(inputs, params)
            if g is not None:
                losses = g(*inputs, beta=params.get('beta', 0.1))
                sched = self.opt_scheduler
                sched.scheduler.step()
                sched._update_step()
                self._log('train', tuple(losses.unbind(0)), epoch_loss_dic, self.train_step)
                self.train_step += 1
                continue
            self.opt_scheduler.optimizer_zero_grad()
            outputs = self.model('train', *inputs, **params)
            # the backward pass on THIS thread: the autograd engine's device thread costs 0.5-1.3 ms of hand-offs per step for ~100 nodes
            # that only enqueue kernels (B = 256 eager: 40.7k -> 44.5k samples/s, B = 128: 21k -> 26k; neutral at B = 512, GPU-bound)
            with torch.autograd.set_multithreading_enabled(False):
                outputs[0].backward()
            if self.grad_sync is not None:
                self.grad_sync.all_reduce_grads()
            self._clip_and_step()
            self._log('train', outputs, epoch_loss_dic, self.train_step)
            self.train_step += 1
            if self.freeze_gc_after_steps is not None and self.train_step == self.freeze_gc_after_steps:
                from ...optim import freeze_gc
                freeze_gc()
        self._flush_logs()
        return epoch_loss_dic

    def eval(self):
        """validation pass: same call as training (mode 'train', module.py:170) without gradients; the
        parameter schedulers are frozen in 'val' mode"""
        self.model.eval()
        self.param_scheduler.eval()
        epoch_loss_dic = self._init_loss_dic()
        for batch in self.data_loaders.val_loader:
            inputs = self._batch_to_inputs(batch)
            with torch.no_grad():
                outputs = self.model('train', *inputs, **self.param_scheduler.step())
            self._log('val', outputs, epoch_loss_dic, self.val_step)
            self.val_step += 1
        self._flush_logs()
        return epoch_loss_dic

    @property
    def is_main(self):
        """rank 0 of a data-parallel job (every rank holds identical weights: only one writes files / logs)"""
        import torch.distributed as dist
        return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0

    def _rank(self):
        import torch.distributed as dist
        return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0

    def _world(self):
        import torch.distributed as dist
        return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1

    def _barrier(self):
        import torch.distributed as dist
        if self.parallel and dist.is_available() and dist.is_initialized():
            dist.barrier()

    def save_model(self, fn):
        """weights only, as the reference (module.py:179-183); written by rank 0"""
        if self.is_main:
            torch.save(self.model.state_dict(), fn)
        self._barrier()

    # ---- full-state checkpoints: weights + optimiser (Adam moments, step count, lr) + LR scheduler
    # + parameter-scheduler counters + the trainer's epoch / step counters + everything that decides the NEXT batches and noise:
    # the model's Philox key and draw counter, python's `random` state (the teacher-forcing coins, ptvae.py:395-428), torch's
    # device generator (the default eps source) and the generators of device-resident loaders (DeviceBatcher.gen)
    def _loader_gens(self):
        out = {}
        for k in ('train_loader', 'val_loader'):
            g = getattr(getattr(self.data_loaders, k, None), 'gen', None)
            if isinstance(g, torch.Generator):
                out[k] = g
        return out

    def save_checkpoint(self, fn):
        import random
        opt, sch = self.opt_scheduler.optimizer, self.opt_scheduler.scheduler
        m = self.model
        rng = {'philox': getattr(m, '_philox', None), 'draws': getattr(m, '_draws', 0), 'python_random': random.getstate(),
               'torch_cpu': torch.get_rng_state(),
               'torch_cuda': torch.cuda.get_rng_state(self.device) if torch.cuda.is_available() and torch.device(self.device).type == 'cuda' else None,
               'loaders': {k: g.get_state() for k, g in self._loader_gens().items()}}
        state = {'model': m.state_dict(), 'optimizer': opt.state_dict(), 'lr_scheduler': sch.state_dict(),
                 'opt_scheduler_step': self.opt_scheduler._step, 'param_scheduler': self.param_scheduler.state_dict(),
                 'epoch': self.epoch, 'train_step': self.train_step, 'val_step': self.val_step, 'rng': rng}
        stamp = {'world': self._world(), 'train_step': self.train_step, 'epoch': self.epoch}
        state['sidecar_stamp'] = stamp
        if self.is_main:
            torch.save(state, fn)
            # sidecars of ranks that do not exist in THIS run (a checkpoint overwritten by fewer processes) must not survive beside it
            import glob
            import os
            for old in glob.glob(glob.escape(fn) + '.rng*'):
                tail = old[len(fn) + 4:]
                if tail.isdigit() and int(tail) >= stamp['world']:
                    os.remove(old)
        else:
            # every other rank keeps ITS random state next to the checkpoint: the Philox sample offset (rank * per-GPU batch), the device
            # generator and the loader generators differ by rank, and restoring rank 0's on all ranks would give every replica the same
            # noise for different samples and the same shard of the data.  Stamped with the run it belongs to (round-4 advice): a
            # sidecar left over from another world size / step is ignored on load
            torch.save({'rng': rng, 'stamp': dict(stamp, rank=self._rank())}, '%s.rng%d' % (fn, self._rank()))
        self._barrier()

    def load_checkpoint(self, fn):
        import random
        state = torch.load(fn, map_location=self.device, weights_only=False)
        self.model.load_state_dict(state['model'])
        self.opt_scheduler.optimizer.load_state_dict(state['optimizer'])
        self.opt_scheduler.scheduler.load_state_dict(state['lr_scheduler'])
        self.opt_scheduler._step = state['opt_scheduler_step']
        self.param_scheduler.load_state_dict(state['param_scheduler'])
        self.epoch, self.train_step, self.val_step = state['epoch'], state['train_step'], state['val_step']
        self._resumed = True                                   # run() continues from these counters unless told otherwise
        rng = state.get('rng')
        if rng is not None and not self.is_main:
            import os
            own = '%s.rng%d' % (fn, self._rank())
            side = torch.load(own, map_location=self.device, weights_only=False) if os.path.exists(own) else None
            want = dict(state.get('sidecar_stamp') or {}, rank=self._rank())
            if side is not None and side.get('stamp') == want and want.get('world') == self._world():
                rng = side['rng']
            else:
                # no per-rank block (checkpoint written by a single process): the shared parts only -- the noise seed and draw counter
                # (this rank's own sample offset stays) and the coin stream (identical on every rank by construction)
                if hasattr(self.model, '_philox') and rng.get('philox') is not None and self.model._philox is not None:
                    self.model._philox = (rng['philox'][0], self.model._philox[1])
                    self.model._draws = rng['draws']
                random.setstate(rng['python_random'])
                rng = None
        if rng is not None:
            if hasattr(self.model, '_philox'):
                self.model._philox, self.model._draws = rng['philox'], rng['draws']
            random.setstate(rng['python_random'])
            torch.set_rng_state(rng['torch_cpu'].cpu())
            if rng.get('torch_cuda') is not None and torch.cuda.is_available():
                torch.cuda.set_rng_state(rng['torch_cuda'].cpu(), self.device)
            gens = self._loader_gens()
            for k, st in rng.get('loaders', {}).items():
                if k in gens:
                    gens[k].set_state(st.cpu())

    def epoch_report(self, start_time, end_time, train_loss, valid_loss):
        mins, secs = epoch_time(start_time, end_time)
        for line in (f'Epoch: {self.epoch + 1:02} | Time: {mins}m {secs}s', f'\tTrain Loss: {train_loss:.3f}',
                     f'\t Valid. Loss: {valid_loss:.3f}'):
            print(line, flush=True)

    def run(self, start_epoch=None, start_train_step=None, start_val_step=None):
        """n_epoch x (train, eval); checkpoints '<name>_epoch.pt' every epoch, '<name>_valid.pt' on a new
        best validation loss, '<name>_final.pt' at the end (module.py:195-213).  The start_* arguments are the reference's
        (default 0 there); None = 0 on a fresh trainer, the restored counters after load_checkpoint()."""
        keep = getattr(self, '_resumed', False)
        self.epoch = start_epoch if start_epoch is not None else (self.epoch if keep else 0)
        self.train_step = start_train_step if start_train_step is not None else (self.train_step if keep else 0)
        self.val_step = start_val_step if start_val_step is not None else (self.val_step if keep else 0)
        best = float('inf')
        for _ in range(self.n_epoch):
            tic = time.time()
            train_loss, val_loss = self.train()['loss'], self.eval()['loss']
            toc = time.time()
            self.save_model(self.path_mng.epoch_model_path(self.name))
            if val_loss < best:
                best = val_loss
                self.save_model(self.path_mng.valid_model_path(self.name))
            if self.is_main:
                self.epoch_report(tic, toc, train_loss, val_loss)
            self.epoch += 1
        self.save_model(self.path_mng.final_model_path(self.name))
        if self.is_main:
            print('Model saved.')
