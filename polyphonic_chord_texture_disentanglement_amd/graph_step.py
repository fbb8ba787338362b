"""The whole train step as ONE hipGraph launch.

A teacher-forced step is ~300 kernel launches on 5 HIP streams whose order, shapes and arguments depend only on (batch size,
precision, teacher-forcing pattern): the host needed 6-7 ms of Python / ctypes to enqueue what the GPU runs in 9 ms at B = 512
(and was the wall below B = 256).  `GraphedTrainStep` captures

    zero_grad -> model('train', x, c, pr_mat, tfr, beta, weights) -> loss.backward() -> fused global-norm clip + Adam

once (reference loop body: amc_dl/torch_plus/module.py:129-150) -- sibling streams, persistent recurrences and the zero-skip
decisions (device-side) included -- and replays it with one `hipGraphLaunch` per step.  What changes from step to step lives on
the device and is written before each replay:

  * the batch: copied into static input tensors (or produced in place by DeviceBatcher-style code into `step.inputs`);
  * the reparameterisation noise: two static eps tensors, refilled by `ptv_philox_normal` (keyed by seed, draw counter, global
    sample index: the same numbers the eager path draws) or by the caller's `eps_source`;
  * beta (KL annealing, train.py:56-58), the learning rate and Adam's bias corrections: a 4-float device array the loss and
    optimiser kernels read instead of their by-value arguments (`ptv_step_params`).

Only steps with a FIXED teacher-forcing pattern can be replayed (tfr = 1, or tfr = 0 with the step loop on its persistent
kernels): the python coin flips of ptvae.py:395-428 pick kernels on the host.  `matches()` tells the trainer whether a step can
use the graph; otherwise it runs eagerly -- same kernels, same results.  Data parallel: the RCCL all-reduce is issued between two
graphs (backward | optimiser) on the capture stream, so the collective itself is never captured.
"""
import math

import torch

from . import functional as F_
from ._lib import call, lib, ptr, stream_ptr


class GraphedTrainStep:
    def __init__(self, model, optimizer, batch_size, clip=1.0, tfr=(1., 1., 1.), weights=(1, 0.5), grad_sync=None, warmup=2, device=None):
        self.model, self.opt, self.clip = model, optimizer, clip
        self.tfr = tuple(float(t) for t in tfr)
        assert all(t in (0.0, 1.0) for t in self.tfr), 'only fixed teacher-forcing patterns replay (tfr 0 or 1): coin flips choose kernels on the host'
        self.weights = list(weights)
        self.sync = grad_sync if (grad_sync is not None and grad_sync.active) else None
        dev = torch.device(device) if device is not None else optimizer.flat_p.device
        self.dev, self.B = dev, int(batch_size)
        B = self.B
        self.inputs = (torch.zeros(B, 32, 16, 6, device=dev, dtype=torch.int64), torch.zeros(B, 8, 36, device=dev),
                       torch.zeros(B, 32, 128, device=dev))
        self.params = torch.zeros(4, device=dev)                       # beta, lr, 1 - b1^t, sqrt(1 - b2^t)
        # pinned staging ring for them: a slot is rewritten only after the copy that last read it has run (its event), so the host may
        # run up to RING steps ahead of the GPU without racing the asynchronous copies
        self.RING = 8
        self._host_params = [torch.zeros(4).pin_memory() for _ in range(self.RING)]
        self._host_events = [None] * self.RING
        self._slot = 0
        self._user_eps = None
        self.eps = {}
        self.losses = None
        self.graphs = None
        self.warmup = warmup
        self.replays = 0

    # ---- what a replay cannot change
    def matches(self, x, tfr, weights):
        return (x.shape[0] == self.B and tuple(float(t) for t in tfr) == self.tfr and [float(w) for w in weights] == [float(w) for w in self.weights])

    def _eps_source(self, name, shape, device):
        t = self.eps.get(name)
        if t is None:
            t = self.eps[name] = torch.zeros(shape, device=device)
        return t

    def _fill_eps(self):
        """the noise the eager path would draw at this point, into the static eps tensors"""
        m = self.model
        user = self._user_eps
        for name in ('chd', 'rhy'):                                    # draw order of DisentangleVAE.run (model.py:45-48)
            t = self.eps.get(name)
            if t is None:
                continue
            if user is not None:
                t.copy_(user(name, t.shape, t.device))
            elif m._philox is not None:
                call('ptv_philox_normal', ptr(t), t.shape[0], t.shape[1], m._philox[0], m._draws, m._philox[1], stream_ptr())
                m._draws += 1
            else:
                t.normal_()

    def _write_params(self, beta):
        g = self.opt.param_groups[0]
        t = self.opt.step_count + 1
        i = self._slot = (self._slot + 1) % self.RING
        if self._host_events[i] is not None:
            self._host_events[i].synchronize()
        h = self._host_params[i]
        # (the bias corrections exactly as ptv_clip_adam_step forms them: the betas arrive there as C floats and are raised in double)
        b1 = torch.tensor(g['betas'][0], dtype=torch.float32).item()
        b2 = torch.tensor(g['betas'][1], dtype=torch.float32).item()
        h[0], h[1] = float(beta), float(g['lr'])
        h[2], h[3] = 1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t)
        self.params.copy_(h, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._host_events[i] = ev

    def _draw_coins(self):
        """the eager step draws its teacher-forcing coins from python's `random` (ptvae.py:395-428: 32 x 14 + 31 + 8 draws, in
        DisentangleVAE.run's order) even when the ratio is 0 or 1: a replayed step consumes the same draws, so eager and replayed
        steps can be mixed without shifting the coin stream"""
        m = self.model
        m.decoder.draw_coins(self.tfr[0], self.tfr[1])
        m.chd_decoder.draw_coins(self.tfr[2])

    def _body_backward(self, beta):
        self.opt.zero_grad()
        out = self.model('train', *self.inputs, tfr1=self.tfr[0], tfr2=self.tfr[1], tfr3=self.tfr[2], beta=beta, weights=self.weights)
        out[0].backward()
        return out

    def _body_step(self):
        self.opt.clip_and_step(self.clip)

    def _eager(self, beta):
        out = self._body_backward(beta)
        if self.sync is not None:
            self.sync.all_reduce_grads()
        self._body_step()
        return out

    def _capture(self, beta):
        import random
        m, opt = self.model, self.opt
        cur = torch.cuda.current_stream()
        # warm-up steps (lazy initialisations, allocator, kernel attributes: all outside capture) are REAL steps: the training state
        # they touch is put back afterwards, so building the graph is invisible to the run
        saved = (opt.flat_p.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.step_count, m._draws, random.getstate(),
                 torch.cuda.get_rng_state(self.dev), opt.grad_scale)
        # (warm-up and capture share ONE stream: the library's per-stream reduction workspaces are allocated on a stream's first use,
        # which must not happen inside the capture)
        s = self._stream = torch.cuda.Stream(device=self.dev)
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            for _ in range(max(1, self.warmup)):
                self._fill_eps()
                self._eager(beta)
        cur.wait_stream(s)
        torch.cuda.synchronize(self.dev)
        opt.flat_p.copy_(saved[0]); opt.exp_avg.copy_(saved[1]); opt.exp_avg_sq.copy_(saved[2])
        opt.step_count, m._draws, opt.grad_scale = saved[3], saved[4], saved[7]
        if self.sync is not None:
            # the optimiser graph bakes grad_scale in BY VALUE: under data parallelism every replay is preceded by an all-reduce(SUM),
            # so the captured value must be 1/world whatever the optimiser held before its first exchange (round-3 advice: a graph
            # built before any eager step captured 1.0 and every replayed step clipped / stepped on world x the gradient)
            opt.grad_scale = 1.0 / self.sync.world
        random.setstate(saved[5])
        torch.cuda.set_rng_state(saved[6], self.dev)
        # The captured step must record exactly what a steady-state step enqueues: the transposed bf16 shadows and the fragment-packed
        # weights are rebuilt (their stamps are stale after every optimiser step), the plain bf16 shadow is NOT (the Adam kernel of the
        # previous step wrote it).  So: bring the plain shadow up to date eagerly, then mark it fresh and everything else stale.
        opt.mark_dirty()
        call('ptv_cast_bf16', ptr(opt.flat_p), ptr(opt.flat_p16), opt.arena.total, stream_ptr())
        torch.cuda.synchronize(self.dev)
        opt._plain_stamp = opt._stamp()
        opt._shadow_stamp = None
        lib().ptv_step_params(self.params.data_ptr())
        fb0 = F_.ordered_fallbacks()
        cap = F_.whole_step_capture()
        cap.__enter__()
        early = None
        if self.sync is not None:                                      # collectives are issued between the graphs, never captured
            early, self.sync.early = self.sync.early, False
        try:
            pool = torch.cuda.graph_pool_handle()
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1, pool=pool, stream=self._stream):
                cap.origin()
                out = self._body_backward(beta if beta != 0 else 1e-30)   # (a non-zero by-value beta marks the calls that read the device beta)
                self.losses = torch.stack([o.detach().reshape(()) for o in out])
                F_.join_captured_streams()
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=pool, stream=self._stream):
                cap.origin()
                self._body_step()
                F_.join_captured_streams()
        finally:
            cap.__exit__()
            lib().ptv_step_params(None)
            if early is not None:
                self.sync.early = early
        self.graphs = (g1, g2)
        # a reduction that found no workspace inside the capture fell back to fp32 atomics: the replayed step would no longer be the
        # eager step bit for bit (the warm-up steps above exist to make every first-use allocation happen outside the capture)
        self.capture_fallbacks = F_.ordered_fallbacks() - fb0
        if self.capture_fallbacks and F_.ORDERED_STRICT:
            raise RuntimeError('GraphedTrainStep: %d reductions were captured on the atomics fallback' % self.capture_fallbacks)
        # the capture pass itself executed nothing: undo its python-side bookkeeping (the first replay is the step)
        opt.step_count -= 1
        random.setstate(saved[5])

    def __call__(self, x, c, pr_mat, beta=0.1):
        """one optimisation step on (x, c, pr_mat); returns the 11 losses of train.py:54-55 as a device tensor [11] (static: read or
        copy it before the next call)"""
        m, opt = self.model, self.opt
        for dst, src in zip(self.inputs, (x, c, pr_mat)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self._user_eps = m.eps_source
        m.eps_source = self._eps_source
        try:
            if self.graphs is None:
                self._write_params(beta)
                self._capture(beta)
            elif opt.flat_p16 is not None and getattr(self, '_replay_stamp', None) != opt._stamp():
                # parameters were written behind the graph's back since the last replay (load_state_dict / load_checkpoint / manual
                # edits): the captured forward reads the plain bf16 shadow the PREVIOUS Adam kernel wrote -- bring it up to date
                call('ptv_cast_bf16', ptr(opt.flat_p), ptr(opt.flat_p16), opt.arena.total, stream_ptr())
            self._write_params(beta)
            self._fill_eps()
            self._draw_coins()
            self.graphs[0].replay()
            if self.sync is not None:
                self.sync.all_reduce_grads()
            self.graphs[1].replay()
        finally:
            m.eps_source = self._user_eps
        opt.step_count += 1
        opt._opt_called = True
        opt.mark_dirty()               # python-side stamps know nothing of the replayed kernels (an eager step after this re-casts the shadows)
        self._replay_stamp = opt._stamp()
        self.replays += 1
        return self.losses
