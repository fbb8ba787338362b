"""Flat-buffer optimiser for the train step: clip_grad_norm_ + Adam in two HIP launches.

Replaces `torch.nn.utils.clip_grad_norm_(model.parameters(), clip)` + `torch.optim.Adam.step()`
(reference amc_dl/torch_plus/module.py:142-144, train.py:50; Adam defaults beta (0.9, 0.999),
eps 1e-8, no weight decay) with `ptv_grad_sumsq` + `ptv_clip_adam_step` over ONE flat fp32 buffer
each for parameters, gradients and the two moments: 7 x 109 MB of HBM traffic per step, no host
sync (the norm stays on the device).  The flat gradient buffer is also the RCCL all-reduce bucket.

Parameters are re-pointed to views of the flat parameter buffer (state_dict keys/shapes are
unchanged).  Gradients: `GradArena` hands the backward kernels zero-initialised views of the flat
gradient buffer, so `.grad` of every parameter already lives in the bucket when backward ends.
"""
import os
import weakref

import torch

from ._lib import call, ptr, stream_ptr

_ARENA_OF = {}          # id(param) -> GradArena
SHADOW_T_ASYNC = True  # transposed bf16 shadows refreshed on a sibling stream (read by the backward only)
ADAM_SHADOW = True      # the Adam kernel also writes the bf16 operand copy of the parameters
_SHADOW_OF = {}         # param data_ptr -> (FusedClipAdam, offset, numel): bf16 copies of the flat parameter buffer


class GradArena:
    def __init__(self, params):
        self.params = list(params)
        dev = self.params[0].device
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 7) // 8 * 8            # every view 16-byte aligned in the fp32 AND the bf16 copy
        self.total = off
        self.flat = torch.zeros(off, device=dev, dtype=torch.float32)
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._shapes = [tuple(p.shape) for p in self.params]
        self._strides = [tuple(p.contiguous().stride()) if p.dim() else () for p in self.params]
        self._handed = set()
        self.epoch = 0                  # bumped by zero(): what an in-flight early all-reduce (dist.GradSync) was started under
        self.in_flight = None           # callable(param) -> bool set by dist.GradSync: p's range is being all-reduced right now
        self.on_zero = []               # callables run by zero() before the bucket is cleared (GradSync drops stale early parts)
        for p in self.params:
            _ARENA_OF[id(p)] = self

    def view(self, p):
        i = self._index[id(p)]
        # (one op instead of slice + view: 81 of these per step)    fresh TensorImpl: autograd may adopt it as .grad
        return self.flat.as_strided(self._shapes[i], self._strides[i], self.offsets[i])

    def take(self, p):
        """zeroed gradient view for p, once per zero(); None if already handed out this step."""
        if id(p) in self._handed:
            # a second contribution to p in this step (shared weight, two forwards before one backward, gradient accumulation):
            # autograd will ADD it into the arena view -- which must not be on the wire already
            if self.in_flight is not None and self.in_flight(p):
                raise RuntimeError('GradArena: a second gradient contribution arrived for a parameter whose slice of the bucket is already '
                                   'being all-reduced (dist.GradSync early exchange requires every weight to be used once per step and one '
                                   'backward per zero_grad; set PTV_EARLY_ALLREDUCE=0 for shared weights or gradient accumulation)')
            return None
        self._handed.add(id(p))
        return self.view(p)

    def zero(self):
        for f in self.on_zero:
            f()
        self.epoch += 1
        self.flat.zero_()
        self._handed.clear()

    def holds_all_grads(self):
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * o:
                return False
        return True

    def gather_grads(self):
        """slow path (a gradient was produced outside the arena): copy .grad into the bucket"""
        for p in self.params:
            v = self.view(p)
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
            p.grad = v


def weight_shadow(p):
    """bf16 copy of parameter p (a view of the optimiser's bf16 shadow of the flat parameter buffer), or None.
    In bf16 precision every weight is an MFMA operand only: converting the 27M parameters once per step
    replaces a conversion in every tile load of every product."""
    ent = _SHADOW_OF.get(p.data_ptr())
    opt = ent[0]() if ent is not None else None
    if opt is None or ent[2] != p.numel() or not p.is_contiguous():
        return None
    off = ent[1]
    opt.refresh_if_param_stale(p)
    if off in opt._row_padded:                       # rows not a multiple of 8 elements: separate row-padded copy
        return opt._row_padded[off][1][:, :p.shape[1]]
    return opt.flat_p16[off:off + p.numel()].view(p.shape)


def weight_shadow_t(p):
    """bf16 TRANSPOSED copy of a 2-D parameter p [N,K] -> [K,N] (K-contiguous operand of the dX products
    dY . W and of the BPTT step), or None"""
    ent = _SHADOW_OF.get(p.data_ptr())
    opt = ent[0]() if ent is not None else None
    if opt is None or p.dim() != 2 or ent[2] != p.numel() or not p.is_contiguous():
        return None
    off = ent[1]
    if off not in opt._mat_offsets:
        return None
    opt.refresh_if_param_stale(p)
    if opt._t_event is not None:
        # (once per stream and refresh: a stream that has waited orders everything it runs afterwards -- this was 25 event waits and
        # current-stream lookups per step)
        from . import functional as F_
        sid = F_.stream_ptr()
        if sid not in opt._t_waited:
            F_.wait_event(F_.cur_stream(), opt._t_event)
            if not F_.WHOLE_STEP_CAPTURE:
                opt._t_waited.add(sid)
    return opt.flat_pT16[off:off + p.numel()].view(p.shape[1], p.shape[0])


def refresh_weight_shadows():
    """bring the bf16 operand copies of every registered flat parameter buffer up to date (called at the start of each
    forward / inference entry point, on the caller's stream, before work forks to sibling streams).  A no-op when the
    parameters have not changed since the last cast: the copies carry a stamp (optimiser step count + the parameters'
    in-place version counters, which `load_state_dict` / `copy_` / `uniform_` bump)."""
    live = {}
    for key, e in list(_SHADOW_OF.items()):
        opt = e[0]()
        if opt is None:
            del _SHADOW_OF[key]                      # its optimiser (and flat buffers) are gone
        else:
            live[id(opt)] = opt
    for opt in live.values():
        opt.refresh_shadow_if_stale()


def is_arena_view(p, g):
    """g is the gradient view the arena handed out for parameter p in this step"""
    a = _ARENA_OF.get(id(p))
    if a is None or g is None:
        return False
    return g.data_ptr() == a.flat.data_ptr() + 4 * a.offsets[a._index[id(p)]]


def grad_buffer(p):
    """Used by functional.py: arena view when the parameter is registered, else a fresh zeros."""
    a = _ARENA_OF.get(id(p))
    if a is not None:
        v = a.take(p)
        if v is not None:
            return v
    return torch.zeros_like(p, memory_format=torch.contiguous_format)


_GC_FROZEN = False


def freeze_gc():
    """One full collection, then move everything alive to the permanent generation (gc.freeze()): later collections walk only what was
    created since, so no ~100-ms generation-2 pass lands inside a training loop.  Process-wide and idempotent (a second call is a no-op
    until unfreeze_gc()); cyclic garbage among the objects frozen now is not collected until then."""
    global _GC_FROZEN
    if _GC_FROZEN:
        return False
    import gc
    gc.collect()
    gc.freeze()
    _GC_FROZEN = True
    return True


def unfreeze_gc():
    global _GC_FROZEN
    import gc
    gc.unfreeze()
    _GC_FROZEN = False


def reserve_step_memory(batch, device=None, gb_per_512=(12.0, 3.0)):
    """Warm the caching allocator for train steps of `batch` samples: one large block on the current stream and one on each sibling pool
    stream (functional.Side), allocated and released -- they stay cached per stream and later requests are carved out of them.  Without it
    the allocator grows by hipMalloc (a device-wide synchronisation, ~70 ms) for as long as the step's working set keeps changing: measured
    at B = 512 a step around the seventh takes 77 ms instead of 7 (the first full garbage collection frees reference cycles that held
    blocks until then).  Sizes: the steady state at B = 512 holds 5.6 GB on the step's stream and <= 1.7 GB per pool stream; HBM is 288 GB.
    Returns the bytes reserved."""
    from . import functional as F_
    dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
    scale = max(batch, 64) / 512.0
    main = F_.cur_stream()
    F_.Side(0)                                               # (creates the pool streams in their fixed order)
    total = 0
    streams = [(main, gb_per_512[0])] + [(st, gb_per_512[1]) for key, st in sorted(F_._CHILD_STREAMS.items(), key=lambda kv: str(kv[0]))
                                         if key[0] == 'pool' and key[1] == dev.index]
    for st, gb in streams:
        with torch.cuda.stream(st):
            n = int(gb * scale * 2 ** 30)
            big = torch.empty(n, dtype=torch.uint8, device=dev)
            small = [torch.empty(512 * 1024, dtype=torch.uint8, device=dev) for _ in range(128)]      # the small-block pool (2-MB segments)
            total += n + 128 * 512 * 1024
            del big, small
    return total


class FusedClipAdam(torch.optim.Optimizer):
    """Adam with global-norm clipping fused in (`clip_and_step(clip)`); `step()` alone = no clipping."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_steps_in_flight=2, gc_freeze_at_step=None):
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        # host-side pacing, both explicit (round-5 advice): at most `max_steps_in_flight` optimiser steps queued ahead of the GPU (None: never
        # wait); `gc_freeze_at_step` = n calls optim.freeze_gc() -- process-wide, once -- after the n-th step (None, the default: the
        # collector is left alone; a training script that wants steady step times calls optim.freeze_gc() itself after its warm-up)
        self.MAX_STEPS_IN_FLIGHT = max_steps_in_flight
        self.GC_FREEZE_AT_STEP = gc_freeze_at_step
        ps = [p for g in self.param_groups for p in g['params']]
        assert len(self.param_groups) == 1 and all(p.is_cuda and p.dtype == torch.float32 for p in ps), \
            'FusedClipAdam: one group of fp32 cuda parameters expected (move the model to the GPU first)'
        self.arena = GradArena(ps)
        self.flat_p = torch.empty(self.arena.total, device=ps[0].device, dtype=torch.float32).zero_()
        for p, o in zip(ps, self.arena.offsets):
            dst = self.flat_p[o:o + p.numel()].view(p.shape)
            dst.copy_(p.data)
            p.data = dst
        self.flat_p16 = torch.empty(self.arena.total, device=ps[0].device, dtype=torch.bfloat16)
        self.flat_pT16 = torch.empty(self.arena.total, device=ps[0].device, dtype=torch.bfloat16)
        self._mats = [(p, o) for p, o in zip(ps, self.arena.offsets) if p.dim() == 2 and p.shape[0] % 8 == 0 and p.shape[1] >= 8]
        self._mat_offsets = {o for _, o in self._mats}
        self._tdesc = None
        # matrices whose rows are not a multiple of 8 elements (dur_hid_linear [64, 642]): a bf16 view of the flat
        # buffer would start every row off the 16-byte grid, so they get their own copy with rows padded to 8
        self._row_padded = {o: (p, torch.zeros(p.shape[0], (p.shape[1] + 7) // 8 * 8, device=p.device, dtype=torch.bfloat16))
                            for p, o in zip(ps, self.arena.offsets) if p.dim() == 2 and p.shape[1] % 8 != 0 and p.shape[1] >= 64}
        for p, o in zip(ps, self.arena.offsets):
            _SHADOW_OF[p.data_ptr()] = (weakref.ref(self), o, p.numel())
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.sumsq = torch.zeros(1, device=ps[0].device, dtype=torch.float32)
        self.step_count = 0
        self._dirty = 0
        self._shadow_stamp = None
        self._fast_key, self._pver = None, {}
        self._t_event = None            # recorded after the last transposed-shadow refresh when that ran on a sibling stream
        self._t_waited = set()          # raw streams that already wait for it
        self._plain_stamp = None        # stamp at which flat_p16 (the untransposed bf16 copy) was last written
        self.grad_scale = 1.0           # set to 1/world_size when the bucket holds a SUM over ranks
        self.refresh_shadow()

    def _stamp(self):
        return (self.step_count, self._dirty, sum(p._version for p in self.arena.params))

    def mark_dirty(self):
        """call after writing parameters through a path torch cannot see (raw pointers, `.data` tricks)"""
        self._dirty += 1

    def refresh_shadow_if_stale(self):
        if self._shadow_stamp != self._stamp():
            self.refresh_shadow()

    def refresh_if_param_stale(self, p):
        """per-weight form of the check (runs ~60 times per step): the copies of p are current iff no optimiser step / mark_dirty()
        happened since the last refresh AND p itself was not written in place since (its version counter; aliases share it) -- the
        full stamp sums the version counters of all 81 parameters (8 us a time)"""
        if self._fast_key == (self.step_count, self._dirty) and self._pver.get(p.data_ptr()) == p._version:
            return
        self.refresh_shadow_if_stale()
        if self._pver.get(p.data_ptr()) != p._version:       # the stamp was unchanged but this check failed: p is an unknown alias
            self._pver[p.data_ptr()] = p._version

    def refresh_shadow(self):
        self._shadow_stamp = self._stamp()
        self._fast_key = (self.step_count, self._dirty)
        self._pver = {q.data_ptr(): q._version for q in self.arena.params}
        st = stream_ptr()
        if self._plain_stamp != self._shadow_stamp:          # (after an optimiser step the Adam kernel has written flat_p16 already)
            call('ptv_cast_bf16', ptr(self.flat_p), ptr(self.flat_p16), self.arena.total, st)
            self._plain_stamp = self._shadow_stamp
        if self._mats:                                           # transposed copies of the matrices: one launch
            if self._tdesc is None:
                d, t = [], 0
                for p, o in self._mats:
                    d += [o, p.shape[0], p.shape[1], t]
                    t += ((p.shape[0] + 31) // 32) * ((p.shape[1] + 31) // 32)
                self._tdesc = (torch.tensor(d, dtype=torch.int64, device=self.flat_p.device), t)
            def transposes():
                call('ptv_transpose_cast_bf16_batched', ptr(self.flat_p), ptr(self.flat_pT16), ptr(self._tdesc[0]), len(self._mats),
                     self._tdesc[1], stream_ptr())
            from . import functional as F_
            self._t_event = None
            self._t_waited = set()
            if SHADOW_T_ASYNC and F_.OVERLAP and not F_.capturing_part():
                # only the backward pass reads the transposed copies (dX products, BPTT): refresh them on a sibling stream, off the
                # head of the step; weight_shadow_t() makes its caller's stream wait for the event
                side = F_.Side(F_.SHADOW_T_SLOT)
                side(transposes)
                self._t_event = F_.record_event(side.s)
                self._t_waited = set()
            else:
                transposes()
        for p, buf in self._row_padded.values():
            buf[:, :p.shape[1]].copy_(p.data)

    def zero_grad(self, set_to_none=True):
        from .functional import reset_deferred
        reset_deferred()                   # side-stream gradient work of an earlier (possibly aborted) backward
        for p in self.arena.params:
            p.grad = None
        self.arena.zero()

    # ---- checkpointing: Adam moments and the step count live outside torch's per-parameter `state`, so the generic
    # Optimizer.state_dict() would lose them (the reference saves weights only, module.py:179-183)
    def state_dict(self):
        n = [p.numel() for p in self.arena.params]
        unpad = lambda flat: torch.cat([flat[o:o + k] for o, k in zip(self.arena.offsets, n)]).cpu()
        g = self.param_groups[0]
        return {'format': 'FusedClipAdam/1', 'step_count': self.step_count, 'lr': g['lr'], 'betas': tuple(g['betas']),
                'eps': g['eps'], 'initial_lr': g.get('initial_lr'), 'numel': n,
                'exp_avg': unpad(self.exp_avg), 'exp_avg_sq': unpad(self.exp_avg_sq)}

    def load_state_dict(self, state):
        assert state.get('format') == 'FusedClipAdam/1', 'not a FusedClipAdam state'
        n = [p.numel() for p in self.arena.params]
        assert list(state['numel']) == n, 'optimizer state belongs to a different parameter list'
        off = 0
        for o, k in zip(self.arena.offsets, n):
            self.exp_avg[o:o + k].copy_(state['exp_avg'][off:off + k])
            self.exp_avg_sq[o:o + k].copy_(state['exp_avg_sq'][off:off + k])
            off += k
        self.step_count = int(state['step_count'])
        g = self.param_groups[0]
        g['lr'], g['betas'], g['eps'] = float(state['lr']), tuple(state['betas']), float(state['eps'])
        if state.get('initial_lr') is not None:
            g['initial_lr'] = float(state['initial_lr'])

    def grad_norm(self):
        """pre-clip global L2 norm of the last step (device tensor; reading it syncs)"""
        return self.sumsq.sqrt() * self.grad_scale

    def clip_and_step(self, clip):
        a = self.arena
        if not a.holds_all_grads():
            a.gather_grads()
        g = self.param_groups[0]
        self.step_count += 1
        self._opt_called = True            # what torch's LRScheduler checks to order step() calls
        from . import functional as F_
        F_.mark('opt:start')
        st = stream_ptr()
        # one library call: global gradient norm + clipped Adam update (gradnorm_clip_adam_step).  The kernel also writes the
        # bf16 operand copy of the updated parameters (one more 55-MB stream in a 760-MB pass) -- the next forward then only
        # re-transposes the matrices instead of re-reading all 109 MB first
        call('ptv_gradnorm_clip_adam_step', ptr(self.flat_p), ptr(a.flat), ptr(self.exp_avg), ptr(self.exp_avg_sq), a.total,
             ptr(self.sumsq), float(self.grad_scale), float(clip if clip is not None else 0.0), float(g['lr']),
             float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), self.step_count, ptr(self.flat_p16) if ADAM_SHADOW else None, st)
        if ADAM_SHADOW:
            self._plain_stamp = self._stamp()
        self._throttle()

    # The host enqueues a step in 5.5 ms, the GPU runs it in 7: a loop that never synchronises runs further and further ahead, and every
    # block that crossed streams (record_stream) stays unusable until the GPU has passed it -- the allocator then grows by hipMalloc, a
    # device-wide synchronisation (36 of them in 120 steps, one ~60-ms stall every ~40 steps: round-5 probe, profiles/r05_allocator_stall.txt).  So the host waits for
    # the step before the previous one to finish before it enqueues the next: never a bubble (two whole steps stay queued), bounded memory.
    MAX_STEPS_IN_FLIGHT = 2
    # A FULL garbage collection walks every tracked Python object of the process (the module tree, the autograd closures, torch's own
    # registries): ~100 ms, and the first one falls around the 20th step -- measured as one 15-18 ms "step" average per 12-step round,
    # 3 runs of 3 (scripts/ab_step.py).  optim.freeze_gc() (explicit, process-wide, idempotent; optim.unfreeze_gc() undoes it) moves what
    # is alive to the permanent generation: later collections only walk what was created since.  The optimiser does NOT do that behind
    # the caller's back: `gc_freeze_at_step` (constructor) opts in, bench.py and TrainingInterface.train() call freeze_gc() themselves.
    GC_FREEZE_AT_STEP = None

    def _throttle(self):
        if torch.cuda.is_current_stream_capturing():
            return                                  # (inside a graph capture: no host waits, and no collection -- releasing blocks there kills it)
        if self.GC_FREEZE_AT_STEP is not None and self.step_count >= self.GC_FREEZE_AT_STEP:
            freeze_gc()
        if self.MAX_STEPS_IN_FLIGHT is None:
            return
        q = self.__dict__.setdefault('_inflight', [])
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        q.append(ev)
        if len(q) > self.MAX_STEPS_IN_FLIGHT:
            import time
            t0 = time.perf_counter()
            while len(q) > self.MAX_STEPS_IN_FLIGHT:
                q.pop(0).synchronize()
            self.throttle_wait_s = getattr(self, 'throttle_wait_s', 0.0) + (time.perf_counter() - t0)     # (bench.py: host time that is waiting, not work)

    def step(self, closure=None):
        assert closure is None
        self.clip_and_step(0.0)
