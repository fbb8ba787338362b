"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce of the flat gradient bucket
(27.3M fp32 = 109 MB over xGMI) in two parts per step.  Replaces the reference's nn.DataParallel
(amc_dl/torch_plus/module.py:67-68,152-159): samples are independent through forward/backward, so
the only exchange is the gradient SUM; the 1/world factor is folded into the clip+Adam kernel.

Gradients become final in a known order during the backward pass: the PianoTree decoder's node finishes first (32 % of the
27.3M parameters, at 7.4 of the 9.8 ms step), then the two encoders' bi-GRUs (53 %, at 8.6 ms), then the note-summary
GRU; what follows each of them is latency-bound BPTT that leaves the xGMI links idle.  So each of these slices of the bucket
starts its all-reduce the moment its last weight-gradient product is enqueued (functional.GRAD_READY_HOOK, on a communication
stream that waits for the producing streams), under the rest of the backward pass; `all_reduce_grads()` then reduces what is
left (heads, CNN, embedding, chord decoder: a few per cent) and waits for the early parts.
xGMI is point-to-point: a ring all-reduce is bound per link, so one large early message beats many small buckets here.
`backend='nccl'` is RCCL on ROCm; the CPU tests use gloo.  PTV_EARLY_ALLREDUCE=0: one all-reduce after the backward pass.

Contract of the early exchange (checked, not assumed): a slice leaves only if this step's backward holds the ONLY contribution
to its parameters -- each weight used once per step, one backward per zero_grad().  A second contribution to a slice that is on the
wire raises (GradArena.take); a backward that starts with gradients already present (accumulation) never starts an early part
(`p.grad is None` at the hook sites); zero_grad() waits for and drops early parts that no all_reduce_grads() consumed; and
all_reduce_grads() refuses early parts that were started under another zero_grad() epoch.  Every rank must issue the same
collectives in the same order: the hook fires from the same autograd nodes everywhere because the only data-dependent branch of
the step -- the teacher-forcing coins -- is drawn from ONE python `random` stream that all ranks seed alike (bench.py, train.py);
all_reduce_grads() cross-checks the ranges once per step when PTV_DP_CHECK=1."""
import os
import weakref

import torch
import torch.distributed as dist


def merge_ranges(ranges):
    """sorted, disjoint union of half-open (start, end) ranges; touching ranges fuse"""
    out = []
    for a, b in sorted(r for r in ranges if r[1] > r[0]):
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return [tuple(r) for r in out]


def complement_ranges(ranges, total):
    """what merge_ranges(ranges) leaves of [0, total)"""
    out, pos = [], 0
    for a, b in merge_ranges(ranges):
        if a > pos:
            out.append((pos, a))
        pos = max(pos, b)
    if pos < total:
        out.append((pos, total))
    return out


class GradSync:
    def __init__(self, model, optimizer=None, group=None):
        self.model = model
        self.optimizer = optimizer
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._bucket = None
        self._early = []                                     # [((start, end), work, arena epoch)] all-reduces started during the backward pass
        self.check = os.environ.get('PTV_DP_CHECK', '0') == '1'
        self._comm = None
        self.early = os.environ.get('PTV_EARLY_ALLREDUCE', '1') != '0'
        # PTV_DP_FORCE=1: run the exchange on a group of ONE rank too (tests: the RCCL / stream mechanics on a 1-GPU box)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get('PTV_DP_FORCE') == '1')
        # diagnosis of a multi-GPU run (bench.py --gpus N reports it): event-timed span of all_reduce_grads() on the caller's stream =
        # the part of the exchange the backward pass did NOT hide, and the byte counts of the early slices / the remainder
        self.timing = False
        self._spans = []
        self.last_early_bytes, self.last_remainder_bytes = [], 0
        # CUs the persistent recurrences leave to RCCL's channel kernels (csrc/gru_persist.hip sizes its grids to one 96-KB workgroup
        # per CU).  Default 0: a collective kernel co-resides with a persistent workgroup (64 KB of LDS and half the registers of a
        # CU stay free), and if it cannot, the part of a persistent grid that is not resident yet waits for the collective to finish
        # -- a stall bounded by the all-reduce itself (~1 ms), never a deadlock: the exchange does not depend on the launch it
        # delays (tests/test_gpu_zz_dist.py pins CUs under a step to show it).  > 0 halves the persistent grids (row groups are a
        # power of two): measured slower on one GPU, kept as the knob for a node where the stall shows up in `exposed_allreduce_ms`.
        self.cu_reserve = int(os.environ.get('PTV_PERSIST_CU_RESERVE', '0'))
        if self.active:
            if self.cu_reserve and torch.cuda.is_available():
                from . import functional as F_
                F_.set_persist_cu_reserve(self.cu_reserve)
            self.broadcast_parameters()
            if self.early and optimizer is not None and hasattr(optimizer, 'arena'):
                from . import functional as F_
                me = weakref.ref(self)
                F_.GRAD_READY_HOOK = lambda params, streams: (me() is not None) and me().grads_ready(params, streams)
                optimizer.arena.in_flight = lambda p: (me() is not None) and me()._in_flight(p)
                optimizer.arena.on_zero.append(lambda: (me() is not None) and me().drop_early())

    def _in_flight(self, p):
        a = self.optimizer.arena
        i = a._index.get(id(p))
        if i is None:
            return False
        lo, hi = a.offsets[i], a.offsets[i] + p.numel()
        return any(lo < e[0][1] and e[0][0] < hi for e in self._early)

    def drop_early(self):
        """zero_grad(): early parts nobody consumed (a backward without all_reduce_grads(), an aborted step) must finish before
        the bucket is cleared under them, and must not be mistaken for the next step's"""
        early, self._early = self._early, []
        for e in early:
            e[1].wait()
        if early and self._comm is not None:
            torch.cuda.current_stream().wait_stream(self._comm)

    def broadcast_parameters(self, src=0):
        """replica equality by construction, not by every rank happening to seed alike: rank `src`'s weights everywhere"""
        opt = self.optimizer
        if opt is not None and getattr(opt, 'flat_p', None) is not None:
            dist.broadcast(opt.flat_p, src, group=self.group)
            opt.mark_dirty()                                 # the bf16 operand shadows are re-cast before their next use
            return
        for p in self.model.parameters():
            dist.broadcast(p.data, src, group=self.group)

    def _flat_bucket(self):
        """(flat tensor holding every gradient, needs_scatter)"""
        opt = self.optimizer
        if opt is not None and hasattr(opt, 'arena'):
            if not opt.arena.holds_all_grads():
                opt.arena.gather_grads()
            return opt.arena.flat, False
        params = [p for p in self.model.parameters() if p.requires_grad]
        n = sum(p.numel() for p in params)
        if self._bucket is None or self._bucket.numel() != n:
            self._bucket = torch.empty(n, device=params[0].device, dtype=torch.float32)
        off = 0
        for p in params:
            k = p.numel()
            if p.grad is None:
                self._bucket[off:off + k].zero_()
            else:
                self._bucket[off:off + k].copy_(p.grad.reshape(-1))
            off += k
        return self._bucket, True

    def grads_ready(self, params, streams=()):
        """the gradients of `params` (registered with the optimiser's arena, adopted in place) are final once the work enqueued so
        far on the current stream and on `streams` has run: start their all-reduce now, on the communication stream"""
        opt = self.optimizer
        if not self.active or not self.early or opt is None or not hasattr(opt, 'arena'):
            return False
        a = opt.arena
        ranges = []
        for p in params:
            i = a._index.get(id(p))
            if i is None:
                continue
            ranges.append((a.offsets[i], a.offsets[i] + (p.numel() + 7) // 8 * 8))
        ranges = [r for r in merge_ranges(ranges) if not any(r[0] < e[0][1] and e[0][0] < r[1] for e in self._early)]
        if not ranges:
            return False
        if a.flat.is_cuda:
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=a.flat.device)
            self._comm.wait_stream(torch.cuda.current_stream())
            for s in streams:
                self._comm.wait_stream(s)
            with torch.cuda.stream(self._comm):
                for r in ranges:
                    self._early.append((r, dist.all_reduce(a.flat[r[0]:min(r[1], a.total)], op=dist.ReduceOp.SUM, group=self.group,
                                                           async_op=True), a.epoch))
        else:
            for r in ranges:
                self._early.append((r, dist.all_reduce(a.flat[r[0]:min(r[1], a.total)], op=dist.ReduceOp.SUM, group=self.group,
                                                       async_op=True), a.epoch))
        return True

    def all_reduce_grads(self):
        if not self.active:
            return
        early, self._early = self._early, []
        flat, scatter = self._flat_bucket()
        span = None
        if self.timing and flat.is_cuda:
            span = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            span[0].record()
        self.last_early_bytes = [4 * (min(e[0][1], flat.numel()) - e[0][0]) for e in early]
        self.last_remainder_bytes = 4 * flat.numel() - sum(self.last_early_bytes)
        try:
            return self._all_reduce(early, flat, scatter)
        finally:
            if span is not None:
                span[1].record()
                self._spans.append(span)

    def exchange_report(self):
        """{'exposed_allreduce_ms': mean GPU time the caller's stream spent inside all_reduce_grads() per step (waiting for the early
        parts + reducing the remainder), 'early_slices_bytes', 'remainder_bytes', ...}; synchronises"""
        spans, self._spans = self._spans, []
        if spans:
            torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in spans]
        return {'exposed_allreduce_ms': (sum(ms) / len(ms)) if ms else None, 'steps_timed': len(ms),
                'early_slices_bytes': list(self.last_early_bytes), 'remainder_bytes': int(self.last_remainder_bytes),
                'bucket_bytes': int(sum(self.last_early_bytes) + self.last_remainder_bytes), 'ranks': self.world,
                'backend': dist.get_backend(self.group) if dist.is_initialized() else None, 'early_exchange': bool(self.early),
                'persist_cu_reserve': self.cu_reserve}

    def _all_reduce(self, early, flat, scatter):
        if early:
            epoch = self.optimizer.arena.epoch
            if any(e[2] != epoch for e in early):
                for e in early:
                    e[1].wait()
                raise RuntimeError('GradSync: an early all-reduce from another zero_grad() epoch is still pending')
        if self.check:
            # every rank must have started the same early ranges (else the collectives below pair up wrongly): compare a digest
            sig = torch.tensor(([float(len(early))] + [float(v) for e in early for v in e[0]] + [0.0] * 32)[:33], dtype=torch.float64)
            lo, hi = sig.clone(), sig.clone()
            if flat.is_cuda and dist.get_backend(self.group) == 'nccl':
                lo, hi = lo.to(flat.device), hi.to(flat.device)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
            if not torch.equal(lo.cpu(), hi.cpu()):
                raise RuntimeError('GradSync: ranks started different early all-reduce ranges (%s here)' % [e[0] for e in early])
        if early and not scatter:
            # (every rank started the same early ranges in the same order: the hook fires from the same autograd node everywhere)
            for r in complement_ranges([e[0] for e in early], flat.numel()):
                dist.all_reduce(flat[r[0]:r[1]], op=dist.ReduceOp.SUM, group=self.group)
            for _, w, _e in early:
                w.wait()                                 # RCCL: the current stream waits; gloo: the host does
            if flat.is_cuda and self._comm is not None:
                torch.cuda.current_stream().wait_stream(self._comm)
        else:
            for _, w, _e in early:
                w.wait()
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        opt = self.optimizer
        if not scatter:
            opt.grad_scale = 1.0 / self.world            # folded into ptv_clip_adam_step
            return
        flat.div_(self.world)
        off = 0
        for p in (p for p in self.model.parameters() if p.requires_grad):
            k = p.numel()
            if p.grad is None:
                p.grad = flat[off:off + k].view_as(p).clone()
            else:
                p.grad.copy_(flat[off:off + k].view_as(p))
            off += k

    def mean_scalars(self, losses):
        """module.py:152-159 semantics: the reported loss is the mean of the replicas' scalars."""
        if not self.active:
            return losses
        t = torch.stack([l.detach().reshape(()) for l in losses])
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        t /= self.world
        return tuple(t.unbind(0))
