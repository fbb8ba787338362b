"""torch.autograd glue over the HIP library: one autograd.Function per block of the train step.

Every forward/backward body below is a sequence of C-ABI calls (`_lib.call`) on raw device
pointers; torch supplies memory, streams and the autograd graph only.  Internal activations are
STEP-MAJOR ([step][row][feature]) so that every recurrent step and every weight-gradient product
sees one contiguous matrix; API tensors are converted at the boundary (Transpose01).

Reference call sites are cited per function (paths are into /root/reference).
"""
import contextlib
import ctypes
import os
import time
import weakref

import torch

from . import _lib as _libmod
from ._lib import call, check, lib, prec_code, ptr, stream_ptr


# torch.cuda.current_stream() costs ~10 us of python (device-index resolution, lazy-init checks) and the step asks ~60 times
_get_cur = getattr(torch._C, '_cuda_getCurrentStream', None)
_get_dev = getattr(torch._C, '_cuda_getDevice', None)
_set_stream = getattr(torch._C, '_cuda_setStream', None)


def cur_stream():
    if _get_cur is None or _get_dev is None:
        return torch.cuda.current_stream()
    sid, di, dt = _get_cur(_get_dev())
    return torch.cuda.Stream(stream_id=sid, device_index=di, device_type=dt)

F32 = torch.float32
BF16 = torch.bfloat16
FUSED_DUR = True            # bf16 precision, H = 64: the 5-step duration GRU runs as one kernel (csrc/dur.hip)
BF16_STORAGE = True         # bf16 precision: tensors that only feed MFMA operands / epilogues live as bf16 in HBM
# bf16 precision: small-M recurrences (time GRU, encoder bi-GRUs; forward and BPTT) as ONE persistent launch per sequence
# (csrc/gru_persist.hip) wherever the shape fits one workgroup per CU.  Measured on one box, teacher-forced train step:
# B = 512: 28.7-30.1k samples/s with the per-step kernels, 31.7-32.9k persistent; B = 128: 14.5k -> 17.2k; B = 256: 23.1k -> 26.1k.
PERSIST = '1'            # '1' | '0'
# set while graph_step.GraphedTrainStep captures a WHOLE train step (forward + backward + optimiser) into one hipGraph: every fork is
# joined again before the capture ends, so the sibling-stream tricks and the persistent launches (ordered by captured event edges)
# stay on.  A capture of PART of a step (the decoder forward, ptvae.py) must not leave forks open and keeps them off.
WHOLE_STEP_CAPTURE = False


def capturing_part():
    """the current stream is being captured into a graph that does NOT span the whole step"""
    return (not WHOLE_STEP_CAPTURE) and torch.cuda.is_current_stream_capturing()


# Whole-step capture keeps the sibling streams, so its cross-stream edges must all be edges INSIDE the capture: waiting for a stream
# or an event that carries only work from before the capture is an error there (hipErrorStreamCaptureIsolation) -- and is not needed
# (the capture starts after a device-wide join).  Every wait of this module goes through these two helpers.
_CAPTURE_GEN = [0]          # bumped when a whole-step capture begins: events recorded before it are never waited for inside it


def stream_is_capturing(s):
    with torch.cuda.stream(s):
        return torch.cuda.is_current_stream_capturing()


_ORIGIN = [None]            # the capturing (origin) stream of a whole-step capture


def _via_origin(waiter, record):
    """torch 2.10's bundled HIP runtime (7.0) recurses forever in hipStreamEndCapture when two NON-origin streams of a capture wait
    for each other in both directions (a nested fork + join, the persistent launches' turn-taking chain); the origin stream may
    wait and be waited for freely (scripts/micro/graph_patterns.py).  So inside a whole-step capture an edge between two sibling
    streams is routed through the origin: origin waits for the source, the waiter waits for the origin."""
    o = _ORIGIN[0]
    record(o)                                                # origin <- source
    ev = torch.cuda.Event()
    ev.record(o)
    waiter.wait_event(ev)                                    # waiter <- origin


def wait_stream(waiter, waited):
    if WHOLE_STEP_CAPTURE:
        if not stream_is_capturing(waited):
            return                                           # nothing captured on it: no edge to add
        o = _ORIGIN[0]
        if o is not None and waiter != o and waited != o:
            return _via_origin(waiter, lambda org: org.wait_stream(waited))
    waiter.wait_stream(waited)


def record_event(stream=None):
    ev = torch.cuda.Event()
    stream = stream if stream is not None else cur_stream()
    ev.record(stream)
    ev.ptv_gen = _CAPTURE_GEN[0] if WHOLE_STEP_CAPTURE else -1
    ev.ptv_stream = stream
    return ev


def wait_event(stream, ev):
    if WHOLE_STEP_CAPTURE:
        if getattr(ev, 'ptv_gen', -1) != _CAPTURE_GEN[0]:
            return                                           # recorded before this capture began
        o = _ORIGIN[0]
        if o is not None and stream != o and getattr(ev, 'ptv_stream', o) != o:
            if ev.ptv_stream == stream:
                return                                       # same stream: ordered already
            return _via_origin(stream, lambda org: org.wait_event(ev))
    stream.wait_event(ev)


def join_captured_streams():
    """end of a whole-step capture: every sibling stream that was forked into the capture joins the capturing stream (a fork left
    open -- e.g. side work whose consumer did not run in this step -- would fail the capture)"""
    cur = cur_stream()
    for s in list(_CHILD_STREAMS.values()):
        if s != cur and stream_is_capturing(s):
            cur.wait_stream(s)


class whole_step_capture:
    """with whole_step_capture(): ... inside torch.cuda.graph(...): the module's stream helpers switch to capture-safe behaviour"""

    def __enter__(self):
        global WHOLE_STEP_CAPTURE
        WHOLE_STEP_CAPTURE = True
        _CAPTURE_GEN[0] += 1
        _PERSIST_LAST.clear()
        _ORIGIN[0] = None
        return self

    def origin(self):
        """call first thing inside torch.cuda.graph(): the current stream is the capture's origin"""
        _ORIGIN[0] = cur_stream()

    def __exit__(self, *exc):
        global WHOLE_STEP_CAPTURE
        WHOLE_STEP_CAPTURE = False
        _ORIGIN[0] = None
        return False


def _empty(*shape, dev, dtype=F32):
    return torch.empty(*shape, device=dev, dtype=dtype)


def _act_dtype(prec, H=8):
    """storage dtype of saved gates / gate gradients / input-side pre-activations"""
    return BF16 if (prec == 1 and BF16_STORAGE and H % 8 == 0) else F32


def _bf(t):
    return 1 if (t is not None and t.dtype == BF16) else 0


def _W(p, prec):
    """weight as an MFMA operand: its bf16 shadow in bf16 precision when the optimiser keeps one"""
    if prec == 1 and BF16_STORAGE and p.dim() >= 2 and p.shape[-1] % 8 == 0:
        from .optim import weight_shadow
        w = weight_shadow(p)
        if w is not None:
            return w
    return p


# Small zero-initialised buffers (reduction targets, counters, sync words: ~40 per step, 8 bytes to 2 KB each) are slices of a chunk
# that was cleared ONCE when it was allocated, handed out once and never reused -- one fill launch per 256 KB instead of one per
# buffer.  One bump chunk per (stream, dtype, fill value): a slice is requested on the stream whose earlier work cleared the chunk.
# Not inside a graph capture (the fill has to be a node of the graph) and not for anything over 16 KB
# (the persistent launches' sync words are 10 KB).
_ZCHUNK = {}
_ZCHUNK_BYTES = 256 * 1024


def _small_filled(n, dtype, value, dev):
    if n * 4 > 16384 or torch.cuda.is_current_stream_capturing():
        return torch.full((n,), value, device=dev, dtype=dtype)
    key = (stream_ptr(), dtype, value, dev.index if isinstance(dev, torch.device) else str(dev))
    ent = _ZCHUNK.get(key)
    n4 = (n + 3) // 4 * 4                                    # 16-byte aligned slices
    if ent is None or ent[1] + n4 > ent[0].numel():
        ent = _ZCHUNK[key] = [torch.full((_ZCHUNK_BYTES // 4,), value, device=dev, dtype=dtype), 0]
    out = ent[0][ent[1]:ent[1] + n]
    ent[1] += n4
    return out


def _zeros(*shape, dev):
    n = 1
    for d_ in shape:
        n *= d_
    if n * 4 <= 4096:
        return _small_filled(n, F32, 0, dev).view(*shape)
    return torch.zeros(*shape, device=dev, dtype=F32)


def _izeros(n, dev):
    return _small_filled(n, torch.int32, 0, dev)


def _ineg1(dev):
    """device int initialised to -1 (the `top` words the zero-skipping kernels raise with atomicMax)"""
    return _small_filled(1, torch.int32, -1, dev)


def _pad8(n):
    return (n + 7) // 8 * 8


def _row_dense(t):
    """t [..., C] is a [rows, C] matrix with one constant row stride >= C (contiguous, or rows padded)"""
    if t.dim() < 2 or t.stride(-1) != 1 or t.stride(-2) < t.shape[-1]:
        return False
    return all(t.stride(i) == t.stride(i + 1) * t.shape[i + 1] for i in range(t.dim() - 2))


def _rows2d(t):
    """[..., C] -> [rows, C] view when row-dense (keeps a padded row stride), else a contiguous copy"""
    if _row_dense(t):
        return t.as_strided((t.numel() // t.shape[-1], t.shape[-1]), (t.stride(-2), 1), t.storage_offset())
    return t.contiguous().view(-1, t.shape[-1])


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, (t.shape, t.stride())
    return t.stride(0)


CHAIN_PRIO = True
_SIDE_DEPTH = [0, 0]          # [nesting depth of Side calls on this thread, priority state last sent to the library]


def _forget_prio():
    _SIDE_DEPTH[1] = -1          # (unknown: _chain_prio sends the wanted state again)


if _forget_prio not in _libmod.ERROR_HOOKS:
    _libmod.ERROR_HOOKS.append(_forget_prio)


def _chain_prio():
    """products of a latency chain (not inside a Side call) raise their wave priority: tell the library when the state changes"""
    if CHAIN_PRIO:
        want = 1 if _SIDE_DEPTH[0] == 0 else 0
        if want != _SIDE_DEPTH[1]:
            lib().ptv_gemm_priority(want)
            _SIDE_DEPTH[1] = want


def gemm(a, b, out=None, *, ta=False, tb=False, bias=None, alpha=1.0, acc=False, act=0, prec=0, splitk=0, out_dtype=F32, m_top=None,
         m_unit=0, out_blocked=False):
    """out[M,N] = act(alpha * op(a) . op(b)^T + bias) (+ out);  see ptv_gemm in include/ptvae_hip.h.
    a / b / out may be bf16 tensors (bf16 precision only)."""
    M, K = (a.shape[1], a.shape[0]) if ta else (a.shape[0], a.shape[1])
    N, Kb = (b.shape[1], b.shape[0]) if tb else (b.shape[0], b.shape[1])
    assert K == Kb, (a.shape, b.shape, ta, tb)
    if out is None:
        assert not acc
        out = _empty(M, N, dev=a.device, dtype=out_dtype)
    assert tuple(out.shape) == (M, N), (out.shape, M, N)
    dt = _bf(a) | (_bf(b) << 1) | (_bf(out) << 2)
    if out_blocked:                                         # out holds the [M, N] result column-blocked by w = 32 (True) or 16: [N/w][M][w] (ptv_gemm dtypes bit 3 / 4)
        assert out_blocked in (True, 16, 32) and N % 32 == 0 and out.is_contiguous()
        dt |= 16 if out_blocked == 16 else 8
    _chain_prio()
    if m_top is not None:                                   # rows of a from (m_top + 1) * m_unit on are zero (device int)
        call('ptv_gemm_mtop', prec, int(ta), int(tb), M, N, K, ptr(a), _ld(a), ptr(b), _ld(b), ptr(out), _ld(out),
             ptr(bias), float(alpha), int(acc), int(act), int(-1 if _bf(out) else splitk), dt, ptr(m_top), int(m_unit), stream_ptr())
        return out
    call('ptv_gemm', prec, int(ta), int(tb), M, N, K, ptr(a), _ld(a), ptr(b), _ld(b), ptr(out), _ld(out),
         ptr(bias), float(alpha), int(acc), int(act), int(-1 if _bf(out) else splitk), dt, stream_ptr())
    return out


def copy2d(dst, src, *, alpha=1.0, acc=False, rows=None, cols=None, lds=None):
    rows = dst.shape[0] if rows is None else rows
    cols = dst.shape[1] if cols is None else cols
    call('ptv_copy2d', ptr(dst), _ld(dst), ptr(src), (_ld(src) if lds is None else lds), rows, cols,
         float(alpha), int(acc), stream_ptr())
    return dst


def colsum(out, a, sel=None, groups=1):
    """out[g, n] += sum_{rows with sel==g} a[row, n]"""
    call('ptv_colsum', ptr(out), ptr(a), _ld(a), a.shape[0], a.shape[1], ptr(sel), groups, _bf(a), stream_ptr())
    return out


def sum_steps(x3, out=None, acc=False, t_top=None):
    """out = sum over the leading axis; t_top (device int): the planes after it are known to be zero"""
    T = x3.shape[0]
    n = x3[0].numel()
    assert x3.is_contiguous()
    if out is None:
        out = _empty(*x3.shape[1:], dev=x3.device)
    call('ptv_sum_steps_top', ptr(out), ptr(x3), n, T, n, int(acc), _bf(x3), ptr(t_top), stream_ptr())
    return out


def transpose01(x):
    D0, D1 = x.shape[0], x.shape[1]
    W = x[0, 0].numel()
    x = x.contiguous()
    out = torch.empty((D1, D0) + tuple(x.shape[2:]), device=x.device, dtype=F32)
    call('ptv_transpose01', ptr(out), ptr(x), D0, D1, W, stream_ptr())
    return out


def _WT(p, prec):
    """transposed bf16 shadow [K_in, N_out] of a weight [N_out, K_in] (bf16 precision, if the optimiser keeps one)"""
    if prec == 1 and BF16_STORAGE and p.dim() == 2:
        from .optim import weight_shadow_t
        return weight_shadow_t(p)
    return None


def gemm_dx(dy, w, cols=None, out=None, *, acc=False, prec=0, m_top=None, m_unit=0, out_blocked=False):
    """input gradient of a Linear: dy . w[:, cols].  With a transposed bf16 shadow of w this is a K-contiguous
    (NT) product on bf16 weight tiles; otherwise the fp32 weight is read K-major."""
    wt = _WT(w, prec)
    if wt is not None:
        return gemm(dy, wt if cols is None else wt[cols, :], out, acc=acc, prec=prec, m_top=m_top, m_unit=m_unit, out_blocked=out_blocked)
    return gemm(dy, w if cols is None else w[:, cols], out, tb=True, acc=acc, prec=prec, m_top=m_top, m_unit=m_unit, out_blocked=out_blocked)


def _gru_flags(gates=None, gi=None, gi2=None, dg=None, w=None, ext=None):
    return _bf(gates) | (_bf(gi) << 1) | (_bf(gi2) << 2) | (_bf(dg) << 3) | (_bf(w) << 4) | (_bf(ext) << 6)


# ---------------------------------------------------------------------------------------------
# persistent weight-stationary recurrences (csrc/gru_persist.hip): one launch per sequence (or per pair of
# bi-GRU directions).  A persistent grid spins on arrival counters, so two of them must never be half-resident at
# the same time: every launch waits for the event of the previous one, whatever stream that ran on.
# ---------------------------------------------------------------------------------------------
_PERSIST_LAST = {}          # device index -> event recorded after the last persistent launch
# A backward node that is itself one C entry point made of other composites (functional_free.DecoderStepFn.backward ->
# ptv_decoder_free_bwd) COLLECTS its stages instead of launching them: while _DEFER is a list, every stage appends (kind, payload, thunk)
# -- payload = the C tables (and the tensors they point to, kept alive), thunk = the launch as it would have run here.  A stage that cannot
# go behind the C ABI flushes first (the thunks run, in order, and collecting stops): nothing is ever reordered.
_DEFER = None


def _defer_or_run(kind, payload, thunk):
    if _DEFER is not None:
        _DEFER.append((kind, payload, thunk))
        return 0
    return thunk()


def _defer_flush():
    """run what was collected so far, in order, and stop collecting"""
    global _DEFER
    items, _DEFER = _DEFER, None
    for kind, payload, thunk in items or ():
        rc = thunk()
        if isinstance(rc, int):
            check(rc, 'deferred ' + kind)


_PERSIST_SYNC = []          # (sync words, error index) of recent launches, for persist_check()
_CLUSTER_SYNC = []          # arrival counters + error word (last) of recent cluster-mode step loops (functional_free), for persist_check()
_PERSIST_OK = {}


def _parr(ts):
    return (ctypes.c_void_p * len(ts))(*[(t.data_ptr() if t is not None else None) for t in ts])


def _larr(vs):
    return (ctypes.c_long * len(vs))(*[int(v) for v in vs])


def _iarr(vs):
    return (ctypes.c_int * len(vs))(*[int(v) for v in vs])


# Persistent launches take turns (two spinning grids must never be half-resident together), so a short sequence on a sibling stream
# makes the long one next to it wait (the chord encoder's 8 steps in front of the texture encoder's 32).  Sending the short sequences
# to the per-step kernels instead (they co-reside with a persistent grid) measured SLOWER, 9.68 vs 9.57 ms: every supported sequence
# runs persistent.


def persist_supported(NC, M, H, T=None):
    if str(PERSIST).lower() in ('0', 'false', 'off') or capturing_part():
        return False
    key = (NC, M, H, torch.cuda.current_device())
    if key not in _PERSIST_OK:
        _PERSIST_OK[key] = bool(lib().ptv_gru_persist_supported(NC, M, H))
    return _PERSIST_OK[key]


def set_persist_cu_reserve(cus):
    """CUs the persistent grids leave free (ptv_gru_persist_cu_reserve; dist.GradSync sets it from PTV_PERSIST_CU_RESERVE)"""
    check(lib().ptv_gru_persist_cu_reserve(int(cus)), 'ptv_gru_persist_cu_reserve')
    _PERSIST_OK.clear()
    _SPLITK_OK.clear()


class _PersistTurn:
    """with _PersistTurn(): <one persistent launch on the current stream>"""

    def __enter__(self):
        self.cur = cur_stream()
        ev = _PERSIST_LAST.get(self.cur.device.index)
        if ev is not None:
            wait_event(self.cur, ev)
        return self

    def __exit__(self, *exc):
        _PERSIST_LAST[self.cur.device.index] = record_event(self.cur)
        return False


def _persist_sync(NC, dev):
    """zeroed sync words of one launch: word 0 = error flag, word 16*(1+g) = arrival counter of row group g"""
    sync = _izeros(16 * (33 + 128), dev)       # error word, 32 row-group counters, 128 team counters (split-K BPTT)
    _PERSIST_SYNC.append(sync)
    if len(_PERSIST_SYNC) > 64:
        del _PERSIST_SYNC[:32]
    return sync


ORDERED_STRICT = os.environ.get('PTV_ORDERED_STRICT', os.environ.get('PTV_PTR_CHECKS', '0')) == '1'      # (on in the test suite)


def ordered_fallbacks(reset=False):
    """reductions that ran on fp32 atomics although the ordered (bit-reproducible) mode is on -- 0 after any step, or the step's last
    bits depend on arrival order (ptv_ordered_fallbacks; asserted by the determinism tests and by GraphedTrainStep after capture)"""
    return int(lib().ptv_ordered_fallbacks(int(bool(reset))))


def persist_check():
    """raises if a bounded spin of a recent persistent launch gave up (synchronises: tests / the end of a bench)"""
    bad = [i for i, sw in enumerate(_PERSIST_SYNC) if int(sw[0].item()) != 0]
    del _PERSIST_SYNC[:]
    badc = [i for i, sw in enumerate(_CLUSTER_SYNC) if int(sw[-1].item()) != 0]
    del _CLUSTER_SYNC[:]
    if bad:
        raise RuntimeError('persistent GRU launch gave up waiting for its row group (recent launches %s)' % bad)
    if badc:
        raise RuntimeError('cluster-mode note loop: a member gave up waiting for its panel (recent forward passes %s)' % badc)


def gru_persist_fwd(M, H, T, chains):
    """chains: dicts with gi (bf16 [T,M,3H] view), gi_step, gi_ld, gi2, gi2_step, gi2_ld, w16 (bf16 [3H,H]), b_hh, hall, hall16,
    gates (bf16 or None), lengths, reverse.  One launch for all of them."""
    NC = len(chains)
    g = lambda k: [c.get(k) for c in chains]
    dev = chains[0]['hall'].device
    sync = _persist_sync(NC, dev)
    xch = [torch.empty((T + 1) * M * H, device=dev, dtype=BF16) for _ in chains]     # exchanged operand, K-blocked
    with _PersistTurn():
        rc = lib().ptv_gru_persist_fwd(NC, M, H, T, _parr(g('gi')), _larr(g('gi_step')), _larr(g('gi_ld')),
                                       _parr(g('gi2')), _larr([c.get('gi2_step', 0) for c in chains]),
                                       _larr([c.get('gi2_ld', 0) for c in chains]), _parr(g('w16')), _parr(g('b_hh')),
                                       _parr(g('hall')), _parr(g('hall16')), _parr(g('gates')), _parr(g('lengths')),
                                       _iarr([int(bool(c.get('reverse'))) for c in chains]), _parr(xch), ptr(sync), stream_ptr())
    check(rc, 'ptv_gru_persist_fwd')


# BPTT of the persistent recurrences: split-K teams of S workgroups (csrc/gru_persist.hip, pgru_bwd_sk_kernel) read 1/S of the
# exchanged operand per step; 0 = the round-2 kernel (every workgroup reads all of K).  Measured (scripts/bench_persist.py,
# gpurun_out/r04_bench_persist_a.txt; us per launch, S = 0 / 2 / 4): time GRU M = 512 T = 32: 429 / 402 / 382; one encoder's two
# chains M = 512 T = 8: 213 / 172 / 178; M = 1024 T = 32: 770 / 633 / 645; M = 256: 253 / 295 / 271; M = 128: 220 / 271 / 240;
# H = 512 M = 512: 53 / 61 / 60 -- the second hand-off of a step costs what the smaller read saves unless the read is large.  In the
# B = 512 step (scripts/ab_step.py, same process): 9.242 / 9.173 / 9.326 ms.  'auto' = 2 where it wins (H = 1024, M >= 512), else 0.
PERSIST_SPLITK = os.environ.get('PTV_PERSIST_SPLITK', 'auto')
PERSIST_SPLITK = PERSIST_SPLITK if PERSIST_SPLITK == 'auto' else int(PERSIST_SPLITK)
_SPLITK_OK = {}


def persist_splitk(NC, M, H):
    """S of the split-K BPTT for this shape, or 0"""
    S = PERSIST_SPLITK
    if S == 'auto':
        S = 2 if (H >= 1024 and M >= 512) else 0
        if not S and not persist_supported(NC, M, H):
            S = 2                                  # only the split-K kernel takes more than 256 rows per workgroup
    if S not in (2, 4):
        return 0
    key = (NC, M, H, S, torch.cuda.current_device())
    if key not in _SPLITK_OK:
        _SPLITK_OK[key] = bool(lib().ptv_gru_persist_splitk_supported(NC, M, H, S))
    return S if _SPLITK_OK[key] else 0


def gru_persist_bwd(M, H, T, chains):
    """chains: dicts with hall, gates, wt16 (bf16 W_hh^T [H,3H]), dh_ext ([T,M,H] fp32/bf16 view or None), dh_last, dgi, dgh,
    dh0 (or None), reverse"""
    NC = len(chains)
    g = lambda k: [c.get(k) for c in chains]
    ext = g('dh_ext')
    last = g('dh_last')
    dev = chains[0]['hall'].device
    sync = _persist_sync(NC, dev)
    xch = [torch.empty(T * M * 3 * H, device=dev, dtype=BF16) for _ in chains]
    S = persist_splitk(NC, M, H)
    args = (NC, M, H, T, _parr(g('hall')), _parr(g('gates')), _parr(g('wt16')),
            _parr(ext), _larr([e.stride(0) if e is not None else 0 for e in ext]),
            _larr([e.stride(1) if e is not None else 0 for e in ext]),
            _iarr([_bf(e) for e in ext]),
            _parr(last), _larr([l.stride(0) if l is not None else 0 for l in last]),
            _parr(g('dgi')), _parr(g('dgh')), _parr(g('dh0')),
            _iarr([int(bool(c.get('reverse'))) for c in chains]), _parr(xch))
    with _PersistTurn():
        if S:
            n = lib().ptv_gru_persist_part_elems(NC, M, H, S)
            part = [torch.empty(n, device=dev) for _ in chains]
            rc = lib().ptv_gru_persist_bwd_splitk(S, *args, _parr(part), ptr(sync), stream_ptr())
        else:
            rc = lib().ptv_gru_persist_bwd(*args, ptr(sync), stream_ptr())
    check(rc, 'ptv_gru_persist_bwd')


def _persist_fwd_ok(prec, M, H, T, gi, gi2, gates, gi_idx, hall16, w16, NC=1):
    return (prec == 1 and T >= 2 and hall16 is not None and gi.dtype == BF16 and (gi2 is None or gi2.dtype == BF16)
            and (gates is None or gates.dtype == BF16) and gi_idx is None and w16 is not None and w16.dtype == BF16
            and persist_supported(NC, M, H, T))


def gru_fwd(prec, gi, gi_step, gi_ld, w_hh, b_hh, hall, gates, *, gi2=None, gi2_step=0, gi2_ld=0, lengths=None,
            reverse=False, gi_idx=None, T=None, hall16=None, skip_cast0=False):
    T1, M, H = hall.shape
    T = T1 - 1 if T is None else T
    if hall16 is not None:
        w_hh = _W(w_hh, prec)
    if not skip_cast0 and _persist_fwd_ok(prec, M, H, T, gi, gi2, gates, gi_idx, hall16, w_hh):
        return gru_persist_fwd(M, H, T, [dict(gi=gi, gi_step=gi_step, gi_ld=gi_ld, gi2=gi2, gi2_step=gi2_step, gi2_ld=gi2_ld,
                                              w16=w_hh, b_hh=b_hh, hall=hall, hall16=hall16, gates=gates, lengths=lengths,
                                              reverse=reverse)])
    call('ptv_gru_seq_fwd', prec, M, H, T, ptr(gi), gi_step, gi_ld, ptr(gi2), gi2_step, gi2_ld, ptr(w_hh),
         ptr(b_hh), ptr(hall), ptr(hall16), ptr(gates), ptr(lengths), int(reverse), ptr(gi_idx),
         _gru_flags(gates, gi, gi2, w=w_hh) | (32 if skip_cast0 else 0), stream_ptr())


def _hall16(prec, T1, M, H, dev):
    """bf16 shadow of a GRU state buffer (bf16 precision): MFMA operand of every later product on it"""
    return _empty(T1, M, H, dev=dev, dtype=BF16) if _act_dtype(prec, H) == BF16 else None


def gru_bwd(prec, hall, gates, w_hh, *, dh_ext=None, dh_last=None, lr=None, reverse=False, need_dh0=True, allow_persist=True):
    """-> dgi [T,M,3H] (time order), dgh [T,M,3H] (processing order), dh0 [M,H] or None"""
    T1, M, H = hall.shape
    T = T1 - 1
    dev = hall.device
    dt = _act_dtype(prec, H)
    dgi = _empty(T, M, 3 * H, dev=dev, dtype=dt)
    dgh = _empty(T, M, 3 * H, dev=dev, dtype=dt)
    dhz = _empty(2, M, H, dev=dev)
    dh0 = _empty(M, H, dev=dev) if need_dh0 else None
    ext = (ptr(dh_ext), dh_ext.stride(0), dh_ext.stride(1)) if dh_ext is not None else (None, 0, 0)
    last = (ptr(dh_last), dh_last.stride(0)) if dh_last is not None else (None, 0)
    lra = (ptr(lr[0]), lr[1], lr[2], lr[3], ptr(lr[4])) if lr is not None else (None, 0, 0, 0, None)
    if dt == BF16:
        wt = _WT(w_hh, prec)                  # W_hh^T [H,3H] bf16: K-contiguous weight tiles for the BPTT products
        if (allow_persist and wt is not None and lr is None and T >= 2 and gates.dtype == BF16 and persist_supported(1, M, H, T)
                and (dh_ext is None or dh_ext.stride(2) == 1)):
            gru_persist_bwd(M, H, T, [dict(hall=hall, gates=gates, wt16=wt, dh_ext=dh_ext, dh_last=dh_last, dgi=dgi, dgh=dgh,
                                           dh0=dh0, reverse=reverse)])
            return dgi, dgh, dh0
        w_hh = wt if wt is not None else w_hh
    call('ptv_gru_seq_bwd', prec, M, H, T, ptr(hall), ptr(gates), ptr(w_hh), *ext, *last, *lra, ptr(dgi),
         ptr(dgh), ptr(dhz), ptr(dh0), int(reverse), _gru_flags(gates, dg=dgi, w=w_hh, ext=dh_ext), stream_ptr())
    return dgi, dgh, dh0


# ---------------------------------------------------------------------------------------------
# HIP-stream fork/join: independent kernel chains (the two directions of a bi-GRU, the two encoders,
# weight-gradient products vs. the BPTT chain) run on sibling streams so the small, latency-bound
# step kernels overlap with the big GEMMs.  Streams are cached per (parent stream, slot).
# ---------------------------------------------------------------------------------------------
_CHILD_STREAMS = {}
OVERLAP = True              # set False to serialise everything on the caller's stream (debugging)


def _record_stream(obj, stream):
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _record_stream(o, stream)
    elif isinstance(obj, dict):
        for o in obj.values():
            _record_stream(o, stream)
    elif hasattr(obj, '__dict__') and not callable(obj):          # small result holders (e.g. HipNormal)
        for o in vars(obj).values():
            if isinstance(o, (torch.Tensor, list, tuple, dict)):
                _record_stream(o, stream)


# (HIP stream priorities on the pool streams -- the command processor hands free workgroup slots to the highest-priority queue -- measured
# 13.3-19.2 ms per step for ANY non-default value, the same cliff as a fifth hardware queue: every pool stream is a default-priority stream.)


# pool stream of slot % 4.  Merging streams measured at the end of round 5 (scripts/ab_combo.py, 6 interleaved rounds): the note summaries
# and the texture encoder on ONE stream ((0, 2, 2, 3) or (0, 1, 1, 3): four streams with the step's own, one per hardware queue) 7.04-7.12 ms
# per step, the same as five (7.05-7.15); the chord decoder's stream merged into another one 7.57-7.60.
POOL_MAP = (0, 1, 2, 3)


class Side:
    """s = Side(slot); s(fn, *keep_alive) runs fn on the sibling stream after everything enqueued so far
    on the parent; s.join() makes the parent wait for it.  `keep_alive` tensors (or containers of tensors) stay referenced
    until join so the caching allocator cannot hand their memory to later parent-stream work while the sibling's kernels
    are still queued: EVERYTHING fn reads that might be released before the join belongs there -- with defer() that includes
    the node's saved forward state, which the autograd engine drops the moment the node returns.  (Never the gradient buffers
    fn writes: a second reference makes AccumulateGrad clone them -- on its own stream, before the sibling has run --
    instead of adopting them.)"""

    def __init__(self, slot=0, chain=False):
        self.main = cur_stream()
        self.chain = chain          # the call IS a latency chain of the step (the encoders at its head): its products keep the raised wave priority
        # The sibling streams are folded onto a pool of 4 (slot mod 4): the HIP runtime multiplexes all streams of a process onto 4
        # hardware queues anyway (GPU_MAX_HW_QUEUES; with 5 or more the step gets 40 % SLOWER), and which of ~10 private streams
        # end up sharing a queue -- i.e. silently serialise -- is then decided by creation order.  With the pool the sharing is
        # explicit; measured 9.67 vs 9.94 ms per step (pool sizes 2 / 3 / 5 / 7: 10.0 / 9.8 / 10.3 / 10.3).
        key = ('pool', self.main.device.index, POOL_MAP[slot % 4])
        if key not in _CHILD_STREAMS:
            # all pool streams at once, in a FIXED order: which hardware queue a stream lands on follows creation order, and
            # creating them lazily in first-use order made the step time depend on which slot happened to be used first (0.4 ms)
            for k in (1, 2, 0, 3):
                _CHILD_STREAMS.setdefault(('pool', self.main.device.index, k), torch.cuda.Stream(device=self.main.device))
        self.s = _CHILD_STREAMS[key]
        self.keep = []
        self.used = False

    def __call__(self, fn, *keep, after=None):
        """after: an event -- the sibling waits for IT instead of for everything the parent has queued so far"""
        if not OVERLAP:
            return fn()
        if after is not None:
            wait_event(self.s, after)
        else:
            wait_stream(self.s, self.main)
        self.used = True
        self.keep.extend(keep)
        depth = 0 if self.chain else 1
        _SIDE_DEPTH[0] += depth
        _chain_prio()               # (kernels other than the products read the same library state: csrc/notes_persist.hip)
        try:
            # (torch.cuda.stream() is ~15 us of python per use; the parent stream is known)
            if _set_stream is not None:
                # (the stream current NOW is restored afterwards: a handle may be called under another stream than it was built under)
                if _get_cur is not None and _get_dev is not None:
                    psid, pdi, pdt = _get_cur(_get_dev())
                else:
                    pm = torch.cuda.current_stream()
                    psid, pdi, pdt = pm.stream_id, pm.device_index, pm.device_type
                _set_stream(stream_id=self.s.stream_id, device_index=self.s.device_index, device_type=self.s.device_type)
                try:
                    r = fn()
                finally:
                    _set_stream(stream_id=psid, device_index=pdi, device_type=pdt)
            else:
                with torch.cuda.stream(self.s):
                    r = fn()
        finally:
            _SIDE_DEPTH[0] -= depth
            _chain_prio()
        # What fn returns was allocated under the sibling stream and will be read on the parent after join(): tell the caching
        # allocator, or the block goes back to the SIBLING's pool the moment Python drops the tensor and the next allocation there
        # may overwrite it while the parent's reader is still queued (seen as partly wrong dx / gradients once in ~10 runs).
        if self.s != self.main:
            _record_stream(r, self.main)
        return r

    def join(self):
        if self.used:
            wait_stream(self.main, self.s)
        self.keep.clear()
        self.used = False

    def defer(self):
        """join at the END of the running backward pass instead of now: for side work that only produces parameter
        gradients (nothing later in the backward reads them), so it can overlap with whatever the autograd engine
        runs next -- the decoder's weight-gradient products then run under the encoders' latency-bound BPTT chains.
        Outside a backward pass this is join()."""
        if not self.used:
            return self.join()
        try:
            if not _DEFERRED:
                torch.autograd.Variable._execution_engine.queue_callback(_join_deferred)
        except RuntimeError:                      # not inside the autograd engine
            return self.join()
        _DEFERRED.append((self.s, list(self.keep)))
        self.keep.clear()
        self.used = False


_DEFERRED = []
GRAD_READY_HOOK = None      # callable(params, streams) set by dist.GradSync: the gradients of `params` are complete once `streams` drain


TRACE = None          # set to a list by scripts/trace_marks.py: (name, event on the current stream) markers of the backward chain


def mark(name):
    if TRACE is not None:
        e = torch.cuda.Event(enable_timing=True)
        e.record(cur_stream())
        TRACE.append((name, e, time.perf_counter()))


def _join_deferred():
    mark('deferred:join_start')
    cur = cur_stream()
    for s, _keep in _DEFERRED:
        wait_stream(cur, s)
    _DEFERRED.clear()
    mark('deferred:joined')


def reset_deferred():
    """join whatever deferred side-stream work is still registered (a backward pass that raised never ran its end-of-pass
    callback): called by FusedClipAdam.zero_grad() and by the first node of every backward pass (VaeLossFn)"""
    _LOSS_TOP.clear()                       # (a zero-skip bound nobody consumed: e.g. torch.autograd.grad that stopped at the logits)
    if _DEFERRED:
        _join_deferred()


def _gbuf(p):
    """zero-initialised gradient buffer for parameter p (weight-gradient kernels accumulate into it):
    a view of the optimiser's flat gradient bucket when one is registered (optim.GradArena)."""
    from .optim import grad_buffer
    return grad_buffer(p)


def _bgrad(b, a):
    """bias-style gradient: column sums of a [rows, N] into the gradient buffer of parameter b [N]"""
    g = _gbuf(b)
    N = a.shape[1]
    if N < 4 and 64 % N == 0 and a.dtype == F32 and a.is_contiguous() and a.numel() % 64 == 0 and a.shape[0] >= 4096:
        # a very narrow matrix (dur_out_linear's bias: [5 M, 2]) leaves 15 of the column kernel's 16 lanes idle on 4-byte loads -- 115 us
        # teacher-forced, 650 us in the free-running step for 10 MB: summed as [rows * N / 64, 64] with 16-byte loads, then folded 64 -> N
        tmp = colsum(_zeros(1, 64, dev=a.device), a.view(-1, 64))
        colsum(g.view(1, -1), tmp.view(64 // N, N))
        return g
    colsum(g.view(1, -1), a)
    return g


WGRAD_FUSE_BIAS = True
# per-row dead work (round 6): inside loss() the teacher-forced decoder works on its rows (t, b) in the order of descending number of live
# note steps and passes over the (note step, 64-row panel) pairs that hold no target; PTV_SORT_DEC_ROWS=0 = rows in (t, b) order, only
# the batch-wide limit (PTV_DEAD_STEPS).  Needs both decoder composites (it is implemented behind the C ABI only).
SORT_DEC_ROWS = os.environ.get('PTV_SORT_DEC_ROWS', '1') != '0'
# ... stage 2: the weight-gradient products over (note step, sorted row) skip the dead 128-row blocks of every note step too (K segments of
# ptv_wgrad_batch); PTV_WGRAD_SEG=0 = they multiply the zero rows (bit-identical results)
WGRAD_SEG = os.environ.get('PTV_WGRAD_SEG', '1') != '0'
_LAST_SEG_N = None
WGRAD_BATCH = 3            # ptv_wgrad_batch_mode (scripts/ab_step.py): 0 = products one by one, 1 = one launch, 2 = single launches + one reduction, 3 = small ones batched


def _apply_switches():
    """hand module-level switches that live in the library to it (scripts/ab_step.py calls this after every setattr)"""
    lib().ptv_wgrad_batch_mode(int(WGRAD_BATCH))


EMBED_MH_FWD = True    # multi-hot operand of the note_embedding gradient built during the forward
# decoder backward: fork the weight-gradient work BEFORE the chain queues its next dX products (no false dependency on them)?  Measured
# 9.43 vs 9.37 ms: the products then compete with the chain's own dX products for the CUs -- the later fork is the better schedule
# stream slot of a bi-GRU's second direction, forward / backward.  In the backward slot 7 = pool stream 3 is also the stream of the decoder's
# deferred weight-gradient products (Side(3)): until round 5 the reversed directions of the note-summary and encoder BPTTs queued behind
# them there (moving them measured SLOWER in round 4: slot 8 / 6 / 5: 9.40 / 9.39 / 9.29 vs 9.09 ms).  Measured again at the end of round 5
# -- shorter forward, shorter host, the chord encoder itself on stream 3 (model.CHD_ENC_SLOT) -- slot 4 (pool stream 0, the chord decoder's,
# idle in the tail) wins: 7.32-7.43 against 7.49-7.68 ms alone, 6.97-7.15 with the encoder move; slots 5 / 6 the same within noise.
BIGRU_SLOT = 7
BIGRU_SLOT_BWD = 4
DEC_WGRAD_SLOT = 3      # pool stream of the decoder's deferred weight-gradient products
EMBED_MH_SLOT = 7       # pool stream on which the embedding's multi-hot operand is built during the forward
SHADOW_T_SLOT = 7       # pool stream of the transposed bf16 weight shadows' refresh (optim.refresh_weight_shadows)
# note-summary bi-GRU: panels of rows sorted by length (ptv_rows_by_length + the *_perm entry points).  Measured, round 5 (profiles/
# r05_ab_runs.txt): the launches do half the work (mean length 3.8 against a panel maximum of 8) but stay as long as their longest panel --
# 225 / 242 us against 210 / 233 us in situ, step 7.69-7.71 against 7.64-7.67 ms: they are latency-bound per step, and what they leave
# free nobody needs at that moment.  Off; the capability stays (it is the sequence packing of pack_padded_sequence, ptvae.py:446-453).
SORT_ROWS = os.environ.get('PTV_SORT_ROWS', '1') == '1'
# (Round-4 scheduling experiments on the step's tail -- chain-first bi-GRU backward, parameter-gradient products launched when the backward
# pass ends, row kernels taking turns with the persistent launches, forks before / after the chain's dX products -- all measured slower
# than this plain scheme, 8.43 ms per step against 8.48-8.95; their numbers are in DESIGN.md section 4 and profiles/r04_ab_*.txt, their
# code paths were removed in round 5.)
DP_INPLACE = True        # decoder backward accumulates into the loss node's dpitch buffer (no 134-MB copy)
# the backward passes over work whose result is exactly zero: note steps / tiles at which no gradient arrives (the loss ignores the
# padded note slots), panel steps beyond the longest packed note sequence.  Decided on the gradients / lengths themselves, so the
# results do not change; PTV_ZERO_SKIP=0 runs everything dense (bench.py reports that figure next to the headline)
ZERO_SKIP = os.environ.get('PTV_ZERO_SKIP', '1') != '0'
_ZERO_SKIP_SET = []


def zero_skip_sync():
    """hand the current ZERO_SKIP to the library (cheap; called by the decoder backward and the bi-GRU forward)"""
    if not _ZERO_SKIP_SET or _ZERO_SKIP_SET[0] != ZERO_SKIP:
        lib().ptv_zero_skip(int(ZERO_SKIP))
        _ZERO_SKIP_SET[:] = [ZERO_SKIP]


def wgrad_bias(dy, x, gw, gb, prec, k_top=None, k_unit=0, k_rev=0, seg=None, seg_period=0):
    """gw [N_out, N_in] += dy^T . x and gb [N_out] (or None) += column sums of dy: a layer's weight and bias gradient in one pass
    over dy (ptv_wgrad's colsum_a) where the weight-gradient kernel applies; otherwise the product and a column-sum kernel.
    k_top (device int) / k_unit: the rows of dy from (k_top + 1) * k_unit on are zero (ptv_wgrad); seg (device ints) / seg_period: K segments
    of k_unit rows (ptv_wgrad_job.seg_n)"""
    K = dy.shape[0]
    if (WGRAD_FUSE_BIAS and prec == 1 and K >= 512 and gw.dtype == F32 and dy.stride(1) == 1 and x.stride(1) == 1
            and True):
        _chain_prio()
        if seg is not None:                                  # (the job form carries the segments; same bits as the composite's batch of two)
            from ._lib import wgrad_batch
            wgrad_batch([dict(M=dy.shape[1], N=x.shape[1], K=K, A=dy, B=x, C=gw, colsum_a=gb, k_top=k_top, k_unit=int(k_unit) if k_top is not None else 0,
                              k_rev=int(k_rev), seg_n=seg, seg_unit=int(k_unit), seg_period=int(seg_period))])
            return gw, gb
        call('ptv_wgrad', dy.shape[1], x.shape[1], K, ptr(dy), _ld(dy), ptr(x), _ld(x), ptr(gw), _ld(gw), 1.0, 1,
             _bf(dy) | (_bf(x) << 1), 0, ptr(gb), ptr(k_top), int(k_unit) if k_top is not None else 0, int(k_rev), stream_ptr())
    else:
        gemm(dy, x, gw, ta=True, tb=True, acc=True, prec=prec)
        if gb is not None:
            colsum(gb.view(1, -1), dy)
    return gw, gb


def wgrad_bias_batch(items, prec):
    """[(dy, x, gw, gb or None), ...] -> every gw += dy^T . x, gb += column sums of dy, as ONE ptv_wgrad_batch call (one product launch + one
    reduction launch) when the weight-gradient kernel takes them all; the same bits as wgrad_bias() item by item"""
    ok = WGRAD_FUSE_BIAS and prec == 1 and all(dy.shape[0] >= 512 and gw.dtype == F32 and dy.stride(1) == 1 and x.stride(1) == 1
                                               for dy, x, gw, gb in items)
    if not ok or len(items) < 2:
        for dy, x, gw, gb in items:
            wgrad_bias(dy, x, gw, gb, prec)
        return
    from ._lib import wgrad_batch
    _chain_prio()
    wgrad_batch([dict(M=dy.shape[1], N=x.shape[1], K=dy.shape[0], A=dy, B=x, C=gw, colsum_a=gb) for dy, x, gw, gb in items])


# The fused duration GRU does not write its gate planes (5 steps x 4 planes x [M, 64] bf16 = 629 MB at B = 512, 63 % of the forward
# kernel's writes); its backward rebuilds them from the states it reads anyway (csrc/dur_bwd.hip, recompute mode).  0 = save them.
DUR_RECOMPUTE = os.environ.get('PTV_DUR_RECOMPUTE', '1') != '0'


def dur_bwd_fusable(prec, Hd, gates_d):
    return prec == 1 and Hd == 64 and FUSED_DUR and (gates_d is None or gates_d.dtype == BF16)


def dur_bwd_fused(P, G, gates_d, idx, HD, HDo, ddur, wgrad, bgrad, side, tabs=None):
    """backward of the 5-step duration GRU as one kernel (csrc/dur_bwd.hip) -> dHD0 [M, Hd]; the parameter gradients
    come back as per-block partials that a column sum + ptv_dur_bwd_finalize fold into G on a side stream.
    gates_d None: the forward did not save its gates; tabs = (tab0, tab) it was given"""
    _, M, Hd = HD.shape
    dev = HD.device
    nblk = min(256, (M + 63) // 64)
    psz = lib().ptv_dur_gru_bwd_part_size()
    part = _empty(nblk, psz, dev=dev)
    dHD0 = _empty(M, Hd, dev=dev)
    hsrc = HDo if HDo.dtype == BF16 else HD                 # bf16 state copies when the forward kept them (half the reads)
    rc = (ptr(P['dec_dur_gru.bias_hh_l0']), ptr(tabs[0]), ptr(tabs[1])) if gates_d is None else (None, None, None)
    call('ptv_dur_gru_bwd', Hd, M, ptr(gates_d), M * Hd, 4 * M * Hd, ptr(hsrc), M * Hd, int(hsrc.dtype == BF16), ptr(ddur), 10,
         ptr(P['dec_dur_gru.weight_hh_l0']), ptr(P['dur_out_linear.weight']), ptr(idx), M, ptr(dHD0), ptr(part), nblk, *rc,
         stream_ptr())

    def dur_wgrads():
        if HDo.dtype == BF16:                     # one pass over the 5 bf16 state planes instead of 5 split-K products
            G['dur_out_linear.weight'] = _gbuf(P['dur_out_linear.weight'])
            call('ptv_dur_out_wgrad', ptr(ddur), 10, ptr(HDo), M * Hd, ptr(G['dur_out_linear.weight']), M, Hd, stream_ptr())
        else:
            for d in range(5):
                wgrad('dur_out_linear.weight', ddur[:, 2 * d:2 * d + 2], HDo[d + 1])
        bgrad('dur_out_linear.bias', ddur.view(M * 5, 2))
        S = colsum(_zeros(1, psz, dev=dev), part)
        names = ('dec_dur_gru.weight_hh_l0', 'dec_dur_gru.bias_hh_l0', 'dec_dur_gru.bias_ih_l0', 'dec_dur_gru.weight_ih_l0',
                 'dur_sos_token')
        for name in names:
            G[name] = _gbuf(P[name])
        call('ptv_dur_bwd_finalize', ptr(S), *[ptr(G[n]) for n in names], ptr(P['dec_dur_gru.weight_ih_l0']),
             ptr(P['dur_sos_token']), P['dur_sos_token'].numel(), stream_ptr())
    side(dur_wgrads, ddur, part)
    return dHD0



def _as2d(t):
    return t.view(1, -1) if t.dim() == 1 else t


# =============================================================================================
# nn.Linear  (x . W^T + b)
# =============================================================================================
class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2, w, b, prec):
        y = gemm(x2, _W(w, prec), bias=b, prec=prec)
        ctx.save_for_backward(x2, w, b)
        ctx.prec = prec
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, w, b = ctx.saved_tensors
        prec = ctx.prec
        dy = dy.contiguous()
        dx = gemm_dx(dy, w, prec=prec) if ctx.needs_input_grad[0] else None
        if b is not None:
            dw, db = wgrad_bias(dy, x2, _gbuf(w), _gbuf(b), prec)
        else:
            dw, db = gemm(dy, x2, _gbuf(w), ta=True, tb=True, acc=True, prec=prec), None
        return dx, dw, db, None


# =============================================================================================
# Transpose01: batch-major API tensors <-> step-major internals
# =============================================================================================
class Transpose01Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return transpose01(x)

    @staticmethod
    def backward(ctx, dy):
        return transpose01(dy)


# =============================================================================================
# PtvaeDecoder.emb_x  (ptvae.py:531-535)
# =============================================================================================
DEFAULT_GEOM = (32, 16, 130, 5, 130)                    # (num_step, max_simu_note, pitch_range, dur_width, pitch_pad) of init_model()


class EmbedFn(torch.autograd.Function):
    """x [B,S,N,1+D] int64 -> (emb step-major [N,S,B,E], lengths int32 [S*B]); geom = (S, N, P, D, pitch_pad), default the
    32 x 16 x (130+5) grid"""

    @staticmethod
    def forward(ctx, x, w, b, prec=0, geom=DEFAULT_GEOM):
        B = x.shape[0]
        E = w.shape[0]
        S, N, P, D, pad = geom
        assert tuple(x.shape[1:]) == (S, N, 1 + D) and w.shape[1] == P + D, (tuple(x.shape), tuple(w.shape), geom)
        ctx.prec, ctx.geom = prec, geom
        x = x.contiguous()
        emb = _empty(N, S, B, E, dev=w.device)
        lengths = torch.empty(S * B, device=w.device, dtype=torch.int32)
        call('ptv_embed_fwd_geom', ptr(x), ptr(w), ptr(b), ptr(emb), ptr(lengths), B, E, S, N, P, D, pad, stream_ptr())
        ctx.save_for_backward(x, w, b)
        ctx.mark_non_differentiable(lengths)
        # The weight gradient is dy^T . multihot(x), and multihot(x) depends on the input only: in bf16 precision it is built NOW, on a
        # sibling stream under the encoders (bf16: 0 / 1 / 2 are exact), instead of at the very end of the backward pass, where nothing
        # is left to hide it behind
        ctx.mh = ctx.mh_side = None
        if (EMBED_MH_FWD and prec == 1 and E % 8 == 0 and OVERLAP and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
                and not capturing_part()):
            ld = (P + D + 7) // 8 * 8

            def build():
                mh = _empty(B * S * N, ld, dev=w.device, dtype=BF16)
                call('ptv_multihot_geom', ptr(x), ptr(mh), ld, B, S, N, P, D, 1, stream_ptr())
                return mh, record_event()
            ctx.mh, ctx.mh_side = Side(EMBED_MH_SLOT)(build, x)           # (mh_side: the event the backward waits for -- not the whole stream)
        return emb, lengths

    @staticmethod
    def backward(ctx, demb, _dl):
        x, w, b = ctx.saved_tensors
        B, E = x.shape[0], w.shape[0]
        S, N, P, D, _pad = ctx.geom
        slot = _EMB_FAMILY.pop(demb.data_ptr(), None)
        demb2 = demb.contiguous().view(B * S * N, E)
        if ctx.mh is not None:
            mh, ev_mh, ctx.mh, ctx.mh_side = ctx.mh, ctx.mh_side, None, None

            def product():
                wait_event(cur_stream(), ev_mh)
                return wgrad_bias(demb2, mh[:, :P + D], _gbuf(w), _gbuf(b), ctx.prec)     # bias gradient inside the same pass over dy
            from .optim import is_arena_view
            if slot is not None and demb.is_contiguous() and w.grad is None and b.grad is None:
                # the gradient was produced by the note-summary family on a pool stream: the product queues behind it THERE (same slot =
                # same stream: ordered) and joins with it when the backward pass ends
                fam = Side(slot)
                dw, db = fam(product, demb, demb2, mh, x, w, b, after=ev_mh)
                if is_arena_view(w, dw) and is_arena_view(b, db):
                    fam.defer()
                else:
                    fam.join()
                return None, dw, db, None, None
            if slot is not None:
                wait_stream(cur_stream(), _CHILD_STREAMS[('pool', cur_stream().device.index, slot % 4)])
            dw, db = product()
            return None, dw, db, None, None
        if slot is not None:
            wait_stream(cur_stream(), _CHILD_STREAMS[('pool', cur_stream().device.index, slot % 4)])
        ld = (P + D + 7) // 8 * 8
        mh = _empty(B * S * N, ld, dev=w.device)
        call('ptv_multihot_geom', ptr(x), ptr(mh), ld, B, S, N, P, D, 0, stream_ptr())
        dw = gemm(demb2, mh[:, :P + D], _gbuf(w), ta=True, tb=True, acc=True, prec=ctx.prec)
        db = _bgrad(b, demb2)
        return None, dw, db, None, None


# =============================================================================================
# bidirectional GRU, final states only  (RnnEncoder / TextureEncoder / dec_notes_emb_gru)
# =============================================================================================
_BGF = {}
_BRF, _BRB = {}, {}


def _fork_events(cache, n=2):
    """per (device, stream): the fork / join events a composite call needs, created once"""
    cur = cur_stream()
    key = ('ev', cur.device.index, stream_ptr())
    evs = cache.get(key)
    if evs is None:
        evs = [torch.cuda.Event() for _ in range(n)]
        for e in evs:
            e.record(cur)
        cache[key] = evs
    return evs


def _bigru_rows_fwd_composite(x3, lengths, perm, w, out, side, T, M, I, H, dev, seg=None):
    """-> _bigru_forward's result when ptv_bigru_rows_fwd ran its row-kernel branch, else None"""
    if 't' not in _BRF:
        from ._lib import header_enum
        _BRF['t'], _BRF['d'] = header_enum('PtvBrfTensor'), header_enum('PtvBrfDim')
    T_, D_ = _BRF['t'], _BRF['d']
    if (not OVERLAP or side.s == side.main or torch.cuda.is_current_stream_capturing() or not x3.is_contiguous() or x3.dtype != F32
            or H != 128 or I != 128):
        return None
    dims = [0] * D_['PTV_BRF_D_COUNT']
    for k, v in (('M', M), ('T', T), ('H', H), ('I', I)):
        dims[D_['PTV_BRF_D_' + k]] = v
    tens = {'X': x3, 'LENGTHS': lengths, 'PERM': perm, 'OUT': out}
    saved = []
    for d_ in range(2):
        w_ih, w_hh, b_ih, b_hh = w[4 * d_: 4 * d_ + 4]
        pk = notes_packs(w_ih, w_hh, 0)
        hall, h16 = _empty(T + 1, M, H, dev=dev), _empty(T + 1, M, H, dev=dev, dtype=BF16)
        gates = _empty(T, 4, M, H, dev=dev, dtype=BF16)
        saved.append((hall, gates, h16, (lengths if ZERO_SKIP else None), perm, seg))
        tens.update({'PK_WG_H%d' % d_: pk['wg_h'], 'PK_WG_T%d' % d_: pk['wg_t'], 'B_HH%d' % d_: b_hh, 'B_IH%d' % d_: b_ih,
                     'HALL%d' % d_: hall, 'H16_%d' % d_: h16, 'GATES%d' % d_: gates})
    slots = [None] * T_['PTV_BRF_COUNT']
    for k, v in tens.items():
        slots[T_['PTV_BRF_' + k]] = ptr(v)
    evs = _fork_events(_BRF)
    slots[T_['PTV_BRF_FORK_EVENT']], slots[T_['PTV_BRF_JOIN_EVENT']] = evs[0].cuda_event, evs[1].cuda_event
    slots[T_['PTV_BRF_SIDE_STREAM']] = side.s.cuda_stream
    rc = lib().ptv_bigru_rows_fwd((ctypes.c_void_p * len(slots))(*slots), _larr(dims), stream_ptr())
    if rc == -3:
        return None
    check(rc, 'ptv_bigru_rows_fwd')
    _BRF['calls'] = _BRF.get('calls', 0) + 1
    return out, saved


def _bigru_rows_bwd_composite(prec, x3, xf, w, saved, dout, need_dx, dx_acc, side, T, M, I, H):
    """-> _bigru_backward's result when ptv_bigru_rows_bwd ran its row-kernel branch, else None"""
    if 't' not in _BRB:
        from ._lib import header_enum
        _BRB['t'], _BRB['d'] = header_enum('PtvBrbTensor'), header_enum('PtvBrbDim')
    T_, D_ = _BRB['t'], _BRB['d']
    dev = x3.device
    if (prec != 1 or not OVERLAP or side.s == side.main or torch.cuda.is_current_stream_capturing() or H != 128 or I != 128 or T * M < 512
            or xf.dtype != F32 or xf.stride(1) != 1 or xf.stride(0) != I or dout.dtype != F32 or dout.stride(1) != 1):
        _defer_flush()
        return None
    wt_ih = [_WT(w[0], prec), _WT(w[4], prec)] if need_dx else [None, None]
    if need_dx and (wt_ih[0] is None or wt_ih[1] is None):
        _defer_flush()
        return None
    lengths = saved[0][3] if len(saved[0]) > 3 else None
    perm = saved[0][4] if len(saved[0]) > 4 else None
    dims = [0] * D_['PTV_BRB_D_COUNT']
    for k, v in (('M', M), ('T', T), ('H', H), ('I', I), ('DX_ACC', int(dx_acc is not None)), ('DOUT_LD', dout.stride(0))):
        dims[D_['PTV_BRB_D_' + k]] = v
    G = [_gbuf(p_) for p_ in w]
    dx = None
    if need_dx:
        dx = dx_acc if dx_acc is not None else _empty(T * M, I, dev=dev)
    tens = {'X': xf, 'DOUT': dout, 'LENGTHS': lengths, 'PERM': perm, 'DX': dx, 'SEG': saved[0][5] if len(saved[0]) > 5 else None}
    n_scr = lib().ptv_row_gru_persist_scratch_elems(H, M)
    for d_ in range(2):
        hall, gates, h16 = saved[d_][:3]
        pk = notes_packs(w[4 * d_], w[4 * d_ + 1], 0)
        tens.update({'PK_WT%d' % d_: pk['wt'], 'HALL%d' % d_: hall, 'H16_%d' % d_: h16, 'GATES%d' % d_: gates, 'WT_IH%d' % d_: wt_ih[d_],
                     'DGI%d' % d_: _empty(T, M, 3 * H, dev=dev, dtype=BF16), 'DGH%d' % d_: _empty(T, M, 3 * H, dev=dev, dtype=BF16),
                     'SCRATCH%d' % d_: _empty(n_scr, dev=dev, dtype=BF16),
                     'TOP%d' % d_: _ineg1(dev) if (lengths is not None and M % 32 == 0) else None})
    slots = [None] * T_['PTV_BRB_COUNT']
    for k, v in tens.items():
        slots[T_['PTV_BRB_' + k]] = ptr(v)
    for d_ in range(2):
        for j, nm in enumerate(('W_IH', 'W_HH', 'B_IH', 'B_HH')):
            slots[T_['PTV_BRB_G_%s%d' % (nm, d_)]] = ptr(G[4 * d_ + j])
    evs = _fork_events(_BRB)
    slots[T_['PTV_BRB_FORK_EVENT']], slots[T_['PTV_BRB_JOIN_EVENT']] = evs[0].cuda_event, evs[1].cuda_event
    slots[T_['PTV_BRB_SIDE_STREAM']] = side.s.cuda_stream
    arr_, darr_, sp_ = (ctypes.c_void_p * len(slots))(*slots), _larr(dims), stream_ptr()
    rc = _defer_or_run('rows_bwd', (arr_, darr_, (tens, G, saved, dout, xf)), lambda: lib().ptv_bigru_rows_bwd(arr_, darr_, sp_))
    _SIDE_DEPTH[1] = 1
    if rc == -3:                          # (H = I = 128 was checked above: ptv_bigru_rows_bwd's only refusal)
        raise RuntimeError('ptv_bigru_rows_bwd refused a configuration its Python-side checks accepted')
    check(rc, 'ptv_bigru_rows_bwd')
    _BRB['calls'] = _BRB.get('calls', 0) + 1
    return G[0:4] + G[4:8], (dx.view(T, M, I) if need_dx else None)


def _bigru_fwd_composite(prec, xf, lengths, w, w16, out, T, M, I, H, dev):
    """-> _bigru_forward's result when ptv_bigru_final_fwd ran its persistent branch, else None"""
    if 't' not in _BGF:
        from ._lib import header_enum
        _BGF['t'], _BGF['d'] = header_enum('PtvBgfTensor'), header_enum('PtvBgfDim')
    T_, D_ = _BGF['t'], _BGF['d']
    wih16 = [_W(w[0], prec), _W(w[4], prec)]
    if (torch.cuda.is_current_stream_capturing() or xf.dtype not in (F32, BF16) or xf.stride(1) != 1 or xf.stride(0) != I
            or wih16[0].dtype != wih16[1].dtype or not wih16[0].is_contiguous() or not wih16[1].is_contiguous()):
        return None
    dims = [0] * D_['PTV_BGF_D_COUNT']
    for k, v in (('M', M), ('T', T), ('H', H), ('I', I), ('X_BF16', _bf(xf)), ('WIH_F32', int(wih16[0].dtype == F32))):
        dims[D_['PTV_BGF_D_' + k]] = v
    saved = []
    tens = {'X': xf, 'LENGTHS': lengths, 'OUT': out, 'SYNC': _persist_sync(2, dev)}
    for d_ in range(2):
        hall, h16 = _empty(T + 1, M, H, dev=dev), _empty(T + 1, M, H, dev=dev, dtype=BF16)
        gates = _empty(T, 4, M, H, dev=dev, dtype=BF16)
        saved.append((hall, gates, h16))
        tens.update({'W16_IH%d' % d_: wih16[d_], 'B_IH%d' % d_: w[4 * d_ + 2], 'W16_HH%d' % d_: w16[d_], 'B_HH%d' % d_: w[4 * d_ + 3],
                     'GI%d' % d_: _empty(T * M, 3 * H, dev=dev, dtype=BF16), 'HALL%d' % d_: hall, 'H16_%d' % d_: h16, 'GATES%d' % d_: gates,
                     'XCH%d' % d_: torch.empty((T + 1) * M * H, device=dev, dtype=BF16)})
    slots = [None] * T_['PTV_BGF_COUNT']
    for k, v in tens.items():
        slots[T_['PTV_BGF_' + k]] = ptr(v)
    cur = cur_stream()
    prev = _PERSIST_LAST.get(cur.device.index)
    done = torch.cuda.Event()
    if prev is not None:
        slots[T_['PTV_BGF_WAIT_EVENT']] = prev.cuda_event
    done.record(cur)
    slots[T_['PTV_BGF_RECORD_EVENT']] = done.cuda_event
    _chain_prio()
    rc = lib().ptv_bigru_final_fwd((ctypes.c_void_p * len(slots))(*slots), _larr(dims), stream_ptr())
    if rc == -3:
        return None
    check(rc, 'ptv_bigru_final_fwd')
    _PERSIST_LAST[cur.device.index] = done
    _BGF['calls'] = _BGF.get('calls', 0) + 1
    return out, saved


def _bigru_forward(prec, x3, lengths, w):
    """x3 [T,M,I] step-major.  w = (w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r).
    Returns out [M,2H] and the saved state for backward.  The two directions are independent
    chains: the reverse one runs on a sibling stream."""
    T, M, I = x3.shape
    H = w[1].shape[1]
    dev = x3.device
    xf = x3.reshape(T * M, I)
    out = _empty(M, 2 * H, dev=dev)

    def direction(d):
        w_ih, w_hh, b_ih, b_hh = w[4 * d: 4 * d + 4]
        gi = gemm(xf, _W(w_ih, prec), bias=b_ih, prec=prec, out_dtype=_act_dtype(prec, H))          # [T*M, 3H]
        hall = _empty(T + 1, M, H, dev=dev)
        hall[0].zero_()
        h16 = _hall16(prec, T + 1, M, H, dev)
        gates = _empty(T, 4, M, H, dev=dev, dtype=_act_dtype(prec, H))
        gru_fwd(prec, gi, M * 3 * H, 3 * H, w_hh, b_hh, hall, gates, lengths=lengths, reverse=bool(d), hall16=h16)
        copy2d(out[:, d * H:(d + 1) * H], hall[T])
        return hall, gates, h16

    adt = _act_dtype(prec, H)
    w16 = [_W(w[1], prec), _W(w[5], prec)]
    if prec == 1 and T >= 2 and adt == BF16 and w16[0].dtype == BF16 and w16[1].dtype == BF16 and persist_supported(2, M, H, T):
        # both directions in ONE persistent launch (csrc/gru_persist.hip): per step the two chains share the exchange latency
        if BIGRU_BWD_COMPOSITE:
            res = _bigru_fwd_composite(prec, xf, lengths, w, w16, out, T, M, I, H, dev)
            if res is not None:
                return res
        chains, saved = [], []
        for d in range(2):
            w_ih, w_hh, b_ih, b_hh = w[4 * d: 4 * d + 4]
            gi = gemm(xf, _W(w_ih, prec), bias=b_ih, prec=prec, out_dtype=adt)
            hall = _empty(T + 1, M, H, dev=dev)
            hall[0].zero_()
            h16 = _hall16(prec, T + 1, M, H, dev)
            gates = _empty(T, 4, M, H, dev=dev, dtype=adt)
            chains.append(dict(gi=gi, gi_step=M * 3 * H, gi_ld=3 * H, w16=w16[d], b_hh=b_hh, hall=hall, hall16=h16, gates=gates,
                               lengths=lengths, reverse=bool(d)))
            saved.append((hall, gates, h16))
        gru_persist_fwd(M, H, T, chains)
        for d in range(2):
            copy2d(out[:, d * H:(d + 1) * H], saved[d][0][T])
        return out, saved

    zero_skip_sync()
    if row_gru_ok(prec, H, I, M, adt) and x3.dtype == F32:
        # many short independent rows (dec_notes_emb_gru: 32*B rows x 16 notes): row-partitioned persistent kernels, one launch per
        # direction for the whole sequence (csrc/notes_persist.hip), input product fused; the directions overlap on sibling streams
        # (optional) rows sorted by length: a 64-row panel then holds rows of (almost) one length and passes over the steps that are masked
        # for ALL of them -- in row order a panel's longest row is nearly always the longest of the batch
        perm, seg = None, None
        if lengths is not None and ZERO_SKIP and SORT_ROWS and T <= 38:
            perm = torch.empty(M, device=dev, dtype=torch.int32)
            call('ptv_rows_by_length', ptr(lengths), ptr(perm), M, T, stream_ptr())
            if WGRAD_SEG and M % 128 == 0 and lib().ptv_wgrad_seg_supported(T * M, M):
                # (round 6) ... and the live prefix of every position in that order: the weight_hh products clip to it (ptv_bigru_rows_bwd)
                len_s = torch.empty(M, device=dev, dtype=torch.int32)
                call('ptv_gather_rows', ptr(len_s), ptr(lengths), ptr(perm), M, 1, 0, 0, 1, stream_ptr())
                seg = torch.empty(T, device=dev, dtype=torch.int32)
                call('ptv_rows_seg_counts', ptr(len_s), M, T, ptr(seg), stream_ptr())

        def rows(d):
            w_ih, w_hh, b_ih, b_hh = w[4 * d: 4 * d + 4]
            pk = notes_packs(w_ih, w_hh, 0)
            hall = _empty(T + 1, M, H, dev=dev)
            hall[0].zero_()
            h16 = _empty(T + 1, M, H, dev=dev, dtype=BF16)
            gates = _empty(T, 4, M, H, dev=dev, dtype=BF16)
            call('ptv_row_gru_persist_fwd_perm', H, ptr(pk['wg_h']), ptr(pk['wg_t']), ptr(b_hh), ptr(b_ih), None, ptr(x3), M * I,
                 ptr(lengths) if lengths is not None else None, ptr(perm), ptr(hall), ptr(h16), ptr(gates), out.data_ptr() + 4 * d * H,
                 2 * H, M, T, d, stream_ptr())
            # (the backward must skip the same fully masked panel steps, with the same row order)
            return hall, gates, h16, (lengths if ZERO_SKIP else None), perm, seg
        side = Side(BIGRU_SLOT)
        if BIGRU_BWD_COMPOSITE:
            res = _bigru_rows_fwd_composite(x3, lengths, perm, w, out, side, T, M, I, H, dev, seg)
            if res is not None:
                return res
        rev = side(lambda: rows(1), x3, out)
        fwd = rows(0)
        side.join()
        return out, [fwd, rev]

    side = Side(BIGRU_SLOT)
    rev = side(lambda: direction(1), xf, out)
    fwd = direction(0)
    side.join()
    return out, [fwd, rev]


BIGRU_BWD_COMPOSITE = os.environ.get('PTV_BWD_COMPOSITES', '1') != '0'    # the encoders' bi-GRU backward through ptv_bigru_final_bwd (one C call)
_BGB = {}


def _bigru_bwd_composite(prec, x3, xf, w, saved, dout, need_dx, dx_acc, wts, side, T, M, I, H):
    """-> _bigru_backward's result when ptv_bigru_final_bwd ran its persistent branch, else None"""
    if 't' not in _BGB:
        from ._lib import header_enum
        _BGB['t'], _BGB['d'] = header_enum('PtvBgbTensor'), header_enum('PtvBgbDim')
    T_, D_ = _BGB['t'], _BGB['d']
    dev = x3.device
    if (T * M < 512 or not OVERLAP or side.s == side.main or torch.cuda.is_current_stream_capturing() or xf.stride(1) != 1
            or xf.stride(0) != I or xf.dtype not in (F32, BF16) or dout.dtype != F32 or dout.stride(1) != 1 or dout.stride(0) != 2 * H
            or any(sv[0].dtype != F32 or sv[2] is None or sv[1].dtype != BF16 for sv in saved[:2])):
        return None
    wt_ih = [_WT(w[0], prec), _WT(w[4], prec)] if need_dx else [None, None]
    if need_dx and (wt_ih[0] is None or wt_ih[1] is None):
        return None
    S = persist_splitk(2, M, H)
    dims = [0] * D_['PTV_BGB_D_COUNT']
    for k, v in (('M', M), ('T', T), ('H', H), ('I', I), ('X_BF16', _bf(xf)), ('DX_ACC', int(dx_acc is not None)), ('SPLITK', S)):
        dims[D_['PTV_BGB_D_' + k]] = v
    G = [_gbuf(p_) for p_ in w]                           # (w_ih, w_hh, b_ih, b_hh) x 2 directions
    dx = None
    if need_dx:
        dx = dx_acc if dx_acc is not None else _empty(T * M, I, dev=dev)
    n_part = lib().ptv_gru_persist_part_elems(2, M, H, S) if S else 0
    tens = {'X': xf, 'DOUT': dout, 'DX': dx, 'SYNC': _persist_sync(2, dev)}
    for d_ in range(2):
        hall, gates, h16 = saved[d_][:3]
        tens.update({'HALL%d' % d_: hall, 'H16_%d' % d_: h16, 'GATES%d' % d_: gates, 'WT_HH%d' % d_: wts[d_], 'WT_IH%d' % d_: wt_ih[d_],
                     'DGI%d' % d_: _empty(T, M, 3 * H, dev=dev, dtype=BF16), 'DGH%d' % d_: _empty(T, M, 3 * H, dev=dev, dtype=BF16),
                     'XCH%d' % d_: torch.empty(T * M * 3 * H, device=dev, dtype=BF16),
                     'PART%d' % d_: torch.empty(n_part, device=dev) if S else None})
    slots = [None] * T_['PTV_BGB_COUNT']
    for k, v in tens.items():
        slots[T_['PTV_BGB_' + k]] = ptr(v)
    for d_ in range(2):
        for j, nm in enumerate(('W_IH', 'W_HH', 'B_IH', 'B_HH')):
            slots[T_['PTV_BGB_G_%s%d' % (nm, d_)]] = ptr(G[4 * d_ + j])
    cur = cur_stream()
    evs = _BGB.get(('ev', cur.device.index, stream_ptr()))
    if evs is None:                                   # fork / join events of this stream, created once
        evs = [torch.cuda.Event(), torch.cuda.Event()]
        for e in evs:
            e.record(cur)
        _BGB[('ev', cur.device.index, stream_ptr())] = evs
    slots[T_['PTV_BGB_FORK_EVENT']], slots[T_['PTV_BGB_JOIN_EVENT']] = evs[0].cuda_event, evs[1].cuda_event
    slots[T_['PTV_BGB_SIDE_STREAM']] = side.s.cuda_stream
    prev = _PERSIST_LAST.get(cur.device.index)
    done = torch.cuda.Event()
    if prev is not None:
        slots[T_['PTV_BGB_WAIT_EVENT']] = prev.cuda_event
    done.record(cur)
    slots[T_['PTV_BGB_RECORD_EVENT']] = done.cuda_event
    rc = lib().ptv_bigru_final_bwd((ctypes.c_void_p * len(slots))(*slots), _larr(dims), stream_ptr())
    _SIDE_DEPTH[1] = 1
    if rc == -3:                          # (persist_supported() was asked before the arena views were taken)
        raise RuntimeError('ptv_bigru_final_bwd refused a configuration its Python-side checks accepted')
    check(rc, 'ptv_bigru_final_bwd')
    _PERSIST_LAST[cur.device.index] = done
    _BGB['calls'] = _BGB.get('calls', 0) + 1
    # (the side stream was joined inside the call: what it read may be released in this stream's order)
    return G[0:4] + G[4:8], (dx.view(T, M, I) if need_dx else None)


def _bigru_backward(prec, x3, w, saved, dout, need_dx, dx_acc=None, rev_slot=None):
    """-> ([dw_ih, dw_hh, db_ih, db_hh] x 2 directions, dx [T,M,I] or None).
    dx_acc ([T*M, I] fp32, or None): a gradient that already arrived at x3 from another consumer -- both directions' input-gradient
    products ACCUMULATE into it and it is returned as dx (round 4: autograd used to add the two consumers' 134-MB gradients of the note
    embedding with an ATen kernel, and the two directions' dx met in a copy kernel; now both are the accumulate mode of products that
    run anyway).  The second direction's dx product runs on the caller's stream after the join (its BPTT ran on the sibling stream)."""
    T, M, I = x3.shape
    H = w[1].shape[1]
    xf = x3.reshape(T * M, I)
    late = {}                                            # the second direction's (dgi, top) for its dx product after the join

    def dx_of(d, dgi2, top, first):
        if not need_dx:
            return None
        if first and dx_acc is None:
            return gemm_dx(dgi2, w[4 * d], prec=prec, m_top=top, m_unit=M if top is not None else 0)
        return gemm_dx(dgi2, w[4 * d], out=dx_acc if first else late['dx'], acc=True, prec=prec, m_top=top, m_unit=M if top is not None else 0)

    def direction(d):
        w_ih, w_hh, b_ih, b_hh = w[4 * d: 4 * d + 4]
        hall, gates, h16 = saved[d][:3]
        dgi, dgh, _ = gru_bwd(prec, hall, gates, w_hh, dh_last=dout[:, d * H:(d + 1) * H], reverse=bool(d),
                              need_dh0=False)
        dgi2, dgh2 = dgi.view(T * M, 3 * H), dgh.view(T * M, 3 * H)
        dw_ih, db_ih = wgrad_bias(dgi2, xf, _gbuf(w_ih), _gbuf(b_ih), prec)
        dw_hh, db_hh = wgrad_bias(dgh2, (h16 if h16 is not None else hall)[:T].view(T * M, H), _gbuf(w_hh), _gbuf(b_hh), prec)
        if d:
            late['dgi'], late['top'] = dgi2, None
        return [dw_ih, dw_hh, db_ih, db_hh], (dx_of(0, dgi2, None, True) if d == 0 else None)

    def products(d, dgi, dgh, top=None, with_dx=True):
        """top (device int, from the BPTT kernel): no row is longer than top + 1, so dgi (indexed by time) is zero after that time
        and dgh (indexed by processing step) after that step -- or, in the reversed direction, BEFORE step T - top - 1"""
        w_ih, w_hh, b_ih, b_hh = w[4 * d: 4 * d + 4]
        hall, gates, h16 = saved[d][:3]
        dgi2, dgh2 = dgi.view(T * M, 3 * H), dgh.view(T * M, 3 * H)
        dw_ih, db_ih = wgrad_bias(dgi2, xf, _gbuf(w_ih), _gbuf(b_ih), prec, top, M)
        seg = saved[d][5] if (len(saved[d]) > 5 and top is not None and h16 is not None and saved[d][4] is not None) else None
        dw_hh, db_hh = wgrad_bias(dgh2, (h16 if h16 is not None else hall)[:T].view(T * M, H), _gbuf(w_hh), _gbuf(b_hh), prec, top, M,
                                  k_rev=T if d else 0, seg=seg, seg_period=-T if d else T)
        if d:
            late['dgi'], late['top'] = dgi2, top
        return [dw_ih, dw_hh, db_ih, db_hh], (dx_of(0, dgi2, top, True) if (d == 0 and with_dx) else None)

    side = Side(BIGRU_SLOT_BWD if rev_slot is None else rev_slot)
    wts = [_WT(w[1], prec), _WT(w[5], prec)]
    adt = _act_dtype(prec, H)
    rows_branch = len(saved[0]) > 3 and saved[0][1].dtype == BF16 and saved[0][2] is not None
    if _DEFER is not None and not (rows_branch and BIGRU_BWD_COMPOSITE and not (
            T >= 2 and adt == BF16 and wts[0] is not None and wts[1] is not None and persist_supported(2, M, H, T))):
        _defer_flush()                    # (only the row-kernel composite can be collected: everything else below launches at once)
    if (T >= 2 and adt == BF16 and wts[0] is not None and wts[1] is not None and saved[0][1].dtype == BF16
            and saved[0][2] is not None and persist_supported(2, M, H, T)):
        # BPTT of both directions in ONE persistent launch, then the weight-gradient products of the two on sibling streams
        if BIGRU_BWD_COMPOSITE:
            res = _bigru_bwd_composite(prec, x3, xf, w, saved, dout, need_dx, dx_acc, wts, side, T, M, I, H)
            if res is not None:
                return res
        chains = []
        for d in range(2):
            hall, gates, h16 = saved[d][:3]
            chains.append(dict(hall=hall, gates=gates, wt16=wts[d], dh_ext=None, dh_last=dout[:, d * H:(d + 1) * H],
                               dgi=_empty(T, M, 3 * H, dev=x3.device, dtype=adt), dgh=_empty(T, M, 3 * H, dev=x3.device, dtype=adt),
                               dh0=None, reverse=bool(d)))
        gru_persist_bwd(M, H, T, chains)
        g1, _ = side(lambda: products(1, chains[1]['dgi'], chains[1]['dgh']), xf, dout, chains[1]['dgi'], chains[1]['dgh'])
        g0, dx0 = products(0, chains[0]['dgi'], chains[0]['dgh'])
    elif (len(saved[0]) > 3 and saved[0][1].dtype == BF16 and saved[0][2] is not None):
        # (a forward that ran on the row kernels -- a 4-entry saved state -- left its gates in their private unit-blocked layout, and with
        # lengths the gates of skipped panel steps unwritten: same kernels back)
        if BIGRU_BWD_COMPOSITE:
            res = _bigru_rows_bwd_composite(prec, x3, xf, w, saved, dout, need_dx, dx_acc, side, T, M, I, H)
            if res is not None:
                return res

        def rows(d):
            w_ih, w_hh = w[4 * d], w[4 * d + 1]
            hall, gates, h16 = saved[d][:3]
            lengths = saved[d][3] if len(saved[d]) > 3 else None
            perm = saved[d][4] if len(saved[d]) > 4 else None
            pk = notes_packs(w_ih, w_hh, 0)
            dgi = _empty(T, M, 3 * H, dev=x3.device, dtype=BF16)
            dgh = _empty(T, M, 3 * H, dev=x3.device, dtype=BF16)
            scratch = _empty(lib().ptv_row_gru_persist_scratch_elems(H, M), dev=x3.device, dtype=BF16)
            top = _ineg1(x3.device) if (lengths is not None and M % 32 == 0) else None
            call('ptv_row_gru_persist_bwd_perm', H, ptr(pk['wt']), ptr(hall), ptr(gates), None, dout.data_ptr() + 4 * d * H, dout.stride(0),
                 ptr(lengths) if lengths is not None else None, ptr(perm), ptr(dgi), ptr(dgh), None, ptr(scratch), M, T, d, ptr(top),
                 stream_ptr())
            return products(d, dgi, dgh, top)
        g1, _ = side(lambda: rows(1), xf, dout)
        g0, dx0 = rows(0)
    else:
        g1, _ = side(lambda: direction(1), xf, dout)
        g0, dx0 = direction(0)
    side.join()
    if need_dx:
        late['dx'] = dx_acc if dx_acc is not None else dx0
        # (dgi of the second direction was allocated under the sibling stream and is read here, on the caller's: tell the caching allocator,
        # or its block returns to the sibling's pool when `late` dies and may be overwritten while this product is still queued)
        if side.s != side.main:
            _record_stream([late['dgi'], late['top']], side.main)
        dx_of(1, late['dgi'], late['top'], False)
        dx0 = late['dx']
    return g0 + g1, (dx0.view(T, M, I) if need_dx else None)


# The note embedding has two consumers -- the ground-truth note summaries (this bi-GRU) and the decoder's note tokens -- and autograd
# summed their two 134-MB gradients with an ATen add at the very end of the backward pass.  The teacher-forced decoder node instead
# parks its gradient here and returns None; the summary node (which always runs later: the decoder consumes its output) accumulates
# its own input-gradient products into that buffer and returns it as the one gradient of the embedding.
_EMB_LINK = {}
EMB_LINK = True
# The note-summary BPTT as a stream family of its own (BiGruFinalFn.backward): pool slot, or -1 = on the stream autograd gives the node.
# Round 5, same-box A/B (profiles/r05_ab_runs.txt): the family starts 0.4 ms earlier (at the decoder node's event instead of behind the
# chord decoder's join and the reparametrisation node, which autograd happens to run first) and the tail after dec_bwd:end shrinks from
# 2.65 to 2.3 ms in the trace marks -- but the STEP gets slower, 7.73-7.94 against 7.62-7.65 ms for every placement of its two
# directions (pool 0 / 0, 0 / 3, 0 / 2, 0 / 1): beside the summary kernels the encoders' persistent BPTTs and the deferred deep
# products slow down by more than the earlier start buys.  Off by default; kept as the measured answer to "run the families at once".
SUMMARY_FAMILY_SLOT = int(os.environ.get('PTV_SUMMARY_FAMILY', '-1'))
SUMMARY_FAMILY_REV = 4
_SUMMARY_EV = {}      # data_ptr of the decoder node's note-summary gradient -> event after which it is final
_EMB_FAMILY = {}      # data_ptr of the embedding gradient the family produced -> its pool slot (EmbedFn.backward queues behind it)


def emb_link_arm(emb):
    """called by PtvaeDecoder._summarize before it builds the summary node on `emb`"""
    if EMB_LINK and torch.is_grad_enabled() and emb.requires_grad:
        if len(_EMB_LINK) > 8:
            _EMB_LINK.clear()
        _EMB_LINK[emb.data_ptr()] = {'demb': None}


class BiGruFinalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x3, lengths, prec, *w):
        x3 = x3.contiguous()
        out, saved = _bigru_forward(prec, x3, lengths, w)
        ctx.save_for_backward(x3, *w)
        ctx.saved_state = saved
        ctx.prec = prec
        ctx.link = _EMB_LINK.get(x3.data_ptr())
        return out

    @staticmethod
    def backward(ctx, dout):
        x3, *w = ctx.saved_tensors
        mark('bigru_bwd:start M=%d @%x' % (x3.shape[1], stream_ptr() & 0xffff))
        dx_acc = None
        if ctx.link is not None and ctx.link['demb'] is not None and ctx.needs_input_grad[0]:
            dx_acc, ctx.link['demb'] = ctx.link['demb'].view(-1, x3.shape[2]), None     # the decoder node's gradient of the same tensor
            _EMB_LINK.pop(x3.data_ptr(), None)
        from .optim import is_arena_view
        ev = _SUMMARY_EV.pop(dout.data_ptr(), None)
        if ev is not None and (dx_acc is None or not dout.is_contiguous()):
            ev = None               # (only in the train step's own structure: the decoder node parked its embedding gradient with this node)
        fam = None
        if ev is not None:
            # The note-summary BPTT (the decoder node produced `dout`): a family of its own.  Autograd runs this node on the stream of
            # its forward -- the one the chord encoder's BPTT will want next -- and behind whatever the pass queued there since; the
            # family instead takes a pool stream that is free by now (the chord decoder's), starts from the decoder node's event and
            # is joined when the backward pass ends: it ends in parameter gradients and the embedding's (EmbedFn follows it there).
            fam = Side(SUMMARY_FAMILY_SLOT)
            state = ctx.saved_state
            grads, dx = fam(lambda: _bigru_backward(ctx.prec, x3, w, state, dout, ctx.needs_input_grad[0], dx_acc, SUMMARY_FAMILY_REV),
                            x3, dout, dx_acc, state, *w, after=ev)
        else:
            grads, dx = _bigru_backward(ctx.prec, x3, w, ctx.saved_state, dout.contiguous(), ctx.needs_input_grad[0], dx_acc)
        mark('bigru_bwd:end M=%d @%x' % (x3.shape[1], stream_ptr() & 0xffff))
        ctx.saved_state = None
        # (w is ordered by direction, the products return [ih, hh, b_ih, b_hh] per direction)
        pairs = list(zip(w[0:4], grads[0:4])) + list(zip(w[4:8], grads[4:8]))
        adopted = all(p_.grad is None and is_arena_view(p_, g_) for p_, g_ in pairs)
        if fam is not None:
            streams = (fam.s,)
            if adopted:
                fam.defer()
                if dx is not None:
                    if len(_EMB_FAMILY) > 8:
                        _EMB_FAMILY.clear()
                    _EMB_FAMILY[dx.data_ptr()] = SUMMARY_FAMILY_SLOT
            else:
                fam.join()
                streams = ()
        else:
            streams = ()
        if GRAD_READY_HOOK is not None and adopted:
            GRAD_READY_HOOK([p_ for p_, _ in pairs], streams)   # data parallel: a bi-GRU's 8 gradients are final once these streams drain
        return (dx, None, None) + tuple(grads)


# =============================================================================================
# encoder heads: mu = linear_mu(h), std = exp(linear_var(h))   (ptvae.py:26-28, 119-121)
# =============================================================================================
class EncoderHeadsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, w_mu, b_mu, w_var, b_var, prec):
        mu = gemm(h, _W(w_mu, prec), bias=b_mu, prec=prec)
        sd = gemm(h, _W(w_var, prec), bias=b_var, act=1, prec=prec)
        ctx.save_for_backward(h, w_mu, w_var, mu, sd, b_mu, b_var)
        ctx.prec = prec
        return mu, sd

    @staticmethod
    def backward(ctx, dmu, dsd):
        h, w_mu, w_var, mu, sd, b_mu, b_var = ctx.saved_tensors
        mark('enc_heads_bwd:start @%x' % (stream_ptr() & 0xffff))
        prec = ctx.prec
        B, Z = mu.shape
        dev = h.device
        dmu = None if dmu is None else dmu.contiguous()
        dsd = None if dsd is None else dsd.contiguous()
        gmu, glv = _empty(B, Z, dev=dev), _empty(B, Z, dev=dev)
        call('ptv_reparam_kl_bwd', ptr(mu), ptr(sd), None, None, 0, ptr(dmu), ptr(dsd), 0.0, 1, ptr(gmu), ptr(glv),
             B, Z, stream_ptr())
        dh = gemm_dx(gmu, w_mu, prec=prec)
        gemm_dx(glv, w_var, out=dh, acc=True, prec=prec)
        # both heads' weight and bias gradients: one product launch + one reduction launch (six launches until round 5)
        dw_mu, dw_var, db_mu, db_var = _gbuf(w_mu), _gbuf(w_var), _gbuf(b_mu), _gbuf(b_var)
        wgrad_bias_batch([(gmu, h, dw_mu, db_mu), (glv, h, dw_var, db_var)], prec)
        return dh, dw_mu, db_mu, dw_var, db_var, None


# =============================================================================================
# reparameterize: z = mu + std * eps   (train_utils.py:33-34)
# =============================================================================================
class ReparamFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, sd, eps):
        B, Z = mu.shape
        z = _empty(B, Z, dev=mu.device)
        scratch = _zeros(1, dev=mu.device)
        call('ptv_reparam_kl_fwd', ptr(mu.contiguous()), ptr(sd.contiguous()), ptr(eps), ptr(z), Z, ptr(scratch), B, Z,
             stream_ptr())
        ctx.save_for_backward(mu, sd, eps)
        return z

    @staticmethod
    def backward(ctx, dz):
        mark('reparam_bwd:start @%x' % (stream_ptr() & 0xffff))
        mu, sd, eps = ctx.saved_tensors
        B, Z = mu.shape
        dz = dz.contiguous()
        dmu, dsd = _empty(B, Z, dev=mu.device), _empty(B, Z, dev=mu.device)
        call('ptv_reparam_kl_bwd', ptr(mu), ptr(sd), ptr(eps), ptr(dz), Z, None, None, 0.0, 0, ptr(dmu), ptr(dsd), B, Z,
             stream_ptr())
        return dmu, dsd, None


# =============================================================================================
# TextureEncoder front end: conv + relu + maxpool   (ptvae.py:95-99,112-114)
# =============================================================================================
TXT_ARGMAX = True


class TextureFrontFn(torch.autograd.Function):
    """pr_mat [B,32,128] -> the rows of the reference's raw view of the pooled conv map (ptvae.py:112-114): [B*8, C*29], held with a row
    stride padded to a multiple of 8 floats (fc1's operand: aligned rows for the MFMA loaders -- its weight gradient took the
    element-wise path at 290-float rows, 334 us at the very end of the backward pass)"""

    @staticmethod
    def forward(ctx, pr_mat, w, b):
        B, C = pr_mat.shape[0], w.shape[0]
        pr_mat = pr_mat.contiguous()
        W = C * 29
        feat = _empty(B * 8, _pad8(W), dev=w.device)             # (the kernel zeroes the row padding)
        # which pooled position won, for the backward (1 byte per output instead of recomputing the convolution there)
        arg = torch.empty(B * 8, W, device=w.device, dtype=torch.int8) if (TXT_ARGMAX and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])) else None
        call('ptv_txt_conv_relu_pool_fwd_rows', ptr(pr_mat), ptr(w), ptr(b), ptr(feat), feat.stride(0), B, C, ptr(arg), stream_ptr())
        ctx.save_for_backward(pr_mat, w, b)
        ctx.arg = arg
        return feat[:, :W]

    @staticmethod
    def backward(ctx, dfeat):
        pr_mat, w, b = ctx.saved_tensors
        dw, db = _gbuf(w), _gbuf(b)
        if dfeat.stride(1) != 1:
            dfeat = dfeat.contiguous()
        arg, ctx.arg = ctx.arg, None
        call('ptv_txt_conv_relu_pool_bwd_rows', ptr(pr_mat), ptr(w), ptr(b), ptr(dfeat), dfeat.stride(0), ptr(dw), ptr(db),
             pr_mat.shape[0], w.shape[0], ptr(arg), stream_ptr())
        return None, dw, db


# =============================================================================================
# PtvaeDecoder, teacher-forced (tfr1 = tfr2 = 1): restructured from 2912 dependent cells to
# 32 + 15 + 5 sequential steps.  ptvae.py:336-496.
# =============================================================================================
# ---------------------------------------------------------------------------------------------
# row-partitioned persistent notes GRU (csrc/notes_persist.hip)
# ---------------------------------------------------------------------------------------------
NOTES_PERSIST = True
class PackCache:
    """packed weight copies keyed by the source parameters' addresses, valid while (a) the stamp -- in-place version counters, the
    fused optimiser's step count -- is unchanged and (b) the tensors that were packed are still alive: an address alone can come
    back for a different tensor once the first one is freed (weak references to the tensor objects seen at insertion; the forward
    pass inserts with the module's Parameter objects, the backward's saved-tensor aliases then hit the same entry)"""

    def __init__(self, slots=6):
        self.d, self.slots = {}, slots

    def get(self, srcs, stamp):
        hit = self.d.get(tuple(p.data_ptr() for p in srcs))
        if hit is not None and hit[0] == stamp and all(r() is not None for r in hit[2]):
            return hit[1]
        return None

    def put(self, srcs, stamp, packs):
        if len(self.d) >= self.slots:
            self.d.clear()
        self.d[tuple(p.data_ptr() for p in srcs)] = (stamp, packs, [weakref.ref(p) for p in srcs])
        return packs


_NOTES_PACKS = PackCache()


def pack_mfma_b(w2d, K=None, pairs=False):
    """fp32 [N, >=K] (any row stride) -> MFMA B-fragment-major bf16 (ptv_pack_mfma_b): one contiguous 1-KB load per fragment"""
    N = w2d.shape[0]
    K = w2d.shape[1] if K is None else K
    out = torch.empty(lib().ptv_pack_mfma_b_size(N, K), device=w2d.device, dtype=BF16)
    call('ptv_pack_mfma_b', ptr(w2d), w2d.stride(0), N, K, ptr(out), int(pairs), stream_ptr())
    return out


def pack_multi(jobs):
    """jobs: tuples (src_ptr, ld, N, K, out_ptr, pairs, trans, NT, kb0, KBtot) -- ptv_pack_mfma_b2's arguments -- in one launch"""
    flat = (ctypes.c_long * (10 * len(jobs)))(*[int(v) for j in jobs for v in j])
    check(lib().ptv_pack_mfma_multi(flat, len(jobs), stream_ptr()), 'ptv_pack_mfma_multi')


def param_stamp(params):
    """changes whenever one of `params` may have changed: in-place version counters + the fused optimiser's step count"""
    from .optim import _SHADOW_OF
    ent = _SHADOW_OF.get(params[0].data_ptr())
    opt = ent[0]() if ent is not None else None
    return (sum(p._version for p in params), opt.step_count if opt is not None else -1, opt._dirty if opt is not None else -1)


def notes_packs(w_ih, w_hh, Ht):
    """fragment-major copies of the notes-GRU weights: W_hh, W_ih[:, Ht:] (forward) and W_hh^T (BPTT); cached per parameter version"""
    stamp = param_stamp([w_ih, w_hh])
    hit = _NOTES_PACKS.get([w_ih, w_hh], stamp)
    if hit is not None:
        return hit
    dev = w_hh.device
    H3, H = w_hh.shape
    w_x = w_ih[:, Ht:]
    E = w_x.shape[1]
    size = lambda N, K: lib().ptv_pack_mfma_b_size(N, K)
    pk = dict(wg_h=torch.empty(size(H3, H), device=dev, dtype=BF16), wg_t=torch.empty(size(H3, E), device=dev, dtype=BF16),
              wt=torch.empty(size(H, H3), device=dev, dtype=BF16))
    # (W_hh^T is packed straight from W_hh with transposed reads: no staging transpose; one launch for the three)
    # (H = 512, the notes GRU's forward: plain tiles, pairs = 0 -- the wave-role kernel hands accumulators over through LDS and picks its
    # own lane layout there; the note-summary GRU's forward (H = 128) and both BPTTs: pair-interleaved, their epilogues run in the MFMA
    # lane layout)
    pf = 0 if H == 512 else 1
    pack_multi([(w_hh.data_ptr(), w_hh.stride(0), H3, H, pk['wg_h'].data_ptr(), pf, 0, H3 // 16, 0, (H + 31) // 32),
                (w_x.data_ptr(), w_x.stride(0), H3, E, pk['wg_t'].data_ptr(), pf, 0, H3 // 16, 0, (E + 31) // 32),
                (w_hh.data_ptr(), w_hh.stride(0), H, H3, pk['wt'].data_ptr(), 1, 1, H // 16, 0, (H3 + 31) // 32)])
    return _NOTES_PACKS.put([w_ih, w_hh], stamp, pk)


_HEADS_PACKS = PackCache()
HEADS_FUSED = True
HEADS_WGRAD_FUSED = True


def heads_ok(prec, Hn, NP, Hd, hn16, hd16):
    """the fused per-note heads (csrc/heads.hip): bf16 precision at the init_model() geometry"""
    return HEADS_FUSED and prec == 1 and BF16_STORAGE and (Hn, NP, Hd) == (512, 130, 64) and hn16 is not None and hd16 is not None


def heads_packs(w_p, w_dh):
    """fragment-major bf16 copies of pitch_out_linear / dur_hid_linear for ptv_heads_fwd / ptv_heads_bwd; cached per parameter version"""
    stamp = param_stamp([w_p, w_dh])
    hit = _HEADS_PACKS.get([w_p, w_dh], stamp)
    if hit is not None:
        return hit
    Hn, NP, Hd = w_p.shape[1], w_p.shape[0], w_dh.shape[0]
    dev = w_p.device
    # wcat: B operand of dNSUM = [dP' (130 -> 160) | dHD0 (64)] . [W_p ; W_dh[:, :Hn]] -- rows = the 512 units, k-blocks 0-4 from W_p^T
    # (its rows beyond 130 zero), 5-6 from W_dh[:, :Hn]^T; straight from the fp32 masters (transposed reads), ONE launch for all six packs
    jobs, pk = [], {}

    def job(key, src, N, K, NT, KBtot, kb0=0, pairs=False, trans=False):
        if key not in pk:
            pk[key] = torch.empty(NT * KBtot * 512, device=dev, dtype=BF16)
        jobs.append((src.data_ptr(), src.stride(0), N, K, pk[key].data_ptr(), int(pairs), int(trans), NT, kb0, KBtot))
    w_dp = w_dh[:, Hn:]
    job('wcat', w_p, Hn, NP, Hn // 16, 7, 0, True, True)
    job('wcat', w_dh, Hn, Hd, Hn // 16, 7, 5, True, True)
    job('wp', w_p, NP, Hn, 9, Hn // 32)
    job('wdh', w_dh, Hd, Hn, Hd // 16, Hn // 32)
    job('wdp', w_dp, Hd, NP, Hd // 16, 5)
    job('wdpT', w_dp, NP, Hd, 9, Hd // 32, trans=True)
    pack_multi(jobs)
    return _HEADS_PACKS.put([w_p, w_dh], stamp, pk)


def row_gru_ok(prec, H, I, M, adt):
    """the H = 128 instance of the row-partitioned persistent GRU (dec_notes_emb_gru): worth it when the rows fill the chip"""
    return (NOTES_PERSIST and True and prec == 1 and BF16_STORAGE and H == 128 and I == 128
            and adt == BF16 and M >= 4096)


def notes_persist_ok(prec, Hn, E, gates_dtype=BF16):
    return NOTES_PERSIST and prec == 1 and BF16_STORAGE and Hn == 512 and E == 128 and gates_dtype == BF16


DEC_PARAM_NAMES = [
    'dec_init_input', 'dur_sos_token',
    'z2dec_hid_linear.weight', 'z2dec_hid_linear.bias', 'z2dec_in_linear.weight', 'z2dec_in_linear.bias',
    'dec_notes_emb_gru.weight_ih_l0', 'dec_notes_emb_gru.weight_hh_l0', 'dec_notes_emb_gru.bias_ih_l0',
    'dec_notes_emb_gru.bias_hh_l0', 'dec_notes_emb_gru.weight_ih_l0_reverse', 'dec_notes_emb_gru.weight_hh_l0_reverse',
    'dec_notes_emb_gru.bias_ih_l0_reverse', 'dec_notes_emb_gru.bias_hh_l0_reverse',
    'dec_time_gru.weight_ih_l0', 'dec_time_gru.weight_hh_l0', 'dec_time_gru.bias_ih_l0', 'dec_time_gru.bias_hh_l0',
    'dec_time_to_notes_hid.weight', 'dec_time_to_notes_hid.bias',
    'dec_notes_gru.weight_ih_l0', 'dec_notes_gru.weight_hh_l0', 'dec_notes_gru.bias_ih_l0', 'dec_notes_gru.bias_hh_l0',
    'pitch_out_linear.weight', 'pitch_out_linear.bias',
    'dec_dur_gru.weight_ih_l0', 'dec_dur_gru.weight_hh_l0', 'dec_dur_gru.bias_ih_l0', 'dec_dur_gru.bias_hh_l0',
    'dur_hid_linear.weight', 'dur_hid_linear.bias', 'dur_out_linear.weight', 'dur_out_linear.bias',
]

_CONST = {}


def _onehot2x5(dev):
    key = ('oh25', str(dev))
    if key not in _CONST:
        t = torch.zeros(2, 5, device=dev, dtype=F32)
        t[0, 0] = 1.0
        t[1, 1] = 1.0
        _CONST[key] = t
    return _CONST[key]


def _eye2(dev):
    key = ('eye2', str(dev))
    if key not in _CONST:
        _CONST[key] = torch.eye(2, device=dev, dtype=F32)
    return _CONST[key]


# The teacher-forced decoder forward behind ONE C entry point (ptv_decoder_tf_fwd, csrc/composite.hip): launch sequence, the persistent
# launch's turn and the shape decisions in C++; this side allocates the tensors and fills the two tables.  0 = sequence the launches here.
DEC_COMPOSITE = True
# the note tokens' gradient product (194 us, not an input of the time BPTT) on the sibling stream instead of in front of the BPTT on the
# chain: MEASURED SLOWER (9.0 vs 8.38 ms per step) -- the persistent time BPTT then starts earlier and runs beside more of the bulk
# products, and a persistent grid with company loses more than the chain gained.  Off.
CHD_COMPOSITE = True
_DTF = {}


def _decoder_tf_composite(ctx, z, emb, xs, force_dur, prec, P, W, params, B, R, E, He, Ht, Hn, Hd, NP):
    """-> the node's outputs when ptv_decoder_tf_fwd ran (ctx then holds exactly what the launch-by-launch path leaves on it), else None"""
    if 't' not in _DTF:
        from ._lib import header_enum
        _DTF['t'], _DTF['d'] = header_enum('PtvDtfTensor'), header_enum('PtvDtfDim')
    T_, D_ = _DTF['t'], _DTF['d']
    dev = z.device
    emb3 = emb.view(16, R, E)
    w16 = [W[n] for n in ('z2dec_hid_linear.weight', 'z2dec_in_linear.weight', 'dec_time_gru.weight_ih_l0', 'dec_time_gru.weight_hh_l0',
                          'dec_time_to_notes_hid.weight', 'dec_notes_gru.weight_ih_l0')]
    # (not inside ANY graph capture: the persistent launch's turn is a raw event pair here, and an event recorded before the capture
    # must not be waited for inside it -- the launch-by-launch path's wait_event() knows which edges a capture may keep)
    if (prec != 1 or not BF16_STORAGE or torch.cuda.is_current_stream_capturing() or any(w.dtype != BF16 for w in w16) or emb3.dtype != F32
            or not emb3.is_contiguous() or not FUSED_DUR or not HEADS_FUSED or not NOTES_PERSIST or _act_dtype(prec, Ht) != BF16
            or z.dtype != F32 or not persist_supported(1, B, Ht, 32)):
        return None
    Zs, Zi = z.shape[1], W['z2dec_in_linear.weight'].shape[0]
    dims = [0] * D_['PTV_DTF_D_COUNT']
    for k, v in (('B', B), ('E', E), ('HE', He), ('HT', Ht), ('HN', Hn), ('HD', Hd), ('NP', NP), ('ZS', Zs), ('ZI', Zi), ('LDP', _pad8(NP))):
        dims[D_['PTV_DTF_D_' + k]] = v
    darr = _larr(dims)
    if not lib().ptv_decoder_tf_supported(darr):
        return None
    M = 15 * R
    xs = xs.contiguous()
    NS, NS16 = _empty(33, B, Ht, dev=dev), _empty(33, B, Ht, dev=dev, dtype=BF16)
    z_in, TOKS = _empty(B, Zi, dev=dev), _empty(33, B, 2 * He, dev=dev)
    gi_t, zg = _empty(R, 3 * Ht, dev=dev, dtype=BF16), _empty(B, 3 * Ht, dev=dev, dtype=BF16)
    gates_t = _empty(32, 4, B, Ht, dev=dev, dtype=BF16)
    HN, HN16 = _empty(16, R, Hn, dev=dev), _empty(16, R, Hn, dev=dev, dtype=BF16)
    GC, gates_n = _empty(R, 3 * Hn, dev=dev, dtype=BF16), _empty(15, 4, R, Hn, dev=dev, dtype=BF16)
    pitch = _empty(M, _pad8(NP), dev=dev)[:, :NP]
    HD, HD16 = _empty(6, M, Hd, dev=dev), _empty(6, M, Hd, dev=dev, dtype=BF16)
    tab0, tab = _empty(1, 3 * Hd, dev=dev), _empty(2, 3 * Hd, dev=dev)
    gates_d = None if DUR_RECOMPUTE else _empty(5, 4, M, Hd, dev=dev, dtype=BF16)
    dur = _empty(M, 5, 2, dev=dev)
    idx = torch.empty(5, M, device=dev, dtype=torch.int32)
    xch = torch.empty(33 * B * Ht, device=dev, dtype=BF16)
    sync = _persist_sync(1, dev)
    pk = notes_packs(P['dec_notes_gru.weight_ih_l0'], P['dec_notes_gru.weight_hh_l0'], Ht)
    hp = heads_packs(P['pitch_out_linear.weight'], P['dur_hid_linear.weight'])
    cur = cur_stream()
    prev = _PERSIST_LAST.get(cur.device.index)
    done = torch.cuda.Event()
    tens = {'Z': z, 'EMB': emb3, 'XS': xs, 'FORCE_DUR': force_dur,
            'B_ZHID': P['z2dec_hid_linear.bias'], 'B_ZIN': P['z2dec_in_linear.bias'], 'INIT_INPUT': P['dec_init_input'],
            'B_IH_T': P['dec_time_gru.bias_ih_l0'], 'B_HH_T': P['dec_time_gru.bias_hh_l0'], 'B_T2N': P['dec_time_to_notes_hid.bias'],
            'B_IH_N': P['dec_notes_gru.bias_ih_l0'], 'B_HH_N': P['dec_notes_gru.bias_hh_l0'], 'B_P': P['pitch_out_linear.bias'],
            'B_DH': P['dur_hid_linear.bias'], 'W_HH_D': P['dec_dur_gru.weight_hh_l0'], 'B_HH_D': P['dec_dur_gru.bias_hh_l0'],
            'W_IH_D': P['dec_dur_gru.weight_ih_l0'], 'B_IH_D': P['dec_dur_gru.bias_ih_l0'], 'SOS': P['dur_sos_token'],
            'ONEHOT': _onehot2x5(dev), 'W_OUT_D': P['dur_out_linear.weight'], 'B_OUT_D': P['dur_out_linear.bias'],
            'W16_ZHID': w16[0], 'W16_ZIN': w16[1], 'W16_IH_T': w16[2], 'W16_HH_T': w16[3], 'W16_T2N': w16[4], 'W16_IH_N': w16[5],
            'PK_NOTES_H': pk['wg_h'], 'PK_NOTES_T': pk['wg_t'], 'PK_WP': hp['wp'], 'PK_WDH': hp['wdh'], 'PK_WDP': hp['wdp'],
            'NS': NS, 'NS16': NS16, 'Z_IN': z_in, 'TOKS': TOKS, 'GI_T': gi_t, 'ZG': zg, 'GATES_T': gates_t, 'HN': HN, 'HN16': HN16, 'GC': GC,
            'GATES_N': gates_n, 'PITCH': pitch, 'HD': HD, 'HD16': HD16, 'TAB0': tab0, 'TAB': tab, 'GATES_D': gates_d, 'DUR': dur, 'IDX': idx,
            'XCH': xch, 'SYNC': sync}
    live = live_top_for(dev)
    srt = _LIVE.get('sort') if live is not None else None
    if srt is not None and not (force_dur is None and decoder_bwd_composite_static_ok(prec, B, Ht, Hn, NP, Hd, E, P)):
        srt = None                                            # (the backward composite is the only un-sorter: no sorted forward without it)
    if srt is not None:
        tens.update(PERM=srt['perm'], ROW_LEN=srt['len'], NS16S=_empty(R, Ht, dev=dev, dtype=BF16), TOK_S=_empty(15, R, E, dev=dev), SEG_N=srt.get('seg_n'))
    if live is not None and POISON_DEAD_STEPS:
        _poison(HN16, gates_n, pitch, HD, HD16, gates_d, dur, idx)
        if srt is not None and srt.get('seg_n') is not None:
            _poison(tens['TOK_S'])                            # (gathered for the live blocks only: nobody may read the rest)
    slots = [None] * T_['PTV_DTF_COUNT']
    for k, v in tens.items():
        slots[T_['PTV_DTF_' + k]] = v.data_ptr() if v is not None else None
    slots[T_['PTV_DTF_LIVE_TOP']] = live.data_ptr() if live is not None else None
    # (the two event slots carry hipEvent_t handles, not tensors: events are created lazily -- record / wait once here to have a handle)
    if prev is not None:
        slots[T_['PTV_DTF_WAIT_EVENT']] = prev.cuda_event
    done.record(cur)                      # creates the handle; the library records it again after the persistent launch
    slots[T_['PTV_DTF_RECORD_EVENT']] = done.cuda_event
    _chain_prio()
    mark('dec_fwd:start')
    rc = lib().ptv_decoder_tf_fwd((ctypes.c_void_p * len(slots))(*slots), darr, stream_ptr())
    if rc == -3:
        return None
    check(rc, 'ptv_decoder_tf_fwd')
    _PERSIST_LAST[cur.device.index] = done
    mark('dec_fwd:heads')
    ctx.save_for_backward(z, emb, *params)
    ctx.emb_link = _EMB_LINK.get(emb.data_ptr()) if (ctx.needs_input_grad[2] and ctx.needs_input_grad[1] and emb.is_contiguous()) else None
    ctx.st = dict(B=B, R=R, E=E, He=He, Ht=Ht, Hn=Hn, Hd=Hd, NP=NP, prec=prec, NS=NS, z_in=z_in, NS16=NS16, HN16=HN16, HD16=HD16,
                  TOKS=TOKS, gates_t=gates_t, HN=HN, gates_n=gates_n, gates_n_rowk=True, pitch=pitch, HD=HD, gates_d=gates_d, idx=idx,
                  dur_tabs=(tab0, tab), dur16_only=True, live_top=live)
    if srt is not None:
        srt['used'] = True                # (the loss node now takes the targets in the same row order: _cached_targets)
        ctx.st['sorted'] = dict(perm=srt['perm'], len=srt['len'], NS16S=tens['NS16S'], TOK_S=tens['TOK_S'], seg_n=srt.get('seg_n'))
        _DTF['sorted_calls'] = _DTF.get('sorted_calls', 0) + 1
    _DTF['calls'] = _DTF.get('calls', 0) + 1
    ctx.mark_non_differentiable(idx)
    # (returned, never stored on ctx: outputs referenced from their own grad_fn are a cycle that only the garbage collector frees -- at an
    # arbitrary later moment, e.g. inside a graph capture, where releasing blocks that other streams used records events and kills the capture)
    return pitch.view(15, 32, B, NP), dur, idx


class DecoderTFFn(torch.autograd.Function):
    """(z [B,Zs], emb step-major [16,32,B,E], xs [32B, 2He] ground-truth note summaries (BiGruFinalFn over
    emb, ptvae.py:446-453), force_dur_idx or None, *params)
    -> pitch logits step-major [15,32,B,130], dur logits [15*32*B, 5, 2], dur argmax indices"""

    @staticmethod
    def forward(ctx, z, emb, xs, force_dur, prec, *params):
        P = dict(zip(DEC_PARAM_NAMES, params))
        W = {n: _W(p, prec) for n, p in P.items()}            # bf16 shadows of the weights as MFMA operands
        dev = z.device
        z = z.contiguous()
        B = z.shape[0]
        R = 32 * B
        E = emb.shape[-1]
        He = W['dec_notes_emb_gru.weight_hh_l0'].shape[1]
        Ht = W['dec_time_gru.weight_hh_l0'].shape[1]
        Hn = W['dec_notes_gru.weight_hh_l0'].shape[1]
        Hd = W['dec_dur_gru.weight_hh_l0'].shape[1]
        NP = W['pitch_out_linear.weight'].shape[0]              # 130
        S = ctx                                                  # stash everything on ctx

        if DEC_COMPOSITE:
            outs = _decoder_tf_composite(ctx, z, emb, xs, force_dur, prec, P, W, params, B, R, E, He, Ht, Hn, Hd, NP)
            if outs is not None:
                return outs

        # --- z -> initial time state, z_in  (ptvae.py:435-437)
        NS = _empty(33, B, Ht, dev=dev)
        gemm(z, W['z2dec_hid_linear.weight'], NS[0], bias=P['z2dec_hid_linear.bias'], prec=prec)
        z_in = gemm(z, W['z2dec_in_linear.weight'], bias=P['z2dec_in_linear.bias'], prec=prec)

        mark('dec_fwd:start')
        emb3 = emb.view(16, R, E)
        xs = xs.contiguous()

        # --- time GRU inputs: token_t = [init ; xs[t-1]], z_in broadcast over t  (ptvae.py:457-462,476-478)
        TOKS = _empty(33, B, 2 * He, dev=dev)
        copy2d(TOKS[0], P['dec_init_input'].view(1, -1), lds=0)
        copy2d(TOKS[1:].view(R, 2 * He), xs)
        w_ih_t = W['dec_time_gru.weight_ih_l0']
        adt_t = _act_dtype(prec, Ht)
        gi_t = gemm(TOKS[:32].view(R, 2 * He), w_ih_t[:, :2 * He], prec=prec, out_dtype=adt_t)  # [32*B, 3Ht]
        zg = gemm(z_in, w_ih_t[:, 2 * He:], bias=P['dec_time_gru.bias_ih_l0'], prec=prec, out_dtype=adt_t)
        gates_t = _empty(32, 4, B, Ht, dev=dev, dtype=_act_dtype(prec, Ht))
        NS16 = _hall16(prec, 33, B, Ht, dev)
        gru_fwd(prec, gi_t, B * 3 * Ht, 3 * Ht, W['dec_time_gru.weight_hh_l0'], P['dec_time_gru.bias_hh_l0'], NS,
                gates_t, gi2=zg, gi2_step=0, gi2_ld=3 * Ht, hall16=NS16)
        mark('dec_fwd:time_gru')
        NSf = NS[1:].view(R, Ht)                                               # notes_summary rows (t, b)
        NSf_op = NS16[1:].view(R, Ht) if NS16 is not None else NSf              # same values as an MFMA operand

        # --- notes GRU: h0 = dec_time_to_notes_hid(ns); input [ns | token], ns part hoisted (ptvae.py:374-398)
        HN = _empty(16, R, Hn, dev=dev)
        gemm(NSf_op, W['dec_time_to_notes_hid.weight'], HN[0], bias=P['dec_time_to_notes_hid.bias'], prec=prec)
        w_ih_n = W['dec_notes_gru.weight_ih_l0']
        adt = _act_dtype(prec, Hn)
        rowk = notes_persist_ok(prec, Hn, E, adt) and emb3.dtype == F32 and emb3.is_contiguous()
        # (the row kernel reads the hoisted part column-blocked by 16: one contiguous kilobyte per wave access instead of half cache lines)
        GC = gemm(NSf_op, w_ih_n[:, :Ht], bias=P['dec_notes_gru.bias_ih_l0'], prec=prec, out_dtype=adt, out_blocked=16 if rowk else False)      # [R, 3Hn]
        gates_n = _empty(15, 4, R, Hn, dev=dev, dtype=adt)
        HN16 = _hall16(prec, 16, R, Hn, dev)
        gates_n_rowk = rowk
        if rowk:
            # ONE launch for the 15 note steps, 64 rows per workgroup, token product fused (csrc/notes_persist.hip)
            pk = notes_packs(P['dec_notes_gru.weight_ih_l0'], P['dec_notes_gru.weight_hh_l0'], Ht)
            # the dead-step limit is taken only when EVERY later stage of the chain honours it (fused heads, fused duration GRU with
            # recomputed gates, and their backward halves): the generic heads / per-step duration GRU and their weight-gradient sums
            # read every row, so rows this launch leaves unwritten would meet zero gradients as NaN bit patterns (round-5 advice)
            chain_live = heads_ok(prec, Hn, NP, Hd, HN16, _act_dtype(prec, Hd) == BF16 or None) and prec == 1 and Hd == 64 and FUSED_DUR
            live = live_top_for(dev) if chain_live else None
            if live is not None and POISON_DEAD_STEPS:
                _poison(HN16, gates_n)
            call('ptv_notes_gru_persist_fwd_top', ptr(pk['wg_h']), ptr(pk['wg_t']), ptr(P['dec_notes_gru.bias_hh_l0']), ptr(GC), ptr(emb3),
                 ptr(HN), ptr(HN16), ptr(gates_n), R, 15, ptr(live), stream_ptr())
        else:
            live = None
            GT = gemm(emb3[:15].view(15 * R, E), w_ih_n[:, Ht:], prec=prec, out_dtype=adt)                    # [15R, 3Hn]
            gru_fwd(prec, GT, R * 3 * Hn, 3 * Hn, W['dec_notes_gru.weight_hh_l0'], P['dec_notes_gru.bias_hh_l0'], HN,
                    gates_n, gi2=GC, gi2_step=0, gi2_ld=3 * Hn, hall16=HN16)
        mark('dec_fwd:notes_gru')
        NSUM = HN[1:].view(15 * R, Hn)
        NSUM_op = HN16[1:].view(15 * R, Hn) if HN16 is not None else NSUM

        # --- pitch head + duration GRU initial state (ptvae.py:343-352)
        M = 15 * R
        # logits rows padded to a multiple of 8 floats: 130-wide rows would put every row of this tensor (an operand of
        # four more products) off the 16-byte grid and force element-wise loads / stores
        pitch = _empty(M, _pad8(NP), dev=dev)[:, :NP]
        HD = _empty(6, M, Hd, dev=dev)
        HD16 = _hall16(prec, 6, M, Hd, dev)
        fused_heads = heads_ok(prec, Hn, NP, Hd, HN16, HD16)
        if fused_heads:
            # ONE pass over the note summaries for both Linears; the logits feed the second product from LDS (csrc/heads.hip)
            hp = heads_packs(P['pitch_out_linear.weight'], P['dur_hid_linear.weight'])
            if live is not None and POISON_DEAD_STEPS:
                _poison(pitch, HD, HD16)
            call('ptv_heads_fwd_top', ptr(NSUM_op), ptr(hp['wp']), ptr(hp['wdh']), ptr(hp['wdp']), ptr(P['pitch_out_linear.bias']),
                 ptr(P['dur_hid_linear.bias']), ptr(pitch), pitch.stride(0), ptr(HD[0]), ptr(HD16[0]), M, ptr(live), R, stream_ptr())
        else:
            gemm(NSUM_op, W['pitch_out_linear.weight'], pitch, bias=P['pitch_out_linear.bias'], prec=prec)          # [M,130]
            w_dh = W['dur_hid_linear.weight']
            gemm(NSUM_op, w_dh[:, :Hn], HD[0], bias=P['dur_hid_linear.bias'], prec=prec)
            gemm(pitch, w_dh[:, Hn:], HD[0], acc=True, prec=prec)

        mark('dec_fwd:heads')
        # --- 5-step duration GRU with argmax feedback (ptvae.py:353-367)
        w_ih_d, b_ih_d = W['dec_dur_gru.weight_ih_l0'], P['dec_dur_gru.bias_ih_l0']
        tab0 = gemm(P['dur_sos_token'].view(1, -1), w_ih_d, bias=b_ih_d, prec=0)       # [1, 3Hd]  (tiny: exact)
        tab = gemm(_onehot2x5(dev), w_ih_d, bias=b_ih_d, prec=0)                       # [2, 3Hd]
        fused_dur = prec == 1 and Hd == 64 and FUSED_DUR
        gates_d = None if (fused_dur and DUR_RECOMPUTE and HD16 is not None) else _empty(5, 4, M, Hd, dev=dev, dtype=_act_dtype(prec, Hd))
        dur = _empty(M, 5, 2, dev=dev)
        idx = torch.empty(5, M, device=dev, dtype=torch.int32)
        dur2 = dur.view(M, 10)
        if fused_dur:
            # one kernel for the 5 steps + output layer + argmax feedback (dur.hip)
            live_d = live                                   # (None unless the fused heads took the limit too: chain_live above)
            if live_d is not None and POISON_DEAD_STEPS:
                _poison(gates_d, dur, idx)
            call('ptv_dur_gru_fwd_top', Hd, M, ptr(HD[0]), Hd, ptr(P['dec_dur_gru.weight_hh_l0']), ptr(P['dec_dur_gru.bias_hh_l0']),
                 ptr(tab0), ptr(tab), ptr(P['dur_out_linear.weight']), ptr(P['dur_out_linear.bias']),
                 None if HD16 is not None else ptr(HD[1]), M * Hd,          # fp32 states stay in registers when the bf16
                 ptr(HD16[1]) if HD16 is not None else None, ptr(gates_d), M * Hd, 4 * M * Hd, _bf(gates_d),   # copies exist
                 ptr(dur2), 10, ptr(idx), M, ptr(force_dur) if force_dur is not None else None, M, ptr(live_d), R, stream_ptr())
            if HD16 is not None and not fused_heads:
                call('ptv_cast_bf16', ptr(HD[0]), ptr(HD16[0]), M * Hd, stream_ptr())       # slot 0 of the shadow
        else:
            for d in range(5):
                gi, gi_ld, gi_idx = (tab0, 0, None) if d == 0 else (tab, 3 * Hd, idx[d - 1])
                gru_fwd(prec, gi, 0, gi_ld, P['dec_dur_gru.weight_hh_l0'], P['dec_dur_gru.bias_hh_l0'], HD[d:d + 2],
                        gates_d[d], gi_idx=gi_idx, T=1, hall16=HD16[d:d + 2] if HD16 is not None else None, skip_cast0=d > 0)
                call('ptv_dur_out_token', ptr(HD[d + 1]), Hd, ptr(P['dur_out_linear.weight']), ptr(P['dur_out_linear.bias']),
                     ptr(dur2[:, 2 * d:]), 10, ptr(idx[d]), ptr(force_dur[d]) if force_dur is not None else None, M,
                     stream_ptr())

        S.save_for_backward(z, emb, *params)
        # (see _EMB_LINK: only when the summaries really are a function of this very embedding and will receive a gradient from this node)
        S.emb_link = _EMB_LINK.get(emb.data_ptr()) if (S.needs_input_grad[2] and S.needs_input_grad[1] and emb.is_contiguous()) else None
        S.st = dict(B=B, R=R, E=E, He=He, Ht=Ht, Hn=Hn, Hd=Hd, NP=NP, prec=prec, NS=NS, z_in=z_in, NS16=NS16, HN16=HN16,
                    HD16=HD16,
                    TOKS=TOKS, gates_t=gates_t, HN=HN, gates_n=gates_n, gates_n_rowk=gates_n_rowk, pitch=pitch, HD=HD, gates_d=gates_d, idx=idx,
                    dur_tabs=(tab0, tab), live_top=live,
                    dur16_only=bool(prec == 1 and Hd == 64 and FUSED_DUR and HD16 is not None))   # HD[1:] never written
        S.mark_non_differentiable(idx)
        return pitch.view(15, 32, B, NP), dur, idx

    @staticmethod
    def backward(ctx, dpitch, ddur, _didx):
        z, emb, *params = ctx.saved_tensors
        P = dict(zip(DEC_PARAM_NAMES, params))
        st = ctx.st
        ctx.st = None
        R, E = st['R'], st['E']
        dz, demb, dTOKS, G, side = decoder_bwd_core(P, st, z, emb.view(16, R, E)[:15].view(15 * R, E), dpitch, ddur)
        mark('dec_bwd:end')
        # parameter gradients only: joined when the backward pass ends -- but only if autograd ADOPTS the tensors
        # (p.grad is None and the buffer is this step's arena view); an accumulation `p.grad += g` would run on this
        # node's stream without a dependency on the side stream
        from .optim import is_arena_view
        if all(G[n] is None or (P[n].grad is None and is_arena_view(P[n], G[n])) for n in DEC_PARAM_NAMES):
            streams = (side.s,)
            side.defer()
            if GRAD_READY_HOOK is not None:               # data parallel: this slice of the gradient bucket can leave now (dist.GradSync)
                GRAD_READY_HOOK([P[n] for n in DEC_PARAM_NAMES if G[n] is not None], streams)
        else:
            side.join()
        B, He = st['B'], st['He']
        if st.get('ev_dtoks') is not None:
            if len(_SUMMARY_EV) > 8:
                _SUMMARY_EV.clear()
            _SUMMARY_EV[dTOKS[1:].data_ptr()] = st['ev_dtoks']
        demb_out = demb.view(16, 32, B, E)
        if ctx.emb_link is not None and ctx.needs_input_grad[2] and demb.dtype == F32 and demb.is_contiguous():
            ctx.emb_link['demb'] = demb                   # the summary node accumulates into it and returns it (BiGruFinalFn.backward)
            demb_out = None
        return (dz, demb_out, dTOKS[1:].view(R, 2 * He), None, None) + tuple(G[n] for n in DEC_PARAM_NAMES)


# decoder_bwd_core's fused bf16 path through ptv_decoder_tf_bwd (one C call: ~50 launches, four forks, the persistent turn);
# PTV_BWD_COMPOSITES=0: both backward composites off -- the same launches sequenced from Python (bit-identical)
DEC_BWD_COMPOSITE = os.environ.get('PTV_BWD_COMPOSITES', '1') != '0'
_DTB = {}
_DTB_G = (('W_ZHID', 'z2dec_hid_linear.weight'), ('B_ZHID', 'z2dec_hid_linear.bias'), ('W_ZIN', 'z2dec_in_linear.weight'),
          ('B_ZIN', 'z2dec_in_linear.bias'), ('INIT_INPUT', 'dec_init_input'), ('W_IH_T', 'dec_time_gru.weight_ih_l0'),
          ('W_HH_T', 'dec_time_gru.weight_hh_l0'), ('B_IH_T', 'dec_time_gru.bias_ih_l0'), ('B_HH_T', 'dec_time_gru.bias_hh_l0'),
          ('W_T2N', 'dec_time_to_notes_hid.weight'), ('B_T2N', 'dec_time_to_notes_hid.bias'), ('W_IH_N', 'dec_notes_gru.weight_ih_l0'),
          ('W_HH_N', 'dec_notes_gru.weight_hh_l0'), ('B_IH_N', 'dec_notes_gru.bias_ih_l0'), ('B_HH_N', 'dec_notes_gru.bias_hh_l0'),
          ('W_P', 'pitch_out_linear.weight'), ('B_P', 'pitch_out_linear.bias'), ('W_DH', 'dur_hid_linear.weight'),
          ('B_DH', 'dur_hid_linear.bias'), ('W_OUT_D', 'dur_out_linear.weight'), ('B_OUT_D', 'dur_out_linear.bias'),
          ('W_IH_D', 'dec_dur_gru.weight_ih_l0'), ('W_HH_D', 'dec_dur_gru.weight_hh_l0'), ('B_IH_D', 'dec_dur_gru.bias_ih_l0'),
          ('B_HH_D', 'dec_dur_gru.bias_hh_l0'), ('SOS', 'dur_sos_token'))


def decoder_bwd_composite_static_ok(prec, B, Ht, Hn, NP, Hd, E, P=None):
    """what _decoder_bwd_composite will ask of the configuration (not of the gradients it is handed): a forward on length-sorted rows
    commits the backward to the composite -- the only place that scatters the row order back"""
    return (prec == 1 and ZERO_SKIP and OVERLAP and SUMMARY_FAMILY_SLOT < 0 and DEC_BWD_COMPOSITE and not torch.cuda.is_current_stream_capturing()
            and DUR_RECOMPUTE and HEADS_WGRAD_FUSED and HEADS_FUSED and BF16_STORAGE and (Hn, NP, Hd) == (512, 130, 64)
            and notes_persist_ok(prec, Hn, E, BF16) and persist_supported(1, B, Ht, 32) and 15 * 32 * B * 5 >= 4096 and DP_INPLACE
            and (P is None or all(_WT(P[n], prec) is not None for n in (
                'dec_notes_gru.weight_ih_l0', 'dec_time_to_notes_hid.weight', 'dec_time_gru.weight_ih_l0', 'dec_time_gru.weight_hh_l0',
                'z2dec_hid_linear.weight', 'z2dec_in_linear.weight'))))


def _decoder_bwd_composite(P, st, z, tok_op, dP, ddur, top_h, side, G):
    """-> decoder_bwd_core's result tuple when ptv_decoder_tf_bwd ran the whole sequence, else None (the caller sequences it: same bits)"""
    if 't' not in _DTB:
        from ._lib import header_enum
        _DTB['t'], _DTB['d'] = header_enum('PtvDtbTensor'), header_enum('PtvDtbDim')
    T_, D_ = _DTB['t'], _DTB['d']
    B, R, E, He, Ht, Hn, Hd, NP, prec = (st[k] for k in ('B', 'R', 'E', 'He', 'Ht', 'Hn', 'Hd', 'NP', 'prec'))
    dev = z.device
    M = 15 * R
    HN16, HD16, NS16, gates_n, gates_t = st.get('HN16'), st.get('HD16'), st.get('NS16'), st['gates_n'], st['gates_t']
    if (prec != 1 or not ZERO_SKIP or not OVERLAP or SUMMARY_FAMILY_SLOT >= 0 or torch.cuda.is_current_stream_capturing()
            or not st.get('dur16_only') or st['gates_d'] is not None or st.get('dur_tabs') is None or HD16 is None or HN16 is None
            or NS16 is None or not st.get('gates_n_rowk') or gates_n.dtype != BF16 or gates_t.dtype != BF16 or not HEADS_WGRAD_FUSED
            or not heads_ok(prec, Hn, NP, Hd, HN16, HD16) or not notes_persist_ok(prec, Hn, E, BF16)
            or dP.stride(0) != _pad8(NP) or dP.data_ptr() % 16 or not ddur.is_contiguous() or ddur.dtype != F32
            or tok_op.dtype != F32 or not tok_op.is_contiguous() or z.dtype != F32 or not z.is_contiguous()
            or not persist_supported(1, B, Ht, 32) or side.s == side.main or M * 5 < 4096):       # (M * 5 >= 4096: _bgrad's 64-column path, as in C)
        _defer_flush()
        return None
    wts = [_WT(P[n], prec) for n in ('dec_notes_gru.weight_ih_l0', 'dec_time_to_notes_hid.weight', 'dec_time_gru.weight_ih_l0',
                                      'dec_time_gru.weight_hh_l0', 'z2dec_hid_linear.weight', 'z2dec_in_linear.weight')]
    # (every reason to decline is checked HERE, before a gradient buffer is taken from the arena: GradArena.take hands a view out once per
    # step, a fallback after it would get fresh non-arena buffers and the deferred join would turn into a join -- round-5 advice)
    if any(w is None for w in wts) or st['pitch'].stride(0) != _pad8(NP) or st['idx'].dtype != torch.int32:
        _defer_flush()
        return None
    Zs, Zi = z.shape[1], st['z_in'].shape[1]
    S = persist_splitk(1, B, Ht)
    nblk = min(256, (M + 63) // 64)
    psz = lib().ptv_dur_gru_bwd_part_size()
    dims = [0] * D_['PTV_DTB_D_COUNT']
    for k, v in (('B', B), ('E', E), ('HE', He), ('HT', Ht), ('HN', Hn), ('HD', Hd), ('NP', NP), ('ZS', Zs), ('ZI', Zi), ('LDP', _pad8(NP)),
                 ('NBLK', nblk), ('SPLITK', S)):
        dims[D_['PTV_DTB_D_' + k]] = v
    hp = heads_packs(P['pitch_out_linear.weight'], P['dur_hid_linear.weight'])
    pk = notes_packs(P['dec_notes_gru.weight_ih_l0'], P['dec_notes_gru.weight_hh_l0'], Ht)
    for _, n in _DTB_G:
        G[n] = _gbuf(P[n])
    dz, dtok, dTOKS = _empty(B, Zs, dev=dev), _empty(16, R, E, dev=dev), _empty(33, B, 2 * He, dev=dev)
    tab0, tab = st['dur_tabs']
    tens = {'Z': z, 'TOK_OP': tok_op, 'DP': dP, 'DDUR': ddur, 'TOP_H': top_h,
            'W_HH_D': P['dec_dur_gru.weight_hh_l0'], 'B_HH_D': P['dec_dur_gru.bias_hh_l0'], 'W_IH_D': P['dec_dur_gru.weight_ih_l0'],
            'W_OUT_D': P['dur_out_linear.weight'], 'SOS': P['dur_sos_token'], 'PK_WDPT': hp['wdpT'], 'PK_WCAT': hp['wcat'],
            'PK_NOTES_WT': pk['wt'], 'WT_IH_N': wts[0], 'WT_T2N': wts[1], 'WT_IH_T': wts[2], 'WT_HH_T': wts[3], 'WT_ZHID': wts[4],
            'WT_ZIN': wts[5], 'NS': st['NS'], 'NS16': NS16, 'Z_IN': st['z_in'], 'TOKS': st['TOKS'], 'GATES_T': gates_t, 'HN16': HN16,
            'GATES_N': gates_n, 'PITCH': st['pitch'], 'HD16': HD16, 'TAB0': tab0, 'TAB': tab, 'IDX': st['idx'],
            'DZ': dz, 'DTOK': dtok, 'DTOKS': dTOKS,
            'DHD0': _empty(M, Hd, dev=dev), 'PART': _empty(nblk, psz, dev=dev), 'S': _zeros(1, psz, dev=dev), 'TMP64': _zeros(1, 64, dev=dev),
            'DNSUM': _empty(M, Hn, dev=dev, dtype=BF16), 'DY16': _empty(M, 200, dev=dev, dtype=BF16), 'TMP200': _empty(200, Hn, dev=dev),
            'CS200': _zeros(200, dev=dev), 'DGI_N': _empty(15, R, 3 * Hn, dev=dev, dtype=BF16), 'DGH_N': _empty(15, R, Hn, dev=dev, dtype=BF16),
            'DHN0': _empty(R, Hn, dev=dev), 'SCRATCH_N': _empty(lib().ptv_notes_gru_persist_scratch_elems(R), dev=dev, dtype=BF16),
            'TOP_STEP': _ineg1(dev), 'DGC': _empty(R, 3 * Hn, dev=dev), 'DNS': _empty(R, Ht, dev=dev),
            'DGI_T': _empty(32, B, 3 * Ht, dev=dev, dtype=BF16), 'DGH_T': _empty(32, B, 3 * Ht, dev=dev, dtype=BF16),
            'DZHID': _empty(B, Ht, dev=dev), 'DZG': _empty(B, 3 * Ht, dev=dev), 'DZ_IN': _empty(B, Zi, dev=dev),
            'XCH': torch.empty(32 * B * 3 * Ht, device=dev, dtype=BF16),
            'PART_T': torch.empty(lib().ptv_gru_persist_part_elems(1, B, Ht, S), device=dev) if S else None,
            'SYNC': _persist_sync(1, dev)}
    srt = st.get('sorted')
    if srt is not None:                                   # the forward ran on length-sorted rows: operands in that order, dNS / dtok scattered back
        tens.update(PERM=srt['perm'], ROW_LEN=srt['len'], NS16S=srt['NS16S'], TOK_OP=srt['TOK_S'].view(M, E), DNS_S=_empty(R, Ht, dev=dev),
                    DTOK_S=_empty(15, R, E, dev=dev), SEG_N=srt.get('seg_n'))
    if POISON_DEAD_STEPS and top_h is not None and st.get('live_top') is not None:        # (tests: whatever reads a dead row of these gets NaN -- heads_bwd / the BPTT leave them unwritten)
        _poison(tens['DNSUM'], tens['DGI_N'], tens['DGH_N'], tens['DY16'])
    slots = [None] * T_['PTV_DTB_COUNT']
    for k, v in tens.items():
        slots[T_['PTV_DTB_' + k]] = ptr(v)
    for k, n in _DTB_G:
        slots[T_['PTV_DTB_G_' + k]] = ptr(G[n])
    cur = cur_stream()
    evs = _DTB.get(('ev', cur.device.index))
    if evs is None:                                   # the four fork events, created once (a wait takes the record that precedes it)
        evs = [torch.cuda.Event() for _ in range(4)]
        for e in evs:
            e.record(cur)
        _DTB[('ev', cur.device.index)] = evs
    for i, e in enumerate(evs):
        slots[T_['PTV_DTB_FORK_EVENT%d' % i]] = e.cuda_event
    slots[T_['PTV_DTB_SIDE_STREAM']] = side.s.cuda_stream
    prev = _PERSIST_LAST.get(cur.device.index)
    done = torch.cuda.Event()
    if prev is not None:
        slots[T_['PTV_DTB_WAIT_EVENT']] = prev.cuda_event
    done.record(cur)                      # creates the handle; the library records it again after the persistent launch
    slots[T_['PTV_DTB_RECORD_EVENT']] = done.cuda_event
    mark('dec_bwd:composite')
    arr_, darr_, sp_ = (ctypes.c_void_p * len(slots))(*slots), _larr(dims), stream_ptr()
    # (payload: the tables AND everything they point to that nothing else keeps alive until a collected call runs -- the tensors, and the
    # previous persistent launch's event, which _PERSIST_LAST drops below: a destroyed hipEvent_t in the table was a segfault)
    rc = _defer_or_run('tf_bwd', (arr_, darr_, tens, prev, done, evs), lambda: lib().ptv_decoder_tf_bwd(arr_, darr_, sp_))
    _SIDE_DEPTH[1] = 1                    # (the library's priority state as the call leaves it)
    if rc == -3:                          # (the checks above mirror ptv_decoder_tf_bwd's: a late refusal would leave taken arena views behind)
        raise RuntimeError('ptv_decoder_tf_bwd refused a configuration its Python-side checks accepted')
    check(rc, 'ptv_decoder_tf_bwd')
    _PERSIST_LAST[cur.device.index] = done
    _DTB['calls'] = _DTB.get('calls', 0) + 1
    st['ev_dtoks'] = None
    # the sibling stream's products are queued, not run: everything they read or write stays referenced until the join (the caller defers
    # it to the end of the backward pass) -- EXCEPT the gradient buffers (a second reference makes AccumulateGrad clone them)
    side.used = True
    side.keep.extend([v for k, v in tens.items() if v is not None] + [st, z, tok_op])
    return dz, dtok, dTOKS, G, side


def decoder_bwd_core(P, st, z, tok_op, dpitch, ddur):
    """BPTT of the PianoTree decoder given the saved forward state `st` -- shared by the teacher-forced node (DecoderTFFn) and
    the step-loop node (functional_free.DecoderStepFn: argmax is not differentiable, so with the fed tokens recorded every
    chain is the same batched BPTT).  Chain (duration GRU -> heads -> notes GRU -> time GRU -> z) on the caller's stream; every
    weight / bias gradient product is enqueued on a sibling stream as soon as its operands exist, so the K-deep dW GEMMs
    overlap the latency-bound recurrent steps.
      tok_op [15*R, E]: the note tokens that were FED to the notes GRU (ground-truth embedding rows, or the recorded mix)
    -> (dz, dtok [16,R,E] gradient w.r.t. the fed note tokens (slot 15 zero), dTOKS [33,B,2He] gradient w.r.t. the time-step
        tokens, G parameter gradients by name, side stream handle -- the caller joins or defers it)"""
    B, R, E, He, Ht, Hn, Hd, NP, prec = (st[k] for k in ('B', 'R', 'E', 'He', 'Ht', 'Hn', 'Hd', 'NP', 'prec'))
    W = P                                             # K-major (dX) products read the fp32 weights
    dev = z.device
    M = 15 * R
    G = {n: None for n in DEC_PARAM_NAMES}
    NS, HN, HD, TOKS = st['NS'], st['HN'], st['HD'], st['TOKS']
    # bf16 shadows of the state buffers (bf16 precision) as the activation operands of the dW products
    NSo = st['NS16'] if st.get('NS16') is not None else NS
    HNo = st['HN16'] if st.get('HN16') is not None else HN
    HDo = st['HD16'] if st.get('HD16') is not None else HD
    NSf_op, NSUM_op = NSo[1:].view(R, Ht), HNo[1:].view(M, Hn)
    side = Side(DEC_WGRAD_SLOT)

    def wgrad(name, dy, x, sub=None):
        """G[name][:, sub] += dy^T . x"""
        if G[name] is None:
            G[name] = _gbuf(P[name])
        out = G[name] if sub is None else G[name][:, sub]
        gemm(dy, x, out, ta=True, tb=True, acc=True, prec=prec)

    def bgrad(name, a):
        G[name] = _bgrad(P[name], a)

    def wgrad_b(name, bname, dy, x, sub=None, k_top=None):
        """G[name][:, sub] += dy^T . x and (bname) G[bname] += column sums of dy, one pass over dy; k_top: the rows of dy after note
        step k_top (device int) are zero"""
        if G[name] is None:
            G[name] = _gbuf(P[name])
        if bname is not None and G[bname] is None:
            G[bname] = _gbuf(P[bname])
        wgrad_bias(dy, x, G[name] if sub is None else G[name][:, sub], G[bname] if bname is not None else None, prec, k_top, R)

    hint = _loss_top_hint(dpitch, ddur)
    ddur = (ddur.contiguous() if ddur is not None else _zeros(M, 5, 2, dev=dev)).view(M, 10)
    if (DP_INPLACE and dpitch is not None and dpitch.dtype == F32 and _row_dense(dpitch) and dpitch.stride(-2) == _pad8(NP)
            and dpitch.numel() == M * NP and not dpitch.requires_grad):
        # the loss node hands the logits' gradient over in the logits' own row-padded layout: it becomes the accumulator of the head
        # chain as it is (it has no other consumer; a `retain_grad()` on the logits would see the accumulated values -- PTV_DP_INPLACE=0)
        dP = _rows2d(dpitch)
    else:
        dP = _empty(M, _pad8(NP), dev=dev)[:, :NP]           # row-padded like the logits
        if dpitch is not None:
            copy2d(dP, _rows2d(dpitch))
        else:
            copy2d(dP, _zeros(1, NP, dev=dev), lds=0)

    # the loss ignores the padded note slots (the late note steps of every row): find the last note step that received any gradient
    # (on the gradients themselves) -- the head products below stop there, as the BPTT does on its own
    mark('dec_bwd:start')
    zero_skip_sync()
    top_h = None
    if ZERO_SKIP:
        if st.get('live_top') is not None:
            # the forward stopped at this note step (DisentangleVAE.loss(): the loss ignores everything after it, its gradient there is
            # exactly zero): the limit of everything below -- a scan may report MORE (row padding that is not zero), and the forward
            # tensors hold nothing beyond it
            top_h = st['live_top']
        elif hint is not None:
            top_h = hint                         # the loss node's own bound (its gradients are zero beyond it by construction)
        else:
            top_h = _ineg1(dev)
            call('ptv_last_nonzero_unit', ptr(dP), M, NP, dP.stride(0), R, ptr(top_h), stream_ptr())
            call('ptv_last_nonzero_unit', ptr(ddur), M, 10, 10, R, ptr(top_h), stream_ptr())

    if DEC_BWD_COMPOSITE:
        res = _decoder_bwd_composite(P, st, z, tok_op, dP, ddur, top_h, side, G)
        if res is not None:
            return res
    if st.get('sorted') is not None:
        raise RuntimeError('the decoder forward ran on length-sorted rows (PTV_SORT_DEC_ROWS) but ptv_decoder_tf_bwd, the only place that '
                           'restores the row order, declined this backward pass')
    _defer_flush()                        # (the launch-by-launch sequencing below runs at once)

    # ---- duration GRU (5 steps) ----
    w_out = P['dur_out_linear.weight']
    w_hh_d, w_ih_d = W['dec_dur_gru.weight_hh_l0'], W['dec_dur_gru.weight_ih_l0']
    if dur_bwd_fusable(prec, Hd, st['gates_d']) or st.get('dur16_only'):
        dHD0 = dur_bwd_fused(P, G, st['gates_d'], st['idx'], HD, HDo, ddur, wgrad, bgrad, side, tabs=st.get('dur_tabs'))
    else:
        dgi_d, dgh_d, dHD0 = gru_bwd(prec, HD, st['gates_d'], w_hh_d, lr=(ddur, 2, 10, 2, w_out))

        def dur_wgrads():
            for d in range(5):
                wgrad('dur_out_linear.weight', ddur[:, 2 * d:2 * d + 2], HDo[d + 1])
            bgrad('dur_out_linear.bias', ddur.view(M * 5, 2))
            wgrad('dec_dur_gru.weight_hh_l0', dgh_d.view(5 * M, 3 * Hd), HDo[:5].view(5 * M, Hd))
            bgrad('dec_dur_gru.bias_hh_l0', dgh_d.view(5 * M, 3 * Hd))
            bgrad('dec_dur_gru.bias_ih_l0', dgi_d.view(5 * M, 3 * Hd))
            cs0 = colsum(_zeros(1, 3 * Hd, dev=dev), dgi_d[0])                   # step 0: dense <sos> token
            g = _gbuf(P['dec_dur_gru.weight_ih_l0'])
            gemm(cs0, P['dur_sos_token'].view(1, -1), g, ta=True, tb=True, acc=True, prec=0, splitk=-1)
            G['dur_sos_token'] = _gbuf(P['dur_sos_token'])
            gemm(cs0, w_ih_d, G['dur_sos_token'].view(1, -1), tb=True, prec=0, splitk=-1)
            sel = _zeros(2, 3 * Hd, dev=dev)                                     # steps 1..4: one-hot tokens {0,1}
            for d in range(1, 5):
                colsum(sel, dgi_d[d], sel=st['idx'][d - 1], groups=2)
            gemm(sel, _eye2(dev), g[:, 0:2], ta=True, acc=True, prec=0, splitk=-1)
            G['dec_dur_gru.weight_ih_l0'] = g
        side(dur_wgrads, ddur, dgi_d, dgh_d)

    mark('dec_bwd:dur_bptt')
    # ---- dur_hid_linear([note_summary | est_pitch]) and pitch_out_linear ----
    w_dh, w_p = W['dur_hid_linear.weight'], W['pitch_out_linear.weight']
    # gradient reaching the notes-GRU states: only ever an addend of the BPTT epilogue -> activation dtype
    dNSUM = _empty(M, Hn, dev=dev, dtype=_act_dtype(prec, Hn))
    if POISON_DEAD_STEPS and top_h is not None and st.get('live_top') is not None:
        _poison(dNSUM)
    # (read by the row-partitioned BPTT kernel column-blocked by 32, like its saved gates: whole-kilobyte wave accesses)
    rowk_bwd = bool(st.get('gates_n_rowk') and notes_persist_ok(prec, Hn, E, st['gates_n'].dtype) and dNSUM.dtype == BF16 and HN.dtype == F32)
    fused_heads = (heads_ok(prec, Hn, NP, Hd, st.get('HN16'), st.get('HD16')) and dNSUM.dtype == BF16 and dP.stride(0) % 4 == 0
                   and dP.data_ptr() % 16 == 0 and dHD0.dtype == F32 and dHD0.is_contiguous())
    dY16 = None
    if fused_heads:
        # dP += dHD0 . W_dh[:, Hn:] and dNSUM = dP . W_p + dHD0 . W_dh[:, :Hn] in one pass over dP / dHD0 (csrc/heads.hip)
        hp = heads_packs(P['pitch_out_linear.weight'], P['dur_hid_linear.weight'])
        dY16 = _empty(M, 200, dev=dev, dtype=BF16) if HEADS_WGRAD_FUSED else None
        # (with a limit and the row BPTT as the consumer -- it gets the same limit as its bound -- dNSUM's dead rows stay unwritten: bit 1)
        bound_n = top_h if (rowk_bwd and ZERO_SKIP) else None
        call('ptv_heads_bwd', ptr(dP), dP.stride(0), ptr(dHD0), ptr(hp['wdpT']), ptr(hp['wcat']), ptr(dNSUM),
             int(rowk_bwd) | (2 if bound_n is not None else 0), ptr(dY16), ptr(top_h), R if top_h is not None else 0, M, stream_ptr())
    else:
        gemm_dx(dHD0, w_dh, slice(Hn, None), out=dP, acc=True, prec=prec, m_top=top_h, m_unit=R)         # dP complete

    def head_wgrads():
        if fused_heads and dY16 is not None:
            # [dP | dHD0]^T . note summaries as ONE product (the summaries are read once, the gradients as bf16), then four tiny scatters
            tmp, cs = _empty(200, Hn, dev=dev), _zeros(200, dev=dev)
            _chain_prio()
            call('ptv_wgrad', 200, Hn, M, ptr(dY16), 200, ptr(NSUM_op), _ld(NSUM_op), ptr(tmp), Hn, 1.0, 0, 3, 0, ptr(cs), ptr(top_h),
                 R if top_h is not None else 0, 0, stream_ptr())
            for name, bname, lo, hi, sub in (('pitch_out_linear.weight', 'pitch_out_linear.bias', 0, NP, None),
                                             ('dur_hid_linear.weight', 'dur_hid_linear.bias', 136, 136 + Hd, slice(0, Hn))):
                if G[name] is None:
                    G[name] = _gbuf(P[name])
                if G[bname] is None:
                    G[bname] = _gbuf(P[bname])
                copy2d(G[name] if sub is None else G[name][:, sub], tmp[lo:hi], acc=True)
                copy2d(G[bname].view(1, -1), cs[lo:hi].view(1, -1), acc=True)
            # (from the fp32 dHD0, whose dead rows are real zeros: N = 130 sends the last <= 32 rows through the guarded tail launch, which
            # knows no row limit -- the dead rows of dY16 are never written)
            wgrad_b('dur_hid_linear.weight', None, dHD0, st['pitch'], slice(Hn, None), top_h)
            return
        wgrad_b('dur_hid_linear.weight', 'dur_hid_linear.bias', dHD0, NSUM_op, slice(0, Hn), top_h)
        wgrad_b('dur_hid_linear.weight', None, dHD0, st['pitch'], slice(Hn, None), top_h)
        wgrad_b('pitch_out_linear.weight', 'pitch_out_linear.bias', dP, NSUM_op, None, top_h)
    if not fused_heads:
        gemm_dx(dHD0, w_dh, slice(0, Hn), out=dNSUM, prec=prec, m_top=top_h, m_unit=R, out_blocked=rowk_bwd)                   # [M, Hn]
        gemm_dx(dP, w_p, out=dNSUM, acc=True, prec=prec, m_top=top_h, m_unit=R, out_blocked=rowk_bwd)
    # (forked after the chain's dX products are queued: forking as soon as the operands exist removes a false dependency and measured 0.06 ms
    # SLOWER -- the products then compete with the chain for CUs)
    side(head_wgrads, dHD0, dP, dY16)

    mark('dec_bwd:head_dx')
    # ---- notes GRU (15 steps, batch 32*B) ----
    w_hh_n, w_ih_n = W['dec_notes_gru.weight_hh_l0'], W['dec_notes_gru.weight_ih_l0']
    if rowk_bwd:
        # (the forward ran on the row kernel: its gate planes are in that kernel pair's private layout)
        pk = notes_packs(w_ih_n, w_hh_n, Ht)
        dgi_n = _empty(15, R, 3 * Hn, dev=dev, dtype=BF16)
        dgh_n = _empty(15, R, Hn, dev=dev, dtype=BF16)          # n third only: the r / z thirds of dgh are dgi's
        if POISON_DEAD_STEPS and top_h is not None and st.get('live_top') is not None:
            _poison(dgi_n, dgh_n)
        dHN0 = _empty(R, Hn, dev=dev)
        scratch = _empty(lib().ptv_notes_gru_persist_scratch_elems(R), dev=dev, dtype=BF16)
        top_step = _ineg1(dev) if ZERO_SKIP else None   # <- last note step with a gradient
        call('ptv_notes_gru_persist_bwd_top', ptr(pk['wt']), ptr(st['HN16']), ptr(st['gates_n']), ptr(dNSUM), ptr(dgi_n), ptr(dgh_n), ptr(dHN0),
             ptr(scratch), R, 15, ptr(bound_n if fused_heads else None), ptr(top_step), stream_ptr())
    else:
        top_step = None
        dgi_n, dgh_n, dHN0 = gru_bwd(prec, HN, st['gates_n'], w_hh_n, dh_ext=dNSUM.view(15, R, Hn))
    mark('dec_bwd:notes_bptt')
    dGC = sum_steps(dgi_n, t_top=top_step)                                    # [R, 3Hn]

    # (the gradient of the fed note tokens -- a 245760 x 128 x 1536 product, not an input of the time BPTT -- stays in front of it on the
    # chain: on the sibling stream it measured 9.0 against 8.38 ms per step, profiles/r04_ab_slots.txt)
    dtok = _empty(16, R, E, dev=dev)
    dtok[15].zero_()

    def notes_dx():
        gemm_dx(dgi_n.view(M, 3 * Hn), w_ih_n, slice(Ht, None), out=dtok[:15].view(M, E), prec=prec, m_top=top_step, m_unit=R)
        dNS = gemm_dx(dGC, w_ih_n, slice(0, Ht), prec=prec)                   # [R, Ht]
        w_tn = W['dec_time_to_notes_hid.weight']
        gemm_dx(dHN0, w_tn, out=dNS, acc=True, prec=prec)
        return dtok, dNS

    def notes_wgrads():
        # bias_hh gradient = column sums of dgh (its r and z thirds equal dgi's), taken inside the W_hh products
        if dgh_n.shape[-1] == Hn:                                # persistent BPTT: dgh holds its n third only
            for name in ('dec_notes_gru.weight_hh_l0', 'dec_notes_gru.bias_hh_l0'):
                if G[name] is None:
                    G[name] = _gbuf(P[name])
            gw, gb, h_op = G['dec_notes_gru.weight_hh_l0'], G['dec_notes_gru.bias_hh_l0'], HNo[:15].view(M, Hn)
            # (the rows of dgi / dgh after the last step that received a gradient are zero: the products stop there)
            # (ONE product for the whole [3Hn, Hn] gradient -- ptv_wgrad_cat, the state matrix read once -- measured 0.1 ms per step SLOWER in
            # situ, 7.83-7.90 against 7.73-7.78 ms: the single 768-block launch crowds the time BPTT it runs beside; two launches stay)
            wgrad_bias(dgi_n.view(M, 3 * Hn)[:, :2 * Hn], h_op, gw[:2 * Hn], gb[:2 * Hn], prec, top_step, R)
            wgrad_bias(dgh_n.view(M, Hn), h_op, gw[2 * Hn:], gb[2 * Hn:], prec, top_step, R)
        else:
            wgrad_b('dec_notes_gru.weight_hh_l0', 'dec_notes_gru.bias_hh_l0', dgh_n.view(M, 3 * Hn), HNo[:15].view(M, Hn))
        # (bias gradients as column sums INSIDE the products that read the same gradient matrix -- ptv_wgrad's colsum_a, as
        # ptv_decoder_tf_bwd's batch does: no second pass over dGC / dHN0)
        wgrad_b('dec_notes_gru.weight_ih_l0', 'dec_notes_gru.bias_ih_l0', dGC, NSf_op, slice(0, Ht))
        if top_step is not None:
            if G['dec_notes_gru.weight_ih_l0'] is None:
                G['dec_notes_gru.weight_ih_l0'] = _gbuf(P['dec_notes_gru.weight_ih_l0'])
            wgrad_bias(dgi_n.view(M, 3 * Hn), tok_op, G['dec_notes_gru.weight_ih_l0'][:, Ht:], None, prec, top_step, R)
        else:
            wgrad('dec_notes_gru.weight_ih_l0', dgi_n.view(M, 3 * Hn), tok_op, slice(Ht, None))
        wgrad_b('dec_time_to_notes_hid.weight', 'dec_time_to_notes_hid.bias', dHN0, NSf_op)
    dtok, dNS = notes_dx()
    # (forked BEFORE the time BPTT is queued: forking after it -- the four deep products then start when the persistent launch is done
    # instead of running beside it -- measured 8.48 against 8.18 ms per step)
    side(notes_wgrads, dgi_n, dgh_n, dGC, dHN0, dNSUM)

    mark('dec_bwd:notes_dx')
    # ---- time GRU (32 steps, batch B) ----
    w_hh_t, w_ih_t = W['dec_time_gru.weight_hh_l0'], W['dec_time_gru.weight_ih_l0']
    dgi_t, dgh_t, dzhid = gru_bwd(prec, NS, st['gates_t'], w_hh_t, dh_ext=dNS.view(32, B, Ht))
    mark('dec_bwd:time_bptt')
    dZG = sum_steps(dgi_t)                                                    # [B, 3Ht]
    dz_in = gemm_dx(dZG, w_ih_t, slice(2 * He, None), prec=prec)              # [B, Zi]
    dTOKS = _empty(33, B, 2 * He, dev=dev)
    dTOKS[32].zero_()
    gemm_dx(dgi_t.view(R, 3 * Ht), w_ih_t, slice(0, 2 * He), out=dTOKS[:32].view(R, 2 * He), prec=prec)
    # (the note-summary node's two inputs -- this gradient and, queued earlier on this stream, the parked embedding gradient -- are final
    # HERE: its BPTT family starts from this event, not from the end of whatever else the step has queued by then)
    st['ev_dtoks'] = record_event() if SUMMARY_FAMILY_SLOT >= 0 and OVERLAP else None
    w_zh, w_zi = W['z2dec_hid_linear.weight'], W['z2dec_in_linear.weight']
    dz = gemm_dx(dzhid, w_zh, prec=prec)
    gemm_dx(dz_in, w_zi, out=dz, acc=True, prec=prec)

    def time_wgrads():
        # bias_hh = column sums of dgh over all (t, b) rows, inside the W_hh product; bias_ih = column sums of dZG = sum_t dgi_t inside the
        # z_in product (the same folds as ptv_decoder_tf_bwd's batch; products the weight-gradient kernel does not take -- fp32 precision,
        # K = B < 512 -- fall back to product + column-sum kernel inside wgrad_bias, as the library's WgradGroup does)
        wgrad_b('dec_time_gru.weight_hh_l0', 'dec_time_gru.bias_hh_l0', dgh_t.view(R, 3 * Ht), NSo[:32].view(R, Ht))
        wgrad_b('dec_time_gru.weight_ih_l0', 'dec_time_gru.bias_ih_l0', dZG, st['z_in'], slice(2 * He, None))
        wgrad('dec_time_gru.weight_ih_l0', dgi_t.view(R, 3 * Ht), TOKS[:32].view(R, 2 * He), slice(0, 2 * He))
        wgrad_b('z2dec_hid_linear.weight', 'z2dec_hid_linear.bias', dzhid, z)
        wgrad_b('z2dec_in_linear.weight', 'z2dec_in_linear.bias', dz_in, z)
        bgrad('dec_init_input', dTOKS[0])
    side(time_wgrads, dgi_t, dgh_t, dZG, dTOKS, dzhid, dz_in, dNS)
    # the caller defers the join to the end of the backward pass; the node's saved forward state (released when the node returns)
    # is still being read by the products queued above
    side.keep.extend((st, z, tok_op))
    return dz, dtok, dTOKS, G, side


# =============================================================================================
# RnnDecoder (chord decoder), teacher-forced (tfr3 = 1): ptvae.py:51-87
# =============================================================================================
CHD_PARAM_NAMES = ['init_input', 'z2dec_hid.weight', 'z2dec_hid.bias', 'z2dec_in.weight', 'z2dec_in.bias',
                   'gru.weight_ih_l0', 'gru.weight_hh_l0', 'gru.bias_ih_l0', 'gru.bias_hh_l0',
                   'root_out.weight', 'root_out.bias', 'chroma_out.weight', 'chroma_out.bias',
                   'bass_out.weight', 'bass_out.bias']


CHD_BWD_COMPOSITE = os.environ.get('PTV_BWD_COMPOSITES', '1') != '0'     # ChordDecoderTFFn.backward through ptv_chord_decoder_bwd (one C call: 26 launches + the persistent launch's turn)
_CDB = {}


def _chord_decoder_bwd_composite(P, st, z, droot, dchroma, dbass):
    """-> (dz, {name: gradient}) when ptv_chord_decoder_bwd ran, else None (the caller sequences the launches itself: same bits)"""
    if 't' not in _CDB:
        from ._lib import header_enum
        _CDB['t'], _CDB['d'] = header_enum('PtvCdbTensor'), header_enum('PtvCdbDim')
    T_, D_ = _CDB['t'], _CDB['d']
    prec, T, B, H, I = st['prec'], st['T'], st['B'], st['H'], st['I']
    dev = z.device
    hall, gates, toks, z_in = st['hall'], st['gates'], st['toks'], st['z_in']
    dls = []
    for dl in (droot, dchroma, dbass):
        if dl is not None:
            dl = dl.contiguous()
            if dl.dtype != F32:
                return None
        dls.append(dl)
    adt = _act_dtype(prec, H)
    if (torch.cuda.is_current_stream_capturing() or hall.dtype != F32 or gates.dtype != adt or z.dtype != F32 or not z.is_contiguous()
            or not hall.is_contiguous() or not gates.is_contiguous() or not toks.is_contiguous() or not z_in.is_contiguous()):
        return None                      # (inside a capture the persistent launch's turn must go through wait_event(): the Python path)
    w_hh = P['gru.weight_hh_l0']
    wt = _WT(w_hh, prec) if adt == BF16 else None
    persist = bool(adt == BF16 and wt is not None and T >= 2 and persist_supported(1, B, H, T))
    S = persist_splitk(1, B, H) if persist else 0
    Z, Zi = z.shape[1], z_in.shape[1]
    dims = [0] * D_['PTV_CDB_D_COUNT']
    for k, v in (('B', B), ('T', T), ('H', H), ('I', I), ('Z', Z), ('ZI', Zi), ('PREC', prec), ('ACT_BF16', int(adt == BF16)),
                 ('NROOT', P['root_out.weight'].shape[0]), ('NCHROMA', P['chroma_out.weight'].shape[0]), ('NBASS', P['bass_out.weight'].shape[0]),
                 ('PERSIST', int(persist)), ('SPLITK', S)):
        dims[D_['PTV_CDB_D_' + k]] = v
    G = {n: _gbuf(P[n]) for n in CHD_PARAM_NAMES}
    dz = _empty(B, Z, dev=dev)
    tens = {'Z': z, 'W_ZHID': P['z2dec_hid.weight'], 'W_ZIN': P['z2dec_in.weight'], 'W_IH': P['gru.weight_ih_l0'], 'W_HH': w_hh,
            'W_ROOT': P['root_out.weight'], 'W_CHROMA': P['chroma_out.weight'], 'W_BASS': P['bass_out.weight'], 'WT16_HH': wt,
            'HALL': hall, 'GATES': gates, 'TOKS': toks, 'Z_IN': z_in, 'DROOT': dls[0], 'DCHROMA': dls[1], 'DBASS': dls[2], 'DZ': dz,
            'G_INIT_INPUT': G['init_input'], 'G_W_ZHID': G['z2dec_hid.weight'], 'G_B_ZHID': G['z2dec_hid.bias'], 'G_W_ZIN': G['z2dec_in.weight'],
            'G_B_ZIN': G['z2dec_in.bias'], 'G_W_IH': G['gru.weight_ih_l0'], 'G_B_IH': G['gru.bias_ih_l0'], 'G_W_HH': G['gru.weight_hh_l0'],
            'G_B_HH': G['gru.bias_hh_l0'], 'G_W_ROOT': G['root_out.weight'], 'G_B_ROOT': G['root_out.bias'], 'G_W_CHROMA': G['chroma_out.weight'],
            'G_B_CHROMA': G['chroma_out.bias'], 'G_W_BASS': G['bass_out.weight'], 'G_B_BASS': G['bass_out.bias'],
            'DHS': _empty(T * B, H, dev=dev), 'DGI': _empty(T, B, 3 * H, dev=dev, dtype=adt), 'DGH': _empty(T, B, 3 * H, dev=dev, dtype=adt),
            'DHZ': None if persist else _empty(2, B, H, dev=dev), 'DH0': _empty(B, H, dev=dev), 'DZG': _empty(B, 3 * H, dev=dev),
            'DZ_IN': _empty(B, Zi, dev=dev), 'DTOK0': _empty(B, I, dev=dev),
            'XCH': torch.empty(T * B * 3 * H, device=dev, dtype=BF16) if persist else None,
            'PART': torch.empty(lib().ptv_gru_persist_part_elems(1, B, H, S), device=dev) if S else None,
            'SYNC': _persist_sync(1, dev) if persist else None}
    slots = [None] * T_['PTV_CDB_COUNT']
    for k, v in tens.items():
        slots[T_['PTV_CDB_' + k]] = ptr(v)
    cur = cur_stream()
    done = None
    if persist:
        prev = _PERSIST_LAST.get(cur.device.index)
        done = torch.cuda.Event()
        if prev is not None:
            slots[T_['PTV_CDB_WAIT_EVENT']] = prev.cuda_event
        done.record(cur)                  # creates the handle; the library records it again after the persistent launch
        slots[T_['PTV_CDB_RECORD_EVENT']] = done.cuda_event
    _chain_prio()
    rc = lib().ptv_chord_decoder_bwd((ctypes.c_void_p * len(slots))(*slots), _larr(dims), stream_ptr())
    if rc == -3:                          # (persist_supported() was asked before the arena views were taken)
        raise RuntimeError('ptv_chord_decoder_bwd refused a configuration its Python-side checks accepted')
    check(rc, 'ptv_chord_decoder_bwd')
    if done is not None:
        _PERSIST_LAST[cur.device.index] = done
    _CDB['calls'] = _CDB.get('calls', 0) + 1
    # (the scratch tensors die here while the launches are only queued: the caching allocator reuses a block in stream order; the gradient
    # views must NOT be kept anywhere -- a second reference makes AccumulateGrad clone instead of adopt them)
    return dz, G


class ChordDecoderTFFn(torch.autograd.Function):
    """(z_chd [B,Z], c_sm [8,B,36] step-major, *params) -> root [8,B,12], chroma [8,B,24], bass [8,B,12]"""

    @staticmethod
    def forward(ctx, z, c_sm, prec, *params):
        P = dict(zip(CHD_PARAM_NAMES, params))
        dev = z.device
        z = z.contiguous()
        B = z.shape[0]
        T = c_sm.shape[0]
        H = P['gru.weight_hh_l0'].shape[1]
        I = c_sm.shape[2]
        if CHD_COMPOSITE and c_sm.is_contiguous() and c_sm.dtype == F32 and z.dtype == F32 and not capturing_part():
            # the whole forward behind ONE C call (ptv_chord_decoder_fwd, csrc/composite.hip)
            if 'ct' not in _DTF:
                from ._lib import header_enum
                _DTF['ct'], _DTF['cd'] = header_enum('PtvCdfTensor'), header_enum('PtvCdfDim')
            CT, CD = _DTF['ct'], _DTF['cd']
            Zi = P['z2dec_in.weight'].shape[0]
            adt = _act_dtype(prec, H)
            hall, z_in, toks = _empty(T + 1, B, H, dev=dev), _empty(B, Zi, dev=dev), _empty(T, B, I, dev=dev)
            gi, zg = _empty(T * B, 3 * H, dev=dev), _empty(B, 3 * H, dev=dev)
            gates = _empty(T, 4, B, H, dev=dev, dtype=adt)
            nr, nc, nb = (P[n + '_out.weight'].shape[0] for n in ('root', 'chroma', 'bass'))
            root, chroma, bass = _empty(T * B, nr, dev=dev), _empty(T * B, nc, dev=dev), _empty(T * B, nb, dev=dev)
            tens = {'Z': z, 'C_SM': c_sm, 'W_ZHID': P['z2dec_hid.weight'], 'B_ZHID': P['z2dec_hid.bias'], 'W_ZIN': P['z2dec_in.weight'],
                    'B_ZIN': P['z2dec_in.bias'], 'INIT_INPUT': P['init_input'], 'W_IH': P['gru.weight_ih_l0'], 'B_IH': P['gru.bias_ih_l0'],
                    'W_HH': P['gru.weight_hh_l0'], 'B_HH': P['gru.bias_hh_l0'], 'W_ROOT': P['root_out.weight'], 'B_ROOT': P['root_out.bias'],
                    'W_CHROMA': P['chroma_out.weight'], 'B_CHROMA': P['chroma_out.bias'], 'W_BASS': P['bass_out.weight'],
                    'B_BASS': P['bass_out.bias'], 'HALL': hall, 'Z_IN': z_in, 'TOKS': toks, 'GI': gi, 'ZG': zg, 'GATES': gates, 'ROOT': root,
                    'CHROMA': chroma, 'BASS': bass}
            slots = [None] * CT['PTV_CDF_COUNT']
            for k, v in tens.items():
                slots[CT['PTV_CDF_' + k]] = ptr(v)
            dims = [0] * CD['PTV_CDF_D_COUNT']
            for k, v in (('B', B), ('T', T), ('H', H), ('I', I), ('Z', z.shape[1]), ('ZI', Zi), ('PREC', prec), ('GATES_BF16', int(adt == BF16)),
                         ('NROOT', nr), ('NCHROMA', nc), ('NBASS', nb)):
                dims[CD['PTV_CDF_D_' + k]] = v
            _chain_prio()
            check(lib().ptv_chord_decoder_fwd((ctypes.c_void_p * len(slots))(*slots), _larr(dims), stream_ptr()), 'ptv_chord_decoder_fwd')
            _DTF['chd_calls'] = _DTF.get('chd_calls', 0) + 1
            ctx.save_for_backward(z, *params)
            ctx.st = dict(hall=hall, gates=gates, toks=toks, z_in=z_in, prec=prec, T=T, B=B, H=H, I=I)
            return root.view(T, B, -1), chroma.view(T, B, -1), bass.view(T, B, -1)
        hall = _empty(T + 1, B, H, dev=dev)
        gemm(z, P['z2dec_hid.weight'], hall[0], bias=P['z2dec_hid.bias'], prec=prec)
        z_in = gemm(z, P['z2dec_in.weight'], bias=P['z2dec_in.bias'], prec=prec)
        toks = _empty(T, B, I, dev=dev)
        copy2d(toks[0], P['init_input'].view(1, -1), lds=0)
        if T > 1:
            copy2d(toks[1:].view((T - 1) * B, I), c_sm[:T - 1].reshape((T - 1) * B, I))
        w_ih = P['gru.weight_ih_l0']
        gi = gemm(toks.view(T * B, I), w_ih[:, :I], prec=prec)
        zg = gemm(z_in, w_ih[:, I:], bias=P['gru.bias_ih_l0'], prec=prec)
        gates = _empty(T, 4, B, H, dev=dev, dtype=_act_dtype(prec, H))
        gru_fwd(prec, gi, B * 3 * H, 3 * H, P['gru.weight_hh_l0'], P['gru.bias_hh_l0'], hall, gates, gi2=zg, gi2_step=0,
                gi2_ld=3 * H)
        hs = hall[1:].view(T * B, H)
        root = gemm(hs, P['root_out.weight'], bias=P['root_out.bias'], prec=prec)
        chroma = gemm(hs, P['chroma_out.weight'], bias=P['chroma_out.bias'], prec=prec)
        bass = gemm(hs, P['bass_out.weight'], bias=P['bass_out.bias'], prec=prec)
        ctx.save_for_backward(z, *params)
        ctx.st = dict(hall=hall, gates=gates, toks=toks, z_in=z_in, prec=prec, T=T, B=B, H=H, I=I)
        return root.view(T, B, -1), chroma.view(T, B, -1), bass.view(T, B, -1)

    @staticmethod
    def backward(ctx, droot, dchroma, dbass):
        z, *params = ctx.saved_tensors
        mark('chd_dec_bwd:start')
        P = dict(zip(CHD_PARAM_NAMES, params))
        st = ctx.st
        ctx.st = None
        prec, T, B, H, I = st['prec'], st['T'], st['B'], st['H'], st['I']
        dev = z.device
        hall, toks = st['hall'], st['toks']
        if CHD_BWD_COMPOSITE:
            res = _chord_decoder_bwd_composite(P, st, z, droot, dchroma, dbass)
            if res is not None:
                mark('chd_dec_bwd:end')
                return (res[0], None, None) + tuple(res[1][n] for n in CHD_PARAM_NAMES)
        hs = hall[1:].view(T * B, H)
        G = {}
        dhs = None
        for name, dlog in (('root_out', droot), ('chroma_out', dchroma), ('bass_out', dbass)):
            w = P[name + '.weight']
            if dlog is None:
                G[name + '.weight'], G[name + '.bias'] = _gbuf(w), _gbuf(P[name + '.bias'])
                continue
            d2 = dlog.contiguous().view(T * B, -1)
            if dhs is None:
                dhs = gemm(d2, w, tb=True, prec=prec)
            else:
                gemm(d2, w, dhs, tb=True, acc=True, prec=prec)
            # (weight and bias gradient in one pass over the logits' gradient: ptv_wgrad's colsum_a, as ptv_chord_decoder_bwd's batch does)
            G[name + '.weight'], G[name + '.bias'] = wgrad_bias(d2, hs, _gbuf(w), _gbuf(P[name + '.bias']), prec)
        if dhs is None:
            dhs = _zeros(T * B, H, dev=dev)
        w_hh, w_ih = P['gru.weight_hh_l0'], P['gru.weight_ih_l0']
        # (per-step kernels, not the persistent launch: persistent launches take turns, and this short chain on its sibling stream had
        # to wait for the decoder's 32-step time BPTT -- its dz then reached the chord encoder 0.3 ms after the decoder's own, round 4)
        dgi, dgh, dh0 = gru_bwd(prec, hall, st['gates'], w_hh, dh_ext=dhs.view(T, B, H), allow_persist=True)
        G['gru.weight_hh_l0'], G['gru.bias_hh_l0'] = wgrad_bias(dgh.view(T * B, 3 * H), hall[:T].view(T * B, H), _gbuf(w_hh),
                                                                _gbuf(P['gru.bias_hh_l0']), prec)
        dzg = sum_steps(dgi)
        g = _gbuf(w_ih)
        _, G['gru.bias_ih_l0'] = wgrad_bias(dzg, st['z_in'], g[:, I:], _gbuf(P['gru.bias_ih_l0']), prec)
        gemm(dgi.view(T * B, 3 * H), toks.view(T * B, I), g[:, :I], ta=True, tb=True, acc=True, prec=prec)
        G['gru.weight_ih_l0'] = g
        dz_in = gemm(dzg, w_ih[:, I:], tb=True, prec=prec)
        dtok0 = gemm(dgi[0], w_ih[:, :I], tb=True, prec=prec)                     # only the learned start token
        G['init_input'] = _bgrad(P['init_input'], dtok0)
        w_zh, w_zi = P['z2dec_hid.weight'], P['z2dec_in.weight']
        dz = gemm(dh0, w_zh, tb=True, prec=prec)
        gemm(dz_in, w_zi, dz, tb=True, acc=True, prec=prec)
        G['z2dec_hid.weight'], G['z2dec_hid.bias'] = wgrad_bias(dh0, z, _gbuf(w_zh), _gbuf(P['z2dec_hid.bias']), prec)
        G['z2dec_in.weight'], G['z2dec_in.bias'] = wgrad_bias(dz_in, z, _gbuf(w_zi), _gbuf(P['z2dec_in.bias']), prec)
        mark('chd_dec_bwd:end')
        return (dz, None, None) + tuple(G[n] for n in CHD_PARAM_NAMES)


# =============================================================================================
# loss_function  (model.py:57-90, ptvae.py:498-511)
# =============================================================================================
_PERM = {3: (1, 0, 2), 4: (2, 1, 0, 3), 5: (2, 1, 0, 3, 4)}     # API shape <-> step-major memory (self-inverse)


def _chord_perm(t):
    return (1, 0, 2) if t.dim() == 3 else (1, 0, 2, 3)


def _mem_order(ts, perms):
    """API-shaped tensors -> (tensors contiguous in memory order, step_major flag).  Step-major iff
    EVERY tensor is a permuted view of a contiguous step-major buffer (what the decoders return);
    otherwise all are made batch-major contiguous."""
    tp = [t.permute(*p) for t, p in zip(ts, perms)]
    if all(t.is_contiguous() or _row_dense(t) for t in tp):
        return tp, True
    return [t.contiguous() for t in ts], False


WDUR = (1.0, 0.6, 0.4, 0.3, 0.3)          # ptvae.py:519-520


def _pianotree_ce_fwd(pitch, dur, x, sums, st, weighted=False):
    (pitch_m, dur_m), sm = _mem_order([pitch, dur], [_PERM[4], _PERM[5]])
    dev = pitch.device
    B = x.shape[0]
    rows = B * 480
    NP = pitch.shape[-1]
    cached = _cached_targets(x, sm)                          # (DisentangleVAE.loss() computed them before the decoder ran)
    if cached is not None:
        pitch_t, dur_t, counts = cached
    else:
        pitch_t = torch.empty(rows, device=dev, dtype=torch.int32)
        dur_t = torch.empty(rows * 5, device=dev, dtype=torch.int32)
        counts = _izeros(3, dev)                             # valid pitch / duration targets; last note step with any (zero-skip limit)
        call('ptv_pianotree_targets', ptr(x), B, int(sm), ptr(pitch_t), ptr(dur_t), ptr(counts), st)
    call('ptv_ce_fwd', ptr(pitch_m), pitch_m.stride(-2), ptr(pitch_t), rows, NP, 130, ptr(sums[0:]), st)
    if weighted:                              # 5 per-bit-position means, weighted (ptvae.py:512-527)
        gsum = _zeros(5, dev=dev)
        gcnt = _izeros(5, dev)
        call('ptv_ce_group_fwd', ptr(dur_m), 2, ptr(dur_t), rows * 5, 2, 2, 5, ptr(gsum), ptr(gcnt), st)
        call('ptv_wdur_finalize', ptr(gsum), ptr(gcnt), *WDUR, ptr(sums[1:]), ptr(counts[1:]), st)
        return pitch_m, dur_m, sm, pitch_t, dur_t, counts, gcnt
    call('ptv_ce_fwd', ptr(dur_m), 2, ptr(dur_t), rows * 5, 2, 2, ptr(sums[1:]), st)
    return pitch_m, dur_m, sm, pitch_t, dur_t, counts, None


def _pianotree_ce_bwd(pitch_m, dur_m, sm, pitch_t, dur_t, gs, st, gcnt=None):
    NP = pitch_m.shape[-1]
    rows = pitch_t.numel()
    ld = pitch_m.stride(-2)                                  # gradient in the logits' (possibly row-padded) layout
    dpitch = torch.empty(rows, ld, device=pitch_m.device)[:, :NP].as_strided(pitch_m.shape, pitch_m.stride())
    ddur = torch.empty_like(dur_m)
    call('ptv_ce_bwd', ptr(pitch_m), ld, ptr(pitch_t), rows, NP, 130, ptr(gs[0:]), ptr(dpitch), ld, st)
    if gcnt is not None:
        gs5 = _empty(5, dev=pitch_m.device)
        call('ptv_wdur_scales', ptr(gs[1:]), ptr(gcnt), *WDUR, ptr(gs5), st)
        call('ptv_ce_group_bwd', ptr(dur_m), 2, ptr(dur_t), rows * 5, 2, 2, 5, ptr(gs5), ptr(ddur), 2, st)
    else:
        call('ptv_ce_bwd', ptr(dur_m), 2, ptr(dur_t), rows * 5, 2, 2, ptr(gs[1:]), ptr(ddur), 2, st)
    if sm:
        dpitch, ddur = dpitch.permute(*_PERM[4]), ddur.permute(*_PERM[5])
    return dpitch, ddur


_LOSS_TOP = {}

# ---- dead note steps of the forward --------------------------------------------------------------------------------------------
# The loss ignores the padded note slots (CrossEntropyLoss(ignore_index), ptvae.py:498-511): the decoder outputs of the note steps after
# the last one that holds ANY target of the batch are dead values when the caller only wants the loss.  DisentangleVAE.loss() -- run +
# loss_function in one call, nothing of run()'s outputs returned -- computes the targets before the decoder (they depend on x only) and
# arms `live_top`; the teacher-forced decoder node then runs its notes GRU, heads and duration GRU for the live steps only (device-side
# limit: no host sync) and the loss node reuses the targets.  run() itself always computes every step (its outputs ARE the result).
# The backward's zero-skip limit is the same number, so nothing reads the unwritten rows (tests poison them with NaN: POISON_DEAD_STEPS).
DEAD_STEPS = os.environ.get('PTV_DEAD_STEPS', '1') != '0'
POISON_DEAD_STEPS = False
_LIVE = {}


def pianotree_targets(x, step_major):
    """-> (pitch_t [480 B] int32, dur_t [2400 B] int32, counts int32 [3]: valid pitch / duration targets, last note step with any)"""
    B = x.shape[0]
    dev = x.device
    pitch_t = torch.empty(B * 480, device=dev, dtype=torch.int32)
    dur_t = torch.empty(B * 2400, device=dev, dtype=torch.int32)
    counts = _izeros(3, dev)
    call('ptv_pianotree_targets', ptr(x), B, int(step_major), ptr(pitch_t), ptr(dur_t), ptr(counts), stream_ptr())
    return pitch_t, dur_t, counts


def arm_live_top(x):
    """called by DisentangleVAE.loss() before run(): targets of x now (step-major: the teacher-forced decoder's layout), kept for the loss
    node; the decoder node of THIS forward may stop at counts[2].  -> token for disarm_live_top"""
    # (B a multiple of 4: the heads kernel's 128-row blocks must not straddle the limit -- a note step holds 32 B rows)
    if not (DEAD_STEPS and ZERO_SKIP and x.is_cuda and x.dtype == torch.int64 and x.is_contiguous() and x.shape[0] % 4 == 0):
        return None
    B = x.shape[0]
    R = 32 * B
    sort = SORT_DEC_ROWS and DEC_COMPOSITE and DEC_BWD_COMPOSITE and not torch.cuda.is_current_stream_capturing()
    row_live = _izeros(R, x.device) if sort else None
    pitch_t = torch.empty(B * 480, device=x.device, dtype=torch.int32)
    dur_t = torch.empty(B * 2400, device=x.device, dtype=torch.int32)
    counts = _izeros(3, x.device)
    call('ptv_pianotree_targets_rows', ptr(x), B, 1, ptr(pitch_t), ptr(dur_t), ptr(counts), ptr(row_live), stream_ptr())
    pt, dt = pitch_t, dur_t
    _LIVE.pop('sort', None)
    if sort:
        # per-row dead work (round 6): the decoder's rows (t, b) in the order of DESCENDING number of live note steps.  Here: the
        # permutation, the lengths and the loss targets in that order; the decoder node takes them up if its composite runs
        # (_decoder_tf_composite) and then marks them used -- the loss node picks the targets that match the logits it is given
        perm = torch.empty(R, device=x.device, dtype=torch.int32)
        call('ptv_rows_by_length', ptr(row_live), ptr(perm), R, 15, stream_ptr())
        len_s = torch.empty(R, device=x.device, dtype=torch.int32)
        call('ptv_gather_rows', ptr(len_s), ptr(row_live), ptr(perm), R, 1, 0, 0, 1, stream_ptr())
        pt_s, dt_s = torch.empty_like(pitch_t), torch.empty_like(dur_t)
        call('ptv_gather_rows', ptr(pt_s), ptr(pitch_t), ptr(perm), R, 1, R, R, 15, stream_ptr())
        call('ptv_gather_rows', ptr(dt_s), ptr(dur_t), ptr(perm), R, 5, 5 * R, 5 * R, 15, stream_ptr())
        seg_n = None
        if WGRAD_SEG and R % 128 == 0 and lib().ptv_wgrad_seg_supported(15 * R, R):    # (slabs of the products must not straddle a note step)
            # ... and the live prefix of every note step in that order (128-row blocks): the weight-gradient products over (note step, row) clip to it
            seg_n = torch.empty(15, device=x.device, dtype=torch.int32)
            call('ptv_rows_seg_counts', ptr(len_s), R, 15, ptr(seg_n), stream_ptr())
            global _LAST_SEG_N
            _LAST_SEG_N = seg_n                                  # (bench.py's roofline record: the live fraction of the segmented products)
        _LIVE['sort'] = dict(perm=perm, len=len_s, pt=pt_s, dt=dt_s, used=False, x=x.data_ptr(), seg_n=seg_n)
    _LIVE['x'] = (weakref.ref(x), x.data_ptr(), x._version, True, pt, dt, counts)
    _LIVE['top'] = counts[2:3]
    _LIVE['last_counts'] = counts                               # (bench.py's roofline record: how many note steps the launches ran)
    return counts


def disarm_live_top():
    _LIVE.pop('top', None)


def live_top_for(dev):
    t = _LIVE.get('top')
    return t if (t is not None and t.device == dev) else None


def _cached_targets(x, sm):
    ent = _LIVE.get('x')
    if ent is None:
        return None
    ref, p, ver, sm0, pt, dt, counts = ent
    if ref() is x and x.data_ptr() == p and x._version == ver and bool(sm) == sm0:
        _LIVE.pop('x', None)
        srt = _LIVE.pop('sort', None)
        if srt is not None and srt['used'] and srt['x'] == p:      # the decoder node ran on length-sorted rows: its logits are in that order
            return srt['pt'], srt['dt'], counts
        return pt, dt, counts
    return None


def _poison(*tensors):
    for t in tensors:
        if t is None:
            continue
        if t.dtype.is_floating_point:
            t.fill_(float('nan'))
        else:
            t.fill_(0x3fffffff)


def _loss_top_hint(dpitch, ddur):
    """the loss node's zero-skip bound, valid ONLY for the very tensors it returned, unmodified.  The entry holds the two gradient tensors
    themselves (so their memory cannot be handed to another tensor while the entry lives) with their version counters: what arrives must
    have the same storage address, shape, strides and version.  A gradient that autograd accumulated another contribution into (a second
    consumer of the logits) is a new allocation; an in-place tensor hook bumps the version -- in those cases the caller scans the
    gradients themselves.  (Round-4 advice: the key used to be the two addresses alone.  Object identity does not work: the engine hands a
    node's result to the next node through C++, and the Python wrapper it arrives in is a new object.)"""
    ent = _LOSS_TOP.pop('hint', None)
    if ent is None or dpitch is None or ddur is None:
        return None
    gp, gd, vp, vd, top = ent

    def same(a, b, v):
        return (a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.stride() == b.stride() and a.dtype == b.dtype
                and b._version == v and a._version == v)
    return top if (same(gp, dpitch, vp) and same(gd, ddur, vd)) else None


LOSS_TOP_HINT = True


LOSS_COMPOSITE = os.environ.get('PTV_BWD_COMPOSITES', '1') != '0'     # VaeLossFn through ptv_vae_loss_fwd / ptv_vae_loss_bwd (one C call each)
_VL = {}


def _vl_tables():
    if 't' not in _VL:
        from ._lib import header_enum
        _VL['t'], _VL['d'] = header_enum('PtvVlTensor'), header_enum('PtvVlDim')
    return _VL['t'], _VL['d']


def _vl_ok(*ts):
    return all(t.is_cuda and t.dtype == F32 and t.is_contiguous() for t in ts)


def _vae_loss_fwd_composite(pitch, dur, x, c, mu_c, sd_c, mu_r, sd_r, root_m, chroma_m, bass_m, sm_c, root_t, chroma_t, bass_t, sums, out, scal, st):
    """-> (pitch_m, dur_m, sm_p, pitch_t, dur_t, counts) when ptv_vae_loss_fwd ran, else None"""
    T_, D_ = _vl_tables()
    (pitch_m, dur_m), sm_p = _mem_order([pitch, dur], [_PERM[4], _PERM[5]])
    B = x.shape[0]
    NP = pitch.shape[-1]
    if (x.dtype != torch.int64 or not _vl_ok(c, mu_c, sd_c, mu_r, sd_r, root_m, chroma_m, bass_m) or pitch_m.dtype != F32 or dur_m.dtype != F32
            or not dur_m.is_contiguous() or not _row_dense(pitch_m) or pitch_m.numel() != B * 480 * NP or dur_m.numel() != B * 4800):
        return None
    dev = pitch.device
    cached = _cached_targets(x, sm_p)
    if cached is not None:
        pitch_t, dur_t, counts = cached
    else:
        pitch_t = torch.empty(B * 480, device=dev, dtype=torch.int32)
        dur_t = torch.empty(B * 2400, device=dev, dtype=torch.int32)
        counts = _izeros(3, dev)
    dims = [0] * D_['PTV_VL_D_COUNT']
    for k, v in (('B', B), ('Z', mu_c.shape[1]), ('NP', NP), ('LDP', pitch_m.stride(-2)), ('SM_P', int(sm_p)), ('SM_C', int(sm_c)),
                 ('HAVE_TARGETS', int(cached is not None))):
        dims[D_['PTV_VL_D_' + k]] = v
    slots = [None] * T_['PTV_VL_COUNT']
    for k, v in (('X', x), ('C', c), ('PITCH', pitch_m), ('DUR', dur_m), ('MU_C', mu_c), ('SD_C', sd_c), ('MU_R', mu_r), ('SD_R', sd_r),
                 ('ROOT', root_m), ('CHROMA', chroma_m), ('BASS', bass_m), ('PITCH_T', pitch_t), ('DUR_T', dur_t), ('COUNTS', counts),
                 ('ROOT_T', root_t), ('CHROMA_T', chroma_t), ('BASS_T', bass_t), ('SUMS', sums), ('OUT', out)):
        slots[T_['PTV_VL_' + k]] = ptr(v)
    rc = lib().ptv_vae_loss_fwd((ctypes.c_void_p * len(slots))(*slots), _larr(dims), (ctypes.c_double * 6)(*scal), st)
    check(rc, 'ptv_vae_loss_fwd')
    _VL['calls'] = _VL.get('calls', 0) + 1
    return pitch_m, dur_m, sm_p, pitch_t, dur_t, counts


def _vae_loss_bwd_composite(gout, gs, pitch_m, dur_m, mu_c, sd_c, mu_r, sd_r, root_m, chroma_m, bass_m, pitch_t, dur_t, counts, root_t, chroma_t,
                            bass_t, dpitch, ddur, dmu_c, dsd_c, dmu_r, dsd_r, droot, dchroma, dbass, scal, st):
    T_, D_ = _vl_tables()
    B = root_t.numel() // 8
    NP = pitch_m.shape[-1]
    if gout.dtype != F32 or not _row_dense(pitch_m) or not dur_m.is_contiguous():
        return False
    dims = [0] * D_['PTV_VL_D_COUNT']
    for k, v in (('B', B), ('Z', mu_c.shape[1]), ('NP', NP), ('LDP', pitch_m.stride(-2))):
        dims[D_['PTV_VL_D_' + k]] = v
    slots = [None] * T_['PTV_VL_COUNT']
    for k, v in (('PITCH', pitch_m), ('DUR', dur_m), ('MU_C', mu_c), ('SD_C', sd_c), ('MU_R', mu_r), ('SD_R', sd_r), ('ROOT', root_m),
                 ('CHROMA', chroma_m), ('BASS', bass_m), ('PITCH_T', pitch_t), ('DUR_T', dur_t), ('COUNTS', counts), ('ROOT_T', root_t),
                 ('CHROMA_T', chroma_t), ('BASS_T', bass_t), ('GOUT', gout), ('GS', gs), ('DPITCH', dpitch), ('DDUR', ddur), ('DMU_C', dmu_c),
                 ('DSD_C', dsd_c), ('DMU_R', dmu_r), ('DSD_R', dsd_r), ('DROOT', droot), ('DCHROMA', dchroma), ('DBASS', dbass)):
        slots[T_['PTV_VL_' + k]] = ptr(v)
    rc = lib().ptv_vae_loss_bwd((ctypes.c_void_p * len(slots))(*slots), _larr(dims), (ctypes.c_double * 6)(*scal), st)
    check(rc, 'ptv_vae_loss_bwd')
    _VL['bwd_calls'] = _VL.get('bwd_calls', 0) + 1
    return True


class VaeLossFn(torch.autograd.Function):
    """(pitch [B,32,15,130], dur [B,32,15,5,2], mu_c, sd_c, mu_r, sd_r, root [B,8,12], chroma [B,8,12,2],
    bass [B,8,12], x, c, beta, w0, w1) -> the 11 scalars of model.py:67-68 as one [11] tensor."""

    @staticmethod
    def forward(ctx, pitch, dur, mu_c, sd_c, mu_r, sd_r, root, chroma, bass, x, c, beta, w0, w1, weighted_dur=False):
        dev = pitch.device
        B = x.shape[0]
        st = stream_ptr()
        x = x.contiguous()
        c = c.contiguous()
        mu_c, sd_c, mu_r, sd_r = (t.contiguous() for t in (mu_c, sd_c, mu_r, sd_r))
        sums = _zeros(8, dev=dev)
        (root_m, chroma_m, bass_m), sm_c = _mem_order([root, chroma, bass], [_chord_perm(root), _chord_perm(chroma),
                                                                              _chord_perm(bass)])
        root_t = torch.empty(B * 8, device=dev, dtype=torch.int32)
        chroma_t = torch.empty(B * 96, device=dev, dtype=torch.int32)
        bass_t = torch.empty(B * 8, device=dev, dtype=torch.int32)
        Z = mu_c.shape[1]

        def small():
            # the two KL terms and the three chord cross-entropies: six ~5-us launches, next to the PianoTree cross-entropy instead of behind it
            st2 = stream_ptr()
            call('ptv_chord_targets', ptr(c), B, int(sm_c), ptr(root_t), ptr(chroma_t), ptr(bass_t), st2)
            call('ptv_kl_fwd', ptr(mu_c), ptr(sd_c), mu_c.numel(), ptr(sums[2:]), st2)
            call('ptv_kl_fwd', ptr(mu_r), ptr(sd_r), mu_r.numel(), ptr(sums[3:]), st2)
            call('ptv_ce_fwd', ptr(root_m), 12, ptr(root_t), B * 8, 12, -1, ptr(sums[4:]), st2)
            call('ptv_ce_fwd', ptr(chroma_m), 2, ptr(chroma_t), B * 96, 2, -1, ptr(sums[5:]), st2)
            call('ptv_ce_fwd', ptr(bass_m), 12, ptr(bass_t), B * 8, 12, -1, ptr(sums[6:]), st2)
        out = _empty(11, dev=dev)
        ctx.scal = (float(beta), float(w0), float(w1), float(B * Z), float(B * 8), float(B * 96))
        comp = None
        if LOSS_COMPOSITE and not weighted_dur:
            comp = _vae_loss_fwd_composite(pitch, dur, x, c, mu_c, sd_c, mu_r, sd_r, root_m, chroma_m, bass_m, sm_c, root_t, chroma_t, bass_t,
                                           sums, out, ctx.scal, st)
        if comp is not None:
            pitch_m, dur_m, sm_p, pitch_t, dur_t, counts = comp
            ctx.gcnt = None
        else:
            pitch_m, dur_m, sm_p, pitch_t, dur_t, counts, gcnt = _pianotree_ce_fwd(pitch, dur, x, sums, st, weighted_dur)
            ctx.gcnt = gcnt
            small()                               # (on a sibling stream beside the PianoTree cross-entropy: 8.298 against 8.302 ms -- in line)
            call('ptv_loss_finalize', ptr(sums), ptr(counts), *ctx.scal, ptr(out), st)
        ctx.save_for_backward(pitch_m, dur_m, mu_c, sd_c, mu_r, sd_r, root_m, chroma_m, bass_m, pitch_t, dur_t, counts,
                              root_t, chroma_t, bass_t)
        ctx.sm = (sm_p, sm_c)
        return out

    @staticmethod
    def backward(ctx, gout):
        (pitch_m, dur_m, mu_c, sd_c, mu_r, sd_r, root_m, chroma_m, bass_m, pitch_t, dur_t, counts, root_t, chroma_t,
         bass_t) = ctx.saved_tensors
        sm_p, sm_c = ctx.sm
        dev = pitch_m.device
        st = stream_ptr()
        reset_deferred()                        # first node of the backward pass: nothing may be left from an aborted one
        gs = _empty(8, dev=dev)
        dmu_c, dsd_c = torch.empty_like(mu_c), torch.empty_like(sd_c)
        dmu_r, dsd_r = torch.empty_like(mu_r), torch.empty_like(sd_r)
        droot, dchroma, dbass = torch.empty_like(root_m), torch.empty_like(chroma_m), torch.empty_like(bass_m)
        done = False
        if LOSS_COMPOSITE and ctx.gcnt is None:
            NP = pitch_m.shape[-1]
            ld = pitch_m.stride(-2)
            dpitch = torch.empty(pitch_t.numel(), ld, device=dev)[:, :NP].as_strided(pitch_m.shape, pitch_m.stride())
            ddur = torch.empty_like(dur_m)
            done = _vae_loss_bwd_composite(gout.contiguous(), gs, pitch_m, dur_m, mu_c, sd_c, mu_r, sd_r, root_m, chroma_m, bass_m, pitch_t, dur_t,
                                           counts, root_t, chroma_t, bass_t, dpitch, ddur, dmu_c, dsd_c, dmu_r, dsd_r, droot, dchroma, dbass,
                                           ctx.scal, st)
            if done and sm_p:
                dpitch, ddur = dpitch.permute(*_PERM[4]), ddur.permute(*_PERM[5])
        if not done:
            call('ptv_loss_bwd_scales', ptr(gout.contiguous()), ptr(counts), *ctx.scal, ptr(gs), st)

        def small():
            st2 = stream_ptr()
            call('ptv_kl_bwd', ptr(mu_c), ptr(sd_c), mu_c.numel(), ptr(gs[2:]), ptr(dmu_c), ptr(dsd_c), st2)
            call('ptv_kl_bwd', ptr(mu_r), ptr(sd_r), mu_r.numel(), ptr(gs[3:]), ptr(dmu_r), ptr(dsd_r), st2)
            call('ptv_ce_bwd', ptr(root_m), 12, ptr(root_t), root_t.numel(), 12, -1, ptr(gs[4:]), ptr(droot), 12, st2)
            call('ptv_ce_bwd', ptr(chroma_m), 2, ptr(chroma_t), chroma_t.numel(), 2, -1, ptr(gs[5:]), ptr(dchroma), 2, st2)
            call('ptv_ce_bwd', ptr(bass_m), 12, ptr(bass_t), bass_t.numel(), 12, -1, ptr(gs[6:]), ptr(dbass), 12, st2)
        if not done:
            dpitch, ddur = _pianotree_ce_bwd(pitch_m, dur_m, sm_p, pitch_t, dur_t, gs, st, ctx.gcnt)
            small()
        if sm_c:
            droot, dchroma, dbass = (t.permute(*_chord_perm(t)) for t in (droot, dchroma, dbass))
        # zero-skip limit for whoever consumes exactly these two gradients (DecoderTFFn / DecoderStepFn): the last note step with a
        # non-ignored target bounds where they can be non-zero -- known from the forward's target pass, no scan of the gradients
        _LOSS_TOP.clear()
        if LOSS_TOP_HINT and sm_p:
            _LOSS_TOP['hint'] = (dpitch, ddur, dpitch._version, ddur._version, counts[2:3])
        return (dpitch, ddur, dmu_c, dsd_c, dmu_r, dsd_r, droot, dchroma, dbass) + (None,) * 6


class SplitScalarsFn(torch.autograd.Function):
    """out[n] -> n scalars, like Tensor.unbind -- whose backward launches one zero fill per output that got no gradient plus a stack
    (11 losses, `losses[0].backward()`: 10 fills + stack at the very head of the backward pass).  Here an output without gradient costs
    nothing: the incoming scalars are copied into a pooled zero row."""

    @staticmethod
    def forward(ctx, out):
        ctx.set_materialize_grads(False)
        ctx.n = out.shape[0]
        return tuple(out.detach().unbind(0))

    @staticmethod
    def backward(ctx, *gs):
        live = [(i, g) for i, g in enumerate(gs) if g is not None]
        if not live:
            return None
        g = _zeros(ctx.n, dev=live[0][1].device)
        for i, gi in live:
            g[i:i + 1].copy_(gi.reshape(1))
        return g


class ReconLossFn(torch.autograd.Function):
    """PtvaeDecoder.recon_loss (ptvae.py:498-511) -> [3] = (w0*pl + w1*dl, pl, dl)."""

    @staticmethod
    def forward(ctx, pitch, dur, x, w0, w1, weighted_dur=False):
        dev = pitch.device
        st = stream_ptr()
        x = x.contiguous()
        sums = _zeros(8, dev=dev)
        pitch_m, dur_m, sm, pitch_t, dur_t, counts, gcnt = _pianotree_ce_fwd(pitch, dur, x, sums, st, weighted_dur)
        ctx.gcnt = gcnt
        out = _empty(11, dev=dev)
        ctx.scal = (0.0, float(w0), float(w1), 1.0, 1.0, 1.0)
        call('ptv_loss_finalize', ptr(sums), ptr(counts), *ctx.scal, ptr(out), st)
        ctx.save_for_backward(pitch_m, dur_m, pitch_t, dur_t, counts)
        ctx.sm = sm
        return out[1:4].clone()

    @staticmethod
    def backward(ctx, g3):
        pitch_m, dur_m, pitch_t, dur_t, counts = ctx.saved_tensors
        dev = pitch_m.device
        st = stream_ptr()
        g11 = _zeros(11, dev=dev)
        copy2d(g11[1:4].view(1, 3), g3.contiguous().view(1, 3))
        gs = _empty(8, dev=dev)
        call('ptv_loss_bwd_scales', ptr(g11), ptr(counts), *ctx.scal, ptr(gs), st)
        dpitch, ddur = _pianotree_ce_bwd(pitch_m, dur_m, ctx.sm, pitch_t, dur_t, gs, st, ctx.gcnt)
        return dpitch, ddur, None, None, None, None


def recon_loss(x, pitch, dur, w0, w1, weighted_dur=False):
    return ReconLossFn.apply(pitch, dur, x, w0, w1, weighted_dur)


class KlFn(torch.autograd.Function):
    """kl_with_normal (train_utils.py:45-49): mean over all elements of KL(N(mu,sd) || N(0,1))."""

    @staticmethod
    def forward(ctx, mu, sd):
        mu, sd = mu.contiguous(), sd.contiguous()
        s = _zeros(2, dev=mu.device)
        call('ptv_kl_fwd', ptr(mu), ptr(sd), mu.numel(), ptr(s), stream_ptr())
        copy2d(s[1:2].view(1, 1), s[0:1].view(1, 1), alpha=1.0 / mu.numel())
        ctx.save_for_backward(mu, sd)
        return s[1]

    @staticmethod
    def backward(ctx, g):
        mu, sd = ctx.saved_tensors
        gs = _empty(1, dev=mu.device)
        copy2d(gs.view(1, 1), g.contiguous().view(1, 1), alpha=1.0 / mu.numel())
        dmu, dsd = torch.empty_like(mu), torch.empty_like(sd)
        call('ptv_kl_bwd', ptr(mu), ptr(sd), mu.numel(), ptr(gs), ptr(dmu), ptr(dsd), stream_ptr())
        return dmu, dsd


class ChordLossFn(torch.autograd.Function):
    """DisentangleVAE.chord_loss (model.py:70-83) -> [4] = (chord, root, chroma, bass)."""

    @staticmethod
    def forward(ctx, root, chroma, bass, c):
        dev = root.device
        st = stream_ptr()
        B = c.shape[0]
        c = c.contiguous()
        (root_m, chroma_m, bass_m), sm = _mem_order([root, chroma, bass], [_chord_perm(root), _chord_perm(chroma),
                                                                            _chord_perm(bass)])
        root_t = torch.empty(B * 8, device=dev, dtype=torch.int32)
        chroma_t = torch.empty(B * 96, device=dev, dtype=torch.int32)
        bass_t = torch.empty(B * 8, device=dev, dtype=torch.int32)
        call('ptv_chord_targets', ptr(c), B, int(sm), ptr(root_t), ptr(chroma_t), ptr(bass_t), st)
        sums = _zeros(8, dev=dev)
        counts = torch.ones(2, device=dev, dtype=torch.int32)
        call('ptv_ce_fwd', ptr(root_m), 12, ptr(root_t), B * 8, 12, -1, ptr(sums[4:]), st)
        call('ptv_ce_fwd', ptr(chroma_m), 2, ptr(chroma_t), B * 96, 2, -1, ptr(sums[5:]), st)
        call('ptv_ce_fwd', ptr(bass_m), 12, ptr(bass_t), B * 8, 12, -1, ptr(sums[6:]), st)
        out = _empty(11, dev=dev)
        ctx.scal = (0.0, 0.0, 0.0, 1.0, float(B * 8), float(B * 96))
        call('ptv_loss_finalize', ptr(sums), ptr(counts), *ctx.scal, ptr(out), st)
        ctx.save_for_backward(root_m, chroma_m, bass_m, root_t, chroma_t, bass_t, counts)
        ctx.sm = sm
        return out[7:11].clone()

    @staticmethod
    def backward(ctx, g4):
        root_m, chroma_m, bass_m, root_t, chroma_t, bass_t, counts = ctx.saved_tensors
        dev = root_m.device
        st = stream_ptr()
        g11 = _zeros(11, dev=dev)
        copy2d(g11[7:11].view(1, 4), g4.contiguous().view(1, 4))
        gs = _empty(8, dev=dev)
        call('ptv_loss_bwd_scales', ptr(g11), ptr(counts), *ctx.scal, ptr(gs), st)
        droot, dchroma, dbass = torch.empty_like(root_m), torch.empty_like(chroma_m), torch.empty_like(bass_m)
        call('ptv_ce_bwd', ptr(root_m), 12, ptr(root_t), root_t.numel(), 12, -1, ptr(gs[4:]), ptr(droot), 12, st)
        call('ptv_ce_bwd', ptr(chroma_m), 2, ptr(chroma_t), chroma_t.numel(), 2, -1, ptr(gs[5:]), ptr(dchroma), 2, st)
        call('ptv_ce_bwd', ptr(bass_m), 12, ptr(bass_t), bass_t.numel(), 12, -1, ptr(gs[6:]), ptr(dbass), 12, st)
        if ctx.sm:
            droot, dchroma, dbass = (t.permute(*_chord_perm(t)) for t in (droot, dchroma, dbass))
        return droot, dchroma, dbass, None
