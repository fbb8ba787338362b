"""Per-launch GPU timeline of the EAGER B = 512 step without a profiler (rocprofv3's per-launch host cost makes the eager step host-bound
and its timeline a picture of the host): every library call is bracketed by two events on its stream.  `ready` = the moment the stream
reached the launch (everything queued before it on that stream is done), `end` = the launch finished; end - ready = queueing for CUs / LDS
+ execution.  One untraced step is in flight when the traced one is enqueued, as in steady state.
    python scripts/trace_calls.py [t_from_ms] [t_to_ms]       (default: the first 1.4 ms of the step)"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonic_chord_texture_disentanglement_amd import functional as F_  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd import functional_free as FF_  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd import _lib  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.optim import FusedClipAdam  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd import optim as O_  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch  # noqa: E402

T0 = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
T1 = float(sys.argv[2]) if len(sys.argv) > 2 else 1.4
B = int(os.environ.get('B', 512))
dev = torch.device('cuda:0')
torch.manual_seed(0)
random.seed(7)
m = DisentangleVAE.init_model(dev).to(dev).set_precision('bf16')
m.use_philox(7, 0)
opt = FusedClipAdam(m.parameters(), lr=1e-3)
data = tuple(torch.from_numpy(a).to(dev) for a in synth_batch(B, 99))
REC = None
orig_call = _lib.call


def traced_call(name, *args):
    if REC is None:
        return orig_call(name, *args)
    s = F_.cur_stream()
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record(s)
    r = orig_call(name, *args)
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record(s)
    REC.append((name, s.cuda_stream & 0xffff, e0, e1))
    return r


def wrap_fn(mod, fname):
    f = getattr(mod, fname)

    def g(*a, **k):
        if REC is None:
            return f(*a, **k)
        s = F_.cur_stream()
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record(s)
        r = f(*a, **k)
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(s)
        REC.append((fname, s.cuda_stream & 0xffff, e0, e1))
        return r
    setattr(mod, fname, g)


for mod in (F_, FF_, O_):
    if hasattr(mod, 'call'):
        mod.call = traced_call
wrap_fn(F_, 'gru_persist_fwd')
wrap_fn(F_, 'gru_persist_bwd')


def step():
    opt.zero_grad()
    o = m('train', *data, tfr1=1.0, tfr2=1.0, tfr3=1.0, beta=0.1, weights=[1, 0.5])
    o[0].backward()
    opt.clip_and_step(1.0)


for _ in range(5):
    step()
torch.cuda.synchronize()
step()                                   # in flight
start = torch.cuda.Event(enable_timing=True)
start.record(F_.cur_stream())
REC = []
step()
rec, REC = REC, None
end = torch.cuda.Event(enable_timing=True)
end.record(F_.cur_stream())
torch.cuda.synchronize()
print('traced step: %.3f ms, %d library calls' % (start.elapsed_time(end), len(rec)))
streams = {}
rows = []
for name, s, e0, e1 in rec:
    streams.setdefault(s, 's%d' % len(streams))
    rows.append((start.elapsed_time(e0), start.elapsed_time(e1), streams[s], name))
print('%9s %9s %8s  %-4s %s' % ('ready ms', 'end ms', 'us', 'strm', 'call'))
for r0, r1, s, name in sorted(rows):
    if r1 >= T0 and r0 <= T1:
        print('%9.3f %9.3f %8.1f  %-4s %s' % (r0, r1, (r1 - r0) * 1e3, s, name))
