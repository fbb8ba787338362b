"""Where the backward of the teacher-forced B=512 step spends its time on the MAIN stream, unprofiled: functional.mark() events.
PTV_BWD_COMPOSITES=0 python scripts/trace_marks.py    (the marks inside the two decoders' backward passes exist on the launch-by-launch path
only; with the one-call C entry points -- the default -- a node is a single mark)"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonic_chord_texture_disentanglement_amd import functional as F_  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.optim import FusedClipAdam  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch  # noqa: E402

B = int(os.environ.get('BATCH', '512'))
TFR = float(os.environ.get('TFR', '1'))          # 0 = free-running training (train.py's schedule from its third batch on)
dev = torch.device('cuda:0')
torch.manual_seed(0)
random.seed(7)
m = DisentangleVAE.init_model(dev).to(dev).set_precision('bf16')
m.use_philox(7, 0)
opt = FusedClipAdam(m.parameters(), lr=1e-3)
data = tuple(torch.from_numpy(a).to(dev) for a in synth_batch(B, 99))


def step():
    opt.zero_grad()
    F_.mark('step:start')
    o = m('train', *data, tfr1=TFR, tfr2=TFR, tfr3=TFR, beta=0.1, weights=[1, 0.5])
    F_.mark('fwd:end')
    o[0].backward()
    F_.mark('bwd:end')
    opt.clip_and_step(1.0)
    F_.mark('opt:end')


for _ in range(5):
    step()
torch.cuda.synchronize()
from polyphonic_chord_texture_disentanglement_amd.optim import freeze_gc  # noqa: E402
freeze_gc()                                  # (no full garbage collection inside a traced step)
acc = {}
N = 10
for _ in range(N):
    if os.environ.get('SYNC_START') == '1':
        torch.cuda.synchronize()             # traced step starts on an idle GPU: host and GPU columns share their origin (who waits for whom?)
    else:
        step()                               # an untraced step in flight: the traced one is enqueued while the GPU is busy, as in steady state
    F_.TRACE = []
    step()
    torch.cuda.synchronize()
    tr = F_.TRACE
    F_.TRACE = None
    t0, h0 = tr[0][1], tr[0][2]
    for name, e, h in tr:
        acc.setdefault(name, []).append((t0.elapsed_time(e), (h - h0) * 1e3))
print('%-28s %10s %10s   (the host enqueues a step while the GPU still runs the previous one: only differences matter)' % ('mark', 'GPU ms', 'host ms'))
for name, t, h in sorted(((n, sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v)) for n, v in acc.items()), key=lambda x: x[1]):
    print('%-28s %10.3f %10.3f' % (name, t, h))
