"""GPU time of the phases of the teacher-forced B=512 bf16 train step, from events on the main stream in an unprofiled run
(the rocprof trace is host-bound and distorts the picture): forward+loss, backward, clip+Adam.  python scripts/phase_times.py"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.optim import FusedClipAdam  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
TFR = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
dev = torch.device('cuda:0')
torch.manual_seed(0)
random.seed(7)
m = DisentangleVAE.init_model(dev).to(dev).set_precision('bf16')
m.use_philox(7, 0)
opt = FusedClipAdam(m.parameters(), lr=1e-3)
data = tuple(torch.from_numpy(a).to(dev) for a in synth_batch(B, 99))
N = 20
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(N)]


def step(e=None):
    opt.zero_grad()
    if e:
        e[0].record()
    o = m('train', *data, tfr1=TFR, tfr2=TFR, tfr3=TFR, beta=0.1, weights=[1, 0.5])
    if e:
        e[1].record()
    o[0].backward()
    if e:
        e[2].record()
    opt.clip_and_step(1.0)
    if e:
        e[3].record()


for _ in range(5):
    step()
torch.cuda.synchronize()
for i in range(N):
    step(ev[i])
torch.cuda.synchronize()
f = sum(e[0].elapsed_time(e[1]) for e in ev) / N
b = sum(e[1].elapsed_time(e[2]) for e in ev) / N
a = sum(e[2].elapsed_time(e[3]) for e in ev) / N
tot = ev[0][0].elapsed_time(ev[-1][3]) / (N - 1 + 1e-9) * (N - 1) / (N - 1)
print('forward+loss %.2f ms  backward %.2f ms  clip+adam %.2f ms   step (event to event) %.2f ms' % (f, b, a, ev[0][0].elapsed_time(ev[-1][0]) / (N - 1)))
