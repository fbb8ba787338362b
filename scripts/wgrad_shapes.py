"""The weight-gradient products of one B = 512 bf16 train step: N1 x N2 x K, operand dtypes, stream, ready -> end (events).
python scripts/wgrad_shapes.py"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonic_chord_texture_disentanglement_amd import functional as F_  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd import _lib  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.optim import FusedClipAdam  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
random.seed(7)
m = DisentangleVAE.init_model(dev).to(dev).set_precision('bf16')
m.use_philox(7, 0)
opt = FusedClipAdam(m.parameters(), lr=1e-3)
data = tuple(torch.from_numpy(a).to(dev) for a in synth_batch(512, 99))
REC = None
orig = _lib.call


def traced(name, *args):
    if REC is None or name not in ('ptv_wgrad', 'ptv_gemm', 'ptv_gemm_mtop'):
        return orig(name, *args)
    s = F_.cur_stream()
    e0 = torch.cuda.Event(enable_timing=True); e0.record(s)
    r = orig(name, *args)
    e1 = torch.cuda.Event(enable_timing=True); e1.record(s)
    REC.append((name, args, s.cuda_stream & 0xffff, e0, e1))
    return r


F_.call = traced


def step():
    opt.zero_grad()
    o = m('train', *data, tfr1=1.0, tfr2=1.0, tfr3=1.0, beta=0.1, weights=[1, 0.5])
    o[0].backward()
    opt.clip_and_step(1.0)


for _ in range(4):
    step()
torch.cuda.synchronize()
step()
t0 = torch.cuda.Event(enable_timing=True); t0.record(F_.cur_stream())
REC = []
step()
rec, REC = REC, None
torch.cuda.synchronize()
tot = 0.0
print('%-14s %6s %6s %8s  %-9s %5s %8s %8s %7s' % ('call', 'N1/M', 'N2/N', 'K', 'dtypes', 'strm', 'ready', 'end', 'us'))
for name, a, s, e0, e1 in rec:
    us = e0.elapsed_time(e1) * 1e3
    if name == 'ptv_wgrad':
        n1, n2, k, dt = a[0], a[1], a[2], a[11]
        desc = 'dy=%s x=%s' % ('bf16' if dt & 1 else 'f32', 'bf16' if dt & 2 else 'f32')
        tot += us
    else:
        n1, n2, k, dt = a[3], a[4], a[5], a[17]
        desc = 'a=%s b=%s o=%s' % ('bf' if dt & 1 else 'f32', 'bf' if dt & 2 else 'f32', 'bf' if dt & 4 else 'f32')
        if us < 40:
            continue
    print('%-14s %6d %6d %8d  %-18s %4x %8.3f %8.3f %7.1f' % (name, n1, n2, k, desc, s, t0.elapsed_time(e0), t0.elapsed_time(e1), us))
print('ptv_wgrad total (ready->end) %.1f us' % tot)
