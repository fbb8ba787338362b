"""Every ptv_gemm call of one teacher-forced B=512 bf16 train step with its shape and its serialized duration (HIP events,
synchronised per call): which products the step's GEMM time is made of.  python scripts/gemm_shapes.py [B]"""
import collections
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonic_chord_texture_disentanglement_amd import functional as F_  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.optim import FusedClipAdam  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device('cuda:0')
torch.manual_seed(0)
random.seed(7)
m = DisentangleVAE.init_model(dev).to(dev).set_precision('bf16')
m.use_philox(7, 0)
opt = FusedClipAdam(m.parameters(), lr=1e-3)
data = tuple(torch.from_numpy(a).to(dev) for a in synth_batch(B, 99))


def step():
    opt.zero_grad()
    o = m('train', *data, tfr1=1.0, tfr2=1.0, tfr3=1.0, beta=0.1, weights=[1, 0.5])
    o[0].backward()
    opt.clip_and_step(1.0)


for _ in range(3):
    step()
torch.cuda.synchronize()
rec = collections.OrderedDict()
orig = F_.call


def wrapped(name, *args):
    if name != 'ptv_gemm':
        return orig(name, *args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s = torch.cuda.current_stream()
    e0.record(s)
    r = orig(name, *args)
    e1.record(s)
    torch.cuda.synchronize()
    prec, ta, tb, M, N, K = args[:6]
    key = (ta, tb, M, N, K, args[14], args[17])
    t = rec.setdefault(key, [0, 0.0])
    t[0] += 1
    t[1] += e0.elapsed_time(e1) * 1e3
    return r


F_.call = wrapped
step()
F_.call = orig
rows = sorted(rec.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in rec.values())
print('total gemm time (serialized) %.0f us over %d calls' % (tot, sum(v[0] for v in rec.values())))
print('%3s %3s %7s %6s %7s %3s %3s %5s %9s %8s %7s' % ('tA', 'tB', 'M', 'N', 'K', 'acc', 'dt', 'calls', 'us total', 'us/call', 'TF/s'))
for (ta, tb, M, N, K, acc, dt), (n, us) in rows:
    print('%3d %3d %7d %6d %7d %3d %3d %5d %9.0f %8.1f %7.1f' % (ta, tb, M, N, K, acc, dt, n, us, us / n, 2.0 * M * N * K * n / us / 1e6))
