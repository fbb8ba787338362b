"""Feasibility probe: capture the WHOLE teacher-forced train step (zero_grad, forward, backward, clip+Adam) into one hipGraph and
compare replay time with the eager step (same process).  Usage: python scripts/try_graph_step.py [B] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import faulthandler
faulthandler.enable()
import torch  # noqa: E402

from polyphonic_chord_texture_disentanglement_amd import functional as F_  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.optim import FusedClipAdam  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
MODE = sys.argv[3] if len(sys.argv) > 3 else 'full'
if os.environ.get('NO_OVERLAP') == '1':
    F_.OVERLAP = False
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = DisentangleVAE.init_model(dev).to(dev).set_precision('bf16')
opt = FusedClipAdam(m.parameters(), lr=1e-3)
x, c, pr = (torch.from_numpy(a).to(dev) for a in synth_batch(B, 1234))
eps = {'chd': torch.randn(B, 256, device=dev), 'rhy': torch.randn(B, 256, device=dev)}
m.eps_source = lambda name, shape, device: eps[name]


def step():
    opt.zero_grad()
    if MODE == 'fwd':
        with torch.no_grad():
            return m('train', x, c, pr, tfr1=1., tfr2=1., tfr3=1., beta=0.1, weights=[1, 0.5])
    out = m('train', x, c, pr, tfr1=1., tfr2=1., tfr3=1., beta=0.1, weights=[1, 0.5])
    if MODE == 'fwdgrad':
        return out
    out[0].backward()
    if MODE != 'fwdbwd':
        opt.clip_and_step(1.0)
    return out


def timeit(fn, k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3, th / k * 1e3


for _ in range(3):
    step()
print('eager  ms/step %.3f (host %.3f)' % timeit(step, K), flush=True)
loss_eager = float(step()[0])


from polyphonic_chord_texture_disentanglement_amd.graph_step import GraphedTrainStep
m.eps_source = None
m.use_philox(7, 0)
gs = GraphedTrainStep(m, opt, B)
l = gs(x, c, pr)
torch.cuda.synchronize()
print('captured + first replay, losses', [round(float(v), 4) for v in l], flush=True)
print('graph  ms/step %.3f (host %.3f)' % timeit(lambda: gs(x, c, pr), K), flush=True)
F_.persist_check()
print('loss after %d replays %.5f' % (K + 1, float(l[0])))
m2 = DisentangleVAE.init_model(dev).to(dev).set_precision('bf16')
