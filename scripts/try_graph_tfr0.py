"""Probe: the free-running (tfr = 0) train step replayed from a whole-step hipGraph vs eager."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from polyphonic_chord_texture_disentanglement_amd import functional as F_
from polyphonic_chord_texture_disentanglement_amd.graph_step import GraphedTrainStep
from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE
from polyphonic_chord_texture_disentanglement_amd.optim import FusedClipAdam
from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = DisentangleVAE.init_model(dev).to(dev).set_precision('bf16')
opt = FusedClipAdam(m.parameters(), lr=1e-3)
x, c, pr = (torch.from_numpy(a).to(dev) for a in synth_batch(B, 1234))
m.use_philox(7, 0)

def step():
    opt.zero_grad()
    out = m('train', x, c, pr, tfr1=0., tfr2=0., tfr3=0., beta=0.1, weights=[1, 0.5])
    out[0].backward()
    opt.clip_and_step(1.0)
    return out

def timeit(fn, k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3, th / k * 1e3

for _ in range(3): step()
print('eager ms/step %.3f (host %.3f)' % timeit(step, 6), flush=True)
le = float(step()[0].detach())
gs = GraphedTrainStep(m, opt, B, tfr=(0., 0., 0.))
l = gs(x, c, pr)
torch.cuda.synchronize()
print('captured; loss %.4f (eager %.4f)' % (float(l[0]), le), flush=True)
print('graph ms/step %.3f (host %.3f)' % timeit(lambda: gs(x, c, pr), 6), flush=True)
F_.persist_check()
print('loss %.4f' % float(l[0]))
