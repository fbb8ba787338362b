import os, sys, traceback
sys.path.insert(0, os.getcwd())
import torch
from polyphonic_chord_texture_disentanglement_amd import functional as F_
orig = F_.colsum
def colsum(out, a, sel=None, groups=1):
    lda = F_._ld(a); N = a.shape[1]
    vec = (lda & 3) == 0 and lda >= ((N + 3) & ~3) and (a.data_ptr() & (7 if a.dtype == torch.bfloat16 else 15)) == 0
    if not vec:
        fr = traceback.extract_stack(limit=6)
        print('SLOW colsum rows=%d N=%d lda=%d dtype=%s ptr%%16=%d  <- %s' % (a.shape[0], N, lda, a.dtype, a.data_ptr() % 16,
              ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in reversed(fr[:-1]))), flush=True)
    return orig(out, a, sel, groups)
F_.colsum = colsum
from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE
from polyphonic_chord_texture_disentanglement_amd.optim import FusedClipAdam
from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch
dev = torch.device('cuda:0')
m = DisentangleVAE.init_model(dev).to(dev).set_precision('bf16')
opt = FusedClipAdam(m.parameters(), lr=1e-3)
x, c, pr = (torch.from_numpy(a).to(dev) for a in synth_batch(512, 1234))
m.use_philox(7, 0)
for tfr in (1., 0.):
    print('tfr', tfr)
    opt.zero_grad()
    out = m('train', x, c, pr, tfr1=tfr, tfr2=tfr, tfr3=tfr, beta=0.1, weights=[1, 0.5])
    out[0].backward()
    opt.clip_and_step(1.0)
torch.cuda.synchronize()
