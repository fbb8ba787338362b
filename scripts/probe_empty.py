"""census of torch.empty / zeros / full calls of one train step by call site (host cost: ~7 us each under the caching allocator)"""
import collections, os, random, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonic_chord_texture_disentanglement_amd import functional as F_
from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE
from polyphonic_chord_texture_disentanglement_amd.optim import FusedClipAdam
from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch
dev = torch.device('cuda:0')
m = DisentangleVAE.init_model(dev).to(dev).set_precision('bf16'); m.use_philox(7, 0); random.seed(7)
opt = FusedClipAdam(m.parameters(), lr=1e-3)
x, c, pr = (torch.from_numpy(t).to(dev) for t in synth_batch(512, 1234))
def step():
    opt.zero_grad(); o = m('train', x, c, pr, tfr1=1., tfr2=1., tfr3=1., beta=0.1, weights=[1, 0.5]); o[0].backward(); opt.clip_and_step(1.0)
for _ in range(3): step()
cnt = collections.Counter(); byt = collections.Counter()
for name in ('empty', 'zeros', 'full', 'empty_like', 'zeros_like'):
    orig = getattr(torch, name)
    def wrap(*a, _o=orig, _n=name, **k):
        t = _o(*a, **k)
        fr = traceback.extract_stack(limit=4)
        site = ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in reversed(fr[:-1]))
        cnt[(_n, site)] += 1; byt[(_n, site)] += t.numel() * t.element_size()
        return t
    setattr(torch, name, wrap)
step(); torch.cuda.synchronize()
print('total', sum(cnt.values()))
for k, v in cnt.most_common(45):
    print('%3d  %8.1f MB  %s  %s' % (v, byt[k] / 2**20, k[0], k[1]))
