import sys, random, torch, traceback, collections
sys.path.insert(0, '.')
from polyphonic_chord_texture_disentanglement_amd import functional as F_
from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE
from polyphonic_chord_texture_disentanglement_amd.optim import FusedClipAdam
from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch
dev = torch.device('cuda:0'); torch.manual_seed(0); random.seed(7)
m = DisentangleVAE.init_model(dev).to(dev).set_precision('bf16'); m.use_philox(7, 0)
opt = FusedClipAdam(m.parameters(), lr=1e-3)
data = tuple(torch.from_numpy(a).to(dev) for a in synth_batch(512, 99))
def step():
    opt.zero_grad()
    o = m('train', *data, tfr1=1.0, tfr2=1.0, tfr3=1.0, beta=0.1, weights=[1, 0.5])
    o[0].backward(); opt.clip_and_step(1.0)
for _ in range(3): step()
log = collections.Counter()
oz, ozeros, ofull = torch.Tensor.zero_, torch.zeros, torch.Tensor.fill_
def where():
    for f in traceback.extract_stack()[::-1]:
        if 'polyphonic' in f.filename: return '%s:%d' % (f.filename.split('/')[-1], f.lineno)
    return '?'
def z(self):
    if self.is_cuda and self.numel() * self.element_size() > 1 << 20: log[(where(), self.numel() * self.element_size())] += 1
    return oz(self)
def zs(*a, **k):
    t = ozeros(*a, **k)
    if t.is_cuda and t.numel() * t.element_size() > 1 << 20: log[(where(), t.numel() * t.element_size())] += 1
    return t
def wrapf(name, obj):
    o = getattr(obj, name)
    def f(*a, **k):
        t = o(*a, **k)
        if isinstance(t, torch.Tensor) and t.is_cuda and t.numel() * t.element_size() > 1 << 20: log[(name + ' ' + where(), t.numel() * t.element_size())] += 1
        return t
    setattr(obj, name, f)
for nm in ('full', 'ones', 'zeros_like', 'full_like', 'ones_like'): wrapf(nm, torch)
for nm in ('fill_', 'new_zeros', 'new_full', 'new_ones'): wrapf(nm, torch.Tensor)
torch.Tensor.zero_ = z; torch.zeros = zs
step(); torch.cuda.synchronize()
for (w, n), c in sorted(log.items(), key=lambda x: -x[0][1]): print('%-40s %8.1f MB x %d' % (w, n / 1e6, c))
