"""Which ATen operators still launch kernels inside one eager teacher-forced train step, and from which line of the package?
A TorchDispatchMode records every aten op that touches a CUDA tensor (name, output elements, innermost package frame) over one
step at B = 512 bf16.  Views / metadata ops are dropped.  Usage: python scripts/aten_sites.py [B]"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.optim import FusedClipAdam  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = DisentangleVAE.init_model(dev).to(dev).set_precision('bf16')
opt = FusedClipAdam(m.parameters(), lr=1e-3)
x, c, pr = (torch.from_numpy(a).to(dev) for a in synth_batch(B, 1234))
m.use_philox(7, 0)

NO_KERNEL = ('view', 'as_strided', 'select', 'slice', 'narrow', 'transpose', 'permute', 't.default', 'unsqueeze', 'squeeze', 'expand',
             'empty', 'detach', 'alias', 'reshape', 'unbind', 'split', 'record_stream', '_unsafe_view', 'is_pinned', 'lift_fresh',
             'resize_', 'set_', 'unfold', 'chunk', 'size', 'stride', 'sym_', 'prim', '_local_scalar_dense', 'is_same_size')
sites = collections.Counter()
elems = collections.Counter()


class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(k in name for k in NO_KERNEL):
            return out
        ts = [t for t in list(args) + [out] if isinstance(t, torch.Tensor)]
        if not any(t.is_cuda for t in ts):
            return out
        fr = [f for f in traceback.extract_stack() if 'polyphonic_chord_texture_disentanglement_amd' in f.filename
              and 'scripts' not in f.filename]
        where = '%s:%d' % (os.path.basename(fr[-1].filename), fr[-1].lineno) if fr else '?'
        n = out.numel() if isinstance(out, torch.Tensor) else 0
        sites[(name, where)] += 1
        elems[(name, where)] += n
        return out


def step():
    opt.zero_grad()
    out = m('train', x, c, pr, tfr1=1., tfr2=1., tfr3=1., beta=0.1, weights=[1, 0.5])
    out[0].backward()
    opt.clip_and_step(1.0)


for _ in range(3):
    step()
torch.cuda.synchronize()
with torch.autograd.set_multithreading_enabled(False):
    with Rec():
        step()
torch.cuda.synchronize()
tot = collections.Counter()
for (name, where), k in sorted(sites.items(), key=lambda kv: (-kv[1], kv[0])):
    print('%3d x %-34s %-28s %12d elems' % (k, name.replace('aten.', ''), where, elems[(name, where)]))
    tot[name] += k
print('--- by op:', ', '.join('%s %d' % (n.replace('aten.', ''), k) for n, k in tot.most_common()))
