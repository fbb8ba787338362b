"""Same-process A/B of COMBINATIONS of module-level switches (functional / model / ptvae), alternating per round like ab_step.py:
    python scripts/ab_combo.py "model.CHD_ENC_SLOT=1,functional.BIGRU_SLOT_BWD=7" "model.CHD_ENC_SLOT=0,functional.BIGRU_SLOT_BWD=4" ..."""
import importlib, os, random, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PK = 'polyphonic_chord_texture_disentanglement_amd.'
from polyphonic_chord_texture_disentanglement_amd import functional as F_
from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE
from polyphonic_chord_texture_disentanglement_amd.optim import FusedClipAdam, reserve_step_memory
from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch
combos = []
for arg in sys.argv[1:]:
    c = []
    for kv in arg.split(','):
        k, v = kv.split('='); mod, name = k.split('.')
        val = tuple(int(ch) for ch in v[1:]) if v.startswith('t') and v[1:].isdigit() else eval(v)      # t0123 = the tuple (0, 1, 2, 3)
        c.append((importlib.import_module(PK + mod), name, val))
    combos.append((arg, c))
dev = torch.device('cuda:0'); torch.manual_seed(0)
m = DisentangleVAE.init_model(dev).to(dev).set_precision('bf16'); m.use_philox(7, 0); random.seed(7)
opt = FusedClipAdam(m.parameters(), lr=1e-3); reserve_step_memory(512, dev)
data = [tuple(torch.from_numpy(t).to(dev) for t in synth_batch(512, 1234 + i)) for i in range(2)]
def step(i):
    x, c, pr = data[i % 2]; opt.zero_grad()
    o = m('train', x, c, pr, tfr1=1., tfr2=1., tfr3=1., beta=0.1, weights=[1, 0.5]); o[0].backward(); opt.clip_and_step(1.0)
res = {a: [] for a, _ in combos}
for r in range(int(os.environ.get('ROUNDS', 3))):
    for a, c in combos:
        for mod, name, v in c:
            assert hasattr(mod, name), name
            setattr(mod, name, v)
        for i in range(3): step(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(12): step(i)
        torch.cuda.synchronize(); res[a].append((time.perf_counter() - t0) / 12 * 1e3)
        F_.persist_check()
for a, _ in combos:
    print('%-70s %s  best %.3f' % (a, ' '.join('%.3f' % t for t in res[a]), min(res[a])), flush=True)
