import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
from helpers import full_params
from polyphonic_chord_texture_disentanglement_amd import functional as F_, model as M
from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch
DEV = 'cuda:0'
x, c, pr = (torch.from_numpy(a).to(DEV) for a in synth_batch(16, 99))
for comp in (True, False):
    F_.DEC_COMPOSITE = comp
    m = M.DisentangleVAE.init_model(torch.device(DEV)); m.load_state_dict(full_params()); m.to(DEV).set_precision('bf16')
    for dead in (False, True):
        F_.DEAD_STEPS = dead; F_.POISON_DEAD_STEPS = dead
        m.use_philox(5, 0); m.zero_grad()
        losses = m.loss(x, c, pr, 1., 1., 1., 0.1, [1, 0.5])
        losses[0].backward(); torch.cuda.synchronize()
        bad = [k for k, p in m.named_parameters() if not torch.isfinite(p.grad).all()]
        print('composite', comp, 'dead', dead, 'loss', float(losses[0].detach()), 'nonfinite:', bad, flush=True)
