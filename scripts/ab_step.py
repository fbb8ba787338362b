"""Same-process A/B of the headline train step (B = 512, bf16, teacher-forced) over values of a module-level switch of
`functional` (or an environment-free attribute), alternating the settings so that box drift cancels:
    python scripts/ab_step.py PERSIST_SPLITK=0,2,4 [--tfr 1.0] [--batch 512] [--rounds 3] [--steps 12]
Prints ms/step per setting and round."""
import argparse
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonic_chord_texture_disentanglement_amd import functional as F_  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.optim import FusedClipAdam  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('switch')
    ap.add_argument('--tfr', type=float, default=1.0)
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--steps', type=int, default=12)
    ap.add_argument('--module', default='functional')
    a = ap.parse_args()
    name, vals = a.switch.split('=')
    vals = [eval(v) for v in vals.split(',')]
    mod = F_
    if a.module != 'functional':
        import importlib
        mod = importlib.import_module('polyphonic_chord_texture_disentanglement_amd.' + a.module)
    assert hasattr(mod, name), name
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    m = DisentangleVAE.init_model(dev).to(dev).set_precision('bf16')
    m.use_philox(7, 0)
    random.seed(7)
    opt = FusedClipAdam(m.parameters(), lr=1e-3)
    from polyphonic_chord_texture_disentanglement_amd.optim import reserve_step_memory
    reserve_step_memory(a.batch, dev)                          # (no hipMalloc -- a 70-ms device-wide sync -- inside a timed round)
    data = [tuple(torch.from_numpy(t).to(dev) for t in synth_batch(a.batch, 1234 + i)) for i in range(2)]

    def step(i):
        x, c, pr = data[i % 2]
        opt.zero_grad()
        o = m('train', x, c, pr, tfr1=a.tfr, tfr2=a.tfr, tfr3=a.tfr, beta=0.1, weights=[1, 0.5])
        o[0].backward()
        opt.clip_and_step(1.0)
        return o

    for i in range(4):                                         # warm-up, then no full garbage collection inside a timed round (optim.freeze_gc)
        step(i)
    from polyphonic_chord_texture_disentanglement_amd.optim import freeze_gc
    freeze_gc()
    res = {v: [] for v in vals}
    for r in range(a.rounds):
        for v in vals:
            setattr(mod, name, v)
            if hasattr(mod, '_apply_switches'):
                mod._apply_switches()
            for i in range(3):
                o = step(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(a.steps):
                o = step(i)
            torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t0) / a.steps * 1e3)
            F_.persist_check()
    for v in vals:
        print('%s=%r: %s ms/step  (best %.3f, %.0f samples/s)  loss %.5f' % (name, v, ' '.join('%.3f' % t for t in res[v]), min(res[v]),
                                                                         a.batch / min(res[v]) * 1e3, o[0].item()), flush=True)


if __name__ == '__main__':
    main()
