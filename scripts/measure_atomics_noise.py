"""Run-to-run noise floor of the weight gradients (fp32 atomics in wgrad.hip / split-K gemm.hip): the SAME teacher-forced step N
times in one process, per-parameter max |g_i - g_0| / max |g_0| and the relative spread of the global norm.  Feeds
tests/helpers.ATOMICS_RTOL.  Usage: python scripts/measure_atomics_noise.py [reduced|full] [fp32|bf16] [runs] [B]"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

from polyphonic_chord_texture_disentanglement_amd.optim import FusedClipAdam  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch  # noqa: E402


def main():
    geo = sys.argv[1] if len(sys.argv) > 1 else 'reduced'
    prec = sys.argv[2] if len(sys.argv) > 2 else 'fp32'
    runs = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    B = int(sys.argv[4]) if len(sys.argv) > 4 else (6 if geo == 'reduced' else 64)
    dev = 'cuda:0'
    if geo == 'reduced':
        from test_host_surface import build_reduced
        m = build_reduced(dev).to(dev)
    else:
        from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE
        m = DisentangleVAE.init_model(torch.device(dev)).to(dev)
    m.set_precision(prec)
    opt = FusedClipAdam(m.parameters(), lr=1e-3)
    x, c, pr = (torch.from_numpy(a).to(dev) for a in synth_batch(B, 321))
    flats, norms, losses = [], [], []
    for i in range(runs):
        m.use_philox(seed=7, sample_offset=0)
        random.seed(7)
        opt.zero_grad()
        ls = m('train', x, c, pr, tfr1=1., tfr2=1., tfr3=1., beta=0.1, weights=[1, 0.5])
        ls[0].backward()
        torch.cuda.synchronize()
        flats.append(opt.arena.flat.clone())
        norms.append(float(opt.arena.flat.double().pow(2).sum().sqrt()))
        losses.append([float(v) for v in ls])
    worst, worst_name = 0.0, None
    names = [n for n, _ in m.named_parameters()]
    for n, p, o in zip(names, opt.arena.params, opt.arena.offsets):
        g0 = flats[0][o:o + p.numel()]
        mx = float(g0.abs().max())
        for f in flats[1:]:
            d = float((f[o:o + p.numel()] - g0).abs().max()) / max(mx, 1e-30)
            if d > worst:
                worst, worst_name = d, n
    print(json.dumps({'geometry': geo, 'precision': prec, 'B': B, 'runs': runs, 'max_rel_to_tensor_max': worst, 'tensor': worst_name,
                      'gnorm_rel_spread': (max(norms) - min(norms)) / norms[0],
                      'loss_abs_spread': max(max(abs(a - b) for a, b in zip(l, losses[0])) for l in losses)}))


if __name__ == '__main__':
    main()
