"""Generate golden vectors by running the REFERENCE itself (build container only).

Imports `/root/reference` unmodified (two stub modules stand in for the absent third-party
`pretty_midi` and `tensorboardX`, which are only touched by off-path MIDI / logging helpers),
builds `DisentangleVAE` in a reduced and in the full `init_model()` configuration, runs the
train-step path on the synthetic generator with recorded eps / coin flips, and writes small
`.npz` fixtures next to this script.  Nothing of the reference is copied: fixtures hold inputs
and expected outputs only.

    python tests/golden/make_golden.py        # needs /root/reference; ~2 min on 8 vCPU
"""
import os
import random
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference'

from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch, fill_state_dict  # noqa: E402


def import_reference():
    pm = types.ModuleType('pretty_midi')
    for n in ('Note', 'PrettyMIDI', 'Instrument'):
        setattr(pm, n, type(n, (), {'__init__': lambda self, *a, **k: None}))
    sys.modules['pretty_midi'] = pm
    tb = types.ModuleType('tensorboardX')
    tb.SummaryWriter = type('SummaryWriter', (), {'__init__': lambda self, *a, **k: None,
                                                  'add_scalar': lambda self, *a, **k: None})
    sys.modules['tensorboardX'] = tb
    sys.path.insert(0, REF)
    import model as ref_model      # noqa
    import ptvae as ref_ptvae      # noqa
    import amc_dl.torch_plus as ref_tp   # noqa
    return ref_model, ref_ptvae, ref_tp


class EpsRecorder:
    """Record the N(0,1) draws of Normal.rsample (train_utils.py:33-34) without touching the
    reference: rsample == loc + scale * randn(shape) drawn from the global generator."""

    def __init__(self):
        self.eps = []
        self._orig = torch.distributions.Normal.rsample

    def __enter__(self):
        rec = self

        def rsample(self_, sample_shape=torch.Size()):
            shape = self_._extended_shape(sample_shape)
            e = torch.randn(shape, dtype=self_.loc.dtype)
            rec.eps.append(e.clone())
            return self_.loc + e * self_.scale
        torch.distributions.Normal.rsample = rsample
        return self

    def __exit__(self, *a):
        torch.distributions.Normal.rsample = self._orig


class CoinRecorder:
    def __init__(self, seed):
        self.rng = random.Random(seed)
        self.coins = []
        self._orig = random.random

    def __enter__(self):
        def rnd():
            v = self.rng.random()
            self.coins.append(v)
            return v
        random.random = rnd
        return self

    def __exit__(self, *a):
        random.random = self._orig


def build_reduced(ref_model, ref_ptvae):
    torch.manual_seed(0)
    dev = torch.device('cpu')
    chd_enc = ref_ptvae.RnnEncoder(36, 32, 16)
    rhy_enc = ref_ptvae.TextureEncoder(24, 32, 16, 3)
    chd_dec = ref_ptvae.RnnDecoder(z_input_dim=16, hidden_dim=24, z_dim=16)
    dec = ref_ptvae.PtvaeDecoder(device=dev, note_emb_size=20, z_size=32, dec_emb_hid_size=12,
                                 dec_time_hid_size=40, dec_notes_hid_size=28, dec_z_in_size=16,
                                 dec_dur_hid_size=8)
    return ref_model.DisentangleVAE('disvae', dev, chd_enc, rhy_enc, dec, chd_dec)


def run_case(m, B, data_seed, rng_seed, tfr, beta=0.1, weights=(1, 0.5), with_grads=True):
    x, c, pr = synth_batch(B, data_seed)
    xt, ct, prt = torch.from_numpy(x), torch.from_numpy(c), torch.from_numpy(pr)
    m.zero_grad()
    torch.manual_seed(rng_seed)
    with EpsRecorder() as er, CoinRecorder(rng_seed) as cr:
        outs = m.run(xt, ct, prt, *tfr)
        losses = m.loss_function(xt, ct, *outs, beta, list(weights))
    pitch_outs, dur_outs, dist_chd, dist_rhy, root, chroma, bass = outs
    res = OrderedDict(
        x=x, c=c, pr_mat=pr, eps_chd=er.eps[0].numpy(), eps_rhy=er.eps[1].numpy(),
        coins=np.array(cr.coins, dtype=np.float64), tfr=np.array(tfr, dtype=np.float64),
        beta=np.float64(beta), weights=np.array(weights, dtype=np.float64),
        losses=np.array([l.item() for l in losses], dtype=np.float64),
        pitch_outs=pitch_outs.detach().numpy(), dur_outs=dur_outs.detach().numpy(),
        mu_chd=dist_chd.mean.detach().numpy(), std_chd=dist_chd.scale.detach().numpy(),
        mu_rhy=dist_rhy.mean.detach().numpy(), std_rhy=dist_rhy.scale.detach().numpy(),
        recon_root=root.detach().numpy(), recon_chroma=chroma.detach().numpy(),
        recon_bass=bass.detach().numpy())
    if with_grads:
        losses[0].backward()
        for n, p in m.named_parameters():
            res['grad.' + n] = p.grad.detach().numpy().copy()
    return res


def slim(res, keep_full=False):
    """Full-config cases: keep 64-element slices + checksums of the big outputs, grad norms."""
    out = OrderedDict()
    for k, v in res.items():
        if k.startswith('grad.'):
            out['gnorm.' + k[5:]] = np.float64(np.sqrt((v.astype(np.float64) ** 2).sum()))
            out['gsum.' + k[5:]] = np.float64(v.astype(np.float64).sum())
        elif k in ('pitch_outs', 'dur_outs') and not keep_full:
            flat = v.reshape(-1)
            idx = np.linspace(0, flat.size - 1, 256).astype(np.int64)
            out[k + '.idx'] = idx
            out[k + '.val'] = flat[idx]
            out[k + '.sum'] = np.float64(flat.astype(np.float64).sum())
            out[k + '.abssum'] = np.float64(np.abs(flat.astype(np.float64)).sum())
        else:
            out[k] = v
    return out


def main():
    ref_model, ref_ptvae, ref_tp = import_reference()
    from amc_dl.torch_plus.train_utils import scheduled_sampling, kl_anealing

    # ---- 1. reduced configuration: everything stored ----------------------------------------
    m = build_reduced(ref_model, ref_ptvae)
    sd = OrderedDict((k, v.detach().numpy().copy()) for k, v in m.state_dict().items())
    np.savez_compressed(os.path.join(HERE, 'reduced_state.npz'), **sd)
    for name, tfr, seed in (('tf1', (1., 1., 1.), 7), ('tf0', (0., 0., 0.), 8),
                            ('tfh', (0.5, 0.5, 0.5), 9)):
        res = run_case(m, 3, 100 + seed, seed, tfr)
        np.savez_compressed(os.path.join(HERE, 'reduced_%s.npz' % name), **res)
        print('reduced', name, res['losses'][:4])

    # reduced: free-running inference_decode with its argmax trace
    torch.manual_seed(21)
    z_chd, z_rhy = torch.randn(3, 16), torch.randn(3, 16)
    est_x = m.inference_decode(z_chd, z_rhy)
    np.savez_compressed(os.path.join(HERE, 'reduced_infer.npz'), z_chd=z_chd.numpy(),
                        z_rhy=z_rhy.numpy(), est_x=est_x)

    # reduced: 3-step training trace (Adam lr 1e-3, clip 1, MinExponentialLR), tfr=1
    m = build_reduced(ref_model, ref_ptvae)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sched = ref_tp.MinExponentialLR(opt, gamma=0.9999, minimum=1e-5)
    trace = OrderedDict()
    for step in range(3):
        x, c, pr = synth_batch(3, 200 + step)
        xt, ct, prt = torch.from_numpy(x), torch.from_numpy(c), torch.from_numpy(pr)
        opt.zero_grad()
        torch.manual_seed(300 + step)
        with EpsRecorder() as er, CoinRecorder(300 + step):
            losses = m('train', xt, ct, prt, tfr1=1., tfr2=1., tfr3=1., beta=0.1, weights=[1, 0.5])
        losses[0].backward()
        gn = torch.nn.utils.clip_grad_norm_(m.parameters(), 1)
        opt.step()
        sched.step()
        trace['eps_chd.%d' % step] = er.eps[0].numpy()
        trace['eps_rhy.%d' % step] = er.eps[1].numpy()
        trace['losses.%d' % step] = np.array([l.item() for l in losses])
        trace['gnorm.%d' % step] = np.float64(gn.item())
        trace['lr.%d' % step] = np.float64(opt.param_groups[0]['lr'])
        trace['psum.%d' % step] = np.array([p.detach().double().sum().item()
                                            for p in m.parameters()])
        trace['pabs.%d' % step] = np.array([p.detach().double().abs().sum().item()
                                            for p in m.parameters()])
    np.savez_compressed(os.path.join(HERE, 'reduced_train3.npz'), **trace)
    print('train3 losses', [trace['losses.%d' % s][0] for s in range(3)])

    # ---- 2. full init_model() configuration, deterministic filler weights --------------------
    torch.manual_seed(0)
    mf = ref_model.DisentangleVAE.init_model(torch.device('cpu'))
    mf.decoder.device = torch.device('cpu')
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in mf.state_dict().items())
    mf.load_state_dict(fill_state_dict(shapes, seed=1234))
    np.savez_compressed(os.path.join(HERE, 'full_shapes.npz'),
                        names=np.array(list(shapes.keys())),
                        shapes=np.array([str(s) for s in shapes.values()]))
    for name, B, tfr, seed in (('tf1_b4', 4, (1., 1., 1.), 11), ('tf1_b16', 16, (1., 1., 1.), 12),
                               ('tf0_b4', 4, (0., 0., 0.), 13)):
        res = run_case(mf, B, 500 + seed, seed, tfr)
        out = slim(res)
        del out['x'], out['c'], out['pr_mat']          # regenerated from (B, data_seed)
        out['B'] = np.int64(B)
        out['data_seed'] = np.int64(500 + seed)
        np.savez_compressed(os.path.join(HERE, 'full_%s.npz' % name), **out)
        print('full', name, res['losses'])

    # SURVEY §8(c) anchor: default-init full model, synth_batch(4, seed=1), seeds (0, 7)
    torch.manual_seed(0)
    ma = ref_model.DisentangleVAE.init_model(torch.device('cpu'))
    ma.decoder.device = torch.device('cpu')
    x, c, pr = synth_batch(4, 1)
    torch.manual_seed(7)
    random.seed(7)
    anchor = ma.loss(torch.from_numpy(x), torch.from_numpy(c), torch.from_numpy(pr),
                     1., 1., 1., 0.1, (1, 0.5))
    print('anchor', [a.item() for a in anchor])

    # default initialisation under torch.manual_seed(0): per-tensor checksums (init-order parity)
    np.savez_compressed(os.path.join(HERE, 'full_init.npz'),
                        names=np.array([k for k, _ in ma.state_dict().items()]),
                        psum=np.array([v.double().sum().item() for v in ma.state_dict().values()]),
                        pabs=np.array([v.double().abs().sum().item() for v in ma.state_dict().values()]))

    # ---- 3. schedules ----------------------------------------------------------
    import warnings
    warnings.simplefilter('ignore')
    steps = np.array([0, 1, 2, 3, 4, 70, 71, 72, 73, 74])
    tf1 = ref_tp.TeacherForcingScheduler(0.6, 0)
    tf2 = ref_tp.TeacherForcingScheduler(0.5, 0)
    bsch = ref_tp.TeacherForcingScheduler(0.1, 0., f=kl_anealing)
    rows = []
    for i in range(75):
        rows.append([tf1.step(), tf2.step(), bsch.step()])
    rows = np.array(rows, dtype=np.float64)
    lin = torch.nn.Linear(2, 2)
    opt = torch.optim.SGD(lin.parameters(), lr=1e-3)
    sched = ref_tp.MinExponentialLR(opt, gamma=0.9999, minimum=1e-5)
    lrs = []
    for i in range(5):
        opt.step()
        sched.step()
        lrs.append(opt.param_groups[0]['lr'])
    np.savez_compressed(os.path.join(HERE, 'schedules.npz'), steps=steps, table=rows[steps],
                        lrs=np.array(lrs, dtype=np.float64),
                        anchor=np.array([a.item() for a in anchor], dtype=np.float64))
    print('schedules', rows[:3], lrs[:3])


if __name__ == '__main__':
    main()
