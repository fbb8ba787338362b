"""GPU parity of the SURVEY.md section 8(f) rows: the device data contract (f2), PtvaeEncoder (f3), the inference / demo
family (f1), the weighted duration loss and optimiser-state checkpointing (f4) -- against fixtures generated from the
reference (tests/golden/make_golden_r2.py) and the CPU oracle."""
import numpy as np
import pytest
import torch

from helpers import load_npz, reduced_params
from polyphonic_chord_texture_disentanglement_amd import model as M
from polyphonic_chord_texture_disentanglement_amd.synthetic import fill_state_dict, synth_batch, synth_raw_bank
from test_host_surface import build_reduced

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


# ---------------------------------------------------------------------------------------------- f2
def test_device_batch_transform_bit_exact_vs_reference_fixture():
    from polyphonic_chord_texture_disentanglement_amd.dataset_loaders import batch_transform
    g = load_npz('data_contract.npz')
    pr, chord, shift = (torch.from_numpy(g[k]).to(DEV) for k in ('pr', 'chord', 'shift'))
    pr_mat, x, c = batch_transform(pr, chord, shift, check=True)
    assert np.array_equal(pr_mat.cpu().numpy(), g['pr_mat'])
    assert np.array_equal(x.cpu().numpy(), g['x'])
    assert np.array_equal(c.cpu().numpy(), g['c'])
    assert x.dtype == torch.int64 and pr_mat.dtype == torch.float32 and c.dtype == torch.float32
    # gather form: sample b takes item index[b]
    index = torch.tensor([5, 0, 23, 5], device=DEV, dtype=torch.int32)
    pm2, x2, c2 = batch_transform(pr, chord, shift[index.long()], index)
    assert np.array_equal(x2.cpu().numpy(), g['x'][[5, 0, 23, 5]]) and np.array_equal(c2.cpu().numpy(), g['c'][[5, 0, 23, 5]])
    # more than 14 onsets in a step: the reference raises IndexError
    bad = torch.zeros(1, 32, 128, dtype=torch.uint8, device=DEV)
    bad[0, 3, 10:25] = 2
    with pytest.raises(IndexError):
        batch_transform(bad, chord[:1], None, check=True)


def test_device_batch_transform_large_batch_vs_oracle_and_properties():
    """B = 4096 items x random shifts against the numpy oracle (bit-exact), plus size-independent properties: transposing by
    s then reading the grid equals shifting the un-transposed grid's pitches by s (no wrap in 36..95 +- 6); chords stay one-hot"""
    from oracle import data_oracle as do
    from polyphonic_chord_texture_disentanglement_amd.dataset_loaders import batch_transform
    pr, chord = synth_raw_bank(512, 9)
    rng = np.random.RandomState(1)
    index = rng.randint(0, 512, 4096).astype(np.int32)
    shift = rng.randint(-6, 6, 4096).astype(np.int32)
    d = lambda a: torch.from_numpy(a).to(DEV)
    pr_mat, x, c = batch_transform(d(pr), d(chord), d(shift), d(index), check=True)
    sub = rng.choice(4096, 64, replace=False)
    pm_o, x_o, c_o = do.batch_transform(pr[index[sub]], chord[index[sub]], shift[sub])
    assert np.array_equal(pr_mat.cpu().numpy()[sub], pm_o) and np.array_equal(x.cpu().numpy()[sub], x_o)
    assert np.array_equal(c.cpu().numpy()[sub], c_o)
    pm0, x0, _ = batch_transform(d(pr), d(chord), None, d(index))
    xs, x0 = x.cpu().numpy(), x0.cpu().numpy()
    note = x0[..., 0] < 128
    assert np.array_equal(xs[..., 0][note], (x0[..., 0] + shift[:, None, None])[note])          # pitches move by the shift
    assert np.array_equal(xs[..., 1:], x0[..., 1:])                                              # durations do not
    assert (pr_mat.sum((1, 2)).cpu().numpy() == pm0.sum((1, 2)).cpu().numpy()).all()
    cc = c.cpu().numpy()
    assert (cc[:, :, :12].sum(-1) == 1).all() and (cc[:, :, 24:].sum(-1) == 1).all()


def test_device_batcher_serves_an_epoch_of_every_item_and_shift():
    from polyphonic_chord_texture_disentanglement_amd.dataset_loaders import DeviceBatcher, MusicDataLoaders, TrainingVAE
    pr, chord = synth_raw_bank(20, 4)
    db = DeviceBatcher(pr, chord, 64, seed=1, device=DEV)
    assert len(db) == (20 * 12 + 63) // 64
    total, seen = 0, set()
    for _, _, pr_mat, x, c, _ in db:
        assert x.shape[1:] == (32, 16, 6) and pr_mat.shape[1:] == (32, 128) and c.shape[1:] == (8, 36)
        total += x.shape[0]
        seen.update(np.round(pr_mat.sum((1, 2)).cpu().numpy(), 3).tolist())
    assert total == 240 and len(seen) >= 15
    loaders = MusicDataLoaders.get_loaders(3345, 16, 16, device_bank=(pr, chord))
    assert len(loaders.val_loader) == 1 and len(loaders.train_loader) == (18 * 12 + 15) // 16


# ---------------------------------------------------------------------------------------------- f3
@pytest.mark.parametrize('tag', ['reduced', 'full'])
def test_ptvae_encoder_vs_reference_golden(tag):
    from polyphonic_chord_texture_disentanglement_amd.ptvae import PtvaeEncoder
    g = load_npz('ptvae_encoder_%s.npz' % tag)
    kw = dict(note_emb_size=20, enc_notes_hid_size=12, enc_time_hid_size=16, z_size=8) if tag == 'reduced' else {}
    enc = PtvaeEncoder(torch.device(DEV), **kw)
    shapes = {str(n): tuple(int(t) for t in s.strip('()').split(',') if t.strip()) for n, s in zip(g['names'], g['shapes'])}
    assert list(enc.state_dict().keys()) == list(shapes.keys())
    enc.load_state_dict(fill_state_dict(shapes, 4321))
    enc.to(DEV)
    dist, emb, lengths = enc(torch.from_numpy(g['x']).to(DEV))
    assert emb.shape == (3, 32, 16, enc.note_emb_size) and lengths.shape == (3, 32)
    np.testing.assert_allclose(dist.mean.detach().cpu().numpy(), g['mean'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(dist.scale.detach().cpu().numpy(), g['scale'], rtol=0, atol=2e-5)
    assert np.array_equal(lengths.cpu().numpy(), g['lengths'])
    ((dist.mean * torch.from_numpy(g['w1']).to(DEV)).sum() + (dist.scale * torch.from_numpy(g['w2']).to(DEV)).sum()).backward()
    for k, p in enc.named_parameters():
        if tag == 'reduced':
            ref = g['grad.' + k]
            np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=0, atol=2e-6 + 2e-4 * np.abs(ref).max(), err_msg=k)
        else:
            gn, ref = float(p.grad.double().pow(2).sum().sqrt()), float(g['gnorm.' + k])
            assert abs(gn - ref) <= 1e-5 + 2e-3 * ref, (k, gn, ref)
    mu, sd, e2 = enc(torch.from_numpy(g['x']).to(DEV), return_iterators=True)
    assert torch.equal(mu, dist.mean.detach()) or (mu - dist.mean).abs().max() < 1e-6


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
@pytest.mark.parametrize('tag', ['train32', 'small', 'steps24'])
def test_ptvae_encoder_on_other_grid_geometries_vs_reference_golden(tag, prec):
    """ptvae.py:127-147 accepts any grid geometry; `train32` is train.py:32's own PtvaeEncoder(z_size=256, max_pitch=39-8) at full
    widths (the persistent / row kernels engage), `small` a 32 x 12 x (34+4) grid, `steps24` a 24 x 10 grid through
    encoder(multihot, lengths).  Expected values: the reference (tests/golden/make_golden_r4.py geom)."""
    from polyphonic_chord_texture_disentanglement_amd.ptvae import PtvaeEncoder
    from test_oracle_vs_golden import GEOM_CASES
    g = load_npz('ptvae_encoder_geom.npz')
    kw = GEOM_CASES[tag]
    enc = PtvaeEncoder(torch.device(DEV), **kw)
    shapes = {str(n): tuple(int(t) for t in s.strip('()').split(',') if t.strip()) for n, s in zip(g[tag + '.names'], g[tag + '.shapes'])}
    assert list(enc.state_dict().keys()) == list(shapes.keys())
    enc.load_state_dict(fill_state_dict(shapes, 977))
    enc.to(DEV)
    enc.precision = prec
    x = torch.from_numpy(g[tag + '.x']).to(DEV)
    B, S, N = x.shape[:3]
    tol = 2e-5 if prec == 'fp32' else 2e-2
    if tag == 'steps24':
        ref_len = torch.from_numpy(g[tag + '.lengths']).to(DEV)
        assert torch.equal(enc.get_len_index_tensor(x), ref_len)
        mh = enc.index_tensor_to_multihot_tensor(x)
        assert np.array_equal(mh.cpu().numpy(), g[tag + '.multihot'])
        dist, emb = enc.encoder(torch.from_numpy(g[tag + '.multihot']).to(DEV), ref_len)
        lengths = ref_len
    else:
        dist, emb, lengths = enc(x)
    assert emb.shape == (B, S, N, enc.note_emb_size) and lengths.shape == (B, S)
    assert np.array_equal(lengths.cpu().numpy(), g[tag + '.lengths'])
    np.testing.assert_allclose(dist.mean.detach().cpu().numpy(), g[tag + '.mean'], rtol=0, atol=tol)
    np.testing.assert_allclose(dist.scale.detach().cpu().numpy(), g[tag + '.scale'], rtol=0, atol=tol)
    if tag == 'train32':
        np.testing.assert_allclose(emb.detach().reshape(-1)[::997].cpu().numpy(), g[tag + '.embedded.slice'], rtol=0, atol=tol)
    else:
        np.testing.assert_allclose(emb.detach().cpu().numpy(), g[tag + '.embedded'], rtol=0, atol=tol)
    ((dist.mean * torch.from_numpy(g[tag + '.w1']).to(DEV)).sum() + (dist.scale * torch.from_numpy(g[tag + '.w2']).to(DEV)).sum()).backward()
    gtol = 2e-4 if prec == 'fp32' else 4e-2
    for k, p in enc.named_parameters():
        if tag == 'train32':
            gn, ref = float(p.grad.double().pow(2).sum().sqrt()), float(g['%s.gnorm.%s' % (tag, k)])
            assert abs(gn - ref) <= 1e-5 + (2e-3 if prec == 'fp32' else 3e-2) * ref, (k, gn, ref)
            idx, val, gmax = g['%s.gslice.%s.idx' % (tag, k)], g['%s.gslice.%s.val' % (tag, k)], float(g['%s.gmax.%s' % (tag, k)])
            got = p.grad.reshape(-1)[torch.from_numpy(idx).to(DEV)].cpu().numpy()
            np.testing.assert_allclose(got, val, rtol=0, atol=2e-6 + gtol * gmax, err_msg=k)
        else:
            ref = g['%s.grad.%s' % (tag, k)]
            np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=0, atol=2e-6 + gtol * np.abs(ref).max(), err_msg=k)


# ---------------------------------------------------------------------------------------------- f1
def _eps(g, key):
    return lambda name, shape, device: torch.from_numpy(g['%s.eps_%s' % (key, name)]).to(device)


def test_inference_family_vs_reference_golden():
    g = load_npz('reduced_family.npz')
    m = build_reduced(DEV).to(DEV)
    t = lambda k: torch.from_numpy(g[k]).to(DEV)
    pr1, c1, pr2, c2 = t('pr1'), t('c1'), t('pr2'), t('c2')

    def same(est, ref, tol=0.999):
        assert est.shape == ref.shape and est.dtype == np.int64
        assert (est == ref).mean() >= tol, (est == ref).mean()

    same(m.inference(pr1, c1, sample=False), g['inference_mean'])
    for fr, fc, key in ((True, True, 'swap_tt'), (True, False, 'swap_tf'), (False, True, 'swap_ft'), (False, False, 'swap_ff')):
        same(m.swap(pr1, pr2, c1, c2, fr, fc), g[key])
    m.eps_source = _eps(g, 'inference_sample')
    same(m.inference(pr1, c1, sample=True), g['inference_sample'])
    same(m.posterior_sample(pr1, c1), g['inference_sample'])                   # scale None + both sampled == inference(sample=True)
    m.eps_source = _eps(g, 'posterior_scaled')
    same(m.posterior_sample(pr1, c1, scale=0.5, sample_chd=True, sample_txt=False), g['posterior_scaled'])
    m.eps_source = _eps(g, 'prior_chd')
    same(m.prior_sample(pr1, c1, sample_chd=True, sample_rhy=False, scale=0.7), g['prior_chd'])
    m.eps_source = None
    same(m.interp(pr1, c1, pr2, c2, interp_chd=True, interp_rhy=False, int_count=5), g['interp_chd'], 0.998)
    same(m.interp(pr1, c1, pr2, c2, interp_chd=True, interp_rhy=True, int_count=4), g['interp_both'], 0.998)
    zs = m.interp_z(t('z_chd1'), t('z_chd2'), 5)
    np.testing.assert_allclose(zs.cpu().numpy(), g['interp_z_chd'], rtol=0, atol=2e-5)
    path = m.interp_path(g['z_chd1'][0], g['z_chd2'][0], 7)
    np.testing.assert_allclose(path.cpu().numpy(), g['interp_path'], rtol=0, atol=2e-5)
    assert np.array_equal(m.gt_sample(t('x1')), g['gt_sample'])
    # MIDI-free note extraction: the predicted grid -> piano-roll and (pitch, start, end) tuples
    grid = g['inference_mean'][0]
    pr, notes = m.decoder.grid_to_pr_and_notes(grid, bpm=60., start=0.)
    assert pr.shape == (32, 128) and all(len(n) == 3 and n[2] > n[1] for n in notes)
    assert len(notes) == int((pr > 0).sum()) or len(notes) >= int((pr > 0).sum())
    x1 = g['x1'][0]
    pr_gt, notes_gt = m.decoder.grid_to_pr_and_notes(x1)
    assert np.array_equal((pr_gt > 0), (g['pr1'][0] > 0))                         # ground-truth grid -> its own piano-roll
    assert len(m.decoder.pr_to_notes(g['pr1'][0])) == int((g['pr1'][0] >= 1).sum())


# ---------------------------------------------------------------------------------------------- f4
def test_weighted_duration_loss_vs_reference_golden():
    g = load_npz('reduced_wdur.npz')
    m = build_reduced(DEV).to(DEV)
    m.eps_source = lambda name, shape, device: torch.from_numpy(g['eps_' + name]).to(device)
    x, c, pr = (torch.from_numpy(g[k]).to(DEV) for k in ('x', 'c', 'pr_mat'))
    m.zero_grad()
    outs = m.run(x, c, pr, 1., 1., 1.)
    losses = m.loss_function(x, c, *outs, 0.1, [1, 0.5], weighted_dur=True)
    np.testing.assert_allclose(np.array([l.item() for l in losses]), g['losses'], rtol=0, atol=1e-5)
    losses[0].backward()
    for k, p in m.named_parameters():
        ref = g['grad.' + k]
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=0, atol=2e-6 + 2e-4 * np.abs(ref).max(), err_msg=k)
    r = m.decoder.recon_loss(x, outs[0].detach(), outs[1].detach(), weights=(1, 0.5), weighted_dur=True)
    np.testing.assert_allclose(np.array([v.item() for v in r]), g['losses'][1:4], rtol=0, atol=1e-5)


def test_optimizer_and_trainer_state_checkpoint_resumes_bit_identically(tmp_path, monkeypatch):
    """the reference saves weights only (module.py:179-183).  Here a checkpoint also carries Adam moments + step
    count, the LR scheduler and the parameter schedulers' counters: train 2 steps, save, train 2 more; a fresh trainer restored
    from the checkpoint reproduces those 2 steps exactly."""
    from polyphonic_chord_texture_disentanglement_amd.amc_dl import torch_plus as tp
    from polyphonic_chord_texture_disentanglement_amd.amc_dl.torch_plus.train_utils import kl_anealing
    from polyphonic_chord_texture_disentanglement_amd.dataset_loaders import MusicDataLoaders, TrainingVAE
    from polyphonic_chord_texture_disentanglement_amd.optim import FusedClipAdam
    monkeypatch.chdir(tmp_path)

    def make():
        m = build_reduced(DEV).to(DEV)
        m.eps_source = lambda name, shape, device: torch.zeros(shape, device=device)
        opt = FusedClipAdam(m.parameters(), lr=1e-3)
        osch = tp.OptimizerScheduler(opt, tp.MinExponentialLR(opt, gamma=0.9, minimum=1e-5), 1)
        ps = tp.ParameterScheduler(tfr1=tp.ConstantScheduler(1.), tfr2=tp.ConstantScheduler(1.), tfr3=tp.ConstantScheduler(1.),
                                   beta=tp.TeacherForcingScheduler(0.1, 0., f=kl_anealing), weights=tp.ConstantScheduler([1, 0.5]))
        loaders = MusicDataLoaders.get_loaders(11, bs_train=3, bs_val=3, n_train_batch=2, n_val_batch=1)
        pm = tp.LogPathManager(None)
        sw = tp.SummaryWriters(M.LOSS_NAMES, {'loss': None}, pm.writer_path)
        return TrainingVAE(torch.device(DEV), m, False, pm, loaders, sw, osch, ps, 1), m, opt

    tr, m, opt = make()
    tr.train()                                                  # 2 optimisation steps
    ck = str(tmp_path / 'ckpt.pt')
    tr.save_checkpoint(ck)
    before = [p.detach().clone() for p in m.parameters()]
    lr_saved, step_saved = opt.param_groups[0]['lr'], opt.step_count
    loss_a = tr.train()
    after_a = [p.detach().clone() for p in m.parameters()]

    tr2, m2, opt2 = make()
    tr2.load_checkpoint(ck)
    assert opt2.step_count == step_saved == 2 and abs(opt2.param_groups[0]['lr'] - lr_saved) < 1e-15
    assert tr2.train_step == 2 and tr2.param_scheduler.schedulers['beta']._step == 2 if hasattr(tr2.param_scheduler, 'schedulers') else True
    for a, b in zip(before, m2.parameters()):
        assert torch.equal(a, b.detach())
    assert torch.equal(opt.exp_avg.new_tensor(0).expand(0), opt.exp_avg.new_tensor(0).expand(0))
    loss_b = tr2.train()
    for a, b in zip(after_a, m2.parameters()):
        assert (a - b.detach()).abs().max() <= 1e-7
    assert abs(loss_a['loss'] - loss_b['loss']) < 1e-5
    # a plain FusedClipAdam.state_dict() round trip keeps the moments
    sd = opt2.state_dict()
    assert sd['exp_avg'].numel() == sum(p.numel() for p in m2.parameters()) and sd['step_count'] == 4


def test_checkpoint_restores_noise_coin_and_loader_streams(tmp_path, monkeypatch):
    """what decides the NEXT batches and noise is part of the checkpoint -- the model's Philox key + draw counter,
    python's `random` state (teacher-forcing coins), the device loader's generator -- and run() continues from the restored
    epoch / step counters.  A trainer restored from the checkpoint sees the same batches, eps and coins as the one that went on."""
    import random
    from polyphonic_chord_texture_disentanglement_amd.amc_dl import torch_plus as tp
    from polyphonic_chord_texture_disentanglement_amd.dataset_loaders import MusicDataLoaders, TrainingVAE
    from polyphonic_chord_texture_disentanglement_amd.optim import FusedClipAdam
    from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_raw_bank
    monkeypatch.chdir(tmp_path)
    bank = synth_raw_bank(12, 5)

    class Spy(TrainingVAE):
        seen = None

        def _batch_to_inputs(self, batch):
            inputs = super()._batch_to_inputs(batch)
            self.seen.append((inputs[0].sum().item(), self.model._draws, random.getstate()[1][:4]))
            return inputs

    def make():
        m = build_reduced(DEV).to(DEV)
        m.use_philox(seed=9, sample_offset=0)
        opt = FusedClipAdam(m.parameters(), lr=1e-3)
        osch = tp.OptimizerScheduler(opt, tp.MinExponentialLR(opt, gamma=0.9, minimum=1e-5), 1)
        ps = tp.ParameterScheduler(tfr1=tp.ConstantScheduler(.5), tfr2=tp.ConstantScheduler(.5), tfr3=tp.ConstantScheduler(.5),
                                   beta=tp.ConstantScheduler(0.1), weights=tp.ConstantScheduler([1, 0.5]))
        loaders = MusicDataLoaders.get_loaders(11, bs_train=16, bs_val=4, device_bank=bank)
        pm = tp.LogPathManager(None)
        sw = tp.SummaryWriters(M.LOSS_NAMES, {'loss': None}, pm.writer_path)
        t = Spy(torch.device(DEV), m, False, pm, loaders, sw, osch, ps, 1)
        t.seen = []
        return t, m

    random.seed(123)
    tr, m = make()
    tr.run()                                                    # one epoch (train + eval), counters advance
    assert tr.epoch == 1 and tr.train_step > 0
    ck = str(tmp_path / 'ck.pt')
    tr.save_checkpoint(ck)
    steps_saved = (tr.epoch, tr.train_step, tr.val_step)
    tr.seen = []
    loss_a = tr.train()
    seen_a = list(tr.seen)

    random.seed(999)                                            # a fresh process would start somewhere else entirely
    tr2, m2 = make()
    tr2.load_checkpoint(ck)
    assert (tr2.epoch, tr2.train_step, tr2.val_step) == steps_saved and m2._draws == seen_a[0][1] and m2._philox == (9, 0)
    loss_b = tr2.train()
    assert [s[0] for s in tr2.seen] == [s[0] for s in seen_a]       # same batches (loader generator restored)
    assert [s[1] for s in tr2.seen] == [s[1] for s in seen_a]       # same Philox draw numbers
    assert [s[2] for s in tr2.seen] == [s[2] for s in seen_a]       # same coin stream
    assert abs(loss_a['loss'] - loss_b['loss']) <= 1e-4 * max(1.0, abs(loss_a['loss']))
    # run() after a restore continues the counters instead of resetting them to the reference's default 0
    n_train = tr2.train_step
    tr2.n_epoch = 0
    tr2.run()
    assert tr2.train_step == n_train and tr2.epoch == steps_saved[0]
    tr2.run(start_epoch=0, start_train_step=0, start_val_step=0)
    assert (tr2.epoch, tr2.train_step, tr2.val_step) == (0, 0, 0)


def test_posterior_sample_draws_both_latents_in_reference_order():
    """model.py:150-172: get_zs_from_dists([dist_chd, dist_rhy], True) always draws chd then rhy; the unsampled latent is then
    overridden by its mean.  So (a) the draw counter / generator advances by two whatever the flags say and (b) the sampled
    latent sees the noise it would see with both flags on (VERDICT r2 missing #5; the f1 golden injects eps by name and cannot
    see the order)."""
    g = load_npz('reduced_family.npz')
    m = build_reduced(DEV).to(DEV)
    pr1, c1 = torch.from_numpy(g['pr1']).to(DEV), torch.from_numpy(g['c1']).to(DEV)
    seen = []
    real = m._rsample

    def spy(name, dist):
        z = real(name, dist)
        seen.append((name, m._draws, z.clone()))
        return z
    m._rsample = spy
    m.use_philox(seed=3, sample_offset=0)
    m.posterior_sample(pr1, c1, scale=0.5, sample_chd=True, sample_txt=True)
    both = list(seen)
    for flags in ((True, False), (False, True)):
        seen.clear()
        m.use_philox(seed=3, sample_offset=0)
        m.posterior_sample(pr1, c1, scale=0.5, sample_chd=flags[0], sample_txt=flags[1])
        assert [s[0] for s in seen] == ['chd', 'rhy'] and m._draws == 2
        for a, b in zip(seen, both):
            # same draws in the same order as with both flags on (the encoders' means carry run-to-run rounding: not bitwise)
            assert (a[2] - b[2]).abs().max() < 1e-5
    # torch's device generator (the default eps source, as the reference's global generator): rhy-only sampling must consume the chd
    # draw first
    m._philox = None
    seen.clear()
    torch.manual_seed(5)
    m.posterior_sample(pr1, c1, scale=0.5, sample_chd=False, sample_txt=True)
    z_rhy = seen[1][2]
    torch.manual_seed(5)
    e_chd = torch.randn(z_rhy.shape, device=DEV)
    e_rhy = torch.randn(z_rhy.shape, device=DEV)
    dist_chd, dist_rhy = m.inference_encode(pr1, c1)
    want = dist_rhy.mean + 0.5 * dist_rhy.scale * e_rhy
    assert (z_rhy - want).abs().max() < 1e-6 and (z_rhy - (dist_rhy.mean + 0.5 * dist_rhy.scale * e_chd)).abs().max() > 1e-3


# ---------------------------------------------------------------------------------------------- a4 / a9-a11 / f3 method surface
def test_reference_helper_methods_vs_reference_golden(monkeypatch):
    """the reference's helper METHODS (ptvae.py:292-428, 190-206), called directly as reference-side code would, against what the
    reference's own methods returned on the reduced model (tests/golden/make_golden_r4.py `methods`)"""
    import random as _random
    from helpers import CoinList, reduced_params
    from polyphonic_chord_texture_disentanglement_amd.ptvae import PtvaeEncoder
    from test_host_surface import build_reduced
    g = load_npz('reduced_methods.npz')
    m = build_reduced(DEV)
    m.load_state_dict(reduced_params())
    m.to(DEV)
    dec = m.decoder
    x = torch.from_numpy(g['x']).to(DEV)
    lengths = dec.get_len_index_tensor(x)
    assert lengths.dtype == torch.int64 and np.array_equal(lengths.cpu().numpy(), g['lengths'])
    mh = dec.index_tensor_to_multihot_tensor(x)
    assert mh.shape == (3, 32, 16, 135) and np.array_equal(mh.cpu().numpy(), g['multihot'])
    assert np.array_equal(dec.get_sos_token().cpu().numpy(), g['sos'])
    assert np.array_equal(dec.dur_ind_to_dur_token(torch.from_numpy(g['dur_inds1']).to(DEV), 3).cpu().numpy(), g['dur_token'])
    tok = dec.pitch_dur_ind_to_note_token(torch.from_numpy(g['pitch_inds']).to(DEV), torch.from_numpy(g['dur_inds']).to(DEV).float(), 3)
    np.testing.assert_allclose(tok.cpu().numpy(), g['note_token'], rtol=0, atol=2e-6)
    ep, ed = dec.decode_note(torch.from_numpy(g['note_summary']).to(DEV), 3)
    np.testing.assert_allclose(ep.cpu().numpy(), g['decode_note.pitch'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(ed.cpu().numpy(), g['decode_note.durs'], rtol=0, atol=2e-5)
    ns, notes = torch.from_numpy(g['notes_summary']).to(DEV), torch.from_numpy(g['notes']).to(DEV)
    seq = CoinList(g['decode_notes.coins'])
    monkeypatch.setattr(_random, 'random', seq)
    po, do, pn, ln = dec.decode_notes(ns, 3, notes, False, 0.5)
    assert seq.i == 14
    for got, key in ((po, 'pitch'), (do, 'durs'), (pn, 'predicted'), (ln, 'lengths')):
        np.testing.assert_allclose(got.cpu().numpy(), g['decode_notes.' + key], rtol=0, atol=2e-5, err_msg=key)
    monkeypatch.setattr(_random, 'random', CoinList([0.5] * 14))          # (the reference draws its 14 coins in inference mode too)
    po, do, pn, ln = dec.decode_notes(ns, 3, None, True, 0.)
    for got, key in ((po, 'pitch'), (do, 'durs'), (pn, 'predicted'), (ln, 'lengths')):
        np.testing.assert_allclose(got.cpu().numpy(), g['decode_notes_inf.' + key], rtol=0, atol=2e-5, err_msg=key)
    monkeypatch.undo()
    with pytest.raises(AssertionError):
        dec.decode_notes(ns, 3, notes, True, 0.)                 # ptvae.py:377-378
    ge = load_npz('ptvae_encoder_reduced.npz')
    enc = PtvaeEncoder(torch.device(DEV), note_emb_size=20, enc_notes_hid_size=12, enc_time_hid_size=16, z_size=8)
    shapes = {str(n): tuple(int(t) for t in s.strip('()').split(',') if t.strip()) for n, s in zip(ge['names'], ge['shapes'])}
    enc.load_state_dict(fill_state_dict(shapes, 4321))
    enc.to(DEV)
    dist, emb = enc.encoder(enc.index_tensor_to_multihot_tensor(x), enc.get_len_index_tensor(x))
    np.testing.assert_allclose(dist.mean.detach().cpu().numpy(), g['enc.mean'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(dist.scale.detach().cpu().numpy(), g['enc.scale'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(emb.detach().cpu().numpy(), g['enc.embedded'], rtol=0, atol=2e-5)
    dist.mean.sum().backward()                                   # the entry point is differentiable like the reference's
    assert enc.note_embedding.weight.grad is not None and float(enc.note_embedding.weight.grad.abs().sum()) > 0


def test_new_entry_points_reject_what_they_cannot_run():
    """error behaviour of round 4's entry points: argument errors come back as status codes before anything is launched, never as a
    launch with wild sizes -- the geometry-general embedding (dur_width > 8, row stride too short), the composites (shape outside the
    specialised kernels: PTV_ERR_UNSUPPORTED = -3, missing table slots: PTV_ERR_ARG = -1), the duration BPTT without its gate tables"""
    import ctypes
    from polyphonic_chord_texture_disentanglement_amd._lib import lib, ptr, stream_ptr, header_enum
    L = lib()
    x = torch.zeros(2, 4, 3, 10, device=DEV, dtype=torch.int64)
    w, b = torch.zeros(8, 43, device=DEV), torch.zeros(8, device=DEV)
    emb = torch.zeros(3, 4, 2, 8, device=DEV)
    st = stream_ptr()
    assert L.ptv_embed_fwd_geom(ptr(x), ptr(w), ptr(b), ptr(emb), None, 2, 8, 4, 3, 34, 9, 34, st) == -1        # dur_width 9 > 8
    assert L.ptv_embed_fwd_geom(ptr(x), ptr(w), ptr(b), ptr(emb), None, 0, 8, 4, 3, 34, 5, 34, st) == -1        # empty batch
    assert L.ptv_multihot_geom(ptr(x), ptr(emb), 10, 2, 4, 3, 34, 5, 0, st) == -1                                 # ld < P + D
    D, T = header_enum('PtvDtfDim'), header_enum('PtvDtfTensor')
    dims = [0] * D['PTV_DTF_D_COUNT']
    for k, v in (('B', 4), ('E', 20), ('HE', 12), ('HT', 40), ('HN', 28), ('HD', 8), ('NP', 130), ('ZS', 16), ('ZI', 8), ('LDP', 136)):
        dims[D['PTV_DTF_D_' + k]] = v
    darr = (ctypes.c_long * len(dims))(*dims)
    assert L.ptv_decoder_tf_supported(darr) == 0                                                                # reduced widths: not this kernel set
    slots = (ctypes.c_void_p * T['PTV_DTF_COUNT'])()
    assert L.ptv_decoder_tf_fwd(slots, darr, st) == -3
    for k, v in (('E', 128), ('HE', 128), ('HT', 1024), ('HN', 512), ('HD', 64), ('ZS', 512), ('ZI', 256), ('B', 512)):
        dims[D['PTV_DTF_D_' + k]] = v
    darr = (ctypes.c_long * len(dims))(*dims)
    assert L.ptv_decoder_tf_supported(darr) == 1
    assert L.ptv_decoder_tf_fwd(slots, darr, st) == -1                                                          # supported shape, empty table
    CT, CD = header_enum('PtvCdfTensor'), header_enum('PtvCdfDim')
    assert L.ptv_chord_decoder_fwd((ctypes.c_void_p * CT['PTV_CDF_COUNT'])(), (ctypes.c_long * CD['PTV_CDF_D_COUNT'])(), st) == -1
    M, H = 64, 64
    z = torch.zeros(6 * M * H, device=DEV)
    zi = torch.zeros(5 * M, device=DEV, dtype=torch.int32)
    part = torch.zeros(L.ptv_dur_gru_bwd_part_size(), device=DEV)
    assert L.ptv_dur_gru_bwd(H, M, None, M * H, 4 * M * H, ptr(z), M * H, 0, ptr(z), 10, ptr(z), ptr(z), ptr(zi), M, ptr(z), ptr(part), 1,
                             None, None, None, st) == -1                                                         # recompute mode needs b_hh / tab0 / tab
    torch.cuda.synchronize()
