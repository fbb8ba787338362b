"""CPU: the host-side mirror keeps the reference's class surface, state_dict keys, default
initialisation and schedules.  No kernels run here (the model refuses CPU tensors)."""
import warnings

import numpy as np
import pytest
import torch

from helpers import full_shapes, load_npz
from polyphonic_chord_texture_disentanglement_amd import model as M
from polyphonic_chord_texture_disentanglement_amd import ptvae as P
from polyphonic_chord_texture_disentanglement_amd.amc_dl import torch_plus as tp
from polyphonic_chord_texture_disentanglement_amd.amc_dl.torch_plus.train_utils import kl_anealing, scheduled_sampling


def build_reduced(device='cpu'):
    torch.manual_seed(0)
    chd_enc = P.RnnEncoder(36, 32, 16)
    rhy_enc = P.TextureEncoder(24, 32, 16, 3)
    chd_dec = P.RnnDecoder(z_input_dim=16, hidden_dim=24, z_dim=16)
    dec = P.PtvaeDecoder(device=device, note_emb_size=20, z_size=32, dec_emb_hid_size=12, dec_time_hid_size=40,
                         dec_notes_hid_size=28, dec_z_in_size=16, dec_dur_hid_size=8)
    return M.DisentangleVAE('disvae', device, chd_enc, rhy_enc, dec, chd_dec)


def test_state_dict_keys_and_shapes_match_reference():
    m = M.DisentangleVAE.init_model(torch.device('cpu'))
    ref = full_shapes()
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert list(got.keys()) == list(ref.keys())
    assert got == dict(ref)
    assert sum(p.numel() for p in m.parameters()) == 27310079


def test_default_init_reproduces_reference_rng_stream():
    g = load_npz('full_init.npz')
    torch.manual_seed(0)
    m = M.DisentangleVAE.init_model(torch.device('cpu'))
    sd = m.state_dict()
    assert [str(n) for n in g['names']] == list(sd.keys())
    psum = np.array([v.double().sum().item() for v in sd.values()])
    pabs = np.array([v.double().abs().sum().item() for v in sd.values()])
    np.testing.assert_allclose(psum, g['psum'], rtol=0, atol=1e-9)
    np.testing.assert_allclose(pabs, g['pabs'], rtol=1e-12, atol=0)


def test_reduced_config_init_is_bit_identical():
    ref = load_npz('reduced_state.npz')
    sd = build_reduced().state_dict()
    assert list(sd.keys()) == list(ref.keys())
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), ref[k]), k


def test_mode_dispatch_and_aliases():
    m = build_reduced()
    assert M.PolyphonicVAE is M.DisentangleVAE
    with pytest.raises(NotImplementedError):
        m('bogus-mode')
    x = torch.zeros(2, 32, 16, 6, dtype=torch.long)
    with pytest.raises(RuntimeError, match='no CPU'):          # product path fails loudly off-GPU
        m('train', x, torch.zeros(2, 8, 36), torch.zeros(2, 32, 128), tfr1=1., tfr2=1., tfr3=1.)


def test_schedules_match_reference_tables():
    g = load_npz('schedules.npz')
    tf1 = tp.TeacherForcingScheduler(0.6, 0)
    tf2 = tp.TeacherForcingScheduler(0.5, 0)
    beta = tp.TeacherForcingScheduler(0.1, 0., f=kl_anealing)
    ps = tp.ParameterScheduler(tfr1=tf1, tfr2=tf2, beta=beta, weights=tp.ConstantScheduler([1, 0.5]))
    rows = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for _ in range(75):
            d = ps.step()
            rows.append([d['tfr1'], d['tfr2'], d['beta']])
            assert d['weights'] == [1, 0.5]
    np.testing.assert_allclose(np.array(rows)[g['steps']], g['table'], rtol=1e-12, atol=0)
    ps.eval()                                                    # frozen in eval mode (scheduler.py:10-16)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        a, b = ps.step(), ps.step()
    assert a == b
    lin = torch.nn.Linear(2, 2)
    opt = torch.optim.SGD(lin.parameters(), lr=1e-3)
    sch = tp.MinExponentialLR(opt, gamma=0.9999, minimum=1e-5)
    lrs = []
    for _ in range(5):
        opt.step()
        sch.step()
        lrs.append(opt.param_groups[0]['lr'])
    np.testing.assert_allclose(lrs, g['lrs'], rtol=1e-12)
    opt2 = torch.optim.SGD(lin.parameters(), lr=1e-3)
    sch2 = tp.MinExponentialLR(opt2, gamma=0.5, minimum=1e-5)
    for _ in range(20):
        opt2.step()
        sch2.step()
    assert opt2.param_groups[0]['lr'] == 1e-5                    # floor


def test_path_manager_and_writers(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    pm = tp.LogPathManager(None)
    assert pm.epoch_model_path('disvae').endswith('models/disvae_epoch.pt')
    assert pm.valid_model_path('disvae').endswith('disvae_valid.pt')
    assert pm.final_model_path('disvae').endswith('disvae_final.pt')
    names = ['loss', 'recon_loss']
    sw = tp.SummaryWriters(names, {'loss': None}, pm.writer_path)
    sw.write_task('train', {'loss': 1.0, 'recon_loss': 2.0}, 0)
    assert sw.all_tags['train'] == {'train_loss': (0, 1)}


def test_reference_method_names_exist_on_the_decoder_and_encoder():
    """every method of the reference's PtvaeDecoder / PtvaeEncoder (ptvae.py:125-215, 218-575) is defined under its own name"""
    for n in ('get_len_index_tensor', 'index_tensor_to_multihot_tensor', 'get_sos_token', 'dur_ind_to_dur_token',
              'pitch_dur_ind_to_note_token', 'decode_note', 'decode_notes', 'decoder', 'forward', 'recon_loss', 'emb_x',
              'output_to_numpy', 'pr_to_notes', 'grid_to_pr_and_notes'):
        assert callable(getattr(P.PtvaeDecoder, n, None)), n
    for n in ('get_len_index_tensor', 'index_tensor_to_multihot_tensor', 'encoder', 'forward'):
        assert callable(getattr(P.PtvaeEncoder, n, None)), n
    with pytest.raises((AssertionError, RuntimeError)):            # no CPU fallback: the helpers refuse host tensors too
        build_reduced().decoder.get_len_index_tensor(torch.zeros(1, 32, 16, 6, dtype=torch.long))


def test_checkpoint_keeps_each_ranks_random_state(tmp_path, monkeypatch):
    """round-3 advice: rank 0 writes the checkpoint -- a data-parallel resume must not hand rank 0's Philox sample offset / generator
    states to every rank (identical noise for different samples).  Every other rank saves its own block next to the file and restores
    it; a checkpoint without such a block (written by a single process) restores the shared parts only (seed, draw counter, coins)."""
    import random
    from polyphonic_chord_texture_disentanglement_amd.amc_dl.torch_plus.module import TrainingInterface

    class Stub:
        def state_dict(self):
            return {}

        def load_state_dict(self, sd):
            pass

    class Model(Stub):
        _philox, _draws = None, 0

    def trainer(rank, offset, draws):
        t = object.__new__(TrainingInterface)
        t.model = Model()
        t.model._philox, t.model._draws = (9, offset), draws
        t.device = torch.device('cpu')
        t.parallel = False
        t.data_loaders = None
        t.epoch = t.train_step = t.val_step = 0
        t.param_scheduler = Stub()
        t.opt_scheduler = type('O', (), {'optimizer': Stub(), 'scheduler': Stub(), '_step': 0})()
        monkeypatch.setattr(TrainingInterface, 'is_main', property(lambda self: self._r == 0))
        monkeypatch.setattr(TrainingInterface, '_rank', lambda self: self._r)
        t._r = rank
        return t

    fn = str(tmp_path / 'ck.pt')
    t0, t1 = trainer(0, 0, 5), trainer(1, 16, 5)
    random.seed(3)
    t0.save_checkpoint(fn)
    t1.save_checkpoint(fn)                                       # rank 1: only its own random-state block
    assert (tmp_path / 'ck.pt.rng1').exists()
    r1 = trainer(1, 999, 0)
    r1.load_checkpoint(fn)
    assert r1.model._philox == (9, 16) and r1.model._draws == 5   # ITS offset, not rank 0's
    r0 = trainer(0, 999, 0)
    r0.load_checkpoint(fn)
    assert r0.model._philox == (9, 0) and r0.model._draws == 5
    (tmp_path / 'ck.pt.rng1').unlink()                           # a single-process checkpoint resumed on two ranks
    r1 = trainer(1, 16, 0)
    random.seed(12345)
    r1.load_checkpoint(fn)
    assert r1.model._philox == (9, 16) and r1.model._draws == 5   # seed + draw counter shared, sample offset kept
    assert random.getstate() == torch.load(fn, weights_only=False)['rng']['python_random']      # the coin stream is the checkpoint's


def test_dead_helpers_of_the_reference_model_exist_and_compute_what_it_defines():
    """model.py:22-40 (`confuse_prmat`, `get_chroma`: defined, never called -- both call sites are commented out).  Plain tensor code:
    checked on the CPU against the reference's own expressions restated inline."""
    m = build_reduced('cpu')
    g = torch.Generator().manual_seed(1)
    pr = ((torch.rand(3, 32, 128, generator=g) < 0.05).float() * torch.randint(1, 9, (3, 32, 128), generator=g).float())
    pad = torch.zeros(3, 32, 4)
    ref = torch.log(torch.cat([pr, pad], -1).view(3, 32, -1, 12).sum(-2).view(3, 8, 4, 12).sum(-2).float() + 1)
    assert torch.equal(m.get_chroma(pr.clone()), ref) and ref.shape == (3, 8, 12)
    torch.manual_seed(5)
    out = m.confuse_prmat(pr.clone())
    torch.manual_seed(5)
    nz = torch.nonzero(pr.long())
    eps = ((2 * torch.randint(0, 2, (nz.size(0),))) - 1).long()
    exp = pr.clone()
    exp[nz[:, 0], nz[:, 1], torch.clamp(nz[:, 2] + eps, min=0, max=127)] = exp[nz[:, 0], nz[:, 1], nz[:, 2]]
    assert torch.equal(out, exp) and (out != pr).any()


def test_loss_node_zero_skip_hint_matches_only_the_very_tensors():
    """functional._loss_top_hint: the loss node's zero-skip bound must be honoured for exactly the gradient tensors it produced -- the same
    storage, shape, strides and version, whatever Python wrapper they arrive in (the engine hands them over through C++) -- and for nothing
    else: not a copy, not an in-place modified tensor, not a view with another shape.  (Round 5: an identity-based key never matched and the
    decoder silently scanned the gradients every step.)"""
    from polyphonic_chord_texture_disentanglement_amd import functional as F_
    dp, dd, top = torch.zeros(6, 8), torch.zeros(6, 10), torch.tensor([3], dtype=torch.int32)

    def arm():
        F_._LOSS_TOP.clear()
        F_._LOSS_TOP['hint'] = (dp, dd, dp._version, dd._version, top)
    arm()
    assert F_._loss_top_hint(dp.view(6, 8), dd.view(6, 10)) is top          # new wrapper objects of the same tensors: accepted
    assert F_._loss_top_hint(dp, dd) is None                                 # consumed: one use
    arm()
    assert F_._loss_top_hint(dp.clone(), dd) is None                         # a copy (autograd accumulated another contribution)
    arm()
    dp.add_(0)
    assert F_._loss_top_hint(dp, dd) is None                                 # modified in place (a tensor hook): version bumped
    arm()
    assert F_._loss_top_hint(dp.view(8, 6), dd) is None                      # same storage, other shape
    arm()
    assert F_._loss_top_hint(None, dd) is None
    F_._LOSS_TOP.clear()


def test_grad_arena_views_are_fresh_objects_over_the_bucket():
    """optim.GradArena.view: a new tensor object per call (autograd adopts a gradient only when nothing else references it) that aliases
    the parameter's 16-byte aligned range of the flat bucket with the parameter's shape"""
    from polyphonic_chord_texture_disentanglement_amd.optim import GradArena
    ps = [torch.nn.Parameter(torch.randn(3, 5)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 3, 4))]
    a = GradArena(ps)
    for i, p in enumerate(ps):
        v1, v2 = a.view(p), a.view(p)
        assert v1 is not v2 and v1.shape == p.shape and v1.is_contiguous()
        assert v1.data_ptr() == a.flat.data_ptr() + 4 * a.offsets[i] and a.offsets[i] % 8 == 0
        v1.fill_(i + 1)
        assert float(a.flat[a.offsets[i]:a.offsets[i] + p.numel()].sum()) == (i + 1) * p.numel()
    assert a.take(ps[0]) is not None and a.take(ps[0]) is None               # handed out once per zero()
    a.zero()
    assert float(a.flat.abs().sum()) == 0 and a.take(ps[0]) is not None
