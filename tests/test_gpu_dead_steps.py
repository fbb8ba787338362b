"""DisentangleVAE.loss() lets the teacher-forced decoder stop at the last note step that holds a target (functional.arm_live_top): the loss
ignores the padded note slots (ptvae.py:498-511), so the later steps' outputs are dead values there.  Checked three ways: (i) with the
unwritten rows of every forward tensor POISONED with NaN, losses and every gradient are bit-identical to the dense path -- nothing reads
them; (ii) the reference golden through loss(); (iii) run() on its own still computes every step."""
import numpy as np
import pytest
import torch

from helpers import full_params
from polyphonic_chord_texture_disentanglement_amd import functional as F_
from polyphonic_chord_texture_disentanglement_amd import model as M
from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _model():
    m = M.DisentangleVAE.init_model(torch.device(DEV))
    m.load_state_dict(full_params())
    m.to(DEV).set_precision('bf16')
    return m


def _step(m, x, c, pr, seed):
    m.use_philox(seed, 0)
    m.zero_grad()
    losses = m.loss(x, c, pr, 1., 1., 1., 0.1, [1, 0.5])
    losses[0].backward()
    torch.cuda.synchronize()
    return [l.detach().clone() for l in losses], {k: p.grad.detach().clone() for k, p in m.named_parameters()}


@pytest.mark.parametrize('B,composite', [(16, True), (16, False), (64, True)])
def test_dead_note_steps_are_never_read(B, composite, monkeypatch):
    monkeypatch.setattr(F_, 'SORT_DEC_ROWS', False)              # (rows in (t, b) order: bit-identity with the dense path; sorted rows: next test)
    x, c, pr = (torch.from_numpy(a).to(DEV) for a in synth_batch(B, 99))
    top = int(((x[..., 1:, 0] != 130) | (x[..., 1:, 1:] != 2).any(-1)).any(0).any(0).nonzero().max())
    assert top < 14                                              # (the synthetic data holds at most 8 of 16 note slots: there ARE dead steps)
    monkeypatch.setattr(F_, 'DEC_COMPOSITE', composite)
    m = _model()
    monkeypatch.setattr(F_, 'DEAD_STEPS', False)
    l0, g0 = _step(m, x, c, pr, 5)
    monkeypatch.setattr(F_, 'DEAD_STEPS', True)
    monkeypatch.setattr(F_, 'POISON_DEAD_STEPS', True)
    poisoned = []
    orig = F_._poison
    monkeypatch.setattr(F_, '_poison', lambda *t: (poisoned.append(len(t)), orig(*t)))
    l1, g1 = _step(m, x, c, pr, 5)
    assert poisoned, 'the decoder node did not take the live-step path'
    for a, b in zip(l0, l1):
        assert torch.equal(a, b), (a, b)
    for k in g0:
        assert torch.isfinite(g1[k]).all(), k
        assert torch.equal(g0[k], g1[k]), (k, (g0[k] - g1[k]).abs().max())


@pytest.mark.parametrize('B', [16, 64, 512, 48, 20, 4])       # (48, 20: note steps of 1536 / 640 rows -- segments of 3 / 5 x a power of two; 4: one block)
def test_length_sorted_rows_skip_dead_blocks_and_change_nothing(B, monkeypatch):
    """round 6, per-row dead work: inside loss() the decoder works on its rows (t, b) sorted by the number of live note steps and passes over
    the (note step, 128-row block) pairs without a target; the loss gets its targets in the same order, the gradients of the time states
    and of the fed tokens are scattered back.  With EVERY buffer the skipped blocks leave unwritten poisoned with NaN: losses equal to the
    unsorted step's to fp32 summation order (1e-6), every gradient finite and equal to it within the reordering of the K-deep sums"""
    from polyphonic_chord_texture_disentanglement_amd.optim import FusedClipAdam
    x, c, pr = (torch.from_numpy(a).to(DEV) for a in synth_batch(B, 99))
    m = _model()
    opt = FusedClipAdam(m.parameters(), lr=1e-3)                 # (the backward composite reads the optimiser's transposed weight shadows)
    res = {}
    for srt in (False, True):
        monkeypatch.setattr(F_, 'SORT_DEC_ROWS', srt)
        monkeypatch.setattr(F_, 'POISON_DEAD_STEPS', srt)
        n0 = F_._DTF.get('sorted_calls', 0)
        F_._LAST_SEG_N = None
        m.use_philox(5, 0)
        opt.zero_grad()
        losses = m.loss(x, c, pr, 1., 1., 1., 0.1, [1, 0.5])
        losses[0].backward()
        torch.cuda.synchronize()
        assert (F_._DTF.get('sorted_calls', 0) - n0 == 1) == srt    # the sorted path ran exactly when asked to
        assert (F_._LAST_SEG_N is not None) == srt                  # ... with the row segments of stage 2
        res[srt] = ([l.detach().clone() for l in losses], {k: p.grad.detach().clone() for k, p in m.named_parameters()})
    (l0, g0), (l1, g1) = res[False], res[True]
    for a, b in zip(l0, l1):
        assert abs(a.item() - b.item()) <= 2e-6 * max(1.0, abs(a.item())), (a, b)
    for k in g0:
        assert torch.isfinite(g1[k]).all(), k
        sc = max(g0[k].abs().max().item(), 1e-12)
        assert (g0[k] - g1[k]).abs().max().item() <= 2e-3 * sc, (k, (g0[k] - g1[k]).abs().max().item() / sc)
    F_.persist_check()


@pytest.mark.parametrize('switch', ['HEADS_FUSED', 'FUSED_DUR'])
def test_a_chain_that_is_not_limit_aware_end_to_end_stays_dense(switch, monkeypatch):
    """round-5 advice: with the generic heads (HEADS_FUSED off) or the per-step duration GRU (FUSED_DUR off) the launch-by-launch forward
    must not take the dead-step limit -- those stages and their weight-gradient sums read every row.  With the poison armed, loss() then
    poisons nothing and gives finite gradients that equal the DEAD_STEPS-off run bit for bit."""
    B = 16
    x, c, pr = (torch.from_numpy(a).to(DEV) for a in synth_batch(B, 99))
    monkeypatch.setattr(F_, 'DEC_COMPOSITE', False)
    monkeypatch.setattr(F_, switch, False)
    m = _model()
    monkeypatch.setattr(F_, 'DEAD_STEPS', False)
    l0, g0 = _step(m, x, c, pr, 5)
    monkeypatch.setattr(F_, 'DEAD_STEPS', True)
    monkeypatch.setattr(F_, 'POISON_DEAD_STEPS', True)
    poisoned = []
    orig = F_._poison
    monkeypatch.setattr(F_, '_poison', lambda *t: (poisoned.append(len(t)), orig(*t)))
    l1, g1 = _step(m, x, c, pr, 5)
    assert not poisoned, 'a chain with a dense stage took the live-step limit'
    for a, b in zip(l0, l1):
        assert torch.equal(a, b), (a, b)
    for k in g0:
        assert torch.isfinite(g1[k]).all(), k
        assert torch.equal(g0[k], g1[k]), k


def test_live_steps_equal_the_last_target_and_run_stays_dense(monkeypatch):
    B = 16
    x, c, pr = (torch.from_numpy(a).to(DEV) for a in synth_batch(B, 7))
    m = _model()
    seen = {}
    orig = F_.live_top_for
    monkeypatch.setattr(F_, 'live_top_for', lambda dev: seen.setdefault('t', orig(dev)))
    m.use_philox(3, 0)
    outs = m.run(x, c, pr, 1., 1., 1.)
    assert seen['t'] is None                                      # run(): its outputs are the result -- every step computed
    assert torch.isfinite(outs[0]).all() and torch.isfinite(outs[1]).all()
    seen.clear()
    m.loss(x, c, pr, 1., 1., 1., 0.1, [1, 0.5])
    top = int(((x[..., 1:, 0] != 130) | (x[..., 1:, 1:] != 2).any(-1)).any(0).any(0).nonzero().max())
    assert seen['t'] is not None and int(seen['t'].item()) == top
    seen.clear()
    with torch.no_grad():                                         # nothing armed without a backward pass to come
        m.loss(x, c, pr, 1., 1., 1., 0.1, [1, 0.5])
    assert seen.get('t') is None
    seen.clear()
    m.loss(x, c, pr, 0., 0., 0., 0.1, [1, 0.5])                   # the free-running node takes no limit
    assert seen.get('t') is None


@pytest.mark.parametrize('B', [6, 3])
def test_batches_that_do_not_fill_whole_head_blocks_run_every_step(B):
    """a note step holds 32 B rows and the fused heads work in 128-row blocks: with B not a multiple of 4 loss() keeps the decoder dense
    (same results as run() + loss_function())"""
    x, c, pr = (torch.from_numpy(a).to(DEV) for a in synth_batch(B, 3))
    m = _model()
    m.use_philox(9, 0)
    m.zero_grad()
    l1 = m.loss(x, c, pr, 1., 1., 1., 0.1, [1, 0.5])
    l1[0].backward()
    g1 = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    m.use_philox(9, 0)
    m.zero_grad()
    l0 = m.loss_function(x, c, *m.run(x, c, pr, 1., 1., 1.), 0.1, [1, 0.5])
    l0[0].backward()
    torch.cuda.synchronize()
    for a, b in zip(l0, l1):
        assert torch.equal(a, b)
    for k, p in m.named_parameters():
        assert torch.equal(p.grad, g1[k]), k


@pytest.mark.parametrize('case', ['full_tf1_b16', 'full_tf1_b512'])
def test_loss_entry_point_vs_reference_golden(case):
    import bench
    r = bench.golden_parity('bf16', torch.device(DEV), case=case, via_loss=True)
    assert r['max_abs_dloss'] < 3e-4, r
    assert r['rel_gradnorm_err'] < 5e-3 and r['worst_tensor_gradnorm_rel_err'] < 2e-2, r
