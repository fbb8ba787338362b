"""CPU, world_size 2 over gloo: the data-parallel exchange (dist.GradSync) reproduces the reference's
DataParallel semantics -- loss = mean of the replicas' scalars, gradient = mean of the replicas'
gradients (amc_dl/torch_plus/module.py:67-68,152-159).  The per-rank work is done by
the CPU oracle here (the HIP model needs a GPU); what is under test is the N>1 host logic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import reduced_params
from oracle.ptvae_oracle import Oracle
from polyphonic_chord_texture_disentanglement_amd.dist import GradSync
from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch


class _Holder(torch.nn.Module):
    def __init__(self, params):
        super().__init__()
        self.ps = torch.nn.ParameterList([torch.nn.Parameter(p.detach().clone()) for p in params.values()])


def _shard_loss_and_grads(rank, B_local):
    params = reduced_params(requires_grad=True)
    x, c, pr = synth_batch(2 * B_local, 77)
    sl = slice(rank * B_local, (rank + 1) * B_local)
    gen = torch.Generator().manual_seed(5)
    eps = torch.randn(2, 2 * B_local, 16, generator=gen)          # keyed by GLOBAL sample index
    losses = Oracle(params).loss(torch.from_numpy(x[sl]), torch.from_numpy(c[sl]), torch.from_numpy(pr[sl]),
                                 1., 1., 1., 0.1, [1, 0.5], eps[0, sl], eps[1, sl], lambda: 0.0)
    losses[0].backward()
    return params, losses


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    params, losses = _shard_loss_and_grads(rank, 2)
    holder = _Holder(params)
    for hp, p in zip(holder.ps, params.values()):
        hp.grad = p.grad.clone()
    sync = GradSync(holder)
    assert sync.world == world
    sync.all_reduce_grads()
    mean_losses = sync.mean_scalars(losses)
    if rank == 0:
        torch.save({'grads': [p.grad.clone() for p in holder.ps], 'losses': [float(l) for l in mean_losses]}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_grad_sync_world2_matches_mean_of_shards(tmp_path):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out = str(tmp_path / 'rank0.pt')
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    shards = [_shard_loss_and_grads(r, 2) for r in range(2)]
    want_losses = np.mean([[float(l) for l in s[1]] for s in shards], axis=0)
    np.testing.assert_allclose(got['losses'], want_losses, rtol=0, atol=1e-6)
    for i, g in enumerate(got['grads']):
        want = (list(shards[0][0].values())[i].grad + list(shards[1][0].values())[i].grad) / 2
        np.testing.assert_allclose(g.numpy(), want.numpy(), rtol=0, atol=1e-7)


def test_single_process_is_a_no_op():
    p = torch.nn.Linear(3, 2)
    p.weight.grad = torch.ones_like(p.weight)
    s = GradSync(p)
    assert s.world == 1
    s.all_reduce_grads()
    assert torch.equal(p.weight.grad, torch.ones_like(p.weight))
    l = (torch.tensor(1.0), torch.tensor(2.0))
    assert s.mean_scalars(l) is l


# ---------------------------------------------------------------------------------------------
# the PRODUCT branch of GradSync: gradients already live in the optimiser's flat arena (optim.GradArena), the bucket
# that is all-reduced IS arena.flat, and the 1/world factor is handed to the optimiser as grad_scale (dist.py:39-47).
# On CPU the arena is real and the optimiser is a stand-in that applies grad_scale the way ptv_clip_adam_step does.
# ---------------------------------------------------------------------------------------------
class _ArenaOpt:
    def __init__(self, params):
        from polyphonic_chord_texture_disentanglement_amd.optim import GradArena
        self.arena = GradArena(params)
        self.grad_scale = 1.0
        self.flat_p = None

    def mark_dirty(self):
        pass


def _fake_shard(rank):
    """the model's parameter list with per-rank pseudo-gradients (the exchange logic is under test here, not the model)"""
    params = reduced_params(requires_grad=True)
    g = torch.Generator().manual_seed(100 + rank)
    for p in params.values():
        p.grad = torch.randn(p.shape, generator=g)
    return params


def _arena_worker(rank, world, port, out, early=False):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), PTV_DP_CHECK='1')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    params = _fake_shard(rank)
    holder = _Holder(params)
    if rank == 1:                                           # replicas that start apart: GradSync must broadcast rank 0's weights
        with torch.no_grad():
            for hp in holder.ps:
                hp.add_(1.0)
    opt = _ArenaOpt(list(holder.ps))
    sync = GradSync(holder, opt)
    for hp, p in zip(holder.ps, params.values()):
        assert torch.equal(hp.detach(), p.detach())          # equal to rank 0's (unperturbed) weights again
    ps, gs = list(holder.ps), list(params.values())
    k0, k1 = 2, 2 + len(ps) // 2                             # a slice from the middle of the bucket is complete first ...

    def write(idx):                                          # what the backward kernels do: write into the arena view
        for i in idx:
            v = opt.arena.take(ps[i])
            v.copy_(gs[i].grad)
            ps[i].grad = v
    write(range(k0, k1))
    if early:                                                # ... and leaves while the rest is still being produced (dist.py)
        # contract of the early exchange (dist.py docstring): parts nobody consumed are waited for and dropped by zero()
        assert sync.grads_ready(ps[k0:k1]) and len(sync._early) > 0
        opt.arena.zero()
        assert sync._early == []
        for i in range(k0, k1):
            ps[i].grad = None
        write(range(k0, k1))
        assert sync.grads_ready(ps[k0:k1])
        assert not sync.grads_ready(ps[k0:k1])               # not twice
        with pytest.raises(RuntimeError, match='second gradient contribution'):
            opt.arena.take(ps[k0])                           # a shared weight's second use while its slice is on the wire
        assert opt.arena.take(ps[0]) is not None and opt.arena.take(ps[0]) is None     # not on the wire: plain "already handed out"
        opt.arena._handed.discard(id(ps[0]))
    write([i for i in range(len(ps)) if not k0 <= i < k1])
    assert opt.arena.holds_all_grads()
    sync.all_reduce_grads()
    assert sync._early == []
    assert opt.grad_scale == 1.0 / world and opt.arena.holds_all_grads()
    if rank == 0:
        torch.save({'flat': opt.arena.flat.clone(), 'offsets': opt.arena.offsets, 'scale': opt.grad_scale}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_range_helpers():
    from polyphonic_chord_texture_disentanglement_amd.dist import complement_ranges, merge_ranges
    assert merge_ranges([(8, 16), (0, 8), (24, 32), (30, 40), (5, 5)]) == [(0, 16), (24, 40)]
    assert complement_ranges([(8, 16), (24, 40)], 48) == [(0, 8), (16, 24), (40, 48)]
    assert complement_ranges([], 10) == [(0, 10)] and complement_ranges([(0, 10)], 10) == []


@pytest.mark.parametrize('early', [False, True])
def test_grad_sync_arena_branch_world2(tmp_path, early):
    """early: a slice of the bucket starts its all-reduce before the rest of the gradients exist (GradSync.grads_ready, what
    functional.GRAD_READY_HOOK calls when the decoder's backward node is done); the result is the same SUM"""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out = str(tmp_path / 'arena0.pt')
    mp.spawn(_arena_worker, args=(2, port, out, early), nprocs=2, join=True)
    got = torch.load(out)
    shards = [_fake_shard(r) for r in range(2)]
    for i, off in enumerate(got['offsets']):
        g0, g1 = (list(s.values())[i].grad for s in shards)
        seg = got['flat'][off:off + g0.numel()].view_as(g0) * got['scale']          # SUM in the bucket, 1/world in the optimiser
        np.testing.assert_allclose(seg.numpy(), ((g0 + g1) / 2).numpy(), rtol=0, atol=1e-7)


def test_philox_oracle_known_answers_and_sharding_invariance():
    """oracle/rng_oracle.py against the Random123 known-answer vectors of philox4x32-10, and the property the DDP design
    needs: eps of global rows [a, b) does not depend on how [0, N) is cut into shards"""
    from oracle.rng_oracle import philox4x32_10, philox_normal
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for c, k, want in kat:
        got = philox4x32_10(np.array(c, dtype=np.uint32), np.array(k, dtype=np.uint32))
        assert tuple(int(v) for v in got) == want
    whole = philox_normal(64, 30, 7, 5)
    parts = np.concatenate([philox_normal(24, 30, 7, 5, 0), philox_normal(40, 30, 7, 5, 24)])
    assert np.array_equal(whole, parts)
    assert not np.array_equal(whole, philox_normal(64, 30, 7, 6))
    big = philox_normal(4096, 256, 7, 0)
    assert abs(big.mean()) < 5e-3 and abs(big.std() - 1) < 5e-3


def _run_bench(args, **env):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, **env)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        e.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    return p.returncode, [json.loads(l) for l in lines], p.stderr


def test_bench_launches_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent starts two fresh ranks (before any GPU call), they rendezvous over
    gloo, take the MAX over ranks and rank 0 prints the ONE line (PTV_BENCH_LAUNCH_ONLY=1 stops before the model, which needs the
    GPU; the full step through this launcher runs in tests/test_gpu_zz_dist.py)"""
    rc, lines, err = _run_bench(['--gpus', '2', '--steps', '1', '--warmup', '0'], PTV_DIST_BACKEND='gloo', PTV_BENCH_LAUNCH_ONLY='1')
    assert rc == 0, err[-2000:]
    assert len(lines) == 1 and lines[0]['n_gpus'] == 2 and lines[0]['launch_only'] is True
    assert lines[0]['max_over_ranks'] >= 1.0            # rank 1 contributed (its value carries +rank)


def test_bench_launcher_fails_when_a_rank_fails():
    """no GPU here: without the launch-only switch every rank dies at its first device call; the launcher must return non-zero
    and print no result line"""
    if torch.cuda.is_available():
        pytest.skip('needs a box without a GPU')
    rc, lines, err = _run_bench(['--gpus', '2', '--steps', '1', '--warmup', '0', '--no-extras', '--no-cpu-baseline'], PTV_DIST_BACKEND='gloo')
    assert rc != 0 and lines == []
