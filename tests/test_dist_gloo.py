"""CPU, world_size 2 over gloo: the data-parallel exchange (dist.GradSync) reproduces the reference's
DataParallel semantics -- loss = mean of the replicas' scalars, gradient = mean of the replicas'
gradients (amc_dl/torch_plus/module.py:67-68,152-159).  The per-rank work is done by
the CPU oracle here (the HIP model needs a GPU); what is under test is the N>1 host logic."""
import os
import socket

import numpy as np
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
